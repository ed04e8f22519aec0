import os, sys, torch
sys.path.insert(0, "/root/repo")
from realpdebench_amd import ops, _lib
f = dict(device="cuda", dtype=torch.float32)
def timeit(name, fn, flops, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:56s} {ms:8.4f} ms  {flops / ms / 1e9:7.2f} TF/s", flush=True)
for M, N, K in ((4096, 1024, 1024), (4096, 2048, 1024), (4096, 1024, 1280), (8192, 1024, 1024), (16384, 512, 512), (32768, 256, 256)):
    G, A = torch.randn(M, N, **f), torch.randn(M, K, **f)
    if ops.gemm_tn_split_bf16(M, N, K):
        sp = ops.gemm_tn_splits(M, N, K)
        part = torch.empty(sp, N * K + N, **f)
        tot = torch.empty(N * K + N, **f)
        timeit(f"M={M} N={N} K={K} split-bf16 sp={sp} (+reduce)", lambda: (ops.gemm_tn(G, A, part, M, N, K), ops.reduce_partials(part, sp, N * K + N, out_f32=tot)), 2 * M * N * K)
    sp32 = _lib.query("rpb_gemm_tn_splits", M, N, K, 0)
    part32 = torch.empty(sp32, N * K + N, **f)
    tot32 = torch.empty(N * K + N, **f)
    timeit(f"M={M} N={N} K={K} fp32 MFMA sp={sp32} (+reduce)",
           lambda: (_lib.call("rpb_gemm_tn", G.data_ptr(), A.data_ptr(), part32.data_ptr(), M, N, K, N, K, 0, 0, 0, 0, torch.cuda.current_stream().cuda_stream),
                    ops.reduce_partials(part32, sp32, N * K + N, out_f32=tot32)), 2 * M * N * K)
