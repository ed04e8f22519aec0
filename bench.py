#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X FNO3d path (contract: see the task statement / DESIGN.md).

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run)

One "step" = one full training iteration of the reference's hot loop (train.py:321-334: forward, MSE mean,
backward, Adam, cosine LR) on a synthetic batch of BASELINE.json configs[1]: FNO3d cylinder-shaped
[B=32/GPU, 20, 128, 128, 2], modes (4,12,16), width 64, 4 layers, fp32.  Inputs are resident in HBM when the
timed region starts.  Rank 0 prints ONE JSON line with the whole-job samples/s, the roofline of the dominant
kernel (HIP-event timed inside the timed region) and a CPU baseline (the oracle timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3     # fp32-input MFMA = fp32 vector peak
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="trajectories per GPU (weak scaling)")
    ap.add_argument("--rollout-steps", type=int, default=10, help="N_autoregressive of configs/cylinder/fno.yaml")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="print per-kernel HIP-event table to stderr")
    ap.add_argument("--no-transolver", action="store_true", help="skip the secondary Transolver measurement")
    ap.add_argument("--no-galerkin", action="store_true", help="skip the secondary Galerkin Transformer measurement")
    ap.add_argument("--no-unet", action="store_true", help="skip the secondary U-Net measurement")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline_worker():
    """Child process: the oracle (PyTorch-CPU restatement of the reference) does two training steps at B=1."""
    from oracle import fno3d_oracle as O
    shape, modes, width, n_layers = (20, 128, 128, 2), (4, 12, 16), 64, 4
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    Bc = 1
    sd = O.init_state_dict(modes, n_layers, width, shape, shape, seed=0)
    g = torch.Generator().manual_seed(0)
    nsteps = 2                                              # ~12-15 s of CPU work on the GPU box's 16 usable cores
    batches = [(torch.randn(Bc, *shape, generator=g), torch.randn(Bc, *shape, generator=g)) for _ in range(nsteps)]
    t0 = time.time()
    O.train_steps(sd, batches, modes, n_layers, shape, shape, lr0=1e-4, t_max=4000)
    dt = time.time() - t0
    print(json.dumps({"value": nsteps * Bc / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                      "sample": f"{nsteps} train steps (fwd+bwd+Adam) of the CPU oracle at B={Bc}, same shape and model, "
                                f"{dt:.1f} s, no warm-up"}))


def cpu_baseline():
    """Bounded: runs in a child process with a hard timeout so the default bench always finishes in minutes."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True,
                           text=True, timeout=300)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:        # timeout / parse error: report it, never hang the bench
        return {"value": None, "unit": "samples/s", "cores": usable_cores(), "kind": "port",
                "sample": f"CPU oracle step did not finish within 300 s ({type(e).__name__})"}


def _flush_c_stdio():
    """RCCL prints its version banner through C stdio, which is fully buffered on a pipe and would otherwise land AFTER
    the JSON line when the process exits; push it out first so the JSON line is the last line of stdout."""
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _conv_model_roofline(achieved_tf):
    """Transolver / U-Net: 70-83 % of the FLOPs are 3x3x3 convolutions, which run on the bf16 MFMA from split fp32 operands
    (hi + mid + lo, six bf16 products per fp32 product, fp32 accumulate: csrc/rpb_conv3x.hip); the rest is fp32 MFMA.
    ``achieved`` counts algorithmic (fp32) FLOPs of the whole step; the two peaks bracket what a step can reach."""
    return {"achieved": achieved_tf, "unit": "TFLOP/s (fp32-equivalent)", "peak_f32_mfma": MFMA_F32_PEAK_TF,
            "peak_split_bf16": MFMA_BF16_PEAK_TF / 6.0, "frac_of_f32_mfma_peak": achieved_tf / MFMA_F32_PEAK_TF,
            "frac_of_split_bf16_peak": achieved_tf / (MFMA_BF16_PEAK_TF / 6.0),
            "conv_arith": "RPB_CONV3_EXACT=1" if os.environ.get("RPB_CONV3_EXACT") == "1" else "split-bf16 (fp32-grade)"}


def bench_unet(dev, steps=2):
    """U-Net at the reference's configs/cylinder/unet.yaml ([12,20,64,128,3], dim = H = 64 -> 64/128/256 channels,
    dim_mults [1,2,4]): train step through the drop-in protocol (HIP forward + taped HIP backward, torch.optim.Adam) and
    eval forward.  fp32 MFMA roofline: SURVEY.md section 8(a8) measured 555 GFLOP forward per sample at 20x64x64 with
    FlopCounterMode on the oracle; the cylinder mesh has twice the cells -> 1.11 TFLOP forward, x3 for a step."""
    import yaml
    from realpdebench_amd.model.unet import Unet3d
    with open(os.path.join(ROOT, "realpdebench_amd", "configs", "cylinder", "unet.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    T, H, W, C = cfg["shape_in"]
    B = int(cfg["train_batch_size"])
    torch.manual_seed(0)
    m = Unet3d(dim=H, out_channels=cfg["shape_out"][-1], dim_mults=cfg["dim_mults"], channels=C, in_time=T,
               out_time=cfg["shape_out"][0]).to(dev)
    x = torch.randn(B, T, H, W, C, device=dev)
    y = torch.randn(B, *cfg["shape_out"], device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"])

    def step():
        opt.zero_grad()
        m.train_loss(x, y).mean().backward()
        opt.step()

    m.train()
    step()
    step()                                  # two warm-up steps: the first one pays allocator growth and code loading
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t_train = (time.perf_counter() - t0) / steps
    m.eval()
    with torch.no_grad():
        m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m(x)
        torch.cuda.synchronize()
    t_fwd = (time.perf_counter() - t0) / steps
    flops_step = 3 * 1.11e12 * B
    del m, opt
    torch.cuda.empty_cache()
    return {"train_samples_per_s": B / t_train, "ms_per_step": 1e3 * t_train, "batch": B,
            "forward_fields_per_s": B * cfg["shape_out"][0] / t_fwd, "ms_per_forward": 1e3 * t_fwd,
            "mfma": _conv_model_roofline(flops_step / t_train / 1e12),
            "config": "configs/cylinder/unet.yaml: [12,20,64,128,3], dim 64, dim_mults [1,2,4], 4 heads x 32"}


def bench_galerkin(dev, steps=3):
    """Galerkin Transformer at the reference's configs/cylinder/galerkin_transformer.yaml (n = 20*64*128 tokens, hidden
    256, freq_dim 128, modes (4,16,20), train_batch_size 16): train step through the drop-in protocol (HIP
    forward/backward, dropout masks drawn, torch.optim.Adam) and eval forward.  Dense FLOPs per token forward:
    2*(768*256 + 2*256*256 + 128*256) Linear + 2*2*256*64 head products = 0.79 MFLOP, x3 for a step; the spectral
    regressor and the head products are HBM-bound."""
    import yaml
    from realpdebench_amd.model.galerkin_transformer import GalerkinTransformer3d
    with open(os.path.join(ROOT, "realpdebench_amd", "configs", "cylinder", "galerkin_transformer.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    T, H, W, Cin = cfg["shape_in"]
    B = int(cfg["train_batch_size"])
    cfg.update(node_feats=Cin, n_targets=cfg["shape_out"][-1])
    torch.manual_seed(0)
    m = GalerkinTransformer3d(**cfg).to(dev)
    x = torch.randn(B, T, H, W, Cin, device=dev)
    y = torch.randn(B, *cfg["shape_out"], device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"])

    def step():
        opt.zero_grad()
        m.train_loss(x, y).mean().backward()
        opt.step()

    m.train()
    step()
    step()                                  # two warm-up steps: the first one pays allocator growth and code loading
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t_train = (time.perf_counter() - t0) / steps
    m.eval()
    with torch.no_grad():
        m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m(x)
        torch.cuda.synchronize()
    t_fwd = (time.perf_counter() - t0) / steps
    tokens = B * T * H * W
    flops_step = 3 * 0.79e6 * tokens
    del m, opt
    torch.cuda.empty_cache()
    return {"train_samples_per_s": B / t_train, "ms_per_step": 1e3 * t_train, "batch": B,
            "forward_fields_per_s": B * cfg["shape_out"][0] / t_fwd, "ms_per_forward": 1e3 * t_fwd,
            "mfma_f32": {"achieved": flops_step / t_train / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": flops_step / t_train / 1e12 / MFMA_F32_PEAK_TF},
            "config": "configs/cylinder/galerkin_transformer.yaml: [16,20,64,128,3], n_hidden 256, 4 heads, freq_dim 128, "
                      "modes (4,16,20), dropout 0.05 / attention 0.5"}


def bench_transolver(dev, B=4, steps=3):
    """Transolver (configs/cylinder/trainsolver.yaml: 20x64x128x3 tokens -> mesh 128x64x20, hidden 256, 8 heads, 16
    slices, 1 layer) -- train step through the drop-in protocol (HIP forward/backward + torch.optim.Adam) and eval
    forward.  Reported next to the FNO headline; MFMA roofline (the two 3x3x3 convolutions are 83 % of the FLOPs)."""
    from realpdebench_amd.model.transolver import Transolver
    torch.manual_seed(0)
    m = Transolver(space_dim=3, n_layers=1, n_hidden=256, n_head=8, fun_dim=0, out_dim=3, slice_num=16, mlp_ratio=4,
                   H=128, W=64, D=20, dropout=0.1).to(dev)
    x = torch.randn(B, 20, 64, 128, 3, device=dev)
    y = torch.randn(B, 20, 64, 128, 3, device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=7e-4)

    def step():
        opt.zero_grad()
        m.train_loss(x, y).mean().backward()
        opt.step()

    m.train()
    step()
    step()                                  # two warm-up steps: the first one pays allocator growth and code loading
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t_train = (time.perf_counter() - t0) / steps
    m.eval()
    with torch.no_grad():
        m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m(x)
        torch.cuda.synchronize()
    t_fwd = (time.perf_counter() - t0) / steps
    tokens = B * 20 * 64 * 128
    flops_step = 3 * 8.56e6 * tokens              # 8.56 MFLOP/token forward, x3 for a step
    del m, opt
    torch.cuda.empty_cache()
    return {"train_samples_per_s": B / t_train, "ms_per_step": 1e3 * t_train, "batch": B,
            "forward_fields_per_s": B * 20 / t_fwd, "ms_per_forward": 1e3 * t_fwd,
            "mfma": _conv_model_roofline(flops_step / t_train / 1e12),
            "config": "Transolver cylinder: tokens 20x64x128 -> mesh (128,64,20), n_hidden 256, 8 heads, 16 slices, "
                      "1 layer, mlp_ratio 4, dropout 0.1, fp32"}


def main():
    a = parse()
    if a.cpu_baseline_worker:
        return cpu_baseline_worker()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback path exists)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    force_dp = os.environ.get("RPB_FORCE_DP") == "1"       # exercise the RCCL code path on a single rank
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from realpdebench_amd import _lib
    from realpdebench_amd.model.fno import FNO3d
    from realpdebench_amd.trainer import Trainer
    from realpdebench_amd.rollout import autoregressive_rollout

    shape, modes, width, L = (20, 128, 128, 2), (4, 12, 16), 64, 4
    torch.manual_seed(0)
    model = FNO3d(*modes, L, width, shape, shape).to(dev)
    if world > 1 or force_dp:
        from realpdebench_amd.dp import DataParallel
        DataParallel(model)
    trainer = Trainer(model, lr=1e-4, num_update=4000, scheduler="cosine")
    B = a.batch
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.randn(B, *shape, device=dev, generator=g)
    y = torch.randn(B, *shape, device=dev, generator=g)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up with every launch timed: find the dominant kernel
    _lib.PROFILE, _lib.PROFILE_ONLY = {}, None
    for _ in range(max(a.warmup, 1)):
        trainer.step(x, y)
    torch.cuda.synchronize()
    warm = _lib.profile_summary()
    dominant = max(warm, key=lambda k: warm[k]["total_ms"])
    if a.profile_all and rank == 0:
        tot = sum(v["total_ms"] for v in warm.values())
        for k, v in sorted(warm.items(), key=lambda kv: -kv[1]["total_ms"]):
            print(f"{k:48s} calls/step {v['calls'] / max(a.warmup, 1):5.1f}  avg {v['avg_ms']:8.3f} ms  "
                  f"{100 * v['total_ms'] / tot:5.1f}%  {v['bytes'] / v['avg_ms'] / 1e6:8.1f} GB/s  "
                  f"{v['flops'] / v['avg_ms'] / 1e9:7.2f} TF/s", file=sys.stderr)

    # ---- timed region: exactly K steps, only the dominant kernel carries HIP events
    _lib.PROFILE, _lib.PROFILE_ONLY = {}, {dominant}
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = trainer.step(x, y)
    barrier()
    dt = time.perf_counter() - t0
    dom = _lib.profile_summary()[dominant]
    _lib.PROFILE = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = 1e3 * dt / a.steps
    value = B * world * a.steps / dt

    # ---- rollout metric (eval.py:311-321), replicas: no collective
    rollout = None
    if not a.no_rollout:
        del trainer
        model._ws = {}
        torch.cuda.empty_cache()
        autoregressive_rollout(model, x, 1)
        barrier()
        t0 = time.perf_counter()
        autoregressive_rollout(model, x, a.rollout_steps)
        barrier()
        rt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([rt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rt = float(t)
        rollout = {"value": B * world * shape[0] * a.rollout_steps / rt, "unit": "fields/s",
                   "n_autoregressive": a.rollout_steps, "ms_per_forward": 1e3 * rt / a.rollout_steps}

    # ---- secondary: Transolver (north_star's second model) at the reference's cylinder config, rank 0 / N=1 only
    transolver = None
    if not a.no_transolver and world == 1:
        transolver = bench_transolver(dev)

    galerkin = None
    if not a.no_galerkin and world == 1:
        galerkin = bench_galerkin(dev)
    unet = None
    if not a.no_unet and world == 1:
        unet = bench_unet(dev)
    if world > 1 or force_dp:
        torch.cuda.synchronize()
        _flush_c_stdio()                                    # every rank empties its C stdio buffer (RCCL banner) ...
        dist.barrier()                                      # ... before rank 0 goes on to print the JSON line
        dist.destroy_process_group()

    if rank == 0:
        ach_gbs = dom["bytes"] / dom["avg_ms"] / 1e6
        ach_tf = dom["flops"] / dom["avg_ms"] / 1e9
        step_bytes = (6.238 * B + 4.03) * 1e9          # algorithmic bytes of one train step
        traffic = None                                  # HBM bytes per launch of the dominant kernel, from the committed
        try:                                            # PMC request-size passes (bench.py cannot run rocprofv3 itself)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_per_launch.json")))
            if B == 32:
                traffic = tj["bytes_per_launch"].get(dominant)
        except Exception:
            pass
        line = {
            "metric": "train-step samples/sec (+ autoregressive rollout fields/sec in 'rollout'), FNO cylinder 128^2",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "FNO3d train step (fwd+MSE+bwd+Adam+cosine), cylinder-shaped [B,20,128,128,2] "
                                   "-> padded 26x134x134, modes (4,12,16), width 64, 4 layers (BASELINE.json configs[1])",
                       "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_traffic.txt" if traffic else None,
                         "avg_launch_ms": dom["avg_ms"], "launches_timed": dom["calls"],
                         "algorithmic_bytes_per_launch": dom["bytes"],
                         "mfma_f32": {"achieved": ach_tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                      "frac": ach_tf / MFMA_F32_PEAK_TF},
                         "whole_step": {"algorithmic_bytes": step_bytes,
                                        "achieved": step_bytes / (ms_per_step * 1e6), "unit": "GB/s",
                                        "frac": step_bytes / (ms_per_step * 1e6) / HBM_PEAK_GBS}},
            "rollout": rollout,
            "transolver": transolver,
            "galerkin_transformer": galerkin,
            "unet": unet,
            "loss": float(loss),
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        _flush_c_stdio()
        print(json.dumps(line), flush=True)                 # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()
