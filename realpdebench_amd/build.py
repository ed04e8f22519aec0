"""Builds ``csrc/librpb_hip.so`` (the C-ABI library declared in ``include/rpb.h``) with hipcc for gfx950.

Cross-compiles without a GPU.  The ``.so`` stays in-tree (git-ignored, but shipped to the GPU box).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "librpb_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = (["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"]
         + os.environ.get("RPB_HIPCC_FLAGS", "").split())


# per-file flags: rpb_pjg.hip keeps every vector instruction in its scalar form (packed fp32 next to MFMAs waits for the matrix pipe)
# rpb_pjx.hip (the eval head rpb_proj_fwd): packed fp32 off -- measured 1.50 -> 1.37 ms (packed fp32 instructions wait for the matrix pipe;
# the same flag costs the wave-pair weight-gradient cell_mix 8 %, so rpb_cmx.hip keeps its packed forms: profiles/r05_kbench_nopk_ab.txt)
NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
EXTRA = {"rpb_pjg.hip": ["-fno-slp-vectorize"], "rpb_pjx.hip": NOPK, "rpb_pjh.hip": NOPK + ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    live = {os.path.basename(s_)[:-4] + ".o" for s_ in sources()}
    for f in os.listdir(objdir):                       # objects whose source is gone must not ship to the GPU box
        if f.endswith(".o") and f not in live:
            os.remove(os.path.join(objdir, f))
    jobs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s in ex.map(cc, jobs):
                if verbose:
                    print("compiled", os.path.basename(s), flush=True)
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print("linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
