"""Static instruction budget of a kernel's hot loop, read off the gfx950 ISA hipcc emits (no GPU needed):

    python tools/isa_budget.py realpdebench_amd/csrc/rpb_cmx.hip 'cmx_kernelILi0ELb0ELb0ELb1'      # eval cell_mix + fused W stage
    python tools/isa_budget.py realpdebench_amd/csrc/rpb_pjh.hip pjh_fwd_kernelILi2

compiles the file to device assembly with the flags realpdebench_amd/build.py uses for it, finds every natural loop of the kernel
(a backward branch to an earlier label), and prints per loop the instruction mix by issue class: MFMA (by shape), packed / scalar-form
vector ALU, transcendentals, v_accvgpr moves, LDS, buffer / global memory, scalar ALU, waits.  With the issue rules measured in
DESIGN.md section 4.0000 (two waves per SIMD: T = 16 M16 + 32 M32 + 2.6 V cycles, added) this gives the cycles one 32-cell wave tile
costs -- the number DESIGN.md section 9 sets against the HBM time of the same tile.

A loop body that holds two variants of an epilogue (the masked last tile of a line next to the unmasked one) is counted whole; the
per-basic-block table (--blocks) shows which blocks are alternatives.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_to_asm(src, extra=()):
    from realpdebench_amd import build as B
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src)[:-4] + ".isa.s")
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA.get(os.path.basename(src), []) + list(extra) + ["--cuda-device-only", "-S", src, "-o", out]
    cmd = [c for c in cmd if c != "-fPIC"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr)
    return out


CLASSES = ("mfma16", "mfma32", "mfma_other", "valu", "valu_pk", "trans", "accmov", "lds", "vmem", "salu", "wait", "nop", "branch", "other")


def classify(op):
    if op.startswith("v_mfma"):
        if "16x16x32" in op:
            return "mfma16"
        if "32x32x16" in op:
            return "mfma32"
        return "mfma_other"
    if op.startswith(("v_accvgpr", )):
        return "accmov"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_lines(asm, pattern):
    lines = open(asm).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    hits = [i for i in starts if pattern in lines[i]]
    if not hits:
        names = [lines[i].split(":")[0] for i in starts]
        raise SystemExit(f"no kernel matching {pattern!r}; kernels: " + ", ".join(names))
    out = []
    for s in hits:
        e = next(i for i in range(s + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
        out.append((lines[s].split(":")[0], lines[s + 1:e]))
    return out


def loops(body):
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    found = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            found.append((label_at[m.group(1)], i, m.group(1)))
    return found


def mix(lines):
    c = dict.fromkeys(CLASSES, 0)
    for l in lines:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if m and not l.lstrip().startswith((".", ";")):
            c[classify(m.group(1))] += 1
    return c


def cycles_two_waves(c):
    """DESIGN.md 4.0000, two waves per SIMD: T = 16 M16 + 32 M32 + 2.6 V (V = every vector-issue instruction of one wave)."""
    V = c["valu"] + c["valu_pk"] + c["trans"] + c["accmov"]
    return 16 * c["mfma16"] + 32 * c["mfma32"], 2.6 * V


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--blocks", action="store_true", help="per basic block of the largest loop")
    ap.add_argument("--flag", action="append", default=[], help="extra compiler flag (e.g. -DCMX_PF2=1)")
    a = ap.parse_args()
    asm = compile_to_asm(os.path.join(ROOT, a.src) if not os.path.isabs(a.src) else a.src, a.flag)
    for name, body in kernel_lines(asm, a.kernel):
        print(f"== {name}: {sum(mix(body).values())} instructions in all")
        ls = sorted(loops(body), key=lambda t: t[0] - t[1])
        seen = []
        for s, e, lab in ls:
            if any(s >= s2 and e <= e2 for s2, e2 in seen):      # nested inside one already printed: still print, indented
                pre = "    inner "
            else:
                pre = "  "
            seen.append((s, e))
            c = mix(body[s:e + 1])
            if c["mfma16"] + c["mfma32"] + c["mfma_other"] == 0 and sum(c.values()) < 40:
                continue
            mp, vp = cycles_two_waves(c)
            nz = {k: v for k, v in c.items() if v}
            print(f"{pre}loop {lab} ({e - s + 1} lines): {nz}")
            print(f"{pre}  matrix-pipe cycles {mp}, vector-issue cycles at two waves per SIMD {vp:.0f}, sum {mp + vp:.0f}")
        if a.blocks and ls:
            s, e, _ = ls[0]
            cur, start = None, s
            for i in range(s, e + 2):
                m = re.match(r"^(\.LBB\d+_\d+):", body[i]) if i <= e else True
                if m:
                    if cur is not None:
                        c = mix(body[start:i])
                        print(f"    block {cur}: " + str({k: v for k, v in c.items() if v}))
                    if i <= e:
                        cur, start = m.group(1), i


if __name__ == "__main__":
    main()
