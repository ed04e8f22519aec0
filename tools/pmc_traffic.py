#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.txt + profiles/traffic_per_launch.json from the request-size PMC passes of tools/collect_profiles.sh:
   python tools/pmc_traffic.py gpurun_out/pmc_r02c_rd.txt gpurun_out/pmc_r02c_wr.txt profiles/r02_pmc_traffic.txt"""
import collections
import json
import os
import re
import sys


def load(fn):
    d = collections.defaultdict(dict)
    for line in open(fn):
        m = re.match(r"(.*?)\s{2,}(TCC_\S+|duration_ms \(profiled pass\))\s+([\d.]+)\s+x(\d+)", line)
        if m:
            d[m.group(1).strip()][m.group(2)] = float(m.group(3))
    return d


rd, wr, out_path = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
ncell, ncrop, G = 32 * 26 * 134 * 134, 32 * 20 * 128 * 128, 32 * 26 * 134          # tools/kbench.py sizes = BASELINE configs[1], B = 32
alg = {
    # cmx_kernel<STATS, BF, FEAT, DFT, WG, SB, C2, H2>
    "void cmx_kernel<0, false, false, false, false, false, 0, false>(CmxArgs)": 4 * (2 * ncell * 64 + G * 32 * 64),
    "void cmx_kernel<1, false, false, false, false, false, 0, false>(CmxArgs)": 4 * (2 * ncell * 64 + G * 32 * 64),
    "void cmx_kernel<1, false, true, false, false, false, 0, false>(CmxArgs)": 4 * (ncell * (8 + 64) + G * 32 * 64),
    "void cmx_kernel<2, false, false, false, false, false, 0, false>(CmxArgs)": 4 * (3 * ncell * 64 + G * 32 * 64),
    "void cmx_kernel<2, false, false, false, true, false, 0, false>(CmxArgs)": 4 * (3 * ncell * 64 + G * 32 * 64),       # + the Conv3d weight gradient (wave pairs)
    "void bwd_row_kernel<64, false>(BwdRowArgs)": 4 * (4 * ncell * 64 + G * 32 * 64),
    "void bwd_row_kernel<64, true>(BwdRowArgs)": 4 * (3 * ncell * 64 + ncell * 8 + G * 32 * 64),         # layer 0 (kbench stores gs)
    # bwr_kernel<GELU, XBN, XGELU, FEAT, NOX>
    "void bwr_kernel<false, true, true, false, false>(BwrArgs)": 4 * (4 * ncell * 64 + G * 32 * 64),
    "void bwr_kernel<true, true, true, false, false>(BwrArgs)": 4 * (4 * ncell * 64 + G * 32 * 64),
    "void bwr_kernel<false, false, false, false, false>(BwrArgs)": 4 * (4 * ncell * 64 + G * 32 * 64),
    "void bwr_kernel<false, false, false, false, true>(BwrArgs)": 4 * (3 * ncell * 64 + G * 32 * 64),    # no weight gradient: x not read
    "void pjf_kernel<2, true, false>(PjfArgs)": 4 * (ncrop * 64 + ncrop * 2 + ncell * 64),
    "void pjf_kernel<2, true, true>(PjfArgs)": 4 * (ncrop * 64 + 2 * ncrop * 2 + ncell * 64),
    "void pjx_head_kernel<false, 2, false>(PjhArgs)": 4 * (ncrop * 64 + ncrop * 2),
    "void pjx_head_kernel<true, 2, false>(PjhArgs)": 4 * (ncrop * 64 + ncrop * 2 + ncrop * 128),
    "void pjx_dgrad_kernel<true>(PjxArgs)": 4 * (ncrop * 64 + ncrop * 128 + ncell * 64),
}
out = ["# HBM traffic per launch of the FNO train step's large kernels from rocprofv3 PMC request-size counters",
       "# (tools/collect_profiles.sh -> tools/pmc_run.sh over tools/kbench.py, B=32 = BASELINE configs[1]); separate --pmc passes, --kernel-trace only:",
       "#   pass rd: TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B      pass wr: TCC_EA0_WRREQ TCC_EA0_WRREQ_64B",
       "# bytes_read = 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B;  bytes_written = 64*WRREQ_64B  (FETCH_SIZE under-reports 128 B requests by 2x on gfx950)",
       "#", f"# {'kernel':52s} {'read GB':>9s} {'written GB':>11s} {'total GB':>9s} {'algorithmic GB':>15s} {'ratio':>6s}"]
tr = {}
for k, a in alg.items():
    r, w = rd.get(k), wr.get(k)
    if not r or not w:
        out.append(f"# {k}: not in this pass")
        continue
    rb = 32 * r.get("TCC_EA0_RDREQ_32B", 0) + 64 * r.get("TCC_EA0_RDREQ_64B", 0) + 128 * r.get("TCC_EA0_RDREQ_128B", 0)
    wb = 64 * w.get("TCC_EA0_WRREQ_64B", 0)
    tr[k] = rb + wb
    out.append(f"  {k[5:57]:52s} {rb / 1e9:9.4f} {wb / 1e9:11.4f} {(rb + wb) / 1e9:9.4f} {a / 1e9:15.4f} {(rb + wb) / a:6.3f}")
c1, cf, c2 = (tr.get(f"void cmx_kernel<{v}>(CmxArgs)") for v in ("1, false, false, false, false, false, 0, false", "1, false, true, false, false, false, 0, false",
                                                                   "2, false, false, false, true, false, 0, false"))
if c1 and cf and c2:
    fam = (3 * c1 + cf + 3 * c2) / 7
    out += ["#", f"# cell_mix family of one train step (3 x <1>, 1 x <1,feat>, 3 x <2,WG>): {fam / 1e9:.4f} GB per launch on average = roofline.traffic of bench.py"]
    json.dump({"source": f"{out_path} (rocprofv3 --pmc TCC_EA0_RDREQ_{{32B,64B,128B}} / TCC_EA0_WRREQ_64B passes over tools/kbench.py, B=32; family "
                         "average over the 7 launches of one train step)",
               "bytes_per_launch": {"cell_mix": fam, "bn_bwd_row": tr.get("void bwr_kernel<false, false, false, false, true>(BwrArgs)"),
                                    "head_fwd_bwd": tr.get("void pjf_kernel<2, true, true>(PjfArgs)")}},
              open(os.path.join(os.path.dirname(out_path), "traffic_per_launch.json"), "w"), indent=1)
open(out_path, "w").write("\n".join(out) + "\n")
print("\n".join(out))
