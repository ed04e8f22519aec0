#!/usr/bin/env python
"""Per-kernel summary (the `--stats` view) of a rocprofv3 ``*_results.db`` (rocpd SQLite) as plain text.

  python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_xxx_kernel_stats.txt
"""
import sqlite3
import sys


def table(c, prefix):
    return [r[0] for r in c.execute("select name from sqlite_master where type='table'") if r[0].startswith(prefix)][0]


def main(path):
    c = sqlite3.connect(path)
    kd, ks = table(c, "rocpd_kernel_dispatch"), table(c, "rocpd_info_kernel_symbol")
    rows = c.execute(f"""
        select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start),
               max(d.end - d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size),
               max(d.workgroup_size_x), max(d.grid_size_x)
        from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':84s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s} "
          f"{'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'wg':>5s} {'grid':>9s}")
    for n, cnt, tot, avg, mn, mx, vg, ag, lds, wg, grid in rows:
        n = n if len(n) <= 84 else n[:81] + "..."
        print(f"{n:84s} {cnt:6d} {tot / 1e6:10.3f} {avg / 1e3:10.1f} {mn / 1e3:10.1f} {mx / 1e3:10.1f} "
              f"{100.0 * tot / total:6.2f} {vg:5d} {ag:5d} {lds:7d} {wg:5d} {grid:9d}")


if __name__ == "__main__":
    main(sys.argv[1])
