#!/usr/bin/env python3
"""Summarise a rocprofv3 results database (rocpd sqlite) into a small text table:
per-kernel call count / average duration, and (if present) per-kernel PMC counter averages.

    python tools/prof_summary.py gpurun_out/prof/run_results.db [--all] > profiles/xyz.txt

Only kernels of this library (ct::*) are listed unless --all is given; names are shortened."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main():
    path = sys.argv[1]
    show_all = "--all" in sys.argv
    db = sqlite3.connect(path)
    cur = db.cursor()
    flt = "" if show_all else "where name like '%ct::%'"
    rows = list(cur.execute(
        f"select name, count(*), avg(duration), min(duration), max(duration), sum(duration), "
        f"max(grid_x), max(workgroup_x), max(vgpr_count), max(sgpr_count), max(lds_size) from kernels {flt} group by name order by sum(duration) desc"))
    print(f"# rocprofv3 kernel-trace summary of {path}")
    print("# full_n / full_avg_us / full_med_us: the dispatches at the kernel's LARGEST grid only (the benchmark's workload, without the small parity-gate launches)")
    print(f"{'kernel':<92} {'calls':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'grid':>9} {'wg':>5} {'vgpr':>5} {'sgpr':>5} {'lds':>6} {'full_n':>7} {'full_avg_us':>11} {'full_med_us':>11}")
    for name, n, avg, mn, mx, tot, gx, wx, vg, sg, lds in rows:
        full = sorted(d for (d,) in cur.execute("select duration from kernels where name = ? and grid_x = ?", (name, gx)))
        favg = sum(full) / len(full) / 1e3 if full else 0.0
        fmed = full[len(full) // 2] / 1e3 if full else 0.0
        print(f"{short(name):<92} {n:>6} {avg/1e3:>9.2f} {mn/1e3:>9.2f} {mx/1e3:>9.2f} {gx:>9} {wx:>5} {vg:>5} {sg:>5} {lds:>6} {len(full):>7} {favg:>11.2f} {fmed:>11.2f}")
    try:
        flt2 = "" if show_all else "where kernel_name like '%ct::%'"
        crow = list(cur.execute(
            f"select kernel_name, counter_name, count(*), avg(value) from counters_collection {flt2} group by kernel_name, counter_name order by kernel_name, counter_name"))
    except sqlite3.Error:
        crow = []
    if crow:
        # full_*: the dispatches at the kernel's LARGEST grid only (round 6: a kernel that the bench also launches on small tensors — the sparse
        # compress on a TinyLlama-shaped checkpoint — would otherwise report a meaningless mixture as its bytes per launch)
        full = {}
        try:
            for name, cname, n, avg in cur.execute(
                    f"select c.kernel_name, c.counter_name, count(*), avg(c.value) from counters_collection c join (select kernel_name as kn, max(grid_size_x) as gx "
                    f"from counters_collection group by kernel_name) m on c.kernel_name = m.kn and c.grid_size_x = m.gx {flt2.replace('kernel_name', 'c.kernel_name')} "
                    f"group by c.kernel_name, c.counter_name"):
                full[(name, cname)] = (n, avg)
        except sqlite3.Error:
            pass
        print("\n# PMC counters: average value per dispatch (full_*: at the kernel's largest grid only)")
        print(f"{'kernel':<92} {'counter':<24} {'dispatches':>10} {'avg_value':>16} {'full_n':>8} {'full_avg_value':>16}")
        for name, cname, n, avg in crow:
            fn, favg = full.get((name, cname), (n, avg))
            print(f"{short(name):<92} {cname:<24} {n:>10} {avg:>16.1f} {fn:>8} {favg:>16.1f}")


if __name__ == "__main__":
    main()
