#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output) as text:
per-kernel dispatch statistics (like `--stats`) and, if present, per-kernel PMC counter means.

    python tools/rocpd_summary.py gpurun_out/prof_r01/a1rough_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys
from collections import defaultdict


def short(name, width=70):
    """70 columns of a kernel name; the step kernels specialised on a task (csrc/env_spec.h) carry the task as their LAST template argument,
    which the cut would drop: it is appended as a tag (`... [Spec_A1_Rough]`)."""
    import re

    m = re.search(r"rl::(Spec_\w+)", name)
    if m:
        tag = " [" + m.group(1) + "]"
        return name[: width - len(tag)] + tag
    return name[:width]


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]  # noqa: E731
    names = {r[0]: r[1] for r in cur.execute(f"select id, display_name from {T('rocpd_info_kernel_symbol')}")}
    regs = {r[0]: r[1:] for r in cur.execute(
        f"select id, sgpr_count, arch_vgpr_count, accum_vgpr_count, group_segment_size, private_segment_size from {T('rocpd_info_kernel_symbol')}")}
    rows = cur.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x, group_segment_size, private_segment_size, event_id from {T('rocpd_kernel_dispatch')}").fetchall()
    stats = defaultdict(list)
    meta = {}
    ev2k = {}
    for k, s, e, g, w, lds, scr, ev in rows:
        stats[k].append(e - s)
        meta[k] = (g, w, lds, scr)
        ev2k[ev] = k
    total = sum(sum(v) for v in stats.values())
    print(f"# {path}")
    print(f"# {len(rows)} dispatches, {total / 1e6:.3f} ms total kernel time")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}  grid wg lds scratch sgpr vgpr* agpr*")
    print("# (* rocprofv3's arch_vgpr_count / accum_vgpr_count columns: NOT the allocation - the env-step kernel allocates 256 VGPR + 54-58 AGPR per llvm-readelf --notes / the .s metadata, tools/kbuild.sh)")
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        g, w, lds, scr = meta[k]
        r = regs.get(k, (0, 0, 0, 0, 0))
        print(f"{short(names.get(k, str(k))):70s} {len(v):6d} {sum(v) / 1e6:10.3f} {sum(v) / len(v) / 1e3:9.2f} {min(v) / 1e3:9.2f} {max(v) / 1e3:9.2f} "
              f"{100 * sum(v) / total:6.2f}  {g} {w} {lds} {scr} {r[0]} {r[1]} {r[2]}")
    npmc = cur.execute(f"select count(*) from {T('rocpd_pmc_event')}").fetchone()[0]
    if npmc:
        pmc_names = {r[0]: r[1] for r in cur.execute(f"select id, name from {T('rocpd_info_pmc')}")}
        acc = defaultdict(lambda: defaultdict(list))
        for ev, pid, val in cur.execute(f"select event_id, pmc_id, value from {T('rocpd_pmc_event')}"):
            if ev in ev2k:
                acc[ev2k[ev]][pmc_names.get(pid, str(pid))].append(val)
        print("\n# PMC counters: mean per dispatch (summed over instances as reported)")
        for k, d in acc.items():
            for n, v in sorted(d.items()):
                print(f"{short(names.get(k, str(k))):70s} {n:24s} mean {sum(v) / len(v):16.2f} n={len(v)}")


if __name__ == "__main__":
    main(sys.argv[1])
