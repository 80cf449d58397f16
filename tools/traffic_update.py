#!/usr/bin/env python
"""profiles/traffic.json entry of one (task, num_envs) from the four rocprofv3 --pmc summaries of a gpurun call (tools/rocpd_summary.py
output: <prefix>_pmc_{fetch,write,sq,wait}.txt) + the kernel-trace summary of the same call:

    python tools/traffic_update.py <key> <num_envs> <prefix> <round tag> "<what ran>"

key: the task id, or `<task id>@<num_envs>` for the large-batch legs.  The counters of the step kernel (`env_kernel<..., 0, SUB, WGW[, Spec]>`
with the most calls) are taken; raw FETCH_SIZE / WRITE_SIZE are KiB (x 1024 here), corrected by `calibration` when read (bench.py)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    """{counter: mean} of the env step kernel with the most dispatches, its call count and its average duration."""
    rows, best = {}, None
    for line in open(path):
        m = re.match(r"^(.*env_kernel<.{10,60}?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s", line)
        if m and (best is None or int(m.group(2)) > best[1]):
            best = (line[:70], int(m.group(2)), float(m.group(4)))
    assert best, path
    for line in open(path):
        if line.startswith(best[0]) and " mean " in line:
            name, val = line[70:].split()[0], float(line.split(" mean ")[1].split()[0])
            rows[name] = val
    return rows, best[1], best[2]


def main():
    key, n_envs, prefix, rnd, what = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    fetch, calls, _ = counters(prefix + "_pmc_fetch.txt")
    write, _, _ = counters(prefix + "_pmc_write.txt")
    sq, _, _ = counters(prefix + "_pmc_sq.txt")
    wait, _, _ = counters(prefix + "_pmc_wait.txt")
    waves = sq["SQ_WAVES"]
    wc = wait["SQ_WAVE_CYCLES"]
    entry = dict(
        fetch_bytes=int(fetch["FETCH_SIZE"] * 1024), write_bytes=int(write["WRITE_SIZE"] * 1024), num_envs=n_envs, round=f"{rnd} ({what})",
        source=f"profiles/{os.path.basename(prefix)}_pmc_{{fetch,write,sq,wait}}.txt ({calls} launches each)",
        sq=dict(wave_quad_cycles=round(sq["SQ_WAVE_CYCLES"] / waves), issuing_frac=round(wait["SQ_ACTIVE_INST_ANY"] / wc, 3),
                valu_frac=round(wait["SQ_ACTIVE_INST_VALU"] / wc, 3), parked_frac=round(wait["SQ_WAIT_ANY"] / wc, 3),
                issue_stall_frac=round(max(0.0, 1.0 - (wait["SQ_ACTIVE_INST_ANY"] + wait["SQ_WAIT_ANY"]) / wc), 3),
                valu_insts_per_wave=round(sq["SQ_INSTS_VALU"] / waves), salu_insts_per_wave=round(sq["SQ_INSTS_SALU"] / waves),
                lds_insts_per_wave=round(sq["SQ_INSTS_LDS"] / waves), vmem_rd_per_wave=round(sq["SQ_INSTS_VMEM_RD"] / waves),
                vmem_wr_per_wave=round(sq["SQ_INSTS_VMEM_WR"] / waves),
                source=f"profiles/{os.path.basename(prefix)}_pmc_sq.txt + _pmc_wait.txt (SQ_WAVE_CYCLES, SQ_ACTIVE_INST_ANY, SQ_ACTIVE_INST_VALU, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_INSTS_* per wavefront)"))
    path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(path))
    old = tj.get(key)
    if old:
        entry["previous"] = {k: old.get(k) for k in ("fetch_bytes", "write_bytes", "round") if k in old}
    tj[key] = entry
    json.dump(tj, open(path, "w"), indent=1)
    print(key, json.dumps({k: v for k, v in entry.items() if k != "previous"})[:600])


if __name__ == "__main__":
    main()
