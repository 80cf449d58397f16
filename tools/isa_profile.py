#!/usr/bin/env python
"""Static profile of one kernel's gfx950 assembly (hipcc -save-temps .s): instruction classes per labelled block, loops
(backward branches), and the `; PHASE <name>` markers the sources can plant with asm volatile("; PHASE x").

    python tools/isa_profile.py kernel.s [--blocks]
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def main():
    path = sys.argv[1]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    label_line = {}
    blocks = OrderedDict()
    cur = "<entry>"
    blocks[cur] = []
    phases = OrderedDict()
    phase = "<start>"
    phases[phase] = Counter()
    detail = Counter()
    for i, ln in enumerate(lines):
        s = ln.strip()
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", s)
        if m:
            cur = m.group(1)
            label_line[cur] = i
            blocks[cur] = []
            continue
        if s.startswith("; PHASE"):
            phase = s[len("; PHASE"):].strip()
            phases.setdefault(phase, Counter())
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        if op == "s_endpgm":
            blocks[cur].append((i, op, s))
            break
        blocks[cur].append((i, op, s))
        c = classify(op)
        phases[phase][c] += 1
        if c == "valu":
            if "dpp" in s or "quad_perm" in s or "row_" in s:
                detail["valu_dpp"] += 1
            if op.startswith("v_mov_b32"):
                detail["v_mov_b32"] += 1
            if op.startswith("v_pk_"):
                detail["v_pk"] += 1
            if op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
                detail["readlane"] += 1
            if op.startswith("v_cndmask"):
                detail["cndmask"] += 1
            if op.startswith("v_fma") or op.startswith("v_fmac") or op.startswith("v_mul_f32") or op.startswith("v_add_f32") or op.startswith("v_sub_f32"):
                detail["fp32_arith"] += 1
    tot = Counter()
    for c in phases.values():
        tot.update(c)
    print("total:", dict(tot))
    print("detail:", dict(detail))
    print("phases:")
    for p, c in phases.items():
        if sum(c.values()):
            print(f"  {p:28s} valu {c['valu']:6d} acc {c['accvgpr']:5d} salu {c['salu']:5d} lds {c['lds']:5d} vmem {c['vmem']:4d} wait {c['waitcnt']:4d} br {c['branch']:4d}")
    # loops: backward branches
    print("loops (backward branches):")
    for name, ins in blocks.items():
        for i, op, s in ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = s.split()[-1]
                if tgt in label_line and label_line[tgt] < i:
                    body = Counter()
                    for n2, ins2 in blocks.items():
                        for j, op2, s2 in ins2:
                            if label_line[tgt] <= j <= i:
                                body[classify(op2)] += 1
                    print(f"  {tgt} <- line {i}: body {dict(body)}")
    if show_blocks:
        for name, ins in blocks.items():
            c = Counter(classify(op) for _, op, _ in ins)
            if sum(c.values()) > 40:
                print(f"  block {name:14s} @{label_line.get(name, 0):6d}: {dict(c)}")


if __name__ == "__main__":
    main()
