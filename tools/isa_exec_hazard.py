#!/usr/bin/env python
"""Static detector for the hipcc miscompile class that has hit the env-step kernel twice (profiles/r02_launch_bounds64_miscompile.txt,
profiles/r03d_pin_desc_miscompile.txt): a vector-register write that executes under a STALE (empty or narrowed) EXEC mask on the
skip path of a divergent region.

How the defect arises (LLVM AMDGPU, SILowerControlFlow + register allocation): the end of a divergent `if` restores EXEC with
`s_or_b64 exec, exec, <saved>` at the top of its "flow" block.  When an inner region ends right where an outer one ends, the inner
restore is removed as redundant (`-amdgpu-remove-redundant-endcf`, default on): the inner `if` becomes a bare
`s_and_b64 exec, exec, vcc ; s_cbranch_execz FLOW`, and FLOW is an EMPTY block in front of the outer restore.  The register
allocator runs afterwards.  When it splits a live range around the region it may put the reload (`v_accvgpr_read_b32 vX, aY`, a
scratch load, a rematerialised `v_mov`) into that empty block - where EXEC is 0 on the skip path and only the inner region's lanes
otherwise.  Lanes of the OUTER region that clobbered vX as a temporary never get the value back; after the outer restore the full
wavefront reads vX.  Nothing in the source can rule this out (which value is split depends on register pressure), which is why the
symptom moved between the push timer (round 2) and the commands / heading flags (round 3) with unrelated edits.

The check: for every region skip - an `s_cbranch_execz L` that directly follows the instruction(s) narrowing EXEC for the region
(`s_and_b64 exec, exec, vcc`, `s_and_saveexec_b64`, `s_or_saveexec_b64` + `s_xor_b64 exec`, `s_mov_b64 exec`) -, walk from L to the
first instruction that writes EXEC; any vector-register write on the way executes under the stale mask -> hazard.  (An
`s_cbranch_execz` that does NOT follow an EXEC write is something else: with EXEC = 0 nothing a block does has an effect, and the
compiler routes such a wavefront through any convenient real code block - e.g. the cases of a switch it could not prove uniform.
Those targets are full of vector writes and harmless.)  `__graft_entry__.build()` runs it on the device assembly of the env library (built with
`-mllvm -amdgpu-remove-redundant-endcf=false`, which removes the empty-flow-block shape altogether) and refuses a build with a hit.

    python tools/isa_exec_hazard.py <file.s> [...]        exit code 1 when a hazard is found"""
import re
import sys

LABEL = re.compile(r"^(\.LBB\d+_\d+|[A-Za-z_][\w.$]*):")
VWRITE = re.compile(r"^(v_(?!cmp_|cmpx_|nop|readlane|readfirstlane|writelane)|ds_read|ds_bpermute|ds_permute|ds_swizzle|ds_consume|ds_append|global_load|buffer_load|scratch_load|flat_load|"
                    r"global_atomic\w*\s+v|buffer_atomic|image_|v_accvgpr)")
EXEC_WRITE = re.compile(r"^(s_\w+\s+exec\b|s_\w*saveexec\w*\s|v_cmpx_)")
BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+|s_endpgm|s_setpc|s_swappc)")


def parse(path):
    """-> {function: [(line_no, label | None, instruction | None)]}"""
    funcs, cur, name = {}, None, None
    with open(path) as f:
        for no, raw in enumerate(f, 1):
            line = raw.split(";", 1)[0].strip()
            if not line:
                continue
            m = LABEL.match(line)
            if m:
                lab = m.group(1)
                if not lab.startswith(".L"):
                    name, cur = lab, []
                    funcs[name] = cur
                elif cur is not None:
                    cur.append((no, lab, None))
                continue
            if cur is None or line.startswith("."):
                continue
            cur.append((no, None, line))
    return funcs


def region_window(items, i):
    """the (up to three) instructions in front of items[i] that fall through to it: an unconditional branch ends the window - what stands
    before it belongs to another path (round 5: a region that ENDS in `s_or_b64 exec ; s_branch L`, followed by a block entered through a
    uniform `s_cbranch_vccz` only, whose first instruction is the EXEC = 0 pass-through `s_cbranch_execz`)"""
    window = []
    for _, _, x in reversed(items[max(0, i - 6):i]):
        if x and x.startswith(("s_branch", "s_endpgm", "s_setpc")):
            break
        if x and not x.startswith(("s_waitcnt", "s_nop")):
            window.append(x)
    return window[:3]


def hazards(items):
    labels = {lab: i for i, (_, lab, _) in enumerate(items) if lab}
    out = []
    for i, (no, _, ins) in enumerate(items):
        if not ins or not ins.startswith("s_cbranch_execz"):
            continue
        # a region skip: EXEC was narrowed by one of the (up to three) instructions in front of the branch
        prev = region_window(items, i)
        if not any(EXEC_WRITE.match(x) for x in prev):
            continue
        target = ins.split()[-1]
        j = labels.get(target)
        if j is None:
            continue
        for no2, _, ins2 in items[j:]:
            if ins2 is None:
                continue
            if EXEC_WRITE.match(ins2) or BRANCH.match(ins2):
                break
            if VWRITE.match(ins2):
                out.append((no, target, no2, ins2))
                break
    return out


def count_skips(items):
    n = 0
    for i, (_, _, ins) in enumerate(items):
        if ins and ins.startswith("s_cbranch_execz"):
            n += any(EXEC_WRITE.match(x) for x in region_window(items, i))
    return n


def main(paths):
    bad = 0
    for p in paths:
        funcs = parse(p)
        n_br = 0
        for name, items in funcs.items():
            n_br += count_skips(items)
            for no, target, no2, ins2 in hazards(items):
                bad += 1
                print(f"{p}:{no2}: HAZARD in {name[:90]}: '{ins2}' runs under the stale EXEC of the skip path 's_cbranch_execz {target}' (line {no})")
        print(f"{p}: {len(funcs)} functions, {n_br} execz skips checked, {bad} hazards so far")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
