#!/usr/bin/env python
"""Do the kernel SHAPES of one lane program compile to the same floating-point arithmetic?

The step kernel of an instance is built for single- and for four-wavefront workgroups (template argument WGW): the same source, so the
same arithmetic - tests/test_gpu_canary.py holds the two to the same bits on the GPU.  Under -ffp-contract=fast that is not a law: the
back end fuses a multiply-add or not depending on what else uses the product, and its unrolling / CSE decisions can differ between the
two instantiations (round 5: one ulp in a reset env's heading target after the kernel prologue changed).  This tool compares the
MULTISET of floating-point opcodes of every such pair in the device assembly (`hipcc -save-temps`): a difference means the pair's bits
may differ by contraction round-off (the canary then falls back to its 2e-5 step-by-step comparison), equality that they very likely
do not.  __graft_entry__.build() runs it and records the result in csrc/build_info.json; it never refuses a build.

    python tools/isa_shape_arith.py kernel1.s [kernel2.s ...]
"""
import collections
import re
import sys

FLOAT_OP = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad|max|min|med3|rcp|rsq|sqrt|exp|log|sin|cos|floor|fract|trunc|rndne|div|ldexp|frexp|pk_)[a-z0-9_]*f(16|32|64)|^v_cvt_")


def kernels(path):
    """{mangled kernel name: Counter of opcodes} of the env kernels in one assembly file."""
    s = open(path).read()
    out = {}
    for m in re.finditer(r"^(_ZN12_GLOBAL__N_110env_kernel[^:\n]*):", s, flags=re.M):
        i = m.end()
        j = s.index("s_endpgm", i)
        c = collections.Counter()
        for line in s[i:j].split("\n"):
            line = line.strip()
            if line and not line.startswith((";", ".")) and not line.endswith(":"):
                c[line.split()[0]] += 1
        out[m.group(1)] = c
    return out


def mismatches(paths):
    """[(kernel name of the single-wavefront shape, {opcode: (count WGW 1, count WGW 4)})] for the pairs whose float arithmetic differs."""
    ks = {}
    for p in paths:
        ks.update(kernels(p))
    bad, pairs = [], 0
    for name, c1 in ks.items():
        m = re.search(r"(Li0ELi\d+E)Li1E", name)  # RESET = 0, SUB, WGW = 1
        if not m:
            continue
        twin = name.replace(m.group(0), m.group(1) + "Li4E")
        if twin not in ks:
            continue
        pairs += 1
        c4 = ks[twin]
        d = {k: (c1.get(k, 0), c4.get(k, 0)) for k in set(c1) | set(c4) if c1.get(k, 0) != c4.get(k, 0) and FLOAT_OP.match(k)}
        if d:
            bad.append((name, d))
    return pairs, bad


if __name__ == "__main__":
    n, bad = mismatches(sys.argv[1:])
    print(f"{n} kernel pairs (one / four wavefronts per workgroup), {len(bad)} with different floating-point opcode counts")
    for name, d in bad:
        print(" ", name[40:130], d)
