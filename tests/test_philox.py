"""Philox4x32-10: Random123 known-answer vectors (Salmon et al., SC'11, kat_vectors) for the oracle's
numpy implementation and for the lane program's C implementation, and agreement of the two
`uniform01` conventions (24-bit mantissa, (env, counter, stream, index>>2) counter layout)."""
import ctypes

import numpy as np

from oracle import philox as px

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_oracle_philox_known_answers():
    for ctr, key, want in KAT:
        got = px.philox4x32(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert tuple(int(g) for g in got) == want


def test_lane_program_philox_known_answers(emu_lib):
    lib = ctypes.CDLL(emu_lib)
    for ctr, key, want in KAT:
        c = (ctypes.c_uint32 * 4)(*ctr)
        k = (ctypes.c_uint32 * 2)(*key)
        o = (ctypes.c_uint32 * 4)()
        lib.rl_test_philox(c, k, o)
        assert tuple(o) == want


def test_uniform_conventions_agree(emu_lib):
    lib = ctypes.CDLL(emu_lib)
    lib.rl_test_uniform01.restype = ctypes.c_float
    lib.rl_test_uniform01.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed = int(rng.integers(0, 2**63))
        env, ctr, stream, idx = (int(v) for v in rng.integers(0, 2**31, 4))
        a = lib.rl_test_uniform01(seed, env, ctr, stream, idx)
        b = float(px.uniform(seed, env, ctr, stream, idx))
        assert a == b and 0.0 <= a < 1.0
