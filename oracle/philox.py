"""TEST INFRASTRUCTURE ONLY (oracle).  Philox4x32-10 counter-based RNG (Salmon et al., SC'11),
vectorised numpy.  Pinned against the Random123 known-answer vectors in tests/test_philox.py.

uniform(seed, env, counter, stream, index) = u24(philox(ctr=(env, counter, stream, index>>2),
key=(seed_lo, seed_hi))[index & 3]) with u24(x) = (x >> 8) * 2^-24, exactly representable in fp32,
so the fp64 oracle and the fp32 kernel draw identical samples.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)

# RNG streams
STREAM_RESET, STREAM_COMMAND, STREAM_PUSH, STREAM_NOISE, STREAM_STARTUP, STREAM_ACTION = 1, 2, 3, 4, 5, 6
# draw indices inside STREAM_RESET
IDX_WRENCH, IDX_JPOS, IDX_JVEL, IDX_KP, IDX_KD, IDX_POSE, IDX_VEL, IDX_CMD, IDX_CMD_TIME, IDX_PUSH_TIME, IDX_LEVEL = (
    0, 8, 40, 72, 104, 136, 142, 148, 154, 155, 156)
# draw indices inside STREAM_STARTUP
IDX_BUCKET, IDX_MASS_ADD, IDX_MASS_SCALE, IDX_COM = 0, 64, 128, 192
GLOBAL_ENV = 0xFFFFFFFF


def philox4x32(c0, c1, c2, c3, k0, k1):
    c = [np.asarray(x, dtype=np.uint32) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c


def uniform(seed, env, counter, stream, index):
    """U[0,1) with 24 bits; all arguments broadcast."""
    env, counter, stream, index = np.broadcast_arrays(
        np.asarray(env, dtype=np.uint32), np.asarray(counter, dtype=np.uint32),
        np.asarray(stream, dtype=np.uint32), np.asarray(index, dtype=np.uint32))
    out = philox4x32(env, counter, stream, index >> np.uint32(2), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    sel = index & np.uint32(3)
    x = np.where(sel == 0, out[0], np.where(sel == 1, out[1], np.where(sel == 2, out[2], out[3])))
    return (x >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)


def uniform_range(seed, env, counter, stream, index, lo, hi):
    return lo + (hi - lo) * uniform(seed, env, counter, stream, index)
