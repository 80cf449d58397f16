"""Every tracked fixture under tests/golden/ is what its committed generator writes TODAY.

`tests/golden/*.npz` claim to be "outputs of the reference's own functions, made by a committed script".  That stays true only while the
script, the oracle that records the states and the fixture move together: in round 3 `terms_g1.npz` silently stopped reproducing (the
oracle's G1 trajectory changed with the self-collision pass and only the episode-statistics fixture was regenerated).  This test re-runs
each `tools/gen_golden_*.py` into a scratch directory (`RL_GOLDEN_DIR`) and demands ZERO difference, array by array, against the tracked
files.  Needs `/root/reference` (the generators import the reference's modules): skipped where it is absent (the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the generators import the reference's own modules")

# generator -> the fixtures it writes (extra_terms reads the TRACKED terms_go2.npz for its recorded state)
GENERATORS = {
    "gen_golden_terms.py": ["terms_a1.npz", "terms_go2.npz", "terms_g1.npz", "terms_a1_handstand.npz", "terms_tita.npz", "terms_go2w.npz"],
    "gen_golden_command_levels.py": ["command_levels.npz"],
    "gen_golden_symmetry.py": ["symmetry_anymal.npz"],
    "gen_golden_extra_terms.py": ["terms_extra.npz"],
}


def differences(path_a, path_b):
    a, b = np.load(path_a, allow_pickle=True), np.load(path_b, allow_pickle=True)
    if set(a.files) != set(b.files):
        return [f"keys differ: {sorted(set(a.files) ^ set(b.files))}"]
    out = []
    for k in a.files:
        x, y = a[k], b[k]
        if x.dtype == object or y.dtype == object:
            if repr(x.tolist()) != repr(y.tolist()):
                out.append(f"{k}: object arrays differ")
        elif x.shape != y.shape or x.dtype != y.dtype:
            out.append(f"{k}: {x.dtype}{x.shape} vs {y.dtype}{y.shape}")
        elif not np.array_equal(x, y, equal_nan=x.dtype.kind == "f"):
            d = np.abs(x.astype(np.float64) - y.astype(np.float64)).max() if x.dtype.kind in "fiub" else "n/a"
            out.append(f"{k}: max abs diff {d}")
    return out


@pytest.mark.parametrize("script", sorted(GENERATORS))
def test_generator_reproduces_its_tracked_fixtures(script, tmp_path):
    env = dict(os.environ, RL_GOLDEN_DIR=str(tmp_path))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    wrote = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(".npz"))
    assert wrote == sorted(GENERATORS[script]), f"{script} wrote {wrote}"
    for name in wrote:
        diff = differences(os.path.join(str(tmp_path), name), os.path.join(GOLDEN, name))
        assert not diff, f"tests/golden/{name} is stale - `python tools/{script}` writes something else now: {diff[:6]}"


def test_every_tracked_fixture_has_a_generator():
    made = {f for fs in GENERATORS.values() for f in fs} | {"episode_stats_A1.npz", "episode_stats_G1.npz"}  # tools/gen_golden_episode_stats.py: see below
    tracked = {f for f in os.listdir(GOLDEN) if f.endswith(".npz")}
    assert tracked == made, f"fixtures without a (listed) generator: {sorted(tracked - made)}; listed but not tracked: {sorted(made - tracked)}"
    for script in list(GENERATORS) + ["gen_golden_episode_stats.py"]:
        assert os.path.isfile(os.path.join(ROOT, "tools", script))
