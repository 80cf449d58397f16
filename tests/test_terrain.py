"""Property tests of the terrain generator (`robot_lab_amd/terrain.py`, product path: the world every env steps on).

The reference builds its world with IsaacLab's `TerrainImporterCfg(terrain_type="generator", terrain_generator=ROUGH_TERRAINS_CFG)`
(`VEL/velocity_env_cfg.py:34,47-66`); the generator itself is third-party and absent, so `terrain.py` restates it from SURVEY.md
Appendix B9 and CANNOT be pinned to upstream outputs ("parity unpinned" for this file).  What can be pinned are the properties the
cfg states and the hot path relies on: sub-terrain proportions by column, difficulty by row, stair geometry, origins on the
platforms, a flat border, continuous tile seams, and the numbers `rl_env_desc.terrain` derives from it."""
import numpy as np
import pytest

from robot_lab_amd.scene import TERRAIN_HSCALE, build_world, load_bundle
from robot_lab_amd.terrain import _pyramid_stairs, generate_terrain, plane_env_origins

TASK = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"


@pytest.fixture(scope="module")
def world():
    desc, extra = load_bundle(TASK)
    gen = extra["terrain_generator"]
    h, origins, x0, y0 = generate_terrain(gen, 0, TERRAIN_HSCALE)
    return gen, h, origins, x0, y0


def _tile(gen, h, r, c):
    n = int(round(gen["size"][0] / TERRAIN_HSCALE))
    nb = int(round(gen["border_width"] / TERRAIN_HSCALE))
    return h[nb + r * n: nb + (r + 1) * n + 1, nb + c * n: nb + (c + 1) * n + 1]


def test_the_compiled_cfg_is_the_reference_rough_cfg(world):
    """ROUGH_TERRAINS_CFG as SURVEY App. B9 lists it: 10 x 20 tiles of 8 m, 20 m border, six sub-terrains 0.2/0.2/0.2/0.2/0.1/0.1."""
    gen = world[0]
    assert (gen["num_rows"], gen["num_cols"], tuple(gen["size"]), gen["border_width"]) == (10, 20, (8.0, 8.0), 20.0)
    assert [(k, v["proportion"]) for k, v in gen["sub_terrains"].items()] == [
        ("pyramid_stairs", 0.2), ("pyramid_stairs_inv", 0.2), ("boxes", 0.2), ("random_rough", 0.2), ("hf_pyramid_slope", 0.1), ("hf_pyramid_slope_inv", 0.1)]
    assert gen["sub_terrains"]["pyramid_stairs"]["step_height_range"] == [0.05, 0.23]
    assert gen["sub_terrains"]["hf_pyramid_slope"]["slope_range"] == [0.0, 0.4]


def test_grid_dimensions_and_border(world):
    gen, h, origins, x0, y0 = world
    n, nb = int(round(8.0 / TERRAIN_HSCALE)), int(round(20.0 / TERRAIN_HSCALE))
    assert h.shape == (10 * n + 2 * nb + 1, 20 * n + 2 * nb + 1) and h.dtype == np.float32
    assert (x0, y0) == (-(10 * 8.0 / 2 + 20.0), -(20 * 8.0 / 2 + 20.0))
    # the border is flat ground at z = 0 all around the tile grid
    for strip in (h[:nb], h[-nb:], h[:, :nb], h[:, -nb:]):
        assert np.all(strip == 0.0)
    assert np.isfinite(h).all() and np.abs(h).max() < 5.0


def test_columns_follow_the_proportions(world):
    """Column c takes the first sub-terrain whose cumulative proportion exceeds c / num_cols [UPSTREAM curriculum layout]:
    4 + 4 + 4 + 4 + 2 + 2 columns in cfg order.  Identified by what only that kind can look like."""
    gen, h, origins, _, _ = world
    kinds = []
    for c in range(20):
        t = _tile(gen, h, 9, c)  # hardest row: the shapes are unmistakable
        centre, rim = t[80, 80], t[4, 80]
        levels = np.unique(np.round(t[80, 20:80], 4))
        if c < 8:
            kinds.append("stairs_up" if centre > 0.5 else "stairs_down" if centre < -0.5 else "?")
            assert len(levels) >= 5  # discrete steps along the line from the rim to the platform
        elif c < 12:
            kinds.append("boxes" if abs(centre) < 1e-6 and np.abs(t).max() <= 0.2 + 1e-6 and len(np.unique(np.round(t, 4))) > 20 else "?")
        elif c < 16:
            kinds.append("rough" if np.abs(t).max() <= 0.1 + 1e-6 and np.abs(t).max() >= 0.02 else "?")
        else:
            kinds.append("slope_up" if centre > 0.3 else "slope_down" if centre < -0.3 else "?")
        assert abs(rim) <= 0.25  # the rim of every tile is (near) ground level
    assert kinds == ["stairs_up"] * 4 + ["stairs_down"] * 4 + ["boxes"] * 4 + ["rough"] * 4 + ["slope_up"] * 2 + ["slope_down"] * 2


def test_stair_height_is_linear_in_difficulty():
    """step height = lo + difficulty * (hi - lo) (MeshPyramidStairsTerrainCfg.step_height_range), every riser the same."""
    cfg = dict(step_height_range=(0.05, 0.23), step_width=0.3, border_width=1.0, platform_width=3.0)
    n = int(round(8.0 / TERRAIN_HSCALE)) + 1
    for d in (0.0, 0.25, 0.5, 1.0):
        z, oz = _pyramid_stairs(cfg, d, n, TERRAIN_HSCALE, 8.0, False)
        want = 0.05 + d * 0.18
        line = z[n // 2, : n // 2 + 1]  # from the rim to the centre along x = 0
        risers = np.diff(line)
        risers = risers[risers > 1e-9]
        np.testing.assert_allclose(risers, want, rtol=1e-9)
        num_steps = int((8.0 - 2 * 1.0 - 3.0) // (2 * 0.3))
        assert len(risers) == num_steps + 1
        assert oz == pytest.approx((num_steps + 1) * want) and z[n // 2, n // 2] == pytest.approx(oz)
        zi, ozi = _pyramid_stairs(cfg, d, n, TERRAIN_HSCALE, 8.0, True)
        np.testing.assert_allclose(zi, -z)
        assert ozi == pytest.approx(-oz)
        # a tread is step_width wide: 0.3 m = 6 samples between consecutive risers
        idx = np.nonzero(np.diff(line) > 1e-9)[0]
        assert np.all(np.diff(idx) == int(round(0.3 / TERRAIN_HSCALE)))


def test_difficulty_grows_with_the_row(world):
    """curriculum=True: difficulty of row r is (r + U(0, 1)) / num_rows - monotone in r for every column."""
    gen, h, origins, _, _ = world
    for c in (0, 4, 16, 18):  # stairs up / down, slopes up / down: |platform height| is a monotone function of the difficulty
        z = np.abs(origins[:, c, 2])
        assert np.all(np.diff(z) > 0), (c, z)
    amp = np.array([np.abs(_tile(gen, h, r, 8)).max() for r in range(10)])  # boxes: heights within +-(lo + d (hi - lo))
    assert amp[-1] > amp[0] and amp[0] <= 0.05 + 0.1 * 0.15 + 1e-6 and amp[-1] <= 0.2 + 1e-6
    amp = np.array([np.abs(_tile(gen, h, r, 12)).max() for r in range(10)])  # random_rough: noise_range does not depend on the difficulty
    assert np.all(amp <= 0.1 + 1e-6) and np.all(amp >= 0.08)


def test_origins_sit_on_the_platforms(world):
    """Env origins [UPSTREAM TerrainImporter.terrain_origins] are the tile centres at the height of the centre platform: a robot
    spawned at origin + default height stands on ground, on every tile."""
    gen, h, origins, x0, y0 = world
    assert origins.shape == (10, 20, 3)
    for r in range(10):
        for c in range(20):
            ox, oy, oz = origins[r, c]
            assert (ox, oy) == ((r + 0.5) * 8.0 - 40.0, (c + 0.5) * 8.0 - 80.0)
            ix, iy = int(round((ox - x0) / TERRAIN_HSCALE)), int(round((oy - y0) / TERRAIN_HSCALE))
            patch = h[ix - 5: ix + 6, iy - 5: iy + 6]  # +-0.25 m around the centre
            if c < 8 or c >= 16:
                np.testing.assert_allclose(patch, oz, atol=1e-6)  # stairs / slopes: a flat platform at exactly the origin height
            else:
                assert abs(oz) < 1e-9 and np.abs(patch).max() <= 0.1 + 1e-6  # boxes: flat platform at 0; rough: noise <= 0.1 around 0


def test_tile_seams_are_continuous(world):
    """Neighbouring tiles share their boundary samples and every tile's rim is ground level: no cliff at a seam (a robot that
    crosses into the next tile - terrain_levels_vel moves it there - must not fall off an edge that the cfg does not have)."""
    gen, h, _, _, _ = world
    n, nb = int(round(8.0 / TERRAIN_HSCALE)), int(round(20.0 / TERRAIN_HSCALE))
    for r in range(1, 10):
        seam = h[nb + r * n - 1: nb + r * n + 2, nb: nb + 20 * n]
        assert np.abs(np.diff(seam, axis=0)).max() <= 0.25  # at most one stair riser / box edge
    for c in range(1, 20):
        seam = h[nb: nb + 10 * n, nb + c * n - 1: nb + c * n + 2]
        assert np.abs(np.diff(seam, axis=1)).max() <= 0.25


def test_generation_is_deterministic_and_seeded(world):
    gen, h, origins, _, _ = world
    h2, o2, _, _ = generate_terrain(gen, 0, TERRAIN_HSCALE)
    assert np.array_equal(h, h2) and np.array_equal(origins, o2)
    h3, _, _, _ = generate_terrain(gen, 1, TERRAIN_HSCALE)
    assert not np.array_equal(h, h3)


def test_build_world_matches_the_descriptor():
    """What rl_env_create receives: the heightfield's shape / scale / offsets in the descriptor are those of the generated array;
    env origins are not passed for a generated terrain - the kernel takes them from `terrain_origins[level][type]` (startup draws
    level <= max_init_terrain_level, velocity_env_cfg.py:51; csrc/rl_env_host.h startup())."""
    desc, extra = load_bundle(TASK)
    h, terrain_origins, env_origins = build_world(desc, extra, 64, 0)
    td = desc.terrain
    assert (td.nx, td.ny) == h.shape and td.hscale == pytest.approx(TERRAIN_HSCALE)
    assert (td.num_rows, td.num_cols, td.tile_size, td.border) == (10, 20, 8.0, 20.0)
    assert (td.x0, td.y0) == (-60.0, -100.0) and not td.is_plane
    assert np.asarray(terrain_origins).shape == (10, 20, 3) and env_origins is None
    assert h.flags["C_CONTIGUOUS"] and h.dtype == np.float32  # the kernel indexes hf[ix * ny + iy]


def test_plane_origins_grid():
    """Flat tasks: [UPSTREAM TerrainImporter._compute_env_origins_grid] - a centred grid with the cfg's env_spacing."""
    o = plane_env_origins(64, 2.5)
    assert o.shape == (64, 3) and np.all(o[:, 2] == 0)
    assert np.allclose(o.mean(axis=0), 0.0, atol=1e-6)
    assert len(np.unique(np.round(o[:, :2], 3), axis=0)) == 64
    d = np.linalg.norm(o[None, :, :2] - o[:, None, :2], axis=-1) + np.eye(64) * 1e9
    assert d.min() == pytest.approx(2.5)
