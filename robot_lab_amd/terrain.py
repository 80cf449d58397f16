"""Terrain generation (host logic, numpy): the world the hot path reads.

[UPSTREAM] The reference uses IsaacLab's `TerrainImporterCfg(terrain_type="generator",
terrain_generator=ROUGH_TERRAINS_CFG)` (`VEL/velocity_env_cfg.py:34,47-66`) or a plane
(`.../unitree_a1/flat_env_cfg.py:18-19`).  The generator is not in /root/reference; this file
restates it from SURVEY.md Appendix B9.  Upstream mixes triangle meshes (stairs, boxes) with
heightfields; this simulator's world is ONE heightfield, so the mesh sub-terrains are rasterised
at `hscale` = 0.05 m (stair width 0.3 m and box grid 0.45 m are both multiples of it) and vertical
faces become one-cell ramps.

Layout: sub-terrain tile (row r = difficulty level, col c = type) is centred at
x = (r + 0.5 - rows/2) * size, y = (c + 0.5 - cols/2) * size; a flat border surrounds the grid.
Heights are sampled at grid points: h[ix, iy] at (x0 + ix*hscale, y0 + iy*hscale).
"""
from __future__ import annotations

import math

import numpy as np


def plane_env_origins(num_envs: int, spacing: float) -> np.ndarray:
    """[UPSTREAM TerrainImporter._compute_env_origins_grid] grid of env origins on a plane."""
    rows = int(np.ceil(num_envs / int(np.sqrt(num_envs))))
    cols = int(np.ceil(num_envs / rows))
    ii, jj = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    o = np.zeros((num_envs, 3), dtype=np.float32)
    o[:, 0] = -(ii.flatten()[:num_envs] - (rows - 1) / 2) * spacing
    o[:, 1] = (jj.flatten()[:num_envs] - (cols - 1) / 2) * spacing
    return o


def _tile_coords(n, hscale, size):
    c = (np.arange(n) * hscale) - size / 2
    return np.meshgrid(c, c, indexing="ij")


def _pyramid_stairs(cfg, difficulty, n, hscale, size, inverted):
    lo, hi = cfg["step_height_range"]
    h = lo + difficulty * (hi - lo)
    w, border, platform = cfg["step_width"], cfg["border_width"], cfg["platform_width"]
    num_steps = int((size - 2 * border - platform) // (2 * w))
    X, Y = _tile_coords(n, hscale, size)
    d = np.maximum(np.abs(X), np.abs(Y))
    inner = size / 2 - border
    k = np.floor((inner - d) / w + 1e-9)  # -1 outside the stairs, 0.. on the rings
    level = np.clip(k + 1, 0, num_steps + 1)
    z = level * h
    if inverted:
        z = -z
    return z, (-(num_steps + 1) * h if inverted else (num_steps + 1) * h)


def _random_grid(cfg, difficulty, n, hscale, size, rng):
    lo, hi = cfg["grid_height_range"]
    gh = lo + difficulty * (hi - lo)
    gw, platform = cfg["grid_width"], cfg["platform_width"]
    nb = int(size / gw)
    border = (size - nb * gw) / 2
    heights = rng.uniform(-gh, gh, size=(nb, nb))
    X, Y = _tile_coords(n, hscale, size)
    ix = np.floor((X + size / 2 - border) / gw).astype(int)
    iy = np.floor((Y + size / 2 - border) / gw).astype(int)
    inside = (ix >= 0) & (ix < nb) & (iy >= 0) & (iy < nb)
    z = np.where(inside, heights[np.clip(ix, 0, nb - 1), np.clip(iy, 0, nb - 1)], 0.0)
    z = np.where(np.maximum(np.abs(X), np.abs(Y)) <= platform / 2, 0.0, z)
    return z, 0.0


def _random_uniform(cfg, difficulty, n, hscale, size, rng, vscale):
    lo, hi = cfg["noise_range"]
    step, border = cfg["noise_step"], cfg["border_width"]
    ds = cfg.get("downsampled_scale") or 0.1
    levels = np.arange(lo, hi + 0.5 * step, step)
    m = int(round(size / ds)) + 1
    coarse = rng.choice(levels, size=(m, m))
    # linear interpolation from the coarse grid to our sampling
    c = np.arange(n) * hscale / ds
    i0 = np.clip(np.floor(c).astype(int), 0, m - 2)
    f = c - i0
    rows = coarse[i0] * (1 - f)[:, None] + coarse[i0 + 1] * f[:, None]
    z = rows[:, i0] * (1 - f)[None] + rows[:, i0 + 1] * f[None]
    X, Y = _tile_coords(n, hscale, size)
    z = np.where(np.maximum(np.abs(X), np.abs(Y)) > size / 2 - border, 0.0, z)
    return np.round(z / vscale) * vscale, 0.0


def _pyramid_slope(cfg, difficulty, n, hscale, size, inverted, vscale):
    lo, hi = cfg["slope_range"]
    slope = lo + difficulty * (hi - lo)
    if inverted:
        slope = -slope
    platform, border = cfg["platform_width"], cfg["border_width"]
    inner = size - 2 * border
    hmax = slope * inner / 2
    X, Y = _tile_coords(n, hscale, size)
    xx = np.clip((inner / 2 - np.abs(X)) / (inner / 2), 0.0, 1.0)
    yy = np.clip((inner / 2 - np.abs(Y)) / (inner / 2), 0.0, 1.0)
    z = hmax * xx * yy
    zpf = hmax * (1 - platform / inner) ** 2
    z = np.clip(z, min(0.0, zpf), max(0.0, zpf))
    return np.round(z / vscale) * vscale, float(np.round(zpf / vscale) * vscale)


def generate_terrain(gen: dict, seed: int = 0, hscale: float = 0.05, vscale: float = 0.005):
    """gen: the dict `cfg_compile.compile_spec` emits under "terrain_generator".
    Returns heights float32 [nx, ny], origins float32 [rows, cols, 3], x0, y0."""
    rows, cols = gen["num_rows"], gen["num_cols"]
    size, border = float(gen["size"][0]), float(gen["border_width"])
    rng = np.random.default_rng(seed)
    n = int(round(size / hscale))
    nb = int(round(border / hscale))
    nx, ny = rows * n + 2 * nb + 1, cols * n + 2 * nb + 1
    heights = np.zeros((nx, ny), dtype=np.float32)
    origins = np.zeros((rows, cols, 3), dtype=np.float32)
    x0, y0 = -(rows * size / 2 + border), -(cols * size / 2 + border)
    names = list(gen["sub_terrains"].keys())
    props = np.array([gen["sub_terrains"][k]["proportion"] for k in names], dtype=np.float64)
    props = np.cumsum(props / props.sum())
    dlo, dhi = gen.get("difficulty_range", (0.0, 1.0))
    for c in range(cols):
        sub = names[int(np.min(np.where(c / cols + 0.001 < props)[0]))]
        cfg = gen["sub_terrains"][sub]
        for r in range(rows):
            if gen.get("curriculum", True):
                difficulty = (r + rng.uniform()) / rows
            else:
                difficulty = rng.uniform()
            difficulty = dlo + (dhi - dlo) * difficulty
            kind = cfg["kind"]
            if kind == "pyramid_stairs":
                z, oz = _pyramid_stairs(cfg, difficulty, n + 1, hscale, size, False)
            elif kind == "pyramid_stairs_inv":
                z, oz = _pyramid_stairs(cfg, difficulty, n + 1, hscale, size, True)
            elif kind == "random_grid":
                z, oz = _random_grid(cfg, difficulty, n + 1, hscale, size, rng)
            elif kind == "random_uniform":
                z, oz = _random_uniform(cfg, difficulty, n + 1, hscale, size, rng, vscale)
            elif kind == "pyramid_slope":
                z, oz = _pyramid_slope(cfg, difficulty, n + 1, hscale, size, False, vscale)
            elif kind == "pyramid_slope_inv":
                z, oz = _pyramid_slope(cfg, difficulty, n + 1, hscale, size, True, vscale)
            elif kind == "plane":
                z, oz = np.zeros((n + 1, n + 1)), 0.0
            else:
                raise NotImplementedError(f"sub-terrain {kind}")
            # tiles share their boundary sample row/col; interior samples win over the (flat) edges
            heights[nb + r * n: nb + (r + 1) * n + 1, nb + c * n: nb + (c + 1) * n + 1] = z
            origins[r, c] = ((r + 0.5) * size - rows * size / 2, (c + 0.5) * size - cols * size / 2, oz)
    return heights, origins, float(x0), float(y0)
