"""Descriptor bundles: (EnvDesc + terrain recipe) <-> JSON files under `robot_lab_amd/data/`.

The bundles are compiled HERE from the reference's cfg classes by `tools/compile_descriptors.py`
(which needs /root/reference) and committed, so that the GPU box - where the reference does not
exist - can build the five BASELINE.json configs without it.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .desc import EnvDesc, desc_from_json, desc_to_json
from .terrain import generate_terrain, plane_env_origins

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
TERRAIN_HSCALE = 0.05


def save_bundle(path: str, desc: EnvDesc, spec: dict):
    blob = dict(desc=json.loads(desc_to_json(desc)), terrain_generator=spec.get("terrain_generator"),
                env_spacing=spec.get("env_spacing", 2.5))
    if spec.get("dropped_contact_bodies"):  # bodies whose collision geometry the lane program cannot host (model/build.py): they never touch the ground
        blob["dropped_contact_bodies"] = list(spec["dropped_contact_bodies"])
    if spec.get("topology"):  # how model/build.py decomposed the link tree into trunk pieces + limb chains (which of its rules, and the result)
        blob["topology"] = spec["topology"]
    with open(path, "w") as f:
        json.dump(blob, f)


def load_bundle(task_id_or_path: str):
    path = task_id_or_path
    if not os.path.isfile(path):
        path = os.path.join(DATA_DIR, task_id_or_path + ".json")
    with open(path) as f:
        blob = json.load(f)
    desc = desc_from_json(json.dumps(blob["desc"]))
    return desc, dict(terrain_generator=blob.get("terrain_generator"), env_spacing=blob.get("env_spacing", 2.5))


def build_world(desc: EnvDesc, extra: dict, num_envs: int, terrain_seed: int = 0):
    """Fills desc.terrain's grid fields; returns (heights|None, terrain_origins|None, env_origins|None)."""
    if desc.terrain.is_plane:
        return None, None, plane_env_origins(num_envs, float(extra.get("env_spacing") or 2.5))
    heights, origins, x0, y0 = generate_terrain(extra["terrain_generator"], terrain_seed, TERRAIN_HSCALE)
    t = desc.terrain
    t.nx, t.ny = heights.shape
    t.hscale, t.x0, t.y0 = TERRAIN_HSCALE, x0, y0
    return np.ascontiguousarray(heights, dtype=np.float32), origins, None
