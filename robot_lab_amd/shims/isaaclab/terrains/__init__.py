"""[UPSTREAM isaaclab.terrains] cfg containers read by the terrain generator of the MI355X env."""
from isaaclab.utils.configclass import GenericCfg, configclass


@configclass
class TerrainImporterCfg:
    prim_path = "/World/ground"
    terrain_type = "generator"
    terrain_generator = None
    max_init_terrain_level = None
    collision_group = -1
    physics_material = None
    visual_material = None
    env_spacing = None
    num_envs = None
    debug_vis = False
    usd_path = None


@configclass
class TerrainGeneratorCfg:
    seed = None
    curriculum = False
    size = (8.0, 8.0)
    border_width = 0.0
    border_height = 1.0
    num_rows = 1
    num_cols = 1
    color_scheme = "none"
    horizontal_scale = 0.1
    vertical_scale = 0.005
    slope_threshold = 0.75
    sub_terrains = None
    difficulty_range = (0.0, 1.0)
    use_cache = False
    cache_dir = "/tmp/isaaclab/terrains"


@configclass
class SubTerrainBaseCfg:
    function = None
    proportion = 1.0
    size = (10.0, 10.0)
    flat_patch_sampling = None


def _sub(kind, **defaults):
    cls = configclass(type(kind, (SubTerrainBaseCfg,), dict(kind=kind, **defaults)))
    return cls


MeshPyramidStairsTerrainCfg = _sub("pyramid_stairs", border_width=0.0, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=1.0, holes=False)
MeshInvertedPyramidStairsTerrainCfg = _sub("pyramid_stairs_inv", border_width=0.0, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=1.0, holes=False)
MeshRandomGridTerrainCfg = _sub("random_grid", grid_width=0.45, grid_height_range=(0.05, 0.2), platform_width=2.0, holes=False)
HfRandomUniformTerrainCfg = _sub("random_uniform", noise_range=(0.02, 0.10), noise_step=0.02, border_width=0.25, downsampled_scale=None)
HfPyramidSlopedTerrainCfg = _sub("pyramid_slope", slope_range=(0.0, 0.4), platform_width=2.0, border_width=0.25, inverted=False)
HfInvertedPyramidSlopedTerrainCfg = _sub("pyramid_slope_inv", slope_range=(0.0, 0.4), platform_width=2.0, border_width=0.25, inverted=True)
MeshPlaneTerrainCfg = _sub("plane")


def __getattr__(name):
    if name.startswith("__") or not name[:1].isupper():
        raise AttributeError(name)
    val = type(name, (GenericCfg,), {"__module__": __name__})
    globals()[name] = val
    return val
