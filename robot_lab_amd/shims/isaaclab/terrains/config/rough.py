"""[UPSTREAM isaaclab.terrains.config.rough] ROUGH_TERRAINS_CFG restated from SURVEY.md Appendix B9
(referenced at `VEL/velocity_env_cfg.py:34,50`)."""
import isaaclab.terrains as tg

ROUGH_TERRAINS_CFG = tg.TerrainGeneratorCfg(
    size=(8.0, 8.0),
    border_width=20.0,
    num_rows=10,
    num_cols=20,
    horizontal_scale=0.1,
    vertical_scale=0.005,
    slope_threshold=0.75,
    use_cache=False,
    sub_terrains={
        "pyramid_stairs": tg.MeshPyramidStairsTerrainCfg(
            proportion=0.2, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=3.0, border_width=1.0, holes=False),
        "pyramid_stairs_inv": tg.MeshInvertedPyramidStairsTerrainCfg(
            proportion=0.2, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=3.0, border_width=1.0, holes=False),
        "boxes": tg.MeshRandomGridTerrainCfg(proportion=0.2, grid_width=0.45, grid_height_range=(0.05, 0.2), platform_width=2.0),
        "random_rough": tg.HfRandomUniformTerrainCfg(proportion=0.2, noise_range=(0.02, 0.10), noise_step=0.02, border_width=0.25),
        "hf_pyramid_slope": tg.HfPyramidSlopedTerrainCfg(proportion=0.1, slope_range=(0.0, 0.4), platform_width=2.0, border_width=0.25),
        "hf_pyramid_slope_inv": tg.HfInvertedPyramidSlopedTerrainCfg(proportion=0.1, slope_range=(0.0, 0.4), platform_width=2.0, border_width=0.25),
    },
)
