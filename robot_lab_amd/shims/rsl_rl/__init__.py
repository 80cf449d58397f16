"""STAND-IN for the third-party `rsl_rl` package (rsl-rl-lib 3.0.1, ETH RSL), which the reference's launch scripts import
(`scripts/reinforcement_learning/rsl_rl/train.py:88`, `play.py:14`) and which cannot be installed here (no network).  Only
`rsl_rl.runners.OnPolicyRunner` with the calls those two scripts make; the PPO update is `robot_lab_amd.ppo` (a restatement of the
library's published update rule, unpinned against it), the collection loop the HIP kernels of this repository.  With the real
library installed this package is never imported (the shims directory is APPENDED to sys.path)."""
__version__ = "3.0.1+robotlabamd.standin"
