import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_LIB = os.path.join(ROOT, "tests", "emu", "librl_env_emu.so")
REFERENCE = "/root/reference/source/robot_lab"


# Step kernels specialised at run time (robot_lab_amd/jit.py) are ON by default for users; the tiers pin them OFF and opt in case by case
# (tests/test_gpu_specs.py JIT_CASES, tests/test_jit.py), so that what a test runs - and how long a tier takes - does not depend on what a
# box's cache holds.  RL_ENV_JIT=1 from outside runs a whole tier on run-time specialised kernels (profiles/r06i_pytest_jit_tier_tail.txt).
os.environ.setdefault("RL_ENV_JIT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference (skipped where it is absent)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The suites need the in-tree libraries (HIP product libraries + the CPU lane emulator): build whatever is missing or
    stale - a no-op after `__graft_entry__.build()`, ~2 minutes of hipcc / g++ on a fresh checkout."""
    import torch

    import __graft_entry__ as g

    # on a GPU box the snapshot's time stamps are not trustworthy (and hipcc minutes are GPU minutes): only fill in what is missing
    g.build(only_missing=torch.cuda.is_available())


@pytest.fixture(scope="session")
def emu_lib():
    """CPU lane emulator (test infrastructure): same lane-program source as the HIP kernel, built with g++."""
    import __graft_entry__ as g

    import torch

    stale = not os.path.isfile(EMU_LIB) if torch.cuda.is_available() else g._stale(
        EMU_LIB, [os.path.join(g.CSRC, h) for h in g.HEADERS] + [os.path.join(g.EMU_DIR, "rl_env_emu.cpp")])
    if stale:
        import subprocess

        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-shared", "-fPIC", "-o", EMU_LIB,
                        os.path.join(g.EMU_DIR, "rl_env_emu.cpp")], check=True)
    return EMU_LIB
