"""Step kernels specialised on a task at RUN time (robot_lab_amd/jit.py; include/rl_env.h "specialising ANY task"), CPU tier: everything up
to the device - the Spec source out of the product library, the plugin cross-compiled for gfx950 (hipcc needs no GPU), the ABI guard, the
registry.  The kernels themselves against the interpreter: tests/test_gpu_specs.py::test_run_time_specialised_kernel_equals_interpreter."""
import ctypes
import os
import subprocess

import pytest

from robot_lab_amd import capi, jit
from robot_lab_amd.scene import build_world, load_bundle

B2 = "RobotLab-Isaac-Velocity-Rough-Unitree-B2-v0"


@pytest.fixture(scope="module")
def lib():
    return capi.load_library()


@pytest.fixture(scope="module")
def cache(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("jit_cache"))
    old = os.environ.get("RL_ENV_JIT_CACHE")
    os.environ["RL_ENV_JIT_CACHE"] = d
    yield d
    if old is None:
        os.environ.pop("RL_ENV_JIT_CACHE", None)
    else:
        os.environ["RL_ENV_JIT_CACHE"] = old


def _desc(task):
    desc, extra = load_bundle(task)
    build_world(desc, extra, 16, 0)
    return desc


def test_library_carries_the_stamp_of_this_tree(lib):
    assert lib.rl_env_abi_stamp().decode() == jit.abi_stamp()


def test_spec_source_from_the_product_library(lib):
    src = jit.spec_source(lib, _desc(B2), "Spec_X", B2, 1234)
    assert src.startswith("struct Spec_X {") and "ID = 1234" in src and "using TP = TopoQuad3;" in src
    # the kinds the built-in Specs never needed (hand-stand terms, base_height_l2, wheel_vel_penalty, feet_distance_*) are specialisable too
    for task in ("RobotLab-Isaac-Velocity-Flat-HandStand-Unitree-A1-v0", "RobotLab-Isaac-Velocity-Rough-DDTRobot-Tita-v0"):
        assert jit.spec_source(lib, _desc(task), "Spec_Y", task, 1235) is not None, lib.rl_env_last_error()
    # a task with a reward kind the specialised evaluation does not implement (action_sync: weight 0 in every shipped cfg) says so and would
    # stay on the interpreter
    from helpers import A1_SYNC_GROUPS, set_action_sync

    a1 = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
    d = _desc(a1)
    set_action_sync(d, "joint_power", A1_SYNC_GROUPS)
    assert jit.spec_source(lib, d, "Spec_W", a1, 1236) is None
    assert "no specialised evaluation" in lib.rl_env_last_error().decode()


def test_plugin_compiles_registers_and_is_cached(lib, cache):
    n0 = lib.rl_env_spec_plugin_count()
    path = jit.build_plugin(lib, _desc(B2), B2, 4)
    assert path and os.path.isfile(path) and os.path.dirname(path) == cache
    so = ctypes.CDLL(path)
    for name in ("rl_spec_plugin_abi", "rl_spec_plugin_id", "rl_spec_plugin_matches", "rl_spec_plugin_launch"):
        assert hasattr(so, name)
    so.rl_spec_plugin_abi.restype = ctypes.c_char_p
    assert so.rl_spec_plugin_abi().decode() == jit.abi_stamp() and so.rl_spec_plugin_id() >= 1000
    t = os.path.getmtime(path)
    assert jit.build_plugin(lib, _desc(B2), B2, 4) == path and os.path.getmtime(path) == t  # second time: the cache
    assert jit.build_plugin(lib, _desc(B2), B2, 2) != path                                    # another lane mapping: another object
    assert jit.specialise(lib, _desc(B2), B2, 4)
    assert lib.rl_env_spec_plugin_count() == n0 + 1
    assert jit.specialise(lib, _desc(B2), B2, 4) and lib.rl_env_spec_plugin_count() == n0 + 1


def test_plugin_built_against_other_headers_is_refused(lib, cache, tmp_path):
    src = jit.plugin_source(jit.spec_source(lib, _desc(B2), "Spec_Z", B2, 4321), "Spec_Z", 4321)
    hip = tmp_path / "z.hip"
    hip.write_text(src)
    out = str(tmp_path / "z.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, *jit.ENV_FLAGS, jit.stamp_flag("0123456789abcdef"), "-DRL_ENV_SPEC_SUB=4", "-shared", "-fPIC", "-o", out, str(hip)], check=True)
    assert lib.rl_env_register_spec_plugin(out.encode()) != 0
    assert "compiled against csrc headers 0123456789abcdef" in lib.rl_env_last_error().decode()
    assert lib.rl_env_register_spec_plugin(b"/nonexistent/plugin.so") != 0


def test_failures_fall_back_to_the_interpreter(lib, monkeypatch):
    """An unwritable cache directory, a compiler that is not there: one log line and False (the caller keeps its interpreter env), never an exception."""
    monkeypatch.setenv("RL_ENV_JIT_CACHE", "/proc/nonexistent/cache")
    assert jit.specialise(lib, _desc(B2), B2, 4) is False
    monkeypatch.delenv("RL_ENV_JIT_CACHE")
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("RL_ENV_JIT_CACHE", "/tmp")
    assert jit.build_plugin(lib, _desc("RobotLab-Isaac-Velocity-Rough-Unitree-B2W-v0"), "b2w", 2) is None
