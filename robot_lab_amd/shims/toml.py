"""`toml` stand-in (robot_lab/tasks/__init__.py imports it; only `load` is used) on top of tomli."""
import tomli


def load(path):
    with open(path, "rb") as f:
        return tomli.load(f)


def loads(s):
    return tomli.loads(s)
