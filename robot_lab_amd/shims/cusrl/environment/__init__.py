"""cusrl.environment stand-in (see robot_lab_amd/shims/cusrl/__init__.py)."""
