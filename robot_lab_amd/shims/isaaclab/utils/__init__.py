from .configclass import GenericCfg, configclass, named_stub  # noqa: F401
