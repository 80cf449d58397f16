"""Minimal `configclass` [UPSTREAM isaaclab.utils.configclass]: every non-callable class attribute
(annotated or not) becomes a per-instance field initialised with a deep copy of its default;
keyword arguments override; `__post_init__` runs last.  Enough for the reference's cfg classes
(`VEL/velocity_env_cfg.py:42-743`) to instantiate and mutate exactly as they do upstream."""
import copy
import types
from dataclasses import MISSING  # noqa: F401


def _is_field(name, val):
    if name.startswith("__"):
        return False
    if isinstance(val, (types.FunctionType, classmethod, staticmethod, property)):
        return False
    if isinstance(val, type):  # nested class definitions / class_type fields stay class attributes
        return False
    return True


def _cfg_init(self, *args, **kwargs):
    order = []
    for klass in reversed(type(self).__mro__):
        for k, v in vars(klass).items():
            if _is_field(k, v):
                if k not in order:
                    order.append(k)
                object.__setattr__(self, k, copy.deepcopy(v))
    if len(args) > len(order):
        raise TypeError(f"{type(self).__name__}: too many positional arguments")
    for k, v in zip(order, args):  # dataclass-style positional fields, definition order
        setattr(self, k, v)
    for k, v in kwargs.items():
        setattr(self, k, v)
    post = getattr(self, "__post_init__", None)
    if post is not None:
        post()


def _cfg_replace(self, **kwargs):
    new = copy.deepcopy(self)
    for k, v in kwargs.items():
        setattr(new, k, v)
    return new


def _cfg_to_dict(self):
    out = {}
    for k, v in vars(self).items():
        if hasattr(v, "to_dict"):
            out[k] = v.to_dict()
        elif callable(v):
            out[k] = f"{getattr(v, '__module__', '')}:{getattr(v, '__name__', repr(v))}"
        elif isinstance(v, dict):
            out[k] = {kk: (vv.to_dict() if hasattr(vv, "to_dict") else vv) for kk, vv in v.items()}
        else:
            out[k] = v
    return out


def _cfg_repr(self):
    return f"{type(self).__name__}({', '.join(f'{k}={v!r}' for k, v in vars(self).items())})"


def configclass(cls):
    cls.__init__ = _cfg_init
    cls.replace = _cfg_replace
    cls.copy = lambda self: copy.deepcopy(self)
    cls.to_dict = _cfg_to_dict
    cls.__repr__ = _cfg_repr
    return cls


class _GenericMeta(type):
    def __getattr__(cls, name):  # nested cfg classes such as RayCasterCfg.OffsetCfg
        if name.startswith("__") or not name[:1].isupper():
            raise AttributeError(name)
        sub = _GenericMeta(name, (GenericCfg,), {"__module__": cls.__module__})
        setattr(cls, name, sub)
        return sub


class GenericCfg(metaclass=_GenericMeta):
    """Cfg object accepting any keyword (placeholder for upstream cfg classes we only read)."""

    __init__ = _cfg_init
    replace = _cfg_replace
    to_dict = _cfg_to_dict
    __repr__ = _cfg_repr

    def copy(self):
        return copy.deepcopy(self)

    def __getattr__(self, name):
        # unknown *data* attributes read as None (upstream defaults we never consume)
        if name.startswith("__"):
            raise AttributeError(name)
        return None


class _Stub:
    """Named placeholder for an upstream function *or* sub-module (e.g. `isaaclab.sensors.patterns`)."""

    def __init__(self, name, module):
        self.__name__ = self.__qualname__ = name
        self.__module__ = module

    def __call__(self, *args, **kwargs):
        raise NotImplementedError(
            f"{self.__module__}.{self.__name__} is an upstream IsaacLab symbol; the MI355X env evaluates it inside its HIP kernels")

    def __getattr__(self, name):
        if name.startswith("__") or not name[:1].isupper():
            raise AttributeError(name)
        sub = _GenericMeta(name, (GenericCfg,), {"__module__": f"{self.__module__}.{self.__name__}"})
        setattr(self, name, sub)
        return sub

    def __repr__(self):
        return f"<upstream {self.__module__}.{self.__name__}>"


def named_stub(name, module="isaaclab"):
    return _Stub(name, module)
