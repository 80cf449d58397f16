import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape, self.dtype = shape, dtype


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        super().__init__(tuple(shape), dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {np.dtype(self.dtype).name})"


class Dict(Space, dict):
    def __init__(self, spaces=None, **kw):
        dict.__init__(self, spaces or {}, **kw)
        Space.__init__(self, None, None)

    @property
    def spaces(self):
        return self

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k!r}: {v!r}" for k, v in self.items()) + ")"
