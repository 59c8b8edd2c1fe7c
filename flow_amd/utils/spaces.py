"""gym.spaces.Box when gym is installed, else a minimal stand-in with the same attributes
(the reference imports gym.spaces.box.Box, flow/envs/ring/accel.py:6; gym is absent in this image)."""
import numpy as np

try:                                             # pragma: no cover - depends on the image
    from gym.spaces import Box, Tuple            # noqa: F401
    HAVE_GYM = True
except Exception:
    HAVE_GYM = False

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1e3)
            hi = np.where(np.isfinite(self.high), self.high, 1e3)
            return np.random.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)

    class Tuple(tuple):
        pass
