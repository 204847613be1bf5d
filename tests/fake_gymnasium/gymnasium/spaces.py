"""Box of the gymnasium stand-in (see gymnasium/__init__.py here)."""
import numpy as np


class Space:
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape)
        self._shape = self.shape
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low.astype(np.float64), self.high.astype(np.float64)).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))
