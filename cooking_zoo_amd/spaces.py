"""Observation / action spaces.  gymnasium's classes are used when gymnasium is importable; otherwise minimal
duck-typed stand-ins with the same attributes (shape, dtype, low, high, n, sample, contains)."""
import numpy as np

try:                                            # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box, Discrete, MultiBinary  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:                               # gymnasium is not installed in the build container
    HAVE_GYMNASIUM = False

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def contains(self, x):
            return 0 <= int(x) < self.n

        def sample(self):
            return int(np.random.randint(self.n))

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiBinary:
        def __init__(self, n):
            self.n = int(n)
            self.shape = (self.n,)
            self.dtype = np.dtype(np.int8)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all((x == 0) | (x == 1)))

        def sample(self):
            return np.random.randint(0, 2, size=self.shape).astype(self.dtype)

        def __repr__(self):
            return f"MultiBinary({self.n})"
