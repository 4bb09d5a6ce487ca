import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class Discrete:
    def __init__(self, n):
        self.n = int(n)

    def sample(self):
        return int(np.random.randint(self.n))


class MultiBinary:
    def __init__(self, n):
        self.n = n
