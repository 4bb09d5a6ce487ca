import numpy as np


def np_random(seed=None):
    return np.random.default_rng(seed), seed
