"""Import-time stand-in for `gymnasium` (NOT installed in the build container).

Generator-side plumbing only: lets tools/gen_golden.py import the unmodified
reference env layer.  Contains no step arithmetic.  Never shipped to the GPU box
as part of the product path and never imported by cooking_zoo_amd.
"""
from . import spaces, utils, envs  # noqa: F401


class Env:
    metadata = {}

    def __init__(self, *a, **k):
        pass
