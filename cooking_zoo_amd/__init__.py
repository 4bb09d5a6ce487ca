"""cooking_zoo_amd: MI355X-native batched implementation of the CookingZoo step() hot path.

Entry points: `cooking_zoo_amd.environment.cooking_env.parallel_env` (drop-in for one env),
`cooking_zoo_amd.vec_env.CookingVecEnv` (thousands of envs, one wavefront each).  When gymnasium is installed the
ids `cooking_zoo_amd:cookingEnv-v1` and `cooking_zoo_amd:cookingEnvMA-v1` are registered like the reference's
(cooking_zoo/__init__.py:3-8).  Importing this package never touches the GPU; creating an environment does.
"""
try:                                             # gymnasium is optional (absent in the build container)
    from gymnasium.envs.registration import register as _register

    _register(id="cookingEnv-v1", entry_point="cooking_zoo_amd.environment:GymCookingEnvironment")
    _register(id="cookingEnvMA-v1", entry_point="cooking_zoo_amd.environment:GymCookingEnvironmentMA")
except Exception:                                # pragma: no cover
    pass

__version__ = "0.1.0"
