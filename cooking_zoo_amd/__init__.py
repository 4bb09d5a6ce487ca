"""cooking_zoo_amd: MI355X-native batched implementation of the CookingZoo step() hot path.

Entry points: `cooking_zoo_amd.environment.cooking_env.parallel_env` (drop-in for one env),
`cooking_zoo_amd.CookingVecEnv` (thousands of envs, one wavefront each), `cooking_zoo_amd.ShardedVecEnv` (one batch over several
GPUs).  When gymnasium is installed the
ids `cookingEnv-v1`, `cookingEnvMA-v1` and `cookingZooEnv-v0` (the raw agent-iterator environment) are registered
like the reference's (cooking_zoo/__init__.py:3-8).  Importing this package never touches the GPU; creating an environment does.
"""
try:                                             # gymnasium is optional (absent in the build container)
    from gymnasium.envs.registration import register as _register

    _register(id="cookingEnv-v1", entry_point="cooking_zoo_amd.environment:GymCookingEnvironment")
    _register(id="cookingEnvMA-v1", entry_point="cooking_zoo_amd.environment:GymCookingEnvironmentMA")
    _register(id="cookingZooEnv-v0", entry_point="cooking_zoo_amd.environment.cooking_env:AECCookingEnvironment")
except Exception:                                # pragma: no cover
    pass

__version__ = "0.1.0"


def __getattr__(name):
    # the batched environments, importable from the package root without loading anything at `import cooking_zoo_amd`
    if name == "CookingVecEnv":
        from cooking_zoo_amd.vec_env import CookingVecEnv
        return CookingVecEnv
    if name == "ShardedVecEnv":
        from cooking_zoo_amd.sharded import ShardedVecEnv
        return ShardedVecEnv
    raise AttributeError(name)
