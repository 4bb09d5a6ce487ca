"""GPU: the direct RCCL path of the C-ABI with a one-rank communicator (the multi-rank case needs >1 GPU)."""
import pytest

pytestmark = pytest.mark.gpu


def test_stats_allgather_single_rank():
    from cooking_zoo_amd import distributed as czd
    from cooking_zoo_amd.vec_env import CookingVecEnv
    env = CookingVecEnv(256, "coop_test", "example", 2, 20, ["TomatoLettuceSalad", "CarrotBanana"],
                        action_scheme="scheme3", num_layouts=8, auto_reset=True)
    env.reset(return_obs=False)
    env.rollout(50, 3)
    env.sync()
    local = env.stats()
    gathered = czd.gather_stats_rccl(env, 1, 0, lambda payload: payload)
    assert gathered == [local]
    assert czd.reduce_stats(gathered)["env_steps"] == local["env_steps"] > 0
    assert local["episodes"] >= 256 * 2
    env.close()
