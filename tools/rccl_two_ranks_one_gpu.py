#!/usr/bin/env python3
"""First contact of the multi-rank RCCL exchange on a ONE-GPU box: two rank processes that both use device 0 (RCCL may refuse
duplicate devices; the answer either way is worth knowing).  Each rank: its shard of 512 envs, cz_comm_init under a deadline,
a rollout, cz_comm_barrier, cz_stats_allgather; rank 0 checks that the gathered structs are what the ranks hold locally."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd import distributed as czd  # noqa: E402

if "--worker" not in sys.argv:
    sys.exit(czd.launch_local(2, [os.path.abspath(__file__), "--worker"], timeout=150.0))

from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rdzv = czd.FileRendezvous.from_env(timeout=120.0)
begin, count = czd.shard_range(512, world, rank)
env = CookingVecEnv(count, "coop_test", "example", 2, 20, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=8,
                    auto_reset=True, device_id=0, env_id_base=begin)
env.reset(return_obs=False)
ok, msg = czd.comm_init_with_deadline(env, world, rank, rdzv, 60.0)
if not ok:
    if rank == 0:
        print(json.dumps({"rccl_two_ranks_on_one_device": "refused", "message": msg}))
    sys.stdout.flush()
    os._exit(0)
env.rollout(60, 5)
env.sync()
L = _native.lib()
_native.check(env._h, L.cz_comm_barrier(env._h))
got = czd.allgather_stats_rccl(env, world)
mine = env.stats()
every = [json.loads(b) for b in rdzv.all_gather(json.dumps(mine).encode())]
if rank == 0:
    print(json.dumps({"rccl_two_ranks_on_one_device": "ok", "allgather_equals_local_stats": got == every, "total_env_steps": czd.reduce_stats(got)["env_steps"]}))
rdzv.close()
env.close()
