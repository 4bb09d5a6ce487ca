#!/usr/bin/env python3
"""GPU box: the largest batch whose runs may go out as overlapped launches (cz_overlap_limit), per kernel instance."""
import sys; sys.path.insert(0,'/root/repo')
from cooking_zoo_amd.vec_env import CookingVecEnv
from cooking_zoo_amd import _native
for lvl,meta,a in (("coop_test","example",2),("large_16x16","large_16x16",4),("huge_32x32","huge_32x32",4)):
    env=CookingVecEnv(64,lvl,meta,a,50,["TomatoSalad"]*a)
    print(lvl, _native.lib().cz_overlap_limit(env._h)); env.close()
