#!/usr/bin/env python3
"""Host cost per default VecEnv.step() call by how the output tensors are obtained (4 096 boards: the kernel is ~3 us, so
the loop is host-bound and the per-call time IS the host cost): a fresh allocation carved by csrc/fastviews.cpp (the default),
the same carved by Python-level torch calls (what runs when _fastviews.so is not built), pooled output sets
(VecEnv(output_pool=4)), the zero-copy form (step_observe_raw), the bare step (step_raw)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qtttgym_amd import VecEnv, vec_env
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
def per_call(fn, reps=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
pooled, fresh = VecEnv(n, seed=1, auto_reset=True, output_pool=4), VecEnv(n, seed=1, auto_reset=True)
a = pooled.sample_actions()
def rebinding(env):
    def f():
        obs, r, tm, tr, info = env.step(a)
    return f
out = {"boards": n, "fastviews_built": vec_env._fastviews is not None,
       "step_raw_us": per_call(lambda: pooled.step_raw(a)),
       "step_observe_raw_us": per_call(lambda: pooled.step_observe_raw(a)),
       "default_step_us": per_call(rebinding(fresh)),
       "step_with_output_pool_4_us": per_call(rebinding(pooled)),
       "carve_alone_us": per_call(lambda: vec_env._carve(n, fresh.device), 2000),
       "carve_in_python_alone_us": per_call(lambda: vec_env._carve_py(n, fresh.device), 2000)}
vec_env._carve = vec_env._carve_py
out["default_step_without_fastviews_us"] = per_call(rebinding(fresh))
print(json.dumps(out))
