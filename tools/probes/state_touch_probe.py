import os, sys, time
sys.path.insert(0, os.getcwd())
from qtttgym_amd import recommended_env; recommended_env(apply=True)
import torch
from qtttgym_amd import VecEnv
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
B, K, W = 1 << 20, 200, 10
T = K + W
env = VecEnv(B, device=dev, seed=1, auto_reset=True)
actions = torch.empty((T, B, 2), dtype=torch.uint8, device=dev)
for t in range(T):
    env.sample_actions(out=actions[t]); env.step_raw(actions[t])
torch.cuda.synchronize()
big = torch.empty(1 << 28, dtype=torch.int32, device=dev)            # 1 GiB: four times the Infinity Cache
big.fill_(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def region():
    env.reset_raw(); env.step_many(actions[:W])
    e0.record(); env.step_many(actions[W:]); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / K
def med(n=7):
    v = sorted(region() for _ in range(n)); return "med %.3f best %.3f" % (v[n // 2], v[0])
def flush():
    big.sum(); torch.cuda.synchronize()
for rnd in range(3):
    print("as it comes                    :", med(), flush=True)
    flush()
    print("after reading 1 GiB (flush)    :", med(), flush=True)
    env.state.view(torch.int64).sum(); torch.cuda.synchronize()
    print("after ONE state.sum()          :", med(), flush=True)
    print("again, nothing in between      :", med(), flush=True)
    flush()
    print("after the flush                :", med(), flush=True)
    env._reward.sum(); env._terminated.sum(); actions.sum(); torch.cuda.synchronize()
    print("after reward/term/actions sums :", med(), flush=True)
    flush()
    print("after the flush                :", med(), flush=True)
