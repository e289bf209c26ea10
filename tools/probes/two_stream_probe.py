"""Config 3 (262 144 boards) is launch-bound.  Are two half-batches, each on its own stream and fed by its own host thread,
faster per step than one launch per step?  (the boards are independent; each stream's chain stays ordered)"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
from qtttgym_amd import recommended_env; recommended_env(apply=True)
import torch
from qtttgym_amd import VecEnv
dev = torch.device("cuda", 0)
K, W = 200, 10
T = K + W
def make(B, off, seed=1):
    env = VecEnv(B, device=dev, seed=seed, auto_reset=True, board_offset=off)
    a = torch.empty((T, B, 2), dtype=torch.uint8, device=dev)
    for t in range(T):
        env.sample_actions(out=a[t]); env.step_raw(a[t])
    return env, a
def one(B, reps=9):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        env, a = make(B, 0)
        torch.cuda.synchronize()
        out = []
        for _ in range(reps):
            env.reset_raw(); env.step_many(a[:W]); torch.cuda.synchronize()
            t0 = time.perf_counter(); env.step_many(a[W:]); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / K * 1e6)
    out.sort(); return out[len(out) // 2], out[0]
def split(B, parts, reps=9):
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    envs = []
    for p, s in enumerate(streams):
        with torch.cuda.stream(s):
            envs.append(make(B // parts, p * (B // parts)))
    torch.cuda.synchronize()
    out = []
    def work(p, go, phase):
        with torch.cuda.stream(streams[p]):
            env, a = envs[p]
            if phase == 0:
                env.reset_raw(); env.step_many(a[:W])
            else:
                go.wait(); env.step_many(a[W:])
    for _ in range(reps):
        th = [threading.Thread(target=work, args=(p, None, 0)) for p in range(parts)]
        [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize()
        go = threading.Barrier(parts + 1)
        th = [threading.Thread(target=work, args=(p, go, 1)) for p in range(parts)]
        [t.start() for t in th]
        go.wait(); t0 = time.perf_counter()
        [t.join() for t in th]; torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / K * 1e6)
    out.sort(); return out[len(out) // 2], out[0]
for B in (262144, 65536, 1048576):
    print("boards %8d  one stream: med %.2f best %.2f us per step (host wall, K=%d)" % ((B,) + one(B) + (K,)), flush=True)
    for parts in (2, 4):
        print("boards %8d  %d streams x %d boards, one host thread each: med %.2f best %.2f" % ((B, parts, B // parts) + split(B, parts)), flush=True)
