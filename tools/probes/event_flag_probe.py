"""K = 20 regions timed by (a) torch.cuda.Event (default flags: a system-scope fence when the event completes) and
(b) raw HIP events created with hipEventDisableSystemFence, alternating, same process."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from qtttgym_amd import recommended_env; recommended_env(apply=True)
import torch
from qtttgym_amd import VecEnv
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
def mk(flags):
    e = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0
    return e
NOFENCE = 0x20000000
B, K, W = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 20, 5
T = K + W
env = VecEnv(B, device=dev, seed=1, auto_reset=True)
actions = torch.empty((T, B, 2), dtype=torch.uint8, device=dev)
for t in range(T):
    env.sample_actions(out=actions[t]); env.step_raw(actions[t])
torch.cuda.synchronize()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h0, h1 = mk(0), mk(0)
n0, n1 = mk(NOFENCE), mk(NOFENCE)
def region(kind):
    torch.cuda.synchronize()
    env.reset_raw(); env.step_many(actions[:W])
    if kind == "torch": t0.record()
    else: assert hip.hipEventRecord(h0 if kind == "hip" else n0, stream) == 0
    env.step_many(actions[W:])
    if kind == "torch": t1.record()
    else: assert hip.hipEventRecord(h1 if kind == "hip" else n1, stream) == 0
    torch.cuda.synchronize()
    if kind == "torch": return t0.elapsed_time(t1) * 1e3 / K
    ms = ctypes.c_float()
    a, b = (h0, h1) if kind == "hip" else (n0, n1)
    assert hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
    return ms.value * 1e3 / K
res = {"torch": [], "hip": [], "nofence": []}
for r in range(200):
    for kind in res: res[kind].append(region(kind))
for kind, v in res.items():
    v.sort()
    print("K=%d  %-8s events: median %.3f  p10 %.3f  min %.3f us per launch" % (K, kind, v[len(v) // 2], v[len(v) // 10], v[0]), flush=True)
