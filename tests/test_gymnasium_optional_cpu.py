"""`qtttgym_amd.Env` is a `gymnasium.Env` with real gymnasium spaces whenever gymnasium is importable, as the reference's
is (qtttgym/env.py:5-8,15,19-25) — and a plain class with declarative stand-ins when it is not (this image has neither
gymnasium nor ray: the REAL libraries were not available to test against; what is tested is the guarded import, with stub
modules of the same shape pre-seeded into sys.modules in a child process — the technique of tests/golden/ref_shim.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_STUBS = r"""
import sys, types
import numpy as np
gym = types.ModuleType("gymnasium")
spaces = types.ModuleType("gymnasium.spaces")
class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape, self.dtype = shape, dtype
    def contains(self, x): raise NotImplementedError
class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        super().__init__((), np.int64); self.n, self.start = int(n), int(start)
    def contains(self, x): return isinstance(x, (int, np.integer)) and self.start <= int(x) < self.start + self.n
    def sample(self, mask=None): return int(np.random.randint(self.n))
class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        super().__init__(tuple(shape), np.dtype(dtype)); self.low, self.high, self.shape = low, high, tuple(shape)
class Tuple(Space):
    def __init__(self, spaces, seed=None):
        super().__init__(); self.spaces = tuple(spaces)
        assert all(isinstance(s, Space) for s in self.spaces), "Tuple wants Space instances"
    def __len__(self): return len(self.spaces)
    def __getitem__(self, i): return self.spaces[i]
    def contains(self, x): return len(x) == len(self.spaces) and all(s.contains(v) for s, v in zip(self.spaces, x))
class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        super().__init__(); self.spaces = dict(spaces or {}, **kw)
        assert all(isinstance(s, Space) for s in self.spaces.values()), "Dict wants Space instances (as gymnasium does)"
    def __getitem__(self, k): return self.spaces[k]
    def keys(self): return self.spaces.keys()
for c in (Space, Discrete, Box, Tuple, Dict):
    setattr(spaces, c.__name__, c)
class Env:
    metadata = {"render_modes": []}
    render_mode = None
    @property
    def unwrapped(self): return self
    def close(self): pass
gym.Env, gym.Space, gym.spaces = Env, Space, spaces
sys.modules["gymnasium"], sys.modules["gymnasium.spaces"] = gym, spaces
if WITH_RAY:
    for name in ("ray", "ray.rllib", "ray.rllib.utils", "ray.rllib.utils.spaces", "ray.rllib.utils.spaces.repeated"):
        sys.modules[name] = types.ModuleType(name)
    class Repeated(Space):
        def __init__(self, child_space, max_len):
            super().__init__(); self.child_space, self.max_len = child_space, max_len
    sys.modules["ray.rllib.utils.spaces.repeated"].Repeated = Repeated
"""

_CHECK = r"""
sys.path.insert(0, %r)
import gymnasium
import qtttgym_amd
from qtttgym_amd import Env, VecEnv, spaces as qs
assert qs.GYMNASIUM is gymnasium and issubclass(Env, gymnasium.Env), Env.__mro__       # env.py:15
env = Env()                                                                           # (no device call: the Board is lazy)
assert isinstance(env, gymnasium.Env) and env.unwrapped is env
G = gymnasium.spaces
a, o = env.action_space, env.observation_space
assert type(a) is G.Tuple and all(type(s) is G.Discrete and s.n == 9 for s in a.spaces) and len(a) == 2   # env.py:19
assert type(o) is G.Dict and set(o.keys()) == {"q_states_p1", "q_states_p2", "classical", "turn"}         # env.py:20-25
for key, max_len in (("q_states_p1", 5), ("q_states_p2", 4)):
    r = o[key]
    assert isinstance(r, G.Space) and r.max_len == max_len
    assert type(r.child_space) is G.Tuple and [s.n for s in r.child_space.spaces] == [9, 9]
    if WITH_RAY:
        from ray.rllib.utils.spaces.repeated import Repeated
        assert type(r) is Repeated
    else:
        assert type(r).__name__ == "GymRepeated" and r.contains([(0, 1), (2, 3)]) and not r.contains([(0, 1)] * 9)
assert type(o["classical"]) is G.Box and o["classical"].shape == (9,) and o["classical"].low == -1 and o["classical"].high == 1
assert np.dtype(o["classical"].dtype) == np.int32 and type(o["turn"]) is G.Discrete and o["turn"].n == 2
assert a.contains((3, 8)) and not a.contains((3, 9))
# reset() / step() keep the reference's signatures on the gymnasium base (env.py:34,55)
import inspect
assert list(inspect.signature(Env.reset).parameters) == ["self", "seed", "options"]
assert list(inspect.signature(Env.step).parameters) == ["self", "action", "verbose"]
print("ok")
"""


def _child(with_ray):
    code = "WITH_RAY = %r\n" % with_ray + _STUBS + _CHECK % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-2500:])


def test_env_is_a_gymnasium_env_with_gymnasium_and_ray_present():
    _child(True)


def test_env_is_a_gymnasium_env_with_gymnasium_alone():
    _child(False)


def test_placeholders_parked_in_sys_modules_are_not_mistaken_for_gymnasium():
    """tests/golden/ref_shim.py parks empty `gymnasium` / `ray` modules in sys.modules to import the reference: this
    package must not take those for the libraries (they have no spaces.Space)."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import ref_shim; ref_shim._install_placeholders(); "
            "from qtttgym_amd import Env, spaces as qs; assert qs.GYMNASIUM is None and Env.__mro__[1] is qs.PlainEnvBase; "
            "o = qs.reference_observation_space(); assert type(o) is qs.Dict and o['q_states_p1'].max_len == 5; print('ok')"
            % (ROOT, os.path.join(ROOT, "tests", "golden")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-2500:])


def test_without_gymnasium_the_stand_ins_are_used():
    import importlib.util
    if importlib.util.find_spec("gymnasium") is not None:
        import pytest
        pytest.skip("gymnasium is installed here")
    from qtttgym_amd import Env, spaces as qs
    assert qs.GYMNASIUM is None and qs.GYM_ENV_BASE is qs.PlainEnvBase and Env.__mro__ == (Env, qs.PlainEnvBase, object)
    assert type(qs.reference_action_space()) is qs.Tuple and type(qs.reference_observation_space()) is qs.Dict
    # what a gym loop touches on the base class besides reset / step (the reference inherits it from gymnasium.Env)
    with Env() as env:
        assert env.unwrapped is env and env.render_mode is None and env.metadata == {"render_modes": []} and env.spec is None
    env.close()
