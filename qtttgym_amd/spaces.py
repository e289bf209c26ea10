"""The space objects the reference declares (env.py:19-25).

The reference's Env is a `gymnasium.Env` whose spaces are `gymnasium.spaces.{Tuple, Discrete, Dict, Box}` and
`ray.rllib.utils.spaces.repeated.Repeated` (env.py:5-8,15).  Neither library is a dependency of this package (both are
absent on the target image), so:
  * when gymnasium is importable, `reference_action_space()` / `reference_observation_space()` build the REAL gymnasium
    spaces, `GYM_ENV_BASE` is `gymnasium.Env` (qtttgym_amd.Env subclasses it), and `Repeated` is ray's class when ray is
    importable too — otherwise a `gymnasium.Space` subclass with ray's fields (child_space, max_len);
  * otherwise the declarative stand-ins below carry the same names and fields, so `env.action_space` /
    `env.observation_space` read the same.  They declare, they do not compute.
Only a library that looks like gymnasium is used (it must have `spaces.Space` with `Discrete` derived from it): a
placeholder module somebody parked in `sys.modules` is not."""
import numpy as np


def _find_gymnasium():
    try:
        import gymnasium
        from gymnasium import spaces
        if not (isinstance(spaces.Space, type) and issubclass(spaces.Discrete, spaces.Space) and isinstance(gymnasium.Env, type)):
            return None, None
        for name in ("Tuple", "Dict", "Box"):
            if not issubclass(getattr(spaces, name), spaces.Space):
                return None, None
        return gymnasium, spaces
    except Exception:                       # noqa: BLE001 — never a hard dependency, whatever a broken install raises
        return None, None


def _find_ray_repeated(gspaces):
    try:
        from ray.rllib.utils.spaces.repeated import Repeated as R
        return R if isinstance(R, type) and issubclass(R, gspaces.Space) else None
    except Exception:                       # noqa: BLE001
        return None


class PlainEnvBase:
    """What a gym loop written against the reference touches on `gymnasium.Env` besides reset / step / render (the
    reference's Env inherits them, env.py:15): close(), `unwrapped`, the context-manager form, and the class attributes
    wrappers read.  Used as Env's base when gymnasium is not importable."""
    metadata = {"render_modes": []}
    render_mode = None
    spec = None

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


GYMNASIUM, GYM_SPACES = _find_gymnasium()
GYM_ENV_BASE = GYMNASIUM.Env if GYMNASIUM is not None else PlainEnvBase
RAY_REPEATED = _find_ray_repeated(GYM_SPACES) if GYMNASIUM is not None else None


class Space:
    def __eq__(self, other):
        return type(self) is type(other) and self.__dict__ == other.__dict__

    def __repr__(self):
        args = ", ".join("%s=%r" % kv for kv in self.__dict__.items())
        return "%s(%s)" % (type(self).__name__, args)


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)

    def contains(self, x):
        return isinstance(x, (int, np.integer)) and 0 <= int(x) < self.n

    def sample(self, rng=None):
        rng = rng or np.random
        return int(rng.randint(self.n))


class Box(Space):
    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype).name


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def contains(self, x):
        return len(x) == len(self.spaces) and all(s.contains(v) for s, v in zip(self.spaces, x))

    def sample(self, rng=None):
        return tuple(s.sample(rng) for s in self.spaces)


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()


class Repeated(Space):
    """ray.rllib.utils.spaces.repeated.Repeated(child_space, max_len) (env.py:20-21)."""

    def __init__(self, child_space, max_len):
        self.child_space = child_space
        self.max_len = int(max_len)


if GYMNASIUM is not None and RAY_REPEATED is None:
    class GymRepeated(GYM_SPACES.Space):
        """ray.rllib.utils.spaces.repeated.Repeated's fields on a gymnasium.Space, for a host that has gymnasium but
        not ray: a list of at most max_len elements of child_space."""

        def __init__(self, child_space, max_len):
            super().__init__()
            self.child_space = child_space
            self.max_len = int(max_len)

        def sample(self, *args, **kwargs):
            n = int(np.random.randint(1, self.max_len + 1))
            return [self.child_space.sample() for _ in range(n)]

        def contains(self, x):
            return isinstance(x, (list, np.ndarray)) and len(x) <= self.max_len and all(self.child_space.contains(c) for c in x)

        def __repr__(self):
            return "Repeated(%r, %d)" % (self.child_space, self.max_len)


def reference_action_space():
    if GYMNASIUM is not None:
        G = GYM_SPACES
        return G.Tuple((G.Discrete(9), G.Discrete(9)))           # env.py:19, the real classes
    return Tuple((Discrete(9), Discrete(9)))                     # env.py:19


def reference_observation_space():
    # env.py:20-25, declared bounds of `classical` kept as the reference states them (-1..1),
    # although the values are -1..8 (the TODO at env.py:18)
    if GYMNASIUM is not None:
        G = GYM_SPACES
        Rep = RAY_REPEATED if RAY_REPEATED is not None else GymRepeated
        return G.Dict({
            "q_states_p1": Rep(G.Tuple((G.Discrete(9), G.Discrete(9))), 5),
            "q_states_p2": Rep(G.Tuple((G.Discrete(9), G.Discrete(9))), 4),
            "classical": G.Box(-1, 1, shape=(9,), dtype=np.int32),
            "turn": G.Discrete(2),
        })
    return Dict({
        "q_states_p1": Repeated(Tuple((Discrete(9), Discrete(9))), 5),
        "q_states_p2": Repeated(Tuple((Discrete(9), Discrete(9))), 4),
        "classical": Box(-1, 1, shape=(9,), dtype=np.int32),
        "turn": Discrete(2),
    })
