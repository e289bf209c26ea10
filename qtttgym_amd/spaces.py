"""Declarative stand-ins for the space objects the reference declares (env.py:19-25).

gymnasium / ray.rllib are not dependencies of this package (they are absent on the target
image); these carry the same names and fields so `env.action_space` / `env.observation_space`
read the same.  They declare, they do not compute."""
import numpy as np


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


def reference_action_space():
    return Tuple((Discrete(9), Discrete(9)))                     # env.py:19


def reference_observation_space():
    # env.py:20-25, declared bounds of `classical` kept as the reference states them (-1..1),
    # although the values are -1..8 (the TODO at env.py:18)
    return Dict({
        "q_states_p1": Repeated(Tuple((Discrete(9), Discrete(9))), 5),
        "q_states_p2": Repeated(Tuple((Discrete(9), Discrete(9))), 4),
        "classical": Box(-1, 1, shape=(9,), dtype=np.int32),
        "turn": Discrete(2),
    })
