"""Test-only loader for the UNMODIFIED reference package at /root/reference.

Only used (a) by ``make_golden.py`` to generate the committed fixtures and (b) by
the live-fuzz tests that run in the build container.  The GPU box has no
/root/reference, so everything here is guarded by ``reference_available()``.

The reference's ``qtttgym/env.py:5-8`` imports ``gymnasium`` and
``ray.rllib...Repeated`` for *space declarations only* (env.py:19-25); neither is
installed here, so empty placeholder modules are pre-seeded into ``sys.modules``.
No arithmetic goes through the placeholders.  The one random draw on the path
(``qeval.py:35``, ``random.choice((lo, hi))``) is pinned by swapping the module
global ``qtttgym.qeval.random`` for a bit source: ``choice(seq) -> seq[bit]``.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("QTTT_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "qtttgym", "board.py"))


class _Space:
    def __init__(self, *args, **kwargs):
        self.args = args
        self.kwargs = kwargs


class _GymEnvBase:
    def __init__(self):
        pass


def _install_placeholders():
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        spaces = types.ModuleType("gymnasium.spaces")
        for name in ("Discrete", "Tuple", "Dict", "Box"):
            setattr(spaces, name, type(name, (_Space,), {}))
        gym.Env = _GymEnvBase
        gym.spaces = spaces
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces
    chain = ("ray", "ray.rllib", "ray.rllib.utils", "ray.rllib.utils.spaces",
             "ray.rllib.utils.spaces.repeated")
    for name in chain:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    rep = sys.modules["ray.rllib.utils.spaces.repeated"]
    if not hasattr(rep, "Repeated"):
        rep.Repeated = type("Repeated", (_Space,), {})


class BitSource:
    """Stands in for the ``random`` module inside the reference's qeval.py."""

    def __init__(self):
        self.bit = 0
        self.calls = 0

    def choice(self, seq):
        assert len(seq) == 2, seq
        self.calls += 1
        return seq[self.bit]


_cached = None


def load_reference():
    """Returns (qtttgym module, BitSource wired into qtttgym.qeval.random)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    _install_placeholders()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import qtttgym  # the reference package, unmodified
    assert os.path.realpath(qtttgym.__file__).startswith(os.path.realpath(REFERENCE_ROOT)), \
        "imported a qtttgym that is not the reference: %s" % qtttgym.__file__
    src = BitSource()
    qtttgym.qeval.random = src
    _cached = (qtttgym, src)
    return _cached
