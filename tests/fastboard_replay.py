"""Replays recorded cases of the single-board façade's host bookkeeping through a build of csrc/fastboard.c, with a ctypes
callback standing in for qttt_board_op_host (no GPU, no torch: this file also runs in a child process under
LD_PRELOAD=libasan, where importing torch would be slow and noisy).

    python tests/fastboard_replay.py <_fastboard.so> <cases.json> [loop_calls]

Used by tests/test_fastboard_cpu.py: in-process against qtttgym_amd/_fastboard.so, and in children against the
ASan + UBSan build and for the leak check (allocated blocks, GC objects and RSS over >= 10^5 calls)."""
import ctypes
import gc
import importlib.util
import json
import random
import resource
import sys


class PlainBoard:
    """The attributes fastboard.c reads and writes (board.py:4-6 of the reference) and nothing else."""

    def __init__(self, moves, board, qstructs):
        self.moves, self.board, self.qstructs = moves, board, qstructs
        self._win = None


def load(path):
    spec = importlib.util.spec_from_file_location("_fastboard", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Harness:
    def __init__(self, fb):
        self.fb = fb
        self.buf_in = (ctypes.c_uint8 * 64)()
        self.buf_out = (ctypes.c_uint8 * 64)()
        self.seen, self.reply, self.rc = None, bytes(64), 0

        @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)
        def op_host(p_in, p_out, n, stream):
            self.seen = ctypes.string_at(p_in, 64)
            if self.rc == 0:
                ctypes.memmove(p_out, self.reply, 64)
            return self.rc
        self._cb = op_host
        fb.init(ctypes.cast(op_host, ctypes.c_void_p).value, ctypes.addressof(self.buf_in), ctypes.addressof(self.buf_out))


def board_of(case):
    return PlainBoard([tuple(m) for m in case["moves"]], list(case["board"]), [set(s) for s in case["qstructs"]])


def replay(cases, fb):
    """Every case: the 41 packed bytes, the attributes afterwards, WHICH list / set objects survive."""
    h = Harness(fb)
    for k, c in enumerate(cases):
        b = board_of(c)
        olds = (b.moves, b.board, b.qstructs, list(b.qstructs))
        h.reply, h.rc, h.seen = bytes.fromhex(c["reply"]), 0, None
        assert fb.board_op(b, c["op"], c["lo"], c["hi"], c["bit"], c["drop"], 0) == 0, k
        assert h.seen[:41] == bytes.fromhex(c["want_in"]), k
        w = c["after"]
        assert b.moves == [tuple(m) for m in w["moves"]] and b.board == w["board"], k
        assert [sorted(s) for s in b.qstructs] == w["qstructs"] and list(b._win[1:]) == w["win"][1:], k
        assert b._win[0] == tuple(w["win"][0]) and type(b._win) is tuple, k
        assert all(type(m) is tuple and all(type(x) is int for x in m) for m in b.moves), k
        assert all(type(s) is set for s in b.qstructs), k
        assert b.moves is olds[0] and b.board is olds[1] and b.qstructs is olds[2], k
        alias = [next((j for j, t in enumerate(olds[3]) if t is s), None) for s in b.qstructs]
        assert alias == c["alias"], (k, alias, c["alias"])
    # attributes of unusual types are declined (-100), nothing is sent: the Python path then deals with them
    base = cases[0]
    for mutate in (lambda b: setattr(b, "board", tuple(b.board)),
                   lambda b: setattr(b, "qstructs", [frozenset({0, 1}), [2, 3]]),
                   lambda b: setattr(b, "moves", [(0.0, 1, 0)]),
                   lambda b: setattr(b, "qstructs", [set(range(12))]),           # not a set of squares
                   lambda b: setattr(b, "qstructs", [{0, -3}]),
                   lambda b: setattr(b, "qstructs", [{0, "x"}]),
                   lambda b: setattr(b, "board", [0] * 5),
                   lambda b: setattr(b, "moves", [(1,)]),
                   lambda b: delattr(b, "moves")):
        b = board_of(base)
        mutate(b)
        h.seen = None
        assert fb.board_op(b, 0, 0, 1, 0, False, 0) == -100 and h.seen is None
    # an error code of the library comes back as it is, attributes untouched
    b = board_of(base)
    snap = (list(b.moves), list(b.board), [set(s) for s in b.qstructs])
    h.rc = 719
    assert fb.board_op(b, 0, 0, 1, 0, False, 0) == 719
    assert (b.moves, b.board, b.qstructs) == snap
    h.rc = 0
    return len(cases)


def loop(cases, fb, calls):
    """`calls` board_op calls cycling through the cases (fresh attribute objects every call, as a caller makes them):
    returns what grew between the end of a warm-up and the end of the run."""
    h = Harness(fb)
    rng = random.Random(5)
    replies = [bytes.fromhex(c["reply"]) for c in cases]

    def run(n):
        for i in range(n):
            c = cases[i % len(cases)]
            b = board_of(c)
            h.reply = replies[rng.randrange(len(replies))]        # any reply on any board: more aliasing paths
            fb.board_op(b, c["op"], c["lo"], c["hi"], c["bit"], c["drop"], 0)
            if i % 997 == 0:                                       # the decline path too
                b.board = tuple(b.board)
                fb.board_op(b, 0, 0, 1, 0, False, 0)
    run(max(2000, calls // 10))
    gc.collect()
    blocks0, objs0, rss0 = sys.getallocatedblocks(), len(gc.get_objects()), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    run(calls)
    gc.collect()
    blocks1, objs1, rss1 = sys.getallocatedblocks(), len(gc.get_objects()), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    return {"calls": calls, "blocks_growth": blocks1 - blocks0, "gc_objects_growth": objs1 - objs0, "maxrss_growth_kb": rss1 - rss0}


if __name__ == "__main__":
    fb = load(sys.argv[1])
    with open(sys.argv[2]) as f:
        cases = json.load(f)
    out = {"replayed": replay(cases, fb)}
    if len(sys.argv) > 3:
        out["loop"] = loop(cases, fb, int(sys.argv[3]))
    print(json.dumps(out))
    print("ok")
