"""The single-record path under the Board / Env façades (include/qttt.h: qttt_board_op, _sync, _host): attributes
mutated in place as the reference mutates them (board.py:19,25,53-69), the stamp poll against the synchronising form,
the poll's fall-back behind a long kernel, and the reference's golden episodes through Board.make_move with the
mailbox at several idle windows and switched off."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------- façade identity
def test_board_attributes_are_mutated_in_place():
    """board.py:19,25 append to .moves, :53-54 write into .board, :56-69 pop / assign / append on .qstructs and
    grow a set with .add: a caller holding the list (or set) objects sees every move."""
    from qtttgym_amd import Board, QEvalClassic

    class Fixed(QEvalClassic):
        def choose(self, lo, hi):
            return hi

    b = Board(Fixed())
    mv, bd, qs = b.moves, b.board, b.qstructs
    b.make_move((0, 1))
    assert b.moves is mv and mv == [(0, 1, 0)] and b.qstructs is qs and qs == [{0, 1}] and b.board is bd
    s0 = qs[0]
    b.make_move((2, 1))
    assert mv == [(0, 1, 0), (1, 2, 1)] and qs[0] is s0 and s0 == {0, 1, 2}          # board.py:68-69: grown in place
    b.make_move((3, 4))
    s1 = qs[1]
    assert qs[0] is s0 and s1 == {3, 4}
    b.make_move((5, 6))
    s2 = qs[2]
    b.make_move((2, 3))                                                              # board.py:58-61: union, a new set
    assert b.qstructs is qs and qs == [{0, 1, 2, 3, 4}, {5, 6}] and qs[1] is s2 and qs[0] is not s0
    b.make_move((0, 4))                                                              # closes a cycle: the component goes
    assert b.qstructs is qs and qs == [{5, 6}] and qs[0] is s2
    assert b.board is bd and bd == [0, 1, 4, 2, 5, -1, -1, -1, -1] and b.moves is mv and len(mv) == 6
    with pytest.raises(Exception):
        b.make_move((0, 5))
    assert len(mv) == 6
    # update_qstructs on its own (the caller appended the move, board.py:19-20) and make_moves keep identity too
    b.moves.append((7, 8, len(b.moves)))
    b.update_qstructs((7, 8))
    assert b.moves is mv and len(mv) == 7 and qs == [{5, 6}, {7, 8}] and qs[0] is s2
    c = Board(Fixed())
    cm, cq = c.moves, c.qstructs
    assert Board.make_moves([c], [(4, 8)]) == [None] and c.moves is cm and cm == [(4, 8, 0)] and c.qstructs is cq


def test_board_op_host_polls_the_stamp_and_equals_board_op_sync():
    """qttt_board_op_host (records in pinned host memory, completion by polling byte 63 of the out records) gives the
    records qttt_board_op_sync gives, for one record, a node's 36 actions, and a batch beyond the polling limit."""
    from qtttgym_amd import Board, QEvalClassic, _native
    from qtttgym_amd.board import _Staging
    L = _native.lib()
    parent = Board(QEvalClassic())
    for mv in ((0, 1), (1, 2), (3, 4), (2, 3), (5, 6), (0, 4)):         # the last one closes a cycle: squares 0..4 go classical
        parent.make_move(mv)
    pairs = [(i, j) for i in range(9) for j in range(i + 1, 9)]
    s = torch.cuda.current_stream().cuda_stream
    for n in (1, 36, 300):
        recs = [_Staging.pack(parent, _native.OP_MAKE_MOVE, *pairs[k % 36], k & 1) for k in range(n)]
        t_in = torch.zeros(64 * n, dtype=torch.uint8).pin_memory()
        t_in.numpy()[:] = np.frombuffer(b"".join(r + bytes(23) for r in recs), dtype=np.uint8)
        a = torch.full((64 * n,), 7, dtype=torch.uint8).pin_memory()
        b = torch.full((64 * n,), 9, dtype=torch.uint8).pin_memory()
        assert L.qttt_board_op_sync(t_in.data_ptr(), a.data_ptr(), n, s) == 0
        assert L.qttt_board_op_host(t_in.data_ptr(), b.data_ptr(), n, s) == 0
        ra, rb = a.numpy().reshape(n, 64), b.numpy().reshape(n, 64)
        assert np.array_equal(ra[:, :42], rb[:, :42]) and np.array_equal(ra[:, 44:51], rb[:, 44:51])   # (bytes 42, 43 are padding)
        assert (rb[:, 63] == (1 if n <= 256 else 9)).all()          # stamped when polled; untouched on the fallback path
        if n > 1:                                                       # both kinds of answers are in the batch
            assert (ra[:, 41] == 1).sum() > 0 and (ra[:, 41] == 0).sum() > 0
    assert L.qttt_board_op_host(None, None, 1, s) == -1 and L.qttt_board_op_host(None, None, 0, s) == 0


_POLL_TIMEOUT_SCRIPT = r"""
import sys, time
sys.path.insert(0, %r)
import torch
from qtttgym_amd import Board, QEvalClassic, VecEnv
ref = Board(QEvalClassic())
ref.make_move((0, 1))
big = VecEnv(1 << 20, seed=3, auto_reset=True)
big.step_random_many(64)
torch.cuda.synchronize()
b = Board(QEvalClassic())
t0 = time.perf_counter()
for _ in range(24):                                     # ~5.5 ms of work on the current stream, not waited for
    big.step_random_many(64)
queued = time.perf_counter() - t0
b.make_move((0, 1))                                     # its launch queues behind them
waited = time.perf_counter() - t0
assert queued < 0.004, "the launches were not asynchronous (%%.1f ms): nothing was queued ahead" %% (queued * 1e3)
assert waited > 0.003, "the record came back before the queued work could have finished (%%.2f ms)" %% (waited * 1e3)
assert b.moves == ref.moves == [(0, 1, 0)] and b.board == ref.board and b.qstructs == ref.qstructs == [{0, 1}]
b.make_move((0, 1))                                     # and the facade keeps working afterwards (poll path again)
assert sorted(b.board[:2]) == [0, 1] and b.qstructs == []
print("ok")
"""


def _run_script(script, **env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-2000:])
    return out


def test_board_op_host_falls_back_to_the_stream_when_the_poll_times_out():
    """include/qttt.h, qttt_board_op_host on its LAUNCH path (QTTT_BOARD_MAILBOX_US=0): with several ms of kernels queued
    ahead on the stream the 2 ms poll gives up and the call synchronises the stream instead — same records, stamped."""
    _run_script(_POLL_TIMEOUT_SCRIPT % ROOT, QTTT_BOARD_MAILBOX_US="0")


_MAILBOX_SCRIPT = r"""
import sys, time, json, random
sys.path.insert(0, %r)
import numpy as np
import torch
from qtttgym_amd import Board, QEvalClassic, _native
from qtttgym_amd import board as board_mod
g = np.load(%r)
class Bits(QEvalClassic):
    def __init__(self, bits): self.bits, self.k = bits, 0
    def choose(self, lo, hi):
        b = int(self.bits[self.k]); return hi if b else lo
kinds = list(g["kind"])
E, T = g["actions"].shape[0], g["actions"].shape[1]
n_calls = 0
for e in list(range(0, E, max(1, E // 150)))[:150]:
    ev = Bits(g["bits"][e]); b = Board(ev)
    for t in range(T):
        a = (int(g["actions"][e, t, 0]), int(g["actions"][e, t, 1]))
        ev.k = t
        try:
            b.make_move(a); n_calls += 1
        except Exception as ex:
            if isinstance(ex, _native.QtttNativeError): raise
        assert b.board == [int(x) for x in g["board"][e, t]], (e, t)
        assert len(b.moves) == int(g["n_moves"][e, t]), (e, t)
        if e %% 7 == 0 and t %% 3 == 0:
            time.sleep(0.0006)                       # longer than the idle window: the wave has left, the next call relaunches it
# a device-wide synchronise right after a call waits for the resident wave at most its idle window (+ slack)
b = Board(QEvalClassic()); b.make_move((0, 1))
t0 = time.perf_counter(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
assert dt < 0.005, dt
print(json.dumps({"calls": n_calls, "sync_after_call_ms": dt * 1e3, "fast": board_mod._stage().fast is not None}))
print("ok")
"""


@pytest.mark.parametrize("mailbox_us", ["100", "0", "20"])
def test_board_facade_on_the_golden_episodes_with_and_without_the_mailbox(mailbox_us):
    """The single-board façade through the bounded mailbox (default window, a short one) and through the launch path:
    150 golden episodes of the reference, step by step, with pauses longer than the idle window in between (the
    resident wave leaves and is launched again), and a device-wide synchronise right after a call."""
    golden = os.path.join(ROOT, "tests", "golden", "step_traces.npz")
    out = _run_script(_MAILBOX_SCRIPT % (ROOT, golden), QTTT_BOARD_MAILBOX_US=mailbox_us)
    info = json.loads(out.stdout.strip().splitlines()[-2])
    assert info["calls"] > 500
    # qtttgym_amd/_fastboard.so is optional (build() goes on without it where Python.h or gcc are missing): in use iff built
    assert info["fast"] is os.path.exists(os.path.join(ROOT, "qtttgym_amd", "_fastboard.so"))
