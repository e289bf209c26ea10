"""The examples run (each is a subprocess, as a user starts them): the root-level PUCT bandit on qttt_expand_rollout, and
the gym loop — in place, with fresh tensors per step (the default VecEnv.step), and as one captured hipGraph."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_root_ucb_search_on_expand_rollout_beats_random():
    """The operator composes into a search loop with torch ops only (select -> one launch -> update): a root-level PUCT
    bandit (mcts.py:281-285) over qttt_expand_rollout wins clearly more often as P1 than a random P1 does (52.8 % + its
    share of the double-line games); measured 93.5 % (2 048 games, 72 iterations x 4 playouts per child)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "ucb_selfplay.py"), "--games", "512", "--iters", "48",
                          "--sims", "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    pct = float(out.stdout.split("(")[2].split("%")[0])
    assert pct > 86.0, out.stdout


@pytest.mark.parametrize("mode", [(), ("--fresh",), ("--graph",)])
def test_gym_loop_example_runs_in_place_with_fresh_tensors_and_as_a_graph(mode):
    """examples/gym_loop.py: the reference's `obs, r, terminated, truncated, info = env.step(action)` loop for N boards
    with a policy that reads the observation on the GPU — in place (copy_obs=False), through the default step()
    (fresh tensors per step) and as ONE captured agent step replayed.  Same seed, same policy: the three modes finish
    the same number of episodes."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "gym_loop.py"), "--boards", "8192", "--steps", "40", *mode],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]
    assert line.startswith("8192 boards x 40 steps") and "episodes finished" in line, line
    episodes = int(line.split(";")[1].split("episodes")[0])
    lines = int(line.split(",")[-1].split("with")[0])
    assert 8192 * 40 // 9 * 0.9 < episodes < 8192 * 40 // 5 and 0 < lines <= episodes, line
    test_gym_loop_example_runs_in_place_with_fresh_tensors_and_as_a_graph.seen.add((episodes, lines))
    assert len(test_gym_loop_example_runs_in_place_with_fresh_tensors_and_as_a_graph.seen) == 1


test_gym_loop_example_runs_in_place_with_fresh_tensors_and_as_a_graph.seen = set()
