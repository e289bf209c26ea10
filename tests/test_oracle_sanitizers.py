"""SURVEY.md §5 (sanitizers): the CPU oracle, compiled with AddressSanitizer + UBSan, replays the
golden traces recorded from the reference (every state field, reward bits, check_win, observation)
and exercises the expand / encode / hash helpers on the same states.  CPU build only: GPU
sanitizers are not available on the pool."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_replay_under_asan_ubsan(golden, tmp_path):
    E, T = golden["bits"].shape
    rec = np.dtype([("actions", "u1", (2,)), ("bit", "u1"), ("board", "i1", (9,)), ("n_moves", "u1"),
                    ("moves", "u1", (9, 2)), ("n_q", "u1"), ("qmask", "<u2", (4,)), ("reward_bits", "<u4"),
                    ("terminated", "u1"), ("p1", "i1"), ("p2", "i1"), ("q_p1", "u1", (5, 2)), ("l1", "u1"),
                    ("q_p2", "u1", (4, 2)), ("l2", "u1"), ("turn", "u1")])
    assert rec.itemsize == 2 + 1 + 9 + 1 + 18 + 1 + 8 + 4 + 1 + 2 + 10 + 1 + 8 + 1 + 1
    a = np.zeros((T, E), dtype=rec)
    g = golden
    a["actions"] = g["actions"].transpose(1, 0, 2)
    a["bit"] = g["bits"].T
    a["board"] = g["board"].transpose(1, 0, 2)
    a["n_moves"] = g["n_moves"].T
    a["moves"] = g["moves"].transpose(1, 0, 2, 3)
    a["n_q"] = g["n_q"].T
    a["qmask"] = g["qmask"].transpose(1, 0, 2)
    a["reward_bits"] = g["reward"].astype(np.float32).view(np.uint32).T
    a["terminated"] = g["terminated"].T
    a["p1"], a["p2"] = g["p1_round"].T, g["p2_round"].T
    a["q_p1"], a["l1"] = g["q_p1"].transpose(1, 0, 2, 3), g["q_p1_len"].T
    a["q_p2"], a["l2"] = g["q_p2"].transpose(1, 0, 2, 3), g["q_p2_len"].T
    a["turn"] = g["turn"].T
    dump = tmp_path / "golden.bin"
    with open(dump, "wb") as f:
        f.write(np.array([E, T], dtype="<u4").tobytes())
        f.write(a.tobytes())
    exe = tmp_path / "san_replay"
    cc = os.environ.get("CC", "gcc")
    build = subprocess.run([cc, "-O1", "-g", "-std=c11", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "oracle"),
                            "-o", str(exe), os.path.join(ROOT, "oracle", "san_replay.c"),
                            os.path.join(ROOT, "oracle", "qttt_oracle.c")], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([str(exe), str(dump)], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "san_replay ok: %d episodes x %d steps" % (E, T) in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
