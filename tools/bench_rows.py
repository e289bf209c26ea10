#!/usr/bin/env python3
"""Timing of the SURVEY §8(f) rows beside the headline bench (not the judged line):
batched expand at BASELINE config 5's batch (65 536), fused rollout and encode at 1 048 576.
Prints one JSON object per row."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from qtttgym_amd import VecEnv  # noqa: E402
from qtttgym_amd import recommended_env  # noqa: E402
recommended_env(apply=True)   # HIP_FORCE_DEV_KERNARG=1 etc., before the first HIP call


def timed(fn, reps=20, warm=3, min_s=float(os.environ.get("QTTT_ROWS_MIN_S", "0.02"))):
    """Seconds per call by HIP events; the window is stretched to >= min_s (a window of 20 calls of a 5 us kernel is
    0.1 ms: the GPU has not left its idle clocks by then — round 4 read 13 us for a 6 us kernel that way) and the
    better of two windows is kept."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3
        if dt < min_s:
            reps = int(reps * min(20.0, max(2.0, 1.2 * min_s / max(dt, 1e-6))))
            continue
        best = dt / reps if best is None else min(best, dt / reps)
    return best if best is not None else dt / reps


def midgame(n, plies, seed=1):
    env = VecEnv(n, seed=seed)
    for _ in range(plies):
        env.step_raw(env.sample_actions())
    return env


def main():
    out = []
    # expand: read 20 B + 1 B, write 2 x 20 B + 1 + 2 + 2 + 16 + 16 = 78 B per pair
    n = 65536
    env = midgame(n, 4)
    act = torch.randint(0, 36, (n,), dtype=torch.uint8, device="cuda")
    # time the raw ABI call with preallocated outputs
    L, st = env._lib, env.state
    c0, c1 = torch.empty_like(st), torch.empty_like(st)
    nch = torch.empty(n, dtype=torch.uint8, device="cuda")
    w = torch.empty((n, 2), dtype=torch.int8, device="cuda")
    tm = torch.empty((n, 2), dtype=torch.bool, device="cuda")
    lg = torch.empty((n, 2), dtype=torch.int64, device="cuda")
    ky = torch.empty((n, 2), dtype=torch.int64, device="cuda")
    sk = torch.empty((n, 2), dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    t = timed(lambda: L.qttt_expand(st.data_ptr(), act.data_ptr(), c0.data_ptr(), c1.data_ptr(), nch.data_ptr(),
                                    w.data_ptr(), tm.data_ptr(), lg.data_ptr(), None, sk.data_ptr(), n, s))
    t_py = timed(lambda: L.qttt_expand(st.data_ptr(), act.data_ptr(), c0.data_ptr(), c1.data_ptr(), nch.data_ptr(),
                                       w.data_ptr(), tm.data_ptr(), lg.data_ptr(), ky.data_ptr(), sk.data_ptr(), n, s))
    out.append({"row": "expand", "boards": n, "us": t * 1e6, "us_with_python_keys": t_py * 1e6, "expansions_per_s": n / t,
                "reference_cpu_step_calls_per_s": 4900})
    for n in (65536, 1 << 20):
        env = midgame(n, 0)
        res = torch.empty(n, dtype=torch.int8, device="cuda")
        pl = torch.empty(n, dtype=torch.uint8, device="cuda")
        t = timed(lambda: env._lib.qttt_rollout(env.state.data_ptr(), 1, 0, 0, res.data_ptr(), pl.data_ptr(), 0, n, s))
        plies = float(pl.float().mean())
        out.append({"row": "rollout_from_empty_board", "boards": n, "us": t * 1e6, "playouts_per_s": n / t,
                    "mean_plies": plies, "env_steps_per_s": n * plies / t})
    # MCTS._rollout's simulation loop (mcts.py:170-176): 10 playouts per leaf, one launch against ten
    n = 65536
    env = midgame(n, 3)
    rm = env.rollout_many(10)
    ro = env.rollout()
    t_many = timed(lambda: env.rollout_many(10, out=rm), reps=20)
    t_loop = timed(lambda: [env.rollout(step_idx0=16 * k, out=ro) for k in range(10)], reps=10)
    out.append({"row": "rollout_10_simulations_per_leaf", "boards": n, "us_one_launch": t_many * 1e6,
                "us_ten_launches": t_loop * 1e6, "playouts_per_s_one_launch": 10 * n / t_many})
    n = 1 << 20
    env = midgame(n, 5)
    vec = torch.empty((n, 18, 10), dtype=torch.float32, device="cuda")
    mask = torch.empty((n, 36), dtype=torch.bool, device="cuda")
    t = timed(lambda: env._lib.qttt_encode(env.state.data_ptr(), vec.data_ptr(), mask.data_ptr(), n, s))
    out.append({"row": "encode", "boards": n, "us": t * 1e6, "boards_per_s": n / t,
                "output_GBps": n * (720 + 36) / t / 1e9})
    # cold kernels a per-step RL loop calls: observation, Board attributes, check_win
    n = 1 << 20
    env = midgame(n, 5)
    t = timed(lambda: env.observ(), reps=10)
    o = env.observ()
    t_raw = timed(lambda: env._lib.qttt_observe(env.state.data_ptr(), o["classical"].data_ptr(), o["q_states_p1"].data_ptr(),
                                                o["q_states_p1_len"].data_ptr(), o["q_states_p2"].data_ptr(),
                                                o["q_states_p2_len"].data_ptr(), o["turn"].data_ptr(), n, s), reps=20)
    out.append({"row": "observe", "boards": n, "us": t * 1e6, "us_kernel_only": t_raw * 1e6, "output_bytes_per_board": 30})
    t = timed(lambda: env.export_boards(), reps=10)
    ex = env.export_boards()
    t_out = timed(lambda: env.export_boards(out=ex), reps=20)
    t_raw = timed(lambda: env._lib.qttt_export(env.state.data_ptr(), ex["moves"].data_ptr(), ex["n_moves"].data_ptr(),
                                               ex["board"].data_ptr(), ex["qmask"].data_ptr(), ex["n_q"].data_ptr(), n, s), reps=20)
    out.append({"row": "export", "boards": n, "us": t * 1e6, "us_out_reuse": t_out * 1e6, "us_kernel_only": t_raw * 1e6,
                "output_bytes_per_board": 37})
    env2 = VecEnv(n, seed=9)
    t_imp = timed(lambda: env2._lib.qttt_import(env2.state.data_ptr(), ex["moves"].data_ptr(), ex["n_moves"].data_ptr(),
                                                ex["board"].data_ptr(), ex["qmask"].data_ptr(), ex["n_q"].data_ptr(), n, s), reps=20)
    out.append({"row": "import", "boards": n, "us_kernel_only": t_imp * 1e6, "input_bytes_per_board": 37})
    t = timed(lambda: env.check_win(), reps=10)
    p1, p2 = env.check_win()
    t_raw = timed(lambda: env._lib.qttt_check_win(env.state.data_ptr(), p1.data_ptr(), p2.data_ptr(), n, s), reps=20)
    t_out = timed(lambda: env.check_win(out=(p1, p2)), reps=20)
    out.append({"row": "check_win", "boards": n, "us": t * 1e6, "us_out_reuse": t_out * 1e6, "us_kernel_only": t_raw * 1e6,
                "output_bytes_per_board": 2})
    t = timed(lambda: env.node_info(python_key=False), reps=10)
    ni = env.node_info()
    t_raw = timed(lambda: env._lib.qttt_node_info(env.state.data_ptr(), ni["winner"].data_ptr(), ni["terminal"].data_ptr(),
                                                  ni["legal"].data_ptr(), None, ni["state_key"].data_ptr(), n, s), reps=20)
    t_py = timed(lambda: env._lib.qttt_node_info(env.state.data_ptr(), ni["winner"].data_ptr(), ni["terminal"].data_ptr(),
                                                 ni["legal"].data_ptr(), ni["key"].data_ptr(), ni["state_key"].data_ptr(), n, s), reps=20)
    t_key = timed(lambda: env._lib.qttt_node_info(env.state.data_ptr(), None, None, None, None, ni["state_key"].data_ptr(), n, s), reps=20)
    ni_native = {k: v for k, v in ni.items() if k != "key"}
    t_out = timed(lambda: env.node_info(out=ni_native), reps=20)
    out.append({"row": "node_info", "boards": n, "us": t * 1e6, "us_out_reuse": t_out * 1e6, "us_kernel_only": t_raw * 1e6,
                "us_kernel_only_with_python_key": t_py * 1e6, "us_kernel_only_state_key_alone": t_key * 1e6,
                "output_bytes_per_board": 18})
    # fused replay (QTTT_FLAG_FUSED): T steps per launch, boards in registers
    for n in (4096, 262144, 1 << 20):
        T = 64
        rec = VecEnv(n, seed=2, auto_reset=True)
        actions = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
        for t_ in range(T):
            rec.sample_actions(out=actions[t_])
            rec.step_raw(actions[t_])
        env = VecEnv(n, seed=2, auto_reset=True)
        r = torch.empty((T, n), dtype=torch.float32, device="cuda")
        tm = torch.empty((T, n), dtype=torch.bool, device="cuda")

        def run(fused):
            env.reset_raw()
            env.step_many(actions, reward=r, terminated=tm, fused=fused)
        tf = timed(lambda: run(True), reps=10)
        tu = timed(lambda: run(False), reps=10)
        out.append({"row": "step_many_T64_every_output_kept", "boards": n, "fused_us_per_step": tf * 1e6 / T,
                    "unfused_us_per_step": tu * 1e6 / T, "fused_steps_per_s": n * T / tf,
                    "unfused_steps_per_s": n * T / tu})
    # fused random-policy stepping (qttt_step_random_many): T steps per launch, in-kernel policy
    for n in (4096, 65536, 262144, 1 << 20):
        T = 64
        env = VecEnv(n, seed=2, auto_reset=True)
        r = torch.empty((T, n), dtype=torch.float32, device="cuda")
        tm = torch.empty((T, n), dtype=torch.bool, device="cuda")
        a = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
        t_keep = timed(lambda: env.step_random_many(T, actions_out=a, reward=r, terminated=tm), reps=10)
        t_rt = timed(lambda: env.step_random_many(T, reward=r, terminated=tm), reps=10)
        t_last = timed(lambda: env.step_random_many(T), reps=10)
        t_one = timed(lambda: env.step_random(), reps=64)
        out.append({"row": "step_random_many_T64", "boards": n, "us_per_step_all_outputs_kept": t_keep * 1e6 / T,
                    "us_per_step_reward_terminated_kept": t_rt * 1e6 / T, "us_per_step_last_only": t_last * 1e6 / T,
                    "us_per_step_launch_by_launch": t_one * 1e6, "steps_per_s_all_outputs_kept": n * T / t_keep,
                    "steps_per_s_last_only": n * T / t_last})
    # launch-per-step on small batches: eager Python loop against a hipGraph of the same launches (VecEnv.capture)
    for n in (4096, 65536, 262144):
        T = 32
        e1 = VecEnv(n, seed=2, auto_reset=True)
        t_eager = timed(lambda: [e1.step_random() for _ in range(T)], reps=10)
        e2 = VecEnv(n, seed=2, auto_reset=True)
        g = e2.capture(T, "random")
        t_graph = timed(lambda: g.replay(), reps=10)
        e3 = VecEnv(n, seed=2, auto_reset=True)
        a = torch.zeros((1, n, 2), dtype=torch.uint8, device="cuda")
        g1 = e3.capture(1, "observe", actions=a)
        t_g1 = timed(lambda: g1.replay(), reps=64)
        e4 = VecEnv(n, seed=2, auto_reset=True)
        t_e1 = timed(lambda: e4.step_observe_raw(a[0]), reps=64)
        out.append({"row": "launch_per_step_eager_vs_hipgraph", "boards": n, "steps_per_graph": T,
                    "step_random_eager_us_per_step": t_eager * 1e6 / T, "step_random_graph_us_per_step": t_graph * 1e6 / T,
                    "step_observe_eager_us_per_step": t_e1 * 1e6, "step_observe_graph_of_one_us_per_step": t_g1 * 1e6})
    # expand and node_info at a batch that fills the chip, turn() = n_moves alone
    n = 1 << 20
    env = midgame(n, 4)
    act = torch.randint(0, 36, (n,), dtype=torch.uint8, device="cuda")
    ex = env.expand(act)
    t = timed(lambda: env.expand(act, out=ex), reps=20)
    t_alloc = timed(lambda: env.expand(act), reps=10)
    out.append({"row": "expand", "boards": n, "us_out_reuse": t * 1e6, "us_allocating": t_alloc * 1e6,
                "expansions_per_s": n / t, "algorithmic_bytes_per_pair": 17 + 32 + 1 + 4 + 32})
    tn = env.turn()
    t = timed(lambda: env.turn(out=tn), reps=20)
    out.append({"row": "turn", "boards": n, "us": t * 1e6, "algorithmic_bytes_per_board": 9})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
