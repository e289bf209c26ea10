"""CPU-side checks: the C-ABI library loads and exports every symbol include/qttt.h declares,
the binding table matches the header, the product never imports the oracle, host-side helpers."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "qttt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qttt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build_hip()
    from qtttgym_amd import _native
    L = ctypes.CDLL(_native.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(_native.SIGNATURES) == syms
    assert L.qttt_abi_version() == _native.ABI_VERSION


def test_state_bytes_and_hash_are_host_callable():
    from qtttgym_amd import _native
    import oracle
    L = _native.lib()
    assert L.qttt_state_bytes(0) == 0
    assert L.qttt_state_bytes(1 << 20) == 16 * (1 << 20)
    assert L.qttt_state_bytes(-1) < 0
    for seed, bid, step in [(0, 0, 0), (1, 123456789, 7), (2**63 + 5, 2**33 + 17, 2**31 + 3)]:
        assert L.qttt_hash(seed, bid, step) == oracle.hash64(seed, bid, step)


def test_env_record_call_validates_before_touching_the_device():
    """qttt_env_step (the struct qttt_env form of the per-step calls): argument errors come back as
    codes without any device work."""
    import ctypes
    from qtttgym_amd import _native
    L = _native.lib()
    assert ctypes.sizeof(_native.EnvRecord) == 112                     # include/qttt.h struct qttt_env, LP64
    assert L.qttt_env_step(None, None, None, 0, _native.ENV_STEP, None) == -1          # QTTT_ERR_NULL
    rec = _native.EnvRecord(n=0)
    assert L.qttt_env_step(ctypes.byref(rec), None, None, 0, 7, None) == -2            # unknown mode: QTTT_ERR_SIZE
    assert L.qttt_env_step(ctypes.byref(rec), None, None, 0, _native.ENV_STEP, None) == 0    # n = 0: nothing to do
    rec.n = -5
    assert L.qttt_env_step(ctypes.byref(rec), None, None, 0, _native.ENV_STEP_RANDOM, None) == -2
    assert L.qttt_env_step(ctypes.byref(rec), None, None, 0, _native.ENV_SAMPLE, None) == -2
    rec.n = 8
    assert L.qttt_env_step(ctypes.byref(rec), None, None, 0, _native.ENV_SAMPLE, None) == -1          # no state, no actions
    assert L.qttt_env_step(ctypes.byref(rec), None, None, 0, _native.ENV_STEP_OBSERVE, None) == -1     # no observation buffers
    assert L.qttt_counter_add(None, 1, None) == -1
    # the fused random stepper and the nullable-output entries: argument errors, no device work
    assert L.qttt_step_random_many(None, 0, 0, 0, 0, None, None, None, 0, None, 8, 4, None) == -1
    assert L.qttt_step_random_many(None, 0, 0, 0, 0, None, None, None, 0, None, -1, 4, None) == -2
    assert L.qttt_step_random_many(None, 0, 0, 0, 0, None, None, None, 0, None, 8, 0, None) == 0             # no steps: nothing to do
    assert L.qttt_export(None, None, None, None, None, None, 8, None) == -1
    assert L.qttt_node_info(None, None, None, None, None, None, 8, None) == -1
    assert L.qttt_expand(None, None, None, None, None, None, None, None, None, None, 8, None) == -1
    assert L.qttt_expand_rollout(None, None, None, None, None, None, None, None, None, None, 0, 0, 0, 1, None, None, 8, None) == -1
    assert L.qttt_expand_rollout(None, None, None, None, None, None, None, None, None, None, 0, 0, 0, 0, None, None, 8, None) == -2
    assert L.qttt_expand_rollout(None, None, None, None, None, None, None, None, None, None, 0, 0, 0, 129, None, None, 8, None) == -2
    # the native position key is host-callable: the empty board's key is the mix of two zero words
    assert L.qttt_state_key(0, 0) == 0 and L.qttt_state_key(1 << 40, 0) != 0
    assert L.qttt_state_key(1 << 40, 5) == L.qttt_state_key((1 << 40) | (1 << 63) | (0xF << 44), 5 | (0x1FF << 32))
    assert L.qttt_export(None, None, None, None, None, None, 0, None) == 0


def test_header_is_plain_c_and_the_env_record_layout_matches_the_binding(tmp_path):
    """include/qttt.h must stay a C header (the reference side would bind it through ctypes / cffi / cgo):
    gcc -std=c99 -pedantic compiles it, and sizeof / offsetof of struct qttt_env as a C compiler sees them are
    the ones of the ctypes mirror in qtttgym_amd/_native.py."""
    import ctypes
    import subprocess
    from qtttgym_amd import _native
    src = tmp_path / "probe.c"
    fields = [f[0] for f in _native.EnvRecord._fields_]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "qttt.h"\nint main(void) {\n'
                   '  printf("%zu", sizeof(qttt_env));\n'
                   + "".join('  printf(" %%zu", offsetof(qttt_env, %s));\n' % f for f in fields)
                   + '  printf(" %d %d", QTTT_ABI_VERSION, QTTT_BOARD_RECORD_BYTES);\n  return 0;\n}\n')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(_native.EnvRecord)
    assert out[1:1 + len(fields)] == [getattr(_native.EnvRecord, f).offset for f in fields]
    assert out[-2] == _native.ABI_VERSION and out[-1] == _native.BOARD_RECORD_BYTES


def test_launch_shape_table_and_overrides():
    """qttt_step_launch_shape / qttt_set_tuning are host logic (no device work): the by-batch-size table
    of DESIGN.md §2 and its overrides."""
    from qtttgym_amd import _native
    L = _native.lib()
    try:
        assert L.qttt_set_tuning(0, 0) == 0
        expect = {1: (1, 256), 4096: (1, 256), 262144: (1, 256), 458752: (1, 256), 458753: (1, 1024),
                  524288: (1, 1024), 524289: (2, 512), 917503: (2, 512), 917504: (2, 1024), 1048576: (2, 1024),
                  1572864: (2, 1024), 1572865: (2, 256), 2097152: (2, 256), 1 << 25: (2, 256)}
        for n, shape in expect.items():
            assert _native.step_launch_shape(n) == shape, n
        assert L.qttt_set_tuning(4, 1024) == 0 and _native.step_launch_shape(1 << 20) == (4, 512)   # 4 per lane: 512 only
        assert _native.step_launch_shape(1 << 20, observe=True) == (2, 1024)    # the observation tiles take <= 2 per lane
        # only boards-per-lane given: the workgroup size stays the table's choice for the batch
        assert L.qttt_set_tuning(2, 0) == 0 and _native.step_launch_shape(1 << 20) == (2, 1024)
        assert _native.step_launch_shape(262144) == (2, 256)
        assert L.qttt_set_tuning(1, 0) == 0 and _native.step_launch_shape(1 << 20) == (1, 1024)
        assert L.qttt_set_tuning(0, 256) == 0 and _native.step_launch_shape(1 << 20) == (2, 256)
        # a call's own QTTT_FLAG_SHAPE bits win over the process-wide default
        assert _native.step_launch_shape(1 << 20, _native.flag_shape(1, 512)) == (1, 512)
        assert _native.step_launch_shape(1 << 20, _native.flag_shape(4, 0)) == (4, 512)
        assert _native.step_launch_shape(1 << 20, _native.flag_shape(4, 0), observe=True) == (2, 1024)
        assert _native.step_launch_shape(1 << 20, 1) == (2, 256)                 # AUTO_RESET alone: no shape bits
        assert L.qttt_set_tuning(0, 0) == 0
        assert _native.step_launch_shape(4096, _native.flag_shape(0, 1024)) == (1, 1024)
        assert _native.flag_shape(2, 1024) == (2 << 8) | (3 << 12) and _native.flag_shape() == 0
        for bad in ((3, 0), (8, 0), (-1, 0), (2, 128), (2, 2048), (0, -256)):
            assert L.qttt_set_tuning(*bad) == -2
        with pytest.raises(RuntimeError):
            _native.step_launch_shape(-1)
    finally:
        assert L.qttt_set_tuning(0, 0) == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "qtttgym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "qttt_oracle" not in src and "libqttt_oracle" not in src, f


def test_no_cpu_fallback():
    import torch
    from qtttgym_amd import VecEnv, _native
    with pytest.raises(_native.QtttNativeError):
        VecEnv(8, device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(_native.QtttNativeError):
            VecEnv(8)


def test_spaces_match_reference_declarations():
    from qtttgym_amd.spaces import reference_action_space, reference_observation_space
    a = reference_action_space()
    assert len(a) == 2 and a[0].n == 9 and a[1].n == 9          # env.py:19
    o = reference_observation_space()
    assert set(o.keys()) == {"q_states_p1", "q_states_p2", "classical", "turn"}
    assert o["q_states_p1"].max_len == 5 and o["q_states_p2"].max_len == 4   # env.py:20-21
    assert o["classical"].shape == (9,) and o["turn"].n == 2
    assert a.contains((3, 8)) and not a.contains((3, 9))


def test_the_reference_package_layout_resolves_by_module_path():
    """qtttgym/__init__.py:1-4 and its four modules: a caller that names `qtttgym.display.displayBoard` (strat_eval.py:44)
    or imports `qtttgym.qeval` finds the same objects under this package."""
    import importlib
    import qtttgym_amd
    for mod, name in (("board", "Board"), ("qeval", "QEvalClassic"), ("display", "displayBoard"), ("env", "Env")):
        m = importlib.import_module("qtttgym_amd." + mod)
        assert getattr(m, name) is getattr(qtttgym_amd, name)
    assert callable(qtttgym_amd.Env._reward) and callable(qtttgym_amd.Env.observ) and callable(qtttgym_amd.Env.render)


def test_display_board_matches_reference_layout(capsys):
    from qtttgym_amd.board import Board, QEvalClassic, displayBoard
    b = Board(QEvalClassic())
    b.moves = [(0, 1, 0), (0, 1, 1), (4, 8, 2)]
    b.board = [1, 0, -1, -1, -1, -1, -1, -1, -1]
    displayBoard(b)
    out = capsys.readouterr().out
    lines = out.splitlines()
    assert lines[0] == "+---+---+---+" and len(lines) == 14
    assert lines[1] == "| o |x x|   |" and lines[2] == "|o1o| 0 |   |" and lines[3] == "| o |x x|   |"
    assert lines[5] == "|   |  2|   |"


def test_oracle_ind2move_table():
    import oracle
    pairs = [oracle.ind2move(a) for a in range(36)]
    assert pairs == [(i, j) for i in range(9) for j in range(i + 1, 9)]


def test_no_kernel_uses_scratch_memory():
    """Every kernel keeps its working set in registers / LDS: private (scratch) memory is both slow
    and the one thing that ever produced a wrong element here (runtime-indexed per-thread arrays
    under 512-thread workgroups in round 1).  Checked from the compiler's own report."""
    import subprocess
    src = os.path.join(ROOT, "qtttgym_amd", "csrc", "qttt_kernels.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
                          "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage",
                          "-o", os.devnull, src], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    assert len(names) == len(scratch) and len(names) >= 25
    assert all(v == 0 for v in scratch), [n for n, v in zip(names, scratch) if v]


def test_bench_self_launch_propagates_a_failing_rank():
    """bench.py --gpus 2 with no launcher starts its own two ranks without importing torch in the
    parent; here (no GPU in the build container, or both ranks on RCCL with one GPU) every rank refuses
    to run, and the parent must come back with that failure instead of hanging or printing a line."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the two ranks would simply run")
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "QTTT_DIST_BACKEND"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--boards", "64",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=300, env=e)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "no HIP device visible" in out.stderr or "needs 2 GPUs" in out.stderr


def test_action_indexing_is_the_lexicographic_pair_order():
    """mcts.py:339-350 index the 36 unordered pairs of squares in lexicographic order ((0,1), (0,2) … (7,8); the table
    is printed in SURVEY.md Appendix A): ind2move enumerates exactly that order, move2ind is its inverse in either
    argument order, and the batched tensor forms agree (host logic, CPU)."""
    import itertools
    import torch
    from qtttgym_amd import ind2move, move2ind
    from qtttgym_amd.actions import action36_to_pairs, pairs_to_action36, legal_mask_to_bool
    assert [ind2move(n) for n in range(36)] == list(itertools.combinations(range(9), 2))
    for n in range(36):
        i, j = ind2move(n)
        assert move2ind(i, j) == n == move2ind(j, i)
    assert [ind2move(n) for n in (0, 7, 8, 35)] == [(0, 1), (0, 8), (1, 2), (7, 8)]
    a = torch.tensor([0, 7, 8, 35, 36, 255], dtype=torch.uint8)
    p = action36_to_pairs(a)
    assert p.tolist() == [[0, 1], [0, 8], [1, 2], [7, 8], [255, 255], [255, 255]]
    back = pairs_to_action36(torch.tensor([[1, 0], [8, 0], [2, 1], [7, 8], [4, 4], [9, 1]], dtype=torch.uint8))
    assert back.tolist() == [0, 7, 8, 35, 255, 255]
    m = legal_mask_to_bool(torch.tensor([0, 1 | (1 << 35), (1 << 36) - 1], dtype=torch.int64))
    assert m.shape == (3, 36) and m[0].sum() == 0 and m[1].nonzero().flatten().tolist() == [0, 35] and bool(m[2].all())


def test_bench_refuses_to_run_without_a_gpu():
    """The judged bench has no CPU path: on a box without a HIP device it says so and exits non-zero (it must never
    fall back to the oracle or to a torch loop)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible here")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-legs", "--no-cpu-baseline", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "no HIP device visible" in (out.stderr + out.stdout) and "{" not in out.stdout
    # and the host package says the same instead of stepping boards on the CPU
    from qtttgym_amd import VecEnv, _native
    with pytest.raises(_native.QtttNativeError):
        VecEnv(4, device="cpu")


def test_integration_md_names_every_exported_symbol():
    """INTEGRATION.md §2's table is the map from the C ABI to the reference's interfaces: every function include/qttt.h
    declares has its row."""
    header = open(os.path.join(ROOT, "include", "qttt.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = sorted(set(re.findall(r"\b(qttt_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) == 30, declared
    missing = [s for s in declared if "`%s" % s not in doc and s not in doc]
    assert not missing, missing
