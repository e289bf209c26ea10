"""Round 5, CPU side: the package leaves the environment alone, the NUMA binding helper on a made-up sysfs tree, the
per-rank bookkeeping collectives with EIGHT gloo ranks, the evaluator plug point's rule."""
import os
import socket
import subprocess
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_importing_the_package_leaves_the_environment_untouched():
    """VERDICT r4 #7 / ADVICE: HIP_FORCE_DEV_KERNARG is the host's choice; recommended_env() is the documented helper."""
    code = ("import os, sys; sys.path.insert(0, %r); before = dict(os.environ); import qtttgym_amd; "
            "from qtttgym_amd import _native; assert dict(os.environ) == before, 'import changed os.environ'; "
            "r = qtttgym_amd.recommended_env(); assert r['HIP_FORCE_DEV_KERNARG'] == '1' and dict(os.environ) == before; "
            "qtttgym_amd.recommended_env(apply=True); assert os.environ['HIP_FORCE_DEV_KERNARG'] == os.environ.get('WANT', '1'); "
            "print('ok')" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("HIP_FORCE_DEV_KERNARG",)}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-1500:]
    # a value the host set itself is kept (setdefault)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(env, HIP_FORCE_DEV_KERNARG="0", WANT="0"))
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-1500:]


def _fake_sysfs(tmp_path, gpus):
    """gpus: [(domain, bus, dev, fn, numa_node, cpulist)]; KFD node 0 is a CPU node, as on real hosts"""
    kfd, pci = tmp_path / "kfd", tmp_path / "pci"
    (kfd / "0").mkdir(parents=True)
    (kfd / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (dom, bus, dev, fn, node, cpus) in enumerate(gpus, 1):
        (kfd / str(i)).mkdir()
        (kfd / str(i) / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain %d\n"
                                                 % ((bus << 8) | (dev << 3) | fn, dom))
        d = pci / ("%04x:%02x:%02x.%x" % (dom, bus, dev, fn))
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
        (d / "local_cpulist").write_text(cpus + "\n")
    return str(kfd), str(pci)


def test_affinity_helper_on_a_made_up_topology(tmp_path, monkeypatch):
    from qtttgym_amd import affinity
    assert affinity.parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    assert affinity.format_cpulist({0, 1, 2, 3, 8, 10, 11}) == "0-3,8,10-11"
    allowed = sorted(os.sched_getaffinity(0))
    half = allowed[:max(1, len(allowed) // 2)]
    kfd, pci = _fake_sysfs(tmp_path, [(0, 5, 0, 0, 0, affinity.format_cpulist(half)),
                                      (0, 0x85, 0, 0, 1, "100000-100003")])
    assert affinity.kfd_gpus(kfd) == ["0000:05:00.0", "0000:85:00.0"]
    for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    a = affinity.bind_to_gpu(0, kfd, pci, apply=False)                  # reports what it would do
    assert a["numa_node"] == 0 and a["bound"] is False
    if len(half) < len(allowed):
        assert a["cpus"] == len(half) and a["cpulist"] == affinity.format_cpulist(half) and a["reason"] == "apply=False"
    b = affinity.bind_to_gpu(1, kfd, pci)                               # local cores not allowed here: nothing bound
    assert b["bound"] is False and b["numa_node"] == 1 and "outside" in b["reason"]
    c = affinity.bind_to_gpu(7, kfd, pci)                               # no such device: nothing bound, no exception
    assert c["bound"] is False and "not found" in c["reason"]
    d = affinity.bind_to_gpu(0, str(tmp_path / "nowhere"), pci)         # unreadable: nothing bound, no exception
    assert d["bound"] is False and d["reason"]
    # a container that hides GPUs by other means than the *_VISIBLE_DEVICES lists: the topology shows two, HIP sees one —
    # which one is not known, nothing is bound
    e = affinity.bind_to_gpu(0, kfd, pci, apply=False, expected_devices=1)
    assert e["bound"] is False and e["numa_node"] is None and "not known" in e["reason"]
    assert affinity.bind_to_gpu(0, kfd, pci, apply=False, expected_devices=2)["numa_node"] == 0
    assert affinity.visible_count(8, {}) == 8 and affinity.visible_count(8, {"HIP_VISIBLE_DEVICES": "3"}) == 1
    assert affinity.visible_count(8, {"ROCR_VISIBLE_DEVICES": "0,1,2,3", "HIP_VISIBLE_DEVICES": "1,0"}) == 2
    assert affinity.visible_count(8, {"HIP_VISIBLE_DEVICES": "GPU-1234"}) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")                    # HIP device 0 is KFD GPU 1
    assert affinity.visible_index(0) == 1 and affinity.bind_to_gpu(0, kfd, pci, apply=False)["numa_node"] == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-abcdef")             # a UUID list is not resolved
    assert affinity.visible_index(0) is None
    assert os.sched_getaffinity(0) == set(allowed)                      # nothing above changed this process


def test_affinity_really_binds_in_a_child_process(tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("one core")
    from qtttgym_amd import affinity
    half = allowed[:len(allowed) // 2]
    kfd, pci = _fake_sysfs(tmp_path, [(0, 5, 0, 0, 0, affinity.format_cpulist(half))])
    code = ("import os, sys, json; sys.path.insert(0, %r); from qtttgym_amd.affinity import bind_to_gpu; "
            "a = bind_to_gpu(0, %r, %r); print(json.dumps([a, sorted(os.sched_getaffinity(0))]))" % (ROOT, kfd, pci))
    env = {k: v for k, v in os.environ.items() if not k.endswith("VISIBLE_DEVICES")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
    import json
    a, now = json.loads(out.stdout)
    assert a["bound"] is True and now == half and a["cpus"] == len(half)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from qtttgym_amd.dist import init_from_env, gather_rank_values, agree, shard_range
    init_from_env(backend="gloo")
    rows = gather_rank_values([1.0 + rank, 0.5 + rank, 10.0 * rank])
    ok_all = agree(True)
    ok_one_fails = agree(rank != 5)                      # rank 5 "failed to prepare": every rank must learn it
    q.put((rank, rows, ok_all, ok_one_fails, shard_range(2097152, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_per_rank_bookkeeping_with_eight_gloo_ranks():
    """The N = 8 shape of bench.py's bookkeeping off the timed path: per-rank timings in rank order on
    every rank, the all-or-none agreement in front of an optional collective, BASELINE config 4's shard offsets."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [[1.0 + r, 0.5 + r, 10.0 * r] for r in range(world)]
    for rank, rows, ok_all, ok_one_fails, (lo, hi) in res:
        assert rows == want and ok_all is True and ok_one_fails is False
        assert (lo, hi) == (rank * 262144, (rank + 1) * 262144)
    from qtttgym_amd.dist import gather_rank_values, agree
    assert gather_rank_values([3, 4]) == [[3.0, 4.0]] and agree(True) is True and agree(False) is False   # no process group


def test_fastboard_agrees_with_the_python_bookkeeping():
    """qtttgym_amd/_fastboard.so (csrc/fastboard.c) against board.py's own pack / _adopt, no GPU: a ctypes callback
    stands in for qttt_board_op_host, so what is compared is exactly the host bookkeeping — the 41 packed bytes, the
    attributes afterwards, and WHICH set / list objects survive (the reference's aliasing, board.py:19,25,53-69)."""
    import ctypes
    import random
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_fastboard()
    from qtttgym_amd import _fastboard
    from qtttgym_amd.board import Board, QEvalClassic, _Staging
    buf_in = (ctypes.c_uint8 * 64)()
    buf_out = (ctypes.c_uint8 * 64)()
    seen, reply = [], [bytes(64)]

    @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)
    def fake_op_host(p_in, p_out, n, stream):
        seen.append(ctypes.string_at(p_in, 64))
        ctypes.memmove(p_out, reply[0], 64)
        return 0
    _fastboard.init(ctypes.cast(fake_op_host, ctypes.c_void_p).value, ctypes.addressof(buf_in), ctypes.addressof(buf_out))
    rng = random.Random(11)

    def random_sets(k):
        out = []
        for _ in range(k):
            out.append(set(rng.sample(range(9), rng.randint(1, 5))))
        return out

    def random_out_record(old_q):
        n = rng.randint(0, 9)
        r = bytearray(64)
        for i in range(9):
            r[2 * i], r[2 * i + 1] = (rng.randrange(9), rng.randrange(9)) if i < n else (255, 255)
        r[18] = n
        for v in range(9):
            r[19 + v] = rng.choice([255, 255] + list(range(9)))
        # new components: some equal to old ones, some grown from old ones, some new
        new = []
        for s in old_q:
            c = rng.random()
            if c < 0.4:
                new.append(set(s))
            elif c < 0.7:
                new.append(set(s) | set(rng.sample(range(9), 2)))
        if rng.random() < 0.5:
            new.append(set(rng.sample(range(9), 2)))
        new = new[:4]
        rng.shuffle(new)
        r[28] = len(new)
        for k, s in enumerate(new):
            m = sum(1 << x for x in s)
            r[30 + 2 * k], r[31 + 2 * k] = m & 255, m >> 8
        r[49], r[50] = rng.choice([255, 4, 6, 8]), rng.choice([255, 5, 7])
        return bytes(r)

    def make_board():
        b = Board(QEvalClassic())
        n = rng.randint(0, 9)
        b.moves = [(rng.randrange(9), rng.randrange(9), i) for i in range(n)]
        b.board = [rng.choice([-1, -1] + list(range(9))) for _ in range(9)]
        b.qstructs = random_sets(rng.randint(0, 4))
        return b

    def clone(b):
        c = Board(b.qeval)
        c.moves, c.board, c.qstructs = list(b.moves), list(b.board), [set(s) for s in b.qstructs]
        return c

    for case in range(400):
        a = make_board()
        b = clone(a)
        op, lo, hi, bit, drop = rng.randrange(3), rng.randrange(9), rng.randrange(9), rng.randrange(2), bool(a.moves) and rng.random() < 0.3
        reply[0] = random_out_record(a.qstructs)
        want_in = _Staging.pack(a, op, lo, hi, bit, drop)
        olds_a = (a.moves, a.board, a.qstructs, list(a.qstructs))
        olds_b = (b.moves, b.board, b.qstructs, list(b.qstructs))
        a._adopt(reply[0])                                            # Python
        assert _fastboard.board_op(b, op, lo, hi, bit, drop, 0) == 0  # C
        assert seen[-1][:41] == want_in, case
        assert (a.moves, a.board, a.qstructs, a._win) == (b.moves, b.board, b.qstructs, b._win), case
        assert all(type(m) is tuple and all(type(x) is int for x in m) for m in b.moves)
        # aliasing: the three attribute objects are the ones the caller held; per component, an old set object survives
        # in b exactly where the corresponding one survives in a
        assert b.moves is olds_b[0] and b.board is olds_b[1] and b.qstructs is olds_b[2]
        ida = [next((j for j, t in enumerate(olds_a[3]) if t is s), None) for s in a.qstructs]
        idb = [next((j for j, t in enumerate(olds_b[3]) if t is s), None) for s in b.qstructs]
        assert ida == idb, (case, ida, idb)
    # attributes of unusual types are declined (-100), nothing is touched: the Python path then deals with them
    odd = make_board()
    odd.board = tuple(odd.board)
    before = len(seen)
    assert _fastboard.board_op(odd, 0, 0, 1, 0, False, 0) == -100 and len(seen) == before
    odd = make_board()
    odd.qstructs = [frozenset({0, 1}), [2, 3]]
    assert _fastboard.board_op(odd, 0, 0, 1, 0, False, 0) == -100
    odd = make_board()
    odd.moves = [(0.0, 1, 0)]
    assert _fastboard.board_op(odd, 0, 0, 1, 0, False, 0) == -100
    # an error code of the library comes back as it is, attributes untouched

    @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)
    def failing(p_in, p_out, n, stream):
        return 719
    _fastboard.init(ctypes.cast(failing, ctypes.c_void_p).value, ctypes.addressof(buf_in), ctypes.addressof(buf_out))
    keep = make_board()
    snap = (list(keep.moves), list(keep.board), [set(s) for s in keep.qstructs])
    assert _fastboard.board_op(keep, 0, 0, 1, 0, False, 0) == 719
    assert (keep.moves, keep.board, keep.qstructs) == snap
