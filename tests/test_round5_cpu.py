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


