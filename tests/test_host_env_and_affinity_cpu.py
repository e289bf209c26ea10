"""Host-side behaviour with no GPU: importing the package leaves os.environ alone (recommended_env() is the documented
helper), and the NUMA binding helper (qtttgym_amd/affinity.py) on a made-up sysfs tree, incl. the *_VISIBLE_DEVICES
remapping levels, and really binding in a child process."""
import os
import subprocess
import sys

import pytest

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
    # HIP_ and CUDA_VISIBLE_DEVICES are ONE level (the CUDA_ name is read only when the HIP_ name is unset);
    # a launcher that exports both with the same permuted list must not see the permutation applied twice
    both = {"HIP_VISIBLE_DEVICES": "2,3,0,1", "CUDA_VISIBLE_DEVICES": "2,3,0,1"}
    assert [affinity.visible_index(k, both) for k in range(4)] == [2, 3, 0, 1] and affinity.visible_count(8, both) == 4
    assert affinity.visible_index(0, {"CUDA_VISIBLE_DEVICES": "5,4"}) == 5                       # alone, the alias counts
    assert affinity.visible_index(0, {"HIP_VISIBLE_DEVICES": "1", "CUDA_VISIBLE_DEVICES": "7"}) == 1   # HIP_ wins
    upper = {"HIP_VISIBLE_DEVICES": "4,5,6,7", "CUDA_VISIBLE_DEVICES": "4,5,6,7"}
    assert affinity.visible_count(8, upper) == 4 and affinity.visible_index(3, upper) == 7
    assert affinity.visible_index(1, dict(both, ROCR_VISIBLE_DEVICES="7,6,5,4")) == 4             # HIP 1 -> ROCr 3 -> KFD 4
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
