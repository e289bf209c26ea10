"""bench.py's JSON line beyond the basic contract (tests/test_bench_contract_gpu.py): the driver's N > 1 command shape
rehearsed with as many gloo ranks as one card may host (per-rank timings, CPU binding), an asymmetric failure in front
of the optional gather, and the tail of the default line carrying BASELINE configs 2 / 3 / 5, the beyond-cache fraction
and the default gym call as plain scalars."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# The GPU boxes of this pool allow at most 6 processes on a card at once, and the pytest process itself is one of them
# once an earlier test has touched the GPU: the driver-shaped N = 8 command is rehearsed with 4 ranks here (4 + pytest
# = 5); its eight-rank bookkeeping runs on CPU gloo ranks in tests/test_dist_bookkeeping_cpu.py.
RANKS_ON_ONE_CARD = 4


def bench(*args, env=None, check=True):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=900, env=e)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if check:
        assert out.returncode == 0, out.stderr[-2000:]
        assert len(lines) == 1, out.stdout
    return out, lines


@pytest.mark.timeout(900)
def test_driver_shape_with_four_gloo_ranks_on_the_one_card():
    """`python bench.py --gpus N --steps 20 --warmup 5` as the driver runs it (self-launched ranks, no torchrun), N = 4
    ranks sharing the one GPU over gloo: every rank is in the line, in rank order."""
    n = RANKS_ON_ONE_CARD
    out, lines = bench("--gpus", str(n), "--steps", "20", "--warmup", "5", "--boards", "131072",
                       env={"QTTT_DIST_BACKEND": "gloo"})
    d = json.loads(lines[0])
    assert d["n_gpus"] == n == d["ranks_seen"] and d["scaling"] == "weak" and d["config"]["self_launched"] is True
    assert d["config"]["boards_total"] == n * 131072 and d["config"]["board_offset_last_rank"] == (n - 1) * 131072
    assert d["config"]["replay_matches_recording"] is True
    pr = d["per_rank_ms_per_step"]
    assert isinstance(pr, list) and len(pr) == n and all(x > 0 for x in pr)
    assert len(d["per_rank_best_region_ms_per_step"]) == n and len(d["per_rank_host_wall_ms_per_step"]) == n
    # value is made from the slowest rank's median region
    assert d["slowest_rank"] == max(range(n), key=lambda r: pr[r])
    assert abs(max(pr) - d["ms_per_step"]) < 1e-12 and abs(d["rank_spread"] - max(pr) / min(pr)) < 1e-9
    assert abs(d["value"] - n * 131072 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    assert all(b <= m + 1e-12 for b, m in zip(d["per_rank_best_region_ms_per_step"], pr))
    a = d["config"]["cpu_affinity"]
    assert isinstance(a, dict) and a["bound"] in (True, False) and (a["bound"] or a["reason"])
    assert d["returns_gather"]["boards_gathered"] == n * 131072


def test_single_rank_line_has_the_per_rank_fields_too():
    out, lines = bench("--boards", "65536", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-legs")
    d = json.loads(lines[0])
    assert d["per_rank_ms_per_step"] == [d["ms_per_step"]] and d["slowest_rank"] == 0 and d["rank_spread"] == 1.0
    assert "cpu_affinity" in d["config"]
    out, lines = bench("--boards", "65536", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-legs",
                       env={"QTTT_BENCH_NO_BIND": "1"})
    assert json.loads(lines[0])["config"]["cpu_affinity"] == {"bound": False, "reason": "QTTT_BENCH_NO_BIND=1"}


@pytest.mark.timeout(600)
def test_one_rank_failing_to_prepare_the_gather_does_not_hang_the_others():
    """an ASYMMETRIC failure (rank 1 only) in front of the optional returns gather — every rank skips the
    gather together (dist.agree), the line is printed with the value, the error is in it."""
    out, lines = bench("--gpus", "2", "--boards", "16384", "--steps", "10", "--warmup", "2",
                       env={"QTTT_DIST_BACKEND": "gloo", "QTTT_BENCH_FAIL_GATHER": "rank1", "QTTT_BENCH_TIMEOUT": "300"})
    d = json.loads(lines[0])
    assert d["ranks_seen"] == 2 and d["value"] > 0 and len(d["per_rank_ms_per_step"]) == 2
    assert "error" in d["returns_gather"] and "another rank" in d["returns_gather"]["error"]
    assert "injected failure" in out.stderr


@pytest.mark.timeout(900)
def test_the_tail_of_the_default_line_carries_the_baseline_configs():
    """a driver that stores only the tail of the (14 KB) line, or only scalar fields of `roofline`, still
    sees BASELINE configs 2 / 3 / 5 and the beyond-cache fraction."""
    out, lines = bench("--cpu-budget", "2")
    line = lines[0]
    tail = line[-800:]
    for k in ("config2_us", "config3_us", "config3_frac", "config5_us", "beyond_cache_frac", "gym_default_us", "gym_default_eager_us"):
        assert '"%s"' % k in tail, (k, tail)
    d = json.loads(line)
    assert list(d)[-1] == "configs"                                   # after legs and cpu_baseline
    r, c = d["roofline"], d["configs"]
    legs = {l["name"]: l for l in d["legs"]}
    assert r["beyond_cache_frac"] == r["beyond_cache"]["frac"] == c["beyond_cache_frac"] == legs["beyond_infinity_cache_16777216_boards"]["frac"]
    assert r["beyond_cache_boards"] == 16777216 and r["beyond_cache_launch_us"] == legs["beyond_infinity_cache_16777216_boards"]["us_per_step"]
    assert all(isinstance(r[k], (int, float)) for k in ("beyond_cache_boards", "beyond_cache_launch_us", "beyond_cache_frac",
                                                         "beyond_cache_frac_of_achievable"))
    assert c["config2_us"] == legs["config2_4096_boards"]["us_per_step"] and c["config3_us"] == legs["config3_262144_boards"]["us_per_step"]
    assert c["config5_us"] == legs["config5_expand_rollout_65536_pairs"]["us_per_unit"]
    # the default gym call: device-paced and by the region clock, beside the zero-copy form
    g = legs["gym_default_1048576_boards"]
    assert c["gym_default_us"] == g["device_paced_us_per_step"] and c["gym_default_eager_us"] == g["us_per_step"]
    assert c["gym_us"] == legs["gym_1048576_boards"]["us_per_step"]
    assert g["replay_matches_recording"] is True and g["mode"] == "gym-default"
    # (qtttgym_amd/_fastviews.so is optional: the line says which way the eight tensors were carved)
    assert ("qtttgym_amd/_fastviews.so" in g["outputs"]) is os.path.exists(os.path.join(ROOT, "qtttgym_amd", "_fastviews.so"))
    assert 0 < g["us_per_step_with_output_pool_4"] < 14.5
    assert g["device_paced_us_per_step"] < 13.0 + 1.5, g       # asked: <= 13 us on a typical box (the pool's boxes differ by 7 %)
    assert abs(c["config1_us"] - d["ms_per_step"] * 1e3) < 1e-9 and abs(c["config1_frac"] - r["frac"]) < 1e-12
