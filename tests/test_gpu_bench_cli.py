"""bench.py from its command line, as the driver runs it (small geometry so each run takes
seconds): the single-GPU line, `--gpus 2` started WITHOUT torch.distributed.run (bench.py
launches its own ranks; here they share cuda:0 and exchange through host memory, the only
multi-rank shape a one-GPU box has), and the sharded engine over RCCL with one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"bench.py {' '.join(flags)} -> {p.returncode}\n{p.stderr[-4000:]}"
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"exactly one line on stdout expected, got {len(lines)}: {p.stdout[-2000:]}"
    return json.loads(lines[0])


SMALL = ["--mem-p", "24", "--batch-log2", "16", "--no-cpu", "--no-pcie", "--kernel-launches", "12",
         "--prewarm-ms", "20"]


def check_common(d, n_gpus, steps, warmup):
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "Mops" and d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["get_loc_correct_last_timed"] == 1.0, d
    assert d["set_loc_correct_last_timed"] == 1.0, d          # the timed region's inserts are in the table
    assert d["table_entries_ok"] is True, d["table_entries"]   # ... and nothing else is
    r = d["roofline"]
    assert r and r["bound"] == "hbm" and r["launches_timed"] >= 8 and 0 < r["frac"] < 1.5
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["launch_us"] * 1e-6) / 1e9, rel=2e-3)


def test_single_gpu_line():
    d = run_bench("--steps", "6", "--warmup", "3", *SMALL)
    check_common(d, 1, 6, 3)
    assert "k_search_mark" in d["roofline"]["kernel"]
    assert d["step_algorithmic"]["bytes_per_step"] > 0
    # roofline.traffic is this run's own (two child passes under rocprofv3 --pmc) -- or the line says why it is not
    r = d["roofline"]
    if r["traffic_not_live_because"] is None:
        assert r["traffic_source"].startswith("live: rocprofv3") and r["traffic"] > 0 and r["traffic_head"] == r["library"], r
    else:
        assert isinstance(r["traffic_not_live_because"], str) and r["traffic_not_live_because"], r


def test_single_gpu_line_with_committed_traffic():
    """--no-live-traffic: the counter passes committed under profiles/ (none for this small geometry's workload key
    is an error neither: the field is null or labelled with its source)."""
    d = run_bench("--steps", "4", "--warmup", "2", "--no-live-traffic", *SMALL)
    r = d["roofline"]
    assert r["traffic_not_live_because"] is None and (r["traffic"] is None or "profiles/" in r["traffic_source"]), r


def test_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment."""
    d = run_bench("--gpus", "2", "--comm", "staged", "--mem-p", "22", "--batch-log2", "16", "--steps", "4",
                  "--warmup", "2", "--no-cpu", "--no-pcie", "--kernel-launches", "8", "--prewarm-ms", "20")
    check_common(d, 2, 4, 2)
    assert d["config"]["mem_p_aggregate"] == 23 and "2 shards" in d["config"]["sharding"]


def test_gpus_8_starts_its_own_ranks():
    """`python bench.py --gpus 8` (BASELINE configs[4]'s shape: 8 shards, aggregate MEM_P = local + 3):
    eight ranks sharing cuda:0, the exchange staged through host memory; the N > 1 line carries the
    roofline of the local slot search and the checks of what the timed region did."""
    d = run_bench("--gpus", "8", "--comm", "staged", "--mem-p", "21", "--batch-log2", "15", "--steps", "4",
                  "--warmup", "2", "--no-cpu", "--no-pcie", "--kernel-launches", "8", "--prewarm-ms", "10", timeout=1500)
    check_common(d, 8, 4, 2)
    assert d["config"]["mem_p_aggregate"] == 24 and "8 shards" in d["config"]["sharding"]
    assert "k_search_slots_mark" in d["roofline"]["kernel"]


def test_one_rank_through_the_rccl_engine():
    d = run_bench("--force-sharded", "--steps", "6", "--warmup", "2", *SMALL)
    check_common(d, 1, 6, 2)
    assert "library-owned RCCL communicator" in d["config"]["launch"]


def test_one_rank_through_the_rccl_engine_reports_its_phases():
    """... and with enough steps the line carries the engine's own device time per phase of a batch
    (mkv_shard_trace / mkv_shard_phases: pack, exchange, local, un-pack)."""
    d = run_bench("--force-sharded", "--steps", "30", "--warmup", "2", *SMALL)
    check_common(d, 1, 30, 2)
    ph = d["phases_us_per_rank"]
    assert isinstance(ph, list) and len(ph) == 1, ph
    assert ph[0]["local"] > 0 and ph[0]["pack"] > 0 and ph[0]["unpack"] > 0, ph


def test_transport_check_fails_loudly_before_the_table_is_built():
    """The sharded run checks its transport in a child process first (RCCL ring, the engine's connect and first
    batches).  With the engine pointed at an RCCL that does not exist the run ends there: non-zero exit, the
    failing phase named, no line on stdout -- not a fall-back line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MKV_RCCL_PATH"] = "/nonexistent/librccl.so"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-sharded", "--steps", "4", "--warmup", "2",
                        *SMALL], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 23, (p.returncode, p.stderr[-2000:])
    assert "transport check" in p.stderr and "C shard engine: create / connect" in p.stderr, p.stderr[-2000:]
    assert not p.stdout.strip()
