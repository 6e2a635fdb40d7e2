"""bench.py as the driver starts it: `python bench.py --gpus N` with N > 1 and no
torch.distributed environment must start the N ranks itself -- as a CHILD process, without
this process ever loading torch or the HIP library -- relay the child's output and return
its exit code.  (The ranks themselves need GPUs: tests/test_gpu_bench_cli.py.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_builds_the_torchrun_command(monkeypatch):
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("MASTER_PORT", raising=False)
    assert bench.launch_ranks(8) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launcher_process_never_loads_torch_or_the_library():
    """`python bench.py --gpus 2` in a process of its own, the rank launch replaced by a stub
    that records which modules the launcher had loaded: no torch, no megakv_amd."""
    code = (
        "import sys, json; sys.argv = ['bench.py', '--gpus', '2', '--steps', '3']\n"
        "import bench\n"
        "def stub(cmd, env=None):\n"
        "    print(json.dumps({'torch': 'torch' in sys.modules, 'mkv': 'megakv_amd' in sys.modules,\n"
        "                      'n': [c for c in cmd if c.startswith('--nproc')]}))\n"
        "    return 3\n"
        "bench.subprocess.call = stub\n"
        "bench.main()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 3, p.stderr[-2000:]
    import json
    seen = json.loads(p.stdout.strip().splitlines()[-1])
    assert seen == {"torch": False, "mkv": False, "n": ["--nproc-per-node=2"]}


def test_counter_passes_are_read_per_dispatch_of_the_timed_steps(tmp_path):
    """roofline.traffic of a run comes from rocprofv3's counter_collection.csv of a child pass (bench.live_traffic):
    the rows of a dispatch are summed (a counter is kept per instance), the warm-up dispatches dropped, the timed
    steps averaged, the searches the child checks itself with behind them left out, other kernels ignored."""
    import bench
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,"
            "Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,SGPR_Count,Counter_Name,Counter_Value"]

    def add(d, kern, counter, values):
        for v in values:
            rows.append(f"{d},{d},1,1,1,1,1024,7,\"{kern}\",256,0,0,52,68,{counter},{v}")

    d = 1
    for _ in range(3):                                      # pre-load: another kernel
        add(d, "void mkv::k_mark<(mkv::Op)1>(mkv::Ctl*)", "FETCH_SIZE", [5.0, 5.0])
        d += 1
    for step in range(2 + 4 + 3):                           # 2 warm-up, 4 timed, 3 checks behind them
        val = 1000.0 if step < 2 else (100.0 + step if step < 6 else 9000.0)
        add(d, "void mkv::k_search_mark<(mkv::Op)1>(uint2 const*, unsigned int*)", "FETCH_SIZE", [val / 2, val / 2])
        add(d, "void mkv::k_search_mark<(mkv::Op)1>(uint2 const*, unsigned int*)", "WRITE_SIZE", [1.0])
        d += 1
        add(d, "void mkv::k_round<(mkv::Op)1, 1>(mkv::Ctl*)", "FETCH_SIZE", [77.0])
        d += 1
    p = tmp_path / "1_counter_collection.csv"
    p.write_text("\n".join(rows) + "\n")
    mean, n, seen = bench.counter_per_launch(str(p), "k_search_mark", "FETCH_SIZE", 2, 4)
    assert (n, seen) == (4, 9) and mean == (102.0 + 103.0 + 104.0 + 105.0) / 4
    mean, n, seen = bench.counter_per_launch(str(p), "k_search_mark", "WRITE_SIZE", 2, 4)
    assert (mean, n) == (1.0, 4)
    assert bench.counter_per_launch(str(p), "k_search<", "FETCH_SIZE", 2, 4) == (None, 0, 0)
    assert bench.counter_per_launch(str(p), "k_search_mark", "FETCH_SIZE", 8, 4)[0] is None     # (too few behind the warm-up)


def test_a_run_under_a_profiler_is_recognised(monkeypatch):
    import bench
    for k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH", "LD_PRELOAD"):
        monkeypatch.delenv(k, raising=False)
    assert not bench.profiler_present()
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.profiler_present()


def test_counter_csv_of_an_unknown_shape_is_an_error_the_caller_survives(tmp_path):
    """(bench.py wraps live_traffic in try / except: a profiler that writes another CSV costs the run its live
    traffic figure, not its line)"""
    import bench
    import pytest
    p = tmp_path / "x_counter_collection.csv"
    p.write_text("Dispatch_Id,Kernel_Name,FETCH_SIZE\n1,k_search_mark,5\n")
    with pytest.raises(KeyError):
        bench.counter_per_launch(str(p), "k_search_mark", "FETCH_SIZE", 0, 1)
    src = open(bench.__file__).read()
    assert "live, live_why = live_traffic(args, fused)" in src and "except Exception as ex:" in src.split("live, live_why = live_traffic(args, fused)")[1][:200]
