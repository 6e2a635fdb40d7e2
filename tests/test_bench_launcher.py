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
