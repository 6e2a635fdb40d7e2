import sys, os, json, subprocess
for extra in (["--overlap","none"], ["--overlap","fused"]):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "3", "--no-cpu", "--workload", "zipf50"] + extra, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().split("\n")[-1])
        print(extra, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])
    except Exception as e:
        print(extra, "ERR", out.stdout[-500:], out.stderr[-1500:])
