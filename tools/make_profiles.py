#!/usr/bin/env python3
"""Turns gpurun_out/profiles_raw/<tag>/ (tools/collect_profiles.sh) into the
committed evidence under profiles/:
  profiles/<round>_<tag>_kernel_stats.csv   rocprofv3 --stats summary (our kernels + top others)
  profiles/<round>_<tag>_pmc.json           per-kernel PMC averages per launch
  profiles/search_traffic.json              HBM bytes per search launch, read by bench.py
usage: make_profiles.py <round> <tag> <workload-key>      e.g.  r01 zipf95_mix zipf95-mix
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(raw, name, suffix):
    f = glob.glob(os.path.join(raw, name, "*", f"*{suffix}"))
    return max(f, key=os.path.getmtime) if f else None


def main():
    rnd, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]
    raw = os.path.join(ROOT, "gpurun_out", "profiles_raw", tag)
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    # --- kernel stats summary
    f = find(raw, "stats", "kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, f"{rnd}_{tag}_kernel_stats.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:14]:
            name = r["Name"].split("(")[0].replace("void ", "")
            w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"], r["StdDev"]])
    # steady-state average of the timed steps only (the stats file also holds the pre-load launches)
    tr = find(raw, "stats", "kernel_trace.csv")
    trows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(trows) if "k_search" in r["Kernel_Name"]]
    steady = collections.defaultdict(list)
    for r in trows[marks[-100] if len(marks) >= 100 else marks[0]:]:
        steady[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    # --- PMC per launch
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in ("pmc_fetch", "pmc_write", "pmc_l2"):
        f = find(raw, name, "counter_collection.csv")
        if not f:
            continue
        crow = list(csv.DictReader(open(f)))
        # keep the steady-state launches: drop the first third of each kernel's dispatches
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in crow:
            if "mkv::" in r["Kernel_Name"]:
                per[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(
                    float(r["Counter_Value"]))
        for k, cs in per.items():
            for c, v in cs.items():
                v = v[len(v) // 3:]
                pmc[k][c] = sum(v) / max(len(v), 1)
    summary = {"round": rnd, "tag": tag, "steady_state_avg_us": {k: round(sum(v) / len(v), 2) for k, v in steady.items()
                                                                 if "mkv::" in k or "fill" in k.lower()},
               "pmc_per_launch": {k: dict(v) for k, v in pmc.items()}}
    json.dump(summary, open(os.path.join(out, f"{rnd}_{tag}_pmc.json"), "w"), indent=1)
    # --- traffic of the dominant (search) kernel, for bench.py
    dom = [k for k in pmc if "k_search" in k]
    if dom:
        k = max(dom, key=lambda x: pmc[x].get("FETCH_SIZE", 0))
        fetch_kb, write_kb = pmc[k].get("FETCH_SIZE", 0.0), pmc[k].get("WRITE_SIZE", 0.0)
        tp = os.path.join(out, "search_traffic.json")
        d = json.load(open(tp)) if os.path.exists(tp) else {}
        d[key] = {"kernel": k, "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
                  "hbm_bytes_per_launch": int((fetch_kb + write_kb) * 1024),
                  "note": "FETCH_SIZE + WRITE_SIZE (KB x 1024), separate PMC passes; calibrated on uniform GETs: "
                          "random 64-B bucket reads are tallied exactly, the coalesced request stream at one "
                          "half (gfx950 FETCH_SIZE rule) -- see DESIGN.md 'measurement'",
                  "l2": {c: pmc[k][c] for c in pmc[k] if c.startswith("TCC_")}, "source": f"profiles/{rnd}_{tag}_pmc.json"}
        json.dump(d, open(tp, "w"), indent=1)
    print(json.dumps(summary, indent=1)[:1500])


if __name__ == "__main__":
    main()
