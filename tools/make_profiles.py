#!/usr/bin/env python3
"""Turns gpurun_out/profiles_raw/<tag>/ (tools/collect_profiles.sh) into the committed
evidence under profiles/:
  profiles/<round>_<tag>_kernel_stats.csv   rocprofv3 --stats summary of the bench command as run
  profiles/<round>_<tag>_timed_region.json  per-kernel durations of the TIMED REGION's dispatches only
                                            (and of the launch-timing pass behind it), plus the PMC
                                            averages per launch over the timed region's dispatches
  profiles/search_traffic.json              HBM bytes per launch of the dominant kernel, read by bench.py
usage: make_profiles.py <round> <tag> <workload-key> <warmup> <steps> <head> [repeats=3]
       e.g.  r04 default zipf95-mix 20 100 $(git rev-parse --short HEAD)

A bench.py run launches the step's kernels in this order: W warm-up steps, R repeats of the K timed
steps (each from the same restored table image), then (after the checks) the launch-timing pass.
One step = one dispatch of the dominant kernel (k_search_mark), so its dispatches number
W + r K .. W + (r+1) K - 1 are repeat r's timed region."""
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


def short(n):
    s = n.split("(")[0].replace("void ", "")
    if "k_round<" in n:          # rounds 1 and 2 are instances of one template (Op, round[, big-pass build]): keep them apart
        s += "round " + n.split("k_round<")[1].split(">")[0].split(",")[1].strip() + ">"
    return s


def library_id(raw):
    """the library build the passes ran with: the `library` word of the line the stats run printed
    (bench.py: roofline.library = the last word of mkv_version(), a digest of the library's sources)"""
    try:
        line = [ln for ln in open(os.path.join(raw, "stats.json")).read().splitlines() if ln.startswith("{")][-1]
        return (json.loads(line).get("roofline") or {}).get("library")
    except Exception:
        return None


def steps_of(rows, key="k_search_mark"):
    """indices of the rows that start a step (the dominant kernel's dispatches)"""
    return [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]


def main():
    rnd, tag, key, W, K, head = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    R = int(sys.argv[7]) if len(sys.argv) > 7 else 3
    raw = os.path.join(ROOT, "gpurun_out", "profiles_raw", tag)
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    dom = "k_search_slots_mark" if "sharded" in tag else ("k_search_mark" if "get100" not in key else "k_search<")
    # --- kernel stats summary of the command as run
    f = find(raw, "stats", "kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, f"{rnd}_{tag}_kernel_stats.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:14]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"], r["StdDev"]])
    # --- the timed region's dispatches
    tr = find(raw, "stats", "kernel_trace.csv")
    trows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
    marks = steps_of(trows, dom)
    marks_set = set(marks)
    summary = {"round": rnd, "tag": tag, "head": head, "library": library_id(raw), "warmup": W, "steps": K,
               "command": open(os.path.join(raw, "stats.err")).read()[:0] or "python3 bench.py " + " ".join(sys.argv[8:]), "repeats": R}

    cycle = ("k_search_mark", "k_search<", "k_round", "k_tail", "k_mark", "k_search_slots")

    def region(lo, hi):
        """the kernels of steps lo .. hi-1: each step = its dominant kernel's dispatch and the
        cycle's kernels that follow it directly"""
        agg = collections.defaultdict(list)
        if hi > len(marks) or hi <= lo:
            return {}
        last_end = 0
        for s_ in range(lo, hi):
            i = marks[s_]
            while i < len(trows) and (i == marks[s_] or (i not in marks_set and any(c in trows[i]["Kernel_Name"] for c in cycle))):
                r = trows[i]
                agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
                last_end = int(r["End_Timestamp"])
                i += 1
        span = (last_end - int(trows[marks[lo]]["Start_Timestamp"])) / 1000.0 / (hi - lo)
        d = {k: {"launches": len(v), "avg_us": round(sum(v) / len(v), 2), "min_us": round(min(v), 2),
                 "max_us": round(max(v), 2)} for k, v in agg.items()}
        d["_step_span_us"] = round(span, 2)
        return d

    reps = [region(W + r * K, W + (r + 1) * K) for r in range(R)]
    spans = [x.get("_step_span_us") for x in reps]
    summary["step_span_us_per_repeat"] = spans
    order = sorted(range(R), key=lambda r: spans[r] if spans[r] is not None else 1e30)
    summary["timed_region"] = reps[order[R // 2]]              # the median repeat (bench.py's `value`)
    summary["timed_region_all_repeats"] = region(W, W + R * K)  # (its span includes the table restores in between)
    summary["timed_region_all_repeats"].pop("_step_span_us", None)
    if len(marks) > W + R * K + 8:
        summary["launch_timing_pass"] = region(W + R * K + 4, len(marks))
    # --- PMC per launch, timed region's dispatches only
    pmc = collections.defaultdict(dict)
    for name in ("pmc_fetch", "pmc_write", "pmc_l2"):
        f = find(raw, name, "counter_collection.csv")
        if not f:
            continue
        crow = list(csv.DictReader(open(f)))
        order = sorted({(int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in crow})
        ids = [d for d, k in order]
        names = {d: k for d, k in order}
        dmarks = [d for d in ids if dom in names[d]]
        if len(dmarks) <= W + R * K:
            continue
        lo, hi = dmarks[W], dmarks[W + R * K]
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in crow:
            d = int(r["Dispatch_Id"])
            if lo <= d < hi and "mkv::" in r["Kernel_Name"]:
                per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in per.items():
            for c, v in cs.items():
                pmc[k][c] = round(sum(v) / max(len(v), 1), 2)
                pmc[k]["_launches"] = len(v)
    summary["pmc_per_launch_timed_region"] = {k: dict(v) for k, v in pmc.items()}
    name = f"{rnd}_{tag}_timed_region.json"
    json.dump(summary, open(os.path.join(out, name), "w"), indent=1)
    # --- traffic of the dominant kernel, for bench.py (labelled with where it came from)
    domk = [k for k in pmc if dom.rstrip("<") in k]
    if domk:
        k = max(domk, key=lambda x: pmc[x].get("FETCH_SIZE", 0))
        fetch_kb, write_kb = pmc[k].get("FETCH_SIZE", 0.0), pmc[k].get("WRITE_SIZE", 0.0)
        tp = os.path.join(out, "search_traffic.json")
        d = json.load(open(tp)) if os.path.exists(tp) else {}
        kern = "k_search_mark" if "k_search_mark" in k else ("k_search_slots_mark" if "slots" in k else "k_search")
        d[f"{key}-{kern}"] = {
            "kernel": k, "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "hbm_bytes_per_launch": int((fetch_kb + write_kb) * 1024), "launches_averaged": pmc[k].get("_launches"),
            "note": "FETCH_SIZE + WRITE_SIZE (KB x 1024) per launch, separate --pmc passes, the timed region's "
                    "dispatches only; calibrated on uniform GETs: random 64-B bucket reads are tallied exactly, "
                    "the coalesced request stream at one half (gfx950 FETCH_SIZE rule) -- DESIGN.md 'measurement'",
            "l2": {c: pmc[k][c] for c in pmc[k] if c.startswith("TCC_")}, "source": name, "head": head,
            "library": library_id(raw)}
        json.dump(d, open(tp, "w"), indent=1)
    print(json.dumps(summary, indent=1)[:2500])


if __name__ == "__main__":
    main()
