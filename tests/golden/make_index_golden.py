"""Writes tests/golden/index_kat.json: small scenarios of the index path as DATA -- geometry,
initial table rows, the operations, and the expected table rows / search results after each.

The reference's kernels cannot run in the build container (nvcc / CUDA runtime are absent), so
the expected values come from the CPU oracle.  The scenarios named survey_* are the behaviours
SURVEY section 8(c) recorded from the reference's own code; their known answers are asserted
HERE, before anything is written, so the file can only be produced by an oracle that reproduces
them.  The remaining scenarios are regression vectors: they pin the oracle and the HIP library
to each other and to this file (tests/test_oracle_kat.py on the CPU, tests/test_gpu_parity.py
on the GPU).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def full_bucket(rows, b, base_sig, base_loc):
    rows[b] = list(range(base_sig, base_sig + 8)) + list(range(base_loc, base_loc + 8))


def run(sc):
    """Applies a scenario's operations with the oracle; returns (expected rows per step, table)."""
    g = O.geom(mem_p=sc["mem_p"], scheme=sc["scheme"], iblock_p=sc["iblock_p"], max_cuckoo=sc["max_cuckoo"])
    t = O.new_table(g)
    for b, row in sc["initial_rows"].items():
        t[int(b)] = np.array(row, dtype=np.uint32)
    init = t.copy() if t.shape[0] <= 4096 else None
    touched = set(int(b) for b in sc["initial_rows"]) if len(sc["initial_rows"]) <= 16 else set()
    steps = []
    for op in sc["ops"]:
        kind = op["op"]
        if kind == "insert":
            blocks = [np.array(b, dtype=np.uint32).reshape(-1, 3) for b in op["blocks"]]
            before = None if init is None else t.copy()
            O.insert(g, t, blocks)
            res = None
        elif kind == "delete":
            before = None if init is None else t.copy()
            O.delete(g, t, np.array(op["requests"], dtype=np.uint32).reshape(-1, 3))
            res = None
        else:
            before = None
            q = np.array(op["requests"], dtype=np.uint32).reshape(-1, 2)
            res = O.search(g, t, q).tolist()
        if before is not None:
            touched |= set(np.nonzero((before != t).any(axis=1))[0].tolist())
        for b in op.get("watch", []):
            touched.add(int(b))
        # few rows: the rows themselves; a whole small table: its digest (the last step also carries the table)
        few = len(touched) <= 16
        steps.append({"rows": {str(b): t[b].tolist() for b in sorted(touched)} if few else None,
                      "table_sha256": hashlib.sha256(t.tobytes()).hexdigest() if init is not None else None,
                      "out": res})
    if init is not None:
        steps[-1]["table"] = t.reshape(-1).tolist()
    return steps, t


def main():
    rng = np.random.default_rng(20261002)
    scs = []

    # ---- the victim moves with the INSERTER's hash (gpu_hash.cu:334-335)
    rows = {}
    full_bucket(rows, 5, 0x5000, 100)
    full_bucket(rows, 0x206, 0x1000, 200)
    scs.append({"name": "survey_victim_relocation", "ref": "gpu_hash.cu:301,334-335,360; SURVEY 8(c)(1)",
                "mem_p": 30, "scheme": 0, "iblock_p": 3, "max_cuckoo": 5, "initial_rows": rows,
                "ops": [{"op": "insert", "blocks": [[[0x203, 5, 777]]], "watch": [0x206, 0x1006, 0x1205, 5]},
                        {"op": "search", "requests": [[0x1003, 0x206], [0x203, 5]]}]})
    # ---- two-choice overflow writes the signature, keeps the victim's loc (:197-209)
    scs.append({"name": "survey_2choice_overflow_stale_loc", "ref": "gpu_hash.cu:197-209; SURVEY 8(c)(2)",
                "mem_p": 30, "scheme": 1, "iblock_p": 3, "max_cuckoo": 5, "initial_rows": dict(rows),
                "ops": [{"op": "insert", "blocks": [[[0x203, 5, 777]]], "watch": [0x206, 5]}]})
    # ---- slot preference -- lowest empty lane >= sig & 7, else lowest empty lane
    scs.append({"name": "survey_slot_preference", "ref": "gpu_hash.cu:301-323; SURVEY 8(c)(3)",
                "mem_p": 20, "scheme": 0, "iblock_p": 3, "max_cuckoo": 5, "initial_rows": {},
                "ops": [{"op": "insert", "blocks": [[[0x100 * (i + 1) + 6, 9, i + 1]]], "watch": [9]} for i in range(4)]})
    # ---- duplicate overwrites loc; delete needs sig AND loc, zeroes sig only;
    #      a search with signature 0 matches emptied slots
    scs.append({"name": "survey_duplicate_delete_sig0", "ref": "gpu_hash.cu:283-296,459-468,50-62; SURVEY 8(c)(4)-(6)",
                "mem_p": 20, "scheme": 0, "iblock_p": 3, "max_cuckoo": 5, "initial_rows": {},
                "ops": [{"op": "insert", "blocks": [[[0x77, 12, 1]]], "watch": [12]},
                        {"op": "insert", "blocks": [[[0x77, 12, 2]]]},
                        {"op": "delete", "requests": [[0x77, 12, 1]]},          # wrong loc: stays
                        {"op": "search", "requests": [[0x77, 12]]},
                        {"op": "delete", "requests": [[0x77, 12, 2]]},          # sig zeroed, loc left behind
                        {"op": "search", "requests": [[0x77, 12], [0, 12]]}]})
    # ---- chain of MAX_CUCKOO_NUM relocations, the last victim dropped
    rows = {}
    for b in range(64):
        full_bucket(rows, b, 0x100 * (b + 1), 0x1000 * (b + 1))
    # (every victim of this table has sig & 7 = its slot: the chain ping-pongs between bucket 6 and
    # itself and the key that is finally dropped is the NEW one -- the table ends as it began)
    scs.append({"name": "chain_drops_last_victim", "ref": "gpu_hash.cu:325-433; SURVEY A.2",
                "mem_p": 12, "scheme": 0, "iblock_p": 3, "max_cuckoo": 5, "initial_rows": rows,
                "ops": [{"op": "insert", "blocks": [[[0xABC5, 3, 42]]], "watch": [3, 6]},
                        {"op": "search", "requests": [[0xABC5, 3], [0x705, 6]]}]})
    # the same with signatures whose low bits differ from their slot: the chain walks on
    rows2 = {}
    for b in range(64):
        rows2[b] = [0x100 * (b + 1) + ((j + b) & 7) for j in range(8)] + list(range(0x1000 * (b + 1), 0x1000 * (b + 1) + 8))
    scs.append({"name": "chain_walks_and_drops", "ref": "gpu_hash.cu:325-433; SURVEY A.2",
                "mem_p": 12, "scheme": 0, "iblock_p": 3, "max_cuckoo": 5, "initial_rows": rows2,
                "ops": [{"op": "insert", "blocks": [[[0xABC5, 3, 42]]], "watch": [3, 6]},
                        {"op": "search", "requests": [[0xABC5, 3]]}]})
    # ---- regression vectors: mixed sequences on small, well-filled tables (hot keys, zero signatures)
    for name, scheme, mem_p, iblock_p, fill in (("mixed_cuckoo", 0, 13, 2, 0.8), ("mixed_2choice", 1, 13, 3, 0.85),
                                                ("mixed_cuckoo_one_region", 0, 12, 0, 0.9)):
        nb = 1 << (mem_p - 6)
        rows = {}
        for b in range(nb):
            occ = rng.random(8) < fill
            sig = np.where(occ, rng.integers(1, 60, size=8), 0)
            loc = np.where(occ, rng.integers(1, 2**32, size=8, dtype=np.uint64), 0)
            rows[b] = [int(x) for x in sig] + [int(x) for x in loc]
        ops = []
        for step in range(3):
            n = 300
            e = np.stack([rng.integers(0, 60, size=n), rng.integers(0, 2**32, size=n, dtype=np.uint64),
                          rng.integers(0, 2**32, size=n, dtype=np.uint64)], axis=1).astype(np.uint32)
            e[: n // 5, :2] = e[0, :2]                                       # a hot key
            region = ((e[:, 1] & (nb - 1)) >> (mem_p - 6 - iblock_p)).astype(int)
            ops.append({"op": "insert", "blocks": [e[region == b].tolist() for b in range(1 << iblock_p)]})
            ops.append({"op": "search", "requests": e[::3, :2].tolist()})
            ops.append({"op": "delete", "requests": e[1::4].tolist()})
        scs.append({"name": name, "ref": "regression vector (oracle)", "mem_p": mem_p, "scheme": scheme,
                    "iblock_p": iblock_p, "max_cuckoo": 5, "initial_rows": rows, "ops": ops})

    out = []
    for sc in scs:
        sc["initial_rows"] = {str(k): v for k, v in sc["initial_rows"].items()}
        steps, t = run(sc)
        sc["expected"] = steps
        out.append(sc)

    # ---- the recorded answers, asserted before the file is written
    by = {s["name"]: s for s in out}
    r = by["survey_victim_relocation"]["expected"]
    assert r[0]["rows"]["518"][3] == 0x203 and r[0]["rows"]["518"][11] == 777           # 0x206
    assert r[0]["rows"]["4102"][3] == 0x1003 and r[0]["rows"]["4102"][11] == 203         # 0x1006 = 5 ^ 0x1003
    assert sum(r[0]["rows"]["4613"]) == 0                                                 # 0x1205 untouched
    assert r[1]["out"] == [[0, 0], [0, 777]]                                              # victim unreachable
    r = by["survey_2choice_overflow_stale_loc"]["expected"]
    assert r[0]["rows"]["518"][3] == 0x203 and r[0]["rows"]["518"][11] == 203
    r = by["survey_slot_preference"]["expected"]
    assert [r[i]["rows"]["9"].index(0x100 * (i + 1) + 6) for i in range(4)] == [6, 7, 0, 1]
    r = by["survey_duplicate_delete_sig0"]["expected"]
    assert r[1]["rows"]["12"][8 + 7] == 2 and r[3]["out"] == [[2, 0]]
    assert r[4]["rows"]["12"][7] == 0 and r[4]["rows"]["12"][15] == 2 and r[5]["out"][0] == [0, 0] and r[5]["out"][1][0] == 2

    with open(os.path.join(HERE, "index_kat.json"), "w") as f:
        json.dump({"source": "CPU oracle (oracle/gpuhash_oracle.c); survey_* answers are SURVEY 8(c)'s, asserted at generation",
                   "row_layout": "sig[0..7] then loc[0..7]", "scenarios": out}, f, separators=(",", ":"))
    print("scenarios:", len(out), "bytes:", os.path.getsize(os.path.join(HERE, "index_kat.json")))


if __name__ == "__main__":
    main()
