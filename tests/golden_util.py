"""Reader / checker of tests/golden/index_kat.json (numpy only: shared by the CPU and GPU tests)."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_index_kat():
    with open(os.path.join(HERE, "golden", "index_kat.json")) as f:
        return json.load(f)["scenarios"]


def initial_rows(sc):
    """(bucket indices int64[k], rows uint32[k,16]) of a scenario's initial table"""
    idx = np.array(sorted(int(b) for b in sc["initial_rows"]), dtype=np.int64)
    rows = np.array([sc["initial_rows"][str(b)] for b in idx], dtype=np.uint32).reshape(-1, 16)
    return idx, rows


def blocks_of(op):
    return [np.array(b, dtype=np.uint32).reshape(-1, 3) for b in op["blocks"]]


def check_step(name, k, exp, rows_of, whole_table, out):
    """exp: the fixture's expectation after operation k.  rows_of(idx) -> uint32[len(idx),16] of the
    table under test; whole_table() -> the complete table (only called for small tables)."""
    if exp["rows"] is not None:
        idx = np.array(sorted(int(b) for b in exp["rows"]), dtype=np.int64)
        if len(idx):
            want = np.array([exp["rows"][str(b)] for b in idx], dtype=np.uint32).reshape(-1, 16)
            got = rows_of(idx)
            assert np.array_equal(got, want), (name, k, "rows", idx[(got != want).any(axis=1)][:4])
    if exp["table_sha256"] is not None:
        t = np.ascontiguousarray(whole_table(), dtype=np.uint32)
        assert hashlib.sha256(t.tobytes()).hexdigest() == exp["table_sha256"], (name, k, "table digest")
        if "table" in exp:
            assert np.array_equal(t.reshape(-1), np.array(exp["table"], dtype=np.uint32)), (name, k, "table")
    if exp["out"] is not None:
        assert np.array_equal(np.asarray(out, dtype=np.uint32), np.array(exp["out"], dtype=np.uint32).reshape(-1, 2)), \
            (name, k, "search results")
