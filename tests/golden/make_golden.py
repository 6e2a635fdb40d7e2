"""Regenerates tests/golden/zipf_kat.json from the REFERENCE's own generator.

Runs oracle/_ref/ref_zipf (built by `make -C oracle ref` from
/root/reference/src/zipf.h, in place) -- so this only works in the build
container; the JSON it writes is the committed fixture the GPU box uses.
"""
import hashlib
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_zipf")

# (n, theta, seed): the app's call is n = preload_cnt - 2, seed 21
# (src/mega_recv.c:659-663); MEM_P=30 -> 26843543, MEM_P=25 -> 838858.
CASES = [(26843543, 0.99, 21), (26843543, 0.0, 21), (838858, 0.99, 21), (838858, 0.0, 21),
         (1000, 0.5, 7), (1000, -1.0, 998)]
COUNT = 100000


def main():
    out = []
    for n, theta, seed in CASES:
        txt = subprocess.check_output([BIN, str(n), repr(theta), str(seed), str(COUNT)], text=True)
        lines = txt.strip().split("\n")
        ranks = [int(x) for x in lines[:COUNT]]
        zetan, eta, thres = (float(x) for x in lines[COUNT].split())
        digest = hashlib.sha256(b"".join(r.to_bytes(8, "little") for r in ranks)).hexdigest()
        out.append({"n": n, "theta": theta, "seed": seed, "first": ranks[:64], "count": COUNT,
                    "sha256_le64": digest, "zetan": zetan, "eta": eta, "thres": thres})
    with open(os.path.join(HERE, "zipf_kat.json"), "w") as f:
        json.dump({"source": "reference src/zipf.h via oracle/ref_zipf_driver.c", "cases": out}, f, indent=1)


if __name__ == "__main__":
    main()
