"""Parity on the BASELINE.json workloads at FULL size, through bench.py's own generator and
engine: MEM_P=30 (2^30-byte table), 26 843 544 keys pre-loaded to load factor 0.2 through
the insert kernel, 2^20-request batches driven exactly as the timed region of bench.py
drives them (bench.Stream -> SingleGpu.stage -> SingleGpu.step = one mkv_cycle), checked
against the CPU oracle after EVERY batch: search results bit for bit, the whole 1 GiB table
image bit for bit.

  configs[1]  100 % GET, uniform keys                      (test_config1_*)
  configs[2]  95/5 GET/SET, Zipf-0.99, S-new and S-upd     (test_config2_*), K-mix and K-ref
  configs[3]  50/50 GET/SET, Zipf-0.99, cuckoo AND 2-choice, S-new and S-upd (test_config3_*)
  + load factor 0.5 (eviction chains and their conflict handling at full size)
  configs[4]  8 shards, Zipf 95/5: 8 ranks sharing the one GPU against ONE aggregate oracle
              table (tests/sharded_check.py)

The reference's own regime for this: libgpuhash/test/insert_test.c:111-244 (insert, then
search-verify, at growing load), src/mega_recv.c:633-768 (the LOCAL_TEST stream)."""
import numpy as np
import pytest
import torch

import bench
import workload as W
from oracle import oracle as O

pytestmark = pytest.mark.gpu

MEM_P = 30
BATCH_LOG2 = 20


def _args(workload, keying, sets, scheme, lf):
    return bench.parse(["--workload", workload, "--keying", keying, "--sets", sets, "--scheme", scheme,
                        "--lf", str(lf), "--mem-p", str(MEM_P), "--batch-log2", str(BATCH_LOG2)])


class Loaded:
    """A pre-loaded index on the GPU and its twin on the oracle."""

    def __init__(self, keying, lf, scheme):
        bench.late_imports()
        import megakv_amd as M
        self.keying, self.lf, self.scheme = keying, lf, scheme
        self.args = _args("zipf95", keying, "new", scheme, lf)
        sc = M.SCHEME_CUCKOO if scheme == "cuckoo" else M.SCHEME_2CHOICE
        self.geom = M.Geometry(mem_p=MEM_P, scheme=sc)
        self.engine = bench.SingleGpu(M.HashIndex(self.geom, device="cuda:0"), self.args)
        self.og = O.geom(mem_p=MEM_P, scheme=O.SCHEME_CUCKOO if scheme == "cuckoo" else O.SCHEME_2CHOICE)
        self.table = O.new_table(self.og)
        sg = bench.Stream(self.args, 0, 1, MEM_P)
        self.P, self.kid = sg.P, sg.keying
        self.next_set = sg.P                      # S-new ids go on from here across the tests of one table
        self.engine.preload(self.P, self.kid)     # bench.py's pre-load: 2^20-key calls of the insert kernel
        step = 1 << 22
        for lo in range(1, self.P, step):
            k = np.arange(lo, min(lo + step, self.P), dtype=np.uint32)
            O.insert(self.og, self.table, W.bin_inserts(W.make_ielem(k, None, self.kid), MEM_P, 3, W.BIN_REGION))
        self.stream = torch.cuda.Stream()
        self.engine.reserve(self.stream)
        self.check_table("after the pre-load")

    def check_table(self, what):
        got = self.engine.table_host()
        if not np.array_equal(got, self.table):
            d = np.flatnonzero((got != self.table).any(axis=1))
            raise AssertionError(f"{what}: table image differs from the oracle's in {len(d)} buckets, first "
                                 f"{int(d[0])}: gpu {got[d[0]].tolist()} oracle {self.table[d[0]].tolist()}")

    def run(self, workload, sets, nbatches, seed_shift=0):
        """nbatches consecutive batches through mkv_cycle, as bench.SingleGpu._launch issues them."""
        args = _args(workload, self.keying, sets, self.scheme, self.lf)
        self.engine.args = args
        sg = bench.Stream(args, seed_shift, 1, MEM_P)          # (rank only shifts the Zipf seeds)
        sg.next_set = self.next_set
        stats = {"gets": 0, "hits": 0, "sets": 0, "evictions": 0}
        for i in range(nbatches):
            gets, sets_ = sg.batch()
            b = self.engine.stage(gets, sets_)
            with torch.cuda.stream(self.stream):
                self.engine.step(b, self.stream, None)
            self.stream.synchronize()
            got = b["out"].cpu().numpy().view(np.uint32)
            exp = O.search(self.og, self.table, gets)
            assert np.array_equal(got, exp), \
                f"{workload}/{sets} batch {i}: {(got != exp).any(axis=1).sum()} search results differ from the oracle's"
            if sets_.shape[0]:
                st = O.insert(self.og, self.table, W.bin_inserts(sets_, MEM_P, 3, W.BIN_REGION))
                stats["evictions"] += st["evictions"]
            self.check_table(f"{workload}/{sets} after batch {i}")
            stats["gets"] += gets.shape[0]
            stats["hits"] += int(((exp[:, 0] != 0) | (exp[:, 1] != 0)).sum())
            stats["sets"] += sets_.shape[0]
        self.next_set = sg.next_set
        return stats


@pytest.fixture(scope="module")
def loaded(request):
    # (module scope + parameters: pytest groups the tests by table and tears one down before it
    # builds the next -- one 1 GiB table at a time on the device and on the host)
    state = Loaded(*request.param)
    yield state
    state.engine = state.table = None
    import gc
    gc.collect()
    torch.cuda.empty_cache()


MIX = ("mix", 0.2, "cuckoo")
REF = ("ref", 0.2, "cuckoo")
TWO = ("mix", 0.2, "2choice")
HALF = ("mix", 0.5, "cuckoo")


@pytest.mark.parametrize("loaded", [MIX, REF], indirect=True)
def test_config1_get100_uniform(loaded):
    """configs[1]: 100 % GET, uniform keys over the pre-loaded ids: every GET hits, the table
    does not change."""
    st = loaded.run("get100", "new", 3)
    assert st["hits"] == st["gets"] == 3 << BATCH_LOG2


@pytest.mark.parametrize("loaded", [MIX, REF], indirect=True)
def test_config2_zipf95_new_keys(loaded):
    """configs[2] as the reference's LOCAL_TEST stream builds it: Zipf-0.99 GETs, fresh
    sequential SET keys (S-new)."""
    st = loaded.run("zipf95", "new", 3)
    assert st["sets"] == 3 * 52428 and st["hits"] == st["gets"]


@pytest.mark.parametrize("loaded", [MIX], indirect=True)
def test_config2_zipf95_updates(loaded):
    """configs[2] with SET keys drawn from the Zipf stream too (S-upd): thousands of SETs of
    one hot key in a batch -- the last one in canonical order must win."""
    loaded.run("zipf95", "upd", 3, seed_shift=1)


@pytest.mark.parametrize("loaded", [MIX, TWO], indirect=True)
def test_config3_zipf50_both_schemes(loaded):
    """configs[3]: 50/50 GET/SET, 524 288 inserts per batch, cuckoo and two-choice."""
    st = loaded.run("zipf50", "new", 3, seed_shift=2)
    assert st["sets"] == 3 << (BATCH_LOG2 - 1)


@pytest.mark.parametrize("loaded", [MIX, TWO], indirect=True)
def test_config3_zipf50_updates(loaded):
    """configs[3] with Zipf SET keys (S-upd): 524 288 SETs of 224 K distinct keys per batch --
    300 K duplicates go through the tail's grid-wide rounds (registrations folded per
    work-group, last-writer elections, hand-over to the in-group rounds)."""
    st = loaded.run("zipf50", "upd", 2, seed_shift=5)
    assert st["sets"] == 2 << (BATCH_LOG2 - 1)


@pytest.mark.parametrize("loaded", [HALF], indirect=True)
def test_load_factor_half_chains_at_full_size(loaded):
    """Load factor 0.5 (67 M keys pre-loaded): inserts meet full buckets, eviction chains run
    and collide with other requests of their batch -- 95/5 and 50/50 batches."""
    a = loaded.run("zipf95", "new", 3, seed_shift=3)
    b = loaded.run("zipf50", "new", 2, seed_shift=4)
    assert a["evictions"] + b["evictions"] > 1000, "this case is here for the eviction chains"


def test_config4_eight_shards_on_one_gpu():
    """configs[4]: aggregate table sharded 8 ways by the top bucket bits, every rank feeding
    its own Zipf-0.99 95/5 batches (bench.Stream with its rank), requests exchanged, served
    by the owning shard, results returned -- 8 ranks sharing the one GPU of the box (the
    exchange staged through host memory; RCCL needs a GPU per rank), two batches in flight,
    against ONE aggregate oracle table of MEM_P = 24 + 3."""
    from tests.sharded_check import run_and_check_zipf
    run_and_check_zipf(world=8, mem_p_local=24, batch_log2=16, steps=4, timeout=420)
