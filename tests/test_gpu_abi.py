"""The drop-in entry points by their reference names (gpu_hash_search /
gpu_hash_insert / gpu_hash_delete, geometry baked in: MEM_P=30, HASH_CUCKOO),
driven the way src/mega_scheduler.c:393-504 drives them, at BASELINE sizes."""
import numpy as np
import pytest
import torch

import megakv_amd as M
import workload as W
from oracle import oracle as O
from tests.gpu_util import dev, host

pytestmark = pytest.mark.gpu


def test_scheduler_cycle_full_size_against_oracle():
    g = M.Geometry(mem_p=30)
    og = O.geom(mem_p=30)
    table = torch.zeros(g.ht_size // 4, dtype=torch.int32, device="cuda")   # cudaMalloc + cudaMemset 0
    exp = O.new_table(og)
    stream = torch.cuda.Stream()
    rng = np.random.default_rng(12)
    n_ins = 8 * 4096
    keys = rng.permutation(np.arange(1, 1 << 22, dtype=np.uint32))[:n_ins * 3]
    with torch.cuda.stream(stream):
        for step in range(3):
            k = keys[step * n_ins:(step + 1) * n_ins]
            e = W.make_ielem(k, None, W.KEYING_MIX)
            blocks = W.bin_inserts(e, 30, 3, W.BIN_APP)             # src/mega_recv.c:476-477
            dblocks = [dev(b) for b in blocks]
            ptrs = torch.tensor([b.data_ptr() for b in dblocks], dtype=torch.int64).cuda()
            cnts = torch.tensor([b.shape[0] for b in dblocks], dtype=torch.int32).cuda()
            # search phase of the previous keys + these (misses for the new ones)
            q = W.make_selem(keys[:(step + 1) * n_ins][-32767:], W.KEYING_MIX)
            dq = dev(q)
            out = torch.zeros((q.shape[0], 2), dtype=torch.int32, device="cuda")  # cudaMemsetAsync
            M.gpu_hash_search(dq, out, table, q.shape[0], 24576, 256, stream)
            exp_out = O.search(og, exp, q)
            # delete phase: a few of the oldest keys
            dk = W.make_ielem(keys[step * 100:(step + 1) * 100], None, W.KEYING_MIX)
            ddk = dev(dk)
            M.gpu_hash_delete(ddk, table, dk.shape[0], 16384, 256, stream)
            O.delete(og, exp, dk)
            # insert phase
            M.gpu_hash_insert(table, ptrs, cnts, 8, stream)
            O.insert(og, exp, blocks)
            stream.synchronize()                                      # cudaDeviceSynchronize
            assert np.array_equal(host(out), exp_out), f"search step {step}"
    got = table.cpu().numpy().view(np.uint32).reshape(-1, 16)
    touched = np.unique(np.nonzero(exp.any(axis=1))[0])
    assert np.array_equal(got[touched], exp[touched])
    assert int(np.count_nonzero(got)) == int(np.count_nonzero(exp))


def test_reference_wrapper_argument_checks_are_kept():
    """gpu_hash.cu:492 asserts num_thread > threads_per_blk: the HIP library
    keeps the abort.  Checked in a child process."""
    import subprocess
    import sys
    code = (
        "import torch, megakv_amd as M\n"
        "t=torch.zeros(1<<28,dtype=torch.int32,device='cuda')\n"
        "q=torch.zeros((4,2),dtype=torch.int32,device='cuda')\n"
        "o=torch.zeros((4,2),dtype=torch.int32,device='cuda')\n"
        "M.gpu_hash_search(q,o,t,4,128,256)\n"
        "torch.cuda.synchronize(); print('survived')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "survived" not in r.stdout


def test_plain_c_scheduler_driver_runs_clean():
    """examples/scheduler_cycle.c (gcc-compiled, linked against lib/libgpuhash.a by
    build()) reproduces the scheduler's call sequence on 4 streams and checks
    hits/misses itself."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "megakv_amd", "_lib", "scheduler_cycle")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    # one worker stream: every insert is ordered, the self-check must be exact
    r = subprocess.run([exe, "12", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "errors 0" in r.stdout
    # Four and sixteen worker streams like the reference (src/mega_scheduler.c:276): there the
    # inserts of different workers race (SURVEY B-1); here the library puts the updates of all
    # streams in issue order, so the self-check is exact whatever the number of streams -- and the
    # update workspaces belong to the device, not to a stream: < 2 GiB for a 1 GiB index.
    import re
    for workers in ("4", "16"):
        r = subprocess.run([exe, "12", workers], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "errors 0" in r.stdout, r.stdout + r.stderr
        mib = float(re.search(r"library memory (\d+) MiB", r.stdout).group(1))
        assert mib < 2048, r.stdout
    # the same through the reference's three entry points on the pinned host buffers themselves
    # (no staging copies: the kernels read requests and write results over PCIe)
    r = subprocess.run([exe, "12", "1", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "errors 0" in r.stdout and "no copies" in r.stdout
    # ... and with the whole cycle's library work as ONE call (mkv_cycle_workers): staged and mapped
    for mapped in ("0", "1"):
        r = subprocess.run([exe, "12", "16", mapped, "1"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "errors 0" in r.stdout and "one library call" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("workers,mapped", [("1", "0"), ("16", "0"), ("16", "1"), ("4", "0")])
def test_deferred_launch_mode_gives_an_unchanged_host_the_fused_cycle(workers, mapped):
    """MKV_DEFER=1: the three drop-in entry points only record, cudaDeviceSynchronize issues the whole
    cycle as one call (mkv_abi.hip).  The same plain-C host, the same call sequence (mode 0 of
    examples/scheduler_cycle.c: per worker and stream copy + memset + gpu_hash_search + copy back,
    gpu_hash_delete, gpu_hash_insert, one synchronisation per cycle, reference
    src/mega_scheduler.c:393-504): 0 errors against its own model and the SAME table, bit for bit, as
    the eager run."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "megakv_amd", "_lib", "scheduler_cycle")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    digests, rates = [], []
    for defer in ("0", "1"):
        env = dict(os.environ, MKV_DEFER=defer)
        r = subprocess.run([exe, "12", workers, mapped, "0", "1"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "errors 0" in r.stdout, r.stdout + r.stderr
        digests.append(re.search(r"table digest ([0-9a-f]+)", r.stdout).group(1))
        rates.append(float(re.search(r"-> ([0-9.]+) Mops", r.stdout).group(1)))
    assert digests[0] == digests[1], (digests, rates)
    print(f"workers {workers} mapped {mapped}: eager {rates[0]} Mops, deferred {rates[1]} Mops")


@pytest.mark.parametrize("workers", ["4", "16"])
def test_deferred_launch_mode_with_a_wait_per_worker_stream(workers):
    """The deferred mode issues a whole cycle -- every worker's search, the updates, the result copies -- on the
    FIRST worker's stream.  A host that waits for its streams one by one (cudaStreamSynchronize, last worker
    first) must still find each worker's results behind that worker's own stream: the flush makes every other
    recorded stream wait for an event behind the cycle (mkv_abi.hip, join_streams_locked).  Same plain-C host,
    staged copies, 0 errors against its own model and the same table digest as the eager run."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "megakv_amd", "_lib", "scheduler_cycle")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    digests = []
    for defer in ("0", "1"):
        env = dict(os.environ, MKV_DEFER=defer)
        r = subprocess.run([exe, "12", workers, "0", "0", "1", "1"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "errors 0" in r.stdout, r.stdout + r.stderr
        digests.append(re.search(r"table digest ([0-9a-f]+)", r.stdout).group(1))
    assert digests[0] == digests[1], digests


def test_cycle_captured_into_a_hipgraph_through_the_c_api():
    """examples/graph_cycle.c (gcc, HIP's C header): one mkv_cycle captured with
    hipStreamBeginCapture, replayed, against the same cycles issued eagerly from the same start:
    identical table, every GET answered with its own location."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "megakv_amd", "_lib", "graph_cycle")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, "24", "16", "50", "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "tables identical" in r.stdout and "wrong results 0" in r.stdout, r.stdout + r.stderr


def test_pipelines_come_and_go_without_leaking_device_memory():
    """The library-owned streams of a pipeline (and of a shard engine) hold no update workspace
    of their own: creating and destroying pipelines leaves the free device memory where it was."""
    import ctypes
    g = M.Geometry(mem_p=24)
    idx = M.HashIndex(g)
    L = idx.lib
    free = []
    for it in range(4):
        pipe = ctypes.c_void_p(0)
        rc = L.mkv_pipe_create(ctypes.byref(pipe), ctypes.byref(idx._g), ctypes.c_void_p(idx.table.data_ptr()),
                               1 << 16, 1 << 14, 8, 3)
        assert rc == 0, L.mkv_pipe_error(pipe)
        L.mkv_pipe_destroy(pipe)
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    assert free[-1] >= free[0] - (8 << 20), free          # (the first creation may size the workspaces)
    assert L.mkv_workspace_bytes() > 0


def test_plain_c_driver_with_other_baked_in_macros():
    """MEM_P / scheme are compile-time in the ABI (like the reference): the archive built
    with -DMEM_P=25 -DHASH_2CHOICE (BASELINE config 1's geometry) behind the same C host."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "megakv_amd", "_lib", "scheduler_cycle_p25_2choice")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, "10", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "errors 0" in r.stdout


def test_updates_of_one_table_from_several_streams_keep_issue_order():
    """The reference's scheduler updates ONE table from a stream per worker
    (src/mega_scheduler.c:393-500).  Here the updates of a table are put in the order the host
    issued them, whatever streams they are on: inserts and deletes alternating over three
    streams with no synchronisation in between, many keys shared between consecutive calls --
    the table equals the oracle's after the calls in issue order."""
    import torch
    import megakv_amd as M
    from oracle import oracle as O
    g = M.Geometry(mem_p=20)
    og = O.geom(mem_p=20)
    idx = M.HashIndex(g)
    exp = O.new_table(og)
    rng = np.random.default_rng(12)
    streams = [torch.cuda.Stream() for _ in range(3)]
    pool = rng.integers(1, 2**32, size=(30000, 3), dtype=np.uint64).astype(np.uint32)
    keep = []
    for call in range(12):
        e = pool[rng.integers(0, len(pool), size=20000)].copy()
        e[:, 2] = rng.integers(1, 2**32, size=len(e), dtype=np.uint64)        # new locations for old keys
        s = streams[call % 3]
        if call % 4 == 3:
            occ = np.argwhere(exp[:, :8] != 0)[::7]                           # delete what is stored: (sig, hash, loc)
            d = np.stack([exp[occ[:, 0], occ[:, 1]], occ[:, 0].astype(np.uint32), exp[occ[:, 0], 8 + occ[:, 1]]], axis=1)
            d = np.ascontiguousarray(d.astype(np.uint32))
            O.delete(og, exp, d)
            t = torch.from_numpy(d.view(np.int32)).cuda()
            keep.append(t)
            idx.delete(t, stream=s)
        else:
            O.insert(og, exp, [e])
            t = torch.from_numpy(e.view(np.int32)).cuda()
            keep.append(t)
            ptrs, cnts, kk = idx.stage_blocks([t])
            keep.append((ptrs, cnts, kk))
            idx.insert_staged(ptrs, cnts, 1, stream=s, hint=len(e))
    torch.cuda.synchronize()
    assert np.array_equal(idx.table_numpy(), exp)


def test_host_threads_updating_their_own_tables_share_the_workspace():
    """Four host threads, each with a stream and a table of its own, insert and delete at the same time:
    the device's one update workspace is shared, so the library lets the threads take turns for the
    stretch in which a call enqueues its kernels (ctypes releases the GIL during the calls: the host
    side really runs in parallel).  Every table equals the oracle's for its own calls."""
    import threading
    import torch
    import megakv_amd as M
    from oracle import oracle as O
    nthreads, calls = 4, 16
    g = M.Geometry(mem_p=18)
    og = O.geom(mem_p=18)
    idxs = [M.HashIndex(g) for _ in range(nthreads)]
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    plans, keep = [], [[] for _ in range(nthreads)]
    for th in range(nthreads):
        rng = np.random.default_rng(100 + th)
        pool = rng.integers(1, 2**32, size=(9000, 3), dtype=np.uint64).astype(np.uint32)
        plan = []
        for call in range(calls):
            e = pool[rng.integers(0, len(pool), size=6000)].copy()
            e[:, 2] = rng.integers(1, 2**32, size=len(e), dtype=np.uint64)
            plan.append(("delete" if call % 5 == 4 else "insert", np.ascontiguousarray(e)))
        plans.append(plan)
        for _, e in plan:                                   # staged before the threads start: only library calls race
            t = torch.from_numpy(e.view(np.int32)).cuda()
            keep[th].append((t, idxs[th].stage_blocks([t])))
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(nthreads)

    def work(th):
        try:
            start.wait()
            for (op, e), (t, (ptrs, cnts, _)) in zip(plans[th], keep[th]):
                if op == "delete":
                    idxs[th].delete(t, stream=streams[th])
                else:
                    idxs[th].insert_staged(ptrs, cnts, 1, stream=streams[th], hint=len(e))
        except Exception as ex:                             # noqa: BLE001
            errors.append((th, repr(ex)))

    ts = [threading.Thread(target=work, args=(th,)) for th in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for th in range(nthreads):
        exp = O.new_table(og)
        for op, e in plans[th]:
            if op == "delete":
                O.delete(og, exp, e)
            else:
                O.insert(og, exp, [e])
        assert np.array_equal(idxs[th].table_numpy(), exp), f"thread {th}: table differs"


def test_host_threads_running_cycles_share_the_cycle_workspace():
    """Four host threads, each with a stream and a table of its own, run mkv_cycle (search + delete +
    insert, the insert's mark pass -- and without deletes its round 1 -- inside the search launch) at
    the same time.  A cycle keeps the device's cycle workspace from its fused launch to its rounds,
    which are enqueued by later parts of the call: the whole call is one turn.  Every table and every
    result equals the oracle's for the thread's own calls."""
    import threading
    import torch
    import megakv_amd as M
    from oracle import oracle as O
    nthreads, calls = 4, 12
    g = M.Geometry(mem_p=18)
    og = O.geom(mem_p=18)
    idxs = [M.HashIndex(g) for _ in range(nthreads)]
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    plans, staged = [], [[] for _ in range(nthreads)]
    for th in range(nthreads):
        rng = np.random.default_rng(300 + th)
        pool = rng.integers(1, 2**32, size=(12000, 3), dtype=np.uint64).astype(np.uint32)
        plan = []
        for call in range(calls):
            e = pool[rng.integers(0, len(pool), size=5000)].copy()
            e[:, 2] = rng.integers(1, 2**32, size=len(e), dtype=np.uint64)
            q = np.ascontiguousarray(pool[rng.integers(0, len(pool), size=7000), :2])
            d = pool[rng.integers(0, len(pool), size=(700 if call % 3 == 2 else 0))].copy()
            plan.append((q, np.ascontiguousarray(d), np.ascontiguousarray(e)))
        plans.append(plan)
        for q, d, e in plan:                                # staged before the threads start: only library calls race
            tq = torch.from_numpy(q.view(np.int32)).cuda()
            td = torch.from_numpy(d.view(np.int32)).cuda() if len(d) else None
            te = torch.from_numpy(e.view(np.int32)).cuda()
            out = torch.full((len(q), 2), 0x5555, dtype=torch.int32, device="cuda")
            staged[th].append((tq, td, te, out, idxs[th].stage_blocks([te])))
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(nthreads)

    def work(th):
        try:
            start.wait()
            for (q, d, e), (tq, td, te, out, (ptrs, cnts, keep)) in zip(plans[th], staged[th]):
                idxs[th].cycle(tq, out, ptrs, cnts, len(keep), delem=td, stream=streams[th], insert_hint=len(e))
        except Exception as ex:                             # noqa: BLE001
            errors.append((th, repr(ex)))

    ts = [threading.Thread(target=work, args=(th,)) for th in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for th in range(nthreads):
        exp = O.new_table(og)
        for (q, d, e), (tq, td, te, out, _) in zip(plans[th], staged[th]):
            want = O.search(og, exp, q)
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want), f"thread {th}: results differ"
            if len(d):
                O.delete(og, exp, d)
            O.insert(og, exp, [e])
        assert np.array_equal(idxs[th].table_numpy(), exp), f"thread {th}: table differs"


@pytest.mark.parametrize("scheme,nworkers", [(0, 5), (1, 3), (0, 16)])
def test_cycle_workers_one_call_for_all_workers_of_a_cycle(scheme, nworkers):
    """mkv_cycle_workers: one search launch over every worker's requests, one delete pass, one insert
    pass with worker-major blocks -- against the oracle run in the documented order (every search on
    the table as the cycle found it; deletes in worker order; inserts in worker, block, index order).
    The workers' keys overlap on purpose: the order is part of the contract."""
    import ctypes
    from megakv_amd.lib import MkvWorker
    from megakv_amd.index import _dev_ptr, _stream_ptr
    g = M.Geometry(mem_p=19, scheme=scheme)
    og = O.geom(mem_p=19, scheme=scheme)
    idx = M.HashIndex(g)
    exp = O.new_table(og)
    rng = np.random.default_rng(900 + nworkers)
    pool = rng.integers(1, 2**32, size=(40000, 3), dtype=np.uint64).astype(np.uint32)
    stream = torch.cuda.Stream()
    for cycle in range(4):
        ws, keep, host_w = (MkvWorker * nworkers)(), [], []
        for w in range(nworkers):
            ns, nd = int(rng.integers(0, 4000)), int(rng.integers(0, 600)) if cycle else 0
            q = np.ascontiguousarray(pool[rng.integers(0, len(pool), size=ns), :2])
            dl = pool[rng.integers(0, len(pool), size=nd)].copy()
            nb = 8
            blocks = []
            for b in range(nb):
                e = pool[rng.integers(0, len(pool), size=int(rng.integers(0, 700)))].copy()
                e[:, 2] = rng.integers(1, 2**32, size=len(e), dtype=np.uint64)
                blocks.append(np.ascontiguousarray(e))
            tq = dev(q) if ns else None
            out = torch.full((max(ns, 1), 2), 0x5555, dtype=torch.int32, device="cuda")
            td = dev(dl) if nd else None
            tb = [dev(b.reshape(-1, 3)) for b in blocks]
            ptrs, cnts, kp = idx.stage_blocks(tb)
            keep.append((tq, out, td, tb, ptrs, cnts, kp))
            host_w.append((q, dl, blocks))
            ws[w].search_in = _dev_ptr(tq).value if ns else None
            ws[w].search_out = _dev_ptr(out).value
            ws[w].num_search = ns
            ws[w].delete_in = _dev_ptr(td).value if nd else None
            ws[w].num_delete = nd
            ws[w].blk_input = _dev_ptr(ptrs).value
            ws[w].blk_elem_num = _dev_ptr(cnts).value
            ws[w].num_blks = nb
        torch.cuda.synchronize()
        rc = idx.lib.mkv_cycle_workers(ctypes.byref(idx._g), _dev_ptr(idx.table), ws, nworkers, _stream_ptr(stream))
        assert rc == 0
        stream.synchronize()
        for w in range(nworkers):
            q = host_w[w][0]
            if len(q):
                want = O.search(og, exp, q)
                assert np.array_equal(keep[w][1].cpu().numpy().view(np.uint32)[: len(q)], want), (cycle, w, "search")
        for w in range(nworkers):
            if len(host_w[w][1]):
                O.delete(og, exp, np.ascontiguousarray(host_w[w][1]))
        O.insert(og, exp, [b for w in range(nworkers) for b in host_w[w][2]])
        assert np.array_equal(idx.table_numpy(), exp), (cycle, "table")


def test_cycle_workers_calls_on_alternating_streams_keep_issue_order():
    """Calls of mkv_cycle_workers on two streams, back to back with no host synchronisation between
    them: the gathered block lists are one set per device, so a call may only rewrite them behind the
    device's last update -- the library orders the calls like any other updates.  Final table against
    the oracle run in issue order."""
    import ctypes
    from megakv_amd.lib import MkvWorker
    from megakv_amd.index import _dev_ptr, _stream_ptr
    g = M.Geometry(mem_p=18)
    og = O.geom(mem_p=18)
    idx = M.HashIndex(g)
    exp = O.new_table(og)
    rng = np.random.default_rng(77)
    pool = rng.integers(1, 2**32, size=(30000, 3), dtype=np.uint64).astype(np.uint32)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    nworkers, calls, keep = 4, [], []
    for c in range(8):
        ws = (MkvWorker * nworkers)()
        host = []
        for w in range(nworkers):
            nd = int(rng.integers(0, 400)) if c else 0
            dl = pool[rng.integers(0, len(pool), size=nd)].copy()
            blocks = []
            for b in range(4):
                e = pool[rng.integers(0, len(pool), size=int(rng.integers(1, 900)))].copy()
                e[:, 2] = rng.integers(1, 2**32, size=len(e), dtype=np.uint64)
                blocks.append(np.ascontiguousarray(e))
            td = dev(dl) if nd else None
            tb = [dev(b.reshape(-1, 3)) for b in blocks]
            ptrs, cnts, kp = idx.stage_blocks(tb)
            keep.append((td, tb, ptrs, cnts, kp))
            host.append((dl, blocks))
            ws[w].search_in = None
            ws[w].search_out = None
            ws[w].num_search = 0
            ws[w].delete_in = _dev_ptr(td).value if nd else None
            ws[w].num_delete = nd
            ws[w].blk_input = _dev_ptr(ptrs).value
            ws[w].blk_elem_num = _dev_ptr(cnts).value
            ws[w].num_blks = 4
        calls.append((ws, host))
    torch.cuda.synchronize()
    for c, (ws, host) in enumerate(calls):                   # issued back to back, streams alternating
        rc = idx.lib.mkv_cycle_workers(ctypes.byref(idx._g), _dev_ptr(idx.table), ws, nworkers, _stream_ptr(streams[c & 1]))
        assert rc == 0
    torch.cuda.synchronize()
    for ws, host in calls:
        for dl, _ in host:
            if len(dl):
                O.delete(og, exp, np.ascontiguousarray(dl))
        O.insert(og, exp, [b for _, blocks in host for b in blocks])
    assert np.array_equal(idx.table_numpy(), exp)


@pytest.mark.filterwarnings("ignore:The CUDA Graph is empty")      # (the refused call leaves the first capture empty: the point)
def test_update_under_capture_behind_another_streams_update_is_refused():
    """Updates issued on different streams are put in issue order with a library-owned event or a
    device synchronisation.  Neither may go into a hipGraph: an update on a stream that is being
    captured, with the device's last update on ANOTHER stream, is refused (nothing enqueued, the
    capture stays valid); once the stream has had an update of its own before the capture, the same
    call is captured, and replaying the graph inserts."""
    import numpy as np
    import torch
    from megakv_amd import Geometry, HashIndex
    from megakv_amd.index import MkvError
    from oracle import oracle as O
    from tests.gpu_util import dev, ogeom, random_requests
    g = Geometry(mem_p=16)
    og = ogeom(g)
    rng = np.random.default_rng(5)
    idx = HashIndex(g)
    exp = O.new_table(og)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    idx.reserve(8192, s1)
    idx.reserve(8192, s2)
    e1, e2 = random_requests(g, rng, 3000, 1 << 31), random_requests(g, rng, 3000, 1 << 31)
    p1, c1, k1 = idx.stage_blocks([dev(e1)])
    p2, c2, k2 = idx.stage_blocks([dev(e2)])
    torch.cuda.synchronize()
    idx.insert_staged(p1, c1, 1, s1, 3000)               # the device's last update: stream 1
    O.insert(og, exp, [e1])
    s1.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s2):
        graph.capture_begin()
        try:
            with pytest.raises(MkvError):
                idx.insert_staged(p2, c2, 1, s2, 3000)   # would need ordering behind stream 1: refused
        finally:
            graph.capture_end()
    torch.cuda.synchronize()
    assert np.array_equal(idx.table_numpy(), exp)        # nothing of it ran
    # one update of its own on stream 2 (ordered behind stream 1's by the library), then the capture
    e3 = random_requests(g, rng, 10, 1 << 31)
    p3, c3, k3 = idx.stage_blocks([dev(e3)])
    torch.cuda.synchronize()
    idx.insert_staged(p3, c3, 1, s2, 10)
    O.insert(og, exp, [e3])
    s2.synchronize()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s2):
        graph2.capture_begin()
        idx.insert_staged(p2, c2, 1, s2, 3000)
        graph2.capture_end()
    torch.cuda.synchronize()
    assert np.array_equal(idx.table_numpy(), exp)        # captured, not run
    graph2.replay()
    torch.cuda.synchronize()
    O.insert(og, exp, [e2])
    assert np.array_equal(idx.table_numpy(), exp)


def test_deferred_mode_keeps_stream_order_for_a_search_behind_its_own_insert():
    """Deferred mode: insert A, search A, insert B, search A+B on ONE stream with no wait in between.  A flush runs
    every recorded search ahead of every recorded update, so a search recorded behind an update of its own stream
    first issues what is recorded (mkv_abi.hip, defer_slot): each search sees the inserts issued before it, as
    the eager calls (and the reference's, stream order) do."""
    from megakv_amd.lib import load_library
    L = load_library()
    g = M.Geometry(mem_p=30)
    table = torch.zeros(g.ht_size // 4, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    rng = np.random.default_rng(77)
    keys = rng.permutation(np.arange(1, 1 << 22, dtype=np.uint32))[:16384]
    keep, outs, qs = [], [], []
    L.mkv_set_defer(1)
    try:
        with torch.cuda.stream(stream):
            for step in range(2):
                e = W.make_ielem(keys[step * 8192:(step + 1) * 8192], None, W.KEYING_MIX)
                blocks = [dev(b) for b in W.bin_inserts(e, 30, 3, W.BIN_APP)]
                ptrs = torch.tensor([b.data_ptr() for b in blocks], dtype=torch.int64).cuda()
                cnts = torch.tensor([b.shape[0] for b in blocks], dtype=torch.int32).cuda()
                q = W.make_selem(keys[:(step + 1) * 8192], W.KEYING_MIX)
                dq = dev(q)
                out = torch.zeros((q.shape[0], 2), dtype=torch.int32, device="cuda")
                keep += [blocks, ptrs, cnts, dq]
                torch.cuda.synchronize()                 # (the staging above is torch's; nothing of the library waits)
                M.gpu_hash_insert(table, ptrs, cnts, 8, stream)
                M.gpu_hash_search(dq, out, table, q.shape[0], 24576, 256, stream)
                outs.append(out)
                qs.append(keys[:(step + 1) * 8192])
        assert L.mkv_flush() == 0
        torch.cuda.synchronize()
    finally:
        L.mkv_set_defer(0)
    for out, k in zip(outs, qs):
        o = host(out)
        loc = np.where(o[:, 0] != 0, o[:, 0], o[:, 1])
        assert np.array_equal(loc, k), "a search recorded behind an insert of its own stream did not see it"
