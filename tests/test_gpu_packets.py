"""SURVEY 8(f) row 3 on the GPU: mkv_parse_packets against the oracle's restatement of the
receiver's parser (src/mega_recv.c:292-545), on synthesised LOCAL_TEST packets and on
malformed ones, and packets -> requests -> index end to end on the device."""
import ctypes

import numpy as np
import pytest
import torch

from megakv_amd import Geometry, HashIndex
import workload as W
from megakv_amd.index import _dev_ptr, _stream_ptr
from megakv_amd.lib import load_library
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def device_parse(pay, lens, key_len, value_len, fold=0, cap=None, with_lens=True):
    L = load_library()
    npk, stride = pay.shape
    cap = cap if cap is not None else npk * 400
    d_pay = torch.from_numpy(pay).cuda()
    d_len = torch.from_numpy(lens).cuda()
    gets = torch.zeros((max(cap, 1), 2), dtype=torch.int32, device="cuda")
    gpk = torch.full((max(cap, 1),), -1, dtype=torch.int32, device="cuda")
    sets = torch.zeros((max(cap, 1), 3), dtype=torch.int32, device="cuda")
    counts = torch.zeros((4,), dtype=torch.int64, device="cuda")
    nb = int(L.mkv_parse_scratch_bytes(npk))
    scratch = torch.zeros((nb // 4 + 4,), dtype=torch.int32, device="cuda")
    rc = L.mkv_parse_packets(_dev_ptr(d_pay), stride, _dev_ptr(d_len) if with_lens else None, npk, key_len,
                             value_len, fold, _dev_ptr(gets), _dev_ptr(gpk), _dev_ptr(sets), cap, cap,
                             _dev_ptr(counts), _dev_ptr(scratch), scratch.numel() * 4, _stream_ptr(None))
    assert rc == 0
    torch.cuda.synchronize()
    c = [int(x) for x in counts.cpu().numpy()]
    g = gets.cpu().numpy().view(np.uint32)[:min(c[0], cap)]
    s = sets.cpu().numpy().view(np.uint32)[:min(c[1], cap)]
    return g, s, gpk.cpu().numpy()[:min(c[0], cap)], c


@pytest.mark.parametrize("num_get,num_set,key_len,value_len,fold", [(95, 5, 8, 8, 0), (100, 0, 8, 8, 0),
                                                                     (50, 50, 16, 64, 1), (0, 3, 8, 8, 0),
                                                                     (95, 5, 32, 512, 1), (1, 1, 128, 1024, 1)])
def test_parse_matches_receiver_rule(num_get, num_set, key_len, value_len, fold):
    P = W.preload_count(25, 0.2)
    syn = W.PacketSynth(P - 2, 0.99, 21, P, num_get, num_set, key_len, value_len, iblock_p=3)
    pay, lens, gk, sk = syn.packets(1000)
    if key_len > 8:                                  # bytes behind the first 8 matter with `fold`
        rng = np.random.default_rng(5)
        noise = rng.integers(0, 256, size=pay.shape, dtype=np.uint8)
        jobs = np.zeros_like(pay, dtype=bool)
        for p in range(0, len(lens), 7):             # a few packets: fill key tails of their first job
            off = 2 + 4 + 8 if num_get else 2 + 8 + 8
            jobs[p, off:off + key_len - 8] = True
        pay = np.where(jobs, noise, pay)
    eg, es, ebp, ebj = O.parse_packets(pay, lens, key_len, value_len, bool(fold))
    g, s, gpk, c = device_parse(pay, lens, key_len, value_len, fold)
    assert c == [len(eg), len(es), ebp, ebj]
    assert np.array_equal(g, eg) and np.array_equal(s, es)
    assert (np.diff(gpk) >= 0).all()                 # arrival order: packet, then job


def test_parse_malformed_packets_like_the_reference():
    rng = np.random.default_rng(11)
    P = W.preload_count(25, 0.2)
    pay, lens, _, _ = W.PacketSynth(P - 2, 0.99, 21, P, 95, 5).packets(600)
    pay = pay.copy()
    for p in range(0, 600, 3):                       # every third packet gets one kind of damage
        kind = (p // 3) % 5
        if kind == 0:
            pay[p, 0] ^= 0xFF                        # wrong magic
        elif kind == 1:
            pay[p, 2 + 12 * int(rng.integers(0, 50))] = 9      # unknown job type somewhere
        elif kind == 2:
            pay[p, 4 + 12 * int(rng.integers(0, 50))] = 7      # a GET with key length 7
        elif kind == 3:
            lens[p] = int(rng.integers(4, lens[p] - 2))         # truncated
        else:
            pay[p, int(rng.integers(2, lens[p])):] = 0          # garbage tail
    eg, es, ebp, ebj = O.parse_packets(pay, lens, 8, 8)
    g, s, _, c = device_parse(pay, lens, 8, 8)
    assert c == [len(eg), len(es), ebp, ebj] and ebp > 100 and ebj > 10
    assert np.array_equal(g, eg) and np.array_equal(s, es)
    # capacities: what does not fit is dropped, the counts still tell
    g2, s2, _, c2 = device_parse(pay, lens, 8, 8, cap=1000)
    assert c2[:2] == c[:2] and np.array_equal(g2, eg[:1000]) and np.array_equal(s2, es[:min(1000, len(es))])


def test_packets_to_index_on_the_device():
    """LOCAL_TEST packets -> parse -> bin SETs into insert blocks -> insert -> search the GETs of
    the same packets: all on the device; table and results equal the oracle's."""
    L = load_library()
    g = Geometry(mem_p=20, scheme=0)
    og = O.geom(mem_p=20, scheme=0)
    P = W.preload_count(20, 0.2)
    idx = HashIndex(g)
    table = O.new_table(og)
    # pre-load keys 1..P-1 through packets of SETs only (config 1: "100 % SET preload")
    pay, lens, _, sk = W.PacketSynth(P - 2, 0.99, 21, 1, 0, 50).packets(P // 61 + 1)   # 61 SETs fill a frame
    eg, es, _, _ = O.parse_packets(pay, lens, 8, 8)
    assert len(eg) == 0 and len(sk) >= P - 1
    _, s, _, c = device_parse(pay, lens, 8, 8)
    assert np.array_equal(s, es)
    # the host would now put the item locations in; here loc = key id stands in for them
    es[:, 2] = sk
    es = es[sk < P]
    blocks = W.bin_inserts(es, 20, 3, W.BIN_APP)
    O.insert(og, table, blocks)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()
    idx.insert([dev(b) for b in blocks])
    assert np.array_equal(idx.table_numpy(), table)
    # a 95/5 stream of packets against it
    pay, lens, gk, sk = W.PacketSynth(P - 2, 0.99, 21, P, 95, 5).packets(400)
    gq, sq, _, _ = device_parse(pay, lens, 8, 8)
    out = idx.search(dev(gq)).cpu().numpy().view(np.uint32)
    assert np.array_equal(out, O.search(og, table, gq))
    assert (out[:, 0] == gk).all()                   # loc = key id: every GET key was pre-loaded
