"""SURVEY 8(f) row 3, CPU side: the LOCAL_TEST packet synthesis (workload/workload.c,
after src/mega_recv.c:633-768) against the oracle's restatement of the receiver's parser
(src/mega_recv.c:292-545) and against the workload definition (K-ref keying)."""
import numpy as np
import pytest

import workload as W
from oracle import oracle as O


@pytest.mark.parametrize("num_get,num_set,key_len,value_len", [(95, 5, 8, 8), (100, 0, 8, 8), (50, 50, 16, 64),
                                                               (0, 7, 8, 8), (95, 5, 32, 512)])
def test_synthesised_packets_parse_back_to_the_workload(num_get, num_set, key_len, value_len):
    P = W.preload_count(25, 0.2)
    syn = W.PacketSynth(P - 2, 0.99, 21, P, num_get, num_set, key_len, value_len, iblock_p=3)
    pay, lens, gk, sk = syn.packets(300)
    # framing: magic, end mark, frame budget of src/mega_recv.c:639,669,708
    assert (pay[:, 0] == 0x34).all() and (pay[:, 1] == 0x12).all()
    for p in range(len(lens)):
        assert pay[p, lens[p] - 2] == 0xFF and pay[p, lens[p] - 1] == 0xFF
        assert 42 + lens[p] <= 1514
    gets, sets, bad_packets, bad_jobs = O.parse_packets(pay, lens, key_len, value_len)
    assert bad_packets == 0 and bad_jobs == 0
    # the GET ids are the reference's Zipf stream, the SET ids count up from P
    if num_get:
        z = W.ZipfKeys(P - 2, 0.99, 21)
        assert np.array_equal(gk, z.keys(len(gk)))
    assert np.array_equal(sk, np.arange(P, P + len(sk), dtype=np.uint32))
    # GET/SET alternate in runs of num_get / num_set across packet borders
    if num_get and num_set:
        assert abs(len(gk) * num_set - len(sk) * num_get) <= max(num_get, num_set) * max(num_get, num_set)
    # parsed requests == the workload definition's (sig, hash) of those ids
    assert np.array_equal(gets, W.make_selem(gk, W.KEYING_REF, 3))
    assert np.array_equal(sets[:, :2], W.make_ielem(sk, None, W.KEYING_REF, 3)[:, :2])


def test_parser_error_paths():
    """wrong magic drops the packet; a job with another key length is skipped; an unknown job
    type ends the packet but keeps what came before (src/mega_recv.c:293-296, 312-315, 535-537)"""
    def pkt(*parts):
        b = b"".join(parts)
        a = np.zeros(1472, np.uint8)
        a[:len(b)] = np.frombuffer(b, np.uint8)
        return a, len(b)
    u16 = lambda v: int(v).to_bytes(2, "little")
    u32 = lambda v: int(v).to_bytes(4, "little")
    key = lambda k: u32(k) + u32(k | 0x20000000)
    get = lambda k: u16(2) + u16(8) + key(k)
    sett = lambda k: u16(3) + u16(8) + u32(8) + key(k) + bytes(8)
    cases = [pkt(u16(0x1234), get(5), sett(6), get(7), u16(0xFFFF)),
             pkt(u16(0x4321), get(5), u16(0xFFFF)),                                  # wrong magic
             pkt(u16(0x1234), get(5), u16(2) + u16(9) + bytes(9), get(8), u16(0xFFFF)),   # bad key length
             pkt(u16(0x1234), get(5), u16(9), get(6), u16(0xFFFF)),                   # unknown job type
             pkt(u16(0x1234), get(5), get(6))]                                        # no end mark
    pay = np.stack([c[0] for c in cases])
    lens = np.array([c[1] for c in cases], np.int32)
    gets, sets, bad_packets, bad_jobs = O.parse_packets(pay, lens, 8, 8)
    assert [int(x) for x in gets[:, 0]] == [5, 7, 5, 8, 5, 5, 6]
    assert [int(x) for x in sets[:, 0]] == [6]
    assert bad_packets == 3 and bad_jobs == 1
