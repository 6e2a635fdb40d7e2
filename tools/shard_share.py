#!/usr/bin/env python3
"""Largest share of one shard (x the average) under the benchmark GET streams, per world size."""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import workload as W
res={}
for G in (8,4,2):
    k=G.bit_length()-1
    mem_p_agg=30+k
    P=W.preload_count(mem_p_agg, 0.2)
    mx=0; mxs=0
    for rank in range(G):
        z=W.ZipfKeys(P-2, 0.99, 21+rank)
        for b in range(130):
            keys=z.keys(996148)
            e=W.make_selem(keys, W.KEYING_MIX)
            sh=((e[:,1] & ((1<<(mem_p_agg-6))-1)) >> (30-6))
            c=np.bincount(sh, minlength=G)/ (len(keys)/G)
            mx=max(mx,c.max())
        print(G, rank, mx, flush=True)
    res[G]=mx
print(res)
