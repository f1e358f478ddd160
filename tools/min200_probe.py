#!/usr/bin/env python3
"""Probe: the bench workload at min_active = 200, PruneActiveTokens owed or run: exported words,
arena-order (general) exports, equality of the lattices."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"])
T, V = wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(N)]
G = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
out = {}
for lazy in (0, 1):
    bt = cabi.LatticeBatch(G, max_batch=N, max_frames=T, **{**wl["opts"], "min_active": 200})
    bt.set_lazy_prune(lazy)
    bt.decode_tensor(dev)
    lats = bt.raw_lattices()
    st = dict(zip(cabi.STAT_NAMES, bt.stats().sum(axis=0).tolist()))
    arcs = [len(lats[i].src) for i in range(N)]
    states = [lats[i].num_states for i in range(N)]
    print("lazy", lazy, "arcs/utt %.0f states/utt %.0f max arcs %d prunes %d links %d kernel_ms %.1f cap %s" % (
        np.mean(arcs), np.mean(states), max(arcs), st["prunes"], st["links"], bt.last_kernel_ms(), bt.capacity()), flush=True)
    out[lazy] = [(lats[i].arc_set(), lats[i].state_set()) for i in range(0, N, 8)]
print("equal:", out[0] == out[1])
