#!/usr/bin/env python3
"""Bench-size equality check: GPU lattices of the first N bench utterances against
the oracle's canonical lattice decoder (tests/ do this on small graphs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
from oracle import oracle
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"]); B, T, V = wl["batch"], wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
G = cabi.Graph(g); dev = torch.from_numpy(np.stack(lps)).cuda()
bt = cabi.LatticeBatch(G, max_batch=B, max_frames=T, **wl["opts"])
bt.init_decoding(); bt.advance_device([dev[i].data_ptr() for i in range(B)], [T]*B, V)
st = bt.stats(); bt.finalize_decoding(); lats = bt.raw_lattices()
og = oracle.Graph(g); bad = 0
for i in range(N):
    can = oracle.CanonLatticeDecoder(og, beam=wl["opts"]["beam"], lattice_beam=wl["opts"]["lattice_beam"])
    can.decode(lps[i]); cl = can.get_raw_lattice()
    same = lats[i].arc_set() == cl.arc_set() and len(lats[i].src) == len(cl.src) and lats[i].state_set() == cl.state_set()
    print(i, "links gpu", st[i, 10], "oracle", can.links_created(), "spill", st[i, 9], "arcs", len(lats[i].src), len(cl.src), "equal" if same else "DIFFERENT")
    bad += not same
print("mismatching utterances:", bad)
