#!/usr/bin/env python3
"""Candidates queued per frame against forward links kept (bench workload, one batch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"]); B, T, V = wl["batch"], wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
G = cabi.Graph(g); dev = torch.from_numpy(np.stack(lps)).cuda()
bt = cabi.LatticeBatch(G, max_batch=B, max_frames=T, **wl["opts"])
bt.init_decoding(); bt.advance_device([dev[i].data_ptr() for i in range(B)], [T] * B, V)
st = bt.stats().sum(0)
print(os.environ.get("KD_HIP_LIB", "libkd_hip.so"), "per frame: emitting candidates queued %.0f, links %.0f, tokens out %.0f" % (
    st[4] / st[0], st[10] / st[0], st[5] / st[0]))
