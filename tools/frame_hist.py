#!/usr/bin/env python3
"""Histogram of frontier sizes (tokens per frame) of a bench workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else "hlg-lattice"
wl = bench.WORKLOADS[kind]
g = synth.hlg_graph(**wl["graph"]) if wl["kind"] == "hlg" else synth.hl_graph(**wl["graph"])
B, T, V = 32, wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
G = cabi.Graph(g); dev = torch.from_numpy(np.stack(lps)).cuda()
cls = cabi.LatticeBatch if wl["lattice"] else cabi.Batch
bt = cls(G, max_batch=B, max_frames=T, **wl["opts"])
bt.init_decoding(); bt.advance_device([dev[i].data_ptr() for i in range(B)], [T]*B, V)
import ctypes as C
cnt = np.zeros((B, T+1), np.int64)
c = C.c_int32(0)
for u in range(B):
    for f in range(T+1):
        cabi.lib().kd_batch_frame_tokens(bt.h, u, f, 0, C.byref(c), None, None); cnt[u, f] = c.value
flat = cnt.ravel()
print(kind, "frames", flat.size, "mean", flat.mean(), "pctl 50/90/95/99/99.9/max", np.percentile(flat, [50, 90, 95, 99, 99.9, 100]))
for lo, hi in [(0,1024),(1024,2048),(2048,3072),(3072,4096),(4096,6144),(6144,8192),(8192,12288),(12288,16384),(16384,32768),(32768,1<<30)]:
    m = (flat >= lo) & (flat < hi)
    print("  [%6d,%6d): %5d frames  %5.2f%%  tokens share %5.1f%%" % (lo, hi, m.sum(), 100*m.mean(), 100*flat[m].sum()/flat.sum()))
big = cnt > 6144
print("frames >6144 per utt:", big.sum(1).tolist())
print("positions of big frames (utt 0..3):", [np.nonzero(big[u])[0].tolist()[:40] for u in range(4)])
