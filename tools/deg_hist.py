#!/usr/bin/env python3
"""Out-degree distribution of the tokens the bench workload expands (small frames)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"])
B, T, V = 16, wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
G = cabi.Graph(g); dev = torch.from_numpy(np.stack(lps)).cuda()
bt = cabi.LatticeBatch(G, max_batch=B, max_frames=T, **wl["opts"])
bt.init_decoding(); bt.advance_device([dev[i].data_ptr() for i in range(B)], [T] * B, V)
rp = g["row_ptr"]; il = g["ilabel"]
n_emit = np.add.reduceat((il != 0).astype(np.int64), rp[:-1]) if False else None
deg_all = np.diff(rp)
# emitting arcs per state
emit = np.zeros(g["num_states"], np.int64)
np.add.at(emit, np.repeat(np.arange(g["num_states"]), deg_all), (il != 0).astype(np.int64))
eps = deg_all - emit
edges = [0, 1, 5, 9, 17, 33, 65, 257, 1 << 30]
cnt = np.zeros(len(edges) - 1); arcs = np.zeros(len(edges) - 1); ecnt = np.zeros(len(edges) - 1); earcs = np.zeros(len(edges) - 1)
nframes = 0; ntok = 0
for u in range(B):
    for f in range(T):
        st, _ = bt.frame_tokens(u, f)
        if len(st) > 1024: continue
        nframes += 1; ntok += len(st)
        d = emit[st]; e = eps[st]
        for k in range(len(edges) - 1):
            m = (d >= edges[k]) & (d < edges[k + 1]); cnt[k] += m.sum(); arcs[k] += d[m].sum()
            m = (e >= edges[k]) & (e < edges[k + 1]); ecnt[k] += m.sum(); earcs[k] += e[m].sum()
print("small frames %d, tokens/frame %.1f, emitting arcs/frame %.1f, eps arcs/frame %.1f" % (nframes, ntok / nframes, arcs.sum() / nframes, earcs.sum() / nframes))
for k in range(len(edges) - 1):
    print("  deg [%4d,%6d): emit tokens/frame %7.2f arcs/frame %8.1f | eps tokens/frame %7.2f arcs/frame %8.1f" % (
        edges[k], min(edges[k + 1], 999999), cnt[k] / nframes, arcs[k] / nframes, ecnt[k] / nframes, earcs[k] / nframes))
