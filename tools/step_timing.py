#!/usr/bin/env python3
"""Host-side timing of the pieces of one bench step, null stream vs a torch stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"]); B, T, V = wl["batch"], wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
G = cabi.Graph(g); dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i].data_ptr() for i in range(B)]
for name, st in (("null", None), ("torch", torch.cuda.Stream())):
    bt = cabi.LatticeBatch(G, max_batch=B, max_frames=T, stream=(st.cuda_stream if st else None), **wl["opts"])
    acc = np.zeros(5)
    for it in range(4):
        t = [time.perf_counter()]
        bt.init_decoding(); t.append(time.perf_counter())
        bt.advance_device(ptrs, [T]*B, V, sync=False); t.append(time.perf_counter())
        bt.sync(); t.append(time.perf_counter())
        bt.finalize_decoding(); t.append(time.perf_counter())
        out = bt.raw_lattices(); t.append(time.perf_counter())
        if it: acc += np.diff(t)
    print(name, "ms: init %.2f launch %.2f sync %.2f finalize %.2f export %.2f total %.2f" % (*(acc/3*1e3), acc.sum()/3*1e3))
