#!/usr/bin/env python3
"""Where the wall time of ONE batch of 128 goes beside its kernel: the enqueue (Python + library),
the wait, the read-back of the lattices.   python tools/single_batch_probe.py [max_batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"])
T, V, N = wl["frames"], wl["V"], 128
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(N)]
G = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i].data_ptr() for i in range(N)]
st = torch.cuda.Stream()
bt = cabi.LatticeBatch(G, max_batch=MB, max_frames=T, stream=st.cuda_stream, **wl["opts"])
rows = []
for it in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bt.decode_device(ptrs, [T] * N, V, sync=False)
    t1 = time.perf_counter()
    bt.sync()
    t2 = time.perf_counter()
    k = bt.last_kernel_ms()
    t3 = time.perf_counter()
    out = bt.raw_lattices()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    rows.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, k, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t0) * 1e3])
r = np.median(np.array(rows[2:]), axis=0)
print("max_batch %d: enqueue %.3f ms, wait %.3f (kernel %.3f by events), elapsed-time call %.3f, raw_lattices %.3f, total %.3f; arcs %d" % (
    MB, r[0], r[1], r[2], r[3], r[4], r[5], int(out.arcs_per_utt.sum())))
