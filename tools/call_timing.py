#!/usr/bin/env python3
"""Host-side time of one library call, by part (one call in flight): enqueue, wait,
lattice read-back.  python tools/call_timing.py [steps_per_call]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kaldi_decoder import cabi, synth  # noqa: E402
import bench  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 7
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"])
B, T, V = wl["batch"] * G, wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
Gr = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i].data_ptr() for i in range(B)]
bt = cabi.LatticeBatch(Gr, max_batch=B, max_frames=T, **wl["opts"])
rows = [T] * B
for it in range(5):
    t = [time.perf_counter()]
    bt.decode_device(ptrs, rows, V, sync=False); t.append(time.perf_counter())
    bt.sync(); t.append(time.perf_counter())
    out = bt.raw_lattices(); t.append(time.perf_counter())
    k = bt.last_kernel_ms()
    print("enqueue %.2f ms  wait %.2f ms (kernel %.2f)  lattices %.2f ms  | capacity %s" % (
        (t[1] - t[0]) * 1e3, (t[2] - t[1]) * 1e3, k, (t[3] - t[2]) * 1e3, bt.capacity()))
