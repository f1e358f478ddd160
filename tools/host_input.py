#!/usr/bin/env python3
"""PCIe-inclusive rate: the same batch advanced from host log-prob matrices
(kd_batch_advance with KD_MEM_HOST stages them per call) versus from matrices
already resident in HBM.  One stream, search only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"]); G = cabi.Graph(g)
B, T, V = wl["batch"], wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i].data_ptr() for i in range(B)]
bt = cabi.LatticeBatch(G, max_batch=B, max_frames=T, **wl["opts"])
def run(host):
    bt.init_decoding()
    if host: bt.advance_host(lps)
    else: bt.advance_device(ptrs, [T] * B, V)
for host in (False, True):
    run(host); run(host)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run(host)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%s input: %.2f ms per batch (search only) = %.2f M audio frames/s; %d MB of log-probs per batch" % (
        "host" if host else "device", dt * 1e3, B * T * 4 / dt / 1e6, B * T * V * 4 >> 20))
