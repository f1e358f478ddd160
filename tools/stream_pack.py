#!/usr/bin/env python3
"""How well do search kernels of independent batches pack onto the GPU?
Launches S batches' searches back to back on S streams from one thread (no
finalize / export), and reports wall time per batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"]); G = cabi.Graph(g)
B, T, V = int(sys.argv[1]) if len(sys.argv) > 1 else 128, wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i].data_ptr() for i in range(B)]
for S in (1, 2, 3, 4, 6):
    streams = [torch.cuda.Stream() for _ in range(S)]
    bts = [cabi.LatticeBatch(G, max_batch=B, max_frames=T, stream=st.cuda_stream, **wl["opts"]) for st in streams]
    def round_():
        for bt in bts:
            bt.init_decoding(); bt.advance_device(ptrs, [T] * B, V, sync=False)
        for bt in bts:
            bt.sync()
    round_(); round_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 4
    for _ in range(R): round_()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (R * S)
    print("B %d streams %d: %.2f ms per batch, kernel %.2f ms  (%.2f M audio frames/s)" % (B, S, dt * 1e3, bts[0].last_kernel_ms(), B * T * 4 / dt / 1e6))
    del bts
