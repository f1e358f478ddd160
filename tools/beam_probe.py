#!/usr/bin/env python3
"""Kernel time of one call of N utterances of the bench workload at another beam (diagnostic:
how a build behaves when every frame is small).  [KD_HIP_LIB=...] python tools/beam_probe.py beam [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-decoder_amd", "python")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kaldi_decoder import cabi, synth  # noqa: E402
import bench  # noqa: E402

beams = [float(a) for a in sys.argv[1].split(",")]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
wl = bench.WORKLOADS["hlg-lattice"]
g = synth.hlg_graph(**wl["graph"])
T, V = wl["frames"], wl["V"]
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    lps = list(ex.map(lambda u: synth.make_logprobs(T, V, u, peak=12.0, lexicon=g["lexicon"]), range(N)))
G = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
for beam in beams:
    bt = cabi.LatticeBatch(G, max_batch=N, max_frames=T, **{**wl["opts"], "beam": beam})
    ms = []
    for _ in range(4):
        bt.decode_tensor(dev)
        ms.append(bt.last_kernel_ms())
    st = bt.stats().sum(axis=0).astype(float)
    print("beam %.1f N %d kernel ms %s tokens/frame %.0f small_frames %.3f  -> %.2f M audio frames/s (one call alone)" % (
        beam, N, ["%.1f" % m for m in ms], st[1] / st[0], st[11] / st[0], N * T * 4 / (min(ms[1:]) * 1e-3) / 1e6))
    del bt
