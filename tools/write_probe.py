#!/usr/bin/env python3
"""One call shape under rocprofv3 --pmc (tools/write_breakdown.sh): calls of G x 128 utterances
of a bench workload, one in flight.  python tools/write_probe.py workload prune_interval [G]
(prune_interval 0: the workload's; hlg-1best ignores it)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kaldi_decoder import cabi, synth  # noqa: E402
import bench  # noqa: E402

name, pi = sys.argv[1], int(sys.argv[2])
G = int(sys.argv[3]) if len(sys.argv) > 3 else 4
wl = bench.WORKLOADS[name]
g = synth.hlg_graph(**wl["graph"])
B, T, V = wl["batch"] * G, wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
Gr = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i].data_ptr() for i in range(B)]
opts = dict(wl["opts"])
if wl["lattice"] and pi:
    opts["prune_interval"] = pi
if len(sys.argv) > 4:
    opts["beam"] = float(sys.argv[4])
bt = (cabi.LatticeBatch if wl["lattice"] else cabi.Batch)(Gr, max_batch=B, max_frames=T, **opts)
for it in range(4):
    bt.decode_device(ptrs, [T] * B, V, sync=False)
    bt.sync()
    print("kernel %.2f ms" % bt.last_kernel_ms())
st = bt.stats().sum(0)
print("per utterance-frame: arcs %.0f tokens out %.0f links %.0f queued %.0f  global-table frames %.4f" % (
    (st[3] + st[6]) / st[0], st[5] / st[0], st[10] / st[0], st[4] / st[0], st[9] / st[0]))
