#!/usr/bin/env python3
"""Work counters per frame of one call of the bench workload (or hlg-large): arcs scanned,
skipped by the label index, index probes, links -- and the launch's duration.
    [KD_HIP_LIB=...] python tools/stats_probe.py [workload] [utterances]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-decoder_amd", "python")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kaldi_decoder import cabi, synth  # noqa: E402
import bench  # noqa: E402

kind = next((a for a in sys.argv[1:] if not a.isdigit()), "hlg-lattice")
wl = bench.WORKLOADS[kind]
B = next((int(a) for a in sys.argv[1:] if a.isdigit()), wl["batch"])
g = synth.hlg_graph(**wl["graph"]) if wl["kind"] == "hlg" else synth.hl_graph(**wl["graph"])
T, V = wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
G = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
bt = (cabi.LatticeBatch if wl["lattice"] else cabi.Batch)(G, max_batch=B, max_frames=T, **wl["opts"])
ms = []
for _ in range(3):
    bt.decode_device([dev[i].data_ptr() for i in range(B)], [T] * B, V)
    ms.append(bt.last_kernel_ms())
st = bt.stats().sum(axis=0).astype(np.float64)
d = dict(zip(cabi.STAT_NAMES, st))
f = d["frames"]
print(kind, "B", B, "kernel ms", ["%.2f" % m for m in ms])
print("  per frame: tokens_in %.0f expanded %.0f arcs_scanned %.0f skipped %.0f (read %.0f) probes %.0f "
      "eps_arcs %.0f kept %.0f links %.0f tokens_out %.0f small_frames %.3f spill %.4f" % (
          d["tokens_in"] / f, d["tokens_expanded"] / f, d["arcs_scanned"] / f, d["arcs_skipped"] / f,
          (d["arcs_scanned"] - d["arcs_skipped"]) / f, d["index_probes"] / f, d["eps_arcs_scanned"] / f,
          d["arcs_kept"] / f, d["links"] / f, d["tokens_out"] / f, d["small_frames"] / f, d["spill_frames"] / f))
