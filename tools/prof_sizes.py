#!/usr/bin/env python3
"""Where a launch's frames spend their cycles, by the SIZE of the frontier they start from
(diagnostic build: make -C kaldi-decoder_amd knob NAME=prof_size KNOBS="-DKD_PROFILE -DKD_PROF_SIZE").

    KD_HIP_LIB=libkd_hip_prof_size.so python tools/prof_sizes.py [workload] [batch]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
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
opts = dict(wl["opts"])
if os.environ.get("KD_PRUNE_INTERVAL"):
    opts["prune_interval"] = int(os.environ["KD_PRUNE_INTERVAL"])
if os.environ.get("KD_MIN_ACTIVE"):  # (200: LatticeFasterDecoderConfig's default)
    opts["min_active"] = int(os.environ["KD_MIN_ACTIVE"])
bt = (cabi.LatticeBatch if wl["lattice"] else cabi.Batch)(G, max_batch=B, max_frames=T, **opts)
for _ in range(2):
    bt.decode_device([dev[i].data_ptr() for i in range(B)], [T] * B, V)
cyc = bt.phase_cycles().sum(0)
sums, cnts = cyc[:12].astype(np.float64), cyc[12:24].astype(np.float64)
names = ["0", "1", "2-3", "4-7", "8-15", "16-31", "32-63", "64-127", "128-255", "256-511", "512-1023", ">=1024"]
print(kind, "B", B, "kernel_ms %.2f" % bt.last_kernel_ms(), "frames", int(cnts.sum()),
      "cycles/utt-frame (frames only) %.0f" % (sums.sum() / max(cnts.sum(), 1)))
for n, s_, c_ in zip(names, sums, cnts):
    if c_:
        print("  frontier %-9s %6.2f %% of frames  %9.0f cycles each  %5.1f %% of the frames' cycles"
              % (n, 100 * c_ / cnts.sum(), s_ / c_, 100 * s_ / sums.sum()))
