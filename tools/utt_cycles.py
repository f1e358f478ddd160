#!/usr/bin/env python3
"""Per-utterance cycle totals of the diagnostic build, against how many of its
frames were spill frames (run with KD_HIP_LIB=libkd_hip_prof.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else "hlg-lattice"
wl = bench.WORKLOADS[kind]
g = synth.hlg_graph(**wl["graph"]) if wl["kind"] == "hlg" else synth.hl_graph(**wl["graph"])
B, T, V = wl["batch"], wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
G = cabi.Graph(g); dev = torch.from_numpy(np.stack(lps)).cuda()
cls = cabi.LatticeBatch if wl["lattice"] else cabi.Batch
bt = cls(G, max_batch=B, max_frames=T, **wl["opts"])
for _ in range(2):
    bt.init_decoding(); bt.advance_device([dev[i].data_ptr() for i in range(B)], [T]*B, V)
cyc, st = bt.phase_cycles(), bt.stats()
tot = cyc.sum(1)
order = np.argsort(-tot)
print("utt  Mcyc  spillMcyc spill_frames tokens_out arcs_scanned arcs_kept")
for u in list(order[:12]) + list(order[-4:]):
    print("%3d %6.1f %6.1f %5d %9d %10d %9d" % (u, tot[u]/1e6, cyc[u,14]/1e6, st[u,9], st[u,5], st[u,3], st[u,4]))
print("corr(total, spill frames) %.3f" % np.corrcoef(tot, st[:,9])[0,1])
nz = st[:,9] > 0
print("cycles per spill frame: %.0f" % (cyc[nz,14].sum() / st[nz,9].sum()))
print("non-spill Mcyc per utt: min %.1f med %.1f max %.1f" % tuple(np.percentile((tot-cyc[:,14])/1e6,[0,50,100])))
