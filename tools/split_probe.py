#!/usr/bin/env python3
"""Probe of the two-kernel launch: calls of N utterances of a bench workload, one in flight,
under several settings of the split -- time per call, hand-overs per utterance, share of the
frames run by the small kind, and where the two kinds' workgroups spent their cycles
(kd_batch_sched_profile).
   python tools/split_probe.py [N] [setting ...]     setting = name:lo:hi:prune_small  or  name:off
   KD_WL=hlg-large python tools/split_probe.py 1024"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
settings = sys.argv[2:] or ["off:off", "d256:256:256:0", "lo128:128:256:0", "lo32:32:256:0", "prune:256:256:1"]
wl = bench.WORKLOADS[os.environ.get("KD_WL", "hlg-lattice")]
g = synth.hlg_graph(**wl["graph"])
T, V = wl["frames"], wl["V"]
distinct = min(N, 512)
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(distinct)]
G = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i % distinct].data_ptr() for i in range(N)]
for st in settings:
    f = st.split(":")
    bt = cabi.LatticeBatch(G, max_batch=N, max_frames=T, **wl["opts"])
    if f[1] == "off":
        bt.set_split(enable=0)
    else:
        bt.set_split(enable=1, min_batch=1, mig_lo=int(f[1]), mig_hi=int(f[2]), prune_small=int(f[3]), stay=int(f[4]) if len(f) > 4 else 0)
    ms = []
    for it in range(4):
        bt.decode_device(ptrs, [T] * N, V)
        ms.append(bt.last_kernel_ms())
        if it == 0:
            p0 = bt.sched_profile()
    p1 = bt.sched_profile()
    sts = dict(zip(cabi.STAT_NAMES, bt.stats().sum(axis=0).tolist()))
    line = "%-8s ms %s  handovers/utt %.1f  small-kind frames %.1f%%" % (
        f[0], " ".join("%.1f" % m for m in ms), sts["handovers"] / N, 100.0 * sts["small_wg_frames"] / max(1, sts["frames"]))
    print(line, bt.split_info(), flush=True)
    if f[1] != "off":
        for kind in ("small", "large"):
            d = {k: (p1[k][kind] - p0[k][kind]) / 3 for k in p1}
            tot = d["idle"] + d["pop"] + d["setup"] + d["busy"]
            n = max(1.0, d["resumed"] + d["stolen"] + d["fresh"])
            print("   %-5s wgs %3d  idle %4.1f%%  pop %4.1f%%  setup %4.1f%%  busy %4.1f%%   per take: pop %5.0f setup %6.0f busy %7.0f cycles;  takes/call fresh %d resumed %d stolen %d" % (
                kind, d["workgroups"], 100 * d["idle"] / tot, 100 * d["pop"] / tot, 100 * d["setup"] / tot, 100 * d["busy"] / tot,
                d["pop"] / n, d["setup"] / n, d["busy"] / n, d["fresh"], d["resumed"], d["stolen"]), flush=True)
    del bt
