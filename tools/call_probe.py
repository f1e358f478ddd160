#!/usr/bin/env python3
"""Probe: one LatticeBatch call of N utterances (distinct or repeated) with chosen pool sizes.
   python tools/call_probe.py N [arena_shift] [no_grow] [repeat_of]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
N = int(sys.argv[1]); shift = int(sys.argv[2]) if len(sys.argv) > 2 else 17
no_grow = int(sys.argv[3]) if len(sys.argv) > 3 else 0
distinct = int(sys.argv[4]) if len(sys.argv) > 4 else min(N, 256)
wl = bench.WORKLOADS[os.environ.get("KD_WL", "hlg-lattice")]
g = synth.hlg_graph(**wl["graph"])
T, V = wl["frames"], wl["V"]
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(distinct)]
G = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i % distinct].data_ptr() for i in range(N)]
opts = dict(wl["opts"])
bt = cabi.LatticeBatch(G, max_batch=N, max_frames=T, arena_tokens=N << shift, arena_links=N << (shift + 1),
                       no_grow=no_grow, **opts)
for it in range(3):
    t0 = time.time()
    try:
        bt.decode_device(ptrs, [T] * N, V)
    except RuntimeError as e:
        print("decode failed:", e)
        import ctypes as C
        out = np.zeros((N, 24), np.int64)
        rc = cabi.lib().kd_batch_phase_cycles(bt.h, out)
        print("phase_cycles rc", rc)
        bad = np.nonzero((out[:, 0] == 0x7777) | (out[:, 0] == 0x8888))[0]
        for u in bad[:12]:
            print("utt", u, [hex(int(x)) if k == 0 else int(x) for k, x in enumerate(out[u, :13])])
        print("status", bt.status()[bad[:12]] if len(bad) else None, "decoded", bt.num_frames_decoded()[:8])
        sys.exit(1)
    lats = bt.raw_lattices()
    print("iter", it, "ok; kernel_ms %.2f wall %.3f" % (bt.last_kernel_ms(), time.time() - t0), bt.capacity(), flush=True)
a = [len(lats[i].src) for i in range(min(N, distinct))]
b = [len(lats[i + distinct].src) for i in range(min(distinct, N - distinct))] if N > distinct else []
print("arcs first copies", a[:8], "second copies", b[:8], "equal" if a[:len(b)] == b else "DIFFER")
