#!/usr/bin/env python3
"""Do two CUs that share an instruction cache (and a scalar cache) slow each other down?
One batch of 128 utterances of the bench workload -- 128 workgroups, one per CU -- on CU-masked
streams (hipExtStreamCreateWithCUMask; mask bit i is CU i / 8 of XCD i % 8, tools/ubench/cumask.hip):
   all      the whole chip (the dispatcher's own placement)
   even     CUs 0, 2, 4, ... of every XCD: every workgroup has its CU pair's caches to itself
   pairs    CUs 0..15 of every XCD: every pair of CUs holds two workgroups
   python tools/cu_pair_probe.py [workload]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hlg-lattice"]
g = synth.hlg_graph(**wl["graph"])
T, V, N = wl["frames"], wl["V"], 128
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(N)]
G = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i].data_ptr() for i in range(N)]
hip = C.CDLL("libamdhip64.so")
cus = torch.cuda.get_device_properties(0).multi_processor_count


def masked(pred):
    words = (cus + 31) // 32
    m = (C.c_uint32 * words)()
    n = 0
    for i in range(cus):
        if pred(i % 8, i // 8):
            m[i // 32] |= 1 << (i % 32)
            n += 1
    st = C.c_void_p()
    r = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), m)
    assert r == 0, r
    return st.value, n


for name, pred in (("all", None), ("even", lambda x, c: c % 2 == 0), ("odd", lambda x, c: c % 2 == 1),
                   ("pairs", lambda x, c: c < 16), ("pairs_hi", lambda x, c: c >= 16),
                   ("quads", lambda x, c: c % 4 < 2)):
    if pred is None:
        stream, n = torch.cuda.Stream().cuda_stream, cus
    else:
        stream, n = masked(pred)
    bt = cabi.LatticeBatch(G, max_batch=N, max_frames=T, stream=stream, **wl["opts"])
    ms = []
    for it in range(6):
        bt.decode_device(ptrs, [T] * N, V)
        ms.append(bt.last_kernel_ms())
    print("%-9s %3d CUs  kernel ms %s   median of the last five %.2f" % (
        name, n, " ".join("%.2f" % m for m in ms), float(np.median(ms[1:]))), flush=True)
    del bt
