#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: LatticeFasterDecoder over a large-LM HLG-like graph
(> 2 GB of 16-byte arcs), beam 17, B = 128, T = 250.  One-off measurement, not a bench line.

    python tools/cfg5_probe.py [n_words] [streams]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from kaldi_decoder import cabi, synth

n_words = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, T, V = 128, 250, 500
t0 = time.time()
g = synth.hlg_graph(V=V, n_words=n_words, n_hist=n_words, avg_succ=50, seed=2)
info = synth.describe(g)
print("graph: %d states, %d arcs (%.2f GB of arc records) built in %.0f s" % (
    info["states"], info["arcs"], info["arcs"] * 16 / 1e9, time.time() - t0), flush=True)
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
G = cabi.Graph(g)
dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i].data_ptr() for i in range(B)]
streams = [torch.cuda.Stream() for _ in range(n_streams)]
bts = [cabi.LatticeBatch(G, max_batch=B, max_frames=T, stream=st.cuda_stream, beam=17.0,
                         lattice_beam=8.0, min_active=0, arena_tokens=(1 << 31) - 1, arena_links=(1 << 31) - 1, table_slots=1 << 19)
       for st in streams]

def step(bt):
    bt.init_decoding(); bt.advance_device(ptrs, [T] * B, V, sync=False); bt.sync()
    k = bt.last_kernel_ms(); bt.finalize_decoding(); return bt.raw_lattices(), k

for bt in bts: step(bt)
torch.cuda.synchronize()
pool = ThreadPoolExecutor(n_streams)
n = 4 * n_streams
t1 = time.perf_counter()
res = list(pool.map(lambda i: step(bts[i % n_streams]), range(n)))
torch.cuda.synchronize()
el = time.perf_counter() - t1
st = bts[0].stats().sum(0)
print("steps %d  ms/step %.2f  audio frames/s %.2fM  kernel_ms %.1f" % (
    n, el / n * 1e3, n * B * T * 4 / el / 1e6, np.mean([k for _, k in res])))
print("per frame: tokens in %.0f, arcs scanned %.0f, links %.0f; max tokens %d; spill frames %d of %d" % (
    st[1] / st[0], (st[3] + st[6]) / st[0], st[10] / st[0], bts[0].stats()[:, 8].max(), st[9], st[0]))
lat = res[-1][0]
print("lattice arcs per utterance %.0f" % np.mean([len(l.src) for l in lat]))
