#!/usr/bin/env python3
"""Fixed cost of a frame: tiny frontiers (CTC topology, and a small HLG with
epsilon arcs), one utterance per CU.  Prints microseconds per frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np, torch
from kaldi_decoder import cabi, synth

def run(name, g, lex, T, B, lattice, **opts):
    V = int(g["ilabel"].max())
    lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=lex) for i in range(B)]
    G = cabi.Graph(g); dev = torch.from_numpy(np.stack(lps)).cuda()
    cls = cabi.LatticeBatch if lattice else cabi.Batch
    bt = cls(G, max_batch=B, max_frames=T, **opts)
    for _ in range(2):
        bt.init_decoding(); bt.advance_device([dev[i].data_ptr() for i in range(B)], [T] * B, V)
    st = bt.stats()
    print("%-28s B %3d: %.2f us/frame  tokens/frame %.1f arcs/frame %.1f eps_exp/frame %.1f small %d/%d" % (
        name + (" lattice" if lattice else ""), B, bt.last_kernel_ms() * 1e3 / T, st[:, 5].sum() / st[:, 0].sum(),
        st[:, 3].sum() / st[:, 0].sum(), st[:, 7].sum() / st[:, 0].sum(), st[:, 11].sum(), st[:, 0].sum()))
    if "prof" in os.environ.get("KD_HIP_LIB", ""):
        cyc = bt.phase_cycles().sum(0) / (B * T)
        print("   cycles/frame by bucket:", " ".join("%d:%d" % (i, v) for i, v in enumerate(cyc) if v >= 1), "total", int(cyc.sum()))

T = 1000
for lattice in (False, True):
    kw = dict(lattice_beam=8.0) if lattice else {}
    run("H8 beam 2", synth.ctc_topo(8), None, T, 1, lattice, beam=2.0, **kw)
    run("H500 beam 8", synth.ctc_topo(500), None, T, 1, lattice, beam=8.0, **kw)
    g = synth.hlg_graph(V=40, n_words=200, n_hist=200, avg_succ=8, seed=4)
    run("HLG small beam 6", g, g["lexicon"], T, 1, lattice, beam=6.0, **kw)
    run("HLG small beam 6", g, g["lexicon"], T, 128, lattice, beam=6.0, **kw)
