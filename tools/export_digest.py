#!/usr/bin/env python3
"""Digest of everything a fixed batch exports (lattice arrays in their exported ORDER, 1-best
paths, FinalRelativeCost) -- printed as one line, so that two builds of the library
(KD_HIP_LIB=...) can be compared byte for byte: tests/test_gpu_fuzz.py runs it against the
shipped library and against the timing-perturbation build."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-decoder_amd", "python")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kaldi_decoder import cabi, synth  # noqa: E402


def main():
    h = hashlib.sha256()
    g = synth.hlg_graph(V=60, n_words=400, n_hist=400, avg_succ=10, seed=7)
    G = cabi.Graph(g)
    B, T, V = 12, 120, 60
    lps = [synth.make_logprobs(T, V, 20 + i, peak=3.0 if i % 3 else 6.0, lexicon=g["lexicon"]) for i in range(B)]
    dev = torch.from_numpy(np.stack(lps)).cuda()
    n_arcs = 0
    for pi in (7, 25):
        lb = cabi.LatticeBatch(G, max_batch=B, beam=12.0, lattice_beam=7.0, min_active=0, prune_interval=pi)
        lb.decode_tensor(dev)
        lats = lb.raw_lattices()
        for i in range(B):
            lat = lats[i]
            for f in ("src", "dst", "ilabel", "olabel", "graph_cost", "ac_cost", "state_frame", "state_graph",
                      "final_state", "final_graph"):
                h.update(np.ascontiguousarray(getattr(lat, f)).tobytes())
            n_arcs += len(lat.src)
        h.update(lb.final_relative_cost().tobytes())
        # the un-finalized export of a stepwise decode (the general, host-ordered export)
        lb.init_decoding(B)
        lb.advance_device([dev[i].data_ptr() for i in range(B)], [T] * B, V, max_num_frames=50)
        for lat in lb.raw_lattices():
            h.update(np.ascontiguousarray(lat.src).tobytes() + np.ascontiguousarray(lat.dst).tobytes())
    bt = cabi.Batch(G, max_batch=B, beam=12.0, max_active=300)
    bt.decode_tensor(dev)
    for p in bt.best_paths():
        h.update(p.ilabel.tobytes() + p.state.tobytes() + p.ac_cost.tobytes())
    print("digest", h.hexdigest(), "arcs", n_arcs)


if __name__ == "__main__":
    main()
