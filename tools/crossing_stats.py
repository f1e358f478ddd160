#!/usr/bin/env python3
"""Per utterance: the sizes of the frontiers its frames start from (canonical oracle, CPU), the
lengths of the runs of frames below / above a threshold and the number of crossings -- what a
scheduler that runs small frames three to a CU and large ones on a whole CU has to live with
("measure first").

    python tools/crossing_stats.py [workload] [n_utts] [lo] [hi]

`hi`: a frame that starts from more than `hi` tokens must run on a large workgroup; `lo`: a
large workgroup hands an utterance back when a frame starts from at most `lo` (hysteresis).
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np
from concurrent.futures import ProcessPoolExecutor
import bench
from kaldi_decoder import synth
from oracle import oracle

kind = sys.argv[1] if len(sys.argv) > 1 else "hlg-lattice"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
wl = bench.WORKLOADS[kind]
g = synth.hlg_graph(**wl["graph"]) if wl["kind"] == "hlg" else synth.hl_graph(**wl["graph"])
T, V = wl["frames"], wl["V"]
G = None

def sizes(u):
    global G
    if G is None:
        G = oracle.Graph(g)
    lp = synth.make_logprobs(T, V, u, peak=12.0, lexicon=g["lexicon"])
    o = wl["opts"]
    d = oracle.CanonLatticeDecoder(G, beam=o["beam"], lattice_beam=o["lattice_beam"],
                                   min_active=o.get("min_active", 0))
    d.decode(lp)
    return [len(d.frame_tokens(f)[0]) for f in range(T + 1)]

if __name__ == "__main__":
    with ProcessPoolExecutor(8) as ex:
        cnt = np.array(list(ex.map(sizes, range(N))))
    np.save("/tmp/w/sizes_%s.npy" % kind, cnt)
    n = cnt[:, :T]  # frame f starts from the tokens of frame-plus-one index f
    print(kind, "utts", N, "mean", n.mean(), "pctl 50/75/90/99", np.percentile(n, [50, 75, 90, 99]))
    for thr in (64, 128, 192, 256, 384, 512):
        big = n > thr
        cross = (big[:, 1:] != big[:, :-1]).sum(1)
        print("thr %4d: frames above %5.1f%%  crossings per utt %5.1f (of %d frames)  mean run below %5.1f above %5.1f"
              % (thr, 100 * big.mean(), cross.mean(), T,
                 (~big).sum() / max(1, ((~big)[:, 1:] & big[:, :-1]).sum() + (~big)[:, 0].sum()),
                 big.sum() / max(1, (big[:, 1:] & ~big[:, :-1]).sum() + big[:, 0].sum())))
    for lo, hi in ((64, 256), (128, 256), (32, 256), (16, 256), (128, 512), (64, 192), (32, 128), (16, 128), (8,256)):
        # hysteresis: state L (large) entered when n > hi, left when n <= lo
        moves = 0; frames_large = 0; frames_small_n = []
        for u in range(N):
            large = False
            for f in range(T):
                x = n[u, f]
                if not large and x > hi: large = True; moves += 1
                elif large and x <= lo: large = False; moves += 1
                frames_large += large
        print("lo %4d hi %4d: migrations per utt %5.1f, frames on large %5.1f%%" % (lo, hi, moves / N, 100 * frames_large / (N * T)))
