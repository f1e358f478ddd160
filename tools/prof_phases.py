#!/usr/bin/env python3
"""Per-phase cycle breakdown of the search kernel (diagnostic build only).

    make -C kaldi-decoder_amd prof
    KD_HIP_LIB=libkd_hip_prof.so python tools/prof_phases.py [hlg-lattice|hlg-1best|hl] [batch]

Never quote this build's run time: the stamps forbid overlaps the real kernel
has.  Read the shares.  (KD_HIP_LIB=libkd_hip_prof_tail.so breaks the lattice sweeps' time
down into buckets 16-18 -- time that buckets 21 and 23 hold too: its sum counts it twice.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kaldi_decoder import cabi, synth  # noqa: E402
import bench  # noqa: E402

NAMES = ['0 frame start/ll row', '1 GetCutoff', '2 prepass', '3 emit filter: items+small',
         '4 emit filter: big loop', '5 emit insert', '6 eps round setup', '7 eps filter: items+small',
         '8 eps filter: big loop', '9 eps insert', '10 resolve R1 + alloc', '11 C1 write tokens',
         '12 stats fold + barrier', '13 clear/loop', '14 spill-mode frames', '15 aborted LDS attempts',
         '16 R2 back-pointers', '17 link records', '18 best token + frame index',
         '19 small (LDS-resident) frames', '20 tail: final costs', '21 tail: FinalizeDecoding',
         '22 tail: GetRawLattice', '23 periodic sweep + collector']


def run(kind, B=None):
    wl = bench.WORKLOADS[kind]
    g = synth.hlg_graph(**wl["graph"]) if wl["kind"] == "hlg" else synth.hl_graph(**wl["graph"])
    B = B or wl["batch"]
    T, V = wl["frames"], wl["V"]
    lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
    G = cabi.Graph(g)
    dev = torch.from_numpy(np.stack(lps)).cuda()
    cls = cabi.LatticeBatch if wl["lattice"] else cabi.Batch
    opts = dict(wl["opts"])
    if os.environ.get("KD_PRUNE_INTERVAL"):
        opts["prune_interval"] = int(os.environ["KD_PRUNE_INTERVAL"])
    if os.environ.get("KD_MIN_ACTIVE"):  # (200: LatticeFasterDecoderConfig's default)
        opts["min_active"] = int(os.environ["KD_MIN_ACTIVE"])
    bt = cls(G, max_batch=B, max_frames=T, **opts)
    for _ in range(2):
        bt.decode_device([dev[i].data_ptr() for i in range(B)], [T] * B, V)
    cyc, st = bt.phase_cycles(), bt.stats()
    tot = cyc.sum(0)
    allc = tot.sum()
    print(kind, 'B', B, 'kernel_ms %.2f' % bt.last_kernel_ms(), 'spill_frames', st[:, 9].sum(),
          'small_frames', st[:, 11].sum(), 'of', st[:, 0].sum(),
          'cycles/utt-frame %.0f' % (allc / (B * T)))
    if st[:, 11].sum():
        print('  cycles per small frame (bucket 19, or all but 14 with the _t1 build): %.0f | %.0f'
              % (tot[19] / st[:, 11].sum(), (allc - tot[14]) / st[:, 11].sum()))
    for n, v in zip(NAMES, tot):
        print('  %-28s %5.1f%%  %8.0f cyc/utt-frame' % (n, 100 * v / allc, v / (B * T)))
    tcyc = cyc.sum(1)
    print('  per-utt Mcyc: min %.1f med %.1f max %.1f' % (tcyc.min() / 1e6, np.median(tcyc) / 1e6,
                                                          tcyc.max() / 1e6))


if __name__ == "__main__":
    kinds = [a for a in sys.argv[1:] if not a.isdigit()] or ["hlg-lattice"]
    nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
    for k in kinds:
        run(k, nums[0] if nums else None)
