#!/usr/bin/env python3
"""Randomised parity runs of the GPU decoders against the oracle's canonical decoders:
random graphs, options (beams, max/min_active, prune_interval), batch sizes, pool sizes
(small ones grow), one-launch Decode or chunked AdvanceDecoding, and -- round 6 -- the call dealt
out over the two kinds of workgroups or run by one kernel, with random hand-over thresholds
(frontiers of a handful of tokens change hands every few frames) and pruning on either kind.

    python tools/fuzz_gpu.py [seconds] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-decoder_amd", "python")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kaldi_decoder import cabi, synth  # noqa: E402
from oracle import oracle  # noqa: E402

INT_MAX = 2**31 - 1


def graph(rng):
    if rng.random() < float(os.environ.get("KD_FUZZ_HEAVY", "0.04")):  # (share of heavy cases)
        # heavy: tens of thousands of states under flat posteriors -- frontiers far beyond
        # the LDS tiers (the global-table tier, table growth, oversized pool blocks)
        V = int(rng.integers(8, 24))
        return synth.random_graph(int(rng.integers(8000, 50000)), float(rng.uniform(3, 5)), V,
                                  float(rng.uniform(0, 0.2)), int(rng.integers(0, 10**6))), None, V
    k = rng.integers(0, 5)
    if k == 0:
        V = int(rng.integers(4, 40))
        return synth.ctc_topo(V), None, V
    if k == 1:
        V = int(rng.integers(8, 24))
        # a third of them with epsilon cycles of positive weight (kd_graph_create refuses the
        # others: csrc/kd_graph_check.h): the closure's re-queueing and the lattice sweep's
        # fixpoint on cyclic links
        cyc = rng.random() < 0.33
        return synth.random_graph(int(rng.integers(50, 3000)), float(rng.uniform(2, 5)), V,
                                  float(rng.uniform(0.2, 0.5) if cyc else rng.uniform(0, 0.35)),
                                  int(rng.integers(0, 10**6)), eps_back=0.5 if cyc else 0.0,
                                  eps_back_min_weight=float(rng.choice([0.125, 0.5, 2.0]))), None, V
    V = int(rng.integers(20, 60))
    if k == 2:
        g = synth.hl_graph(V=V, n_words=int(rng.integers(30, 600)), seed=int(rng.integers(0, 10**6)))
    else:
        n = int(rng.integers(30, 400))
        g = synth.hlg_graph(V=V, n_words=n, n_hist=n, avg_succ=int(rng.integers(3, 12)),
                            seed=int(rng.integers(0, 10**6)))
    return g, g["lexicon"], V


def same_lattice(gl, cl):
    return (gl.ok == cl.ok and gl.num_states == cl.num_states and gl.state_set() == cl.state_set() and
            gl.arc_set() == cl.arc_set() and len(gl.src) == len(cl.src) and gl.final_set() == cl.final_set())


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t_end = time.time() + budget
    n_cases = n_lat = n_1b = n_bad = n_split = n_hand = n_full = n_gc = n_paid = 0
    keep_going = os.environ.get("KD_FUZZ_KEEP_GOING") == "1"  # (diagnosis: count mismatches, do not stop)
    while time.time() < t_end:
        g, lex, V = graph(rng)
        if rng.random() < 0.25:
            # emitting arcs with negative weights (a weight-pushed graph): the label index's
            # lower bound uses the graph's lowest emitting weight
            w = g["weight"].copy()
            emit = g["ilabel"] != 0
            w[emit] -= np.float32(rng.uniform(0.1, 2.0)) * (rng.random(int(emit.sum())) < 0.5).astype(np.float32)
            g = dict(g, weight=w.astype(np.float32))
        G, OG = cabi.Graph(g), oracle.Graph(g)
        heavy = g["num_states"] >= 8000
        B = int(rng.integers(1, 4 if heavy else 9))
        T = int(rng.integers(1, 30 if heavy else 90))
        peak = float(rng.uniform(0.3, 3) if heavy else rng.uniform(0.5, 9))
        lps = [synth.make_logprobs(max(1, T - int(rng.integers(0, 10))), V, int(rng.integers(0, 10**6)),
                                   peak=peak, lexicon=lex) for _ in range(B)]
        dev = [torch.from_numpy(lp).cuda() for lp in lps]
        beam = float(rng.uniform(8, 25) if heavy else rng.uniform(2, 14))
        ma = INT_MAX if rng.random() < 0.6 else int(rng.integers(5, 400))
        mi = 0 if rng.random() < 0.5 else int(rng.integers(1, min(ma, 40)))
        small = rng.random() < 0.4
        cap = dict(arena_tokens=int(rng.integers(256, 4096)), max_frames=int(rng.integers(1, 16)),
                   table_slots=1 << 14) if small else {}
        one_launch = rng.random() < 0.5
        chunk = int(rng.integers(1, 25))
        # two kinds of workgroups: two calls in three are split, at thresholds that fuzz-sized
        # frontiers cross (KD_FUZZ_SPLIT=0/1 forces it off / on)
        fs = os.environ.get("KD_FUZZ_SPLIT")
        split = rng.random() < 0.67 if fs is None else fs == "1"
        hi = int(rng.choice([1, 2, 4, 8, 16, 64, 256]))
        lo = int(rng.integers(0, hi + 1))
        psm = int(rng.integers(0, 2))
        stay = int(rng.random() < 0.75)  # (else surplus workgroups leave a call of a few utterances at once)
        lazy = int(rng.integers(0, 2))   # PruneActiveTokens owed (run when a pool fills) or run when its frame comes

        def deal(b):
            try:
                if split:
                    b.set_split(enable=1, min_batch=1, mig_lo=lo, mig_hi=hi, prune_small=psm, stay=stay)
                else:
                    b.set_split(enable=0)
            except RuntimeError as e:
                if "second kernel" not in str(e):
                    raise
        tag = dict(B=B, T=T, V=V, beam=beam, ma=ma, mi=mi, small=small, one=one_launch, chunk=chunk,
                   states=g["num_states"], split=split, lo=lo, hi=hi, psm=psm, stay=stay, lazy=lazy)
        if rng.random() < 0.6:
            lbeam = float(rng.uniform(1, 8))
            pi = int(rng.integers(1, 30)) if rng.random() < 0.7 else 10**6
            opts = dict(beam=beam, lattice_beam=lbeam, max_active=ma, min_active=mi)
            if small:
                cap["arena_links"] = int(rng.integers(256, 4096))
            # pools that are not to grow (small cases): `no_grow`, or a memory budget equal to what
            # they start with (KD_GC_TOKENS / KD_GC_LINKS, read when the batch is created) -- the
            # owed prunes are then the garbage collection, paid under pool pressure or by the frame
            # that finds no room; a decode that does not fit even so may end in "pool full"
            gc = int(rng.integers(0, 4)) if small else 0  # 0, 1: pools grow; 2: no_grow; 3: budget
            if gc == 2:  # (nothing grows then: the frame index is sized for the utterance)
                cap["no_grow"] = 1
                cap["max_frames"] = T
            if gc == 3:
                os.environ["KD_GC_TOKENS"] = str(cap["arena_tokens"])
                os.environ["KD_GC_LINKS"] = str(cap["arena_links"])
            tag["gc"] = gc
            try:
                lb = cabi.LatticeBatch(G, max_batch=B, prune_interval=pi, **opts, **cap)
            finally:
                os.environ.pop("KD_GC_TOKENS", None)
                os.environ.pop("KD_GC_LINKS", None)
            deal(lb)
            lb.set_lazy_prune(lazy)
            try:
                if one_launch:
                    lb.decode_device([d.data_ptr() for d in dev], [lp.shape[0] for lp in lps], V)
                else:
                    lb.init_decoding(B)
                    for _ in range(0, T, chunk):
                        lb.advance_host(lps, max_num_frames=chunk)
                    lb.finalize_decoding()
            except RuntimeError as e:
                if gc != 2 or "full" not in str(e):  # (token / link pool, or the tables of a heavy case)
                    raise
                n_full += 1
                n_cases += 1
                continue
            n_gc += int(gc >= 2)
            n_paid += int(gc >= 2 and lazy and int(lb.stats()[:, cabi.STAT_NAMES.index("prunes")].sum()) > 0)
            lats, frc = lb.raw_lattices(), lb.final_relative_cost()
            n_hand += int(lb.stats()[:, cabi.STAT_NAMES.index("handovers")].sum())
            for i, lp in enumerate(lps):
                can = oracle.CanonLatticeDecoder(OG, **opts)
                can.decode(lp)
                if not same_lattice(lats[i], can.get_raw_lattice()) or not (
                        frc[i] == can.final_relative_cost() or
                        (np.isinf(frc[i]) and np.isinf(can.final_relative_cost()))):
                    print("LATTICE MISMATCH", tag, dict(lbeam=lbeam, pi=pi, utt=i), flush=True)
                    n_bad += 1
                    if not keep_going:
                        sys.exit(1)
            n_lat += 1
        else:
            opts = dict(beam=beam, max_active=ma, min_active=mi)
            bt = cabi.Batch(G, max_batch=B, **opts, **cap)
            deal(bt)
            if one_launch:
                bt.decode_device([d.data_ptr() for d in dev], [lp.shape[0] for lp in lps], V)
            else:
                bt.init_decoding(B)
                for _ in range(0, T, chunk):
                    bt.advance_host(lps, max_num_frames=chunk)
            uf = bool(rng.integers(0, 2))
            paths, rf = bt.best_paths(uf), bt.reached_final()
            n_hand += int(bt.stats()[:, cabi.STAT_NAMES.index("handovers")].sum())
            for i, lp in enumerate(lps):
                d = oracle.FasterDecoder(OG, policy=1, **opts)
                d.decode(lp)
                p = d.get_best_path(uf)
                q = paths[i]
                if (q.ok != p.ok or q.ilabel.tolist() != p.ilabel.tolist() or q.state.tolist() != p.state.tolist()
                        or q.ac_cost.tolist() != p.ac_cost.tolist() or q.graph_cost.tolist() != p.graph_cost.tolist()
                        or tuple(q.final_weight) != tuple(p.final_weight) or bool(rf[i]) != d.reached_final()):
                    print("PATH MISMATCH", tag, dict(uf=uf, utt=i), flush=True)
                    n_bad += 1
                    if not keep_going:
                        sys.exit(1)
            n_1b += 1
        n_cases += 1
        n_split += int(split)
    if n_bad:
        print(f"fuzz FAILED: {n_bad} mismatching utterances in {n_cases} cases")
        sys.exit(1)
    if n_split >= 50 and n_hand == 0:
        print(f"fuzz FAILED: {n_split} cases dealt out over two kinds of workgroups and not one hand-over")
        sys.exit(1)
    print(f"fuzz ok: {n_cases} cases ({n_lat} lattice, {n_1b} 1-best; {n_split} over two kinds of "
          f"workgroups, {n_hand} hand-overs; {n_gc} in pools not to grow of which {n_paid} paid owed prunes, "
          f"{n_full} more ended in 'pool full') in {budget:.0f} s")


if __name__ == "__main__":
    main()
