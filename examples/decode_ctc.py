#!/usr/bin/env python3
"""icefall-style CTC decoding with the drop-in package, on synthetic inputs.

The call sequence is the one of icefall's jit_pretrained_decode_with_{H,HL,HLG}.py
(the scripts the reference's README points to): build a decodable from the
network's log-probs, decode with FasterDecoder(fst, FasterDecoderOptions(max_active=3000)),
check reached_final(), take get_best_path() and its linear symbol sequence; for an H
graph the token ids are ilabel - 1 with repeats collapsed and blanks dropped, for
HL / HLG the word ids are the olabels.  Hypotheses are scored against the word
sequence the synthetic posteriors were drawn from (word error rate), and the
lattice decoder's best path must agree with the 1-best decoder's.

    python examples/decode_ctc.py [H|HL|HLG] [num_utts]

Needs a GPU and the built library (python __graft_entry__.py).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import kaldi_decoder as kd  # noqa: E402
from kaldi_decoder import synth  # noqa: E402
from kaldi_decoder.scoring import ctc_collapse, edit_distance  # noqa: E402


def main(kind=None, n_utts=None, graph_path=None, results=None):
    """Returns (WER, utterances on which the lattice decoder's best path equals the 1-best
    decoder's).  With `graph_path` the graph goes through an OpenFst binary file, as a real
    setup's HLG.fst does.  `results` (a list) receives, per utterance, what the 1-best decoder
    returned: (log_probs, isyms, osyms, (graph cost, acoustic cost), reached_final) -- and its
    first entry is the graph dictionary the file was written from (tests compare with the
    oracle's decode of THAT graph)."""
    kind = kind or (sys.argv[1] if len(sys.argv) > 1 else "HLG")
    n_utts = n_utts or (int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    V, T = 60, 120
    if kind == "H":
        g, lex = synth.ctc_topo(V), None
    elif kind == "HL":
        g = synth.hl_graph(V=V, n_words=500, seed=1)
        lex = g["lexicon"]
    else:
        g = synth.hlg_graph(V=V, n_words=500, n_hist=500, avg_succ=10, seed=2)
        lex = g["lexicon"]
    if results is not None:
        results.append(g)
    fst = kd.fst_from_dict(g)
    if graph_path:  # a real setup: kd.Fst.read("HLG.fst")
        fst.write(graph_path)
        fst = kd.Fst.read(graph_path)
    decoder = kd.FasterDecoder(fst, kd.FasterDecoderOptions(max_active=3000))
    lat_decoder = kd.LatticeFasterDecoder(fst, kd.LatticeFasterDecoderConfig(beam=16.0, lattice_beam=8.0))
    errs = words = agree = 0
    for utt in range(n_utts):
        log_probs = synth.make_logprobs(T, V, utt, peak=10.0, lexicon=lex)
        decodable = kd.DecodableCtc(log_probs)
        decoder.decode(decodable)
        if not decoder.reached_final():
            print(f"utt {utt}: no final state reached")
        ok, best_path = decoder.get_best_path()
        ok2, isyms, osyms, total = kd.get_linear_symbol_sequence(best_path)
        hyp = ctc_collapse(isyms) if kind == "H" else list(osyms)
        if results is not None:
            results.append((log_probs, list(isyms), list(osyms), (float(total[0]), float(total[1])),
                            decoder.reached_final()))
        lat_decoder.decode(kd.DecodableCtc(log_probs))
        ok3, lat_best = lat_decoder.get_best_path()
        _, li, lo, _ = kd.get_linear_symbol_sequence(lat_best)
        agree += (ctc_collapse(li) if kind == "H" else list(lo)) == hyp
        ref = synth.hidden_word_sequence(T, V, utt, lexicon=lex) if lex is not None else None
        if ref is not None:
            errs += edit_distance(ref, hyp)
            words += len(ref)
        print(f"utt {utt}: {len(hyp)} symbols, cost {float(total[0]) + float(total[1]):.3f}: {hyp[:12]}{' ...' if len(hyp) > 12 else ''}")
    print(f"lattice best path == 1-best on {agree}/{n_utts} utterances")
    if words:
        print(f"WER against the hidden word sequences: {100.0 * errs / words:.2f} % ({errs}/{words})")
    return (errs / words if words else None), agree


if __name__ == "__main__":
    main()
