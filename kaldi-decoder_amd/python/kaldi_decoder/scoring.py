"""What icefall's CTC decode scripts do with a best path (the callers of the reference,
SURVEY.md 8(b) "Who calls it" / 8(f)-4): token post-processing for an H graph, and the
word error rate of hypotheses against references."""

from __future__ import annotations

import numpy as np


def edit_distance(ref, hyp):
    """Levenshtein distance between two sequences (substitutions, insertions and
    deletions all cost 1)."""
    ref, hyp = list(ref), list(hyp)
    d = np.arange(len(hyp) + 1)
    for i, x in enumerate(ref, 1):
        prev, d[0] = d[0], i
        for j, y in enumerate(hyp, 1):
            prev, d[j] = d[j], min(d[j] + 1, d[j - 1] + 1, prev + (x != y))
    return int(d[len(hyp)])


def ctc_collapse(ilabels):
    """Token ids of a best path through an H (CTC topology) graph: token = ilabel - 1
    (ilabel 0 is epsilon), repeats collapsed, blanks (token 0) dropped."""
    toks = [int(i) - 1 for i in ilabels if i > 0]
    out = []
    for k, t in enumerate(toks):
        if t != 0 and (k == 0 or t != toks[k - 1]):
            out.append(t)
    return out


def word_error_rate(refs, hyps):
    """(WER, errors, reference words) over pairs of sequences."""
    errs = sum(edit_distance(r, h) for r, h in zip(refs, hyps))
    words = sum(len(list(r)) for r in refs)
    return (errs / words if words else 0.0), errs, words
