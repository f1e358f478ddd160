"""Exact (unpruned) Viterbi over a CSR graph in plain numpy.

An independent known-answer generator for the 1-best path: no beam, no hash
list, no token objects -- a dense float64 cost vector per frame, relaxed with
``np.minimum.at``.  With an infinite beam the reference's FasterDecoder
(/root/reference/kaldi-decoder/csrc/faster-decoder.cc:155-241) must return
exactly this path; candidate costs use the same expression
``(cost + weight) + acoustic`` in float64 so the minima agree bit for bit.
"""

import numpy as np


def _closure(cost, bp, src, dst, w, arc_ids):
    """Bellman-Ford over epsilon arcs; bp gets the lowest arc id achieving the min."""
    if len(src) == 0:
        return
    while True:
        cand = cost[src] + w
        new = cost.copy()
        np.minimum.at(new, dst, cand)
        changed = new < cost
        if not changed.any():
            return
        hit = changed[dst] & (cand == new[dst])
        # assign in reverse so the lowest arc id is written last
        idx = np.nonzero(hit)[0][::-1]
        bp[dst[idx]] = arc_ids[idx]
        cost[:] = new


def viterbi(g, log_probs):
    """Returns dict(ok, ilabel, olabel, state, cost, final, reached_final)."""
    S = g["num_states"]
    row_ptr = g["row_ptr"].astype(np.int64)
    arc_src = np.repeat(np.arange(S), np.diff(row_ptr))
    il, ol = g["ilabel"], g["olabel"]
    w64 = g["weight"].astype(np.float64)
    nxt = g["nextstate"].astype(np.int64)
    em = np.nonzero(il != 0)[0]
    ep = np.nonzero(il == 0)[0]
    T = log_probs.shape[0]
    cost = np.full(S, np.inf)
    cost[g["start"]] = 0.0
    bps = []
    bp = np.full(S, -1, np.int64)
    _closure(cost, bp, arc_src[ep], nxt[ep], w64[ep], ep)
    bps.append(bp)
    for t in range(T):
        ac = (-log_probs[t, il[em] - 1]).astype(np.float32).astype(np.float64)
        cand = (cost[arc_src[em]] + w64[em]) + ac
        new = np.full(S, np.inf)
        np.minimum.at(new, nxt[em], cand)
        bp = np.full(S, -1, np.int64)
        hit = np.isfinite(cand) & (cand == new[nxt[em]])
        idx = np.nonzero(hit)[0][::-1]
        bp[nxt[em][idx]] = em[idx]
        cost = new
        _closure(cost, bp, arc_src[ep], nxt[ep], w64[ep], ep)
        bps.append(bp)
    fin = g["final"].astype(np.float64)
    reached = bool(np.any(np.isfinite(cost) & np.isfinite(fin)))
    if not np.isfinite(cost).any():
        return {"ok": False}
    tot = cost + fin if reached else cost
    s = int(np.argmin(tot))
    best_state, best_total = s, float(tot[s])
    arcs = []
    layer = T
    while True:
        a = bps[layer][s]
        if a < 0:
            break
        arcs.append(int(a))
        s = int(arc_src[a])
        if il[a] != 0:
            layer -= 1
    assert s == g["start"] and layer == 0
    arcs = np.asarray(arcs[::-1], np.int64)
    return {
        "ok": True,
        "arcs": arcs,
        "ilabel": il[arcs],
        "olabel": ol[arcs],
        "state": nxt[arcs].astype(np.int32),
        "cost": float(cost[best_state]),
        "total": best_total,
        "final": float(fin[best_state]) if reached else 0.0,
        "reached_final": reached,
    }
