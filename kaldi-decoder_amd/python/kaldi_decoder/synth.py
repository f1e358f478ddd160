"""Seeded synthetic decoding graphs and CTC log-prob matrices.

No LibriSpeech data, kaldifst or real H/HL/HLG files exist offline, so every
test and the benchmark run on graphs built here:

* ``ctc_topo(V)``       -- the dense CTC topology H: V states, V*V arcs.
* ``hl_graph(...)``     -- H o L like: a determinised token trie per lexicon with
                           blank/self-loop structure and epsilon word-end arcs.
* ``hlg_graph(...)``    -- H o L o G like: one trie per bigram history plus a
                           unigram trie reached through backoff epsilon arcs.
* ``make_logprobs(...)``-- peaky CTC posteriors that follow a hidden path.

Graphs are plain CSR dictionaries (numpy arrays); arc input labels follow the
reference's convention that ilabel = token id + 1 and 0 is epsilon
(/root/reference/kaldi-decoder/csrc/decodable-ctc.cc:22-29).
"""

from __future__ import annotations

import numpy as np

INF = np.float32(np.inf)


def _to_csr(num_states, start, src, ilabel, olabel, weight, dst, final):
    src = np.asarray(src, dtype=np.int64)
    order = np.argsort(src, kind="stable")
    counts = np.bincount(src, minlength=num_states)
    row_ptr = np.zeros(num_states + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    assert row_ptr[-1] < 2**31
    return {
        "num_states": int(num_states),
        "start": int(start),
        "row_ptr": row_ptr.astype(np.int32),
        "ilabel": np.asarray(ilabel, dtype=np.int32)[order],
        "olabel": np.asarray(olabel, dtype=np.int32)[order],
        "weight": np.asarray(weight, dtype=np.float32)[order],
        "nextstate": np.asarray(dst, dtype=np.int32)[order],
        "final": np.asarray(final, dtype=np.float32),
    }


def graph_from_arcs(num_states, start, arcs, finals):
    """arcs: iterable of (src, dst, ilabel, olabel, weight); finals: {state: cost}."""
    arcs = list(arcs)
    final = np.full(num_states, INF, dtype=np.float32)
    for s, w in finals.items():
        final[s] = w
    if arcs:
        src, dst, il, ol, w = zip(*arcs)
    else:
        src = dst = il = ol = w = ()
    return _to_csr(num_states, start, src, il, ol, w, dst, final)


def ctc_topo(V: int):
    """Standard CTC topology: state i == "last emitted token i" (0 = blank).

    Arc i->j carries ilabel j+1 and olabel j unless i == j (a repeat) or j is
    blank, where the olabel is epsilon.  All states final with weight 0.
    """
    i = np.repeat(np.arange(V), V)
    j = np.tile(np.arange(V), V)
    ol = np.where(i != j, j, 0)
    return _to_csr(V, 0, i, j + 1, ol, np.zeros(V * V, np.float32), j,
                   np.zeros(V, np.float32))


def random_graph(num_states, avg_degree, V, eps_frac, seed, n_olabels=50,
                 final_frac=0.3, max_weight=3.0, eps_back=0.0, eps_back_min_weight=0.25):
    """Unstructured sparse graph with epsilon arcs and final weights.

    Epsilon arcs go from a lower to a higher state id -- no epsilon cycles -- except the
    fraction `eps_back` of them, which lead to ANY state and weigh at least
    `eps_back_min_weight`: every epsilon cycle then holds such an arc and (weights being
    non-negative) has a positive total weight, the kind of cycle the reference's
    ProcessNonemitting terminates on by strict improvement (faster-decoder.cc:97-113).
    Every state gets at least one emitting arc.
    """
    rng = np.random.default_rng(seed)
    S = num_states
    n_arcs = int(S * avg_degree)
    src = np.concatenate([np.arange(S), rng.integers(0, S, n_arcs - S)])
    dst = rng.integers(0, S, n_arcs)
    il = rng.integers(1, V + 1, n_arcs)
    is_eps = rng.random(n_arcs) < eps_frac
    is_eps[:S] = False
    is_eps &= src < S - 1
    fwd = src + 1 + (dst % np.maximum(S - 1 - src, 1))
    back = np.zeros(n_arcs, bool)
    if eps_back > 0.0:  # (drawn only when asked for: graphs of earlier seeds stay what they were)
        back = is_eps & (rng.random(n_arcs) < eps_back)
    dst = np.where(is_eps & ~back, fwd, dst)
    il = np.where(is_eps, 0, il)
    ol = np.where(rng.random(n_arcs) < 0.4, rng.integers(1, n_olabels + 1, n_arcs), 0)
    w = (rng.random(n_arcs) * max_weight).astype(np.float32)
    w = np.where(back, np.maximum(w, np.float32(eps_back_min_weight)), w).astype(np.float32)
    perm = rng.permutation(n_arcs)
    final = np.where(rng.random(S) < final_frac,
                     (rng.random(S) * 2.0).astype(np.float32), INF).astype(np.float32)
    return _to_csr(S, 0, src[perm], il[perm], ol[perm], w[perm], dst[perm], final)


def _zipf_probs(n, s=1.0):
    p = 1.0 / np.arange(1, n + 1) ** s
    return p / p.sum()


def make_lexicon(n_words, V, seed, max_len=6):
    """Token sequences for words 1..n_words (row 0 unused); tokens in 1..V-1."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, max_len + 1, n_words + 1)
    toks = rng.integers(1, V, (n_words + 1, max_len))
    toks[np.arange(max_len)[None, :] >= lens[:, None]] = -1
    lens[0] = 0
    toks[0] = -1
    return toks.astype(np.int64), lens.astype(np.int64)


def _trie_graph(n_hist, pair_hist, pair_word, pair_cost, pair_dest, toks, lens,
                V, backoff_cost, hist_final, seed):
    """One token trie per history state; see hl_graph / hlg_graph."""
    rng = np.random.default_rng(seed)
    P = len(pair_hist)
    max_len = toks.shape[1]
    plen = lens[pair_word]
    BLANK_IL = 1  # blank token 0 -> ilabel 1

    src, dst, il, ol, w = [], [], [], [], []

    def add(s, d, i, o, ww):
        n = len(s)
        src.append(np.asarray(s, np.int64))
        dst.append(np.asarray(d, np.int64))
        il.append(np.broadcast_to(np.asarray(i, np.int64), (n,)))
        ol.append(np.broadcast_to(np.asarray(o, np.int64), (n,)))
        w.append(np.broadcast_to(np.asarray(ww, np.float32), (n,)))

    next_state = n_hist
    # node id (global, per depth) of each pair's prefix; depth 0 == history state
    parent_E = pair_hist.copy()  # state that emits on a *different* token
    parent_B = pair_hist.copy()  # state that emits on any token
    parent_tok = np.full(P, -1, np.int64)
    parent_key = pair_hist.copy()
    end_E = np.zeros(P, np.int64)
    end_B = np.zeros(P, np.int64)
    for d in range(max_len):
        sel = np.nonzero(plen > d)[0]
        if len(sel) == 0:
            break
        t = toks[pair_word[sel], d]
        key = parent_key[sel] * V + t
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        n_nodes = len(uniq)
        E = next_state + 2 * np.arange(n_nodes)
        B = E + 1
        next_state += 2 * n_nodes
        rep = sel[first]
        tt = t[first]
        # entering arcs from the parent node
        pE, pB, pt = parent_E[rep], parent_B[rep], parent_tok[rep]
        if d == 0:
            add(pB, E, tt + 1, 0, 0.0)
        else:
            diff = pt != tt
            add(pE[diff], E[diff], tt[diff] + 1, 0, 0.0)
            add(pB, E, tt + 1, 0, 0.0)
        # word-end epsilon arcs are appended between the structural arcs so
        # that epsilon and emitting arcs interleave inside a state's arc list
        ends = np.nonzero(plen[sel] == d + 1)[0]
        pe = sel[ends]
        end_E[pe] = E[inv[ends]]
        end_B[pe] = B[inv[ends]]
        add(end_E[pe], pair_dest[pe], 0, pair_word[pe], pair_cost[pe])
        add(E, E, tt + 1, 0, 0.0)            # token self-loop
        add(E, B, BLANK_IL, 0, 0.0)          # token -> blank
        add(B, B, BLANK_IL, 0, 0.0)          # blank self-loop
        add(end_B[pe], pair_dest[pe], 0, pair_word[pe], pair_cost[pe])
        parent_E[sel] = E[inv]
        parent_B[sel] = B[inv]
        parent_tok[sel] = t
        parent_key[sel] = (next_state - 2 * n_nodes - n_hist) // 2 + inv + n_hist
    hs = np.arange(n_hist)
    add(hs, hs, BLANK_IL, 0, 0.0)            # blank self-loop on history states
    if n_hist > 1:
        add(hs[1:], np.zeros(n_hist - 1, np.int64), 0, 0, backoff_cost)
    S = next_state
    final = np.full(S, INF, np.float32)
    final[:n_hist] = hist_final
    del rng
    return _to_csr(S, 0, np.concatenate(src), np.concatenate(il), np.concatenate(ol),
                   np.concatenate(w), np.concatenate(dst), final)


def hl_graph(V=500, n_words=20000, seed=1, max_len=6):
    """H o L like graph: one determinised token trie looping through a root.

    V=500, n_words=20000 gives about 1e5 states / 3e5 arcs (BASELINE cfg2).
    """
    toks, lens = make_lexicon(n_words, V, seed, max_len)
    rng = np.random.default_rng(seed + 7)
    words = np.arange(1, n_words + 1)
    cost = (rng.random(n_words) * 2.0).astype(np.float32)
    g = _trie_graph(1, np.zeros(n_words, np.int64), words, cost,
                    np.zeros(n_words, np.int64), toks, lens, V, 0.0,
                    np.zeros(1, np.float32), seed)
    g["lexicon"] = (toks, lens)
    return g


def hlg_graph(V=500, n_words=20000, n_hist=None, avg_succ=35, seed=2, max_len=6):
    """H o L o G like graph with a bigram-ish G.

    History state 0 is the unigram state (a trie over all words, cost
    -log p_uni); history h>0 is "last word was h" with a Zipf-sized successor
    set and a backoff epsilon arc to state 0.  Word-end epsilon arcs carry the
    word as olabel and its LM cost.  n_words=20000, n_hist=20000, avg_succ=35
    gives about 5e6 states / 1.5e7 arcs (BASELINE cfg3).
    """
    if n_hist is None:
        n_hist = n_words
    n_hist = min(n_hist, n_words)
    toks, lens = make_lexicon(n_words, V, seed, max_len)
    rng = np.random.default_rng(seed + 11)
    p_uni = _zipf_probs(n_words)
    words = np.arange(1, n_words + 1)

    def dest_of(wd):
        return np.where(wd <= n_hist, wd, 0).astype(np.int64)

    # unigram pairs
    ph = [np.zeros(n_words, np.int64)]
    pw = [words]
    pc = [(-np.log(p_uni)).astype(np.float32)]
    # bigram pairs
    k = np.clip(rng.zipf(1.3, n_hist) + 1, 2, 12 * avg_succ)
    k = np.maximum(2, (k * (avg_succ / max(k.mean(), 1e-9))).astype(np.int64))
    k = np.minimum(k, n_words)
    tot = int(k.sum())
    bh = np.repeat(np.arange(1, n_hist + 1), k)
    bw = rng.choice(words, size=tot, p=p_uni)
    # de-duplicate (history, word)
    key = bh.astype(np.int64) * (n_words + 1) + bw
    _, first = np.unique(key, return_index=True)
    first.sort()
    bh, bw = bh[first], bw[first]
    bc = (-np.log(np.clip(rng.random(len(bh)), 1e-3, 1.0)) * 0.8 + 0.3).astype(np.float32)
    ph.append(bh)
    pw.append(bw)
    pc.append(bc)
    pair_hist = np.concatenate(ph)
    pair_word = np.concatenate(pw)
    pair_cost = np.concatenate(pc)
    hist_final = (rng.random(n_hist + 1) * 1.0).astype(np.float32)
    g = _trie_graph(n_hist + 1, pair_hist, pair_word, pair_cost, dest_of(pair_word),
                    toks, lens, V, np.float32(1.5), hist_final, seed)
    g["lexicon"] = (toks, lens)
    return g


def _hidden(T, V, seed, lexicon=None, p_blank=0.6):
    """(per-frame hot token, the hidden words whose tokens were all emitted)."""
    rng = np.random.default_rng(seed)
    words, ends = [], []
    if lexicon is None:
        seq = rng.integers(1, V, T + 1)
    else:
        toks, lens = lexicon
        n_words = len(lens) - 1
        p = _zipf_probs(n_words)
        seq = []
        while len(seq) < T + 1:
            wd = int(rng.choice(n_words, p=p)) + 1
            seq.extend(toks[wd, :lens[wd]].tolist())
            words.append(wd)
            ends.append(len(seq))
        seq = np.asarray(seq)
    is_blank = rng.random(T) < p_blank
    pos = np.cumsum(~is_blank) - 1
    hot = np.where(is_blank, 0, seq[np.clip(pos, 0, len(seq) - 1)])
    emitted = int((~is_blank).sum())
    return hot.astype(np.int64), [w for w, e in zip(words, ends) if e <= emitted]


def hidden_token_path(T, V, seed, lexicon=None, p_blank=0.6):
    """Per-frame hot token: blank w.p. p_blank else the next token of a hidden
    word sequence (or of a random token sequence when no lexicon is given)."""
    return _hidden(T, V, seed, lexicon, p_blank)[0]


def hidden_word_sequence(T, V, utt_id=0, lexicon=None, p_blank=0.6):
    """The word ids behind make_logprobs(T, V, utt_id, lexicon=lexicon): the words whose
    tokens the hot path emitted completely ([] without a lexicon)."""
    return _hidden(T, V, 1000 + utt_id + 500000, lexicon, p_blank)[1]


def make_logprobs(T, V, utt_id=0, peak=12.0, lexicon=None, p_blank=0.6):
    """[T, V] float32 log-softmax rows; rng seed is 1000 + utt_id."""
    seed = 1000 + utt_id
    hot = hidden_token_path(T, V, seed + 500000, lexicon, p_blank)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((T, V), dtype=np.float32)
    x[np.arange(T), hot] += np.float32(peak)
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m, dtype=np.float32)
    lse = np.log(e.sum(axis=1, keepdims=True, dtype=np.float32), dtype=np.float32)
    return np.ascontiguousarray((x - m - lse).astype(np.float32))


def describe(g):
    deg = np.diff(g["row_ptr"].astype(np.int64))
    n_eps = int((g["ilabel"] == 0).sum())
    return {
        "states": g["num_states"],
        "arcs": int(g["row_ptr"][-1]),
        "eps_arcs": n_eps,
        "max_degree": int(deg.max()),
        "mean_degree": float(deg.mean()),
        "final_states": int(np.isfinite(g["final"]).sum()),
    }
