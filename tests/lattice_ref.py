"""An INDEPENDENT statement of what a finalized raw lattice is -- not a restatement of the
reference's algorithm (oracle/kd_oracle.cc is that), but of its result, computed another way.

With a search beam so wide that nothing is pruned during the search, the forward link
(frame f, state s) --arc a--> (f', d) survives PruneForwardLinksFinal / PruneForwardLinks
(ref lattice-simple-decoder.cc:228-305, :425-520) iff

        alpha(f, s) + w(a) + ac(f, a) + beta(f', d) - best  <=  lattice_beam

where alpha is the exact forward Viterbi cost of (f, s), beta the exact backward cost to the end
of the utterance (final costs included; 0 everywhere when no final state was reached,
ref :449-456), and best = min_s alpha(T, s) + final(s).  [The reference's recursion is
link_extra = next.extra + ((tok.tot + ac + graph) - next.tot) (:257-260) with tok.tot = alpha
and next.extra = alpha(next) + beta(next) - best by induction from the last frame.]  A token
survives iff one of its links does (last frame: iff alpha + final - best <= lattice_beam).

alpha and beta are computed here in float64 by Bellman-Ford relaxation over dense per-frame
arrays (epsilon arcs relax inside a frame until nothing changes: epsilon cycles of positive
weight are fine), with no token lists, no hash containers, no running cutoffs and no notion of
order.  The decoders compute in float32, so membership is asserted outside a margin around the
threshold; inside it either answer is accepted, and the tests check that hardly any link
falls there.
"""
import numpy as np

INF = np.inf


def lattice_by_viterbi(g, log_probs, lattice_beam):
    """-> (links, tokens, best): links = {(f, s, f', d, ilabel, olabel): extra}, tokens =
    {(f, s): extra} over EVERY reachable link / token; best = the best total cost."""
    S = int(g["num_states"])
    row_ptr = np.asarray(g["row_ptr"])
    src = np.repeat(np.arange(S), np.diff(row_ptr))
    dst = np.asarray(g["nextstate"], np.int64)
    il = np.asarray(g["ilabel"], np.int64)
    ol = np.asarray(g["olabel"], np.int64)
    w = np.asarray(g["weight"], np.float64)
    final = np.asarray(g["final"], np.float64)
    lp = np.asarray(log_probs, np.float64)
    T = lp.shape[0]
    eps = il == 0
    e_src, e_dst, e_w = src[eps], dst[eps], w[eps]
    m_src, m_dst, m_w, m_il = src[~eps], dst[~eps], w[~eps], il[~eps]

    def close_fwd(a):
        while True:
            new = a.copy()
            np.minimum.at(new, e_dst, a[e_src] + e_w)
            if not (new < a).any():
                return a
            a = new

    def close_bwd(b):
        while True:
            new = b.copy()
            np.minimum.at(new, e_src, b[e_dst] + e_w)
            if not (new < b).any():
                return b
            b = new

    alpha = np.full((T + 1, S), INF)
    alpha[0, g["start"]] = 0.0
    alpha[0] = close_fwd(alpha[0])
    for f in range(T):
        nxt = np.full(S, INF)
        np.minimum.at(nxt, m_dst, alpha[f, m_src] + m_w - lp[f, m_il - 1])
        alpha[f + 1] = close_fwd(nxt)
    reached = np.isfinite(alpha[T])
    any_final = bool((reached & np.isfinite(final)).any())
    fin = final if any_final else np.zeros(S)
    beta = np.full((T + 1, S), INF)
    beta[T] = close_bwd(np.where(reached, fin, INF))
    for f in range(T - 1, -1, -1):
        b = np.full(S, INF)
        np.minimum.at(b, m_src, m_w - lp[f, m_il - 1] + beta[f + 1, m_dst])
        beta[f] = close_bwd(b)
    best = float((alpha[T] + np.where(reached, fin, INF)).min())
    links, tokens = {}, {}
    for f in range(T + 1):
        ok = np.isfinite(alpha[f, e_src])
        for k in np.nonzero(ok)[0]:
            x = alpha[f, e_src[k]] + e_w[k] + beta[f, e_dst[k]] - best
            key = (f, int(e_src[k]), f, int(e_dst[k]), 0, int(ol[eps][k]))
            links[key] = min(links.get(key, INF), float(x))
        if f < T:
            ok = np.isfinite(alpha[f, m_src])
            for k in np.nonzero(ok)[0]:
                x = alpha[f, m_src[k]] + m_w[k] - lp[f, m_il[k] - 1] + beta[f + 1, m_dst[k]] - best
                key = (f, int(m_src[k]), f + 1, int(m_dst[k]), int(m_il[k]), int(ol[~eps][k]))
                links[key] = min(links.get(key, INF), float(x))
        for s in np.nonzero(np.isfinite(alpha[f]))[0]:
            tokens[(f, int(s))] = float(alpha[f, s] + beta[f, s] - best)
    return links, tokens, best


def check_lattice(lat, g, log_probs, lattice_beam, margin):
    """Asserts that `lat` (a RawLattice of cabi or of the oracle) is the lattice the
    definition above gives: every link / token with extra < lattice_beam - margin is there,
    none with extra > lattice_beam + margin is.  Returns how many links lie inside the
    margin (either answer accepted)."""
    links, tokens, _ = lattice_by_viterbi(g, log_probs, lattice_beam)
    f, s = lat.state_frame, lat.state_graph
    have = set(zip(f[lat.src].tolist(), s[lat.src].tolist(), f[lat.dst].tolist(), s[lat.dst].tolist(),
                   lat.ilabel.tolist(), lat.olabel.tolist()))
    # (parallel arcs with equal labels between the same two states are one key here)
    must = {k for k, x in links.items() if x < lattice_beam - margin}
    may = {k for k, x in links.items() if x <= lattice_beam + margin}
    assert must <= have, sorted(must - have)[:5]
    assert have <= may, sorted(have - may)[:5]
    toks_have = set(zip(f.tolist(), s.tolist()))
    t_must = {k for k, x in tokens.items() if x < lattice_beam - margin}
    t_may = {k for k, x in tokens.items() if x <= lattice_beam + margin}
    assert t_must <= toks_have, sorted(t_must - toks_have)[:5]
    assert toks_have <= t_may, sorted(toks_have - t_may)[:5]
    return len(may) - len(must)
