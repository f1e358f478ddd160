"""Loading tests/golden/*.npz (see tests/golden/make_golden.py: oracle-generated,
parity unpinned by the reference)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRAPH_KEYS = ("row_ptr", "ilabel", "olabel", "weight", "nextstate", "final")


def names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load(name):
    """(fixture, graph dict, options).  Large cases store the generator call instead of
    the arrays (`gen`) and a digest of what it produced when the fixture was made."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    if "gen" in z.files:
        import hashlib
        import json

        from kaldi_decoder import synth
        spec = json.loads(str(z["gen"]))
        g = getattr(synth, spec["graph"])(**spec["graph_kwargs"])
        lk = dict(spec["logprobs"])
        lex = g["lexicon"] if lk.pop("lexicon") else None
        lp = synth.make_logprobs(lk["T"], lk["V"], lk["utt_id"], peak=lk["peak"], lexicon=lex)
        h = hashlib.sha256()
        for k in GRAPH_KEYS:
            h.update(np.ascontiguousarray(g[k]).tobytes())
        h.update(np.ascontiguousarray(lp).tobytes())
        assert h.hexdigest() == str(z["gen_sha256"]), f"{name}: the input generator drifted"
        z = dict(z)
        z["logprobs"] = lp
        g = {**{k: np.asarray(g[k]) for k in GRAPH_KEYS}, "num_states": int(g["num_states"]),
             "start": int(g["start"])}
    else:
        g = {k: z[k] for k in GRAPH_KEYS}
        g["num_states"] = int(z["num_states"])
        g["start"] = int(z["start"])
    opts = {}
    for k, v in zip(z["opt_names"].tolist(), z["opt_values"].tolist()):
        opts[k] = int(v) if k in ("max_active", "min_active") else float(v)
    return z, g, opts


def assert_path(p, z, prefix, tag):
    assert bool(p.ok) == bool(z[prefix + "ok"]), tag
    for k in ("ilabel", "olabel", "state", "graph_cost", "ac_cost"):
        assert getattr(p, k).tolist() == z[prefix + k].tolist(), (tag, k)
    assert np.asarray(p.final_weight, np.float32).tolist() == z[prefix + "final_weight"].tolist(), tag


def lattice_rows(lat):
    rows = sorted(lat.arc_set())
    return np.array(rows, np.float64).reshape(len(rows), -1)


def lattice_forward_costs(rows, start_state):
    """Forward (alpha) cost of every lattice state, in the decoder's own float32
    arithmetic: emitting arcs `(cur + ac) + graph` (ref lattice-simple-decoder.cc:383-385),
    epsilon arcs `cur + graph` (:170-172).  `rows` are arc rows (src frame, src graph
    state, dst frame, dst graph state, ilabel, olabel, graph, ac).  A token's best
    incoming link has link_extra_cost == its own extra cost, so it survives whenever the
    token does: on a pruned raw lattice these are exactly the tokens' tot_cost."""
    alpha = {(0, int(start_state)): np.float32(0.0)}
    by_frame = {}
    for r in rows:
        by_frame.setdefault(int(r[2]), []).append(r)
    inf = np.float32(np.inf)
    for f in sorted(by_frame):
        arcs = by_frame[f]
        changed = True
        first = True
        while changed:
            changed = False
            for r in arcs:
                eps = int(r[0]) == int(r[2])
                if not first and not eps:
                    continue
                cur = alpha.get((int(r[0]), int(r[1])), inf)
                if cur == inf:
                    continue
                if eps:
                    c = np.float32(cur + np.float32(r[6]))
                else:
                    c = np.float32(np.float32(cur + np.float32(r[7])) + np.float32(r[6]))
                k = (int(r[2]), int(r[3]))
                if c < alpha.get(k, inf):
                    alpha[k] = c
                    changed = True
            first = False
    return alpha


def assert_extras_beyond_cutoff(ref_rows, created_rows, frame_best, beam, start_state, tag=""):
    """SURVEY section 7 ("order-dependent pruning"), made exact.  The reference tightens
    its cutoff while it iterates, so it also creates links under a looser, order-dependent
    cutoff; the order-free decoder creates exactly the links below the value the running
    cutoff ends the frame at, (best cost of the destination frame) + beam.  Hence for every
    arc of the reference-order lattice:
        forward cost <  frame-best + beam   <=>   the order-free decoder created that link
    (`created_rows`: its links BEFORE lattice-beam pruning).  The survey's shorter form --
    "every reference-only arc of the pruned lattices is beyond the cutoff" -- is false:
    a link both decoders create can survive FinalizeDecoding in the reference only, because
    the reference's extra arcs give its destination a cheaper continuation.  Returns
    (#reference arcs beyond the cutoff, #created by both but pruned by the order-free one).
    """
    created = {tuple(r) for r in np.asarray(created_rows).tolist()}
    ref = [tuple(r) for r in np.asarray(ref_rows).tolist()]
    alpha = lattice_forward_costs(ref, start_state)
    beyond = 0
    for r in ref:
        cur = alpha[(int(r[0]), int(r[1]))]
        eps = int(r[0]) == int(r[2])
        fwd = np.float32(cur + np.float32(r[6])) if eps else \
            np.float32(np.float32(cur + np.float32(r[7])) + np.float32(r[6]))
        cutoff = np.float32(np.float32(frame_best[int(r[2])]) + np.float32(beam))
        assert (fwd < cutoff) == (r in created), (tag, r, float(fwd), float(cutoff))
        beyond += not (fwd < cutoff)
    return beyond
