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
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
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
