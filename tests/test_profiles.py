"""The committed profiles of the current round hang together (no GPU): every summary names
the library it is of, and they all name the same one (round 5's
counters were of the build before its last kernel change); the bench line carries what the
measurement contract asks for; the trace's average launch agrees with the bench's own HIP-event
figure."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _latest_round():
    rounds = sorted({os.path.basename(f)[:3] for f in glob.glob(os.path.join(P, "r??_summary.json"))})
    return rounds[-1]


def test_profiles_of_the_round_name_one_library():
    rnd = _latest_round()
    if rnd < "r06":
        return  # (rounds 1-5 did not record it)
    hashes = {}
    for f in sorted(glob.glob(os.path.join(P, f"{rnd}_*.json"))):
        d = json.load(open(f))
        if isinstance(d, dict) and "library_sha256" in d:
            hashes[os.path.basename(f)] = d["library_sha256"]
    assert len(hashes) >= 3, hashes  # summary, large-LM summary, counters (+ min_active 200)
    assert len(set(hashes.values())) == 1 and None not in hashes.values(), hashes


def test_bench_line_of_the_round_carries_the_contract():
    rnd = _latest_round()
    d = json.loads(open(os.path.join(P, f"{rnd}_bench_line.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert d["parity"]["one_best_mismatches"] == 0 and d["vs_baseline"] is None
    s = json.load(open(os.path.join(P, f"{rnd}_summary.json")))
    # the kernel trace of the same command and the bench's HIP events see the same launch
    assert abs(s["avg_launch_ms"] - r["kernel_ms"]) / r["kernel_ms"] < 0.1, (s["avg_launch_ms"], r["kernel_ms"])
    if rnd >= "r06":
        assert d["value_prune_eager"] and d["value_prune_eager"] < d["value"] * 1.02
        assert any("roofline" in x for x in d["side_runs"])
