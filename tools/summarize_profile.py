#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs that a GPU run left under gpurun_out/ into the
committed summaries under profiles/ (per round).

    python tools/summarize_profile.py r01
    python tools/summarize_profile.py r04 large     (tools/profile_large.sh: the large-LM workload)

Expects gpurun_out/prof_<round>_trace (--kernel-trace --stats),
gpurun_out/prof_<round>_fetch (--pmc FETCH_SIZE) and gpurun_out/prof_<round>_write
(--pmc WRITE_SIZE), each produced by its own rocprofv3 run of
`python3 bench.py --no-cpu` as MI355X_MICROARCH.md prescribes (FETCH_SIZE and
WRITE_SIZE do not fit one pass).  HBM bytes per launch follow that guide's
gfx950 rule: FETCH_SIZE is in KiB and tallies 128-byte read requests at 64
bytes, so reads are 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "SearchKernel"


def one(pattern):
    # rocprofv3 writes <dir>/<prefix>_*.csv, or <dir>/<host>/<pid>_*.csv without -o
    hits = (glob.glob(os.path.join(ROOT, "gpurun_out", pattern)) +
            glob.glob(os.path.join(ROOT, "gpurun_out", pattern.replace("/*/", "/"))))
    if not hits:
        raise SystemExit(f"missing gpurun_out/{pattern}")
    return max(hits, key=os.path.getmtime)  # a directory may hold several runs: newest


def full_launch(row):
    """A launch of a whole call (steps_per_call x B utterances: one workgroup per CU);
    the bench also launches single batches of B utterances for its single-batch figure."""
    g = row.get("Grid_Size_X") or row.get("Grid_Size")
    w = row.get("Workgroup_Size_X") or row.get("Workgroup_Size")
    try:
        return int(g) // int(w) >= 256
    except (TypeError, ValueError, ZeroDivisionError):
        return True


def counter(path, name):
    rows = [r for r in csv.DictReader(open(path))
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name and full_launch(r)]
    vals = [float(r["Counter_Value"]) for r in rows]
    return sum(vals) / len(vals), len(vals)


def lib_hash(name):
    p = os.path.join(ROOT, "gpurun_out", name)
    return open(p).read().strip() if os.path.exists(p) else None


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    large = len(sys.argv) > 2 and sys.argv[2] == "large"
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    # (directory names as tools/profile_round.sh and tools/profile_large.sh make them)
    d_trace = f"prof_{rnd}_large" if large else f"prof_{rnd}_trace"
    d_fetch, d_write = (f"prof_{rnd}_large_fetch", f"prof_{rnd}_large_write") if large else (
        f"prof_{rnd}_fetch", f"prof_{rnd}_write")
    name = f"{rnd}_hlg_large" if large else rnd
    stats = one(f"{d_trace}/*/*_kernel_stats.csv")
    trace = one(f"{d_trace}/*/*_kernel_trace.csv")
    shutil.copy(stats, os.path.join(out, f"{name}_kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(trace)) if KERNEL in r["Kernel_Name"]]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6  # noqa: E731
    full = sorted(dur(r) for r in rows if full_launch(r))
    single = sorted(dur(r) for r in rows if not full_launch(r))
    # GPU time by kernel (everything the process launched)
    by = {}
    for r in csv.DictReader(open(trace)):
        by[r["Kernel_Name"][:60]] = by.get(r["Kernel_Name"][:60], 0.0) + dur(r)
    tot = sum(by.values())
    fetch_kb, nf = counter(one(f"{d_fetch}/*/*_counter_collection.csv"), "FETCH_SIZE")
    write_kb, nw = counter(one(f"{d_write}/*/*_counter_collection.csv"), "WRITE_SIZE")
    bench = {}
    bj = os.path.join(ROOT, "gpurun_out", f"{d_trace}.json")
    if os.path.exists(bj):
        txt = open(bj).read().strip().splitlines()
        if txt:
            bench = json.loads(txt[-1])
    summary = {
        "round": rnd,
        "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --headline-only "
                   + ("--workload hlg-large " if large else "") +
                   "(+ one run each with --pmc FETCH_SIZE / --pmc WRITE_SIZE)",
        "kernel": KERNEL,
        "full_launches": len(full),
        "avg_launch_ms": sum(full) / len(full),
        "min_launch_ms": full[0], "max_launch_ms": full[-1],
        "single_batch_launches": len(single),
        "avg_single_batch_launch_ms": sum(single) / len(single) if single else None,
        "gpu_time_share_by_kernel": {k: v / tot for k, v in sorted(by.items(), key=lambda kv: -kv[1])},
        "FETCH_SIZE_KiB_per_launch": fetch_kb,
        "WRITE_SIZE_KiB_per_launch": write_kb,
        "pmc_launches_averaged": [nf, nw],
        "hbm_read_bytes_per_launch": 2 * fetch_kb * 1024,
        "hbm_write_bytes_per_launch": write_kb * 1024,
        "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
        "note": "per launch of a whole call (the bench's steps_per_call x B utterances).  Reads use "
                "the guide's gfx950 x2 FETCH_SIZE correction, calibrated there on wide coalesced 16 "
                "B/lane streams; this kernel's reads are 16 B/lane gathers, so the absolute read "
                "figure is an upper estimate (raw counter is half of it).  Counter passes serialise "
                "the launches; the trace pass does not (launches of calls in flight overlap).",
        "bench_line_under_profiler": bench,
        # sha256[:16] of libkd_hip.so on the box that made the traces (tools/check_profiles.py)
        "library_sha256": lib_hash(f"{d_trace.replace('_trace', '')}_lib.sha256"),
    }
    with open(os.path.join(out, f"{name}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "bench_line_under_profiler"}, indent=1))


if __name__ == "__main__":
    main()
