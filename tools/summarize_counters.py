#!/usr/bin/env python3
"""gpurun_out/pmc_<round>_*/ (tools/pmc_counters.sh) -> profiles/<round>_search_counters.json:
per-launch averages over the full-size launches of SearchKernel."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))
import bench  # noqa: E402  (the workload tools/call_timing.py runs: its options go into the note)
OPTS = bench.WORKLOADS["hlg-lattice"]["opts"]
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{rnd}_*", "*counter_collection.csv"))):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "SearchKernel" not in r["Kernel_Name"]:
            continue
        g = r.get("Grid_Size_X") or r.get("Grid_Size")
        w = r.get("Workgroup_Size_X") or r.get("Workgroup_Size")
        if int(g) // int(w) < 256:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
d = {"note": "rocprofv3 --pmc passes (tools/pmc_counters.sh) over tools/call_timing.py 4: calls of 512 "
             f"utterances x 250 frames (one launch each, one in flight, prune_interval {OPTS['prune_interval']}), "
             "averages per launch of SearchKernel<1> (search + periodic pruning + FinalizeDecoding + "
             "GetRawLattice)",
     "counters": out}
if "SQ_INSTS_VALU" in out:
    arcs = 8155.0 * 512 * 250  # arcs scanned per launch (bench line: arcs_scanned_per_frame x utterance-frames)
    d["wave_instructions_per_arc_scanned"] = {k: out[k] / arcs for k in
                                              ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                               "SQ_INSTS_VMEM_WR") if k in out}
def ratio(a, b):
    return out[a] / out[b] if a in out and b in out and out[b] else None


# where a wave's cycles go (SQ_WAVE_CYCLES and the ACTIVE / WAIT counters are in the same unit)
d["derived"] = {
    "issuing (ACTIVE_INST_ANY / WAVE_CYCLES)": ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
    "waiting for anything (WAIT_ANY / WAVE_CYCLES)": ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
    "waiting at s_waitcnt (WAIT_INST_ANY / WAVE_CYCLES)": ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
    "... of which for LDS (WAIT_INST_LDS / WAVE_CYCLES)": ratio("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"),
    "VALU share of wave-cycles": ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"),
    "scalar share": ratio("SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES"),
    "LDS share": ratio("SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"),
    "VMEM share": ratio("SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES"),
    "FLAT share": ratio("SQ_ACTIVE_INST_FLAT", "SQ_WAVE_CYCLES"),
    "MISC share (barriers, sleeps, branches)": ratio("SQ_ACTIVE_INST_MISC", "SQ_WAVE_CYCLES"),
    "VALU lane utilisation (THREAD_CYCLES_VALU / (64 x ACTIVE_INST_VALU))":
        (out["SQ_THREAD_CYCLES_VALU"] / (64.0 * out["SQ_ACTIVE_INST_VALU"])
         if "SQ_THREAD_CYCLES_VALU" in out and out.get("SQ_ACTIVE_INST_VALU") else None),
    "LDS bank-conflict cycles / LDS active cycles": ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
    "average vector-memory instructions in flight x cycles per instruction (INST_LEVEL_VMEM / INSTS_VMEM)":
        ratio("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM"),
    "average LDS instruction latency (INST_LEVEL_LDS / INSTS_LDS)": ratio("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS"),
    "branches per VALU instruction": ratio("SQ_INSTS_BRANCH", "SQ_INSTS_VALU"),
}
if "TCC_HIT_sum" in out and "TCC_REQ_sum" in out:
    d["l2_hit_rate"] = out["TCC_HIT_sum"] / out["TCC_REQ_sum"]
_h = os.path.join(ROOT, "gpurun_out", f"pmc_{rnd}_lib.sha256")
d["library_sha256"] = open(_h).read().strip() if os.path.exists(_h) else None  # (tools/check_profiles.py)
json.dump(d, open(os.path.join(ROOT, "profiles", f"{rnd}_search_counters.json"), "w"), indent=1)
print(json.dumps(d, indent=1))
