#!/usr/bin/env python3
"""bench.py -- throughput of the token-passing search on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch of synthetic utterances whose
log-prob matrices are already resident in HBM: InitDecoding, AdvanceDecoding
over all frames (one persistent kernel launch), GetBestPath traceback and the
read-back of the 1-best label sequences (N > 1: plus the gather of every rank's
results to rank 0).  Prints ONE JSON line (rank 0).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
HLG-like graph (~5e6 states / ~1.3e7 arcs, seed 2), beam 15, batch 128 per GPU,
T = 250 decoder frames (10 s of audio at 40 ms/frame).  Audio frames = decoder
frames x 4 (subsampling), as in BASELINE.md.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kaldi-decoder_amd", "python"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
SUBSAMPLING = 4

WORKLOADS = {
    # name: (graph kwargs, decoder options, batch, frames, V)
    "hlg": dict(kind="hlg", graph=dict(V=500, n_words=20000, n_hist=20000, avg_succ=50, seed=2),
                opts=dict(beam=15.0), batch=128, frames=250, V=500,
                desc="FasterDecoder 1-best over synthetic LibriSpeech-like HLG "
                     "(bigram G, 20k words, V=500), beam=15, B=128/GPU, T=250"),
    "hl": dict(kind="hl", graph=dict(V=500, n_words=20000, seed=1),
               opts=dict(beam=16.0, max_active=3000), batch=64, frames=250, V=500,
               desc="FasterDecoder 1-best over synthetic HL (20k-word trie, V=500), beam=16, "
                    "max_active=3000, B=64/GPU, T=250"),
    "small": dict(kind="hlg", graph=dict(V=100, n_words=2000, n_hist=2000, avg_succ=10, seed=2),
                  opts=dict(beam=12.0), batch=16, frames=100, V=100,
                  desc="small HLG-like smoke workload"),
}


def algorithmic_bytes(stats, V):
    """16*A_scan + 8*R + 12*N_in + 16*N_out + 4*V per utterance-frame."""
    s = stats.sum(axis=0)
    frames, n_in, expanded, arcs, _, n_out, eps_arcs, eps_exp = (int(x) for x in s[:8])
    a_scan = arcs + eps_arcs
    r = expanded + eps_exp
    return 16 * a_scan + 8 * r + 12 * n_in + 16 * n_out + 4 * V * frames, dict(
        arcs_scanned_per_frame=a_scan / max(frames, 1),
        tokens_in_per_frame=n_in / max(frames, 1),
        tokens_out_per_frame=n_out / max(frames, 1),
        max_tokens=int(stats[:, 8].max()), spill_frames=int(s[9]))


def profiled_traffic(workload_desc):
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary
    (profiles/rNN_summary.json, written by tools/summarize_profile.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes of this same command); None if there is
    no summary for this workload."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("bench_line_under_profiler", {}).get("config", {}).get("workload") == workload_desc:
            best = (d["hbm_bytes_per_launch"], os.path.basename(path))
    return best


def cpu_baseline(g, lps, opts, gpu_paths, budget_s=20.0):
    """The oracle's reference-order FasterDecoder (policy 0), one thread, on a
    bounded sample of the same utterances; also checks 1-best parity."""
    from oracle import oracle
    og = oracle.Graph(g)
    t_used, frames, n, mism = 0.0, 0, 0, 0
    for i, lp in enumerate(lps):
        d = oracle.FasterDecoder(og, policy=0, **opts)
        t0 = time.perf_counter()
        d.decode(lp)
        p = d.get_best_path()
        t_used += time.perf_counter() - t0
        frames += lp.shape[0]
        n += 1
        gp = gpu_paths[i]
        if p.ilabel.tolist() != gp.ilabel.tolist() or p.olabel.tolist() != gp.olabel.tolist():
            mism += 1
        if t_used > budget_s:
            break
    return dict(value=frames * SUBSAMPLING / t_used, unit="audio_frames/s", cores=1, kind="port",
                sample=f"{n} utterances x {lps[0].shape[0]} frames of the bench batch, "
                       "oracle FasterDecoder policy 0 (reference order), 1 thread"), \
        dict(utts_checked=n, one_best_mismatches=mism)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="hlg", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from kaldi_decoder import cabi, sharded, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    B = args.batch or wl["batch"]
    T, V = wl["frames"], wl["V"]
    t0 = time.time()
    g = synth.hlg_graph(**wl["graph"]) if wl["kind"] == "hlg" else synth.hl_graph(**wl["graph"])
    info = synth.describe(g)
    # this rank's utterances: global ids [rank*B, (rank+1)*B) (weak scaling)
    lps = [synth.make_logprobs(T, V, rank * B + i, peak=12.0, lexicon=g["lexicon"])
           for i in range(B)]
    G = cabi.Graph(g, device=local_rank)
    dev = torch.from_numpy(np.stack(lps)).cuda()
    ptrs = [dev[i].data_ptr() for i in range(B)]
    rows = [T] * B
    bt = cabi.Batch(G, max_batch=B, max_frames=T, **wl["opts"])
    setup_s = time.time() - t0

    def step():
        bt.init_decoding()
        bt.advance_device(ptrs, rows, V, sync=False)
        bt.sync()
        paths = bt.best_paths()
        if world > 1:
            sharded.gather_paths([(p.ilabel, p.olabel) for p in paths], dst=0)
        return paths

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms = []
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        paths = step()
        kernel_ms.append(bt.last_kernel_ms())  # HIP events on the launch stream
    fence()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        stats = bt.stats()  # counters of the last step (reset by InitDecoding)
        alg_bytes, per_frame = algorithmic_bytes(stats, V)
        k_ms = float(np.mean(kernel_ms))
        traffic = profiled_traffic(wl["desc"])
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        frames_total = world * B * T * args.steps
        out = {
            "metric": "decoded audio frames/sec (whole node), LibriSpeech HLG beam=15; "
                      "1-best WER parity",
            "value": frames_total * SUBSAMPLING / elapsed,
            "unit": "audio_frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl["desc"], "graph_states": info["states"],
                       "graph_arcs": info["arcs"], "batch_per_gpu": B, "frames": T,
                       "decoder_frames_per_s": frames_total / elapsed, **per_frame},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": "SearchKernel", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "setup_s": setup_s,
        }
        if not args.no_cpu:
            cb, parity = cpu_baseline(g, lps, wl["opts"], paths)
            out["cpu_baseline"] = cb
            out["parity"] = parity
            out["gpu_over_cpu_1thread"] = out["value"] / cb["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
