#!/usr/bin/env python3
"""bench.py -- throughput of the token-passing search on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch of B synthetic utterances whose
log-prob matrices are already resident in HBM.  Steps are handed to the library
--steps-per-call at a time (kd_batch_decode: ONE kernel launch whose persistent
workgroups, one per CU, take utterances from a work queue until none is left).  Default workload (BASELINE.json
configs[2], the configuration the metric is quoted on): LatticeFasterDecoder over
an HLG-like graph, beam 15, lattice-beam 8, batch 128 per GPU, T = 250 decoder
frames (10 s of audio at 40 ms/frame): InitDecoding, AdvanceDecoding over all
frames (one persistent kernel launch), FinalizeDecoding (device-side lattice
pruning) and the export of the pruned raw lattices to host arrays; with N > 1
also the gather of every rank's 1-best label sequences to rank 0.  Prints ONE
JSON line (rank 0).  Audio frames = decoder frames x 4 (subsampling), as in
BASELINE.md.
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

HLG = dict(V=500, n_words=20000, n_hist=20000, avg_succ=50, seed=2)
WORKLOADS = {
    "hlg-lattice": dict(kind="hlg", graph=HLG, lattice=True,
                        opts=dict(beam=15.0, lattice_beam=8.0, min_active=0, prune_interval=25),
                        batch=128, frames=250, V=500,
                        desc="LatticeFasterDecoder over synthetic LibriSpeech-like HLG (bigram G, "
                             "20k words, V=500), beam=15, lattice_beam=8, B=128/GPU, T=250; headline at "
                             "LatticeSimpleDecoderConfig's cutoff (min_active=0, max_active=INT_MAX: the one "
                             "lattice decoder the reference implements), LatticeFasterDecoderConfig's own "
                             "defaults (min_active=200) and max_active=10000 as side figures"),
    # BASELINE configs[4] on one GPU: the same generator at 200 k words -- 52.8 M states,
    # 141 M arcs = 2.26 GB of 16-byte arc records (stresses HBM: the graph is nine times the
    # Infinity Cache), beam 17
    "hlg-large": dict(kind="hlg", graph=dict(V=500, n_words=200000, n_hist=200000, avg_succ=50, seed=2),
                      lattice=True, opts=dict(beam=17.0, lattice_beam=8.0, min_active=0, prune_interval=25),
                      batch=128, frames=250, V=500,
                      desc="LatticeFasterDecoder over synthetic large-LM HLG (bigram G, 200k words, "
                           "141M arcs = 2.26 GB, V=500), beam=17, lattice_beam=8, B=128/GPU, T=250 "
                           "(min_active=0, max_active=INT_MAX)"),
    "hlg-1best": dict(kind="hlg", graph=HLG, lattice=False, opts=dict(beam=15.0),
                      batch=128, frames=250, V=500,
                      desc="FasterDecoder 1-best over synthetic LibriSpeech-like HLG "
                           "(bigram G, 20k words, V=500), beam=15, B=128/GPU, T=250"),
    "hl": dict(kind="hl", graph=dict(V=500, n_words=20000, seed=1), lattice=False,
               opts=dict(beam=16.0, max_active=3000), batch=64, frames=250, V=500,
               desc="FasterDecoder 1-best over synthetic HL (20k-word trie, V=500), beam=16, "
                    "max_active=3000, B=64/GPU, T=250"),
    "small": dict(kind="hlg", graph=dict(V=100, n_words=2000, n_hist=2000, avg_succ=10, seed=2),
                  lattice=True, opts=dict(beam=12.0, lattice_beam=6.0, min_active=0),
                  batch=16, frames=100, V=100, desc="small HLG-like lattice smoke workload"),
}


def algorithmic_bytes(stats, V):
    """16*A_scan + 8*R + 12*N_in + 16*N_out + 4*V (+ 20*L) per utterance-frame,
    with A_scan = the arcs the kernel READ: since round 5 a token whose cost leaves room for the
    frame's likeliest label only looks its arcs of that label up in the label index instead of
    reading all of them (KD_STAT_ARCS_SKIPPED); the index nodes it reads (16 bytes each) count as
    arcs do.  `arcs_scanned_per_frame` keeps its meaning -- the arcs the reference's loop visits,
    equal to the oracle's count -- and the third value returned is the figure by rounds 1-4's
    definition (every scanned arc read), for comparison across rounds."""
    s = stats.sum(axis=0)
    frames, n_in, expanded, arcs, _, n_out, eps_arcs, eps_exp = (int(x) for x in s[:8])
    links = int(s[10])
    skipped, probes = (int(s[12]), int(s[13])) if len(s) > 13 else (0, 0)
    a_scan = arcs + eps_arcs
    a_read = a_scan - skipped
    r = expanded + eps_exp
    rest = 8 * r + 12 * n_in + 16 * n_out + 4 * V * frames + 20 * links
    return (16 * (a_read + probes) + rest), dict(
        arcs_scanned_per_frame=a_scan / max(frames, 1),
        arcs_read_per_frame=a_read / max(frames, 1),
        index_nodes_read_per_frame=probes / max(frames, 1),
        arcs_read_per_link=a_read / max(links, 1),
        tokens_in_per_frame=n_in / max(frames, 1),
        tokens_out_per_frame=n_out / max(frames, 1),
        links_per_frame=links / max(frames, 1),
        max_tokens=int(stats[:, 8].max()), spill_frames=int(s[9])), (16 * a_scan + rest)


def profiled_traffic(workload_desc, utt_frames_per_launch, prune_interval, min_active=0):
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary
    (profiles/rNN_summary.json, written by tools/summarize_profile.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes of this same command); None if there is
    no summary for this workload and these options.  The counters are per launch: a
    summary taken on launches of another size (--steps-per-call) is scaled by the
    utterance-frames per launch, which is what the traffic is proportional to, and the
    source says so."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        line = d.get("bench_line_under_profiler", {})
        cfg = line.get("config", {})
        if cfg.get("workload") != workload_desc or cfg.get("prune_interval") != prune_interval:
            continue
        if (cfg.get("min_active") or 0) != (min_active or 0):  # (rNN_min200: another configuration)
            continue
        if "hbm_bytes_per_launch" not in d:  # (a trace-only summary: no counter passes)
            continue
        theirs = line.get("roofline", {}).get("utterance_frames_per_launch")
        src = os.path.basename(path)
        if theirs and theirs != utt_frames_per_launch:
            best = (d["hbm_bytes_per_launch"] * utt_frames_per_launch / theirs,
                    f"{src} (measured on launches of {theirs} utterance-frames, scaled to {utt_frames_per_launch})")
        else:
            best = (d["hbm_bytes_per_launch"], src)
    return best


def lattice_best_path(lat):
    """Tropical shortest path over a cabi.RawLattice (numpy): relax all arcs until
    stable (the lattice is acyclic and shallow inside a frame)."""
    n = lat.num_states
    if n == 0:
        return None
    cost = np.full(n, np.inf)
    cost[lat.start] = 0.0
    w = lat.graph_cost.astype(np.float64) + lat.ac_cost.astype(np.float64)
    for _ in range(4 * int(lat.state_frame.max()) + 64):
        new = cost.copy()
        np.minimum.at(new, lat.dst, cost[lat.src] + w)
        if not (new < cost).any():
            break
        cost = new
    cand = cost[lat.src] + w
    hit = np.isfinite(cand) & (cand == cost[lat.dst])
    back = np.full(n, -1, np.int64)
    idx = np.nonzero(hit)[0][::-1]
    back[lat.dst[idx]] = idx
    fin = cost[lat.final_state] + lat.final_graph
    if not len(fin) or not np.isfinite(fin).any():
        return None
    s = int(lat.final_state[int(np.argmin(fin))])
    il, ol = [], []
    while s != lat.start:
        a = int(back[s])
        il.append(int(lat.ilabel[a]))
        ol.append(int(lat.olabel[a]))
        s = int(lat.src[a])
    return [x for x in il[::-1] if x], [x for x in ol[::-1] if x]


def usable_cores():
    """Host cores this process may actually use: the affinity mask, cut by the container's
    CPU quota (cgroup cpu.max / cfs_quota_us) -- the GPU box reports 256 logical cores and
    grants 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(g, lps, wl, gpu_out, pool=None, budget_s=20.0):
    """The oracle's reference-order restatement (policy 0), one thread, on a bounded
    sample of the same utterances; also checks 1-best parity with the GPU output."""
    from oracle import oracle
    og = _CPU.get("og") or oracle.Graph(g)
    t_used, frames, n, mism, not_subset = 0.0, 0, 0, 0, 0
    for i, lp in enumerate(lps):
        t0 = time.perf_counter()
        if wl["lattice"]:
            d = oracle.LatticeSimpleDecoder(og, beam=wl["opts"]["beam"],
                                            lattice_beam=wl["opts"]["lattice_beam"])
            d.decode(lp)
            raw = d.get_raw_lattice()
            p = d.get_best_path()
            t_used += time.perf_counter() - t0
            best = lattice_best_path(gpu_out[i])
            if best is None or best[0] != p.isyms.tolist() or best[1] != p.osyms.tolist():
                mism += 1
            if not gpu_out[i].arc_set() <= raw.arc_set():
                not_subset += 1
        else:
            d = oracle.FasterDecoder(og, policy=0, **wl["opts"])
            d.decode(lp)
            p = d.get_best_path()
            t_used += time.perf_counter() - t0
            gp = gpu_out[i]
            if p.ilabel.tolist() != gp.ilabel.tolist() or p.olabel.tolist() != gp.olabel.tolist():
                mism += 1
        frames += lp.shape[0]
        n += 1
        if t_used > budget_s:
            break
    what = ("LatticeSimpleDecoder (reference order) incl. FinalizeDecoding, GetRawLattice and "
            "shortest path" if wl["lattice"] else "FasterDecoder policy 0 (reference order)")
    par = dict(utts_checked=n, one_best_mismatches=mism)
    if wl["lattice"]:
        par["gpu_lattice_not_subset_of_reference"] = not_subset
    one = frames * SUBSAMPLING / t_used
    res = dict(value=one, unit="audio_frames/s", cores=1, kind="port",
               sample=f"{n} utterances x {lps[0].shape[0]} frames of the bench batch, oracle "
                      f"{what}, 1 thread")
    if pool is not None:
        # the same decode on every host core: one PROCESS per core (forked before the GPU
        # was touched; each builds its own decoder objects over the inherited graph arrays,
        # so neither the GIL nor one allocator is shared), every process decoding the same
        # number of utterances of the sample
        procs, per_proc = pool._processes, 8
        jobs = [[(k * per_proc + j) % n for j in range(per_proc)] for k in range(procs)]
        # (each job first decodes one utterance untimed: a forked worker's first pass over
        # the inherited graph is a storm of minor page faults; the clock -- CLOCK_MONOTONIC,
        # common to the processes -- runs from the first timed start to the last end)
        done = pool.map(_cpu_worker, jobs, chunksize=1)
        frames_all = sum(f for f, _, _ in done)
        t_all = max(e for _, _, e in done) - min(b for _, b, _ in done)
        allc = frames_all * SUBSAMPLING / t_all
        res["all_cores"] = dict(value=allc, unit="audio_frames/s", cores=procs,
                                scaling_efficiency=allc / (procs * one),
                                host_logical_cores=os.cpu_count(),
                                sample=f"{procs} processes (the cores the container grants) x "
                                       f"{per_proc} utterances of the same sample")
    return res, par


_CPU = {}


def _cpu_worker(idx):
    """(in a forked worker) decodes the sample's utterances `idx` -- the first of them once
    more beforehand, untimed; returns (decoder frames, start, end) of the timed part."""
    from oracle import oracle
    wl, lps, og = _CPU["wl"], _CPU["lps"], _CPU["og"]

    def decode(lp):
        if wl["lattice"]:
            d = oracle.LatticeSimpleDecoder(og, beam=wl["opts"]["beam"],
                                            lattice_beam=wl["opts"]["lattice_beam"])
            d.decode(lp)
            d.get_raw_lattice()
            d.get_best_path()
        else:
            d = oracle.FasterDecoder(og, policy=0, **wl["opts"])
            d.decode(lp)
            d.get_best_path()
        return lp.shape[0]

    if idx:
        decode(lps[idx[0]])
    t0 = time.perf_counter()
    frames = sum(decode(lps[i]) for i in idx)
    return frames, t0, time.perf_counter()


def synth_graph(wl):
    from kaldi_decoder import synth
    return synth.hlg_graph(**wl["graph"]) if wl["kind"] == "hlg" else synth.hl_graph(**wl["graph"])


def shared_graph(wl, name, rank, world, job, timeout_s=3600.0):
    """The workload's graph for this rank.  One rank: synthesised here.  Several: rank 0
    synthesises it ONCE and hands it to the others through a file in /dev/shm -- this package's
    own OpenFst `const` file, read back with Fst.read (the large-LM graph takes a minute to
    synthesise on one core; eight ranks doing that at once on the cores a container grants would
    be most of the job's run time) -- plus the lexicon the log-probs are drawn from.  The file is
    written under another name and renamed (the readers poll for it); rank 0 removes it when
    every rank has passed the first barrier (main())."""
    if world <= 1:
        return synth_graph(wl), None
    import kaldi_decoder as kd
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"kd_bench_{name}_{job}")
    path, lex_path = base + ".fst", base + ".lex.npz"
    if rank == 0:
        g = synth_graph(wl)
        kd.fst_from_dict(g).write(path + ".tmp", "const")
        toks, lens = g["lexicon"]
        np.savez(lex_path + ".tmp.npz", toks=toks, lens=lens)
        os.replace(lex_path + ".tmp.npz", lex_path)
        os.replace(path + ".tmp", path)  # (last: its presence says both files are complete)
        return g, (path, lex_path)
    t_end = time.time() + timeout_s
    while not os.path.exists(path):
        if time.time() > t_end:
            raise SystemExit(f"rank {rank}: the graph file {path} never appeared")
        time.sleep(0.2)
    f = kd.Fst.read(path)
    row_ptr, ilabel, olabel, weight, nextstate, final = f.arrays()
    lex = np.load(lex_path)
    g = {"num_states": int(f.num_states), "start": int(f.start), "row_ptr": row_ptr, "ilabel": ilabel,
         "olabel": olabel, "weight": weight, "nextstate": nextstate, "final": final,
         "lexicon": (lex["toks"], lex["lens"])}
    return g, None


def spawn_ranks(n):
    """`bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): start the N
    ranks as a CHILD `python -m torch.distributed.run` of this same command line and exit with
    its code.  Nothing here has imported torch or made a HIP call -- a process that has touched
    the GPU must never be replaced by another (no exec), and this one has not and does not."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this host driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    # stdout is inherited: rank 0's JSON line is this process's JSON line
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=56)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="hlg-lattice", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch (utterances per step)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the side measurements whose launches are of another shape than the "
                         "headline's (prune_interval off, host input): what tools/profile_round.sh "
                         "runs under rocprofv3, so that the trace's average launch is the headline's")
    ap.add_argument("--no-large-lm", action="store_true",
                    help="skip the large-LM workload (BASELINE configs[4] on one GPU) that the default "
                         "invocation runs as a child process after the headline and embeds as `large_lm`")
    ap.add_argument("--steps-per-call", type=int, default=8,
                    help="steps (batches of B utterances) handed to the library in one "
                         "kd_batch_decode call: one kernel launch whose persistent workgroups take "
                         "all their utterances from one work queue")
    ap.add_argument("--inflight", type=int, default=2,
                    help="library calls in flight, each on a batch object and stream of its own; "
                         "one host thread enqueues them (the decode call is asynchronous) and "
                         "collects them in order")
    ap.add_argument("--min-seconds", type=float, default=6.0,
                    help="repeat the K timed steps until the timed region lasts this long")
    ap.add_argument("--prune-interval", type=int, default=0,
                    help="lattice workloads: PruneActiveTokens every this many frames (0: the "
                         "workload's setting)")
    ap.add_argument("--min-active", type=int, default=-1,
                    help="lattice workloads: the headline at this min_active instead of the workload's "
                         "(200 = LatticeFasterDecoderConfig's default: what tools/profile_round.sh rNN_min200 "
                         "profiles; the default bench line carries that configuration as a side run)")
    ap.add_argument("--search-only", action="store_true",
                    help="diagnostic: time InitDecoding + AdvanceDecoding only (not a bench line)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the result gather: nccl (= RCCL, the product) "
                         "or gloo (tests: the exchange crosses host memory)")
    ap.add_argument("--one-device", action="store_true",
                    help="test only: every rank uses GPU 0 (needs --backend gloo: RCCL refuses two "
                         "ranks on one device) -- the world > 1 code path on a one-GPU box")
    ap.add_argument("--verify-gather", action="store_true",
                    help="after the timed region every rank sends one more call's lattices through "
                         "the gather; rank 0 decodes the other ranks' utterances itself and compares "
                         "what arrived with that, array for array (`gather_verified` in the line)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    if args.one_device and args.backend != "gloo":
        raise SystemExit("--one-device needs --backend gloo")

    from kaldi_decoder import synth
    rank = int(os.environ.get("RANK", "0"))
    wl = WORKLOADS[args.workload]
    t0 = time.time()
    g, shared_files = shared_graph(wl, args.workload, rank, int(os.environ.get("WORLD_SIZE", "1")),
                                   os.environ.get("MASTER_PORT", str(os.getppid())))
    # The CPU leg's worker processes are forked HERE, before this process loads the HIP
    # runtime: they inherit the graph and the sample's log-probs and never touch the GPU.
    cpu_pool = None
    if rank == 0 and not args.no_cpu and not args.search_only:
        import multiprocessing
        B0, T0, V0 = args.batch or wl["batch"], wl["frames"], wl["V"]
        n_sample = max(1, args.steps_per_call) * B0
        from oracle import oracle as _oracle
        # (the oracle's graph object is built once, before the fork: the workers share its
        # pages instead of holding a copy of a multi-GB graph each)
        _CPU.update(g=g, wl=wl, og=_oracle.Graph(g),
                    lps=[synth.make_logprobs(T0, V0, u, peak=12.0, lexicon=g["lexicon"])
                         for u in range(min(n_sample, 512))])
        cpu_pool = multiprocessing.get_context("fork").Pool(usable_cores())

    import torch
    import torch.distributed as dist
    from kaldi_decoder import cabi, sharded

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)  # (before any other HIP call of this process)
    if world > 1 and args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif world > 1:
        dist.init_process_group("gloo")
    # the few control words the ranks exchange (repeat count, times) live where the backend
    # wants them: device memory for RCCL, host memory for gloo
    ctl_dev = "cuda" if args.backend == "nccl" else "cpu"

    B = args.batch or wl["batch"]
    T, V = wl["frames"], wl["V"]
    G = max(1, args.steps_per_call)
    n_fly = max(1, args.inflight)
    info = synth.describe(g)
    # this rank's utterances (weak scaling): every call in flight decodes its OWN G * B
    # utterances -- global ids (rank * n_fly + slot) * G * B + i -- so no two launches walk
    # the same arcs or share log-prob rows in L2 / the Infinity Cache
    per_slot = G * B
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(32, usable_cores())) as ex:
        all_lps = list(ex.map(lambda u: synth.make_logprobs(T, V, u, peak=12.0, lexicon=g["lexicon"]),
                              range(rank * n_fly * per_slot, (rank + 1) * n_fly * per_slot)))
    Gr = cabi.Graph(g, device=local_rank)
    devs = [torch.from_numpy(np.stack(all_lps[s * per_slot:(s + 1) * per_slot])).cuda()
            for s in range(n_fly)]
    ptrs_of = [[d[i].data_ptr() for i in range(per_slot)] for d in devs]
    streams = [torch.cuda.Stream() for _ in range(n_fly)]
    cls = cabi.LatticeBatch if wl["lattice"] else cabi.Batch
    opts = dict(wl["opts"])
    if wl["lattice"] and args.prune_interval:
        opts["prune_interval"] = args.prune_interval
    if wl["lattice"] and args.min_active >= 0:
        opts["min_active"] = args.min_active
    bts = [cls(Gr, max_batch=per_slot, max_frames=T, stream=st.cuda_stream, **opts)
           for st in streams]
    setup_s = time.time() - t0

    def enqueue(slot, n_steps):
        """One library call: n_steps batches of B utterances, one kernel launch; returns at
        once (kd_batch_decode_async)."""
        n = n_steps * B
        bts[slot].decode_device(ptrs_of[slot][:n], [T] * n, V, finalize=not args.search_only,
                                lattice=not args.search_only, best_path=not args.search_only,
                                sync=False)

    def collect(slot):
        """Waits for the slot's call and takes its results to the host."""
        bt = bts[slot]
        bt.sync()
        k_ms = bt.last_kernel_ms()  # HIP events around the launch, on its stream
        if args.search_only:
            return None, k_ms
        out = bt.raw_lattices() if wl["lattice"] else bt.best_paths()
        return out, k_ms

    def call(slot, n_steps):
        enqueue(slot, n_steps)
        return collect(slot)

    # The only exchange between ranks: finished lattices go to rank 0 (north_star: "RCCL over
    # xGMI only for gather of final lattices").  The batches export into device buffers lent
    # to them (kd_batch_set_output_buffer), so a call's result is handed to RCCL as device
    # memory -- no host packing, no upload -- and the exchange (an all_gather of sizes, then
    # grouped point-to-point receives of exactly those sizes on rank 0) runs on a WORKER thread
    # and a stream of its own (sharded.ResultGatherer): the thread that enqueues the decode
    # calls never waits for another rank.  With one rank the same path runs and the exchange
    # is empty.  (1-best workloads: the label sequences, packed on the host -- a few KB.)
    # (a process group of its own: the worker's collectives never interleave with the
    # barriers and reductions the decode thread issues on the default group)
    gather_group = dist.new_group() if world > 1 else None
    gatherer = sharded.ResultGatherer(dst=0, group=gather_group, keep=False)
    out_bufs = None
    if wl["lattice"] and not args.search_only:
        out_bufs = [torch.zeros(per_slot << 13, dtype=torch.int64, device="cuda") for _ in range(n_fly)]
        for bt_, buf_ in zip(bts, out_bufs):
            bt_.set_output_buffer(buf_)
        # a lent buffer does not grow (the library's own pool does): size it on one untimed
        # call per slot -- doubled until the call's lattices fit, with room to spare
        for slot_ in range(n_fly):
            while True:
                try:
                    bts[slot_].decode_device(ptrs_of[slot_], [T] * per_slot, V)
                    used_, _ = bts[slot_].raw_lattice_table()
                    if 2 * used_ <= out_bufs[slot_].numel():
                        break
                except RuntimeError as e:
                    if "output buffer" not in str(e):
                        raise
                out_bufs[slot_] = torch.zeros(2 * out_bufs[slot_].numel(), dtype=torch.int64, device="cuda")
                bts[slot_].set_output_buffer(out_bufs[slot_])

    def deliver(out, slot):
        if out is None:
            return
        if out_bufs is not None:
            used, table = bts[slot].raw_lattice_table()
            with torch.cuda.stream(streams[slot]):  # (ahead of the slot's next launch, which rewrites the buffer)
                payload = sharded.pool_payload(out_bufs[slot], used, table)
                ready = torch.cuda.Event()
                ready.record(streams[slot])
            gatherer.submit(payload, ready)
        elif world > 1:
            sharded.gather_paths([(p.ilabel, p.olabel) for p in out], dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        dist.barrier()  # (every rank has read the graph file)
        if shared_files:
            for f_ in shared_files:
                try:
                    os.remove(f_)
                except OSError:
                    pass

    # A run of K steps is ceil(K / G) calls (the last one may be short), kept n_fly deep in
    # flight by ONE host thread: the library's decode call only enqueues, so the thread
    # hands over call i + n_fly before it waits for call i -- while it copies one call's
    # lattices out, the workgroups of the next calls keep every CU busy (a launch's
    # workgroups leave as its work queue runs dry and the next launch's take their CUs).
    def run_steps(n):
        sizes = [G] * (n // G) + ([n % G] if n % G else [])
        res = []
        for i in range(min(n_fly, len(sizes))):
            enqueue(i % n_fly, sizes[i])
        for i in range(len(sizes)):
            r = collect(i % n_fly)
            deliver(r[0], i % n_fly)  # (before the slot's next launch rewrites its output buffer)
            if i + n_fly < len(sizes):
                enqueue(i % n_fly, sizes[i + n_fly])
            res.append(r + (sizes[i], i % n_fly))
        gatherer.flush()  # (the exchange of every call of this run has completed)
        return res

    def verify_gather():
        """One more call per rank through the gather, kept on rank 0, which then decodes every
        other rank's utterances of that call ITSELF and compares what arrived with its own
        export, array for array: the exchange hands over the right bytes of the right rank."""
        if out_bufs is None:
            return None
        vg = sharded.ResultGatherer(dst=0, group=gather_group, keep=True)
        call(0, G)
        used, table = bts[0].raw_lattice_table()
        with torch.cuda.stream(streams[0]):
            payload = sharded.pool_payload(out_bufs[0], used, table)
            ready = torch.cuda.Event()
            ready.record(streams[0])
        vg.submit(payload, ready)
        got = vg.close()
        if rank != 0:
            return None
        fields = ("src", "dst", "ilabel", "olabel", "graph_cost", "ac_cost", "state_frame",
                  "state_graph", "final_state", "final_graph")
        bad = n_utts = 0
        for r in range(world):
            theirs = sharded.unpack_pool(got[0][r])
            if r == 0:
                mine = bts[0].raw_lattices()
            else:
                lps_r = [synth.make_logprobs(T, V, u, peak=12.0, lexicon=g["lexicon"])
                         for u in range(r * n_fly * per_slot, r * n_fly * per_slot + per_slot)]
                dev_r = torch.from_numpy(np.stack(lps_r)).cuda()
                bts[0].decode_device([dev_r[i].data_ptr() for i in range(per_slot)], [T] * per_slot, V)
                mine = bts[0].raw_lattices()
            assert len(theirs) == len(mine) == per_slot
            for i in range(per_slot):
                a, b = theirs[i], mine[i]
                n_utts += 1
                if a.num_states != b.num_states or a.start != b.start or not all(
                        np.array_equal(getattr(a, f), getattr(b, f)) for f in fields):
                    bad += 1
        return dict(ranks=world, utterances=n_utts, mismatches=bad)

    # every slot's batch gets at least one untimed full call when warmup is requested
    # (first launches fault in the pools and size the read-back buffers)
    run_steps(max(args.warmup, 2 * n_fly * G if args.warmup else 0))
    fence()
    # The timed region is the K requested steps, repeated until it lasts at least
    # --min-seconds (a 20-step region is 90 ms: too short for any outside clock or
    # utilisation sampler to corroborate).  Every repeat is K steps; "steps" in the
    # output line is the number actually timed.
    t_probe = time.perf_counter()
    run_steps(args.steps)
    fence()
    t_probe = time.perf_counter() - t_probe
    repeats = max(1, int(np.ceil(args.min_seconds / max(t_probe, 1e-6))))
    if world > 1:  # every rank must run the same number of collectives
        tr = torch.tensor([repeats], dtype=torch.int64, device=ctl_dev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        repeats = int(tr.item())
    steps_timed = args.steps * repeats
    fence()
    t_start = time.perf_counter()
    results = run_steps(steps_timed)
    fence()
    elapsed = time.perf_counter() - t_start
    launches = [(k, n) for _, k, n, _ in results]
    # one more full call (untimed) whose output the CPU leg checks and whose counters
    # give the algorithmic bytes of a launch
    n_last = G
    out, _ = call(0, G)
    bt = bts[0]
    lps = all_lps[:per_slot]
    stats = bt.stats() if not args.search_only else None
    # The workloads run at the reference's default prune_interval (25): PruneActiveTokens
    # every 25 frames plus the collector that keeps an utterance's footprint at its pruned
    # lattice and a window of recent frames.  The same pipeline with the interval above the
    # utterance length (no periodic sweeps, no collector; the lattices are the same either
    # way -- the knob trades time for memory, and a 10 s utterance needs 5 MB) is reported
    # next to the headline as `value_prune_interval_off`.
    def side_value(extra, label, lazy_prune=None):
        """The headline's pipeline with other decoder options (batch objects of its own, the
        same streams, utterances and lent buffers): audio frames/s, and the counters of one call."""
        nonlocal bts
        keep = bts
        try:
            bts = [cls(Gr, max_batch=per_slot, max_frames=T, stream=st.cuda_stream,
                       **{**wl["opts"], **extra}) for st in streams]
            if out_bufs is not None:
                for bt_, buf_ in zip(bts, out_bufs):
                    bt_.set_output_buffer(buf_)
            if lazy_prune is not None:
                for bt_ in bts:
                    bt_.set_lazy_prune(lazy_prune)
            run_steps(2 * n_fly * G)
            fence()
            # (as long a timed region as the headline's: the pipeline's fill and drain -- one call's
            # duration with fewer than n_fly calls in flight -- must not weigh more here than there)
            n_side = max(args.steps, int(steps_timed * 0.6) // (n_fly * G) * (n_fly * G), n_fly * G)
            t_side = time.perf_counter()
            run_steps(n_side)
            fence()
            v = world * B * T * n_side * SUBSAMPLING / (time.perf_counter() - t_side)
            st_ = bts[0].stats()
            sm = st_.sum(axis=0)
            # (the call's own launch, by the library's HIP events, and the same roofline
            # arithmetic as the headline's -- VERDICT round 5, item 5)
            k_ms_ = float(bts[0].last_kernel_ms())
            ab_, pf_, _ = algorithmic_bytes(st_, V)
            return dict(value=v, unit="audio_frames/s", options=label,
                        tokens_in_per_frame=float(sm[1]) / max(float(sm[0]), 1.0),
                        tokens_expanded_per_frame=float(sm[2]) / max(float(sm[0]), 1.0),
                        max_tokens=int(st_[:, 8].max()),
                        kernel_ms=k_ms_,
                        roofline=dict(bound="hbm", achieved=ab_ / (k_ms_ * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                                      unit="GB/s", frac=ab_ / (k_ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      algorithmic_bytes_per_launch=ab_),
                        arcs_scanned_per_frame=pf_["arcs_scanned_per_frame"],
                        arcs_read_per_frame=pf_["arcs_read_per_frame"],
                        links_per_frame=pf_["links_per_frame"], spill_frames=pf_["spill_frames"])
        finally:
            bts = keep

    # The workloads run at the reference's default prune_interval (25): PruneActiveTokens
    # every 25 frames plus the collector that keeps an utterance's footprint at its pruned
    # lattice and a window of recent frames.  The same pipeline with the interval above the
    # utterance length (no periodic sweeps, no collector; the lattices are the same either
    # way -- the knob trades time for memory, and a 10 s utterance needs 5 MB) is reported
    # next to the headline as `value_prune_interval_off`.
    value_pi_off = side_min200 = side_max10k = side_eager = None
    if wl["lattice"] and not args.search_only and not args.prune_interval and not args.headline_only:
        value_pi_off = side_value({"prune_interval": 1 << 20}, "prune_interval off")["value"]
        # The headline runs at the reference's prune_interval (25) with the library's default
        # policy: PruneActiveTokens is OWED when its frame comes and run when it shows -- an
        # export of the un-finalized lattice, a pool that is full (kd_batch_set_lazy_prune,
        # DESIGN section 4): a Decode() of a 10 s utterance in the default pools never pays.  The
        # same pipeline with every prune run when its frame comes, like the reference's loop
        # (lattice-simple-decoder.cc:57-59; rounds 1-5's headline):
        side_eager = side_value({}, "PruneActiveTokens run every prune_interval frames (kd_batch_set_lazy_prune(0))",
                                lazy_prune=0)
        # ... and at the options of the decoder the workload is named after: the reference
        # DECLARES LatticeFasterDecoder (config only: lattice-faster-decoder.h:48-55 defaults
        # min_active = 200, max_active = INT_MAX) and implements LatticeSimpleDecoder, whose
        # cutoff (no min / max_active) is the headline's; SURVEY 8(d) config 3 also asks for
        # max_active = 10000
        side_min200 = side_value({"min_active": 200}, "LatticeFasterDecoderConfig defaults: min_active=200, max_active=INT_MAX")
        side_max10k = side_value({"min_active": 200, "max_active": 10000}, "min_active=200, max_active=10000")
    # a full call alone on the GPU (no other launch sharing the CUs): what the launch's
    # own duration is when nothing stretches it
    alone_ms = []
    for _ in range(3):
        fence()
        alone_ms.append(call(0, G)[1])
    alone_ms = float(np.median(alone_ms))
    # The PCIe-inclusive rate: the same call fed host matrices (numpy, pageable), which the
    # library stages to the device first -- never the headline (the boundary hands over
    # device pointers when the acoustic model runs on the GPU), reported for comparison.
    host_input = None
    if not args.search_only and world == 1 and not args.headline_only:
        # one call at a time (what round 2 reported) ...
        ts = []
        for _ in range(3):
            fence()
            t1 = time.perf_counter()
            bts[0].decode_host(all_lps[:per_slot])
            (bts[0].raw_lattices() if wl["lattice"] else bts[0].best_paths())
            ts.append(time.perf_counter() - t1)
        serial_s = float(np.median(ts))
        # ... and the same pipeline as the headline's, fed host matrices: the calls are kept
        # n_fly deep (kd_batch_decode_host_async), so one batch's rows cross PCIe while the
        # other batches' launches run
        def host_round(n_calls):
            for i in range(min(n_fly, n_calls)):
                bts[i % n_fly].decode_host(all_lps[(i % n_fly) * per_slot:(i % n_fly + 1) * per_slot], sync=False)
            for i in range(n_calls):
                bt_i = bts[i % n_fly]
                bt_i.sync()
                (bt_i.raw_lattices() if wl["lattice"] else bt_i.best_paths())
                if i + n_fly < n_calls:
                    bt_i.decode_host(all_lps[(i % n_fly) * per_slot:(i % n_fly + 1) * per_slot], sync=False)
        host_round(n_fly)
        fence()
        n_calls = 3 * n_fly
        t1 = time.perf_counter()
        host_round(n_calls)
        fence()
        piped_s = (time.perf_counter() - t1) / n_calls
        host_input = dict(value=per_slot * T * SUBSAMPLING / piped_s, unit="audio_frames/s",
                          ms_per_call=piped_s * 1e3,
                          one_call_at_a_time=dict(value=per_slot * T * SUBSAMPLING / serial_s,
                                                  ms_per_call=serial_s * 1e3),
                          note=f"calls of {per_slot} utterances, {n_fly} in flight, log-probs in pageable "
                               f"host memory ({per_slot * T * V * 4 / 2**20:.0f} MB per call)")
    # one batch of B utterances alone on the GPU, nothing else in flight: what a caller
    # who submits a single batch gets
    single = []
    for _ in range(5):
        fence()
        t1 = time.perf_counter()
        _, k1 = call(0, 1)
        torch.cuda.synchronize()
        single.append(((time.perf_counter() - t1) * 1e3, k1))
    single_batch_ms = float(np.median([a for a, _ in single]))
    single_kernel_ms = float(np.median([b for _, b in single]))
    per_rank_elapsed = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        ts_all = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts_all, t)
        per_rank_elapsed = [float(x.item()) for x in ts_all]
        elapsed = max(per_rank_elapsed)  # (the job is as slow as its slowest rank)
    gather_ms = gatherer.mean_ms()
    verified = verify_gather() if args.verify_gather else None

    # BASELINE configs[4] on this GPU, in front of the driver too: the one workload whose graph
    # (2.26 GB of arc records) does not fit the Infinity Cache.  A CHILD process of its own
    # (another graph, other pools), started when this process's GPU work is done.
    large_lm = None
    if (rank == 0 and world == 1 and args.workload == "hlg-lattice" and not args.search_only
            and not args.headline_only and not args.no_large_lm and not args.batch):
        import subprocess
        t_l = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "hlg-large",
                                "--no-cpu", "--headline-only", "--min-seconds", "5", "--steps", str(n_fly * G)],
                               capture_output=True, text=True, timeout=1500)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            large_lm = dict(value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"],
                            frac=d["roofline"]["frac"], kernel_ms=d["roofline"]["kernel_ms"],
                            launch_alone_frac=d["roofline"]["launch_alone"]["frac"],
                            traffic=d["roofline"]["traffic"], traffic_source=d["roofline"]["traffic_source"],
                            algorithmic_bytes_per_launch=d["roofline"]["algorithmic_bytes_per_launch"],
                            workload=d["config"]["workload"], graph_arcs=d["config"]["graph_arcs"],
                            arcs_scanned_per_frame=d["config"]["arcs_scanned_per_frame"],
                            links_per_frame=d["config"]["links_per_frame"],
                            steps_timed=d["steps_timed"], setup_s=d["setup_s"], wall_s=time.time() - t_l)
            for k in ("arcs_read_per_frame", "index_nodes_read_per_frame", "arcs_read_per_link"):
                if k in d["config"]:
                    large_lm[k] = d["config"][k]
        except Exception as e:  # (reported, not fatal: the headline stands on its own)
            large_lm = dict(error=repr(e)[:300])

    if rank == 0 and args.search_only:
        print(json.dumps({"diagnostic": "search only", "ms_per_step": elapsed / steps_timed * 1e3,
                          "launch_ms": float(np.mean([k for k, n in launches if n == G] or [0])),
                          "steps_per_call": G, "inflight": n_fly}))
    elif rank == 0:
        alg_bytes, per_frame, alg_bytes_all_arcs = algorithmic_bytes(stats, V)  # of one full launch (n_last steps)
        k_ms = float(np.mean([k for k, n in launches if n == n_last] or [k for k, _ in launches]))
        traffic = profiled_traffic(wl["desc"], n_last * B * T, opts.get("prune_interval"), opts.get("min_active", 0))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        frames_total = world * B * T * steps_timed
        cap = bt.capacity()
        res = {
            "metric": "decoded audio frames/sec (whole node), LibriSpeech HLG beam=15; "
                      "1-best WER parity",
            "value": frames_total * SUBSAMPLING / elapsed,
            "unit": "audio_frames/s",
            "n_gpus": world,
            # every rank's own figure (its utterances over its own time); `value` is all
            # ranks' utterances over the slowest rank's time
            "per_rank_value": [B * T * steps_timed * SUBSAMPLING / e for e in per_rank_elapsed],
            "gather_ms_per_call": gather_ms,
            "gather_backend": (args.backend if world > 1 else None),
            "gather_verified": verified,
            # `steps` is what --steps asked for; the timed region repeats them `repeats` times
            # (until it lasts --min-seconds): `steps_timed` steps went into `value`
            "steps": args.steps,
            "steps_timed": steps_timed,
            "repeats": repeats,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps_timed * 1e3,
            "timed_seconds": elapsed,
            "value_prune_interval_off": value_pi_off,
            "value_prune_eager": side_eager["value"] if side_eager else None,
            "value_min_active_200": side_min200["value"] if side_min200 else None,
            "value_max_active_10000": side_max10k["value"] if side_max10k else None,
            "side_runs": [x for x in (side_eager, side_min200, side_max10k) if x],
            "large_lm": large_lm,
            "host_input": host_input,
            "single_batch_ms": single_batch_ms,
            "single_batch_value": B * T * SUBSAMPLING / (single_batch_ms * 1e-3),
            "single_batch_kernel_ms": single_kernel_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if wl["lattice"] else "f64",
            "data": "synthetic",
            "config": {"workload": wl["desc"], "graph_states": info["states"],
                       "graph_arcs": info["arcs"], "batch_per_gpu": B, "frames": T,
                       "prune_interval": (args.prune_interval or wl["opts"].get("prune_interval")),
                       "min_active": opts.get("min_active", 0),
                       "prune_policy": ("owed: run on un-finalized export or when a pool is full "
                                        "(library default; prunes run in the timed region: %d)"
                                        % int(stats[:, len(cabi.STAT_NAMES) - 1].sum())) if wl["lattice"] else None,
                       "steps_per_call": G, "utterances_per_launch": n_last * B,
                       "calls_in_flight": n_fly, "workgroups_per_launch": cap["workgroups"],
                       "device_mb_per_call": bt.device_bytes / 2**20,
                       "device_mb_per_step": bt.device_bytes / 2**20 / G,
                       "pool_grow_events": cap["grow_events"],
                       "decoder_frames_per_s": frames_total / elapsed, **per_frame},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": "SearchKernel<1> (search + FinalizeDecoding + GetRawLattice "
                                   "of every utterance of the call)" if wl["lattice"]
                                   else "SearchKernel",
                         "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # rounds 1-4 priced every arc the reference's loop visits as read; the
                         # label index (round 5) reads a quarter of them: the same launch's
                         # fraction by the old definition, for comparison across rounds only
                         "frac_if_every_scanned_arc_were_read": alg_bytes_all_arcs / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch_every_scanned_arc": alg_bytes_all_arcs,
                         # the same launch with nothing else in flight (the timed region
                         # keeps `calls_in_flight` launches on the GPU, which share the CUs
                         # and stretch one another)
                         "launch_alone": {"kernel_ms": alone_ms,
                                          "achieved": alg_bytes / (alone_ms * 1e-3) / 1e9,
                                          "frac": alg_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "utterance_frames_per_launch": n_last * B * T,
                         # with more than one call in flight the launches share the CUs;
                         # the device as a whole moves this many algorithmic bytes per
                         # second of wall time
                         "device_achieved": alg_bytes / (n_last * B * T) * (B * T * steps_timed) / elapsed / 1e9,
                         "device_frac": alg_bytes / (n_last * B * T) * (B * T * steps_timed) / elapsed / 1e9 / HBM_PEAK_GBS},
            "setup_s": setup_s,
        }
        if wl["lattice"]:
            res["config"]["lattice_arcs_per_utt"] = float(np.mean(out.arcs_per_utt))
            res["config"]["lattice_states_per_utt"] = float(np.mean(out.states_per_utt))
        if not args.no_cpu:
            cb, parity = cpu_baseline(g, lps, wl, out, cpu_pool)
            res["cpu_baseline"] = cb
            res["parity"] = parity
            res["gpu_over_cpu_1thread"] = res["value"] / cb["value"]
        print(json.dumps(res))
    if cpu_pool is not None:
        cpu_pool.close()
    gatherer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # one rank of several failing must end the job, not leave the others waiting in a
        # collective until its timeout: torch.distributed.run tears the group down when a
        # rank exits non-zero -- so exit now, without the interpreter's orderly shutdown
        # (which would join the gather thread and the process group first)
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(1)
        raise
