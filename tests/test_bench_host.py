"""Host-side pieces of bench.py that need no GPU: which PMC summary the `roofline.traffic`
figure is taken from (and how it follows the launch size), the workloads' options."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-decoder_amd", "python")]
import bench  # noqa: E402


def test_lattice_workloads_run_at_the_reference_default_prune_interval():
    # lattice-simple-decoder.h: prune_interval = 25; the bench line's headline is that config
    for name, wl in bench.WORKLOADS.items():
        if wl["lattice"] and "prune_interval" in wl["opts"]:
            assert wl["opts"]["prune_interval"] == 25, name
    hl = bench.WORKLOADS["hlg-lattice"]
    assert (hl["opts"]["beam"], hl["opts"]["lattice_beam"], hl["batch"]) == (15.0, 8.0, 128)


def test_traffic_comes_from_a_summary_of_the_same_workload_and_options(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    line = {"config": {"workload": "W", "prune_interval": 25},
            "roofline": {"utterance_frames_per_launch": 256000}}
    (prof / "r07_summary.json").write_text(json.dumps(
        {"hbm_bytes_per_launch": 64e9, "bench_line_under_profiler": line}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    # same launch size: the figure as measured
    assert bench.profiled_traffic("W", 256000, 25) == (64e9, "r07_summary.json")
    # another launch size: scaled by utterance-frames, and the source says so
    v, src = bench.profiled_traffic("W", 128000, 25)
    assert v == 32e9 and "scaled" in src and "256000" in src
    # other options or another workload: no figure rather than a wrong one
    assert bench.profiled_traffic("W", 256000, 1 << 20) is None
    assert bench.profiled_traffic("other", 256000, 25) is None
    # a trace-only summary of the workload (no counter passes, e.g. r03_hlg_large_summary.json)
    # is passed over, whichever way the file names sort
    line2 = {"config": {"workload": "L", "prune_interval": 25},
             "roofline": {"utterance_frames_per_launch": 256000}}
    (prof / "r07_large_summary.json").write_text(json.dumps({"bench_line_under_profiler": line2}))
    (prof / "r08_w_trace_summary.json").write_text(json.dumps({"bench_line_under_profiler": line}))
    assert bench.profiled_traffic("L", 256000, 25) is None
    assert bench.profiled_traffic("W", 256000, 25) == (64e9, "r07_summary.json")


def test_committed_summary_matches_the_bench_line_next_to_it():
    """profiles/rNN_summary.json is taken under rocprofv3 on the command the bench line
    comes from: same workload, options and launch size; the two clocks agree."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))[-1:]:
        s = json.load(open(path))
        b = json.load(open(path.replace("_summary", "_bench_line")))
        under = s["bench_line_under_profiler"]
        assert under["config"]["workload"] == b["config"]["workload"]
        assert under["config"]["prune_interval"] == b["config"]["prune_interval"]
        assert under["roofline"]["utterance_frames_per_launch"] == b["roofline"]["utterance_frames_per_launch"]
        assert b["roofline"]["traffic"] == s["hbm_bytes_per_launch"]
        # average launch: rocprofv3's trace against the HIP events of the SAME (profiled) run
        assert abs(s["avg_launch_ms"] - under["roofline"]["kernel_ms"]) / under["roofline"]["kernel_ms"] < 0.05
        # ... and against the un-profiled line: the same launches, stretched by however much the
        # calls in flight overlap -- between the launch-alone duration and the bench's own figure
        # (under the profiler the host thread is slower and the calls overlap less)
        lo = 0.95 * min(b["roofline"]["launch_alone"]["kernel_ms"], under["roofline"]["launch_alone"]["kernel_ms"])
        assert lo <= s["avg_launch_ms"] <= 1.05 * b["roofline"]["kernel_ms"]


def test_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 8` as the driver may run it (no torchrun around it, WORLD_SIZE
    unset): bench.py starts `python -m torch.distributed.run` on its own command line as a child
    process -- before torch is imported or any HIP call is made -- and exits with its code."""
    import subprocess
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "16", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    had_torch = "torch" in sys.modules
    try:
        bench.main()
        raise AssertionError("main() should have exited with the child's code")
    except SystemExit as e:
        assert e.code == 7
    assert had_torch or "torch" not in sys.modules  # (nothing was imported on the way)
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "16", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_ranks_share_one_synthesis_of_the_graph_through_a_file(monkeypatch):
    """`bench.py --gpus N`: rank 0 synthesises the workload's graph once and writes it as this
    package's OpenFst const file (plus the lexicon); the other ranks read it back with Fst.read
    -- the same arrays, bit for bit (SURVEY 8(f)-1 exercised on the way)."""
    import numpy as np
    wl = bench.WORKLOADS["small"]
    g0, files = bench.shared_graph(wl, "small", 0, 2, "unittest%d" % os.getpid())
    try:
        assert files and all(os.path.exists(f) for f in files)
        g1, none = bench.shared_graph(wl, "small", 1, 2, "unittest%d" % os.getpid(), timeout_s=5.0)
        assert none is None
        for k in ("row_ptr", "ilabel", "olabel", "weight", "nextstate", "final"):
            assert np.array_equal(np.asarray(g0[k]), np.asarray(g1[k])), k
        assert g0["num_states"] == g1["num_states"] and g0["start"] == g1["start"]
        assert np.array_equal(g0["lexicon"][0], g1["lexicon"][0]) and np.array_equal(g0["lexicon"][1], g1["lexicon"][1])
    finally:
        for f in files or ():
            if os.path.exists(f):
                os.remove(f)
    # one rank: no file at all
    g, files = bench.shared_graph(wl, "small", 0, 1, "unused")
    assert files is None and g["num_states"] == g0["num_states"]
