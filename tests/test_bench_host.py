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
        # average launch: rocprofv3's trace against the HIP events of the bench line
        assert abs(s["avg_launch_ms"] - b["roofline"]["kernel_ms"]) / b["roofline"]["kernel_ms"] < 0.05


def test_library_is_not_built_with_the_option_that_stalled_the_large_workload():
    """DESIGN §5 "A 5 % that is not taken": -amdgpu-prealloc-sgpr-spill-vgprs made the bench
    5 % faster and the large-LM workload stall or fault in 2 runs of 3.  Whoever turns it
    back on has to have run `bench.py --workload hlg-large` a number of times first."""
    mk = open(os.path.join(ROOT, "kaldi-decoder_amd", "Makefile")).read()
    flags = [ln for ln in mk.splitlines() if ln.startswith("HIPFLAGS")]
    assert flags and all("prealloc-sgpr-spill-vgprs" not in ln for ln in flags)
