#!/bin/bash
# tools/side_workloads.sh rNN -- on the GPU box: the bench lines of the 1-best workloads
# (BASELINE configs[1]: FasterDecoder over the HL graph, B = 64; the 1-best decoder over the HLG)
# of the shipped library, two runs each
R=${1:-r06}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
H=$(sha256sum kaldi-decoder_amd/python/kaldi_decoder/lib/libkd_hip.so | cut -c1-16)
for w in hl hlg-1best; do
  for rep in 1 2; do
    timeout 600 python bench.py --workload $w --no-cpu > gpurun_out/side_${R}_${w}_$rep.json 2> gpurun_out/side_${R}_${w}_$rep.err || tail -3 gpurun_out/side_${R}_${w}_$rep.err
    python - gpurun_out/side_${R}_${w}_$rep.json $H <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
d["library_sha256"] = sys.argv[2]
json.dump(d, open(sys.argv[1], "w"))
print(d["config"]["workload"][:60], "%.2f M" % (d["value"] / 1e6), "ms/step %.3f" % d["ms_per_step"],
      "roofline", d.get("roofline", {}).get("frac"), "parity", d.get("parity"))
PY
  done
done
