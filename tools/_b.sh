timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_now.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_now.json'))
print(d['value'], d.get('value_prune_interval_off'), d['ms_per_step'], d['single_batch_ms'], d['roofline']['frac'], d['parity'])
PY
