timeout 600 python bench.py --no-cpu --prune-interval 1048576 2>/dev/null | tail -1 > gpurun_out/b_off.json
timeout 600 python bench.py --no-cpu 2>/dev/null | tail -1 > gpurun_out/b_25.json
python - <<'PY'
import json
for f in ('b_off','b_25'):
    d=json.load(open('gpurun_out/%s.json'%f))
    print(f, d['value'], d.get('value_prune_interval_off'), d['ms_per_step'], d['single_batch_ms'], d['config']['pool_grow_events'], d['config']['prune_interval'])
PY
