#!/usr/bin/env python3
"""Condensed view of gpurun_out/ab.log (tools/ab.sh)."""
import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ab.log"):
    l = l.rstrip()
    if l.startswith('{'):
        d = json.loads(l)
        if "roofline" in d:
            print('   value %.2fM ms/step %.2f kernel_ms %.2f frac %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
        else:
            print('  ', l)
    elif l.startswith('==') or 'Error' in l or 'passed' in l or 'failed' in l or 'cyc/utt' in l or 'cycles' in l:
        if 'cyc/utt' in l and ' 0.0%' in l: continue
        print(l)
