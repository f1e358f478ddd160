#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV directory: for every SearchKernel dispatch its
start, end and how many other SearchKernel dispatches were running when it started."""
import csv, glob, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + '/*kernel_trace.csv')[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']) for r in rows if 'Search' in r['Kernel_Name'])
t0 = ks[0][0]
busy = 0
for i, (s, e, q) in enumerate(ks):
    running = sum(1 for (s2, e2, _) in ks[:i] if e2 > s)
    if i >= len(ks) - int(sys.argv[2] if len(sys.argv) > 2 else 12):
        print('%9.2f -> %9.2f  dur %7.2f ms  q %s  running at start: %d' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, running))
span = ks[-1][1] - ks[0][0]
tot = sum(e - s for s, e, _ in ks)
print('dispatches %d, sum of durations / span = %.2f' % (len(ks), tot / span))
