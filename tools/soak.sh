#!/bin/bash
# tools/soak.sh [runs] [workload] -- the bench command over and over (short timed region): counts
# runs that fail or hang.  (Round 3: the lattice sweep's descriptor race showed as one run in six.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
N=${1:-30}
W=${2:-hlg-lattice}
ok=0; bad=0
for i in $(seq 1 $N); do
  timeout 240 python bench.py --no-cpu --headline-only --workload $W --min-seconds 0.6 > /tmp/b.json 2> /tmp/b.err
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i rc=$rc"; tail -2 /tmp/b.err; fi
done
echo "soak $W: ok=$ok bad=$bad of $N"
