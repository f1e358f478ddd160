#!/bin/bash
# tools/variants_round.sh -- on the GPU box: the GPU suite under the policies that are not the default --
# every prune run when its frame comes; every call dealt out over two kinds of workgroups that stay
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
echo "== the GPU suite with every prune run when its frame comes (KD_PRUNE_LAZY=0)"
KD_PRUNE_LAZY=0 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
if [ "$1" != "eager" ]; then
echo "== the GPU suite with every call dealt out over two kinds of workgroups that stay (KD_SPLIT=1, min batch 1)"
KD_SPLIT=1 KD_SPLIT_MIN_BATCH=1 KD_SPLIT_STAY=1 KD_MIG_LO=3 KD_MIG_HI=6 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
fi
