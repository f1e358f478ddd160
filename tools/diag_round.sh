#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
echo "== sizes, bench workload (prunes owed)"; KD_HIP_LIB=libkd_hip_prof_size.so timeout 300 python tools/prof_sizes.py hlg-lattice 128 2>&1 | grep -v amdgpu.ids | tail -16
echo "== sizes, min_active 200"; KD_MIN_ACTIVE=200 KD_HIP_LIB=libkd_hip_prof_size.so timeout 300 python tools/prof_sizes.py hlg-lattice 128 2>&1 | grep -v amdgpu.ids | tail -16
echo "== phases, bench workload"; KD_HIP_LIB=libkd_hip_prof.so timeout 300 python tools/prof_phases.py hlg-lattice 128 2>&1 | grep -v amdgpu.ids | tail -32
echo "== prof_tail WITH -amdgpu-prealloc-sgpr-spill-vgprs, prunes run when their frames come"; KD_PRUNE_LAZY=0 KD_HIP_LIB=libkd_hip_prof_tail_pre.so timeout 300 python tools/prof_phases.py hlg-lattice 128 2>&1 | grep -v amdgpu.ids | tail -12; echo "rc=$?"
echo "== phases, min_active 200"; KD_MIN_ACTIVE=200 KD_HIP_LIB=libkd_hip_prof.so timeout 300 python tools/prof_phases.py hlg-lattice 128 2>&1 | grep -v amdgpu.ids | tail -32
