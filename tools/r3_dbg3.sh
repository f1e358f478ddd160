#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
echo "== prof 512 (2 slots)"; KD_HIP_LIB=libkd_hip_prof.so timeout 300 python tools/prof_phases.py hlg-lattice 512 2>&1 | grep -v "^/opt"
echo "== prof_ms1 256 (1 slot, wave tier on)"; KD_HIP_LIB=libkd_hip_prof_ms1.so timeout 300 python tools/prof_phases.py hlg-lattice 256 2>&1 | grep -v "^/opt"
