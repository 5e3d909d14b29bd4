#!/bin/bash
# same-box A/B: round 5's final library (build/libfigdraw_hip_r05g.so, built from commit bc298da) against the tree's
export TMPDIR=/tmp; o=gpurun_out/s11; mkdir -p $o
python3 tools/ab_kernels.py 3 r05g tree 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee $o/ab_kernels.txt
FIGDRAW_HIP_LIB=$PWD/build/libfigdraw_hip_r05g.so python3 tools/perf_configs.py 10 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
bash tools/cfg_ab.sh "2 9" r05g 2>&1 | grep "config\|==" | tee $o/cfg_ab.txt
