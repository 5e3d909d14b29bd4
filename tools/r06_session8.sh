#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s8; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; grep -n "passed\|failed" $o/pytest_gpu.txt | tail -3
for d in 1 0 1 0; do
FDH_DIRECT=$d python tools/perf_configs.py 1 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > $o/cfg1_$d.json
python -c "
import json; d=json.load(open('$o/cfg1_$d.json'))['config1']; print('FDH_DIRECT=$d', d['frame_us'], d['kernel_us'], d.get('parity_vs_oracle'))"
done
