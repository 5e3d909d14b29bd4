#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s7; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; tail -4 $o/pytest_gpu.txt
python tools/perf_configs.py 2 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > $o/cfg2.json
python -c "
import json; d=json.load(open('$o/cfg2.json'))['config2']; print(d['frame_us'], d['kernel_us'], d['roofline']['frac'], d['parity_vs_oracle'])"
python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; python -c "
import json; d=json.load(open('$o/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['one_frame_at_a_time']['ms_per_step'], d['cpu_baseline']['parity_max_lsb'])"
