#!/bin/bash
# round 6: the committed profile set of the round's library (profiles/r06_f_*, r06_configs.json, r06_config<n>_kernel_stats.csv)
export TMPDIR=/tmp; root=$(pwd)
bash tools/refresh_profiles.sh r06_f > gpurun_out/r06_f_refresh.log 2>&1
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r06_f/r06_f_bench_driver_style_$i.json; done
bash tools/refresh_configs.sh r06 > gpurun_out/r06_configs_refresh.log 2>&1
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3 > gpurun_out/r06_gpu_suite.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_f/r06_f_bench*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['frac'], d['one_frame_at_a_time']['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
d=json.load(open('gpurun_out/r06_configs.json'))
for k,v in d.items(): print(k, v['frame_us'], v['kernel_us']['bin'], v['kernel_us']['composite_phase0'], v['kernel_us']['composite_all'], v['roofline']['frac'], v.get('parity_vs_oracle'))
PY
cat gpurun_out/r06_gpu_suite.txt
