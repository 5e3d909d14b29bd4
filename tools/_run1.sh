python -m pytest tests -m gpu -q 2>&1 | tail -3
python3 tools/perf_configs.py > gpurun_out/cfg_new.json 2>/dev/null; python3 -c "
import json;d=json.load(open('gpurun_out/cfg_new.json'))
for k,v in d.items(): print(k, v['frame_us'], v['kernel_us']['composite_phase0'], v['parity_vs_oracle'])"
