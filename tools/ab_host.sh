#!/bin/bash
# GPU-box A/B of two builds through bench.py's dynamic path (the C player links the in-tree library: the file is swapped in place)
# usage: bash tools/ab_host.sh build/libfigdraw_hip_<name>.so [rounds]
other=$1; n=${2:-2}
cp figdraw_amd/libfigdraw_hip.so /tmp/lib_tree.so
one() { python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; j=json.loads(sys.stdin.read()); d=j['dynamic_path']; print('$1', j['value'], j['per_call_path']['value'], 'one', j['one_frame_at_a_time']['ms_per_step'], 'record', d['host_record_us'], 'prepare', d['host_prepare_us'], 'issue', d['host_issue_us'], 'batches', j['batches_ms'][:5])"; }
for i in $(seq $n); do
  cp /tmp/lib_tree.so figdraw_amd/libfigdraw_hip.so; rm -f build/libfdh_call_player.so; one tree
  cp $other figdraw_amd/libfigdraw_hip.so; rm -f build/libfdh_call_player.so; one other
done
cp /tmp/lib_tree.so figdraw_amd/libfigdraw_hip.so
