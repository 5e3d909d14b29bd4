#!/bin/bash
# GPU-box helper: SQ / TA counters of the config-4 frame (tools/cfg4.py), one rocprofv3 --pmc pass per counter set
root=$(pwd); export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
  "VALUBusy SALUBusy" ; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_cfg4_$i; rm -rf $out; mkdir -p $out
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o run -- python3 $root/tools/cfg4.py > $out.log 2>&1)
  f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 $root/tools/pmc_summary.py $f | grep -A9 "k_composite_tiles" || tail -3 $out.log
done
