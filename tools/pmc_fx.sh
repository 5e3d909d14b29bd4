#!/bin/bash
# GPU-box helper: where do k_blur_fx's wave cycles go?  SQ counters for the bench frame replayed on one context (rocprofv3 --pmc passes,
# kernel-trace only), reduced to the fused blur's launches.  usage: bash tools/pmc_fx.sh [tag]
tag=${1:-fx}
root=$(pwd)
export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAIT_IFETCH" ; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_${tag}_$i
  rm -rf $out; mkdir -p $out
  (cd /tmp && N=8 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o run -- python3 $root/tools/one_frame.py > $out.log 2>&1)
  f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_blur_fx" not in k and "k_composite_tiles<4, true>" not in k: continue
    k = "k_blur_fx" if "k_blur_fx" in k else "k_composite_tiles<4,true>"
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
done
