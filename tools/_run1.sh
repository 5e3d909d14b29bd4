export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/dyn; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o r -- python3 $root/tools/dynamic_probe.py > $out/log.txt 2>&1)
f=$(ls $out/*kernel_trace.csv $out/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_gaps.py $f | head -24
