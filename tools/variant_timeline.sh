#!/bin/bash
# bench + kernel timeline for library variants: bash tools/variant_timeline.sh <lib.so|default> ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset BSAREC_LIB; else export BSAREC_LIB=$R/$v; fi
  echo "=== $v"
  rm -rf $R/gpurun_out/tlv
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tlv -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > $R/gpurun_out/tlv.log 2>&1 || { tail -5 $R/gpurun_out/tlv.log; exit 1; }
  python3 $R/tools/timeline.py $(find $R/gpurun_out/tlv -name "*kernel_trace.csv") 50
done
rm -rf $R/gpurun_out/tlv
