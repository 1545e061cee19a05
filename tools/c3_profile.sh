#!/bin/bash
# C3-shape timings + kernel trace of the generic tiled path on the GPU box:  gpurun -- 'bash tools/c3_profile.sh <tag> [f32|bf16|both]'
TAG=${1:-c3}
WHAT=${2:-both}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for DT in f32 bf16; do
  if [ "$WHAT" != both ] && [ "$WHAT" != $DT ]; then continue; fi
  timeout -k 10 120 python3 $R/tools/c3_run.py 6 $DT > $OUT/c3_$DT.log 2>&1 || exit 1
  tail -1 $OUT/c3_$DT.log
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$DT -- python3 $R/tools/c3_run.py 6 $DT > $OUT/trace_$DT.log 2>&1 || exit 1
  find $OUT/trace_$DT -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_c3_$DT.csv \;
  rm -rf $OUT/trace_$DT
  python3 $R/tools/prof_summary.py $OUT/kernel_stats_c3_$DT.csv 7 16
done
