#!/bin/bash
# End-of-round measurement on the GPU box: kernel trace + PMC passes + the bench line (fp32 headline; bf16 when asked).
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r03 [bf16]'
TAG=${1:-r03_x}
RND=${TAG%%_*}
DT=${2:-f32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 200 --warmup 20 --no-graph --no-cpu-baseline --no-roofline --no-secondary --dtype $DT > $OUT/trace_bench.log 2>&1 || exit 1
echo trace done
export BSAREC_PMC_DTYPE=$DT
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch -- python3 $R/tools/pmc_run.py 4 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write -- python3 $R/tools/pmc_run.py 4 > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc/sq1 -- python3 $R/tools/pmc_run.py 4 > $OUT/pmc_sq1.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc/sq2 -- python3 $R/tools/pmc_run.py 4 > $OUT/pmc_sq2.log 2>&1 || exit 1
echo pmc done
cd $R
python3 tools/pmc_aggregate.py $OUT/pmc $OUT/pmc.csv
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
python3 tools/timeline.py $(find $OUT/trace -name '*kernel_trace.csv' | head -1) 50 > $OUT/timeline.txt 2>&1
find $OUT/trace -name '*.csv' ! -name '*stats*' -delete; find $OUT -size +8M -delete
python3 tools/prof_summary.py $OUT/kernel_stats.csv 220 20
cat $OUT/timeline.txt
# the bench line's roofline.traffic comes from the PMC passes of THIS build (same library_sha16)
if [ "$DT" = f32 ]; then cp $OUT/pmc.csv $R/profiles/${RND}_pmc_C1.csv; fi
timeout -k 10 300 python3 bench.py --dtype $DT > $OUT/bench_line.json 2> $OUT/bench.err
cat $OUT/bench_line.json
# ... and the driver's own command line (BENCH_rNN.json): short run, every graph built before the clock
if [ "$DT" = f32 ]; then timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_cmd.json 2>> $OUT/bench.err; cut -c1-260 $OUT/bench_line_driver_cmd.json; fi
