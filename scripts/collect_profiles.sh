#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel trace + separate PMC passes -> gpurun_out/<round>/
#   fp32 batch 32 DenseNet-121 (the headline, BASELINE configs[1]), fp16 batch 128 (configs[2]), ResNet-50 fp8 batch 256 (configs[4]).
#   plus the opt-in fp32 + bf16x6 configuration (batch 32).  PMC passes never combine with tracing domains.  Since round 3 the traced / counted
#   runs are the SAME program bench.py times: hipGraph replays, with the in-flight depth capped (IE_MAX_INFLIGHT_REPLAYS=8: the engine
#   synchronises every 8 replays) -- deep un-synchronised graph queues under rocprofv3 --kernel-trace segfaulted inside hipGraphLaunch
#   (profiles/r02/graph_burst_under_kernel_trace_sigsegv.log; single synchronised launches always traced fine).  IE_PROFILE_EAGER=1 goes
#   back to eager launches of the plan's kernels.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export IE_TUNE_CACHE=$OUT/tune_cache.txt
TRACE_ENV="IE_DISABLE_GRAPH=0 IE_MAX_INFLIGHT_REPLAYS=8"
[ -n "$IE_PROFILE_EAGER" ] && TRACE_ENV="IE_DISABLE_GRAPH=1"
ROUND=$(basename $OUT)
python3 $R/scripts/profile_steps.py 32 > $OUT/steps_b32.txt 2>&1
IE_PRECISION=fp16 python3 $R/scripts/profile_steps.py 128 > $OUT/steps_f16_b128.txt 2>&1
IE_PRECISION=fp8 python3 $R/scripts/profile_steps.py 256 resnet50 > $OUT/steps_resnet50_f8_b256.txt 2>&1
IE_FP32_SPLIT=1 python3 $R/scripts/profile_steps.py 32 > $OUT/steps_f32x6_b32.txt 2>&1
echo "step tables done"
# name | bench arguments | file suffix
CONFIGS=("f32|--model densenet121 --dtype f32 --batch 32|" "f16|--model densenet121 --dtype f16 --batch 128|_f16_b128" "r50f8|--model resnet50 --dtype f8 --batch 256|_resnet50_f8_b256" "f32x6|--model densenet121 --dtype f32x6 --batch 32|_f32x6_b32")
for CFG in "${CONFIGS[@]}"; do
  IFS="|" read -r NAME ARGS SUF <<< "$CFG"
  env $TRACE_ENV rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$NAME -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-sample 0 --no-hostpath --no-secondary $ARGS > $OUT/rocprof_kernel_trace$SUF.log 2>&1
  echo "kernel trace $NAME rc=$?"
  cp $OUT/kt_$NAME/*/*_kernel_stats.csv $OUT/kernel_stats$SUF.csv 2>/dev/null
  rm -rf $OUT/kt_$NAME
  for C in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    N=$(echo $C | cut -d" " -f1)
    env $TRACE_ENV rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${NAME}_$N -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-hostpath --no-secondary $ARGS > $OUT/pmc_$N$SUF.log 2>&1 || echo "PMC pass $N ($NAME) failed"
    python3 $R/scripts/pmc_summary.py $OUT/pmc_${NAME}_$N > $OUT/pmc_$N$SUF.summary.csv
    rm -rf $OUT/pmc_${NAME}_$N
  done
  python3 $R/scripts/make_traffic.py $OUT $SUF > $OUT/traffic$SUF.json
  echo "config $NAME done"
done
# the bench line last, with this run's traffic files already in profiles/<round>/ so that its roofline.traffic comes from the same session
mkdir -p $R/profiles/$ROUND && cp $OUT/traffic*.json $R/profiles/$ROUND/
python3 $R/bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done"
ls -la $OUT
