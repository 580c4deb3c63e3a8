#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel trace + separate PMC passes -> gpurun_out/<round>/
#   fp32 batch 32 (the headline, BASELINE configs[1]) and fp16 batch 128 (configs[2]); PMC passes never combine with tracing.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export IE_TUNE_CACHE=$OUT/tune_cache.txt
python3 $R/bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 $R/bench.py --steps 50 --warmup 10 --dtype f16 --batch 128 --cpu-sample 0 > $OUT/bench_f16_b128.json 2> $OUT/bench_f16_b128.err || exit 1
python3 $R/scripts/profile_steps.py 32 > $OUT/steps_b32.txt 2>&1
IE_PRECISION=fp16 python3 $R/scripts/profile_steps.py 128 > $OUT/steps_f16_b128.txt 2>&1
# hipGraphLaunch crashed inside rocprofv3 kernel tracing on this pool (SIGSEGV in the tool); eager launches of the same plan are traced instead
IE_DISABLE_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-sample 0 --no-hostpath > $OUT/rocprof_kernel_trace.log 2>&1
cp $OUT/kt/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/kt
IE_DISABLE_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt16 -- python3 $R/bench.py --steps 50 --warmup 10 --dtype f16 --batch 128 --cpu-sample 0 --no-hostpath > $OUT/rocprof_kernel_trace_f16.log 2>&1
cp $OUT/kt16/*/*_kernel_stats.csv $OUT/kernel_stats_f16_b128.csv 2>/dev/null
rm -rf $OUT/kt16
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  N=$(echo $C | cut -d" " -f1)
  IE_DISABLE_GRAPH=1 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-hostpath > $OUT/pmc_$N.log 2>&1 || echo "PMC pass $N failed"
  python3 $R/scripts/pmc_summary.py $OUT/pmc_$N > $OUT/pmc_$N.summary.csv
  rm -rf $OUT/pmc_$N
done
for C in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | cut -d" " -f1)
  IE_DISABLE_GRAPH=1 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc16_$N -- python3 $R/bench.py --steps 2 --warmup 1 --dtype f16 --batch 128 --cpu-sample 0 --no-hostpath > $OUT/pmc_${N}_f16.log 2>&1 || echo "PMC pass $N (f16) failed"
  python3 $R/scripts/pmc_summary.py $OUT/pmc16_$N > $OUT/pmc_${N}_f16.summary.csv
  rm -rf $OUT/pmc16_$N
done
python3 $R/scripts/make_traffic.py $OUT > $OUT/traffic.json
python3 $R/scripts/make_traffic.py $OUT _f16 > $OUT/traffic_f16_b128.json
ls -la $OUT
