#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel trace + separate PMC passes -> gpurun_out/r01/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r01}
mkdir -p $OUT
export IE_TUNE_CACHE=$OUT/tune_cache.txt
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 $R/scripts/profile_steps.py 32 > $OUT/steps_b32.txt 2>&1
# hipGraphLaunch crashed inside rocprofv3 kernel tracing on this pool (SIGSEGV in the tool); eager launches of the same plan are traced instead
IE_DISABLE_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-sample 0 --no-hostpath > $OUT/rocprof_kernel_trace.log 2>&1
cp $OUT/kt/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/kt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  N=$(echo $C | cut -d" " -f1)
  IE_DISABLE_GRAPH=1 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-hostpath > $OUT/pmc_$N.log 2>&1 || echo "PMC pass $N failed"
  python3 $R/scripts/pmc_summary.py $OUT/pmc_$N > $OUT/pmc_$N.summary.csv
  rm -rf $OUT/pmc_$N
done
ls -la $OUT
