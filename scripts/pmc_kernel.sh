#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes over a short bench run with the environment's kernel overrides (IE_FORCE_ALGO, ...);
# one summary CSV per pass under gpurun_out/<dir>/.  usage: pmc_kernel.sh <outdir-name> [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export IE_TUNE_CACHE=0
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-hostpath --no-secondary "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
  python3 $R/scripts/pmc_summary.py $OUT/p$i > $OUT/pass$i.csv
  rm -rf $OUT/p$i
done
head -4 $OUT/pass*.csv
