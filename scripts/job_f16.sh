#!/bin/bash
# fp16 dense-block tests + the fp16 B=128 bench line (run on the GPU box from the repo root)
python -m pytest tests -m gpu -q -s -k "dense_block" > gpurun_out/r3_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t3.log
grep -E "dense-block|rel err|passed|failed|Error|error" gpurun_out/r3_t3.log | tail -12
for band in 0 1; do
IE_DENSE_BAND=$band IE_TUNE_LOG=1 python bench.py --dtype f16 --batch 128 --no-secondary --cpu-sample 0 --no-hostpath > gpurun_out/r3_bench_f16.json 2> gpurun_out/r3_bench_f16.err
grep "dense block" gpurun_out/r3_bench_f16.err | head -40
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench_f16.json"))
print(d["value"], d["ms_per_step"], d["kernel_families_ms"])
PY
done
