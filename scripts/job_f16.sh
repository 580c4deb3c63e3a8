#!/bin/bash
# fp16 weights-stationary / dense-block tests + the fp16 B=128 bench line with the tuner's log (run on the GPU box from the repo root)
python -m pytest tests -m gpu -q -s -k "fp16_weights_stationary or dense_block" > gpurun_out/r3_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t3.log
grep -E "passed|failed|Error|error|rc=" gpurun_out/r3_t3.log | tail -6
IE_TUNE_LOG=1 python bench.py --dtype f16 --batch 128 --no-secondary --cpu-sample 0 --no-hostpath > gpurun_out/r3_bench_f16.json 2> gpurun_out/r3_bench_f16.err
grep -E "k=3x3|kh=3|3x3" gpurun_out/r3_bench_f16.err | head -30
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench_f16.json"))
print(d["value"], d["ms_per_step"], d["kernel_families_ms"])
PY
IE_PRECISION=fp16 python scripts/profile_steps.py 128 > gpurun_out/r3_steps_f16.txt 2>&1; grep -E "conv3x3|^#" gpurun_out/r3_steps_f16.txt | head -8
