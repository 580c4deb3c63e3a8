#!/bin/bash
# fp8 suites + the ResNet-50 fp8 bench line with the tuner's log (run on the GPU box from the repo root)
python -m pytest tests -m gpu -q -s -k "fp8 or config4" > gpurun_out/r3_tf8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tf8.log
grep -E "weights-stationary|passed|failed|Error|error|rc=" gpurun_out/r3_tf8.log | tail -14
python bench.py --model resnet50 --dtype f8 --no-secondary --cpu-sample 0 --no-hostpath > gpurun_out/r3_bench_f8.json 2> gpurun_out/r3_bench_f8.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench_f8.json"))
print(d["dtype"], d["value"], d["ms_per_step"], d["kernel_families_ms"])
PY
IE_PRECISION=fp8 python scripts/profile_steps.py 256 resnet50 > gpurun_out/r3_steps_f8.txt 2>&1; grep -E "proj|^#" gpurun_out/r3_steps_f8.txt
