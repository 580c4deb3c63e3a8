#!/bin/bash
# fp16 dense-block tests + the fp16 B=128 bench line (run on the GPU box from the repo root)
python -m pytest tests -m gpu -q -s -k "dense_block or fp16_densenet121 or config2 or config3" > gpurun_out/r3_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t3.log
grep -E "dense-block|rel err|passed|failed|Error|error" gpurun_out/r3_t3.log | tail -30
IE_TUNE_LOG=1 python bench.py --dtype f16 --batch 128 --no-secondary > gpurun_out/r3_bench_f16.json 2> gpurun_out/r3_bench_f16.err
grep "dense block" gpurun_out/r3_bench_f16.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench_f16.json"))
print(d["value"], d["ms_per_step"], d["kernel_families_ms"], d["modelinfer_images_per_s"], d["modelinfer_uint8_images_per_s"])
PY
