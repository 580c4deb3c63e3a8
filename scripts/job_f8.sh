#!/bin/bash
# fp8: ws8 probe on the stage-1 / stage-2 shapes, the fp8 tests, the fp8 bench line
rm -f gpurun_out/r3_ws8.log
for cfg in "256 56 64 256 64" "256 56 256 64" "256 28 128 512 256" "256 28 512 128" "256 14 256 1024 512" "256 14 1024 256"; do
  timeout -k 10 200 build/ws8_probe $cfg >> gpurun_out/r3_ws8.log 2>&1 || echo "probe rc=$? ($cfg)" >> gpurun_out/r3_ws8.log
done
cat gpurun_out/r3_ws8.log
python -m pytest tests -m gpu -q -s -k "fp8 or e4m3 or config4 or private_replicas" > gpurun_out/r3_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t8.log
grep -E "fp8|resnet50|passed|failed|Error|error" gpurun_out/r3_t8.log | tail -30
IE_TUNE_LOG=1 python bench.py --model resnet50 --dtype f8 --no-secondary > gpurun_out/r3_bench_f8.json 2> gpurun_out/r3_bench_f8.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench_f8.json"))
print(d["value"], d["ms_per_step"], d["kernel_families_ms"], d["modelinfer_images_per_s"], d["modelinfer_uint8_images_per_s"])
PY
