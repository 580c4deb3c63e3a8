#!/bin/bash
# stem + pool fusion: its tests, the fp8 / fp16 suites that run through it, and the three bench lines (run on the GPU box from the repo root)
python -m pytest tests -m gpu -q -s -k "stem or fp8 or fp16 or config" > gpurun_out/r3_tstem.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tstem.log
grep -E "stem \+ pool|passed|failed|Error|error|rc=" gpurun_out/r3_tstem.log | tail -14
IE_TUNE_LOG=1 python bench.py > gpurun_out/r3_bench_stem.json 2> gpurun_out/r3_bench_stem.err
grep "stem + pool" gpurun_out/r3_bench_stem.err | head
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench_stem.json"))
print(d["dtype"], d["value"], d["ms_per_step"], d.get("modelinfer_images_per_s"), d["kernel_families_ms"])
for s in d["secondary"]:
    print(s["dtype"], s["value"], s["ms_per_step"], s.get("modelinfer_images_per_s"), s.get("modelinfer_uint8_images_per_s"), s["kernel_families_ms"])
PY
