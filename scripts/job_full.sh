#!/bin/bash
# full GPU suite + load time vs replicas + the default bench line (run on the GPU box from the repo root)
python -m pytest tests -m gpu -q -s > gpurun_out/r3_tfull.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tfull.log
tail -4 gpurun_out/r3_tfull.log
mkdir -p gpurun_out/r03
for p in fp32 fp16 fp8; do python scripts/load_time_replicas.py $p > gpurun_out/r03/load_time_replicas_$p.txt 2>&1; cat gpurun_out/r03/load_time_replicas_$p.txt; done
python bench.py > gpurun_out/r3_bench_full.json 2> gpurun_out/r3_bench_full.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench_full.json"))
print(d["metric"])
print(d["dtype"], d["value"], d["ms_per_step"], d.get("modelinfer_images_per_s"), d["roofline"]["kernel"], d["roofline"]["frac"])
for s in d["secondary"]:
    print(s["dtype"], s["value"], s["ms_per_step"], s.get("modelinfer_images_per_s"), s.get("modelinfer_uint8_images_per_s"), s["roofline"]["kernel"], s["roofline"]["frac"], s["roofline_model"]["frac_of_tight_bound"])
PY
