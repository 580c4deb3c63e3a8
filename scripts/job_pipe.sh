#!/bin/bash
for c in 2 4; do for h in 9 16 24 32 40 44; do
  echo "== fp16 B=128 chunks=$c head=$h"
  IE_PIPELINE_CHUNKS=$c IE_PIPELINE_HEAD=$h python bench.py --dtype f16 --batch 128 --no-secondary --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], 'FLOAT32', d['modelinfer_p50_ms'], d['modelinfer_images_per_s'], 'dev', d['modelinfer_device_ms'], 'UINT8', d['modelinfer_uint8_p50_ms'], d['modelinfer_uint8_images_per_s'], d['modelinfer_pipeline'])"
done; done
