#!/usr/bin/env python3
"""A/B of the pipelined ModelInfer path: p50 of the C call and device ms per forward for (chunks, head) settings.

    python scripts/hostpath_sweep.py [f32|f16] [batch]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _pkg import load_package  # noqa: E402

load_package()
from gpu_ai_inference_server_amd import binding as B  # noqa: E402
from gpu_ai_inference_server_amd.modelgen import models  # noqa: E402

import bench  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
Bsz = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if dtype == "f32" else 128)
os.environ["IE_PRECISION"] = "fp16" if dtype == "f16" else "fp32"
mdir = bench.model_dir("densenet121")
x = models.synthetic_input((Bsz, 3, 224, 224), stream="sweep")
xb = np.clip(x * 255.0, 0, 255).astype(np.uint8)
outs = [B.OutputConfig("fc6_1", [Bsz, 1000, 1, 1])]
for chunks, head in ((0, -1), (2, 1), (2, 2), (2, 8), (2, 14), (2, 16), (2, 30), (4, 14), (4, 30), (2, -1), (4, -1)):
    os.environ["IE_PIPELINE_CHUNKS"] = str(chunks)
    if head >= 0:
        os.environ["IE_PIPELINE_HEAD"] = str(head)
    else:
        os.environ.pop("IE_PIPELINE_HEAD", None)
    m = B.CreateModel(mdir, "densenet_onnx")
    try:
        B.Prepare(m, [[Bsz, 3, 224, 224]], 1)
        row = {"chunks": chunks, "head": head}
        for tag, ins in (("f32", [B.TensorData("data_0", B.DataTypeFloat32, B.Shape([Bsz, 3, 224, 224]), x)]),
                         ("u8", [B.TensorData("data_0", B.DataTypeUint8, B.Shape([Bsz, 3, 224, 224]), xb)])):
            m.InferTimed(ins, outs, 3)
            b0 = B.RuntimeInfo(m)
            t = m.InferTimed(ins, outs, 15)
            b1 = B.RuntimeInfo(m)
            row[tag + "_p50_ms"] = round(float(np.percentile(t, 50)) * 1e3, 3)
            row[tag + "_dev_ms"] = round((b1["device_ms_total"] - b0["device_ms_total"]) / max(1, b1["forwards"] - b0["forwards"]), 3)
            row["used"] = (b1["last_chunks"], b1["last_head_steps"])
        print(json.dumps(row), flush=True)
    finally:
        m.Destroy()
