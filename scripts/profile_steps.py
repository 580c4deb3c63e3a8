#!/usr/bin/env python3
"""Per-step timing table of the DenseNet-121 plan (HIP events around each launch, eager mode)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _pkg import load_package  # noqa: E402

load_package()
import numpy as np  # noqa: E402
from gpu_ai_inference_server_amd import binding as B  # noqa: E402
from gpu_ai_inference_server_amd.modelgen import models  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model_name = sys.argv[2] if len(sys.argv) > 2 else "densenet121"
mdir = bench.model_dir(model_name)
plan = B.DescribeModel(mdir, batch)["plan"]
m = B.CreateModel(mdir, os.path.basename(os.path.dirname(mdir)))
din, dout = B.Prepare(m, [[batch, 3, 224, 224]], 1)
B.CopyToDevice(m, din[0], models.synthetic_input((batch, 3, 224, 224), stream="prof"))
B.RunPrepared(m, 5, True)
prof = B.Profile(m, 10)
tot = sum(p["ms"] for p in prof)
print(f"# batch {batch}: eager forward {tot:.3f} ms, {sum(p['flops'] for p in prof)/tot/1e9:.1f} TFLOP/s overall")
print(f"{'idx':>3} {'kernel':34} {'M':>7} {'N':>5} {'K':>5} {'ms':>8} {'TF/s':>7} {'GB/s':>7}  name")
for i, (p, s) in enumerate(zip(prof, plan["steps"])):
    M = s["out"]["n"] * s["out"]["h"] * s["out"]["w"]
    K = s["k"][0] * s["k"][1] * s["in"]["c"]
    print(f"{i:3d} {p['kernel']:34} {M:7d} {s['out']['c']:5d} {K:5d} {p['ms']:8.4f} {p['flops']/p['ms']/1e9:7.2f} {p['bytes']/p['ms']/1e6:7.0f}  {p['name'][:40]}")
m.Destroy()
