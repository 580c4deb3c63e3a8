#!/usr/bin/env python3
"""One graph-mode forward under a profiler: which captured graph (if any) makes `rocprofv3 --kernel-trace` fall over?

    rocprofv3 --kernel-trace --stats -d <dir> -- python3 scripts/probes/graph_trace_probe.py {mini|densenet|densenet_nofuse} [batch]
"""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _pkg import load_package  # noqa: E402

load_package()
import numpy as np  # noqa: E402
from gpu_ai_inference_server_amd import binding as B  # noqa: E402
from gpu_ai_inference_server_amd.modelgen import models  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "mini"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
root = "/tmp/ie_probe_models"
if which == "mini":
    path = models.write_repo(root, "mini", models.densenet("N", growth=8, blocks=(2, 3), stem=16, image=32, classes=10, seed=5))
    shape = [batch, 3, 32, 32]
else:
    path = models.write_repo(root, "densenet", models.densenet121("N"))
    shape = [batch, 3, 224, 224]
print("loading", which, flush=True)
m = B.CreateModel(path, which)
print("prepare (capture + instantiate)", flush=True)
din, dout = B.Prepare(m, [shape], 1)
B.CopyToDevice(m, din[0], models.synthetic_input(shape, stream="probe"))
print("graph launch", flush=True)
B.RunPrepared(m, 3, True)
print("graph launch done", flush=True)
m.Destroy()
print("PROBE_OK", flush=True)
