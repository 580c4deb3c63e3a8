#!/usr/bin/env python3
"""Kernel-quality probe: one 1x1 conv = plain GEMM [M=H*W*B, K=Cin] x [K, N=Cout] through the engine (autotuned)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _pkg import load_package
load_package()
import numpy as np
from gpu_ai_inference_server_amd import binding as B
from gpu_ai_inference_server_amd.modelgen import onnx_pb as pb

def run(M_hw, cin, cout, k=1, batch=1):
    w = (np.random.RandomState(0).randn(cout, cin, k, k) * 0.02).astype(np.float32)
    w0 = (np.random.RandomState(1).randn(cin, 4, 1, 1) * 0.5).astype(np.float32)
    n0 = pb.node("Conv", ["x", "w0"], ["h"], "c0", [pb.attr_ints("kernel_shape", [1, 1]), pb.attr_ints("pads", [0] * 4), pb.attr_ints("strides", [1, 1])])
    n = pb.node("Conv", ["h", "w"], ["y"], "c", [pb.attr_ints("kernel_shape", [k, k]), pb.attr_ints("pads", [k // 2] * 4), pb.attr_ints("strides", [1, 1])])
    g = pb.graph("g", [n0, n], [pb.tensor("w0", w0), pb.tensor("w", w)], [pb.value_info("x", [batch, 4, M_hw, M_hw])], [pb.value_info("y", [batch, cout, M_hw, M_hw])])
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "m", "1"))
    open(os.path.join(d, "m", "1", "model.onnx"), "wb").write(pb.model(g))
    m = B.CreateModel(os.path.join(d, "m", "1"), "m")
    B.Prepare(m, [[batch, 4, M_hw, M_hw]], 1)
    B.RunPrepared(m, 3, True)
    prof = B.Profile(m, 5)
    for p in prof:
        if p["name"] == "c":
            print(f"M={batch*M_hw*M_hw} K={cin*k*k} N={cout}: {p['kernel']:40s} {p['ms']:.4f} ms  {p['flops']/p['ms']/1e9:.1f} TFLOP/s")
    m.Destroy()

for args in [(128, 4096, 4096), (128, 1024, 1024), (128, 512, 128), (128, 256, 128), (64, 128, 32, 3, 8)]:
    run(*args)
