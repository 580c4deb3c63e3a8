#!/usr/bin/env python3
"""fp32 MFMA calibration: TFLOP/s of a register-resident v_mfma_f32_32x32x2_f32 loop for 1/2/4 independent accumulator chains."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _pkg import load_package  # noqa: E402

load_package()
from gpu_ai_inference_server_amd import binding as B  # noqa: E402

for nacc in (1, 2, 4):
    for bpc in (1, 2):
        print(f"nacc={nacc} blocks_per_cu={bpc}: {B.MfmaPeak(nacc, bpc, 20000):.1f} TFLOP/s")
