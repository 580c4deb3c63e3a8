#!/usr/bin/env python3
"""One-off robustness sweep (not part of the test suite): many seeded random conv graphs through every algorithm choice, both
precisions, against the float64 oracle.  usage: stress_random_convs.py [first_seed] [num_seeds]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _pkg import load_package  # noqa: E402

load_package()
import numpy as np  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from gpu_ai_inference_server_amd import binding as B  # noqa: E402
from gpu_ai_inference_server_amd.modelgen import models  # noqa: E402
from oracle import onnx_oracle as O  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
modes32 = [dict(), dict(IE_AUTOTUNE="0"), dict(IE_FORCE_ALGO="raster"), dict(IE_FORCE_ALGO="igemm", IE_FORCE_TILE="8"),
           dict(IE_FORCE_ALGO="scalar"), dict(IE_FORCE_ALGO="igemm", IE_FORCE_SPLITK="3", IE_FORCE_TILE="3"),
           dict(IE_FORCE_ALGO="ws", IE_FORCE_TILE="2"), dict(IE_FORCE_ALGO="ws", IE_FORCE_TILE="12"), dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE="0"),
           dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE="4"), dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE="6"), dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE="9")]
modes16 = [dict(), dict(IE_AUTOTUNE="0"), dict(IE_FORCE_ALGO="igemm", IE_FORCE_TILE="9"), dict(IE_FORCE_ALGO="ws", IE_FORCE_TILE="0"),
           dict(IE_FORCE_ALGO="ws", IE_FORCE_TILE="5"), dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE="1"), dict(IE_FORCE_ALGO="naive")]
worst = {"fp32": 0.0, "fp16": 0.0}
n = 0
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(first, first + count):
        rs = np.random.RandomState(seed)
        for case in range(6):
            for prec, modes, cins, tol in (("fp32", modes32, (3, 4, 8, 12, 16, 20, 32, 48, 64, 96, 128), T.RTOL),
                                           ("fp16", modes16, (8, 16, 24, 32, 64, 72, 96, 128, 160), T.F16_RTOL)):
                mb, ishape, oshape, desc = T._random_conv_graph(rs, case, cin_choices=cins)
                d = models.write_repo(tmp, f"s{seed}_{case}_{prec}", mb)
                x = rs.rand(*ishape).astype(np.float32)
                ref = O.run(O.load_model(mb), {"x": x}, dtype=np.float64)["out"]
                env = dict(modes[(seed + case) % len(modes)], IE_PRECISION=prec)
                os.environ.update(env)
                try:
                    m = B.CreateModel(d, "s")
                    try:
                        y, dims = T.infer(m, "", "x", x, "out", oshape)
                    finally:
                        m.Destroy()
                finally:
                    for k in env:
                        os.environ.pop(k, None)
                e = T.rel_err(y, ref)
                worst[prec] = max(worst[prec], e)
                n += 1
                if not (dims == list(oshape) and e < tol):
                    print("FAIL", seed, case, prec, env, desc, e)
                    sys.exit(1)
print(f"{n} random conv graphs ok; worst rel err fp32 {worst['fp32']:.2e}, fp16 {worst['fp16']:.2e}")
