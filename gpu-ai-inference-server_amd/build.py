"""hipcc build recipe for libinference_engine.so (gfx950 only, built in-tree so it travels to the GPU box).

    python gpu-ai-inference-server_amd/build.py [--force]

Output: gpu-ai-inference-server_amd/lib/libinference_engine.so, plus copies at the two places the reference's cgo
binding and scripts look for the library (inference_binding.go:7 `-L${SRCDIR}/../../build/inference_engine`,
scripts/run_server.sh:6 `build/inference_engine/lib`) under ./build/ (git-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "lib", "obj")
LIB = os.path.join(PKG, "lib", "libinference_engine.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = ["env.cpp", "onnx_reader.cpp", "plan.cpp", "repository.cpp", "config.cpp", "executor.cpp", "bridge.cpp", "bridge_load.cpp", "bridge_run.cpp", "kernels.hip", "kernels_f16.hip", "kernels_ws.hip", "kernels_ws32.hip", "kernels_stem.hip", "kernels_direct.hip", "kernels_f8.hip", "kernels_fused.hip", "kernels_wino.hip", "kernels_x6.hip", "kernels_block.hip", "kernels_ws8.hip"]
COMMON = ["-std=c++17", "-O3", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-unused-result",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _deps(src: str) -> list[str]:
    d = [os.path.join(CSRC, src)]
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            d.append(os.path.join(CSRC, f))
    for f in os.listdir(os.path.join(ROOT, "include")):
        d.append(os.path.join(ROOT, "include", f))
    return d


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, src + ".o")
    if force or _stale(obj, _deps(src)):
        cmd = [HIPCC, *COMMON, "-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".hip"):
            # MFMA accumulators in ordinary VGPRs: for the 256-thread kernels the compiler otherwise parks them in AGPRs (a v_accvgpr_read / _write per
            # element in every epilogue, and 170 + 64 registers instead of 170: one wave per SIMD fewer); scripts/isa_regs.py tabulates both builds
            cmd[1:1] = ["--offload-arch=" + ARCH, "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", LIB, "-lstdc++fs", "-lpthread",
               "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        for d in (os.path.join(ROOT, "build", "inference_engine"), os.path.join(ROOT, "build", "inference_engine", "lib")):
            os.makedirs(d, exist_ok=True)
            shutil.copy2(LIB, os.path.join(d, "libinference_engine.so"))
    return LIB


def build_harness() -> str:
    """C program replaying the Go binding's call sequence (csrc/harness/replay_binding.c)."""
    src = os.path.join(CSRC, "harness", "replay_binding.c")
    out = os.path.join(PKG, "lib", "replay_binding")
    if _stale(out, [src, LIB, os.path.join(ROOT, "include", "inference_bridge.h")]):
        cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
               "-L" + os.path.dirname(LIB), "-linference_engine", "-Wl,-rpath," + os.path.dirname(LIB), "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"harness build failed:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build_library("--force" in sys.argv))
    if os.path.exists(os.path.join(CSRC, "harness", "replay_binding.c")):
        print(build_harness())
