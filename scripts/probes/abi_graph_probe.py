#!/usr/bin/env python3
"""Pure-ctypes probe over the 21 reference ABI symbols only (works against any build of libinference_engine.so):
load DenseNet-121 through InferenceLoadModel, call ModelInfer a few times (graph replays), shut down.

    [rocprofv3 --kernel-trace --stats -d <dir> --] python3 scripts/probes/abi_graph_probe.py <path/to/libinference_engine.so> [batch]

Used to bisect the round-1 observation "hipGraphLaunch under rocprofv3 --kernel-trace ends in SIGSEGV".
"""
import ctypes as C
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
from gpu_ai_inference_server_amd.modelgen import models  # noqa: E402

lib = C.CDLL(sys.argv[1])
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
root = "/tmp/ie_probe_repo"
models.write_repo(root, "densenet_onnx", models.densenet121("N"))


class Shape(C.Structure):
    _fields_ = [("dims", C.POINTER(C.c_int64)), ("num_dims", C.c_int)]


class TensorData(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data_type", C.c_int), ("shape", Shape), ("data", C.c_void_p), ("data_size", C.c_size_t)]


lib.InferenceInitialize.restype = C.c_void_p
lib.InferenceInitialize.argtypes = [C.c_char_p]
lib.InferenceLoadModel.restype = C.c_bool
lib.InferenceLoadModel.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
lib.GetModelHandle.restype = C.c_void_p
lib.GetModelHandle.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
lib.ModelInfer.restype = C.c_bool
lib.ModelInfer.argtypes = [C.c_void_p, C.POINTER(TensorData), C.c_int, C.POINTER(TensorData), C.c_int, C.POINTER(C.c_void_p)]
lib.ModelDestroy.argtypes = [C.c_void_p]
lib.InferenceShutdown.argtypes = [C.c_void_p]
print("initialize + load", flush=True)
mgr = lib.InferenceInitialize(root.encode())
err = C.c_void_p()
assert lib.InferenceLoadModel(mgr, b"densenet_onnx", None, C.byref(err)), C.string_at(err.value)
h = lib.GetModelHandle(mgr, b"densenet_onnx", None, C.byref(err))
x = models.synthetic_input((batch, 3, 224, 224), stream="probe")
y = np.empty((batch, 1000), np.float32)
idims = (C.c_int64 * 4)(batch, 3, 224, 224)
odims = (C.c_int64 * 4)(batch, 1000, 1, 1)
tin = TensorData(b"data_0", 0, Shape(idims, 4), x.ctypes.data, x.nbytes)
tout = TensorData(b"fc6_1", 0, Shape(odims, 4), y.ctypes.data, y.nbytes)
for i in range(4):
    print("ModelInfer", i, flush=True)
    assert lib.ModelInfer(h, C.byref(tin), 1, C.byref(tout), 1, C.byref(err)), C.string_at(err.value)
print("finite:", bool(np.isfinite(y).all()), flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "burst":
    # what bench.py does: EnginePrepare, then 60 back-to-back hipGraphLaunch calls without a sync in between
    lib.EnginePrepare.restype = C.c_bool
    lib.EnginePrepare.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
    lib.EngineRunPrepared.restype = C.c_bool
    lib.EngineRunPrepared.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    din, dout = (C.c_void_p * 1)(), (C.c_void_p * 1)()
    shp = Shape(idims, 4)
    assert lib.EnginePrepare(h, C.byref(shp), 1, din, dout, 1, C.byref(err)), C.string_at(err.value)
    for rep in range(3):
        print("EngineRunPrepared burst", rep, flush=True)
        assert lib.EngineRunPrepared(h, 60, 1, C.byref(err)), C.string_at(err.value)
lib.ModelDestroy(h)
lib.InferenceShutdown(mgr)
print("PROBE_OK", flush=True)
