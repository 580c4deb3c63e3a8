"""CPU: the C-ABI library loads, exports every symbol the headers declare, has the reference's struct layout,
keeps the reference's error strings, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import MINI, ROOT
from gpu_ai_inference_server_amd import binding as B
from gpu_ai_inference_server_amd import build

HDR = os.path.join(ROOT, "include")


def _declared(header):
    txt = open(os.path.join(HDR, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b([A-Z][A-Za-z]+)\s*\([^;{]*\)\s*;", txt, flags=re.M)


def test_exports_every_declared_symbol(engine_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", engine_lib], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    bridge = _declared("inference_bridge.h")
    ext = _declared("inference_engine_ext.h")
    assert len(bridge) == 23 and set(bridge) == set(B.ABI_SYMBOLS)      # the reference's 21 + ModelLoad/ModelUnload
    assert set(ext) == set(B.EXT_SYMBOLS)
    missing = (set(bridge) | set(ext)) - exported
    assert not missing, missing
    B.lib()                                                             # ctypes binds all of them


def test_struct_layout_matches_reference_header():
    """sizes/offsets the Go binding depends on (measured from the reference header with gcc)."""
    assert C.sizeof(B.CShape) == 16
    assert C.sizeof(B.CTensorData) == 48 and B.CTensorData.data.offset == 32 and B.CTensorData.data_size.offset == 40
    assert B.CTensorData.data_type.offset == 8 and B.CTensorData.shape.offset == 16
    assert C.sizeof(B.CModelConfig) == 64
    assert [getattr(B.CModelConfig, f).offset for f in ("type_", "max_batch_size", "input_names", "num_inputs", "output_names",
                                                        "num_outputs", "instance_count", "dynamic_batching")] == [16, 20, 24, 32, 40, 48, 52, 56]
    assert C.sizeof(B.CModelMetadata) == 72 and B.CModelMetadata.load_time_ns.offset == 64
    assert C.sizeof(B.CModelStats) == 32 and C.sizeof(B.CCudaMemoryInfo) == 24


def test_c_header_compiles_as_c_and_layout(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "inference_bridge.h"\n#include "inference_engine_ext.h"\n#include <stdio.h>\n#include <stddef.h>\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %d %d %d",sizeof(Shape),sizeof(TensorData),sizeof(ModelConfig),'
                   'sizeof(ModelMetadata),sizeof(ModelStats),sizeof(CudaMemoryInfo),(int)DATATYPE_UNKNOWN,(int)DEVICE_GPU,(int)MODEL_CUSTOM);return 0;}\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", HDR, str(src), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout == "16 48 64 72 32 24 8 1 5"


def test_null_handle_error_strings(engine_lib):
    L = B.lib()
    err = C.c_void_p()
    assert not L.InferenceLoadModel(None, b"x", None, C.byref(err))
    assert B._take_error(err) == "Invalid handle or model name"          # bridge:301
    err = C.c_void_p()
    assert not L.ModelInfer(None, None, 0, None, 0, C.byref(err))
    assert B._take_error(err) == "Invalid model handle"                  # bridge:697
    err = C.c_void_p()
    assert not L.GetModelHandle(None, b"x", None, C.byref(err))
    assert B._take_error(err) == "Invalid handle or model name"
    assert not L.ModelIsLoaded(None) and not L.InferenceIsModelLoaded(None, b"x", None)
    assert not L.ModelGetStats(None) and not L.ModelGetMetadata(None)
    L.ModelDestroy(None); L.InferenceShutdown(None); L.FreeErrorMessage(None); L.ModelFreeStats(None); L.ModelFreeMetadata(None)
    # error pointer may be NULL (Go always passes one, C callers need not)
    assert not L.InferenceLoadModel(None, b"x", None, None)


def test_repository_and_manager_semantics(model_repo, tmp_path):
    # version ordering: numeric-descending -> latest of {1,2,10} is 10 (model_repository.cpp:45-53)
    root = tmp_path / "repo"
    for v in ("1", "2", "10"):
        d = root / "m" / v
        d.mkdir(parents=True)
        (d / "config.json").write_text("{}")
    (root / "empty_model").mkdir()
    (root / "not_a_dir.txt").write_text("x")
    mgr = B.NewInferenceManager(str(root))
    assert mgr.ListModels() == ["m"]                                   # only models with >= 1 valid version
    with pytest.raises(RuntimeError, match=r"ONNX file not found at: .*/m/10/model.onnx"):
        mgr.LoadModel("m")                                             # latest = "10", and it has no model.onnx
    with pytest.raises(RuntimeError, match="Model path not found: "):
        mgr.LoadModel("nope")
    with pytest.raises(RuntimeError, match="Model path not found: "):
        mgr.LoadModel("m", "3")
    with pytest.raises(RuntimeError, match="Model not found"):
        mgr.UnloadModel("m")
    assert not mgr.IsModelLoaded("m")
    with pytest.raises(RuntimeError, match="is not loaded"):
        mgr.GetModel("m")
    mgr.Shutdown()
    # InferenceInitialize creates a missing repository directory (model_repository.cpp:10-16)
    newroot = tmp_path / "fresh" / "models"
    m2 = B.NewInferenceManager(str(newroot))
    assert newroot.is_dir() and m2.ListModels() == []
    m2.Shutdown()
    m3 = B.NewInferenceManager(model_repo)
    assert m3.ListModels() == sorted(["test_model", "mini_two_input"] + list(MINI))
    m3.Shutdown()


def test_model_create_without_load_and_stub_backends(model_repo):
    m = B.CreateModel(os.path.join(model_repo, "test_model", "1"), "test_model", load=False, input_names=["input"], output_names=["output"])
    assert not B.lib().ModelIsLoaded(m.handle)
    md = m.GetMetadata()
    assert (md.Name, md.Version, md.Type, md.Inputs, md.Outputs, md.LoadTimeNs) == ("test_model", "1", B.ModelONNX, ["input"], ["output"], 0)
    st = m.GetStats()
    assert (st.InferenceCount, st.TotalInferenceTimeNs, st.MemoryUsageBytes) == (0, 0, 0)
    with pytest.raises(RuntimeError, match="model not loaded"):
        m.Infer([B.TensorData("input", B.DataTypeFloat32, B.Shape([1, 3]), np.ones(3, np.float32))], [B.OutputConfig("output", [1, 2])])
    m.Destroy()
    for t, msg in ((B.ModelTensorFlow, "TensorFlow model loading not implemented"), (B.ModelTensorRT, "TensorRT model loading not implemented"),
                   (B.ModelPyTorch, "PyTorch model loading not implemented"), (B.ModelCustom, "Custom model loading not implemented"),
                   (B.ModelUnknown, "Unsupported model type")):
        with pytest.raises(RuntimeError, match=msg):
            B.CreateModel(os.path.join(model_repo, "test_model", "1"), "x", model_type=t)
    with pytest.raises(RuntimeError, match="Model file not found: /nonexistent"):
        B.CreateModel("/nonexistent", "x")
    with pytest.raises(RuntimeError, match="DEVICE_CPU execution is not provided"):
        B.CreateModel(os.path.join(model_repo, "test_model", "1"), "x", device=B.DeviceCPU)


def test_no_gpu_means_loud_failure_not_fallback(model_repo, engine_lib):
    """On a box without a HIP device the product must refuse to load (it never routes through a CPU path)."""
    if B.IsCUDAAvailable():
        pytest.skip("a GPU is visible here")
    assert B.GetDeviceCount() == 0 and B.GetDeviceInfo(0) == "Unknown device"
    with pytest.raises(RuntimeError, match="failed to get memory information"):
        B.GetMemoryInfo(0)
    mgr = B.NewInferenceManager(model_repo)
    with pytest.raises(RuntimeError, match="No HIP device available: the MI355X engine has no CPU fallback"):
        mgr.LoadModel("test_model")
    assert not mgr.IsModelLoaded("test_model")
    mgr.Shutdown()
    harness = build.build_harness()
    r = subprocess.run([harness, model_repo, "test_model", "input", "output", "2", "1", "3", "--no-gpu"], capture_output=True, text=True)
    assert r.returncode == 0 and "CALL InferenceLoadModel -> 0 error=" in r.stdout, r.stdout + r.stderr
    with pytest.raises(RuntimeError, match="No HIP device available"):
        B.VectorAdd(np.ones(4, np.float32), np.ones(4, np.float32))
