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
    return re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b([A-Z][A-Za-z0-9]+)\s*\([^;{]*\)\s*;", txt, flags=re.M)


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


def test_config_json_is_parsed_not_scraped(tmp_path):
    """The engine reads config.json once with a real JSON parser and only honours TOP-LEVEL keys: a "gpus" inside a nested object,
    inside a string, or in an array element must not switch anything on (round 1 regex-scraped the raw text)."""
    from gpu_ai_inference_server_amd.modelgen import models
    cfg = r'''{
      "name": "decoy", "platform": "onnxruntime_onnx", "version": "1",
      "inputs": [{"name": "x", "dims": [64, 1, 1], "shape": [4, 64, 1, 1], "data_type": "FLOAT32", "gpus": 8, "precision": "fp16"}],
      "outputs": [{"name": "y", "dims": [10], "shape": [4, 10], "data_type": "FLOAT32", "label_filename": "labels \"gpus\": 7 .txt"}],
      "notes": "\"gpus\": 6, \"precision\": \"fp16\", \"dynamic_batching\": true, \"max_batch_size\": 64",
      "nested": {"gpus": 5, "instance_count": 9, "uint8_scale": 3.0, "dynamic_batching": true, "max_batch_size": 32, "fp32_split": true},
      "uint8_bias": -1.5e0, "instance_count": 2, "tune_batches": [1, 8], "unicode": "é😀"
    }'''
    path = models.write_repo(str(tmp_path), "decoy", models.gemm_mlp("N"), config_json=cfg)
    c = B.DescribeModel(path)["config"]
    assert c["present"] and (c["name"], c["platform"], c["version"]) == ("decoy", "onnxruntime_onnx", "1")
    assert c["gpus"] == 0 and c["precision"] == "" and not c["dynamic_batching"] and c["max_batch_size"] == 0
    assert abs(c["uint8_scale"] - 1 / 255) < 1e-7 and c["uint8_bias"] == -1.5 and c["instance_count"] == 2 and c["tune_batches"] == [1, 8]
    assert c["fp32_split"] is False                       # the nested one does not count
    assert c["inputs"] == [{"name": "x", "data_type": "FLOAT32", "label_filename": "", "dims": [64, 1, 1], "shape": [4, 64, 1, 1]}]
    assert c["outputs"][0]["label_filename"] == 'labels "gpus": 7 .txt' and c["outputs"][0]["shape"] == [4, 10]
    # top-level keys are honoured
    path2 = models.write_repo(str(tmp_path), "real", models.gemm_mlp("N"),
                              config_json='{"gpus": 4, "precision": "FP16", "dynamic_batching": true, "max_batch_size": 16, "batch_window_us": 50, "fp32_split": true}')
    c2 = B.DescribeModel(path2)["config"]
    assert (c2["gpus"], c2["precision"], c2["dynamic_batching"], c2["max_batch_size"], c2["batch_window_us"]) == (4, "fp16", True, 16, 50)
    assert c2["fp32_split"] is True
    # numbers far outside the integer range are clamped, never cast blindly: 1e30 -> 2^31 - 1 for counts, 2^31 in lists
    path4 = models.write_repo(str(tmp_path), "huge", models.gemm_mlp("N"), config_json='{"gpus": 1e30, "instance_count": -1e30, "max_batch_size": 1e300, "tune_batches": [1e30, 4, -1e30]}')
    c4 = B.DescribeModel(path4)["config"]
    assert c4["gpus"] == 2147483647 and c4["instance_count"] == -2147483648 and c4["max_batch_size"] == 2147483647 and c4["tune_batches"] == [2147483648, 4, -2147483648]
    # no file: defaults; malformed file: an error that names the position, not a silent default
    path3 = models.write_repo(str(tmp_path), "nocfg", models.gemm_mlp("N"))
    assert B.DescribeModel(path3)["config"]["present"] is False
    for bad in ('{"gpus": 2,}', '{"gpus": 2} trailing', '{"a": "unterminated}', '[1, 2]', '{"x": 01}'):
        pbad = models.write_repo(str(tmp_path), "bad", models.gemm_mlp("N"), config_json=bad)
        with pytest.raises(RuntimeError, match="config.json parse error"):
            B.DescribeModel(pbad)


def test_version_order_can_follow_the_go_server(tmp_path, monkeypatch):
    """model_repository.cpp:45-53 sorts versions numerically ("10" is the latest of 1, 2, 10); the Go server's loadModelConfig sorts the
    same names as strings (main.go:640-655) and reads version "2"'s config.json.  IE_VERSION_ORDER=go makes the engine pick what Go picks."""
    from gpu_ai_inference_server_amd.modelgen import models
    root = str(tmp_path / "repo")
    for v in ("1", "2", "10"):
        models.write_repo(root, "m", models.test_model(), version=v)
    mgr = B.NewInferenceManager(root)
    try:
        with pytest.raises(RuntimeError) as e:        # no GPU here: the load fails AFTER the path was resolved; the message names nothing
            mgr.LoadModel("m", "7")
        assert "Model path not found" in str(e.value)
    finally:
        mgr.Shutdown()
    import ctypes as C2
    lib = B.lib()
    # resolve through the C ABI's own error text: an unknown explicit version names the directory it looked for
    for order, want in ((None, "10"), ("go", "2")):
        if order:
            monkeypatch.setenv("IE_VERSION_ORDER", order)
        else:
            monkeypatch.delenv("IE_VERSION_ORDER", raising=False)
        h = lib.InferenceInitialize(root.encode())
        try:
            err = C2.c_void_p()
            ok = lib.InferenceLoadModel(h, b"m", None, C2.byref(err))
            msg = C2.string_at(err.value).decode() if err.value else ""
            if err.value:
                lib.FreeErrorMessage(err)
            # on the CPU-only container the load fails at the device check, after version resolution; on a GPU box it succeeds
            if ok:
                mh = lib.GetModelHandle(h, b"m", None, None)
                md = lib.ModelGetMetadata(mh)
                got = C2.cast(md, C2.POINTER(C2.c_char_p))[1].decode()
                lib.ModelFreeMetadata(md)
                lib.ModelDestroy(mh)
                assert got == want
            else:
                assert "HIP device" in msg or "no CPU fallback" in msg, msg
        finally:
            lib.InferenceShutdown(h)
