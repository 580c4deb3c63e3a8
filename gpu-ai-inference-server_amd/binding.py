"""ctypes mirror of the reference's Go cgo binding (inference_engine/binding/inference_binding.go).

Go is not available in the build image, so the host side above the C ABI is restated here in Python with the
binding's own names, argument meaning and error behaviour: `InferenceManager` (NewInferenceManager :177,
LoadModel :227, UnloadModel :292, IsModelLoaded :340, ListModels :361, GetModel :387, RunInference :433,
Shutdown :195), `Model.Infer` :521, `GetMetadata` :739, `GetStats` :780, and the device queries :134-175.
It marshals exactly like the Go code: malloc'd C copies of every input payload and dims array, malloc'd
*uninitialised* output buffers sized from `OutputConfig.Shape` (falling back to `Dims`), FLOAT32 only.

The module fails loudly when libinference_engine.so is missing: there is no pure-Python fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass, field
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INFERENCE_ENGINE_LIB", os.path.join(_HERE, "lib", "libinference_engine.so"))

# enums (inference_binding.go:19-50)
DataTypeFloat32, DataTypeInt32, DataTypeInt64, DataTypeUint8, DataTypeInt8, DataTypeString, DataTypeBool, DataTypeFp16, \
    DataTypeUnknown = range(9)
DeviceCPU, DeviceGPU = 0, 1
ModelUnknown, ModelTensorFlow, ModelTensorRT, ModelONNX, ModelPyTorch, ModelCustom = range(6)


class CShape(C.Structure):
    _fields_ = [("dims", C.POINTER(C.c_int64)), ("num_dims", C.c_int)]


class CTensorData(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data_type", C.c_int), ("shape", CShape), ("data", C.c_void_p),
                ("data_size", C.c_size_t)]


class CModelConfig(C.Structure):
    _fields_ = [("name", C.c_char_p), ("version", C.c_char_p), ("type_", C.c_int), ("max_batch_size", C.c_int),
                ("input_names", C.POINTER(C.c_char_p)), ("num_inputs", C.c_int),
                ("output_names", C.POINTER(C.c_char_p)), ("num_outputs", C.c_int),
                ("instance_count", C.c_int), ("dynamic_batching", C.c_bool)]


class CModelMetadata(C.Structure):
    _fields_ = [("name", C.c_char_p), ("version", C.c_char_p), ("model_type", C.c_int),
                ("inputs", C.POINTER(C.c_char_p)), ("num_inputs", C.c_int),
                ("outputs", C.POINTER(C.c_char_p)), ("num_outputs", C.c_int),
                ("description", C.c_char_p), ("load_time_ns", C.c_int64)]


class CModelStats(C.Structure):
    _fields_ = [("inference_count", C.c_int64), ("total_inference_time_ns", C.c_int64),
                ("last_inference_time_ns", C.c_int64), ("memory_usage_bytes", C.c_size_t)]


class CCudaMemoryInfo(C.Structure):
    _fields_ = [("total", C.c_size_t), ("free", C.c_size_t), ("used", C.c_size_t)]


ABI_SYMBOLS = [
    "IsCudaAvailable", "GetDeviceCount", "GetDeviceInfo", "GetMemoryInfo", "InferenceInitialize", "InferenceShutdown",
    "InferenceLoadModel", "InferenceUnloadModel", "InferenceIsModelLoaded", "InferenceListModels",
    "InferenceFreeModelList", "ModelCreate", "ModelDestroy", "ModelIsLoaded", "ModelInfer", "ModelGetMetadata",
    "ModelFreeMetadata", "ModelGetStats", "ModelFreeStats", "FreeErrorMessage", "GetModelHandle",
    "ModelLoad", "ModelUnload",
]
EXT_SYMBOLS = ["EngineDescribeModel", "EnginePrepare", "EngineRunPrepared", "EngineSynchronize", "EngineGetStream",
               "EngineProfile", "EngineGetWeightBlob", "EngineWeightsUpdated", "EngineGetPrecision", "EngineMemcpy", "EngineMfmaPeak", "EngineGetBatcherStats", "EngineGetShardStats", "EngineGetRuntimeInfo", "EngineE4m3RoundTrip", "EnginePlanWeights", "EngineVectorAdd"]

_lib = None
_lib_lock = threading.Lock()
_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]
_libc.free.argtypes = [C.c_void_p]


def lib() -> C.CDLL:
    """Load libinference_engine.so and declare the prototypes of include/inference_bridge.h."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python gpu-ai-inference-server_amd/build.py` "
                               "(there is no fallback implementation)")
        L = C.CDLL(LIB_PATH)
        vp, cp, ep = C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)
        sig = {
            "IsCudaAvailable": (C.c_bool, []), "GetDeviceCount": (C.c_int, []),
            "GetDeviceInfo": (vp, [C.c_int]), "GetMemoryInfo": (CCudaMemoryInfo, [C.c_int]),
            "InferenceInitialize": (vp, [cp]), "InferenceShutdown": (None, [vp]),
            "InferenceLoadModel": (C.c_bool, [vp, cp, cp, ep]), "InferenceUnloadModel": (C.c_bool, [vp, cp, cp, ep]),
            "InferenceIsModelLoaded": (C.c_bool, [vp, cp, cp]),
            "InferenceListModels": (C.POINTER(vp), [vp, C.POINTER(C.c_int)]),
            "InferenceFreeModelList": (None, [C.POINTER(vp), C.c_int]),
            "ModelCreate": (vp, [cp, C.c_int, C.POINTER(CModelConfig), C.c_int, C.c_int, ep]),
            "ModelDestroy": (None, [vp]), "ModelIsLoaded": (C.c_bool, [vp]),
            "ModelInfer": (C.c_bool, [vp, C.POINTER(CTensorData), C.c_int, C.POINTER(CTensorData), C.c_int, ep]),
            "ModelGetMetadata": (C.POINTER(CModelMetadata), [vp]), "ModelFreeMetadata": (None, [C.POINTER(CModelMetadata)]),
            "ModelGetStats": (C.POINTER(CModelStats), [vp]), "ModelFreeStats": (None, [C.POINTER(CModelStats)]),
            "FreeErrorMessage": (None, [vp]), "GetModelHandle": (vp, [vp, cp, cp, ep]),
            "ModelLoad": (C.c_bool, [vp, ep]), "ModelUnload": (C.c_bool, [vp, ep]),
            "EngineDescribeModel": (vp, [cp, C.c_int, ep]),
            "EnginePrepare": (C.c_bool, [vp, C.POINTER(CShape), C.c_int, ep, ep, C.c_int, ep]),
            "EngineRunPrepared": (C.c_bool, [vp, C.c_int, C.c_int, ep]), "EngineSynchronize": (C.c_bool, [vp, ep]),
            "EngineGetStream": (vp, [vp]), "EngineProfile": (vp, [vp, C.c_int, ep]),
            "EngineGetWeightBlob": (C.c_bool, [vp, ep, C.POINTER(C.c_size_t), ep]),
            "EngineWeightsUpdated": (C.c_bool, [vp, ep]), "EngineGetPrecision": (C.c_int, [vp]),
            "EngineVectorAdd": (C.c_bool, [vp, vp, vp, C.c_size_t, ep]),
            "EngineMemcpy": (C.c_bool, [vp, vp, vp, C.c_size_t, C.c_int, ep]),
            "EngineMfmaPeak": (C.c_double, [C.c_int, C.c_int, C.c_int]),
            "EngineGetBatcherStats": (C.c_bool, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
            "EngineGetShardStats": (C.c_bool, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
            "EngineGetRuntimeInfo": (C.c_void_p, [vp, C.c_int, C.POINTER(C.c_void_p)]),
            "EnginePlanWeights": (C.c_void_p, [cp, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]),
            "EngineE4m3RoundTrip": (C.c_bool, [vp, vp, vp, C.c_size_t, C.c_float, C.POINTER(C.c_void_p)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)   # AttributeError here = symbol missing from the .so
            fn.restype = res
            fn.argtypes = args
        _lib = L
        return L


def _take_error(err: C.c_void_p) -> str:
    if not err.value:
        return "unknown error"
    msg = C.string_at(err.value).decode("utf-8", "replace")
    lib().FreeErrorMessage(err)
    return msg


def _take_string(p) -> str:
    s = C.string_at(p).decode("utf-8", "replace")
    lib().FreeErrorMessage(p)
    return s


# ---- device queries (inference_binding.go:134-175) --------------------------------------------------------
def IsCUDAAvailable() -> bool:
    return bool(lib().IsCudaAvailable())


def GetDeviceCount() -> int:
    return int(lib().GetDeviceCount())


def GetDeviceInfo(device_id: int) -> str:
    p = lib().GetDeviceInfo(device_id)
    s = C.string_at(p).decode()
    _libc.free(p)                     # the Go side calls C.free directly (:146)
    return s


@dataclass
class MemoryInfo:
    Total: int
    Free: int
    Used: int


def GetMemoryInfo(device_id: int) -> MemoryInfo:
    m = lib().GetMemoryInfo(device_id)
    if m.total == 0:
        raise RuntimeError("failed to get memory information")   # :165-167
    return MemoryInfo(int(m.total), int(m.free), int(m.used))


@dataclass
class Shape:
    Dims: list


@dataclass
class TensorData:
    Name: str
    DataType: int
    Shape: Shape
    Data: object = None


@dataclass
class OutputConfig:
    Name: str
    Shape: list = field(default_factory=list)
    Dims: list = field(default_factory=list)
    DataType: str = "FLOAT32"
    LabelFilename: str = ""


@dataclass
class ModelMetadata:
    Name: str
    Version: str
    Type: int
    Inputs: list
    Outputs: list
    Description: str
    LoadTimeNs: int


@dataclass
class ModelStats:
    InferenceCount: int
    TotalInferenceTimeNs: int
    LastInferenceTimeNs: int
    MemoryUsageBytes: int


class Model:
    """A model handle (inference_binding.go:106-111)."""

    def __init__(self, handle, name: str = "", version: str = ""):
        self.handle = handle
        self.name = name
        self.version = version

    def Infer(self, inputs: Sequence[TensorData], outputConfigs: Sequence[OutputConfig]) -> list:
        L = lib()
        if not self.handle:
            raise RuntimeError("model not initialized")
        if not L.ModelIsLoaded(self.handle):
            raise RuntimeError("model not loaded")
        if len(inputs) == 0:
            raise RuntimeError("no input tensors provided")
        outs = []
        for oc in outputConfigs:
            shape = list(oc.Shape) or list(oc.Dims)
            if not shape:
                raise RuntimeError(f"no shape defined for output '{oc.Name}'")
            if oc.DataType not in ("FLOAT32", "TYPE_FP32"):
                raise RuntimeError(f"unsupported data type '{oc.DataType}' for output '{oc.Name}'")
            outs.append(TensorData(oc.Name, DataTypeFloat32, Shape(shape), np.zeros(int(np.prod(shape)), np.float32)))

        to_free = []

        def cmalloc(n):
            p = _libc.malloc(max(int(n), 1))
            if not p:
                raise MemoryError
            to_free.append(p)
            return p

        try:
            cin = (CTensorData * len(inputs))()
            cout = (CTensorData * max(len(outs), 1))()
            keep = []
            for i, t in enumerate(inputs):
                if t.DataType not in (DataTypeFloat32, DataTypeUint8):
                    raise RuntimeError(f"unsupported data type for input '{t.Name}'")
                arr = np.ascontiguousarray(t.Data, dtype=np.float32 if t.DataType == DataTypeFloat32 else np.uint8).ravel()
                nm = t.Name.encode()
                keep.append(nm)
                dims = list(t.Shape.Dims)
                dp = cmalloc(8 * len(dims))
                C.memmove(dp, (C.c_int64 * len(dims))(*dims), 8 * len(dims))
                buf = cmalloc(arr.nbytes)
                C.memmove(buf, arr.ctypes.data, arr.nbytes)                      # copy #1 (:607-651)
                cin[i].name = nm
                cin[i].data_type = t.DataType
                cin[i].shape.dims = C.cast(dp, C.POINTER(C.c_int64))
                cin[i].shape.num_dims = len(dims)
                cin[i].data = buf
                cin[i].data_size = arr.nbytes
            for i, t in enumerate(outs):
                nm = t.Name.encode()
                keep.append(nm)
                dims = list(t.Shape.Dims)
                dp = cmalloc(8 * len(dims))
                C.memmove(dp, (C.c_int64 * len(dims))(*dims), 8 * len(dims))
                nbytes = t.Data.nbytes
                buf = cmalloc(nbytes)                                            # uninitialised, like C.malloc (:654-695)
                cout[i].name = nm
                cout[i].data_type = DataTypeFloat32
                cout[i].shape.dims = C.cast(dp, C.POINTER(C.c_int64))
                cout[i].shape.num_dims = len(dims)
                cout[i].data = buf
                cout[i].data_size = nbytes
            err = C.c_void_p()
            ok = L.ModelInfer(self.handle, cin, len(inputs), cout, len(outs), C.byref(err))   # :699-704
            if not ok:
                raise RuntimeError("inference failed: " + _take_error(err))
            for i, t in enumerate(outs):
                C.memmove(t.Data.ctypes.data, cout[i].data, t.Data.nbytes)       # copy back (:717-731)
                nd = cout[i].shape.num_dims
                t.Shape = Shape([int(cout[i].shape.dims[k]) for k in range(nd)])
            return outs
        finally:
            for p in to_free:
                _libc.free(p)

    def InferTimed(self, inputs: Sequence[TensorData], outputConfigs: Sequence[OutputConfig], iters: int) -> list:
        """Seconds spent INSIDE the ModelInfer C call, per call, for `iters` calls.

        Every iteration marshals exactly like Infer() / the Go binding does (fresh C.malloc'ed payload and output buffers, payload
        copied in, buffers freed afterwards: inference_binding.go:590-734) but only the C-ABI call itself is timed, i.e. what the
        engine is responsible for; Infer()'s wall time adds the binding's own malloc + copy, which the build does not change."""
        import time
        L = lib()
        times = []
        for _ in range(iters):
            to_free = []
            try:
                cin = (CTensorData * len(inputs))()
                outs = [(oc.Name.encode(), list(oc.Shape) or list(oc.Dims)) for oc in outputConfigs]
                cout = (CTensorData * max(len(outs), 1))()
                keep = []
                for i, t in enumerate(inputs):
                    arr = np.ascontiguousarray(t.Data, dtype=np.float32 if t.DataType == DataTypeFloat32 else np.uint8).ravel()
                    nm = t.Name.encode()
                    dims = (C.c_int64 * len(t.Shape.Dims))(*t.Shape.Dims)
                    keep += [nm, dims]
                    buf = _libc.malloc(max(arr.nbytes, 1))
                    to_free.append(buf)
                    C.memmove(buf, arr.ctypes.data, arr.nbytes)
                    cin[i].name = nm
                    cin[i].data_type = t.DataType
                    cin[i].shape.dims = C.cast(dims, C.POINTER(C.c_int64))
                    cin[i].shape.num_dims = len(t.Shape.Dims)
                    cin[i].data = buf
                    cin[i].data_size = arr.nbytes
                for i, (nm, shape) in enumerate(outs):
                    dims = (C.c_int64 * len(shape))(*shape)
                    keep += [nm, dims]
                    nbytes = 4 * int(np.prod(shape))
                    buf = _libc.malloc(max(nbytes, 1))
                    to_free.append(buf)
                    cout[i].name = nm
                    cout[i].data_type = DataTypeFloat32
                    cout[i].shape.dims = C.cast(dims, C.POINTER(C.c_int64))
                    cout[i].shape.num_dims = len(shape)
                    cout[i].data = buf
                    cout[i].data_size = nbytes
                err = C.c_void_p()
                t0 = time.perf_counter()
                ok = L.ModelInfer(self.handle, cin, len(inputs), cout, len(outs), C.byref(err))
                times.append(time.perf_counter() - t0)
                if not ok:
                    raise RuntimeError("inference failed: " + _take_error(err))
            finally:
                for p_ in to_free:
                    _libc.free(p_)
        return times

    def GetMetadata(self) -> ModelMetadata:
        L = lib()
        p = L.ModelGetMetadata(self.handle)
        if not p:
            raise RuntimeError("failed to get model metadata")
        try:
            m = p.contents
            return ModelMetadata(m.name.decode(), m.version.decode(), int(m.model_type),
                                 [m.inputs[i].decode() for i in range(m.num_inputs)],
                                 [m.outputs[i].decode() for i in range(m.num_outputs)],
                                 m.description.decode(), int(m.load_time_ns))
        finally:
            L.ModelFreeMetadata(p)

    def GetStats(self) -> ModelStats:
        L = lib()
        p = L.ModelGetStats(self.handle)
        if not p:
            raise RuntimeError("failed to get model statistics")
        try:
            s = p.contents
            return ModelStats(int(s.inference_count), int(s.total_inference_time_ns), int(s.last_inference_time_ns),
                              int(s.memory_usage_bytes))
        finally:
            L.ModelFreeStats(p)

    def Destroy(self):
        if self.handle:
            lib().ModelDestroy(self.handle)
            self.handle = None


def CreateModel(model_path: str, name: str, version: str = "1", model_type: int = ModelONNX, device: int = DeviceGPU,
                device_id: int = 0, input_names: Sequence[str] = (), output_names: Sequence[str] = (),
                load: bool = True) -> Model:
    """createModelInternal (:449-519) + ModelLoad (declared in this build's header)."""
    L = lib()
    cfg = CModelConfig()
    cfg.name = name.encode()
    cfg.version = version.encode()
    cfg.type_ = model_type
    cfg.max_batch_size = 0
    ins = (C.c_char_p * max(len(input_names), 1))(*[s.encode() for s in input_names])
    outs = (C.c_char_p * max(len(output_names), 1))(*[s.encode() for s in output_names])
    cfg.input_names, cfg.num_inputs = ins, len(input_names)
    cfg.output_names, cfg.num_outputs = outs, len(output_names)
    cfg.instance_count = 1
    cfg.dynamic_batching = False
    err = C.c_void_p()
    h = L.ModelCreate(model_path.encode(), model_type, C.byref(cfg), device, device_id, C.byref(err))
    if not h:
        raise RuntimeError("failed to create model: " + _take_error(err))
    m = Model(h, name, version)
    if load:
        err = C.c_void_p()
        if not L.ModelLoad(h, C.byref(err)):
            msg = _take_error(err)
            m.Destroy()
            raise RuntimeError("failed to load model: " + msg)
    return m


def _model_key(name: str, version: str) -> str:
    return name if version == "" else f"{name}:{version}"


class InferenceManager:
    """inference_binding.go:98-103, 177-446."""

    def __init__(self, model_repository_path: str):
        self.handle = lib().InferenceInitialize(model_repository_path.encode())
        if not self.handle:
            raise RuntimeError("failed to initialize inference manager")
        self.loadedModels: dict = {}
        self.loadedModelsMutex = threading.RLock()

    def Shutdown(self):
        with self.loadedModelsMutex:
            for m in self.loadedModels.values():
                m.Destroy()
            self.loadedModels = {}
            if self.handle:
                lib().InferenceShutdown(self.handle)
                self.handle = None

    def LoadModel(self, modelName: str, version: str = "") -> None:
        L = lib()
        err = C.c_void_p()
        cver = version.encode() if version else None
        if not L.InferenceLoadModel(self.handle, modelName.encode(), cver, C.byref(err)):
            raise RuntimeError("failed to load model: " + _take_error(err))
        err = C.c_void_p()
        h = L.GetModelHandle(self.handle, modelName.encode(), cver, C.byref(err))
        if not h:
            raise RuntimeError("model loaded but failed to get handle: " + _take_error(err))
        with self.loadedModelsMutex:
            self.loadedModels[_model_key(modelName, version)] = Model(h, modelName, version)

    def UnloadModel(self, modelName: str, version: str = "") -> None:
        L = lib()
        err = C.c_void_p()
        cver = version.encode() if version else None
        if not L.InferenceUnloadModel(self.handle, modelName.encode(), cver, C.byref(err)):
            raise RuntimeError("failed to unload model: " + _take_error(err))
        with self.loadedModelsMutex:
            m = self.loadedModels.pop(_model_key(modelName, version), None)
            if m:
                m.Destroy()

    def IsModelLoaded(self, modelName: str, version: str = "") -> bool:
        cver = version.encode() if version else None
        return bool(lib().InferenceIsModelLoaded(self.handle, modelName.encode(), cver))

    def ListModels(self) -> list:
        L = lib()
        n = C.c_int(0)
        arr = L.InferenceListModels(self.handle, C.byref(n))
        if not arr or n.value == 0:
            return []
        try:
            return [C.string_at(arr[i]).decode() for i in range(n.value)]
        finally:
            L.InferenceFreeModelList(arr, n.value)

    def GetModel(self, modelName: str, version: str = "") -> Model:
        if not self.IsModelLoaded(modelName, version):
            raise RuntimeError(f"model {modelName} is not loaded")
        with self.loadedModelsMutex:
            m = self.loadedModels.get(_model_key(modelName, version))
        if m is None:
            raise RuntimeError(f"model {modelName} is loaded but no handle is tracked")
        return m

    def RunInference(self, modelName: str, version: str, inputs: Sequence[TensorData],
                     outputConfigs: Sequence[OutputConfig]) -> list:
        return self.GetModel(modelName, version).Infer(inputs, outputConfigs)


def NewInferenceManager(modelRepositoryPath: str) -> InferenceManager:
    return InferenceManager(modelRepositoryPath)


# ---- engine extensions (include/inference_engine_ext.h) ----------------------------------------------------
def DescribeModel(path: str, batch: int = 0) -> dict:
    import json
    err = C.c_void_p()
    p = lib().EngineDescribeModel(path.encode(), batch, C.byref(err))
    if not p:
        raise RuntimeError(_take_error(err))
    return json.loads(_take_string(p))


def Prepare(model: Model, input_shapes: Sequence[Sequence[int]], num_outputs: int = 1):
    """-> (device pointers of the input buffers, device pointers of the output buffers)."""
    L = lib()
    n = len(input_shapes)
    shapes = (CShape * n)()
    keep = []
    for i, s in enumerate(input_shapes):
        a = (C.c_int64 * len(s))(*s)
        keep.append(a)
        shapes[i].dims = C.cast(a, C.POINTER(C.c_int64))
        shapes[i].num_dims = len(s)
    din = (C.c_void_p * n)()
    dout = (C.c_void_p * num_outputs)()
    err = C.c_void_p()
    if not L.EnginePrepare(model.handle, shapes, n, din, dout, num_outputs, C.byref(err)):
        raise RuntimeError("prepare failed: " + _take_error(err))
    return [int(x or 0) for x in din], [int(x or 0) for x in dout]


def RunPrepared(model: Model, iters: int = 1, sync: bool = True) -> None:
    err = C.c_void_p()
    if not lib().EngineRunPrepared(model.handle, iters, 1 if sync else 0, C.byref(err)):
        raise RuntimeError("run failed: " + _take_error(err))


def Synchronize(model: Model) -> None:
    err = C.c_void_p()
    if not lib().EngineSynchronize(model.handle, C.byref(err)):
        raise RuntimeError("synchronize failed: " + _take_error(err))


def GetStream(model: Model) -> int:
    return int(lib().EngineGetStream(model.handle) or 0)


def Profile(model: Model, iters: int = 3) -> list:
    import json
    err = C.c_void_p()
    p = lib().EngineProfile(model.handle, iters, C.byref(err))
    if not p:
        raise RuntimeError("profile failed: " + _take_error(err))
    return json.loads(_take_string(p))


def GetWeightBlob(model: Model):
    ptr = C.c_void_p()
    nbytes = C.c_size_t()
    err = C.c_void_p()
    if not lib().EngineGetWeightBlob(model.handle, C.byref(ptr), C.byref(nbytes), C.byref(err)):
        raise RuntimeError("weight blob unavailable: " + _take_error(err))
    return int(ptr.value or 0), int(nbytes.value)


def WeightsUpdated(model: Model) -> None:
    """Call after rewriting the fp32 blob in place (RCCL broadcast): refreshes the half mirror of fp16 mode."""
    err = C.c_void_p()
    if not lib().EngineWeightsUpdated(model.handle, C.byref(err)):
        raise RuntimeError(_take_error(err))


def Precision(model: Model) -> str:
    return {0: "fp32", 1: "fp16", 2: "fp8"}.get(int(lib().EngineGetPrecision(model.handle)), "unloaded")


def ShardStats(model: Model) -> tuple:
    """(number of in-process replicas a request is sharded over, requests sharded so far)."""
    n, calls = C.c_int(), C.c_int64()
    if not lib().EngineGetShardStats(model.handle, C.byref(n), C.byref(calls)):
        raise RuntimeError("shard stats unavailable")
    return int(n.value), int(calls.value)


def PlanWeights(path: str, batch: int = 1) -> np.ndarray:
    """Host-only: the packed fp32 weight blob the plan steps' offsets index into."""
    n = C.c_size_t()
    err = C.c_void_p()
    p = lib().EnginePlanWeights(path.encode(), int(batch), C.byref(n), C.byref(err))
    if not p:
        raise RuntimeError(_take_error(err))
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value,)).copy()
    finally:
        lib().FreeErrorMessage(p)


def E4m3RoundTrip(x: np.ndarray, scale: float = 1.0):
    """(codes uint8, decoded float32) of the device's e4m3 conversion of x / scale."""
    x = np.ascontiguousarray(x, np.float32).ravel()
    dst = np.empty_like(x)
    codes = np.empty(x.size, np.uint8)
    err = C.c_void_p()
    if not lib().EngineE4m3RoundTrip(x.ctypes.data, dst.ctypes.data, codes.ctypes.data, x.size, float(scale), C.byref(err)):
        raise RuntimeError(_take_error(err))
    return codes, dst


def RuntimeInfo(model: Model, checksums: bool = False) -> dict:
    """Lanes, shards, RCCL broadcast facts, device-time accounting and pipelined-path counters of a loaded model (EngineGetRuntimeInfo)."""
    import json
    err = C.c_void_p()
    p = lib().EngineGetRuntimeInfo(model.handle, 1 if checksums else 0, C.byref(err))
    if not p:
        raise RuntimeError(_take_error(err))
    try:
        return json.loads(C.string_at(p).decode())
    finally:
        lib().FreeErrorMessage(p)


def CopyToDevice(model: Model, dst_dev: int, src: np.ndarray) -> None:
    src = np.ascontiguousarray(src)
    err = C.c_void_p()
    if not lib().EngineMemcpy(model.handle, dst_dev, src.ctypes.data, src.nbytes, 1, C.byref(err)):
        raise RuntimeError(_take_error(err))


def CopyToHost(model: Model, dst: np.ndarray, src_dev: int) -> None:
    assert dst.flags["C_CONTIGUOUS"]
    err = C.c_void_p()
    if not lib().EngineMemcpy(model.handle, dst.ctypes.data, src_dev, dst.nbytes, 2, C.byref(err)):
        raise RuntimeError(_take_error(err))


def MfmaPeak(nacc: int = 4, blocks_per_cu: int = 1, iters: int = 2000) -> float:
    return float(lib().EngineMfmaPeak(nacc, blocks_per_cu, iters))


def BatcherStats(model: Model) -> dict:
    b, r, m = C.c_int64(0), C.c_int64(0), C.c_int(0)
    lib().EngineGetBatcherStats(model.handle, C.byref(b), C.byref(r), C.byref(m))
    return {"device_batches": int(b.value), "coalesced_requests": int(r.value), "max_batch": int(m.value)}


def VectorAdd(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        raise ValueError("Vector sizes does not match")
    out = np.empty_like(a)
    err = C.c_void_p()
    if not lib().EngineVectorAdd(a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size, C.byref(err)):
        raise RuntimeError(_take_error(err))
    return out
