"""MI355X-native batched inference engine behind the reference's `inference_bridge.h` C ABI.

Layout:
  csrc/      C++17 + HIP (gfx950) sources of libinference_engine.so — ONNX reader, graph planner, kernels, C ABI
  binding.py ctypes mirror of the reference's Go cgo binding (inference_engine/binding/inference_binding.go)
  modelgen/  synthetic ONNX model writers + counter-based RNG (test / bench tooling)
  build.py   hipcc build recipe for the shared library
"""
__all__ = ["binding", "build", "modelgen"]
