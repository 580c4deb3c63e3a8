"""Minimal ONNX protobuf *writer* (wire format only, no `onnx` package needed).

The `onnx` Python package is absent from the build container and the GPU box, so the synthetic
models used by tests and bench.py are serialised by hand.  Only the subset of onnx.proto needed for
inference graphs is covered: ModelProto / GraphProto / NodeProto / AttributeProto / TensorProto /
ValueInfoProto.  Field numbers follow onnx.proto (ONNX 1.17, the version the reference's notebook
installs: docs/run_server.ipynb:170).

The matching *reader* used by the product is C++ (csrc/onnx_reader.cpp); the oracle has its own
independent Python reader (oracle/onnx_oracle.py) so the two decoders cross-check each other.
"""
from __future__ import annotations

import struct
from typing import Iterable, Sequence

import numpy as np

# TensorProto.DataType
FLOAT, UINT8, INT8, INT32, INT64, FLOAT16 = 1, 2, 3, 6, 7, 10
# AttributeProto.AttributeType
A_FLOAT, A_INT, A_STRING, A_TENSOR, A_FLOATS, A_INTS = 1, 2, 3, 4, 6, 7


def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return _varint((field << 3) | wire)


def f_varint(field: int, v: int) -> bytes:
    return _key(field, 0) + _varint(int(v))


def f_bytes(field: int, b: bytes) -> bytes:
    return _key(field, 2) + _varint(len(b)) + b


def f_str(field: int, s: str) -> bytes:
    return f_bytes(field, s.encode("utf-8"))


def f_float(field: int, v: float) -> bytes:
    return _key(field, 5) + struct.pack("<f", v)


def tensor(name: str, arr: np.ndarray, raw: bool = True) -> bytes:
    """TensorProto. raw=True -> raw_data (field 9); raw=False -> float_data/int64_data (packed)."""
    arr = np.ascontiguousarray(arr)
    if arr.dtype == np.float32:
        dt = FLOAT
    elif arr.dtype == np.int64:
        dt = INT64
    elif arr.dtype == np.int32:
        dt = INT32
    elif arr.dtype == np.float16:
        dt = FLOAT16
    else:
        raise TypeError(arr.dtype)
    out = b"".join(f_varint(1, d) for d in arr.shape)
    out += f_varint(2, dt)
    if raw:
        out += f_bytes(9, arr.tobytes())
    elif dt == FLOAT:
        out += f_bytes(4, arr.astype("<f4").tobytes())  # packed float_data
    elif dt == INT64:
        out += f_bytes(7, b"".join(_varint(int(x)) for x in arr.ravel()))
    else:
        raise TypeError("non-raw encoding only for float32/int64")
    out += f_str(8, name)
    return out


def attr_int(name: str, v: int) -> bytes:
    return f_str(1, name) + f_varint(3, v) + f_varint(20, A_INT)


def attr_float(name: str, v: float) -> bytes:
    return f_str(1, name) + f_float(2, v) + f_varint(20, A_FLOAT)


def attr_ints(name: str, vs: Sequence[int], packed: bool = False) -> bytes:
    out = f_str(1, name)
    if packed:
        out += f_bytes(8, b"".join(_varint(int(v)) for v in vs))
    else:
        out += b"".join(f_varint(8, v) for v in vs)
    return out + f_varint(20, A_INTS)


def attr_str(name: str, s: str) -> bytes:
    return f_str(1, name) + f_bytes(4, s.encode()) + f_varint(20, A_STRING)


def node(op_type: str, inputs: Iterable[str], outputs: Iterable[str], name: str = "",
         attrs: Iterable[bytes] = ()) -> bytes:
    out = b"".join(f_str(1, i) for i in inputs)
    out += b"".join(f_str(2, o) for o in outputs)
    if name:
        out += f_str(3, name)
    out += f_str(4, op_type)
    out += b"".join(f_bytes(5, a) for a in attrs)
    return out


def value_info(name: str, shape: Sequence[int | str], elem_type: int = FLOAT) -> bytes:
    dims = b""
    for d in shape:
        if isinstance(d, str):
            dims += f_bytes(1, f_str(2, d))       # dim_param (dynamic)
        else:
            dims += f_bytes(1, f_varint(1, d))    # dim_value
    ttype = f_varint(1, elem_type) + f_bytes(2, dims)
    return f_str(1, name) + f_bytes(2, f_bytes(1, ttype))


def graph(name: str, nodes: Iterable[bytes], initializers: Iterable[bytes],
          inputs: Iterable[bytes], outputs: Iterable[bytes]) -> bytes:
    out = b"".join(f_bytes(1, n) for n in nodes)
    out += f_str(2, name)
    out += b"".join(f_bytes(5, t) for t in initializers)
    out += b"".join(f_bytes(11, i) for i in inputs)
    out += b"".join(f_bytes(12, o) for o in outputs)
    return out


def model(graph_bytes: bytes, opset: int = 11, ir_version: int = 6,
          producer: str = "mi355x-inference-engine-modelgen") -> bytes:
    out = f_varint(1, ir_version)
    out += f_str(2, producer)
    out += f_bytes(7, graph_bytes)
    out += f_bytes(8, f_str(1, "") + f_varint(2, opset))
    return out
