"""Counter-based RNG for synthetic weights and inputs.

Specification (so any other implementation regenerates identical bits):

    GOLDEN   = 0x9E3779B97F4A7C15
    fnv1a(s) = 64-bit FNV-1a of the UTF-8 stream name (offset 0xCBF29CE484222325, prime 0x100000001B3)
    mix(z)   = splitmix64 finaliser:  z ^= z>>30; z *= 0xBF58476D1CE4E5B9; z ^= z>>27;
                                      z *= 0x94D049BB133111EB; z ^= z>>31          (all mod 2^64)
    key      = mix(seed * GOLDEN + fnv1a(stream))
    u64(i)   = mix(key + (i + 1) * GOLDEN)
    uniform(i)  = float32( (u64(i) >> 40) * 2^-24 )                 in [0, 1)
    gaussish(i) = float32( (U(4i)+U(4i+1)+U(4i+2)+U(4i+3) - 2) * sqrt(3) )   with U(j) = (u64(j)>>40)*2^-24 in f64
                  (Irwin-Hall(4): mean 0, variance 1, support [-3.46, 3.46]; no transcendental
                   functions, so the bits do not depend on a libm)

Everything is vectorised numpy on uint64 (wrap-around arithmetic).
"""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def fnv1a(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def _key(seed: int, stream: str) -> np.uint64:
    k = (int(seed) * int(GOLDEN) + fnv1a(stream)) & 0xFFFFFFFFFFFFFFFF
    return _mix(np.array([k], dtype=np.uint64))[0]


def u64(seed: int, stream: str, n: int, start: int = 0) -> np.ndarray:
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        return _mix(_key(seed, stream) + i * GOLDEN)


def uniform(seed: int, stream: str, n: int) -> np.ndarray:
    """float32 in [0,1): 24 random mantissa bits."""
    return ((u64(seed, stream, n) >> np.uint64(40)).astype(np.float64) * 2.0 ** -24).astype(np.float32)


def gaussish(seed: int, stream: str, n: int) -> np.ndarray:
    """float32, mean 0 / variance 1 (Irwin-Hall of 4 uniforms)."""
    u = (u64(seed, stream, 4 * n) >> np.uint64(40)).astype(np.float64) * 2.0 ** -24
    s = u.reshape(n, 4).sum(axis=1)
    return ((s - 2.0) * np.sqrt(3.0)).astype(np.float32)
