"""Multi-GPU host logic: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The path shards by independent units (images): rank r serves a contiguous slice of the request batch and no
collective sits on the data path.  The only exchange is the load-time broadcast of the packed fp32 weight blob
(folded BN + repacked conv weights, ~32 MB for DenseNet-121) from rank 0, one RCCL broadcast over xGMI.
The reference has nothing here (device 0 hard-coded, inference_bridge.cpp:346-347); this is the capability
BASELINE.json's north_star adds.
"""
from __future__ import annotations

import numpy as np


def shard_batch(batch: int, world: int) -> list[tuple[int, int]]:
    """Contiguous (start, count) image slices, ceil(B/N) per rank with the tail ranks short or empty.
    The NCHW input is sliced on its leading axis, so every shard is one contiguous host range."""
    if batch < 0 or world <= 0:
        raise ValueError("batch must be >= 0 and world > 0")
    per = -(-batch // world) if batch else 0
    out = []
    for r in range(world):
        s = min(r * per, batch)
        out.append((s, max(0, min(per, batch - s))))
    return out


def scatter_outputs(parts: list[np.ndarray], shards: list[tuple[int, int]], batch: int) -> np.ndarray:
    """Per-request result scatter: rank r's logits land at rows [start, start+count) of the request's output."""
    width = next(p.shape[1:] for p in parts if p.size) if any(p.size for p in parts) else ()
    out = np.zeros((batch,) + tuple(width), np.float32)
    for p, (s, c) in zip(parts, shards):
        if c:
            out[s:s + c] = p[:c]
    return out


class _DevicePtr:
    """Zero-copy view of engine-owned HBM for torch (CUDA array interface; works on ROCm builds of torch)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_weights(dist, blob, src: int = 0) -> None:
    """Broadcast the weight blob from `src` to every rank.

    `blob` is either a (device_ptr, nbytes) pair from binding.GetWeightBlob (RCCL path, tensor aliases the
    engine's own HBM so the broadcast lands in place) or a writable torch tensor (gloo path in the CPU tests).
    """
    import torch
    if isinstance(blob, tuple):
        t = torch.as_tensor(_DevicePtr(*blob), device="cuda")
    else:
        t = blob
    dist.broadcast(t, src=src)


def blob_checksum(arr: np.ndarray) -> int:
    """Order-independent 64-bit checksum of a byte buffer (sum of u32 words mod 2^64) for post-broadcast agreement checks."""
    a = np.frombuffer(np.ascontiguousarray(arr).tobytes(), dtype=np.uint8)
    pad = (-a.size) % 4
    if pad:
        a = np.concatenate([a, np.zeros(pad, np.uint8)])
    return int(a.view(np.uint32).astype(np.uint64).sum() & np.uint64(0xFFFFFFFFFFFFFFFF))
