"""TEST INFRASTRUCTURE ONLY (see oracle/onnx_oracle.py): numpy emulation of OCP FP8 E4M3 ("e4m3fn") and a numpy interpreter of the
engine's fused execution plan, used to check the fp8 precision mode (BASELINE.json configs[4]).

The reference computes nothing in fp8 (its ONNX Runtime session runs the model's own fp32: inference_engine/src/model.cpp:1264-1270),
so there is no reference-side number for this mode: **parity unpinned**.  Two checkers are used instead:

  * `e4m3_encode` / `e4m3_decode` restate the OCP 8-bit Floating Point Specification (OFP8) rev 1.0, format E4M3: 1 sign, 4 exponent
    (bias 7), 3 mantissa bits, subnormals, no infinities, S.1111.111 = NaN, largest finite 448; conversion from binary32 rounds to
    nearest, ties to even, and (saturating mode) clamps to +-448.  The device conversion (v_cvt_pk_fp8_f32) is compared with it
    code for code in tests/test_gpu_parity.py.
  * `run_plan` executes the plan EngineDescribeModel prints (same step list, same NHWC views, the engine's own packed weight blob)
    in float64, optionally quantising exactly where the fp8 engine does (e4m3 tensors with the engine's calibrated scales, e4m3
    weights with per-output-channel scales).  Against it the fp8 kernels must agree up to fp32 accumulation order; against the
    plain float64 oracle the difference is the quantisation error itself, whose bound the tests state.
"""
from __future__ import annotations

import numpy as np

E4M3_MAX = 448.0


def e4m3_decode(codes: np.ndarray) -> np.ndarray:
    c = np.asarray(codes, np.uint8).astype(np.int64)
    sign = np.where(c & 0x80, -1.0, 1.0)
    e = (c >> 3) & 0xF
    m = c & 0x7
    val = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    val = np.where((e == 15) & (m == 7), np.nan, val)
    return (sign * val).astype(np.float64)


_POS = e4m3_decode(np.arange(0, 0x7F, dtype=np.uint8))      # the 127 non-negative finite values, ascending (code == index)


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """Saturating round-to-nearest-even conversion of float values to e4m3 codes (uint8)."""
    x = np.asarray(x, np.float64)
    a = np.minimum(np.abs(x), E4M3_MAX)
    hi = np.searchsorted(_POS, a, side="left").clip(0, len(_POS) - 1)      # first value >= a
    lo = np.maximum(hi - 1, 0)
    dlo, dhi = a - _POS[lo], _POS[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & (hi % 2 == 0))                 # ties go to the even code (even mantissa)
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    code = np.where(_POS[hi] == a, hi.astype(np.uint8), code)
    return (code | np.where(np.signbit(x), 0x80, 0).astype(np.uint8)).astype(np.uint8)


def quantize(x: np.ndarray, scale: float) -> np.ndarray:
    """Real values -> e4m3 codes with a per-tensor scale -> real values again (what an fp8 tensor in HBM represents).
    The device multiplies by the fp32 reciprocal of the scale, so does this."""
    inv = np.float32(1.0) / np.float32(scale)
    q = (np.asarray(x, np.float64).astype(np.float32) * inv).astype(np.float32)
    return e4m3_decode(e4m3_encode(q)) * float(np.float32(scale))


def quantize_rows(w: np.ndarray):
    """[Cout, K] float weights -> (dequantised e4m3 weights [Cout, K] in real units, row scales), as LaunchQuantizeRowsE4m3 does."""
    w32 = np.asarray(w, np.float32)
    amax = np.abs(w32).max(axis=1)
    sc = np.where(amax > 0, (amax / np.float32(E4M3_MAX)).astype(np.float32), np.float32(1.0)).astype(np.float32)
    inv = (np.float32(1.0) / sc).astype(np.float32)
    q = e4m3_decode(e4m3_encode((w32 * inv[:, None]).astype(np.float32)))
    return q * sc[:, None].astype(np.float64), sc


def _view(bufs, v, n):
    """numpy view [N, H, W, C] (NHWC) or [N, C, H, W] (dense NCHW graph I/O) of a planned tensor inside its buffer."""
    b = bufs[v["buf"]]
    if v["nchw"]:
        return b[: n * v["c"] * v["h"] * v["w"]].reshape(n, v["c"], v["h"], v["w"])
    full = b[: n * v["h"] * v["w"] * v["pitch"]].reshape(n, v["h"], v["w"], v["pitch"])
    return full[..., v["c_off"]: v["c_off"] + v["c"]]


def run_plan(plan: dict, blob: np.ndarray, feeds: dict, act_scales=None, fp8: bool = False) -> dict:
    """Execute a plan (EngineDescribeModel(...)["plan"]) in float64 with the engine's packed fp32 weight blob.

    fp8=True: tensors whose view says f8 are passed through e4m3 with act_scales[step index]; conv weights of steps with algo
    "igemm_f8" through per-row e4m3; half tensors ([N, C] vectors) through float16.  Returns {output name: array in ABI order}."""
    blob = np.asarray(blob, np.float32)
    n = plan["inputs"][0]["dims"][0]
    bufs = [np.zeros(int(sz), np.float64) for sz in plan["buffers"]]
    for io in plan["inputs"]:
        x = np.asarray(feeds[io["name"]], np.float64)
        v = io["view"]
        if v["nchw"]:
            _view(bufs, v, n)[...] = x.reshape(n, v["c"], v["h"], v["w"])
        else:
            _view(bufs, v, n)[...] = x.reshape(n, v["c"], v["h"] * v["w"])[:, None, :, :].transpose(0, 1, 3, 2).reshape(n, v["h"], v["w"], v["c"])
    steps = []
    for s in plan["steps"]:          # a fused step carries the convs it replaces: execute those
        if s.get("parts") and s.get("algo") == "dual_f8":
            # projection shortcut + last conv of a bottleneck block as two GEMMs of one launch: the shortcut is added in fp32, never
            # quantised; the result takes the FUSED step's scale
            pj, cv = dict(s["parts"][0]), dict(s["parts"][1])
            pj["_exact_out"] = True
            cv["idx"] = s["idx"]
            steps.extend([pj, cv])
        elif s.get("parts") and s.get("algo") == "stem_pool":
            # stem conv + max pool in one launch: the conv tile is pooled as halfs in LDS and only the pooled tensor is quantised, with the
            # fused step's scale (conv_stem_kernel<POOL>, kernels_stem.hip)
            cv, pl = dict(s["parts"][0]), dict(s["parts"][1])
            if s.get("tile", 1) != 0:
                cv["_exact_out"] = True
                cv["_half_out"] = True
            cv["idx"] = s["idx"]
            pl["idx"] = s["idx"]
            steps.extend([cv, pl])
        else:
            steps.extend(s["parts"] if s.get("parts") else [s])
    for s in steps:
        vin, vout = s["in"], s["out"]
        xin = _view(bufs, vin, n)
        if vin["nchw"]:
            xin = xin.transpose(0, 2, 3, 1)
        xin = np.array(xin, np.float64)
        if s["pre"]:
            sc = blob[s["pre_scale_off"]: s["pre_scale_off"] + vin["c"]].astype(np.float64)
            sh = blob[s["pre_shift_off"]: s["pre_shift_off"] + vin["c"]].astype(np.float64)
            xin = xin * sc + sh
            if s["pre_relu"]:
                xin = np.maximum(xin, 0)
        kh, kw = s["k"]
        sh_, sw_ = s["stride"]
        pt, pl, pb, pr = s["pads"]
        if s["kind"] == "conv":
            cout, cin = vout["c"], vin["c"]
            w = blob[s["w_off"]: s["w_off"] + cout * kh * kw * cin].reshape(cout, kh * kw * cin)
            if fp8 and s.get("algo") == "igemm_f8":
                w, _ = quantize_rows(w)
            elif fp8 and (vin["f16"] or s.get("algo") == "stem"):
                w = w.astype(np.float16)
            w = np.asarray(w, np.float64)
            if fp8 and s.get("algo") == "stem":
                xin = xin.astype(np.float16).astype(np.float64)
            xp = np.pad(xin, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
            oh, ow = vout["h"], vout["w"]
            cols = np.empty((n, oh, ow, kh * kw * cin), np.float64)
            for ky in range(kh):
                for kx in range(kw):
                    cols[..., (ky * kw + kx) * cin: (ky * kw + kx + 1) * cin] = xp[:, ky: ky + sh_ * (oh - 1) + 1: sh_, kx: kx + sw_ * (ow - 1) + 1: sw_, :]
            y = cols.reshape(-1, kh * kw * cin) @ w.T
            y = y.reshape(n, oh, ow, cout)
            if s["bias"]:
                y = y + blob[s["bias_off"]: s["bias_off"] + cout].astype(np.float64)
            if s.get("residual"):
                y = y + np.array(_view(bufs, s["in2"], n), np.float64)
            if s["relu"]:
                y = np.maximum(y, 0)
        elif s["kind"] == "pool":
            oh, ow = vout["h"], vout["w"]
            if s["max"]:
                xp = np.pad(xin, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
                y = np.full((n, oh, ow, vin["c"]), -np.inf)
                for ky in range(kh):
                    for kx in range(kw):
                        y = np.maximum(y, xp[:, ky: ky + sh_ * (oh - 1) + 1: sh_, kx: kx + sw_ * (ow - 1) + 1: sw_, :])
            else:
                xp = np.pad(xin, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
                ones = np.pad(np.ones(xin.shape[1:3]), ((pt, pb), (pl, pr)), constant_values=1.0 if s["count_include_pad"] else 0.0)
                y = np.zeros((n, oh, ow, vin["c"]))
                cnt = np.zeros((oh, ow))
                for ky in range(kh):
                    for kx in range(kw):
                        y += xp[:, ky: ky + sh_ * (oh - 1) + 1: sh_, kx: kx + sw_ * (ow - 1) + 1: sw_, :]
                        cnt += ones[ky: ky + sh_ * (oh - 1) + 1: sh_, kx: kx + sw_ * (ow - 1) + 1: sw_]
                y = y / np.maximum(cnt, 1)[None, :, :, None]
        elif s["kind"] == "gap":
            y = xin.mean(axis=(1, 2), keepdims=True)
        elif s["kind"] == "eltwise":
            y = xin
            if s.get("in2"):
                y = y + np.array(_view(bufs, s["in2"], n), np.float64)
            if s["relu"]:
                y = np.maximum(y, 0)
        elif s["kind"] == "copy":
            y = xin
        else:
            raise NotImplementedError(s["kind"])
        if fp8 and s.get("_half_out"):
            y = y.astype(np.float16).astype(np.float64)
        if fp8 and vout["f8"] and not s.get("_exact_out"):
            y = quantize(y, act_scales[s["idx"]])
        elif fp8 and vout["f16"]:
            y = y.astype(np.float16).astype(np.float64)
        out = _view(bufs, vout, n)
        if vout["nchw"]:
            out[...] = y.transpose(0, 3, 1, 2)
        else:
            out[...] = y
    res = {}
    for io in plan["outputs"]:
        v = io["view"]
        y = np.array(_view(bufs, v, n))
        if not v["nchw"]:
            y = y.transpose(0, 3, 1, 2)
        res[io["name"]] = y.reshape(io["dims"])
    return res
