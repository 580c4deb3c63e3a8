"""Synthetic ONNX model builders (DenseNet-121 and small test graphs).

The reference's `models/densenet_onnx/1/model.onnx` is not in the mount (.MISSING_LARGE_BLOBS:1), so the
benchmark model is rebuilt from its I/O contract (`models/densenet_onnx/1/config.json:5-20`: input `data_0`
[N,3,224,224] FLOAT32 -> output `fc6_1` [N,1000,1,1]) and the published DenseNet-121 architecture
(growth 32, blocks 6/12/24/16, stem 64, bottleneck 4*growth, final 1x1 conv 1024->1000 like the
Caffe-derived model-zoo file).  Weights come from the counter-based RNG in rng.py (seed 121).

`test_model()` rebuilds the reference's 543-byte MLP (scripts/create-test-model.py:25-29: weights are
`np.random.seed(42)` randn draws in the order weight1, bias1, weight2, bias2).
"""
from __future__ import annotations

import os
from typing import Sequence

import numpy as np

from . import onnx_pb as pb
from . import rng


class GraphBuilder:
    def __init__(self, name: str, seed: int):
        self.name = name
        self.seed = seed
        self.nodes: list[bytes] = []
        self.inits: list[bytes] = []
        self.n = 0

    def _uid(self, base: str) -> str:
        self.n += 1
        return f"{base}_{self.n}"

    def init(self, name: str, arr: np.ndarray, raw: bool = True) -> str:
        self.inits.append(pb.tensor(name, arr, raw=raw))
        return name

    # ---- ops ----
    def conv(self, x: str, cin: int, cout: int, k: int, stride: int = 1, pad: int = 0, bias: bool = False,
             name: str | None = None, w_scale: float | None = None) -> str:
        name = name or self._uid("conv")
        fan_in = cin * k * k
        std = w_scale if w_scale is not None else float(np.sqrt(2.0 / fan_in))
        w = rng.gaussish(self.seed, name + "_w", cout * cin * k * k).reshape(cout, cin, k, k) * np.float32(std)
        ins = [x, self.init(name + "_w", w.astype(np.float32))]
        if bias:
            b = (rng.uniform(self.seed, name + "_b", cout) - np.float32(0.5)) * np.float32(0.2)
            ins.append(self.init(name + "_b", b.astype(np.float32)))
        y = name + "_out"
        self.nodes.append(pb.node("Conv", ins, [y], name, [
            pb.attr_ints("dilations", [1, 1]), pb.attr_int("group", 1),
            pb.attr_ints("kernel_shape", [k, k]), pb.attr_ints("pads", [pad] * 4),
            pb.attr_ints("strides", [stride, stride])]))
        return y

    def bn(self, x: str, c: int, name: str | None = None, eps: float = 1e-5, g_center: float = 1.0) -> str:
        name = name or self._uid("bn")
        g = np.float32(g_center) + (rng.uniform(self.seed, name + "_g", c) - np.float32(0.5)) * np.float32(0.2)
        b = (rng.uniform(self.seed, name + "_b", c) - np.float32(0.5)) * np.float32(0.2)
        m = (rng.uniform(self.seed, name + "_m", c) - np.float32(0.5)) * np.float32(0.2)
        v = np.float32(0.5) + rng.uniform(self.seed, name + "_v", c)
        ins = [x] + [self.init(f"{name}_{s}", a.astype(np.float32)) for s, a in
                     (("scale", g), ("B", b), ("mean", m), ("var", v))]
        y = name + "_out"
        self.nodes.append(pb.node("BatchNormalization", ins, [y], name,
                                  [pb.attr_float("epsilon", eps), pb.attr_float("momentum", 0.9)]))
        return y

    def relu(self, x: str) -> str:
        name = self._uid("relu")
        self.nodes.append(pb.node("Relu", [x], [name + "_out"], name))
        return name + "_out"

    def concat(self, xs: Sequence[str], axis: int = 1) -> str:
        name = self._uid("concat")
        self.nodes.append(pb.node("Concat", xs, [name + "_out"], name, [pb.attr_int("axis", axis)]))
        return name + "_out"

    def pool(self, op: str, x: str, k: int, stride: int, pad: int = 0, extra: Sequence[bytes] = ()) -> str:
        name = self._uid(op.lower())
        self.nodes.append(pb.node(op, [x], [name + "_out"], name, [
            pb.attr_ints("kernel_shape", [k, k]), pb.attr_ints("pads", [pad] * 4),
            pb.attr_ints("strides", [stride, stride]), *extra]))
        return name + "_out"

    def gap(self, x: str) -> str:
        name = self._uid("gap")
        self.nodes.append(pb.node("GlobalAveragePool", [x], [name + "_out"], name))
        return name + "_out"

    def simple(self, op: str, xs: Sequence[str], attrs: Sequence[bytes] = (), out: str | None = None) -> str:
        name = self._uid(op.lower())
        y = out or name + "_out"
        self.nodes.append(pb.node(op, xs, [y], name, attrs))
        return y

    def finish(self, inputs: Sequence[tuple[str, Sequence[int | str]]],
               outputs: Sequence[tuple[str, Sequence[int | str]]], opset: int = 11) -> bytes:
        g = pb.graph(self.name, self.nodes, self.inits,
                     [pb.value_info(n, s) for n, s in inputs], [pb.value_info(n, s) for n, s in outputs])
        return pb.model(g, opset=opset)


def densenet(batch: int | str = 1, *, growth: int = 32, blocks: Sequence[int] = (6, 12, 24, 16),
             stem: int = 64, bn_size: int = 4, image: int = 224, classes: int = 1000, seed: int = 121,
             in_name: str = "data_0", out_name: str = "fc6_1", caffe_scale: bool | str = False) -> bytes:
    """DenseNet-BC-style graph (DenseNet-121 with the defaults).

    caffe_scale follows each BatchNormalization with the Mul+Add of a Caffe Scale layer (SURVEY.md §2.3 last row), in the three
    forms exporters of the model-zoo Caffe2 DenseNet-121 produce:
      True            [C,1,1] initializers, numpy broadcasting
      "unsqueeze"     [C] initializers routed through Unsqueeze(axes=[1,2]) nodes (opset >= 7 exports)
      "legacy_axis"   [C] initializers with the opset-6 attributes broadcast=1, axis=1 (the whole model is then opset 6)
    """
    gb = GraphBuilder("densenet", seed)

    def norm(x: str, c: int) -> str:
        y = gb.bn(x, c)
        if caffe_scale:
            nm = gb._uid("scale")
            s = np.float32(1.0) + (rng.uniform(seed, nm + "_s", c) - np.float32(0.5)) * np.float32(0.1)
            t = (rng.uniform(seed, nm + "_t", c) - np.float32(0.5)) * np.float32(0.1)
            if caffe_scale == "unsqueeze":
                su = gb.simple("Unsqueeze", [gb.init(nm + "_s", s)], [pb.attr_ints("axes", [1, 2])])
                tu = gb.simple("Unsqueeze", [gb.init(nm + "_t", t)], [pb.attr_ints("axes", [1, 2])])
                y = gb.simple("Mul", [y, su])
                y = gb.simple("Add", [y, tu])
            elif caffe_scale == "legacy_axis":
                legacy = [pb.attr_int("axis", 1), pb.attr_int("broadcast", 1)]
                y = gb.simple("Mul", [y, gb.init(nm + "_s", s)], legacy)
                y = gb.simple("Add", [y, gb.init(nm + "_t", t)], legacy)
            else:
                y = gb.simple("Mul", [y, gb.init(nm + "_s", s.reshape(c, 1, 1))])
                y = gb.simple("Add", [y, gb.init(nm + "_t", t.reshape(c, 1, 1))])
        return y

    x = gb.conv(in_name, 3, stem, 7, stride=2, pad=3, name="conv1")
    x = gb.relu(norm(x, stem))
    x = gb.pool("MaxPool", x, 3, 2, 1)
    c = stem
    for bi, nl in enumerate(blocks):
        for _ in range(nl):
            y = gb.relu(norm(x, c))
            y = gb.conv(y, c, bn_size * growth, 1)
            y = gb.relu(norm(y, bn_size * growth))
            y = gb.conv(y, bn_size * growth, growth, 3, pad=1)
            x = gb.concat([x, y])
            c += growth
        if bi != len(blocks) - 1:
            y = gb.relu(norm(x, c))
            y = gb.conv(y, c, c // 2, 1)
            x = gb.pool("AveragePool", y, 2, 2, 0)
            c //= 2
    x = gb.relu(norm(x, c))
    x = gb.gap(x)
    # Final classifier as a 1x1 conv with bias so the output is [N, classes, 1, 1]
    # (models/densenet_onnx/1/config.json:17).  Smaller init keeps logits O(1).
    fc = "fc6"
    w = rng.gaussish(seed, fc + "_w", classes * c).reshape(classes, c, 1, 1) * np.float32(np.sqrt(1.0 / c))
    b = (rng.uniform(seed, fc + "_b", classes) - np.float32(0.5)) * np.float32(0.2)
    gb.nodes.append(pb.node("Conv", [x, gb.init(fc + "_w", w.astype(np.float32)), gb.init(fc + "_b", b)],
                            [out_name], fc, [pb.attr_ints("kernel_shape", [1, 1]), pb.attr_ints("pads", [0] * 4),
                                             pb.attr_ints("strides", [1, 1]), pb.attr_int("group", 1),
                                             pb.attr_ints("dilations", [1, 1])]))
    return gb.finish([(in_name, [batch, 3, image, image])], [(out_name, [batch, classes, 1, 1])],
                     opset=6 if caffe_scale == "legacy_axis" else 11)


def densenet121(batch: int | str = 1) -> bytes:
    return densenet(batch)


def test_model() -> bytes:
    """Re-creation of the reference's test MLP (same initializer values, names and node order)."""
    st = np.random.RandomState(42)
    w1 = st.randn(3, 5).astype(np.float32)
    b1 = st.randn(5).astype(np.float32)
    w2 = st.randn(5, 2).astype(np.float32)
    b2 = st.randn(2).astype(np.float32)
    nodes = [pb.node("MatMul", ["input", "weight1"], ["matmul1"], "matmul1"),
             pb.node("Add", ["matmul1", "bias1"], ["hidden"], "add1"),
             pb.node("Relu", ["hidden"], ["relu"], "relu"),
             pb.node("MatMul", ["relu", "weight2"], ["matmul2"], "matmul2"),
             pb.node("Add", ["matmul2", "bias2"], ["output"], "add2")]
    inits = [pb.tensor("weight1", w1), pb.tensor("bias1", b1), pb.tensor("weight2", w2), pb.tensor("bias2", b2)]
    g = pb.graph("test-model", nodes, inits, [pb.value_info("input", [1, 3])], [pb.value_info("output", [1, 2])])
    return pb.model(g, opset=12, ir_version=10, producer="GPU-AI-Inference-Server")


def gemm_mlp(batch: int | str = 4, din: int = 64, dh: int = 96, dout: int = 10, seed: int = 7) -> bytes:
    """Flatten -> Gemm(transB=1) -> Relu -> Gemm(transB=0, alpha/beta) : exercises Gemm attribute handling."""
    gb = GraphBuilder("gemm_mlp", seed)
    w1 = rng.gaussish(seed, "w1", dh * din).reshape(dh, din) * np.float32(np.sqrt(2.0 / din))
    b1 = (rng.uniform(seed, "b1", dh) - np.float32(0.5))
    w2 = rng.gaussish(seed, "w2", dh * dout).reshape(dh, dout) * np.float32(np.sqrt(1.0 / dh))
    b2 = (rng.uniform(seed, "b2", dout) - np.float32(0.5))
    x = gb.simple("Flatten", ["x"], [pb.attr_int("axis", 1)])
    x = gb.simple("Gemm", [x, gb.init("w1", w1), gb.init("b1", b1)], [pb.attr_int("transB", 1)])
    x = gb.relu(x)
    gb.simple("Gemm", [x, gb.init("w2", w2, raw=False), gb.init("b2", b2, raw=False)],
              [pb.attr_float("alpha", 0.5), pb.attr_float("beta", 2.0)], out="y")
    return gb.finish([("x", [batch, din, 1, 1])], [("y", [batch, dout])], opset=11)


def resnet_block(batch: int | str = 2, c: int = 32, image: int = 16, seed: int = 50) -> bytes:
    """conv3x3 s2 -> (conv-bn-relu-conv-bn) + identity -> relu -> maxpool -> output [N,C,H,W] (H,W>1).

    Exercises Conv->BN epilogue folding, residual Add, stand-alone Relu, strided 3x3, NCHW output transform.
    """
    gb = GraphBuilder("resnet_block", seed)
    x = gb.conv("x", 3, c, 3, stride=2, pad=1, bias=True)
    x = gb.relu(gb.bn(x, c))
    y = gb.conv(x, c, c, 3, pad=1)
    y = gb.relu(gb.bn(y, c))
    y = gb.conv(y, c, c, 3, pad=1)
    y = gb.bn(y, c)
    s = gb.simple("Add", [y, x])
    s = gb.relu(s)
    gb.nodes.append(pb.node("MaxPool", [s], ["y"], "final_pool", [
        pb.attr_ints("kernel_shape", [2, 2]), pb.attr_ints("pads", [0] * 4), pb.attr_ints("strides", [2, 2])]))
    return gb.finish([("x", [batch, 3, image, image])], [("y", [batch, c, image // 4, image // 4])], opset=11)


def preact_block(batch: int | str = 2, c: int = 32, image: int = 16, seed: int = 53, final_relu: bool = True) -> bytes:
    """Pre-activation (ResNet-v2) residual blocks: the Add of a block is followed by the NEXT block's BN, so the graph holds
    Conv -> Add -> BN(gamma != 1) -> ReLU chains, and ends in Add -> BN -> [ReLU] -> GlobalAveragePool.

    Regression graph for the planner: a BN that follows an absorbed residual Add must scale the shortcut too
    (relu(s*(conv+res)+t), not relu(s*conv+res+t)), so it cannot be folded into the conv's weights.
    """
    gb = GraphBuilder("preact_block", seed)
    x = gb.conv("x", 3, c, 3, pad=1, bias=True)
    for _ in range(2):
        y = gb.conv(gb.relu(gb.bn(x, c, g_center=1.6)), c, c, 3, pad=1)
        y = gb.conv(gb.relu(gb.bn(y, c, g_center=0.7)), c, c, 3, pad=1, bias=True)
        x = gb.simple("Add", [y, x])
    z = gb.bn(x, c, g_center=2.5)
    if final_relu:
        z = gb.relu(z)
    gb.nodes.append(pb.node("GlobalAveragePool", [z], ["y"], "final_gap"))
    return gb.finish([("x", [batch, 3, image, image])], [("y", [batch, c, 1, 1])], opset=11)


def two_input_graph(batch: int | str = 2, ca: int = 8, cb: int = 16, image: int = 12, seed: int = 54) -> bytes:
    """Two graph inputs declared in the order (b_in, a_in) with different channel counts, merged by Concat after one conv each.

    ModelInfer must place payloads by graph input NAME -> graph index (model.cpp:1174-1190), whatever order the caller lists them in.
    """
    gb = GraphBuilder("two_input", seed)
    ya = gb.relu(gb.conv("a_in", ca, 16, 3, pad=1, bias=True))
    yb = gb.relu(gb.conv("b_in", cb, 16, 1, bias=True))
    x = gb.concat([ya, yb])
    x = gb.conv(x, 32, 24, 3, pad=1)
    gb.nodes.append(pb.node("GlobalAveragePool", [x], ["y"], "final_gap"))
    return gb.finish([("b_in", [batch, cb, image, image]), ("a_in", [batch, ca, image, image])], [("y", [batch, 24, 1, 1])], opset=11)


def resnet(batch: int | str = 1, *, layers: Sequence[int] = (3, 4, 6, 3), width: int = 64, image: int = 224, classes: int = 1000,
           seed: int = 50, in_name: str = "data", out_name: str = "logits") -> bytes:
    """ResNet-v1.5 bottleneck network (ResNet-50 with the defaults; BASELINE.json configs[4] names this architecture).

    conv7x7/s2 -> BN -> ReLU -> maxpool3x3/s2 -> 4 stages of bottlenecks [1x1 -> 3x3 (stride on the 3x3) -> 1x1 (x4)] with a
    projection shortcut (1x1/stride conv + BN) on the first block of a stage and identity shortcuts elsewhere -> global average
    pool -> Flatten -> Gemm.  Exercises what DenseNet does not: residual Add + ReLU, strided 1x1 / 3x3 convs, Cout up to 2048,
    Conv->BN folding without a ReLU, a Gemm classifier.  The last BN of every block gets a small gamma (as zero-init-residual
    training leaves it) so activations stay O(1) through 16 residual additions.
    """
    gb = GraphBuilder("resnet", seed)
    x = gb.conv(in_name, 3, width, 7, stride=2, pad=3, name="conv1")
    x = gb.relu(gb.bn(x, width, name="bn1"))
    x = gb.pool("MaxPool", x, 3, 2, pad=1)
    cin = width
    for si, nblocks in enumerate(layers):
        mid = width * (2 ** si)
        cout = mid * 4
        for bi in range(nblocks):
            stride = 2 if (bi == 0 and si > 0) else 1
            tag = f"s{si + 1}b{bi + 1}"
            y = gb.relu(gb.bn(gb.conv(x, cin, mid, 1, name=tag + "_c1"), mid, name=tag + "_bn1"))
            y = gb.relu(gb.bn(gb.conv(y, mid, mid, 3, stride=stride, pad=1, name=tag + "_c2"), mid, name=tag + "_bn2"))
            y = gb.bn(gb.conv(y, mid, cout, 1, name=tag + "_c3", w_scale=float(0.5 * np.sqrt(2.0 / mid))), cout, name=tag + "_bn3")
            if bi == 0:
                sc = gb.bn(gb.conv(x, cin, cout, 1, stride=stride, name=tag + "_proj"), cout, name=tag + "_bnp")
            else:
                sc = x
            x = gb.relu(gb.simple("Add", [y, sc]))
            cin = cout
    x = gb.gap(x)
    x = gb.simple("Flatten", [x], [pb.attr_int("axis", 1)])
    wfc = rng.gaussish(seed, "fc_w", classes * cin).reshape(classes, cin) * np.float32(np.sqrt(1.0 / cin))
    bfc = (rng.uniform(seed, "fc_b", classes) - np.float32(0.5)) * np.float32(0.2)
    gb.simple("Gemm", [x, gb.init("fc_w", wfc.astype(np.float32)), gb.init("fc_b", bfc.astype(np.float32))],
              [pb.attr_int("transB", 1)], out=out_name)
    return gb.finish([(in_name, [batch, 3, image, image])], [(out_name, [batch, classes])], opset=11)


def resnet50(batch: int | str = 1) -> bytes:
    return resnet(batch)


def write_repo(root: str, name: str, model_bytes: bytes, version: str = "1", config_json: str | None = None) -> str:
    d = os.path.join(root, name, version)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "model.onnx")
    tmp = path + f".tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(model_bytes)
    os.replace(tmp, path)
    if config_json is not None:
        with open(os.path.join(d, "config.json"), "w") as f:
            f.write(config_json)
    return d


def synthetic_input(shape: Sequence[int], seed: int = 20250704, stream: str = "input") -> np.ndarray:
    """Synthetic images in [0,1) (the client's /255 convention: client/test_client.py:189).

    0.6 * (per-image, per-channel coarse grid of random levels, nearest-upsampled by `cell`) + 0.4 * U[0,1)
    fine noise, so different images give visibly different logits (pure white noise averages out through
    the pooling stages).  Only RNG draws, multiplies and adds: no libm dependence.
    """
    shape = tuple(int(d) for d in shape)
    n = int(np.prod(shape))
    fine = rng.uniform(seed, stream, n).reshape(shape)
    if len(shape) != 4 or shape[2] < 8 or shape[3] < 8:
        return fine
    b, c, h, w = shape
    cell = max(h // 7, 1)
    gh, gw = -(-h // cell), -(-w // cell)
    coarse = rng.uniform(seed, stream + "/coarse", b * c * gh * gw).reshape(b, c, gh, gw)
    up = np.repeat(np.repeat(coarse, cell, axis=2), cell, axis=3)[:, :, :h, :w]
    return (np.float32(0.6) * up + np.float32(0.4) * fine).astype(np.float32)
