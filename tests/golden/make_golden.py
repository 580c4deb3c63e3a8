"""Generate committed golden fixtures (run HERE in the build container; never on the GPU box).

    python tests/golden/make_golden.py

Independent arithmetic: every operator is evaluated with **torch CPU in float64** (torch.nn.functional
conv2d / batch_norm / pools), not with the oracle's numpy kernels; only the protobuf *decode* is shared with
oracle/onnx_oracle.py.  Outputs are small .npz files (allow_pickle=False loadable):

  densenet121_b2.npz    : logits float64 [2,1000] for the synthetic DenseNet-121 (seed 121) on the
                          synthetic input (seed 20250704), plus mean/std of every block-end tensor
  mini_*.npz            : full outputs of the small graphs used by the parity tests
  test_model.npz        : the reference's recorded ONNX Runtime 1.21.0 known-answer
                          (docs/run_server.ipynb:174-175) and float64 results for the other probe inputs
                          listed in SURVEY.md §8c

DenseNet-121 is "parity unpinned" by the reference (no model file, never executed there); these fixtures pin
the oracle and the HIP engine to an independent float64 evaluation of the same ONNX graph instead.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _pkg import load_package  # noqa: E402

load_package()
from gpu_ai_inference_server_amd.modelgen import models  # noqa: E402
from oracle import onnx_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run_torch(model: O.Model, feeds: dict, keep=()):
    T = lambda a: torch.from_numpy(np.asarray(a)).to(torch.float64) if np.asarray(a).dtype.kind == "f" \
        else torch.from_numpy(np.asarray(a))
    env = {k: T(v) for k, v in model.inits.items()}
    env.update({k: T(v) for k, v in feeds.items()})
    kept = {}
    for n in model.nodes:
        a = n.attrs
        i = [env[x] if x else None for x in n.inputs]
        if n.op == "Conv":
            p = a.get("pads", [0, 0, 0, 0])
            x = F.pad(i[0], (p[1], p[3], p[0], p[2]))
            y = F.conv2d(x, i[1], i[2] if len(i) > 2 else None, stride=tuple(a.get("strides", [1, 1])))
        elif n.op == "BatchNormalization":
            y = F.batch_norm(i[0], i[3], i[4], i[1], i[2], False, 0.0, a.get("epsilon", 1e-5))
        elif n.op == "Relu":
            y = F.relu(i[0])
        elif n.op == "Concat":
            y = torch.cat(i, dim=a["axis"])
        elif n.op == "MaxPool":
            p = a.get("pads", [0, 0, 0, 0])
            assert p[0] == p[2] and p[1] == p[3]
            y = F.max_pool2d(i[0], tuple(a["kernel_shape"]), tuple(a.get("strides", [1, 1])), (p[0], p[1]))
        elif n.op == "AveragePool":
            p = a.get("pads", [0, 0, 0, 0])
            assert p[0] == p[2] and p[1] == p[3]
            y = F.avg_pool2d(i[0], tuple(a["kernel_shape"]), tuple(a.get("strides", [1, 1])), (p[0], p[1]),
                             count_include_pad=bool(a.get("count_include_pad", 0)))
        elif n.op == "GlobalAveragePool":
            y = i[0].mean(dim=(2, 3), keepdim=True)
        elif n.op == "Gemm":
            A = i[0].T if a.get("transA", 0) else i[0]
            B = i[1].T if a.get("transB", 0) else i[1]
            y = a.get("alpha", 1.0) * (A @ B)
            if len(i) > 2:
                y = y + a.get("beta", 1.0) * i[2]
        elif n.op == "MatMul":
            y = i[0] @ i[1]
        elif n.op in ("Add", "Mul"):
            b = i[1]
            if 0 < model.opset < 7 and a.get("broadcast", 0) and "axis" in a:      # opset-6 form: B aligned at `axis` of A
                b = b.reshape((1,) * a["axis"] + tuple(b.shape) + (1,) * (i[0].ndim - a["axis"] - b.ndim))
            y = i[0] + b if n.op == "Add" else i[0] * b
        elif n.op == "Unsqueeze":
            y = i[0]
            for ax in sorted(a["axes"]):
                y = y.unsqueeze(ax)
        elif n.op == "Flatten":
            y = torch.flatten(i[0], a.get("axis", 1))
        else:
            raise NotImplementedError(n.op)
        env[n.outputs[0]] = y
        if n.outputs[0] in keep:
            kept[n.outputs[0]] = y
    return {o[0]: env[o[0]].numpy() for o in model.outputs}, kept


def _save(name, **arrays):
    """Existing fixtures are left alone (np.savez output is not byte-stable) unless --force is given."""
    path = os.path.join(HERE, name + ".npz")
    if os.path.exists(path) and "--force" not in sys.argv:
        old = np.load(path)
        for k, v in arrays.items():
            assert np.array_equal(old[k], v), f"{name}.{k}: regenerated values differ from the committed fixture"
        return
    np.savez(path, **arrays)


def main():
    torch.set_num_threads(8)
    # ---- test_model ------------------------------------------------------------------
    m = O.load_model(models.test_model())
    probes = np.array([[-0.01349723, -1.0577109, 0.82254493], [1, 1, 1], [1, 2, 3], [0, 0, 0], [-1, 2, -3],
                       [0.5, -0.25, 2.0]], np.float32)
    outs = np.stack([run_torch(m, {"input": p[None]})[0]["output"][0] for p in probes])
    _save("test_model", inputs=probes, outputs_f64=outs,
             ort_recorded_input=np.array([[-0.01349723, -1.0577109, 0.82254493]], np.float32),
             ort_recorded_output=np.array([[-0.6017066, 1.8522782]], np.float32))
    print("test_model", outs)

    # ---- mini graphs -----------------------------------------------------------------
    minis = {
        "mini_densenet": (models.densenet(2, growth=8, blocks=(2, 3), stem=16, image=32, classes=10, seed=5),
                          "data_0", (2, 3, 32, 32)),
        "mini_densenet_scale": (models.densenet(3, growth=12, blocks=(2, 2, 2), stem=24, image=64, classes=17,
                                                seed=6, caffe_scale=True), "data_0", (3, 3, 64, 64)),
        "mini_gemm_mlp": (models.gemm_mlp(4), "x", (4, 64, 1, 1)),
        "mini_resnet_block": (models.resnet_block(2), "x", (2, 3, 16, 16)),
        "mini_preact": (models.preact_block(2), "x", (2, 3, 16, 16)),
        "mini_preact_norelu": (models.preact_block(2, final_relu=False), "x", (2, 3, 16, 16)),
        "mini_densenet_unsqueeze": (models.densenet(3, growth=12, blocks=(2, 2, 2), stem=24, image=64, classes=17,
                                                    seed=6, caffe_scale="unsqueeze"), "data_0", (3, 3, 64, 64)),
        "mini_densenet_legacy": (models.densenet(3, growth=12, blocks=(2, 2, 2), stem=24, image=64, classes=17,
                                                 seed=6, caffe_scale="legacy_axis"), "data_0", (3, 3, 64, 64)),
    }
    for name, (mb, iname, ishape) in minis.items():
        mm = O.load_model(mb)
        x = models.synthetic_input(ishape, stream=name)
        out, _ = run_torch(mm, {iname: x})
        (oname, y), = out.items()
        _save(name, output_f64=y)
        print(name, y.shape, float(np.abs(y).max()))

    # ---- two graph inputs (ordering rule of model.cpp:1174-1190) ---------------------------
    mm = O.load_model(models.two_input_graph(2))
    xa = models.synthetic_input((2, 8, 12, 12), stream="two_input/a")
    xb = models.synthetic_input((2, 16, 12, 12), stream="two_input/b")
    out, _ = run_torch(mm, {"a_in": xa, "b_in": xb})
    _save("mini_two_input", output_f64=out["y"])
    print("mini_two_input", out["y"].shape, float(np.abs(out["y"]).max()))

    # ---- DenseNet-121, batch 2 -------------------------------------------------------
    mb = models.densenet121(2)
    md = O.load_model(mb)
    x = models.synthetic_input((2, 3, 224, 224))
    concat_outs = [n.outputs[0] for n in md.nodes if n.op == "Concat"]
    ends = [concat_outs[5], concat_outs[17], concat_outs[41], concat_outs[57]]
    out, kept = run_torch(md, {"data_0": x}, keep=ends)
    stats = np.array([[float(kept[k].mean()), float(kept[k].std())] for k in ends])
    _save("densenet121_b2", logits_f64=out["fc6_1"].reshape(2, 1000), block_end_stats=stats)
    print("densenet121", out["fc6_1"].reshape(2, 1000)[:, :4], stats)


if __name__ == "__main__":
    main()
