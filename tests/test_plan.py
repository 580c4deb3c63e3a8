"""CPU: the ONNX reader + graph planner (host logic of the engine) through the C ABI's EngineDescribeModel."""
import os

import numpy as np
import pytest

from gpu_ai_inference_server_amd import binding as B
from gpu_ai_inference_server_amd.modelgen import models
from gpu_ai_inference_server_amd.modelgen import onnx_pb as pb
from oracle import onnx_oracle as O


def test_describe_test_model_known_answers(model_repo):
    d = B.DescribeModel(os.path.join(model_repo, "test_model", "1"), 1)
    assert (d["ir_version"], d["opset"], d["producer"], d["num_nodes"], d["num_initializers"]) == (10, 12, "GPU-AI-Inference-Server", 5, 4)
    assert d["inputs"] == [{"name": "input", "elem_type": 1, "dims": [1, 3]}]
    assert d["outputs"] == [{"name": "output", "elem_type": 1, "dims": [1, 2]}]
    assert d["memory_usage_bytes"] == 10485780                       # model.cpp:979-1035 on this graph
    steps = d["plan"]["steps"]
    assert [s["name"] for s in steps] == ["matmul1+add1+relu", "matmul2+add2"]      # MatMul+Add(+Relu) fused into two GEMM steps
    assert steps[0]["relu"] and steps[0]["bias"] and not steps[1]["relu"]


def test_describe_agrees_with_oracle_reader(model_repo):
    for name in ("mini_densenet", "mini_densenet_scale", "mini_gemm_mlp", "mini_resnet_block"):
        path = os.path.join(model_repo, name, "1")
        d = B.DescribeModel(path)
        m = O.load_model(open(os.path.join(path, "model.onnx"), "rb").read())
        assert d["num_nodes"] == len(m.nodes) and d["num_initializers"] == len(m.inits)
        assert [(i["name"], i["dims"]) for i in d["inputs"]] == [(n, s) for n, s, _ in m.inputs]
        assert [(o["name"], o["dims"]) for o in d["outputs"]] == [(n, s) for n, s, _ in m.outputs]
        assert d["memory_usage_bytes"] == O.estimate_memory_usage(m)


def test_densenet121_plan(densenet_repo):
    d = B.DescribeModel(os.path.join(densenet_repo, "densenet_onnx", "1"), 32)
    assert d["inputs"][0] == {"name": "data_0", "elem_type": 1, "dims": [-1, 3, 224, 224]}
    assert d["outputs"][0]["name"] == "fc6_1" and d["outputs"][0]["dims"] == [-1, 1000, 1, 1]
    assert d["memory_usage_bytes"] == 11091872                       # SURVEY §8 a7 known-answer (dynamic dim skipped)
    p = d["plan"]
    # dense blocks 3-4 (M <= 8192 pixels at batch 32) run as fused dense-layer steps: the 3x3 of layer L + the 1x1 of layer L+1 in one
    # launch (23 + 15 of them); expanded back into their two convs, the plan is the familiar one:
    fused = [s for s in p["steps"] if s.get("algo") == "dense_fused"]
    # (and the stem conv + the max pool behind it are one step: conv_stem_kernel<POOL> pools the conv tile in LDS)
    assert len(fused) == 38 and len(p["steps"]) == 126 - 38 - 1
    assert p["steps"][0]["algo"] == "stem_pool" and p["steps"][0]["in"]["nchw"] and [q["kind"] for q in p["steps"][0]["parts"]] == ["conv", "pool"]
    assert all(len(s["parts"]) == 2 and s["parts"][0]["k"] == [3, 3] and s["parts"][1]["k"] == [1, 1] and s["parts"][0]["out"]["c"] == 32 and
               s["out"]["c"] == 128 and s["parts"][0]["in"]["buf"] != s["out"]["buf"] for s in fused)            # bottleneck ping-pong
    assert [s["tile"] for s in fused] == [2] * 23 + [1] * 15
    p["steps"] = [q for s in p["steps"] for q in (s["parts"] if s.get("parts") else [s])]
    kinds = [s["kind"] for s in p["steps"]]
    # 121 convs, stem max-pool + 3 transition avg-pools, 1 global pool; every BN/ReLU/Concat fused away
    assert kinds.count("conv") == 121 and kinds.count("pool") == 4 and kinds.count("gap") == 1 and len(kinds) == 126
    # the graph is 5.668 GFLOP per image; the planner runs each transition's AvgPool BEFORE its 1x1 conv (they commute),
    # which removes 3 x 154 MFLOP: 5.209 GFLOP per image are executed
    assert abs(p["total_flops"] / 32 / 5.209e9 - 1) < 2e-3
    act = sum(s["bytes"] for s in p["steps"]) / 32
    assert 91e6 < act < 98e6                                          # 23.8 M activation elements (+weights/32)
    pools = [s for s in p["steps"] if s["kind"] == "pool"]
    assert [s["pre"] and s["pre_relu"] for s in pools] == [False, True, True, True]     # the transitions' BN+ReLU ride on the pool
    trans = [p["steps"][i + 1] for i, s in enumerate(p["steps"]) if s["kind"] == "pool" and s["pre"]]
    assert [(s["in"]["c"], s["in"]["h"], s["out"]["c"]) for s in trans] == [(256, 28, 128), (512, 14, 256), (1024, 7, 512)]
    # dense block 1: six 3x3 convs write 32-channel slices at offsets 64..224 of one 256-channel NHWC buffer
    grow = [s for s in p["steps"] if s["kind"] == "conv" and s["k"] == [3, 3] and s["out"]["pitch"] == 256]
    assert [s["out"]["c_off"] for s in grow] == [64, 96, 128, 160, 192, 224]
    assert len({s["out"]["buf"] for s in grow}) == 1
    # pre-activation BN+ReLU rides on the consumer conv, Conv->BN->ReLU on the producer
    b1 = [s for s in p["steps"] if s["kind"] == "conv" and s["k"] == [1, 1] and s["in"]["pitch"] == 256]
    assert all(s["pre"] and s["pre_relu"] and s["relu"] and s["bias"] for s in b1[:6])
    assert p["steps"][0]["algo"] == "stem" and p["steps"][0]["in"]["nchw"]              # stem kernel reads the ABI's NCHW directly (expanded parts)
    # default kernel choices before any autotuning: the activations-stationary 1x1 (kernels_direct.hip family) and the Winograd 3x3
    # (eight waves, 2x14 output tiles) for the big layers of blocks 1-2, the direct split-K / window tiles for the small grids, the
    # tiled implicit GEMM for the rest
    algos = [s["algo"] for s in p["steps"][2:] if s["kind"] == "conv"]
    assert set(algos) <= {"igemm_vec", "ws1x1", "wino3x3", "direct"} and algos[0] == "direct" and algos[1] == "wino3x3"
    assert algos.count("wino3x3") == 18 and algos.count("direct") >= 90
    assert {s["tile"] for s in p["steps"] if s.get("algo") == "wino3x3"} == {5}
    tiles = {s["tile"] for s in p["steps"][2:] if s["kind"] == "conv" and s["algo"] == "direct"}
    assert {10, 13, 6} <= tiles, tiles
    assert p["outputs"][0]["dims"] == [32, 1000, 1, 1]
    # recycled activation buffers: far fewer buffers than tensors, working set < 256 MiB Infinity Cache + input
    assert len(p["buffers"]) <= 14 and sum(p["buffers"]) * 4 < 320e6


def test_pool_conv_swap_can_be_disabled(densenet_repo, monkeypatch):
    monkeypatch.setenv("IE_NO_POOL_SWAP", "1")
    p = B.DescribeModel(os.path.join(densenet_repo, "densenet_onnx", "1"), 32)["plan"]
    assert abs(p["total_flops"] / 32 / 5.668e9 - 1) < 2e-3            # 5.668 GFLOP per image, as exported
    assert 88e6 < sum(s["bytes"] for s in p["steps"]) / 32 < 92e6      # 96 MB per image as exported, minus the stem tensor that is never written / read (2 x 3.2 MB)
    assert not any(s["pre"] for s in p["steps"] if s["kind"] == "pool")


def test_batch_mismatch_and_unsupported_ops(tmp_path, model_repo):
    with pytest.raises(RuntimeError, match="ONNX model file not found"):
        B.DescribeModel(str(tmp_path / "missing"))
    # fixed-batch model (test_model is [1,3]) rejects another batch like ORT does
    g = pb.graph("g", [pb.node("Relu", ["x"], ["y"], "r")], [], [pb.value_info("x", [2, 4])], [pb.value_info("y", [2, 4])])
    p = tmp_path / "relu.onnx"
    p.write_bytes(pb.model(g))
    d = B.DescribeModel(str(p), 2)
    assert [s["kind"] for s in d["plan"]["steps"]] == ["eltwise"]
    g = pb.graph("g", [pb.node("Softplus", ["x"], ["y"], "sp")], [], [pb.value_info("x", [1, 4])], [pb.value_info("y", [1, 4])])
    p2 = tmp_path / "bad.onnx"
    p2.write_bytes(pb.model(g))
    with pytest.raises(RuntimeError, match=r"Unsupported ONNX operator: Softplus \(node sp\)"):
        B.DescribeModel(str(p2), 1)
    p3 = tmp_path / "garbage.onnx"
    p3.write_bytes(b"\x0a\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01 not a protobuf")
    with pytest.raises(RuntimeError, match="ONNX parse error"):
        B.DescribeModel(str(p3), 1)


def test_packed_and_unpacked_attribute_encodings(tmp_path):
    """protobuf allows both encodings of repeated ints; ORT accepts both, so must the reader."""
    w = np.ones((4, 4, 3, 3), np.float32)
    for packed in (False, True):
        n = pb.node("Conv", ["x", "w"], ["y"], "c", [pb.attr_ints("kernel_shape", [3, 3], packed), pb.attr_ints("pads", [1, 1, 1, 1], packed),
                                                        pb.attr_ints("strides", [2, 2], packed)])
        g = pb.graph("g", [n], [pb.tensor("w", w, raw=not packed)], [pb.value_info("x", [1, 4, 8, 8])], [pb.value_info("y", [1, 4, 4, 4])])
        f = tmp_path / f"c{int(packed)}.onnx"
        f.write_bytes(pb.model(g))
        s = B.DescribeModel(str(f), 1)["plan"]["steps"][0]
        assert (s["k"], s["stride"], s["pads"], s["out"]["h"], s["out"]["w"]) == ([3, 3], [2, 2], [1, 1, 1, 1], 4, 4)


def test_fp16_plan_keeps_graph_io_fp32(densenet_repo, monkeypatch):
    """fp16 precision mode: every buffer between the graph's fp32 input and fp32 output holds halfs; bytes per image halve."""
    monkeypatch.setenv("IE_PRECISION", "fp16")
    p = B.DescribeModel(os.path.join(densenet_repo, "densenet_onnx", "1"), 32)["plan"]
    assert p["precision"] == "fp16"
    assert not p["inputs"][0]["view"]["f16"] and not p["outputs"][0]["view"]["f16"]
    convs = [s for s in p["steps"] if s["kind"] == "conv"]
    # fp32 NCHW in, half NHWC out -- and the max pool behind the stem rides in the same launch: the 112 x 112 x 64 tensor never exists
    assert convs[0]["algo"] == "stem_pool" and not convs[0]["in"]["f16"] and convs[0]["out"]["f16"] and (convs[0]["out"]["h"], convs[0]["out"]["w"]) == (56, 56)
    assert [q["kind"] for q in convs[0]["parts"]] == ["conv", "pool"] and convs[0]["parts"][0]["algo"] == "stem" and convs[0]["parts"][1]["max"]
    assert not [s for s in p["steps"] if s["kind"] == "pool" and s["max"]] and p["steps"][1]["in_src"] == 0
    assert all(s["in"]["f16"] and s["out"]["f16"] for s in convs[1:-1])
    assert convs[-1]["in"]["f16"] and not convs[-1]["out"]["f16"]                                    # classifier writes fp32 logits
    assert 42e6 < sum(s["bytes"] for s in p["steps"]) / 32 < 47e6                                    # 47.6 MB/img + weights/32, minus the swapped transitions and the stem tensor (2 x 1.6 MB)
    monkeypatch.setenv("IE_NO_STEM_POOL", "1")
    p2 = B.DescribeModel(os.path.join(densenet_repo, "densenet_onnx", "1"), 32)["plan"]
    assert p2["steps"][0]["algo"] == "stem" and p2["steps"][1]["kind"] == "pool" and len(p2["steps"]) == len(p["steps"]) + 1
    monkeypatch.delenv("IE_NO_STEM_POOL")
    assert p["activation_bytes"] < 170e6                                                             # vs 300 MB in fp32
    monkeypatch.setenv("IE_PRECISION", "fp32")
    p32 = B.DescribeModel(os.path.join(densenet_repo, "densenet_onnx", "1"), 32)["plan"]
    assert p32["precision"] == "fp32" and not any(s["in"]["f16"] or s["out"]["f16"] for s in p32["steps"])


def test_forced_algorithms_respect_eligibility(densenet_repo, monkeypatch):
    """IE_FORCE_ALGO only switches convs that the target kernel can run (the rest keep the implicit GEMM)."""
    path = os.path.join(densenet_repo, "densenet_onnx", "1")
    monkeypatch.setenv("IE_FORCE_ALGO", "ws")
    monkeypatch.setenv("IE_FORCE_TILE", "4")          # 32 output channels per workgroup: every K up to 1024 fits in LDS
    p = B.DescribeModel(path, 32)["plan"]
    ws = [s for s in p["steps"] if s.get("algo") == "ws1x1"]
    assert len(ws) >= 58 and all(s["k"] == [1, 1] and s["stride"] == [1, 1] for s in ws)
    assert all(s["algo"] != "ws3x3" for s in p["steps"] if s["kind"] == "conv")                       # fp32: no resident-weight 3x3
    monkeypatch.setenv("IE_PRECISION", "fp16")
    p = B.DescribeModel(path, 32)["plan"]
    assert sum(1 for s in p["steps"] if s.get("algo") == "ws3x3") == 58                                # every growth conv
    monkeypatch.setenv("IE_FORCE_ALGO", "direct")
    monkeypatch.setenv("IE_FORCE_TILE", "0")          # 8 waves x up to 8 chunks
    monkeypatch.setenv("IE_PRECISION", "fp32")
    p = B.DescribeModel(path, 32)["plan"]
    d = [s for s in p["steps"] if s.get("algo") == "direct"]
    assert d and all(8 <= s["k"][0] * s["k"][1] * s["in"]["c"] // 16 <= 64 and s["out"]["n"] * s["out"]["h"] * s["out"]["w"] <= 65536 for s in d)
    monkeypatch.setenv("IE_FORCE_TILE", "6")          # window variant: 8 waves x up to 9 chunks, stride 1, output grid == input grid
    p = B.DescribeModel(path, 32)["plan"]
    d = [s for s in p["steps"] if s.get("algo") == "direct"]
    assert len(d) >= 40 and all(8 <= s["k"][0] * s["k"][1] * s["in"]["c"] // 16 <= 72 and s["out"]["c"] % 16 == 0 and s["stride"] == [1, 1] and
                                (s["out"]["h"], s["out"]["w"]) == (s["in"]["h"], s["in"]["w"]) for s in d)


def test_resnet50_plan_fuses_shortcuts(tmp_path):
    """ResNet-50: 53 convs + Gemm, every BN folded, all 16 residual Add+ReLU pairs folded into a conv epilogue (the shortcut that
    already exists when the conv runs becomes its `in2`), 8.2 GFLOP per image."""
    path = models.write_repo(str(tmp_path), "resnet50", models.resnet50("N"))
    d = B.DescribeModel(path, 32)
    assert d["inputs"][0]["dims"] == [-1, 3, 224, 224] and d["outputs"][0]["dims"] == [-1, 1000]
    p = d["plan"]
    kinds = [s["kind"] for s in p["steps"]]
    # (the max pool rides in the stem conv's launch: plan step 0 is the "stem_pool" step, its parts the conv and the pool)
    assert kinds.count("conv") == 54 and kinds.count("eltwise") == 0 and kinds.count("pool") == 0 and kinds.count("gap") == 1
    assert p["steps"][0]["algo"] == "stem_pool" and [q["kind"] for q in p["steps"][0]["parts"]] == ["conv", "pool"]
    res = [s for s in p["steps"] if s["kind"] == "conv" and s["residual"]]
    assert len(res) == 16 and all(s["relu"] and s["k"] == [1, 1] for s in res)
    for s in res:       # the shortcut has the output's shape and lives in another buffer
        assert (s["in2"]["n"], s["in2"]["c"], s["in2"]["h"], s["in2"]["w"]) == (s["out"]["n"], s["out"]["c"], s["out"]["h"], s["out"]["w"])
        assert s["in2"]["buf"] != s["out"]["buf"]
    # projection blocks: the Add rides on the projection conv (its partner branch is complete by then), 4 of them strided or not
    assert sum(1 for s in res if "proj" in s["name"]) == 4
    assert abs(p["total_flops"] / 32 / 8.18e9 - 1) < 1e-2


def test_bn_after_absorbed_residual_is_not_folded_into_the_conv(tmp_path):
    """Pre-activation blocks: Conv -> Add -> BN -> [ReLU] -> GAP.  Folding that BN into the conv's weights/bias would leave the
    shortcut operand unscaled (relu(s*conv + s*b + t + res) instead of relu(s*(conv+res) + t)); the planner must stop epilogue
    fusion at the residual and hand the BN to the consumer's prologue."""
    for final_relu in (True, False):
        path = models.write_repo(str(tmp_path), f"preact{int(final_relu)}", models.preact_block(2, final_relu=final_relu))
        steps = B.DescribeModel(path, 2)["plan"]["steps"]
        res = [s for s in steps if s["kind"] == "conv" and s["residual"]]
        assert len(res) == 2
        for s in res:
            assert "bn" not in s["name"].split("conv")[-1], s["name"]       # nothing BN-ish after the Add in the fused name
            assert not s["relu"]
        last = steps[-1]
        assert last["kind"] == "gap" and last["pre"] and last["pre_relu"] == final_relu and "bn" in last["name"]
        # the BN between the blocks has two readers of its input (next BN and next Add): it rides on the consumer conv
        assert sum(1 for s in steps if s["kind"] == "eltwise") == 0


def test_caffe_scale_export_forms_plan_identically(tmp_path):
    """[C,1,1] constants, [C] constants through Unsqueeze nodes, and opset-6 broadcast=1/axis=1: three spellings of the same
    Scale layer that must all fold into the neighbouring convs; none may reach the device as a separate step."""
    shapes = {}
    for form in (True, "unsqueeze", "legacy_axis"):
        mb = models.densenet(3, growth=12, blocks=(2, 2, 2), stem=24, image=64, classes=17, seed=6, caffe_scale=form)
        path = models.write_repo(str(tmp_path), f"scale_{form}", mb)
        d = B.DescribeModel(path, 3)
        assert d["opset"] == (6 if form == "legacy_axis" else 11)
        steps = d["plan"]["steps"]
        assert all(s["kind"] in ("conv", "pool", "gap") for s in steps)
        shapes[form] = [(s["kind"], s["k"], s["in"]["c"], s["out"]["c"], s["pre"], s["pre_relu"], s["relu"], s["bias"]) for s in steps]
    assert shapes[True] == shapes["unsqueeze"] == shapes["legacy_axis"]
    # opset-6 Mul without broadcast=1 against a differently shaped constant is an error, as in the operator's definition
    gb = models.GraphBuilder("bad", 1)
    y = gb.simple("Mul", ["x", gb.init("s", np.ones(4, np.float32))], out="y")
    f = tmp_path / "bad_legacy.onnx"
    f.write_bytes(gb.finish([("x", [1, 4, 2, 2])], [("y", [1, 4, 2, 2])], opset=6))
    with pytest.raises(RuntimeError, match="broadcast attribute is not set"):
        B.DescribeModel(str(f), 1)


def test_two_input_graph_plan(tmp_path):
    path = models.write_repo(str(tmp_path), "two", models.two_input_graph("N"))
    d = B.DescribeModel(path, 4)
    assert [i["name"] for i in d["inputs"]] == ["b_in", "a_in"]            # graph order, not alphabetical
    assert [i["dims"] for i in d["plan"]["inputs"]] == [[4, 16, 12, 12], [4, 8, 12, 12]]


def test_plan_interpreter_reproduces_the_onnx_oracle(tmp_path):
    """oracle/fp8.py run_plan executes the FUSED plan (EngineDescribeModel's step list + the packed weight blob, both host-only) in
    float64.  Agreement with the ONNX-operator oracle proves the planner's rewrites on the CPU: BN folding into weights / prologues,
    Scale merging, residual absorption, the AvgPool <-> 1x1 swap, concat-by-placement, buffer recycling."""
    from conftest import MINI
    from oracle import fp8 as F
    for name, (mk, iname, ishape) in MINI.items():
        mb = mk(models)
        path = models.write_repo(str(tmp_path), name, mb)
        plan = B.DescribeModel(path, ishape[0])["plan"]
        blob = B.PlanWeights(path, ishape[0])
        assert blob.size == plan["weight_floats"]
        x = models.synthetic_input(ishape, stream=name)
        (ref,) = O.run(O.load_model(mb), {iname: x}, dtype=np.float64).values()
        (y,) = F.run_plan(plan, blob, {iname: x}).values()
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() / np.abs(ref).max() < 5e-7, name


def test_e4m3_emulation_follows_the_ofp8_format():
    from oracle import fp8 as F
    v = F.e4m3_decode(np.arange(256, dtype=np.uint8))
    assert v[0x7E] == 448 and v[0xFE] == -448 and np.isnan(v[0x7F]) and np.isnan(v[0xFF])          # no infinities, one NaN per sign
    assert v[0x01] == 2.0 ** -9 and v[0x08] == 2.0 ** -6 and v[0x38] == 1.0                         # smallest subnormal, smallest normal, one
    fin = ~np.isnan(v)
    back = F.e4m3_encode(v[fin])
    assert np.array_equal(back[1:127], np.arange(1, 127, dtype=np.uint8)) and np.array_equal(F.e4m3_decode(back), v[fin])
    # round to nearest, ties to even mantissa; saturation instead of overflow
    assert F.e4m3_decode(F.e4m3_encode(np.array([17.0, 19.0, 21.0, 1.5 * 2 ** -9, 2.5 * 2 ** -9, 460.0, 1e9, -1e9]))).tolist() == \
        [16.0, 20.0, 20.0, 2 * 2.0 ** -9, 2 * 2.0 ** -9, 448.0, 448.0, -448.0]
    q, sc = F.quantize_rows(np.array([[0.5, -2.0, 1.0], [0.0, 0.0, 0.0]], np.float32))
    assert np.allclose(sc, [2.0 / 448, 1.0]) and np.allclose(q[0], [0.5, -2.0, 1.0]) and not q[1].any()


def test_fp8_plan_structure_and_rejections(tmp_path, densenet_repo, monkeypatch):
    """fp8 precision mode (BASELINE configs[4]): e4m3 tensors between the stem and the global pool, halfs for [N, C] vectors, fp32 graph
    I/O; graphs the fp8 kernels cannot run are rejected at plan time with a reason (never handed to a kernel that would misread bytes)."""
    monkeypatch.setenv("IE_PRECISION", "fp8")
    path = models.write_repo(str(tmp_path), "resnet50", models.resnet50("N"))
    p = B.DescribeModel(path, 4)["plan"]
    assert p["precision"] == "fp8"
    convs = [s for s in p["steps"] if s["kind"] == "conv"]
    assert convs[0]["algo"] == "stem_pool" and not convs[0]["in"]["f8"] and convs[0]["out"]["f8"] and len(convs[0]["parts"]) == 2
    assert all(q["idx"] == 0 for q in convs[0]["parts"]) and convs[0]["parts"][1]["in_src"] == 0          # run as two launches they share the fused step's scale
    # 53 convs + the classifier; the projection shortcut of stage 1 and the block's last 1x1 are ONE step (two GEMMs of one launch:
    # the shortcut tensor is never written), the strided projections of stages 2-4 keep their own step
    dual = [s for s in convs if s["algo"] == "dual_f8"]
    assert len(dual) == 1 and len(dual[0]["parts"]) == 2 and not dual[0]["residual"] and dual[0]["parts"][1]["residual"] and dual[0]["parts"][0]["stride"] == [1, 1]
    assert all(s["algo"] in ("igemm_f8", "dual_f8") and s["in"]["f8"] and s["out"]["f8"] for s in convs[1:-1]) and len(convs) == 53
    assert sum(1 for s in convs if s["residual"]) == 15 and all(s["in2"]["f8"] for s in convs if s["residual"])
    monkeypatch.setenv("IE_NO_DUAL_F8", "1")
    convs1 = [s for s in B.DescribeModel(path, 4)["plan"]["steps"] if s["kind"] == "conv"]
    assert len(convs1) == 54 and sum(1 for s in convs1 if s["residual"]) == 16 and not [s for s in convs1 if s["algo"] == "dual_f8"]
    monkeypatch.delenv("IE_NO_DUAL_F8")
    gap = [s for s in p["steps"] if s["kind"] == "gap"][0]
    assert gap["in"]["f8"] and gap["out"]["f16"] and convs[-1]["in"]["f16"] and not convs[-1]["out"]["f16"] and not convs[-1]["out"]["f8"]
    assert all(s["in_src"] >= 0 for s in p["steps"][1:]) and p["steps"][0]["in_src"] == -1
    # e4m3 storage: a quarter of the fp32 activation bytes
    monkeypatch.setenv("IE_PRECISION", "fp32")
    p32 = B.DescribeModel(path, 4)["plan"]
    assert p["activation_bytes"] < 0.3 * p32["activation_bytes"]
    monkeypatch.setenv("IE_PRECISION", "fp8")
    with pytest.raises(RuntimeError, match="fp8 precision: .*prologue"):
        B.DescribeModel(os.path.join(densenet_repo, "densenet_onnx", "1"), 2)
    with pytest.raises(RuntimeError, match="fp8 precision"):
        B.DescribeModel(models.write_repo(str(tmp_path), "blk", models.resnet_block(2)), 2)
