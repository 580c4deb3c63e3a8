"""CPU: pin the oracle (oracle/onnx_oracle.py) against the reference's known-answer and the float64 fixtures."""
import os

import numpy as np
import pytest

from conftest import MINI, ROOT
from gpu_ai_inference_server_amd.modelgen import models
from oracle import onnx_oracle as O

GOLD = os.path.join(ROOT, "tests", "golden")


def test_reference_test_model_file_decodes():
    """The committed copy of the reference fixture models/test_model/1/model.onnx."""
    m = O.load_model(open(os.path.join(GOLD, "test_model", "1", "model.onnx"), "rb").read())
    assert (m.ir_version, m.opset, m.producer) == (10, 12, "GPU-AI-Inference-Server")
    assert [n.op for n in m.nodes] == ["MatMul", "Add", "Relu", "MatMul", "Add"]
    assert [(n, s) for n, s, _ in m.inputs] == [("input", [1, 3])]
    assert [(n, s) for n, s, _ in m.outputs] == [("output", [1, 2])]
    # weights are np.random.seed(42) randn draws in creation order (scripts/create-test-model.py:25-29)
    st = np.random.RandomState(42)
    for name, shape in (("weight1", (3, 5)), ("bias1", (5,)), ("weight2", (5, 2)), ("bias2", (2,))):
        np.testing.assert_array_equal(m.inits[name], st.randn(*shape).astype(np.float32))
    # modelgen's re-creation carries identical initializers
    mine = O.load_model(models.test_model())
    for k in m.inits:
        np.testing.assert_array_equal(m.inits[k], mine.inits[k])


def test_oracle_reproduces_recorded_onnxruntime_answer():
    """docs/run_server.ipynb:174-175: ORT 1.21.0 CPU, input [[-0.01349723,-1.0577109,0.82254493]] -> [[-0.6017066, 1.8522782]]."""
    g = np.load(os.path.join(GOLD, "test_model.npz"))
    m = O.load_model(open(os.path.join(GOLD, "test_model", "1", "model.onnx"), "rb").read())
    y = O.run(m, {"input": g["ort_recorded_input"]})["output"]
    np.testing.assert_array_equal(y, g["ort_recorded_output"])          # bit-exact in fp32
    for x, ref in zip(g["inputs"], g["outputs_f64"]):
        y = O.run(m, {"input": x[None]})["output"][0]
        np.testing.assert_allclose(y, ref, rtol=2e-6, atol=1e-6)


def test_plumbing_known_answers():
    m = O.load_model(open(os.path.join(GOLD, "test_model", "1", "model.onnx"), "rb").read())
    assert O.estimate_memory_usage(m) == 10485780
    d = O.load_model(models.densenet(1, growth=4, blocks=(1,), stem=8))
    assert O.estimate_memory_usage(d) == 11091872                     # [1,3,224,224] in + [1,1000,1,1] out + 10 MiB
    assert O.sort_versions(["1", "2", "10"])[0] == "10"               # model_repository.cpp:45-53
    assert O.sort_versions(["a", "c", "b"]) == ["c", "b", "a"]


def test_model_infer_semantics():
    m = O.load_model(open(os.path.join(GOLD, "test_model", "1", "model.onnx"), "rb").read())
    x = np.ones((1, 3), np.float32)
    ok, err, outs = O.model_infer(m, [("input", (1, 3), x.tobytes())], [("output", 8)])
    assert ok and outs[0][0] == [1, 2]
    np.testing.assert_allclose(np.frombuffer(outs[0][1], np.float32), [-1.6748662, 2.0709436], rtol=1e-6)
    ok, err, _ = O.model_infer(m, [("data_0", (1, 3), x.tobytes())], [("output", 8)])
    assert not ok and err == "Unexpected input name: data_0"
    ok, err, _ = O.model_infer(m, [], [("output", 8)])
    assert not ok and err == "Expected 1 inputs, got 0"
    # short payload: the tensor is zero-initialised and only data_size bytes are copied in (bridge:746-748)
    ok, _, outs = O.model_infer(m, [("input", (1, 3), x.tobytes()[:4])], [("output", 4)])
    y_full = O.run(m, {"input": np.array([[1, 0, 0]], np.float32)})["output"]
    assert ok and np.frombuffer(outs[0][1], np.float32)[0] == y_full[0, 0] and len(outs[0][1]) == 4


@pytest.mark.parametrize("name", sorted(MINI))
def test_oracle_vs_float64_fixture_mini(name):
    mk, iname, ishape = MINI[name]
    m = O.load_model(mk(models))
    x = models.synthetic_input(ishape, stream=name)
    ref = np.load(os.path.join(GOLD, name + ".npz"))["output_f64"]
    (y,) = O.run(m, {iname: x}).values()
    assert y.shape == ref.shape
    np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-5)
    (y64,) = O.run(m, {iname: x}, dtype=np.float64).values()
    np.testing.assert_allclose(y64, ref, rtol=1e-10, atol=1e-12)


def test_oracle_vs_float64_fixture_densenet121():
    """DenseNet-121 is parity-unpinned by the reference; the oracle is cross-checked against torch-CPU float64."""
    g = np.load(os.path.join(GOLD, "densenet121_b2.npz"))
    m = O.load_model(models.densenet121(2))
    x = models.synthetic_input((2, 3, 224, 224))
    y = O.run(m, {"data_0": x})["fc6_1"].reshape(2, 1000)
    rel = np.abs(y - g["logits_f64"]).max() / np.abs(g["logits_f64"]).max()
    assert rel < 1e-4, rel


def test_torch_cpu_walk_matches_numpy_oracle():
    """bench.py's faster CPU baseline (the oracle's graph walk on torch-CPU primitives) computes the same function."""
    pytest.importorskip("torch")
    for mb, feeds in ((models.densenet(2, growth=8, blocks=(2, 3), stem=16, image=32, classes=10, seed=5), "data_0"),
                      (models.resnet(2, layers=(1, 1, 1, 1), width=8, image=32, classes=7, seed=3), "data")):
        m = O.load_model(mb)
        x = models.synthetic_input((2, 3, 32, 32), stream="torchwalk")
        a = list(O.run(m, {feeds: x}, dtype=np.float64).values())[0]
        b = list(O.run_torch_cpu(m, {feeds: x}).values())[0]
        assert np.abs(a - b).max() / np.abs(a).max() < 1e-5
