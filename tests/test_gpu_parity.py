"""GPU (-m gpu): the HIP engine, called through the C ABI exactly as the Go binding would, against the oracle and the
committed float64 fixtures.  Tolerance: north_star's 1e-3 relative fp32 bar is the contract; the fp32 MFMA path is an exact
fp32 FMA chain, so the tests hold it to 2e-4 of the output scale (and report the observed figure)."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

from conftest import MINI, ROOT
from gpu_ai_inference_server_amd import binding as B
from gpu_ai_inference_server_amd import build
from gpu_ai_inference_server_amd.modelgen import models
from oracle import onnx_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
RTOL = 2e-4   # of max|ref| ; contract (BASELINE.json north_star): 1e-3


def rel_err(y, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(y, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.fixture(scope="module")
def mgr(model_repo):
    assert B.IsCUDAAvailable(), "no HIP device: the gpu tests must run on the MI355X box"
    m = B.NewInferenceManager(model_repo)
    yield m
    m.Shutdown()


def infer(mgr_or_model, name, iname, x, oname, oshape):
    ins = [B.TensorData(iname, B.DataTypeFloat32, B.Shape(list(x.shape)), x)]
    outs = [B.OutputConfig(oname, Shape=list(oshape), DataType="FLOAT32")]
    if isinstance(mgr_or_model, B.Model):
        r = mgr_or_model.Infer(ins, outs)
    else:
        r = mgr_or_model.RunInference(name, "", ins, outs)
    d = r[0].Data
    dims = r[0].Shape.Dims
    return (d.reshape(dims) if int(np.prod(dims)) == d.size else d), dims


def test_device_queries():
    assert B.GetDeviceCount() >= 1
    info = B.GetDeviceInfo(0)
    assert info.startswith("Device 0: ") and "(Compute Capability " in info      # cuda_utils.cu:51-54 format
    mi = B.GetMemoryInfo(0)
    assert mi.Total > 100e9 and mi.Used == mi.Total - mi.Free                    # 288 GB HBM3E part
    assert B.GetDeviceInfo(4096) == "Unknown device"


def test_vector_add_smoke():
    """test/cuda_test.cpp:33-63: 1 M ones + twos -> threes."""
    a = np.ones(1_000_000, np.float32)
    b = np.full(1_000_000, 2.0, np.float32)
    np.testing.assert_array_equal(B.VectorAdd(a, b), np.full(1_000_000, 3.0, np.float32))


def test_test_model_known_answers(mgr):
    g = np.load(os.path.join(GOLD, "test_model.npz"))
    mgr.LoadModel("test_model")
    try:
        assert mgr.IsModelLoaded("test_model")
        y, dims = infer(mgr, "test_model", "input", g["ort_recorded_input"], "output", [1, 2])
        assert dims == [1, 2]
        # the one result the reference recorded from ONNX Runtime 1.21.0 (docs/run_server.ipynb:174-175)
        np.testing.assert_allclose(y, g["ort_recorded_output"], rtol=1e-6, atol=1e-7)
        for x, ref in zip(g["inputs"], g["outputs_f64"]):
            y, _ = infer(mgr, "test_model", "input", x[None], "output", [1, 2])
            np.testing.assert_allclose(y[0], ref, rtol=2e-6, atol=1e-6)
        m = mgr.GetModel("test_model")
        st = m.GetStats()
        assert st.InferenceCount == 1 + len(g["inputs"]) and st.MemoryUsageBytes == 10485780
        assert st.TotalInferenceTimeNs >= st.LastInferenceTimeNs > 0
        md = m.GetMetadata()
        assert (md.Name, md.Version, md.Type, md.Inputs, md.Outputs) == ("test_model", "1", B.ModelONNX, ["input"], ["output"])
        assert md.LoadTimeNs > 0
        with pytest.raises(RuntimeError, match="Model already loaded"):
            mgr.LoadModel("test_model")
    finally:
        mgr.UnloadModel("test_model")
    assert not mgr.IsModelLoaded("test_model")


def test_infer_validation_errors(mgr):
    mgr.LoadModel("test_model")
    try:
        m = mgr.GetModel("test_model")
        x = np.ones((1, 3), np.float32)
        n0 = m.GetStats().InferenceCount
        with pytest.raises(RuntimeError, match="Unexpected input name: data_0"):
            infer(m, "", "data_0", x, "output", [1, 2])
        two = [B.TensorData("input", 0, B.Shape([1, 3]), x)] * 2
        with pytest.raises(RuntimeError, match="Expected 1 inputs, got 2"):
            m.Infer(two, [B.OutputConfig("output", [1, 2])])
        assert m.GetStats().InferenceCount == n0          # validation failures are not counted (model.cpp:566-569)
        with pytest.raises(RuntimeError, match="Got invalid dimensions for input: input"):
            infer(m, "", "input", np.ones((2, 3), np.float32), "output", [2, 2])
        assert m.GetStats().InferenceCount == n0 + 1      # backend failures are (model.cpp:607-610)
        # output buffer larger than produced: tail is zero-filled, dims clamp to the caller's rank
        y, dims = infer(m, "", "input", x, "output", [1, 6])
        assert dims == [1, 2] and np.all(y.ravel()[2:] == 0) and abs(y.ravel()[0] + 1.6748662) < 1e-5
        # output buffer smaller than produced: only data_size bytes are written (bridge:810)
        y, _ = infer(m, "", "input", x, "output", [1])
        assert abs(y.ravel()[0] + 1.6748662) < 1e-5
    finally:
        mgr.UnloadModel("test_model")


@pytest.mark.parametrize("name", sorted(MINI))
def test_mini_graphs_vs_oracle_and_fixture(mgr, name):
    mk, iname, ishape = MINI[name]
    om = O.load_model(mk(models))
    oname, oshape, _ = om.outputs[0]
    x = models.synthetic_input(ishape, stream=name)
    ref64 = np.load(os.path.join(GOLD, name + ".npz"))["output_f64"]
    (yo,) = O.run(om, {iname: x}).values()
    mgr.LoadModel(name)
    try:
        y, dims = infer(mgr, name, iname, x, oname, oshape)
        assert dims == list(oshape)
        e64, eo = rel_err(y, ref64), rel_err(y, yo)
        print(f"{name}: rel err vs float64 fixture {e64:.2e}, vs oracle fp32 {eo:.2e}")
        assert e64 < RTOL and eo < RTOL
    finally:
        mgr.UnloadModel(name)


@pytest.mark.parametrize("tile", range(16))
@pytest.mark.parametrize("algo", ["igemm", "scalar"])
def test_every_igemm_tile_and_loader(model_repo, tile, algo):
    """Each MFMA tile configuration x both operand loaders on graphs with ragged M / Cout / Cin tails."""
    os.environ["IE_FORCE_TILE"] = str(tile)
    os.environ["IE_FORCE_ALGO"] = algo
    try:
        for name in ("mini_densenet_scale", "mini_resnet_block", "mini_gemm_mlp"):
            mk, iname, ishape = MINI[name]
            om = O.load_model(mk(models))
            oname, oshape, _ = om.outputs[0]
            x = models.synthetic_input(ishape, stream=name)
            ref64 = np.load(os.path.join(GOLD, name + ".npz"))["output_f64"]
            m = B.CreateModel(os.path.join(model_repo, name, "1"), name)
            try:
                y, _ = infer(m, "", iname, x, oname, oshape)
                assert rel_err(y, ref64) < RTOL, (name, tile, algo, rel_err(y, ref64))
            finally:
                m.Destroy()
    finally:
        del os.environ["IE_FORCE_TILE"], os.environ["IE_FORCE_ALGO"]


@pytest.mark.parametrize("in_launch", ["0", "1"])
@pytest.mark.parametrize("splitk", [2, 3, 7])
def test_split_k_reduction(model_repo, splitk, in_launch):
    """K-tiles split over several workgroups; slabs combined in-launch by the last-arriving workgroup (agent-scope
    release/acquire + ticket counter) or by the two-pass reduce kernel.  Both sum in slice order: deterministic."""
    os.environ["IE_FORCE_SPLITK"] = str(splitk)
    os.environ["IE_FORCE_ALGO"] = "igemm"
    os.environ["IE_SPLITK_IN_LAUNCH"] = in_launch
    try:
        for name, tile in (("mini_densenet_scale", 5), ("mini_resnet_block", 4), ("mini_gemm_mlp", 3)):
            os.environ["IE_FORCE_TILE"] = str(tile)
            mk, iname, ishape = MINI[name]
            om = O.load_model(mk(models))
            oname, oshape, _ = om.outputs[0]
            x = models.synthetic_input(ishape, stream=name)
            ref64 = np.load(os.path.join(GOLD, name + ".npz"))["output_f64"]
            m = B.CreateModel(os.path.join(model_repo, name, "1"), name)
            try:
                y, _ = infer(m, "", iname, x, oname, oshape)
                y2, _ = infer(m, "", iname, x, oname, oshape)
                assert rel_err(y, ref64) < RTOL, (name, splitk, rel_err(y, ref64))
                np.testing.assert_array_equal(y, y2)          # no atomics: bitwise reproducible
            finally:
                m.Destroy()
    finally:
        for k in ("IE_FORCE_SPLITK", "IE_FORCE_ALGO", "IE_FORCE_TILE", "IE_SPLITK_IN_LAUNCH"):
            os.environ.pop(k, None)


@pytest.mark.parametrize("tile", range(8))
@pytest.mark.parametrize("splitk", [1, 2])
def test_raster_3x3_kernel(model_repo, tile, splitk):
    """LDS-window 3x3 kernel (nine shifted GEMMs over the padded raster): every tile shape, with and without split-K,
    on graphs whose 3x3 convs have ragged channel counts (12, 32, 48) and image sizes 4..32."""
    os.environ.update(IE_FORCE_ALGO="raster", IE_FORCE_TILE=str(tile), IE_FORCE_SPLITK=str(splitk))
    try:
        for name in ("mini_densenet_scale", "mini_resnet_block", "mini_densenet"):
            mk, iname, ishape = MINI[name]
            om = O.load_model(mk(models))
            oname, oshape, _ = om.outputs[0]
            x = models.synthetic_input(ishape, stream=name)
            ref64 = np.load(os.path.join(GOLD, name + ".npz"))["output_f64"]
            d = B.DescribeModel(os.path.join(model_repo, name, "1"), ishape[0])
            assert any(s.get("algo") == "raster3x3" for s in d["plan"]["steps"]), name
            m = B.CreateModel(os.path.join(model_repo, name, "1"), name)
            try:
                y, _ = infer(m, "", iname, x, oname, oshape)
                assert rel_err(y, ref64) < RTOL, (name, tile, splitk, rel_err(y, ref64))
            finally:
                m.Destroy()
    finally:
        for k in ("IE_FORCE_SPLITK", "IE_FORCE_ALGO", "IE_FORCE_TILE"):
            os.environ.pop(k, None)


def test_pool_conv_swap_matches_unswapped_graph(model_repo):
    """Transitions run AvgPool before their 1x1 conv (linear ops commute; 4x fewer conv FLOPs).  Same logits as the graph order up
    to fp32 summation order, and both within the contract of the float64 fixture."""
    name = "mini_densenet_scale"
    mk, iname, ishape = MINI[name]
    x = models.synthetic_input(ishape, stream=name)
    ref64 = np.load(os.path.join(GOLD, name + ".npz"))["output_f64"]
    ys = []
    for env in (dict(), dict(IE_NO_POOL_SWAP="1")):
        os.environ.update(env)
        try:
            d = B.DescribeModel(os.path.join(model_repo, name, "1"), ishape[0])
            assert any(s["kind"] == "pool" and s["pre"] for s in d["plan"]["steps"]) == (not env)
            m = B.CreateModel(os.path.join(model_repo, name, "1"), name)
            try:
                ys.append(infer(m, "", iname, x, "fc6_1", [3, 17, 1, 1])[0])
            finally:
                m.Destroy()
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
    assert rel_err(ys[0], ref64) < RTOL and rel_err(ys[1], ref64) < RTOL
    assert rel_err(ys[0], ys[1]) < 2e-6


def test_naive_kernel_agrees(model_repo):
    os.environ["IE_FORCE_ALGO"] = "naive"
    try:
        name = "mini_densenet"
        mk, iname, ishape = MINI[name]
        x = models.synthetic_input(ishape, stream=name)
        ref64 = np.load(os.path.join(GOLD, name + ".npz"))["output_f64"]
        m = B.CreateModel(os.path.join(model_repo, name, "1"), name)
        y, _ = infer(m, "", iname, x, "fc6_1", [2, 10, 1, 1])
        m.Destroy()
        assert rel_err(y, ref64) < RTOL
    finally:
        del os.environ["IE_FORCE_ALGO"]


@pytest.fixture(scope="module")
def densenet(densenet_repo):
    m = B.CreateModel(os.path.join(densenet_repo, "densenet_onnx", "1"), "densenet_onnx")
    yield m
    m.Destroy()


def test_densenet121_b2_vs_float64_fixture(densenet):
    g = np.load(os.path.join(GOLD, "densenet121_b2.npz"))
    x = models.synthetic_input((2, 3, 224, 224))
    y, dims = infer(densenet, "", "data_0", x, "fc6_1", [2, 1000, 1, 1])
    assert dims == [2, 1000, 1, 1]
    e = rel_err(y.reshape(2, 1000), g["logits_f64"])
    print(f"densenet121 B=2: rel err vs float64 fixture {e:.2e}")
    assert e < RTOL
    assert np.argmax(y.reshape(2, 1000), 1).tolist() == np.argmax(g["logits_f64"], 1).tolist()
    md = densenet.GetMetadata()
    assert md.Inputs == ["data_0"] and md.Outputs == ["fc6_1"]          # names come from the graph (fixes SURVEY §3.5-1)
    assert densenet.GetStats().MemoryUsageBytes == 11091872


def test_densenet121_b32_batch_independence(densenet):
    """BASELINE configs[1] size.  Size-independent properties: every image's logits equal the logits of that image run
    alone (shards are independent units), and a permutation of the batch permutes the outputs."""
    x = models.synthetic_input((32, 3, 224, 224), stream="b32")
    y32, _ = infer(densenet, "", "data_0", x, "fc6_1", [32, 1000, 1, 1])
    y32 = y32.reshape(32, 1000)
    assert np.isfinite(y32).all() and np.abs(y32).max() < 50
    for i in (0, 13, 31):
        y1, _ = infer(densenet, "", "data_0", x[i:i + 1], "fc6_1", [1, 1000, 1, 1])
        assert rel_err(y1.reshape(1000), y32[i]) < 2e-5
    perm = np.random.RandomState(0).permutation(32)
    yp, _ = infer(densenet, "", "data_0", x[perm], "fc6_1", [32, 1000, 1, 1])
    assert rel_err(yp.reshape(32, 1000), y32[perm]) < 2e-5
    # oracle on two of the 32 images (seconds on CPU)
    om = O.load_model(models.densenet121(2))
    yo = O.run(om, {"data_0": x[[3, 27]]})["fc6_1"].reshape(2, 1000)
    assert rel_err(y32[[3, 27]], yo) < RTOL


def test_device_resident_path_matches_host_path(densenet):
    """EnginePrepare/EngineRunPrepared (what bench.py times) produce the same bytes as ModelInfer."""
    x = models.synthetic_input((4, 3, 224, 224), stream="dev")
    y_host, _ = infer(densenet, "", "data_0", x, "fc6_1", [4, 1000, 1, 1])
    din, dout = B.Prepare(densenet, [[4, 3, 224, 224]], 1)
    B.CopyToDevice(densenet, din[0], x)
    B.RunPrepared(densenet, 2, True)
    y = np.empty((4, 1000), np.float32)
    B.CopyToHost(densenet, y, dout[0])
    np.testing.assert_array_equal(y, y_host.reshape(4, 1000))
    prof = B.Profile(densenet, 1)
    assert 60 <= len(prof) <= 126 and all(p["ms"] > 0 for p in prof)          # 126 launches, minus the dense layers fused at this batch size
    ptr, nbytes = B.GetWeightBlob(densenet)
    assert ptr and nbytes > 30e6


def test_weight_blob_aliases_engine_memory_for_rccl(densenet):
    """The N>1 launcher broadcasts the packed weights IN PLACE: the torch tensor handed to RCCL must alias the engine's HBM."""
    import torch
    from gpu_ai_inference_server_amd import sharding
    ptr, nbytes = B.GetWeightBlob(densenet)
    t = torch.as_tensor(sharding._DevicePtr(ptr, nbytes), device="cuda")
    assert t.data_ptr() == ptr and t.numel() == nbytes and t.dtype == torch.uint8
    host = np.empty(nbytes, np.uint8)
    B.CopyToHost(densenet, host, ptr)
    assert sharding.blob_checksum(host) == sharding.blob_checksum(t.cpu().numpy())
    # a write through the torch view is visible to the engine (what a broadcast on a non-root rank does) -- restore afterwards
    first = t[:16].clone()
    t[:16] = 7
    torch.cuda.synchronize()
    B.CopyToHost(densenet, host, ptr)
    assert (host[:16] == 7).all()
    t[:16] = first
    torch.cuda.synchronize()


def test_bench_under_torchrun_single_rank(tmp_path):
    """bench.py through the driver's launch contract (torch.distributed.run, nccl = RCCL) on the one GPU of this box:
    process-group init, in-place weight broadcast, barrier, all-reduce(MAX) of the elapsed time."""
    import json
    import sys
    env = dict(os.environ, IE_BENCH_MODEL_ROOT=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--cpu-sample", "0", "--no-hostpath", "--no-secondary"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["value"] > 1000 and d["scaling"] == "weak" and d["roofline"]["bound"] == "mfma"


def test_concurrent_infer_load_unload(mgr):
    """gin serves each request on its own goroutine: concurrent ModelInfer on one handle plus registry traffic."""
    mgr.LoadModel("mini_densenet")
    mk, iname, ishape = MINI["mini_densenet"]
    x = models.synthetic_input(ishape, stream="mini_densenet")
    ref, _ = infer(mgr, "mini_densenet", iname, x, "fc6_1", [2, 10, 1, 1])
    errs = []

    def worker():
        try:
            for _ in range(20):
                y, _ = infer(mgr, "mini_densenet", iname, x, "fc6_1", [2, 10, 1, 1])
                np.testing.assert_array_equal(y, ref)
                mgr.IsModelLoaded("mini_densenet"); mgr.ListModels()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    h = mgr.GetModel("mini_densenet")
    mgr.UnloadModel("mini_densenet")
    assert not mgr.IsModelLoaded("mini_densenet")
    del h


def test_c_replay_harness_on_gpu(model_repo, engine_lib):
    harness = build.build_harness() if not os.path.exists(os.path.join(os.path.dirname(engine_lib), "replay_binding")) \
        else os.path.join(os.path.dirname(engine_lib), "replay_binding")
    r = subprocess.run([harness, model_repo, "test_model", "input", "output", "2", "1", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(l.split(" -> ", 1) for l in r.stdout.splitlines() if l.startswith("CALL "))
    assert out["CALL InferenceLoadModel"].strip() == "1" and out["CALL InferenceLoadModel(again)"] == "0 error=Model already loaded"
    assert out["CALL ModelInfer"].strip() == "1" and out["CALL ModelIsLoaded(after unload)"] == "0"
    assert out["CALL InferenceUnloadModel(again)"] == "0 error=Model not found"
    vals = [float(v) for v in [l for l in r.stdout.splitlines() if l.startswith("OUTPUT ")][0].split()[1:]]
    np.testing.assert_allclose(vals, [-1.6748662, 2.0709436], rtol=2e-6)   # ones(1,3) probe (test/onnx_test.cpp:92), SURVEY §8c
    assert "count=1 mem=10485780" in out["CALL ModelGetStats"]


def _random_conv_graph(rs, case, cin_choices=(3, 4, 8, 12, 20, 32, 36, 64, 96)):
    """One random Conv (+ optional pre-activation BN/ReLU, bias, post BN/ReLU, concat partner) as an ONNX graph."""
    from gpu_ai_inference_server_amd.modelgen import onnx_pb as pb
    n = int(rs.choice([1, 2, 3, 5]))
    k = int(rs.choice([1, 1, 3, 3, 5, 7]))
    stride = int(rs.choice([1, 1, 2]))
    pad = int(rs.choice([0, k // 2]))
    cin = int(rs.choice(list(cin_choices)))
    cout = int(rs.choice([1, 5, 16, 32, 40, 64, 100, 128]))
    h = int(rs.randint(max(k, 3), 24))
    w = int(rs.randint(max(k, 3), 24))
    pre = bool(rs.randint(2)) and cin % 4 == 0
    post = int(rs.randint(3))          # 0 none, 1 relu, 2 bn+relu
    bias = bool(rs.randint(2))
    nodes, inits = [], []
    x = "x"
    # a first 1x1 conv moves the tensor into the engine's NHWC world so the conv under test takes the vector path
    w0 = (rs.randn(cin, 3, 1, 1) * 0.5).astype(np.float32)
    inits.append(pb.tensor("w0", w0))
    nodes.append(pb.node("Conv", ["x", "w0"], ["h0"], "c0", [pb.attr_ints("kernel_shape", [1, 1])]))
    x = "h0"
    if pre:
        for nm, v in (("g", 1 + 0.1 * rs.randn(cin)), ("b", 0.1 * rs.randn(cin)), ("m", 0.1 * rs.randn(cin)), ("v", 0.5 + rs.rand(cin))):
            inits.append(pb.tensor("pre_" + nm, v.astype(np.float32)))
        nodes.append(pb.node("BatchNormalization", [x, "pre_g", "pre_b", "pre_m", "pre_v"], ["p0"], "prebn", [pb.attr_float("epsilon", 1e-5)]))
        nodes.append(pb.node("Relu", ["p0"], ["p1"], "prerelu"))
        x = "p1"
    wt = (rs.randn(cout, cin, k, k) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)
    inits.append(pb.tensor("w", wt))
    ins = [x, "w"]
    if bias:
        inits.append(pb.tensor("bvec", (0.2 * rs.randn(cout)).astype(np.float32)))
        ins.append("bvec")
    nodes.append(pb.node("Conv", ins, ["y0"], "conv", [pb.attr_ints("kernel_shape", [k, k]), pb.attr_ints("pads", [pad] * 4),
                                                       pb.attr_ints("strides", [stride, stride])]))
    y = "y0"
    if post == 2:
        for nm, v in (("g", 1 + 0.1 * rs.randn(cout)), ("b", 0.1 * rs.randn(cout)), ("m", 0.1 * rs.randn(cout)), ("v", 0.5 + rs.rand(cout))):
            inits.append(pb.tensor("post_" + nm, v.astype(np.float32)))
        nodes.append(pb.node("BatchNormalization", [y, "post_g", "post_b", "post_m", "post_v"], ["y1"], "postbn", [pb.attr_float("epsilon", 1e-5)]))
        y = "y1"
    if post >= 1:
        nodes.append(pb.node("Relu", [y], ["y2"], "postrelu"))
        y = "y2"
    oh = (h + 2 * pad - k) // stride + 1
    ow = (w + 2 * pad - k) // stride + 1
    # concat the result behind a strided copy of itself: exercises channel-offset epilogues into a wider buffer
    nodes.append(pb.node("Concat", [y, y], ["out"], "cat", [pb.attr_int("axis", 1)]))
    g = pb.graph(f"rand{case}", nodes, inits, [pb.value_info("x", [n, 3, h, w])], [pb.value_info("out", [n, 2 * cout, oh, ow])])
    return pb.model(g), (n, 3, h, w), (n, 2 * cout, oh, ow), dict(k=k, stride=stride, pad=pad, cin=cin, cout=cout, h=h, w=w, pre=pre, post=post, bias=bias)


@pytest.mark.parametrize("seed", range(4))
def test_random_conv_graphs_vs_oracle(tmp_path, seed):
    """Seeded random convolutions (kernel 1..7, stride 1/2, ragged channels, with/without the fused BN+ReLU prologue /
    epilogue, duplicated into a Concat) through every algorithm choice the planner/autotuner can make, against the oracle."""
    rs = np.random.RandomState(1000 + seed)
    modes = [dict(), dict(IE_AUTOTUNE="0"), dict(IE_FORCE_ALGO="raster"), dict(IE_FORCE_ALGO="igemm", IE_FORCE_TILE=str(7 + seed)),
             dict(IE_FORCE_ALGO="scalar", IE_FORCE_TILE=str(seed)), dict(IE_FORCE_ALGO="igemm", IE_FORCE_SPLITK="3", IE_FORCE_TILE=str(seed + 2)),
             dict(IE_FORCE_ALGO="ws", IE_FORCE_TILE=str(seed + 1)), dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE=str(seed))]
    worst = 0.0
    for case in range(14):
        mb, ishape, oshape, desc = _random_conv_graph(rs, case)
        d = models.write_repo(str(tmp_path), f"r{seed}_{case}", mb)
        om = O.load_model(mb)
        x = rs.rand(*ishape).astype(np.float32)
        ref = O.run(om, {"x": x}, dtype=np.float64)["out"]
        env = modes[case % len(modes)]
        os.environ.update(env)
        try:
            m = B.CreateModel(d, "r")
            try:
                y, dims = infer(m, "", "x", x, "out", oshape)
            finally:
                m.Destroy()
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
        assert dims == list(oshape)
        e = rel_err(y, ref)
        worst = max(worst, e)
        assert e < RTOL, (desc, env, e)
    print(f"seed {seed}: worst rel err {worst:.2e}")


def test_dynamic_batcher_coalesces_concurrent_requests(densenet_repo):
    """concurrent single-image ModelInfer calls (one per server goroutine) are folded into one device batch and the
    logits are scattered back per caller; each caller sees exactly what it would have got alone."""
    x = models.synthetic_input((8, 3, 224, 224), stream="batcher")
    path = os.path.join(densenet_repo, "densenet_onnx", "1")
    ref_m = B.CreateModel(path, "densenet_onnx")
    assert B.BatcherStats(ref_m)["max_batch"] == 0
    ref = np.stack([infer(ref_m, "", "data_0", x[i:i + 1], "fc6_1", [1, 1000, 1, 1])[0].reshape(1000) for i in range(8)])
    ref_m.Destroy()
    os.environ.update(IE_DYNAMIC_BATCH="8", IE_BATCH_WINDOW_US="200000")
    try:
        m = B.CreateModel(path, "densenet_onnx")
    finally:
        del os.environ["IE_DYNAMIC_BATCH"], os.environ["IE_BATCH_WINDOW_US"]
    try:
        assert B.BatcherStats(m)["max_batch"] == 8
        infer(m, "", "data_0", x[:8], "fc6_1", [8, 1000, 1, 1])       # rows == max_batch bypasses the batcher; warms the B=8 plan
        out = [None] * 8
        errs = []

        def call(i):
            try:
                y, dims = infer(m, "", "data_0", x[i:i + 1], "fc6_1", [1, 1000, 1, 1])
                assert dims == [1, 1000, 1, 1]
                out[i] = y.reshape(1000)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        before = B.BatcherStats(m)
        ts = [threading.Thread(target=call, args=(i,)) for i in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        after = B.BatcherStats(m)
        assert after["coalesced_requests"] - before["coalesced_requests"] == 8
        assert after["device_batches"] - before["device_batches"] < 8          # several callers shared a device batch
        assert rel_err(np.stack(out), ref) < 2e-5
        assert m.GetStats().InferenceCount == 9
        # a 3-image request is padded to the 4-row bucket and still returns its own 3 rows
        y3, dims3 = infer(m, "", "data_0", x[2:5], "fc6_1", [3, 1000, 1, 1])
        assert dims3 == [3, 1000, 1, 1] and rel_err(y3.reshape(3, 1000), ref[2:5]) < 2e-5
    finally:
        m.Destroy()


@pytest.mark.parametrize("tile", range(20))
def test_fp32_weights_stationary_1x1_kernel(tmp_path, tile):
    """conv1x1_ws_f32_kernel (weight slice resident in LDS, persistent workgroups, activations streamed from HBM straight into
    MFMA fragments through a register ring): every {channels per workgroup, waves} variant, multi-round persistent loops (14-19: the six shapes on
    a grid of one workgroup per CU)."""
    mb = models.densenet(3, growth=32, blocks=(2, 2), stem=128, image=112, classes=40, seed=78)     # K = 96 .. 192
    path = models.write_repo(str(tmp_path), "f32ws", mb)
    om = O.load_model(mb)
    x = models.synthetic_input((3, 3, 112, 112), stream="f32ws")
    ref = O.run(om, {"data_0": x}, dtype=np.float64)["fc6_1"]
    os.environ.update(IE_FORCE_ALGO="ws", IE_FORCE_TILE=str(tile))
    try:
        plan = B.DescribeModel(path, 3)["plan"]
        n1 = sum(1 for st in plan["steps"] if st.get("algo") == "ws1x1")
        m = B.CreateModel(path, "f32ws")
        try:
            y, _ = infer(m, "", "data_0", x, "fc6_1", [3, 40, 1, 1])
            y2, _ = infer(m, "", "data_0", x, "fc6_1", [3, 40, 1, 1])
        finally:
            m.Destroy()
    finally:
        for k_ in ("IE_FORCE_ALGO", "IE_FORCE_TILE"):
            os.environ.pop(k_, None)
    assert n1 >= 4, n1
    np.testing.assert_array_equal(y, y2)
    e = rel_err(y, ref)
    print(f"fp32 ws tile {tile}: {n1} convs on the weights-stationary kernel, rel err {e:.2e}")
    assert e < RTOL, (tile, e)


@pytest.mark.parametrize("tile", range(6))
@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_direct_split_k_kernel(tmp_path, tile, prec):
    """conv_direct_kernel (small output grids: K split over the waves of a workgroup, both operands loaded straight from global
    memory into MFMA fragments, partial tiles summed through LDS): every {channels, waves, chunks} variant, 1x1 convs with the
    BN+ReLU prologue and 3x3 convs with zero padding, both precisions."""
    mb = models.densenet(2, growth=32, blocks=(3, 3), stem=256, image=64, classes=24, seed=79)
    path = models.write_repo(str(tmp_path), "direct", mb)
    om = O.load_model(mb)
    x = models.synthetic_input((2, 3, 64, 64), stream="direct")
    ref = O.run(om, {"data_0": x}, dtype=np.float64)["fc6_1"]

    def go():
        plan = B.DescribeModel(path, 2)["plan"]
        nd = sum(1 for st in plan["steps"] if st.get("algo") == "direct")
        m = B.CreateModel(path, "direct")
        try:
            y, _ = infer(m, "", "data_0", x, "fc6_1", [2, 24, 1, 1])
            y2, _ = infer(m, "", "data_0", x, "fc6_1", [2, 24, 1, 1])
        finally:
            m.Destroy()
        return nd, y, y2
    nd, y, y2 = _run_with_env(dict(IE_PRECISION=prec, IE_FORCE_ALGO="direct", IE_FORCE_TILE=str(tile)), go)
    np.testing.assert_array_equal(y, y2)
    e = rel_err(y, ref)
    print(f"direct tile {tile} {prec}: {nd} convs on the direct split-K kernel, rel err {e:.2e}")
    assert nd >= 1, nd
    assert e < (RTOL if prec == "fp32" else F16_RTOL), (tile, prec, e)


@pytest.mark.parametrize("tile", [6, 7, 8, 9])
def test_direct_window_kernel(tmp_path, tile):
    """conv_win_kernel (direct tiles 6-9, fp32, 16x16x4 MFMA tiles): activations of a 16-pixel row block copied once into an LDS
    window (ragged last block, blocks that straddle two images), weights read from the fragment-major mirror, 16 / 32 output channels
    per workgroup, K split over 8 / 4 waves - 1x1 convs with the BN+ReLU prologue reading a slice of the concat buffer and 3x3 convs
    with zero padding."""
    mb = models.densenet(3, growth=32, blocks=(3, 3), stem=256, image=56, classes=24, seed=83)
    path = models.write_repo(str(tmp_path), "win", mb)
    om = O.load_model(mb)
    x = models.synthetic_input((3, 3, 56, 56), stream="win")
    ref = O.run(om, {"data_0": x}, dtype=np.float64)["fc6_1"]

    def go():
        plan = B.DescribeModel(path, 3)["plan"]
        nd = sum(1 for st in plan["steps"] if st.get("algo") == "direct")
        m = B.CreateModel(path, "win")
        try:
            y, _ = infer(m, "", "data_0", x, "fc6_1", [3, 24, 1, 1])
            y2, _ = infer(m, "", "data_0", x, "fc6_1", [3, 24, 1, 1])
            prof = {p_["kernel"] for p_ in B.Profile(m, 1)}
        finally:
            m.Destroy()
        return nd, y, y2, prof
    nd, y, y2, prof = _run_with_env(dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE=str(tile)), go)
    np.testing.assert_array_equal(y, y2)
    e = rel_err(y, ref)
    print(f"direct window tile {tile}: {nd} convs, kernels {sorted(prof)}, rel err {e:.2e}")
    assert nd >= 6, nd
    assert f"conv_win_kernel<f32,t{tile}>" in prof, prof        # ran on the window kernel, not on the implicit-GEMM fallback
    assert e < RTOL, (tile, e)


def test_window_kernel_random_same_size_convs(tmp_path):
    """Random stride-1 'same' convolutions (k = 1/3/5, 16-aligned Cin, Cout % 16 == 0, ragged pixel counts, prologue / bias / epilogue
    variants, channel-offset stores into the Concat buffer) on the window kernel vs the float64 oracle."""
    rs = np.random.RandomState(4242)
    done, worst = 0, 0.0
    for case in range(400):
        mb, ishape, oshape, desc = _random_conv_graph(rs, case, cin_choices=(16, 32, 48, 64, 128))
        if not (desc["stride"] == 1 and desc["pad"] == desc["k"] // 2 and desc["cout"] % 16 == 0 and desc["k"] <= 5):
            continue
        total = desc["k"] ** 2 * desc["cin"] // 16
        ok = [t for t, (tn, wv, mc) in ((6, (1, 8, 9)), (7, (2, 8, 9)), (8, (1, 4, 18)), (9, (2, 4, 18)))
              if wv <= total <= wv * mc and desc["cout"] % (16 * tn) == 0]
        if not ok:
            continue
        tile = ok[case % len(ok)]
        d = models.write_repo(str(tmp_path), f"w{case}", mb)
        x = rs.rand(*ishape).astype(np.float32)
        ref = O.run(O.load_model(mb), {"x": x}, dtype=np.float64)["out"]

        def go():
            m = B.CreateModel(d, "w")
            try:
                y, dims = infer(m, "", "x", x, "out", oshape)
                kern = {p_["kernel"] for p_ in B.Profile(m, 1)}
            finally:
                m.Destroy()
            return y, dims, kern
        y, dims, kern = _run_with_env(dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE=str(tile)), go)
        assert f"conv_win_kernel<f32,t{tile}>" in kern, (desc, kern)
        e = rel_err(y, ref)
        worst = max(worst, e)
        assert dims == list(oshape) and e < RTOL, (desc, tile, e)
        done += 1
        if done >= 16:
            break
    print(f"window kernel: {done} random same-size convs, worst rel err {worst:.2e}")
    assert done >= 10


def _pointwise_graph(rs, n, h, w, cin, cout, pre, bias, relu):
    """x -> 1x1 conv (3 -> cin, moves the tensor into NHWC) -> [BN + ReLU] -> 1x1 conv (cin -> cout) [+ bias] [-> ReLU]."""
    from gpu_ai_inference_server_amd.modelgen import onnx_pb as pb
    nodes, inits = [], []
    inits.append(pb.tensor("w0", (rs.randn(cin, 3, 1, 1) * 0.5).astype(np.float32)))
    nodes.append(pb.node("Conv", ["x", "w0"], ["h0"], "c0", [pb.attr_ints("kernel_shape", [1, 1])]))
    x = "h0"
    if pre:
        for nm, v in (("g", 1 + 0.1 * rs.randn(cin)), ("b", 0.1 * rs.randn(cin)), ("m", 0.1 * rs.randn(cin)), ("v", 0.5 + rs.rand(cin))):
            inits.append(pb.tensor("pre_" + nm, v.astype(np.float32)))
        nodes.append(pb.node("BatchNormalization", [x, "pre_g", "pre_b", "pre_m", "pre_v"], ["p0"], "prebn", [pb.attr_float("epsilon", 1e-5)]))
        nodes.append(pb.node("Relu", ["p0"], ["p1"], "prerelu"))
        x = "p1"
    inits.append(pb.tensor("w", (rs.randn(cout, cin, 1, 1) * np.sqrt(2.0 / cin)).astype(np.float32)))
    ins = [x, "w"]
    if bias:
        inits.append(pb.tensor("bvec", (0.2 * rs.randn(cout)).astype(np.float32)))
        ins.append("bvec")
    nodes.append(pb.node("Conv", ins, ["y0" if relu else "out"], "conv", [pb.attr_ints("kernel_shape", [1, 1])]))
    if relu:
        nodes.append(pb.node("Relu", ["y0"], ["out"], "postrelu"))
    g = pb.graph("pointwise", nodes, inits, [pb.value_info("x", [n, 3, h, w])], [pb.value_info("out", [n, cout, h, w])])
    return pb.model(g)


@pytest.mark.parametrize("tile", [10, 11, 12, 13, 14])
def test_activations_stationary_1x1_kernel(tmp_path, tile):
    """conv1x1_as_kernel (direct tiles 10-14, fp32): the workgroup's 32 (16) pixel rows staged once in LDS, weights streamed from the
    fragment-major mirror through a register ring - K from 16 (shorter than the ring) to 1008 (not a multiple of the ring depth),
    ragged last row block, prologue / bias / ReLU variants, 128 / 64 / 256 output channels per workgroup; plus the mini DenseNet."""
    rs = np.random.RandomState(500 + tile)
    per_wg = {10: 128, 11: 64, 12: 256, 13: 64, 14: 64}[tile]
    worst = 0.0
    for case, (n, h, w, cin) in enumerate([(1, 5, 7, 16), (2, 9, 9, 48), (3, 14, 14, 256), (2, 7, 13, 1008), (5, 6, 6, 144), (1, 12, 11, 400)]):
        cout = per_wg * (1 + case % 2)
        mb = _pointwise_graph(rs, n, h, w, cin, cout, pre=case % 3 != 0, bias=case % 2 == 0, relu=case % 4 != 1)
        d = models.write_repo(str(tmp_path), f"as{tile}_{case}", mb)
        x = rs.rand(n, 3, h, w).astype(np.float32)
        ref = O.run(O.load_model(mb), {"x": x}, dtype=np.float64)["out"]

        def go():
            m = B.CreateModel(d, "as")
            try:
                y, dims = infer(m, "", "x", x, "out", (n, cout, h, w))
                kern = {p_["kernel"] for p_ in B.Profile(m, 1)}
            finally:
                m.Destroy()
            return y, dims, kern
        y, dims, kern = _run_with_env(dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE=str(tile)), go)
        assert f"conv1x1_as_kernel<f32,t{tile}>" in kern, (case, kern)
        e = rel_err(y, ref)
        worst = max(worst, e)
        assert dims == [n, cout, h, w] and e < RTOL, (tile, case, e)
    if tile != 12:
        mb = models.densenet(3, growth=32, blocks=(3, 3), stem=256, image=56, classes=24, seed=87)
        path = models.write_repo(str(tmp_path), "asnet", mb)
        x = models.synthetic_input((3, 3, 56, 56), stream="as")
        ref = O.run(O.load_model(mb), {"data_0": x}, dtype=np.float64)["fc6_1"]

        def go_net():
            m = B.CreateModel(path, "asnet")
            try:
                y, _ = infer(m, "", "data_0", x, "fc6_1", [3, 24, 1, 1])
                kern = {p_["kernel"] for p_ in B.Profile(m, 1)}
            finally:
                m.Destroy()
            return y, kern
        y, kern = _run_with_env(dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE=str(tile)), go_net)
        assert f"conv1x1_as_kernel<f32,t{tile}>" in kern, kern       # the bottleneck 1x1s read a slice of the concat buffer
        worst = max(worst, rel_err(y, ref))
        assert rel_err(y, ref) < RTOL
    print(f"activations-stationary tile {tile}: worst rel err {worst:.2e}")


# ---------------------------------------------------------------------------------------------------------------------
# fp16 precision mode (BASELINE.json configs[2-3]).  The reference never runs fp16 (its ONNX Runtime session computes the
# model's own fp32), so there is no reference-side number to pin: "parity unpinned".  The checker is the float64 oracle /
# fixture; the tolerance is this repo's own statement: activations and weights are rounded to half (2^-11 relative per
# element), every accumulation, BN scale/shift, bias and split-K slab is fp32, so the error is a rounding random walk over
# the layers.  Bound used: 3e-3 of max|ref| per graph (observed 3e-4 ... 1e-3; DenseNet-121 6.5e-4) plus identical top-1 classes.
# ---------------------------------------------------------------------------------------------------------------------
F16_RTOL = 3e-3


def _f16_env(**extra):
    env = dict(IE_PRECISION="fp16")
    env.update(extra)
    return env


def _run_with_env(env, fn):
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k_ in env:
            os.environ.pop(k_, None)


def _f16_densenet_bytes():
    # every channel count / concat offset a multiple of 8: all dense-layer convs take the fp16 MFMA path
    return models.densenet(3, growth=16, blocks=(2, 3, 2), stem=32, image=48, classes=24, seed=77)


@pytest.mark.parametrize("name", sorted(MINI) + ["f16_densenet"])
def test_fp16_mode_graphs_vs_float64_oracle(model_repo, tmp_path, name):
    if name == "f16_densenet":
        mb, iname, ishape = _f16_densenet_bytes(), "data_0", (3, 3, 48, 48)
        path = models.write_repo(str(tmp_path), name, mb)
    else:
        mk, iname, ishape = MINI[name]
        mb = mk(models)
        path = os.path.join(model_repo, name, "1")
    om = O.load_model(mb)
    oname, oshape, _ = om.outputs[0]
    x = models.synthetic_input(ishape, stream=name)
    ref = O.run(om, {iname: x}, dtype=np.float64)[oname]

    def go():
        plan = B.DescribeModel(path, ishape[0])["plan"]
        assert plan["precision"] == "fp16"
        assert all(not st["in"]["f16"] for st in plan["steps"][:1]) and not plan["outputs"][0]["view"]["f16"]   # graph I/O stays fp32
        m = B.CreateModel(path, name)
        try:
            assert B.Precision(m) == "fp16"
            y, dims = infer(m, "", iname, x, oname, oshape)
            y2, _ = infer(m, "", iname, x, oname, oshape)
        finally:
            m.Destroy()
        return plan, y, y2, dims
    plan, y, y2, dims = _run_with_env(_f16_env(), go)
    assert dims == list(oshape)
    np.testing.assert_array_equal(y, y2)
    e = rel_err(y, ref)
    n16 = sum(1 for st in plan["steps"] if st["kind"] == "conv" and st["in"]["f16"] and st["algo"] == "igemm_vec")
    print(f"{name} fp16: rel err vs float64 oracle {e:.2e}; {n16} convs on the fp16 MFMA kernel")
    assert e < F16_RTOL
    if name == "f16_densenet":
        assert n16 >= 8


@pytest.mark.parametrize("tile", range(11))
@pytest.mark.parametrize("splitk", [1, 3])
def test_fp16_every_tile_and_split_k(tmp_path, tile, splitk):
    """Each tile configuration of conv_igemm_f16_kernel (base tiles and K-group tiles), with and without the two-pass split-K."""
    mb = _f16_densenet_bytes()
    path = models.write_repo(str(tmp_path), "f16t", mb)
    om = O.load_model(mb)
    x = models.synthetic_input((3, 3, 48, 48), stream="f16_densenet")
    ref = O.run(om, {"data_0": x}, dtype=np.float64)["fc6_1"]

    def go():
        plan = B.DescribeModel(path, 3)["plan"]
        forced = [st for st in plan["steps"] if st["kind"] == "conv" and st["in"]["f16"] and st["algo"] == "igemm_vec"]
        assert forced and all(st["tile"] == tile and st["splitk"] == splitk for st in forced)
        m = B.CreateModel(path, "f16t")
        try:
            return infer(m, "", "data_0", x, "fc6_1", [3, 24, 1, 1])[0]
        finally:
            m.Destroy()
    y = _run_with_env(_f16_env(IE_FORCE_ALGO="igemm", IE_FORCE_TILE=str(tile), IE_FORCE_SPLITK=str(splitk)), go)
    e = rel_err(y, ref)
    assert e < F16_RTOL, (tile, splitk, e)


@pytest.mark.parametrize("tile", range(18))
@pytest.mark.parametrize("image", [40, 112])
def test_fp16_weights_stationary_kernels(tmp_path, tile, image):
    """conv1x1_ws_f16_kernel (weights in LDS once per persistent workgroup, activations streamed into MFMA fragments) and
    conv3x3_ws_f16_kernel (all weights of the layer resident in LDS, raster window per 64-channel slice): every tile variant (1x1: six shapes x three
    grid policies -- persistent, one row block per wave, one workgroup per CU),
    single-tile and multi-tile persistent loops (image 112 -> 28x28 and 14x14 feature maps, several raster tiles per workgroup)."""
    mb = models.densenet(3, growth=32, blocks=(2, 2), stem=32, image=image, classes=40, seed=78)
    path = models.write_repo(str(tmp_path), "f16ws", mb)
    om = O.load_model(mb)
    x = models.synthetic_input((3, 3, image, image), stream="f16ws")
    ref = O.run(om, {"data_0": x}, dtype=np.float64)["fc6_1"]

    def go():
        plan = B.DescribeModel(path, 3)["plan"]
        n1 = sum(1 for st in plan["steps"] if st.get("algo") == "ws1x1")
        n3 = sum(1 for st in plan["steps"] if st.get("algo") == "ws3x3")
        m = B.CreateModel(path, "f16ws")
        try:
            return n1, n3, infer(m, "", "data_0", x, "fc6_1", [3, 40, 1, 1])[0]
        finally:
            m.Destroy()
    n1, n3, y = _run_with_env(_f16_env(IE_FORCE_ALGO="ws", IE_FORCE_TILE=str(tile)), go)
    assert n1 >= (3 if tile % 6 >= 2 else 2) and n3 == 4, (n1, n3)
    e = rel_err(y, ref)
    print(f"ws tile {tile} image {image}: {n1} 1x1 + {n3} 3x3 convs on the weights-stationary kernels, rel err {e:.2e}")
    assert e < F16_RTOL, (tile, e)


@pytest.mark.parametrize("seed", range(3))
def test_fp16_random_conv_graphs_vs_oracle(tmp_path, seed):
    """Seeded random convolutions with 8-aligned channels (kernel 1..7, stride 1/2, padding, fused BN+ReLU prologue /
    epilogue, channel-offset stores into a Concat buffer) on the fp16 path: autotuned, heuristic, forced K-group tile, split-K."""
    rs = np.random.RandomState(2000 + seed)
    modes = [dict(), dict(IE_AUTOTUNE="0"), dict(IE_FORCE_ALGO="igemm", IE_FORCE_TILE=str(7 + seed)),
             dict(IE_FORCE_ALGO="igemm", IE_FORCE_SPLITK="2", IE_FORCE_TILE=str(seed + 3)), dict(IE_FORCE_ALGO="naive"),
             dict(IE_FORCE_ALGO="ws", IE_FORCE_TILE=str(2 * seed)), dict(IE_FORCE_ALGO="direct", IE_FORCE_TILE=str(2 * seed + 1))]
    worst = 0.0
    for case in range(12):
        mb, ishape, oshape, desc = _random_conv_graph(rs, case, cin_choices=(8, 16, 24, 32, 64, 72, 96, 160))
        d = models.write_repo(str(tmp_path), f"h{seed}_{case}", mb)
        om = O.load_model(mb)
        x = rs.rand(*ishape).astype(np.float32)
        ref = O.run(om, {"x": x}, dtype=np.float64)["out"]
        env = _f16_env(**modes[case % len(modes)])

        def go():
            m = B.CreateModel(d, "h")
            try:
                return infer(m, "", "x", x, "out", oshape)
            finally:
                m.Destroy()
        y, dims = _run_with_env(env, go)
        assert dims == list(oshape)
        e = rel_err(y, ref)
        worst = max(worst, e)
        assert e < F16_RTOL, (desc, env, e)
    print(f"fp16 seed {seed}: worst rel err {worst:.2e}")


@pytest.mark.parametrize("batch,image,stem,blocks,band", [(3, 56, 128, (4, 3), 0), (5, 56, 256, (2, 5), 0), (2, 28, 192, (6,), 0), (9, 112, 192, (2, 4, 3), 0),
                                                        (3, 112, 64, (3, 2), 1), (2, 224, 64, (2, 2, 2), 1), (5, 80, 96, (2, 3), 1), (1, 288, 64, (1,), 1)])
def test_fp16_dense_block_chain_kernel(tmp_path, batch, image, stem, blocks, band):
    """dense_block_f16_kernel (kernels_block.hip): chains of dense layers (BN-ReLU-1x1 to 128, BN-ReLU-3x3 to 32) on 14x14 / 7x7 maps run as ONE
    launch per chain, one workgroup per image, the bottleneck tensor never leaving LDS.  DenseNet-shaped graphs whose blocks start at K0 =
    128 ... 256 channels (odd and even chunk counts, K % 64 == 32 tails, 7- and 2-tile maps, ragged chains cut where a layer would read what
    its predecessor writes too early): against the float64 oracle within F16_RTOL, and against the same plan run layer by layer.
    band = 1 (IE_DENSE_BAND): one layer per launch on larger maps -- STRIP mode on 56x56 / 28x28 / 20x20 maps (dense_strip_f16_kernel: a workgroup
    slides down a strip of rows, ring of bottleneck rows in LDS, ragged last steps / strips) and BAND mode on the 72x72 maps of the last case (a
    workgroup per band of rows, halo rows recomputed; wider maps such as 112x112 do not fit a band and stay layer by layer) -- measured no faster than the streaming kernels, so only formed on request."""
    mb = models.densenet(batch, growth=32, blocks=blocks, stem=stem, image=image, classes=24, seed=91)
    path = models.write_repo(str(tmp_path), "dblock", mb)
    x = models.synthetic_input((batch, 3, image, image), stream="dblock")
    ref = O.run(O.load_model(mb), {"data_0": x}, dtype=np.float64)["fc6_1"].reshape(batch, 24)

    def go():
        m = B.CreateModel(path, "dblock")
        try:
            y = infer(m, "", "data_0", x, "fc6_1", [batch, 24, 1, 1])[0].reshape(batch, 24).copy()
            din, _ = B.Prepare(m, [[batch, 3, image, image]], 1)
            B.CopyToDevice(m, din[0], x)
            B.RunPrepared(m, 1, True)
            return y, [p_["kernel"] for p_ in B.Profile(m, 1)], B.DescribeModel(path, batch)["plan"]
        finally:
            m.Destroy()
    y, kern, plan = _run_with_env(_f16_env(IE_AUTOTUNE="0", IE_DENSE_BAND=str(band)), go)
    y0, kern0, plan0 = _run_with_env(_f16_env(IE_AUTOTUNE="0", IE_NO_DENSE_BLOCK="1"), go)
    chains = [s for s in plan["steps"] if s.get("algo") == "dense_block" and s["tile"] != 0]
    if band:
        assert any(s["in"]["h"] * (s["in"]["w"] + 1) > 224 for s in chains), [(s["in"]["h"], s["in"]["w"]) for s in chains]
    assert chains and not [s for s in plan0["steps"] if s.get("algo") == "dense_block"]
    assert all(len(s["parts"]) % 2 == 0 and s["parts"][0]["k"] == [1, 1] and s["parts"][1]["k"] == [3, 3] for s in chains)
    nk = [k for k in kern if k.startswith("dense_block_f16_kernel")]
    assert len(nk) == len(chains) and not any(k.startswith("dense_block") for k in kern0), kern
    layers = sum(len(s["parts"]) // 2 for s in chains)
    e, e0, d = rel_err(y, ref), rel_err(y0, ref), rel_err(y, y0)
    print(f"dense-block chains B={batch} image={image} stem={stem} blocks={blocks}: {len(chains)} chains / {layers} layers "
          f"({[len(s['parts']) // 2 for s in chains]}), rel err {e:.2e} (layer by layer {e0:.2e}, between the two {d:.2e})")
    assert e < F16_RTOL and e0 < F16_RTOL and d < F16_RTOL


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
@pytest.mark.parametrize("batch,image,stem", [(1, 32, 64), (3, 50, 32), (2, 62, 64), (5, 224, 64), (2, 230, 16), (9, 112, 48)])
def test_stem_and_max_pool_in_one_launch(tmp_path, batch, image, stem, prec):
    """conv_stem_kernel<POOL> (kernels_stem.hip): the 7x7/s2 stem conv and the 3x3/s2/p1 max pool behind it as ONE step (plan algo "stem_pool"), the
    conv tile pooled in LDS -- half and float variants.  Even and odd conv / pooled sizes (ragged 7 x 7 pooled tiles, windows hanging over every
    image edge; widths that are and are not multiples of four: 16-byte and scalar window gathers), 16 ... 64 stem channels: bit-identical logits to
    the same plan run as two launches (IE_NO_STEM_POOL=1: the same MFMA sequence, the same rounding, max is exact) and within the mode's bound of
    the float64 oracle."""
    mb = models.densenet(batch, growth=16, blocks=(2, 2), stem=stem, image=image, classes=24, seed=57)
    path = models.write_repo(str(tmp_path), "stempool", mb)
    x = models.synthetic_input((batch, 3, image, image), stream="stempool")
    ref = O.run(O.load_model(mb), {"data_0": x}, dtype=np.float64)["fc6_1"].reshape(batch, 24)

    def go():
        m = B.CreateModel(path, "stempool")
        try:
            y = infer(m, "", "data_0", x, "fc6_1", [batch, 24, 1, 1])[0].reshape(batch, 24).copy()
            din, _ = B.Prepare(m, [[batch, 3, image, image]], 1)
            B.CopyToDevice(m, din[0], x)
            B.RunPrepared(m, 1, True)
            return y, [p_["kernel"] for p_ in B.Profile(m, 1)], B.DescribeModel(path, batch)["plan"]
        finally:
            m.Destroy()
    y, kern, plan = _run_with_env(dict(IE_PRECISION=prec, IE_AUTOTUNE="0"), go)
    y0, kern0, plan0 = _run_with_env(dict(IE_PRECISION=prec, IE_AUTOTUNE="0", IE_NO_STEM_POOL="1"), go)
    s0 = plan["steps"][0]
    conv_hw = (image + 6 - 7) // 2 + 1
    tag = "f16" if prec == "fp16" else "f32"
    assert s0["algo"] == "stem_pool" and s0["tile"] == 1 and s0["out"]["h"] == (conv_hw - 1) // 2 + 1 and plan0["steps"][0]["algo"] == "stem" and plan0["steps"][1]["kind"] == "pool"
    assert kern[0] == f"conv_stem_kernel<{tag},pool>" and kern0[0] == f"conv_stem_kernel<{tag}>" and kern0[1].startswith("pool_kernel") and len(kern) == len(kern0) - 1, (kern[:2], kern0[:3])
    e = rel_err(y, ref)
    print(f"stem + pool {prec} B={batch} image={image} ({conv_hw} -> {s0['out']['h']}) stem={stem}: rel err {e:.2e}, max |one launch - two launches| {np.abs(y - y0).max():.1e}")
    assert np.array_equal(y, y0)
    assert e < (F16_RTOL if prec == "fp16" else RTOL)


def test_fp16_densenet121_fixture_and_batch_independence(densenet_repo, tmp_path):
    """DenseNet-121 in fp16 mode selected through config.json ("precision": "fp16"), B=2 against the float64 fixture and the
    B=32 size-independent property (each image's logits equal that image run alone, up to summation-order rounding)."""
    import shutil
    root = str(tmp_path / "repo")
    shutil.copytree(densenet_repo, root)
    cfg_path = os.path.join(root, "densenet_onnx", "1", "config.json")
    with open(cfg_path) as f:
        cfg = f.read()
    with open(cfg_path, "w") as f:
        f.write(cfg[:-1] + ',"precision":"fp16"}')
    mgr = B.NewInferenceManager(root)
    try:
        mgr.LoadModel("densenet_onnx")
        g = np.load(os.path.join(GOLD, "densenet121_b2.npz"))
        x = models.synthetic_input((2, 3, 224, 224))
        y, dims = infer(mgr, "densenet_onnx", "data_0", x, "fc6_1", [2, 1000, 1, 1])
        assert dims == [2, 1000, 1, 1]
        e = rel_err(y.reshape(2, 1000), g["logits_f64"])
        print(f"densenet121 fp16 B=2: rel err vs float64 fixture {e:.2e}")
        assert e < F16_RTOL
        assert np.argmax(y.reshape(2, 1000), 1).tolist() == np.argmax(g["logits_f64"], 1).tolist()
        x32 = models.synthetic_input((32, 3, 224, 224), stream="b32")
        y32, _ = infer(mgr, "densenet_onnx", "data_0", x32, "fc6_1", [32, 1000, 1, 1])
        y32 = y32.reshape(32, 1000)
        assert np.isfinite(y32).all() and np.abs(y32).max() < 50
        for i in (0, 17, 31):
            y1, _ = infer(mgr, "densenet_onnx", "data_0", x32[i:i + 1], "fc6_1", [1, 1000, 1, 1])
            assert rel_err(y1.reshape(1000), y32[i]) < F16_RTOL
    finally:
        mgr.Shutdown()


def test_fp16_test_model_known_answer(model_repo):
    """The reference's pinned known answer (docs/run_server.ipynb:174-175) still holds to half precision in fp16 mode."""
    def go():
        m = B.CreateModel(os.path.join(model_repo, "test_model", "1"), "test_model")
        try:
            x = np.array([[-0.01349723, -1.0577109, 0.82254493]], np.float32)
            return infer(m, "", "input", x, "output", [1, 2])[0]
        finally:
            m.Destroy()
    y = _run_with_env(_f16_env(), go)
    np.testing.assert_allclose(np.asarray(y).reshape(-1), [-0.6017066, 1.8522782], rtol=5e-3, atol=5e-3)


def test_unknown_precision_is_a_load_error(model_repo):
    def go():
        with pytest.raises(RuntimeError, match="unsupported precision"):
            B.CreateModel(os.path.join(model_repo, "test_model", "1"), "test_model")
    _run_with_env(dict(IE_PRECISION="int3"), go)


# ---------------------------------------------------------------------------------------------------------------------
# ResNet (BASELINE.json configs[4] names ResNet-50): residual Add + ReLU, strided 1x1 / 3x3 convs, Conv->BN without ReLU, Gemm
# classifier.  The reference holds no ResNet file or output, so this family is "parity unpinned" by the reference as well: the
# checker is the float64 oracle on the same synthetic graph.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", range(5))
def test_fp16_weights_stationary_3x3_wide_outputs(tmp_path, tile):
    """conv3x3_ws_f16_kernel with more than 32 output channels (one resident weight set per 32-channel N-tile, blockIdx.y): the
    64->64 and 128->128 3x3 convs of a ResNet, every tile variant."""
    mb = models.resnet(2, layers=(2, 2, 1, 1), width=64, image=64, classes=20, seed=52)
    path = models.write_repo(str(tmp_path), "resnet_ws3", mb)
    om = O.load_model(mb)
    x = models.synthetic_input((2, 3, 64, 64), stream="resnet_ws3")
    ref = O.run(om, {"data": x}, dtype=np.float64)["logits"]

    def go():
        plan = B.DescribeModel(path, 2)["plan"]
        n3 = [st for st in plan["steps"] if st.get("algo") == "ws3x3"]
        m = B.CreateModel(path, "resnet_ws3")
        try:
            return n3, infer(m, "", "data", x, "logits", [2, 20])[0]
        finally:
            m.Destroy()
    n3, y = _run_with_env(dict(IE_PRECISION="fp16", IE_FORCE_ALGO="ws", IE_FORCE_TILE=str(tile)), go)
    assert len(n3) >= 3 and {st["out"]["c"] for st in n3} >= {64, 128}, [(st["in"]["c"], st["out"]["c"]) for st in n3]
    e = rel_err(y, ref)
    print(f"ws3x3 wide tile {tile}: {len(n3)} convs, rel err {e:.2e}")
    assert e < F16_RTOL


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_resnet_mini_vs_float64_oracle(tmp_path, prec):
    mb = models.resnet(3, layers=(2, 2, 2, 2), width=16, image=64, classes=20, seed=51)
    path = models.write_repo(str(tmp_path), "resnet_mini", mb)
    om = O.load_model(mb)
    x = models.synthetic_input((3, 3, 64, 64), stream="resnet")
    ref = O.run(om, {"data": x}, dtype=np.float64)["logits"]

    def go():
        m = B.CreateModel(path, "resnet_mini")
        try:
            return infer(m, "", "data", x, "logits", [3, 20])
        finally:
            m.Destroy()
    y, dims = _run_with_env(dict(IE_PRECISION=prec), go)
    assert dims == [3, 20]
    e = rel_err(y, ref)
    print(f"resnet mini {prec}: rel err vs float64 oracle {e:.2e}")
    assert e < (RTOL if prec == "fp32" else F16_RTOL)


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_resnet50_b2_vs_float64_oracle(tmp_path, prec):
    """The full 53-conv ResNet-50 graph at 224x224, batch 2, against the float64 oracle (same seeded weights and inputs)."""
    mb = models.resnet50("N")
    path = models.write_repo(str(tmp_path), "resnet50", mb)
    om = O.load_model(mb)
    x = models.synthetic_input((2, 3, 224, 224), stream="resnet50")
    ref = O.run(om, {"data": x}, dtype=np.float64)["logits"]

    def go():
        m = B.CreateModel(path, "resnet50")
        try:
            return infer(m, "", "data", x, "logits", [2, 1000])
        finally:
            m.Destroy()
    y, dims = _run_with_env(dict(IE_PRECISION=prec), go)
    assert dims == [2, 1000]
    e = rel_err(y, ref)
    print(f"resnet50 {prec} B=2: rel err vs float64 oracle {e:.2e}")
    assert e < (RTOL if prec == "fp32" else F16_RTOL)
    assert np.argmax(y, 1).tolist() == np.argmax(ref, 1).tolist()


def test_uint8_ingest_matches_float_path(densenet_repo, tmp_path):
    """a FLOAT32 graph input may be fed as UINT8 bytes (DATATYPE_UINT8 exists in the reference ABI,
    inference_bridge.h:25, but its ModelInfer rejects everything except FLOAT32); the engine uploads the bytes (4x fewer over
    PCIe) and converts on the device with x * uint8_scale + uint8_bias (config.json; default 1/255, 0).  The result must be
    bit-identical to feeding the same converted floats."""
    import shutil
    rs = np.random.RandomState(7)
    xb = rs.randint(0, 256, size=(5, 3, 224, 224)).astype(np.uint8)
    for scale, bias, cfg_extra in ((np.float32(1.0) / np.float32(255.0), np.float32(0.0), ""),
                                   (np.float32(0.017), np.float32(-1.5), ',"uint8_scale":0.017,"uint8_bias":-1.5')):
        root = str(tmp_path / f"repo{len(cfg_extra)}")
        shutil.copytree(densenet_repo, root)
        cfg_path = os.path.join(root, "densenet_onnx", "1", "config.json")
        with open(cfg_path) as f:
            cfg = f.read()
        with open(cfg_path, "w") as f:
            f.write(cfg[:-1] + cfg_extra + "}")
        m = B.CreateModel(os.path.join(root, "densenet_onnx", "1"), "densenet_onnx")
        try:
            xf = xb.astype(np.float32) * scale + bias
            outs = [B.OutputConfig("fc6_1", Shape=[5, 1000, 1, 1], DataType="FLOAT32")]
            yf = m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([5, 3, 224, 224]), xf)], outs)[0].Data.copy()
            yu = m.Infer([B.TensorData("data_0", B.DataTypeUint8, B.Shape([5, 3, 224, 224]), xb)], outs)[0].Data.copy()
            assert np.isfinite(yu).all()
            np.testing.assert_array_equal(yu, yf)
            # a short payload is zero-extended like the reference's zero-initialised Tensor (bytes 0 -> bias)
            yshort = m.Infer([B.TensorData("data_0", B.DataTypeUint8, B.Shape([5, 3, 224, 224]), xb.reshape(-1)[:3 * 224 * 224 * 2])], outs)[0].Data
            xpad = np.zeros_like(xb)
            xpad.reshape(-1)[:3 * 224 * 224 * 2] = xb.reshape(-1)[:3 * 224 * 224 * 2]
            yref = m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([5, 3, 224, 224]), xpad.astype(np.float32) * scale + bias)], outs)[0].Data
            np.testing.assert_array_equal(yshort, yref)
        finally:
            m.Destroy()


def test_roctx_ranges_do_not_disturb_inference(model_repo):
    """IE_ROCTX=1: the marker library is dlopen'ed and every ModelInfer call is wrapped in a range; results are unchanged."""
    code = (
        "import sys, os, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'tests'))\n"
        "from _pkg import load_package; load_package()\n"
        "from gpu_ai_inference_server_amd import binding as B\n"
        f"m = B.CreateModel(os.path.join({model_repo!r}, 'test_model', '1'), 'test_model')\n"
        "x = np.array([[-0.01349723, -1.0577109, 0.82254493]], np.float32)\n"
        "y = m.Infer([B.TensorData('input', B.DataTypeFloat32, B.Shape([1, 3]), x)], [B.OutputConfig('output', [1, 2])])[0].Data\n"
        "print('OUT', float(y[0]), float(y[1]))\n"
        "m.Destroy()\n")
    env = dict(os.environ, IE_ROCTX="1")
    r = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("OUT")][0].split()
    np.testing.assert_allclose([float(out[1]), float(out[2])], [-0.6017066, 1.8522782], rtol=2e-6)


def test_in_process_batch_sharding(densenet_repo):
    """SURVEY §8e (single-process plan): a ModelInfer request is cut into contiguous row slices, one per model replica, run
    concurrently and scattered back by offset.  On the 1-GPU box the replicas all live on device 0 (IE_SHARD_DEVICES=0,0,0), which
    exercises the slicing / threading / scatter logic; uneven slices (7 rows over 3 replicas), FLOAT32 and UINT8 payloads."""
    path = os.path.join(densenet_repo, "densenet_onnx", "1")
    x = models.synthetic_input((7, 3, 224, 224), stream="shard")
    xb = np.clip(x * 255.0, 0, 255).astype(np.uint8)
    outs = [B.OutputConfig("fc6_1", Shape=[7, 1000, 1, 1], DataType="FLOAT32")]
    m = B.CreateModel(path, "densenet_onnx")
    try:
        assert B.ShardStats(m) == (1, 0)
        y1 = m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([7, 3, 224, 224]), x)], outs)[0].Data.copy()
        y1u = m.Infer([B.TensorData("data_0", B.DataTypeUint8, B.Shape([7, 3, 224, 224]), xb)], outs)[0].Data.copy()
    finally:
        m.Destroy()
    os.environ["IE_SHARD_DEVICES"] = "0,0,0"
    try:
        m = B.CreateModel(path, "densenet_onnx")
        try:
            assert B.ShardStats(m)[0] == 3
            r = m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([7, 3, 224, 224]), x)], outs)[0]
            y3 = r.Data.copy()
            assert r.Shape.Dims == [7, 1000, 1, 1]
            y3u = m.Infer([B.TensorData("data_0", B.DataTypeUint8, B.Shape([7, 3, 224, 224]), xb)], outs)[0].Data.copy()
            assert B.ShardStats(m) == (3, 2)
            # fewer rows than replicas: served by the primary alone
            y2 = m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([2, 3, 224, 224]), x[:2])],
                         [B.OutputConfig("fc6_1", Shape=[2, 1000, 1, 1], DataType="FLOAT32")])[0].Data.copy()
            assert B.ShardStats(m) == (3, 2)
        finally:
            m.Destroy()
    finally:
        del os.environ["IE_SHARD_DEVICES"]
    # slices run at other batch sizes (other tiles / summation orders): equal to the unsharded run up to fp32 rounding
    assert rel_err(y3, y1) < 2e-5 and rel_err(y3u, y1u) < 2e-5 and rel_err(y2, y1.reshape(7, -1)[:2].reshape(-1)) < 2e-5


def test_plan_cache_is_bounded(model_repo):
    """A server sees arbitrary batch sizes: plan instances (activation buffers + hipGraph per input shape) are kept in an LRU of
    IE_MAX_PLANS entries; evicted shapes are rebuilt on demand with identical results."""
    mb = models.densenet("N", growth=8, blocks=(2, 2), stem=16, image=32, classes=10, seed=9)

    def go(tmp):
        path = models.write_repo(tmp, "lru", mb)
        m = B.CreateModel(path, "lru")
        try:
            outs = {}
            for rep in range(2):
                for b in (1, 2, 3, 4, 5):
                    x = models.synthetic_input((b, 3, 32, 32), stream=f"lru{b}")
                    y, _ = infer(m, "", "data_0", x, "fc6_1", [b, 10, 1, 1])
                    if rep == 0:
                        outs[b] = y.copy()
                    else:
                        np.testing.assert_array_equal(y, outs[b])       # rebuilt after eviction: same plan, same kernels, same bits
        finally:
            m.Destroy()
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        _run_with_env(dict(IE_MAX_PLANS="2", IE_AUTOTUNE="0"), lambda: go(tmp))


def test_output_dims_survive_plan_eviction_under_concurrency(tmp_path):
    """the sharded / coalesced paths described their outputs from a plan instance AFTER releasing its lane; with a full
    plan cache another request's Prepare frees that instance.  IE_MAX_PLANS=1, two shard lanes on device 0, three threads alternating
    batch sizes through the sharded path (rows >= shards) and the single-lane path (rows < shards): every call must report its own
    dims and its own logits."""
    mb = models.densenet("N", growth=8, blocks=(2, 2), stem=16, image=32, classes=10, seed=9)
    path = models.write_repo(str(tmp_path), "evict", mb)
    xs = {b: models.synthetic_input((b, 3, 32, 32), stream=f"evict{b}") for b in (1, 2, 3, 4, 6)}

    def go():
        m = B.CreateModel(path, "evict")
        try:
            assert B.ShardStats(m)[0] == 2
            ref = {b: infer(m, "", "data_0", xs[b], "fc6_1", [b, 10, 1, 1])[0].copy() for b in xs}
            errs = []

            def worker(order):
                try:
                    for _ in range(12):
                        for b in order:
                            y, dims = infer(m, "", "data_0", xs[b], "fc6_1", [b, 10, 1, 1])
                            assert dims == [b, 10, 1, 1], (b, dims)
                            assert rel_err(y, ref[b]) < 2e-5, b
                except Exception as e:  # noqa: BLE001
                    errs.append(e)
            ts = [threading.Thread(target=worker, args=(o,)) for o in ((1, 4, 2, 6, 3), (6, 3, 1, 2, 4), (2, 6, 4, 3, 1))]
            [t.start() for t in ts]
            [t.join() for t in ts]
            assert not errs, errs
        finally:
            m.Destroy()
    _run_with_env(dict(IE_MAX_PLANS="1", IE_AUTOTUNE="0", IE_SHARD_DEVICES="0,0"), go)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs at their own sizes, and the multi-input ordering rule
# ---------------------------------------------------------------------------------------------------------------------
def _with_precision_config(densenet_repo, tmp_path, extra):
    import shutil
    root = str(tmp_path / "repo")
    shutil.copytree(densenet_repo, root)
    cfg_path = os.path.join(root, "densenet_onnx", "1", "config.json")
    with open(cfg_path) as f:
        cfg = f.read()
    with open(cfg_path, "w") as f:
        f.write(cfg[:-1] + extra + "}")
    return root


def test_config2_fp16_densenet121_b128(densenet_repo, tmp_path):
    """BASELINE configs[2] at its own size: DenseNet-121, fp16 mode, batch 128 through ModelInfer (77 MB FLOAT32 payload).
    Other output grids, other kernel choices and multi-round persistent loops than the B=2 / B=32 cases.  Size-independent
    property: each image's logits equal that image run alone (up to fp16 summation-order rounding); plus the float64 oracle
    on 2 of the 128 images, and a batch permutation."""
    root = _with_precision_config(densenet_repo, tmp_path, ',"precision":"fp16","tune_batches":[1,128]')     # the SEARCHED kernel set bench.py times
    mgr = B.NewInferenceManager(root)
    try:
        mgr.LoadModel("densenet_onnx")
        x = models.synthetic_input((128, 3, 224, 224), stream="b128")
        y, dims = infer(mgr, "densenet_onnx", "data_0", x, "fc6_1", [128, 1000, 1, 1])
        assert dims == [128, 1000, 1, 1]
        y = y.reshape(128, 1000)
        assert np.isfinite(y).all() and np.abs(y).max() < 50
        for i in (0, 61, 100, 127):
            y1, _ = infer(mgr, "densenet_onnx", "data_0", x[i:i + 1], "fc6_1", [1, 1000, 1, 1])
            e = rel_err(y1.reshape(1000), y[i])
            assert e < F16_RTOL, (i, e)
        perm = np.random.RandomState(1).permutation(128)
        yp, _ = infer(mgr, "densenet_onnx", "data_0", x[perm], "fc6_1", [128, 1000, 1, 1])
        assert rel_err(yp.reshape(128, 1000), y[perm]) < F16_RTOL
        om = O.load_model(models.densenet121(2))
        yo = O.run(om, {"data_0": x[[5, 90]]}, dtype=np.float64)["fc6_1"].reshape(2, 1000)
        e = rel_err(y[[5, 90]], yo)
        print(f"densenet121 fp16 B=128: rel err vs float64 oracle on images 5, 90: {e:.2e}")
        assert e < F16_RTOL
        assert np.argmax(y[[5, 90]], 1).tolist() == np.argmax(yo, 1).tolist()
    finally:
        mgr.Shutdown()


def test_config3_fp16_b1024_over_8_shards(densenet_repo, tmp_path):
    """BASELINE configs[3] on the one GPU of this box: fp16 DenseNet-121, one ModelInfer request of 1024 images (616 MB) cut
    into 8 contiguous slices of 128 over 8 replicas (IE_SHARD_DEVICES=0,...: all on device 0 here; on an 8-GPU node the same
    code path puts one replica per device).  Every slice must equal the unsharded engine's answer for those 128 images --
    bit for bit, since replicas run the same plan (the kernel search is pinned off so every replica picks the same kernels)."""
    root = _with_precision_config(densenet_repo, tmp_path, ',"precision":"fp16"')
    path = os.path.join(root, "densenet_onnx", "1")
    base = models.synthetic_input((128, 3, 224, 224), stream="b1024")
    rs = np.random.RandomState(3)
    perms = [np.arange(128)] + [rs.permutation(128) for _ in range(7)]
    x = np.concatenate([base[p_] for p_ in perms], 0)
    assert x.shape == (1024, 3, 224, 224)
    outs128 = [B.OutputConfig("fc6_1", Shape=[128, 1000, 1, 1], DataType="FLOAT32")]
    outs1024 = [B.OutputConfig("fc6_1", Shape=[1024, 1000, 1, 1], DataType="FLOAT32")]

    def unsharded():
        m = B.CreateModel(path, "densenet_onnx")
        try:
            assert B.ShardStats(m) == (1, 0)
            return m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([128, 3, 224, 224]), base)], outs128)[0].Data.reshape(128, 1000).copy()
        finally:
            m.Destroy()

    def sharded():
        m = B.CreateModel(path, "densenet_onnx")
        try:
            assert B.ShardStats(m) == (8, 0)
            r = m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([1024, 3, 224, 224]), x)], outs1024)[0]
            assert r.Shape.Dims == [1024, 1000, 1, 1]
            assert B.ShardStats(m) == (8, 1)
            return r.Data.reshape(1024, 1000).copy()
        finally:
            m.Destroy()
    y128 = _run_with_env(dict(IE_AUTOTUNE="0"), unsharded)
    y = _run_with_env(dict(IE_AUTOTUNE="0", IE_SHARD_DEVICES="0,0,0,0,0,0,0,0"), sharded)
    assert np.isfinite(y).all()
    for k, p_ in enumerate(perms):
        got = y[128 * k:128 * (k + 1)]
        if k == 0:
            np.testing.assert_array_equal(got, y128)        # same rows, same plan: identical bits
        else:
            assert rel_err(got, y128[p_]) < F16_RTOL, k   # a permuted slice: other images share a tile, fp16 rounding only
            assert (np.argmax(got, 1) == np.argmax(y128[p_], 1)).mean() > 0.98


def test_two_input_graph_orders_inputs_by_graph_index(mgr):
    """InferONNX (model.cpp:1174-1190) places each payload at the graph index of its NAME, whatever order the caller lists
    the tensors in; graph order here is (b_in, a_in)."""
    g = np.load(os.path.join(GOLD, "mini_two_input.npz"))["output_f64"]
    xa = models.synthetic_input((2, 8, 12, 12), stream="two_input/a")
    xb = models.synthetic_input((2, 16, 12, 12), stream="two_input/b")
    ta = B.TensorData("a_in", B.DataTypeFloat32, B.Shape([2, 8, 12, 12]), xa)
    tb = B.TensorData("b_in", B.DataTypeFloat32, B.Shape([2, 16, 12, 12]), xb)
    outs = [B.OutputConfig("y", Shape=[2, 24, 1, 1], DataType="FLOAT32")]
    mgr.LoadModel("mini_two_input")
    try:
        m = mgr.GetModel("mini_two_input")
        assert m.GetMetadata().Inputs == ["b_in", "a_in"]
        y_ab = m.Infer([ta, tb], outs)[0].Data.copy()
        y_ba = m.Infer([tb, ta], outs)[0].Data.copy()
        np.testing.assert_array_equal(y_ab, y_ba)
        assert rel_err(y_ab.reshape(g.shape), g) < RTOL
        yo = O.run(O.load_model(models.two_input_graph(2)), {"a_in": xa, "b_in": xb})["y"]
        assert rel_err(y_ab.reshape(yo.shape), yo) < RTOL
        with pytest.raises(RuntimeError, match="Expected 2 inputs, got 1"):
            m.Infer([ta], outs)
        with pytest.raises(RuntimeError, match="Required input tensor not provided: b_in"):
            m.Infer([ta, ta], outs)
        # swapped payloads under the right names are a shape error, not a silent mix-up
        bad_a = B.TensorData("a_in", B.DataTypeFloat32, B.Shape([2, 16, 12, 12]), xb)
        bad_b = B.TensorData("b_in", B.DataTypeFloat32, B.Shape([2, 8, 12, 12]), xa)
        with pytest.raises(RuntimeError, match="Got invalid dimensions for input"):
            m.Infer([bad_a, bad_b], outs)
    finally:
        mgr.UnloadModel("mini_two_input")


# ---------------------------------------------------------------------------------------------------------------------
# Lanes (instance_count), in-process RCCL weight broadcast, pipelined host path, tuning off the request path
# ---------------------------------------------------------------------------------------------------------------------
def test_instance_count_runs_two_requests_side_by_side(densenet_repo, tmp_path):
    """config.json "instance_count": 2 (the field model.h:63 carries and the reference never reads): two execution lanes on the
    device sharing one weight blob; two threads calling ModelInfer overlap (the lane pool's high-water mark reaches 2) and every
    call returns exactly what the serial run returned."""
    root = _with_precision_config(densenet_repo, tmp_path, ',"instance_count":2')
    m = B.CreateModel(os.path.join(root, "densenet_onnx", "1"), "densenet_onnx")
    try:
        info = B.RuntimeInfo(m)
        assert info["lanes"] == 2 and info["shards"] == 1 and info["lane_shares_weights_with"] == [0, 0] and not info["rccl"]["used"]
        xs = [models.synthetic_input((8, 3, 224, 224), stream=f"lane{i}") for i in range(2)]
        serial = [infer(m, "", "data_0", x, "fc6_1", [8, 1000, 1, 1])[0].copy() for x in xs]
        for lane_warm in range(2):          # the second lane plans B=8 on its first use
            infer(m, "", "data_0", xs[0], "fc6_1", [8, 1000, 1, 1])
        outs = {0: [], 1: []}
        errs = []
        gate = threading.Barrier(2)

        def worker(i):
            try:
                gate.wait()
                for _ in range(12):
                    outs[i].append(infer(m, "", "data_0", xs[i], "fc6_1", [8, 1000, 1, 1])[0].copy())
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        for i in range(2):
            for y in outs[i]:
                np.testing.assert_array_equal(y, serial[i])
        assert B.RuntimeInfo(m)["max_in_flight"] == 2
        assert "mi355x-engine precision=fp32 lanes=2" in m.GetMetadata().Description
    finally:
        m.Destroy()


def test_rccl_weight_broadcast_to_a_private_replica(densenet_repo):
    """The in-process sharding fills a replica that owns its weights with ONE ncclBroadcast of the packed blob at load.
    On the 1-GPU box: IE_SHARD_DEVICES=0,0 with IE_SHARD_PRIVATE_WEIGHTS=1 gives the replica its own (zeroed, never uploaded)
    allocation on device 0; RCCL (one rank, out-of-place broadcast) is what moves the 32 MB.  The replica's blob checksum in HBM
    must equal the primary's, and the sharded answer must equal the unsharded one."""
    path = os.path.join(densenet_repo, "densenet_onnx", "1")
    x = models.synthetic_input((8, 3, 224, 224), stream="rccl")
    outs = [B.OutputConfig("fc6_1", Shape=[8, 1000, 1, 1], DataType="FLOAT32")]
    m = B.CreateModel(path, "densenet_onnx")
    try:
        y1 = m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([8, 3, 224, 224]), x)], outs)[0].Data.copy()
        c1 = B.RuntimeInfo(m, checksums=True)["weight_checksums"]
    finally:
        m.Destroy()

    def go():
        m2 = B.CreateModel(path, "densenet_onnx")
        try:
            info = B.RuntimeInfo(m2, checksums=True)
            y2 = m2.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([8, 3, 224, 224]), x)], outs)[0].Data.copy()
            return info, y2, B.ShardStats(m2)
        finally:
            m2.Destroy()
    info, y2, st = _run_with_env(dict(IE_SHARD_DEVICES="0,0", IE_SHARD_PRIVATE_WEIGHTS="1"), go)
    assert info["lanes"] == 2 and info["shards"] == 2 and info["lane_shares_weights_with"] == [0, 1]
    r = info["rccl"]
    assert r["used"] and r["ranks"] == 1 and r["weight_owners"] == 2 and r["bytes"] > 30e6 and r["broadcast_ms"] > 0
    assert info["weight_checksums"][0] == info["weight_checksums"][1] == c1[0]
    assert st == (2, 1)
    assert rel_err(y2, y1) < 2e-5
    # without private weights a same-device replica shares the primary's blob: nothing to broadcast
    info3 = _run_with_env(dict(IE_SHARD_DEVICES="0,0"), lambda: (lambda mm: (B.RuntimeInfo(mm), mm.Destroy())[0])(B.CreateModel(path, "densenet_onnx")))
    assert info3["lane_shares_weights_with"] == [0, 0] and not info3["rccl"]["used"]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp8"])
def test_private_replicas_hold_the_primarys_blob_and_every_mirror(tmp_path, prec):
    """VERDICT r2 #7a, ADVICE r2: four shard lanes on device 0 that each OWN a (zeroed, never uploaded) weight allocation
    (IE_SHARD_PRIVATE_WEIGHTS=1): after the load-time broadcast + WeightsArrived every replica's fp32 blob AND everything derived from
    it (half mirror; fragment-major + Winograd U; e4m3 weights, their row scales / epilogue multipliers and the adopted activation
    scales) must be bit-identical to the primary's, the sharded answer must equal the unsharded one, and an in-place weight update
    (EngineWeightsUpdated) must reach every owner and every captured graph."""
    if prec == "fp8":
        mb, iname, oname = models.resnet("N", layers=(2, 1, 2, 1), width=16, image=64, classes=20, seed=51), "data", "logits"
        oshape, xshape = [8, 20], (8, 3, 64, 64)
    else:
        mb, iname, oname = models.densenet("N", growth=32, blocks=(2, 3, 2), stem=64, image=64, classes=24, seed=78), "data_0", "fc6_1"
        oshape, xshape = [8, 24, 1, 1], (8, 3, 64, 64)
    path = models.write_repo(str(tmp_path), "priv", mb)
    x = models.synthetic_input(xshape, stream="priv")

    def run(update):
        m = B.CreateModel(path, "priv")
        try:
            y = infer(m, "", iname, x, oname, oshape)[0].copy()
            if update:            # scale the whole blob by 1/2 in place on the primary, tell the engine, run again
                ptr, nbytes = B.GetWeightBlob(m)
                w = np.empty(nbytes // 4, np.float32)
                B.CopyToHost(m, w, ptr)
                B.CopyToDevice(m, ptr, (w * np.float32(0.5)).astype(np.float32))
                B.WeightsUpdated(m)
                y2 = infer(m, "", iname, x, oname, oshape)[0].copy()
            else:
                y2 = None
            return y, y2, B.RuntimeInfo(m, checksums=True), B.ShardStats(m)
        finally:
            m.Destroy()
    y1, y1u, info1, _ = _run_with_env(dict(IE_PRECISION=prec), lambda: run(True))
    y4, y4u, info4, st = _run_with_env(dict(IE_PRECISION=prec, IE_SHARD_DEVICES="0,0,0,0", IE_SHARD_PRIVATE_WEIGHTS="1"), lambda: run(True))
    assert info4["lanes"] == 4 and info4["lane_shares_weights_with"] == [0, 1, 2, 3] and st[0] == 4 and st[1] >= 2
    assert info4["rccl"]["used"] and info4["rccl"]["weight_owners"] == 4
    assert len(set(info4["weight_checksums"])) == 1 and info4["weight_checksums"][0] == info1["weight_checksums"][0]
    mc = info4["mirror_checksums"]
    want = {"fp32": {"fragment_major", "winograd_u"}, "fp16": {"half"}, "fp8": {"half", "e4m3", "e4m3_scales", "act_scales"}}[prec]
    assert want <= set(mc[0]), mc[0]
    for k in range(1, 4):
        assert mc[k] == mc[0], (k, mc[k], mc[0])
    assert mc[0] == info1["mirror_checksums"][0]
    tol = {"fp32": 2e-5, "fp16": F16_RTOL, "fp8": F8_EMU_RTOL * 4}[prec]
    assert rel_err(y4, y1) < tol and rel_err(y4u, y1u) < tol
    assert rel_err(y1u, y1) > 1e-3            # the update really changed the answer (so stale graphs / stale replicas would show)


def test_pipelined_host_path_matches_single_shot(densenet_repo):
    """ModelInfer uploads a batch in image ranges and runs the first plan steps per range while the next range uploads; the rest runs
    once on the whole batch.  Same arithmetic, other tile choices for the per-range launches: equal to the single-shot path up to
    fp32 summation order; B=32 FLOAT32 and UINT8 payloads, plus a ragged short payload."""
    path = os.path.join(densenet_repo, "densenet_onnx", "1")
    x = models.synthetic_input((32, 3, 224, 224), stream="pipe")
    xb = np.clip(x * 255.0, 0, 255).astype(np.uint8)
    outs = [B.OutputConfig("fc6_1", Shape=[32, 1000, 1, 1], DataType="FLOAT32")]

    def run(expect_chunks):
        m = B.CreateModel(path, "densenet_onnx")
        try:
            yf = m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([32, 3, 224, 224]), x)], outs)[0].Data.copy()
            info = B.RuntimeInfo(m)
            yu = m.Infer([B.TensorData("data_0", B.DataTypeUint8, B.Shape([32, 3, 224, 224]), xb)], outs)[0].Data.copy()
            short = x.reshape(-1)[:3 * 224 * 224 * 13 + 1000]                     # ends inside image 13: zero-extended
            ys = m.Infer([B.TensorData("data_0", B.DataTypeFloat32, B.Shape([32, 3, 224, 224]), short)], outs)[0].Data.copy()
            assert info["last_chunks"] == expect_chunks, info
            if expect_chunks > 1:
                assert info["last_head_steps"] >= 2 and info["pipelined_calls"][0] >= 1
            return yf, yu, ys
        finally:
            m.Destroy()
    yf, yu, ys = run(2)
    yf0, yu0, ys0 = _run_with_env(dict(IE_PIPELINE_CHUNKS="0"), lambda: run(1))
    assert rel_err(yf, yf0) < 2e-5 and rel_err(yu, yu0) < 2e-5 and rel_err(ys, ys0) < 2e-5
    assert rel_err(ys.reshape(32, 1000)[:13], yf.reshape(32, 1000)[:13]) < 2e-5        # whole images before the cut are unaffected
    assert np.abs(ys.reshape(32, 1000)[14:] - ys.reshape(32, 1000)[14]).max() < 1e-5   # all-zero images give one common answer


def test_dynamic_batcher_feeds_the_sharder(densenet_repo):
    """A coalesced batch is cut over the shard replicas like a single large request (round 1: the two were mutually exclusive)."""
    x = models.synthetic_input((8, 3, 224, 224), stream="batcher")
    path = os.path.join(densenet_repo, "densenet_onnx", "1")
    ref_m = B.CreateModel(path, "densenet_onnx")
    ref = np.stack([infer(ref_m, "", "data_0", x[i:i + 1], "fc6_1", [1, 1000, 1, 1])[0].reshape(1000) for i in range(8)])
    ref_m.Destroy()

    def go():
        m = B.CreateModel(path, "densenet_onnx")
        try:
            assert B.BatcherStats(m)["max_batch"] == 8 and B.ShardStats(m)[0] == 2
            out = [None] * 8
            errs = []

            def call(i):
                try:
                    out[i] = infer(m, "", "data_0", x[i:i + 1], "fc6_1", [1, 1000, 1, 1])[0].reshape(1000)
                except Exception as e:  # noqa: BLE001
                    errs.append(e)
            ts = [threading.Thread(target=call, args=(i,)) for i in range(8)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            assert not errs, errs
            return np.stack(out), B.BatcherStats(m), B.ShardStats(m)
        finally:
            m.Destroy()
    out, bs, ss = _run_with_env(dict(IE_DYNAMIC_BATCH="8", IE_BATCH_WINDOW_US="200000", IE_SHARD_DEVICES="0,0"), go)
    assert bs["coalesced_requests"] == 8 and bs["device_batches"] < 8
    assert ss[1] >= 1                                  # at least one coalesced batch had >= 2 rows and was cut over both replicas
    assert rel_err(out, ref) < 2e-5


def test_requests_never_wait_for_a_kernel_search(tmp_path):
    """The exhaustive kernel search runs at load (declared shape + "tune_batches") and persists beside the model; a request with a
    batch size nobody tuned for takes the cached choice of the nearest pixel count or the planner's default -- the cache file is not
    touched by requests, and a second process finds it."""
    mb = models.densenet("N", growth=16, blocks=(2, 2), stem=32, image=64, classes=10, seed=11)
    path = models.write_repo(str(tmp_path), "tuned", mb, config_json='{"tune_batches": [4]}')
    cache = os.path.join(path, ".ie_tune.fp32.txt")
    m = B.CreateModel(path, "tuned")
    try:
        assert os.path.exists(cache)
        lines = open(cache).read().splitlines()
        assert lines[0].startswith("# ie-tune-v2 ") and len(lines) > 4
        stamp = (os.stat(cache).st_mtime_ns, open(cache).read())
        for b in (4, 5, 3, 16):
            x = models.synthetic_input((b, 3, 64, 64), stream=f"tune{b}")
            y, _ = infer(m, "", "data_0", x, "fc6_1", [b, 10, 1, 1])
            yo = O.run(O.load_model(models.densenet(b, growth=16, blocks=(2, 2), stem=32, image=64, classes=10, seed=11)), {"data_0": x})["fc6_1"]
            assert rel_err(y, yo) < RTOL
        assert (os.stat(cache).st_mtime_ns, open(cache).read()) == stamp
        assert not [f for f in os.listdir(path) if ".tmp" in f]          # written through a temporary + rename, none left behind
    finally:
        m.Destroy()
    m2 = B.CreateModel(path, "tuned")           # same process, fresh model: choices come from the file, nothing is re-searched
    try:
        assert (os.stat(cache).st_mtime_ns, open(cache).read()) == stamp
    finally:
        m2.Destroy()


def test_fused_step_choices_survive_a_restart(tmp_path):
    """The fused steps' timed choices (fp16: dense-block chain vs its plain launches, stem + pool as one launch or two) have short signatures in the tune
    file; a second model of the same directory must find them there and search nothing (the file is not rewritten)."""
    mb = models.densenet("N", growth=32, blocks=(2, 3), stem=64, image=56, classes=10, seed=13)
    path = models.write_repo(str(tmp_path), "tuned16", mb, config_json='{"tune_batches": [4], "precision": "fp16"}')
    cache = os.path.join(path, ".ie_tune.fp16.txt")
    m = B.CreateModel(path, "tuned16")
    try:
        assert os.path.exists(cache)
        stamp = (os.stat(cache).st_mtime_ns, open(cache).read())
        keys = [ln.split(":")[0].split() for ln in stamp[1].splitlines()[1:]]
        assert any(len(k) == 7 for k in keys) and any(len(k) == 10 for k in keys), sorted({len(k) for k in keys})     # stem + pool, dense block
        plan = _run_with_env(dict(IE_PRECISION="fp16"), lambda: B.DescribeModel(path, 4)["plan"])
        assert plan["steps"][0]["algo"] == "stem_pool" and any(s.get("algo") == "dense_block" for s in plan["steps"])
    finally:
        m.Destroy()
    m2 = B.CreateModel(path, "tuned16")
    try:
        assert (os.stat(cache).st_mtime_ns, open(cache).read()) == stamp
    finally:
        m2.Destroy()


# ---------------------------------------------------------------------------------------------------------------------
# fp8 precision mode (BASELINE.json configs[4]: ResNet-50 fp8).  The reference never computes in fp8: parity unpinned.
# Checkers: (1) the OFP8 E4M3 format restated in numpy (oracle/fp8.py) against the device conversion, code for code;
# (2) the plan interpreter quantising exactly where the engine does, with the engine's own calibrated scales - the fp8 kernels
# must agree with it up to fp32 accumulation order.  On shallow graphs that is 3e-4 ... 6e-4 (bound F8_EMU_RTOL); on deep ones a
# handful of e4m3 codes per forward flip between fp32 and float64 accumulation (a value within 1e-6 of a rounding boundary) and
# every flip (a 6 % step in one element) cascades through the layers behind it, so there the emulation is only held to F8_RTOL;
# (3) the plain float64 oracle: the difference is the e4m3 quantisation error of a 50-layer network; this repo's statement of
# the tolerance is F8_RTOL of max|ref| (3 mantissa bits per stored tensor, ~3 % RMS per element, averaged by the dot products).
# ---------------------------------------------------------------------------------------------------------------------
F8_RTOL = 0.08        # ResNet-50 (BASELINE configs[4]): observed 6.1e-2 ... 6.2e-2 per image
F8_RTOL_MINI = 0.12   # the 16-channel-wide mini ResNets: dot products of 16 ... 144 terms and 2x2 final maps average far fewer e4m3 rounding errors (observed 9.2e-2)
F8_EMU_RTOL = 5e-3


def test_e4m3_device_conversion_matches_ofp8():
    from oracle import fp8 as F
    rs = np.random.RandomState(8)
    x = np.concatenate([rs.randn(20000).astype(np.float32) * s for s in (1e-3, 0.05, 1.0, 30.0, 300.0)] +
                       [F.e4m3_decode(np.arange(256, dtype=np.uint8))[np.r_[0:127, 128:255]].astype(np.float32),
                        np.array([17, 19, 21, 1.5 * 2 ** -9, 2.5 * 2 ** -9, 2 ** -10, 2 ** -10 * 1.0001, 447.9, 449, 464, 1e6, -1e6, 0.0], np.float32)])
    for scale in (1.0, 0.0173, 3.5):
        codes, dec = B.E4m3RoundTrip(x, scale)
        q = (x * (np.float32(1.0) / np.float32(scale))).astype(np.float32) if False else (x / np.float32(scale)).astype(np.float32)
        want = F.e4m3_encode(q)
        # +0 / -0 are distinct codes with equal value: compare values, and codes wherever the value is non-zero
        np.testing.assert_array_equal(F.e4m3_decode(codes), F.e4m3_decode(want))
        nz = F.e4m3_decode(want) != 0
        np.testing.assert_array_equal(codes[nz], want[nz])
        np.testing.assert_allclose(dec, F.e4m3_decode(want) * np.float32(scale), rtol=1e-6, atol=0)


def _fp8_run(path, name, x, iname, oname, oshape, env=None):
    def go():
        m = B.CreateModel(path, name)
        try:
            assert B.Precision(m) == "fp8"
            info = B.RuntimeInfo(m)
            y = infer(m, "", iname, x, oname, oshape)[0].copy()
            y2 = infer(m, "", iname, x, oname, oshape)[0]
            np.testing.assert_array_equal(y, y2)
            return y, info
        finally:
            m.Destroy()
    e = dict(IE_PRECISION="fp8")
    e.update(env or {})
    return _run_with_env(e, go)


@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 4, 5, 6])
def test_fp8_resnet_mini_every_tile(tmp_path, tile):
    """A 2-stage bottleneck ResNet (stem 7x7/s2, max pool, 1x1 / 3x3 / strided 3x3 convs, projection and identity shortcuts, global
    pool, Gemm) in fp8 mode, every tile shape of conv_igemm_f8_kernel forced in turn (-1: the planner's own choice + autotune).
    Shallow on purpose: the fp8 emulation is then a tight checker (see the header of this section)."""
    from oracle import fp8 as F
    mb = models.resnet(3, layers=(2, 1), width=16, image=64, classes=20, seed=51)
    path = models.write_repo(str(tmp_path), "resnet_f8", mb)
    x = models.synthetic_input((3, 3, 64, 64), stream="resnet_f8")
    ref = O.run(O.load_model(mb), {"data": x}, dtype=np.float64)["logits"]
    env = {} if tile < 0 else dict(IE_FORCE_TILE=str(tile))
    y, info = _fp8_run(path, "resnet_f8", x, "data", "logits", [3, 20], env)
    assert info["f8_ready"] and all(s > 0 for s in info["f8_act_scales"])

    def plan_and_blob():
        return B.DescribeModel(path, 3)["plan"], B.PlanWeights(path, 3)
    plan, blob = _run_with_env(dict(IE_PRECISION="fp8", **env), plan_and_blob)
    emu = F.run_plan(plan, blob, {"data": x}, act_scales=info["f8_act_scales"], fp8=True)["logits"]
    e_emu, e_ref, emu_ref = rel_err(y, emu), rel_err(y, ref), rel_err(emu, ref)
    print(f"fp8 resnet mini tile {tile}: vs fp8 emulation {e_emu:.2e}, vs float64 oracle {e_ref:.2e} (emulation vs float64 {emu_ref:.2e})")
    assert e_emu < F8_EMU_RTOL and e_ref < F8_RTOL


@pytest.mark.parametrize("image", [64, 50])
def test_fp8_stem_and_max_pool_one_launch_or_two(tmp_path, image):
    """fp8 mode: the stem conv + max pool as one launch (only the pooled tensor is quantised, conv_stem_kernel<POOL>) and as two (IE_NO_STEM_POOL=1:
    the stem's e4m3 output is pooled by pool_f8_kernel) -- each against the emulation of ITS plan (oracle/fp8.py knows both), both inside the fp8
    bound of the float64 oracle; even and odd conv sizes."""
    from oracle import fp8 as F
    mb = models.resnet(3, layers=(2, 1), width=16, image=image, classes=20, seed=59)
    path = models.write_repo(str(tmp_path), "resnet_f8sp", mb)
    x = models.synthetic_input((3, 3, image, image), stream="resnet_f8sp")
    ref = O.run(O.load_model(mb), {"data": x}, dtype=np.float64)["logits"]
    errs = []
    for env in ({}, dict(IE_NO_STEM_POOL="1")):
        y, info = _fp8_run(path, "resnet_f8sp", x, "data", "logits", [3, 20], dict(IE_AUTOTUNE="0", **env))
        plan, blob = _run_with_env(dict(IE_PRECISION="fp8", **env), lambda: (B.DescribeModel(path, 3)["plan"], B.PlanWeights(path, 3)))
        assert (plan["steps"][0]["algo"] == "stem") == bool(env) and (plan["steps"][0]["algo"] == "stem_pool") == (not env)
        emu = F.run_plan(plan, blob, {"data": x}, act_scales=info["f8_act_scales"], fp8=True)["logits"]
        errs.append((rel_err(y, emu), rel_err(y, ref)))
        assert errs[-1][0] < F8_EMU_RTOL and errs[-1][1] < F8_RTOL_MINI
    print(f"fp8 stem + pool image {image}: one launch vs emulation {errs[0][0]:.2e} / float64 {errs[0][1]:.2e}; two launches {errs[1][0]:.2e} / {errs[1][1]:.2e}")


@pytest.mark.parametrize("tile", [100, 101, 102, 103, 104, 105, 107, 109, 200, 201, 202, 203, 204, 206])
def test_fp8_weights_stationary_kernels(tmp_path, tile):
    """conv1x1_ws_f8_kernel (tiles 100-104; 105-109: the same shapes on a grid of one workgroup per CU; also its DUAL form for the projection shortcut of the first block and its STRIDED-input form for the
    stride-2 projection shortcut of the second stage) and conv3x3_ws_f8_kernel (tiles
    200-203; 204-207: one workgroup per CU) forced on a bottleneck ResNet whose channel counts are multiples of 32 (every 1x1 / 3x3 stride-1 conv qualifies): against the fp8
    plan emulation (same quantisation points: kernel correctness), the float64 oracle (stated fp8 bound) and the tiled fp8 kernel's answer."""
    from oracle import fp8 as F
    mb = models.resnet(3, layers=(2, 2), width=32, image=64, classes=20, seed=53)
    path = models.write_repo(str(tmp_path), "resnet_f8w", mb)
    x = models.synthetic_input((3, 3, 64, 64), stream="resnet_f8w")
    ref = O.run(O.load_model(mb), {"data": x}, dtype=np.float64)["logits"]
    env = dict(IE_FORCE_TILE=str(tile))

    def kernels():
        m = B.CreateModel(path, "resnet_f8w")
        try:
            din, _ = B.Prepare(m, [[3, 3, 64, 64]], 1)
            B.CopyToDevice(m, din[0], x)
            B.RunPrepared(m, 1, True)
            return [p_["kernel"] for p_ in B.Profile(m, 1)]
        finally:
            m.Destroy()
    y, info = _fp8_run(path, "resnet_f8w", x, "data", "logits", [3, 20], env)
    y0, _ = _fp8_run(path, "resnet_f8w", x, "data", "logits", [3, 20], dict(IE_FORCE_TILE="3"))
    kern = _run_with_env(dict(IE_PRECISION="fp8", **env), kernels)
    want = "conv1x1_ws_f8_kernel" if tile < 200 else "conv3x3_ws_f8_kernel"
    nws = sum(k.startswith(want) for k in kern)
    assert nws >= (4 if tile < 200 else 2), kern         # (the strided 3x3 of a stage's first block stays on the tiled kernel)
    assert any(k.startswith("conv1x1_ws_f8_kernel<dual") for k in kern), kern          # the first block's projection shortcut: one launch, two GEMMs
    if tile < 200:       # every 1x1 runs weights-stationary, the STRIDED projection shortcut of stage 2 included: only the four 3x3 convs stay tiled
        assert sum(k.startswith("conv_igemm_f8_kernel") for k in kern) == 4, kern
    plan, blob = _run_with_env(dict(IE_PRECISION="fp8", **env), lambda: (B.DescribeModel(path, 3)["plan"], B.PlanWeights(path, 3)))
    emu = F.run_plan(plan, blob, {"data": x}, act_scales=info["f8_act_scales"], fp8=True)["logits"]
    e_emu, e_ref, e_t = rel_err(y, emu), rel_err(y, ref), rel_err(y, y0)
    print(f"fp8 weights-stationary tile {tile}: {nws} launches; vs fp8 emulation {e_emu:.2e}, vs float64 oracle {e_ref:.2e}, vs the tiled kernels {e_t:.2e}")
    assert e_emu < F8_EMU_RTOL and e_ref < F8_RTOL_MINI and e_t < F8_EMU_RTOL


def test_fp8_deep_resnet_mini_vs_float64_oracle(tmp_path):
    """Four stages, 18 convs, final maps of 2x2: quantisation error against float64 within the stated bound."""
    from oracle import fp8 as F
    mb = models.resnet(3, layers=(2, 1, 2, 1), width=16, image=64, classes=20, seed=51)
    path = models.write_repo(str(tmp_path), "resnet_f8d", mb)
    x = models.synthetic_input((3, 3, 64, 64), stream="resnet_f8")
    ref = O.run(O.load_model(mb), {"data": x}, dtype=np.float64)["logits"]
    y, info = _fp8_run(path, "resnet_f8d", x, "data", "logits", [3, 20])
    plan, blob = _run_with_env(dict(IE_PRECISION="fp8"), lambda: (B.DescribeModel(path, 3)["plan"], B.PlanWeights(path, 3)))
    emu = F.run_plan(plan, blob, {"data": x}, act_scales=info["f8_act_scales"], fp8=True)["logits"]
    print(f"fp8 deep resnet mini: vs float64 oracle {rel_err(y, ref):.2e}, vs fp8 emulation {rel_err(y, emu):.2e}, emulation vs float64 {rel_err(emu, ref):.2e}")
    assert rel_err(y, ref) < F8_RTOL_MINI and rel_err(y, emu) < F8_RTOL_MINI


def test_fp8_resnet50_b2_vs_float64_oracle_and_emulation(tmp_path):
    """The full ResNet-50 graph at 224x224 in fp8 mode, batch 2: against the float64 oracle (quantisation error, stated bound) and
    against the float64 plan interpreter that quantises exactly where the engine does (kernel correctness)."""
    from oracle import fp8 as F
    mb = models.resnet50("N")
    path = models.write_repo(str(tmp_path), "resnet50", mb)
    x = models.synthetic_input((2, 3, 224, 224), stream="resnet50")
    ref = O.run(O.load_model(mb), {"data": x}, dtype=np.float64)["logits"]
    y, info = _fp8_run(path, "resnet50", x, "data", "logits", [2, 1000])
    plan, blob = _run_with_env(dict(IE_PRECISION="fp8"), lambda: (B.DescribeModel(path, 2)["plan"], B.PlanWeights(path, 2)))
    emu = F.run_plan(plan, blob, {"data": x}, act_scales=info["f8_act_scales"], fp8=True)["logits"]
    e_emu, e_ref = rel_err(y, emu), rel_err(y, ref)
    top5 = [set(np.argsort(r)[-5:]) for r in ref]
    hit = [int(np.argmax(y[i])) in top5[i] for i in range(2)]
    print(f"resnet50 fp8 B=2: vs fp8 emulation {e_emu:.2e}, vs float64 oracle {e_ref:.2e}; top-1 in the oracle's top-5: {hit}; "
          f"top-1 equal: {np.argmax(y, 1).tolist() == np.argmax(ref, 1).tolist()}")
    assert e_emu < F8_RTOL and e_ref < F8_RTOL and all(hit)


def test_fp8_resnet50_top1_agreement_over_16_images(tmp_path):
    """Top-1 of the fp8 engine against the float64 oracle on 16 ResNet-50 images (random-init weights: the winning margin of an image is
    a few per cent of max|logit|, so this is a sharper check than the norm bound), plus the per-image error bound."""
    mb = models.resnet50("N")
    path = models.write_repo(str(tmp_path), "resnet50", mb)
    x = models.synthetic_input((16, 3, 224, 224), stream="resnet50_top1")
    ref = O.run(O.load_model(mb), {"data": x}, dtype=np.float64)["logits"]
    y, _ = _fp8_run(path, "resnet50", x, "data", "logits", [16, 1000])
    errs = [rel_err(y[i], ref[i]) for i in range(16)]
    agree = int((np.argmax(y, 1) == np.argmax(ref, 1)).sum())
    top5 = sum(int(np.argmax(y[i])) in set(np.argsort(ref[i])[-5:]) for i in range(16))
    print(f"resnet50 fp8 B=16: top-1 agreement {agree}/16, engine top-1 inside the oracle's top-5 {top5}/16, per-image rel err max {max(errs):.2e} median {np.median(errs):.2e}")
    assert max(errs) < F8_RTOL
    assert agree >= 15 and top5 == 16           # >= 90 % top-1 agreement


def test_config4_fp8_resnet50_b256(tmp_path):
    """BASELINE configs[4] at its own size: ResNet-50, fp8 mode, batch 256 through ModelInfer (154 MB FLOAT32 payload) with the
    SEARCHED kernel set ("tune_batches": [1, 256] -- what bench.py times).  Size-independent properties: each image's logits equal
    that image run alone, batch permutation equivariance; plus the float64 oracle on 2 of the 256 images."""
    mb = models.resnet50("N")
    cfg = ('{"name":"resnet50","platform":"onnxruntime_onnx","version":"1","precision":"fp8","tune_batches":[1,256],'
           '"inputs":[{"name":"data","dims":[3,224,224],"shape":[1,3,224,224],"data_type":"FLOAT32"}],'
           '"outputs":[{"name":"logits","dims":[1000],"shape":[1,1000],"data_type":"FLOAT32"}]}')
    models.write_repo(str(tmp_path), "resnet50", mb, config_json=cfg)
    mgr = B.NewInferenceManager(str(tmp_path))
    try:
        mgr.LoadModel("resnet50")
        x = models.synthetic_input((256, 3, 224, 224), stream="resnet50_b256")
        y, dims = infer(mgr, "resnet50", "data", x, "logits", [256, 1000])
        assert dims == [256, 1000] and np.isfinite(y).all()
        for i in (0, 131, 255):
            y1, _ = infer(mgr, "resnet50", "data", x[i:i + 1], "logits", [1, 1000])
            e = rel_err(y1.reshape(1000), y[i])
            assert e < F8_EMU_RTOL * 4, (i, e)       # same quantisation points, other kernels / summation order: a few flipped e4m3 codes
        perm = np.random.RandomState(4).permutation(256)
        yp, _ = infer(mgr, "resnet50", "data", x[perm], "logits", [256, 1000])
        np.testing.assert_array_equal(yp, y[perm])   # same plan, same kernels: images are independent, bit for bit
        ref = O.run(O.load_model(mb), {"data": x[[7, 200]]}, dtype=np.float64)["logits"]
        e = rel_err(y[[7, 200]], ref)
        print(f"resnet50 fp8 B=256: rel err vs float64 oracle on images 7, 200: {e:.2e}")
        assert e < F8_RTOL
    finally:
        mgr.Shutdown()


def test_fp8_inputs_beyond_the_calibrated_range(tmp_path):
    """Inputs 4x larger than anything the load-time calibration saw: activations exceed the 2x headroom, the +-448 clamp of the
    re-quantisation saturates some of them.  The engine must degrade, not break: finite logits, no NaN code (0x7f / 0xff) leaking out
    of a clamp, and an error against the float64 oracle of the SAME scaled input that stays bounded."""
    mb = models.resnet(3, layers=(2, 1, 2, 1), width=16, image=64, classes=20, seed=51)
    path = models.write_repo(str(tmp_path), "resnet_f8d", mb)
    x = models.synthetic_input((3, 3, 64, 64), stream="resnet_f8")
    om = O.load_model(mb)
    out = {}
    for k in (1.0, 4.0, 64.0):
        ref = O.run(om, {"data": x * np.float32(k)}, dtype=np.float64)["logits"]
        y, _ = _fp8_run(path, "resnet_f8d", x * np.float32(k), "data", "logits", [3, 20])
        assert np.isfinite(y).all(), k
        out[k] = rel_err(y, ref)
    print(f"fp8 beyond the calibrated range: rel err at 1x {out[1.0]:.2e}, 4x {out[4.0]:.2e}, 64x (saturating) {out[64.0]:.2e}")
    assert out[1.0] < F8_RTOL_MINI and out[4.0] < 0.5      # 64x: everything clamps; the answer is wrong but finite (asserted above)


def test_fp8_rejects_graphs_it_cannot_run(densenet_repo):
    def go():
        with pytest.raises(RuntimeError, match="fp8 precision"):
            B.CreateModel(os.path.join(densenet_repo, "densenet_onnx", "1"), "densenet_onnx")
    _run_with_env(dict(IE_PRECISION="fp8"), go)


@pytest.mark.parametrize("fuse_tile", ["", "3", "4", "5"])
@pytest.mark.parametrize("batch,image,blocks", [(2, 64, (3, 4)), (5, 56, (2, 3, 2)), (32, 28, (4,)), (8, 112, (3, 2)), (3, 104, (2, 2))])
def test_fused_dense_layer_kernel(tmp_path, batch, image, blocks, fuse_tile):
    """conv_dense_fused_kernel (3x3 growth conv of layer L + 1x1 bottleneck conv of layer L+1 in one launch) on DenseNet-shaped
    graphs with growth 32 / bottleneck 128: both pixel-tile sizes, image widths 16 / 14 / 8 / 7 / 4, ragged last tiles, against the
    float64 oracle; and bit-for-bit determinism plus agreement with the unfused plan (IE_NO_DENSE_FUSE=1)."""
    mb = models.densenet(batch, growth=32, blocks=blocks, stem=64, image=image, classes=12, seed=31)
    path = models.write_repo(str(tmp_path), "fused", mb)
    x = models.synthetic_input((batch, 3, image, image), stream="fused")
    ref = O.run(O.load_model(mb), {"data_0": x}, dtype=np.float64)["fc6_1"]

    def go():
        plan = B.DescribeModel(path, batch)["plan"]
        m = B.CreateModel(path, "fused")
        try:
            y = infer(m, "", "data_0", x, "fc6_1", [batch, 12, 1, 1])[0].copy()
            y2 = infer(m, "", "data_0", x, "fc6_1", [batch, 12, 1, 1])[0]
            prof = B.Profile(m, 1) if False else None
        finally:
            m.Destroy()
        np.testing.assert_array_equal(y, y2)
        return plan, y
    env = dict(IE_AUTOTUNE="0")
    if fuse_tile:
        env["IE_FUSE_PB"] = fuse_tile          # 3 = two-workgroups-per-CU variant of the 16-pixel tile; 4, 5 = wave-specialised variants
    plan, y = _run_with_env(env, go)
    nf = [s for s in plan["steps"] if s.get("algo") == "dense_fused"]
    assert len(nf) >= sum(b - 1 for b in blocks) - 2, (len(nf), blocks)
    plan0, y0 = _run_with_env(dict(IE_AUTOTUNE="0", IE_NO_DENSE_FUSE="1"), go)
    assert not [s for s in plan0["steps"] if s.get("algo") == "dense_fused"]
    e, e0 = rel_err(y, ref), rel_err(y0, ref)
    print(f"fused dense layers B={batch} image={image}: {len(nf)} fused steps (tiles {sorted({s['tile'] for s in nf})}), rel err {e:.2e} (unfused {e0:.2e})")
    assert e < RTOL and e0 < RTOL and rel_err(y, y0) < 2e-5


def test_fp32_split_densenet121_full_size(densenet_repo):
    """BASELINE configs[1] (DenseNet-121, batch 32) with IE_FP32_SPLIT=1: the search may give the 128-channel 1x1 convs to the bf16x6
    kernel.  Same contract as the native fp32 path: the float64 fixture at B=2 and the float64 oracle on two images of the B=32 batch
    within RTOL, an error no worse than the native kernels' (x3 slack for summation order), and batch independence."""
    path = os.path.join(densenet_repo, "densenet_onnx", "1")
    g = np.load(os.path.join(GOLD, "densenet121_b2.npz"))
    x2 = models.synthetic_input((2, 3, 224, 224))
    x32 = models.synthetic_input((32, 3, 224, 224), stream="b32")

    def run():
        m = B.CreateModel(path, "densenet_onnx")
        try:
            y2 = infer(m, "", "data_0", x2, "fc6_1", [2, 1000, 1, 1])[0].reshape(2, 1000).copy()
            y32 = infer(m, "", "data_0", x32, "fc6_1", [32, 1000, 1, 1])[0].reshape(32, 1000).copy()
            y1 = infer(m, "", "data_0", x32[31:32], "fc6_1", [1, 1000, 1, 1])[0].reshape(1000).copy()
            din, _ = B.Prepare(m, [[32, 3, 224, 224]], 1)
            B.CopyToDevice(m, din[0], x32)
            B.RunPrepared(m, 2, True)
            kernels = [p_["kernel"] for p_ in B.Profile(m, 2)]
            return y2, y32, y1, kernels
        finally:
            m.Destroy()
    y2, y32, y1, kern = _run_with_env(dict(IE_FP32_SPLIT="1", IE_TUNE_CACHE="0", IE_TUNE_BATCHES="2,32"), run)      # the search runs at load, never in a request
    n2, n32, _, kern0 = _run_with_env(dict(IE_TUNE_CACHE="0", IE_TUNE_BATCHES="2,32"), run)
    nx = sum(k.startswith("conv1x1_x6_kernel") for k in kern)
    assert nx >= 12 and not any(k.startswith("conv1x1_x6_kernel") for k in kern0), (nx, kern)
    e, e0 = rel_err(y2, g["logits_f64"]), rel_err(n2, g["logits_f64"])
    om = O.load_model(models.densenet121(2))
    yo = O.run(om, {"data_0": x32[[3, 27]]})["fc6_1"].reshape(2, 1000)
    f, f0 = rel_err(y32[[3, 27]], yo), rel_err(n32[[3, 27]], yo)
    print(f"fp32 split: {nx} bf16x6 launches per forward; B=2 vs float64 fixture {e:.2e} (native {e0:.2e}); B=32 images vs oracle {f:.2e} (native {f0:.2e})")
    assert e < RTOL and f < RTOL and e < 3 * e0 + 1e-7 and f < 3 * f0 + 1e-7
    assert np.argmax(y2, 1).tolist() == np.argmax(g["logits_f64"], 1).tolist()
    assert rel_err(y1, y32[31]) < 2e-5 and rel_err(y32, n32) < 2e-5


@pytest.mark.parametrize("tile", [0, 1])
@pytest.mark.parametrize("batch,image,blocks", [(3, 64, (2, 2)), (2, 112, (3,)), (5, 48, (2, 1))])
def test_bf16x6_1x1_kernel(tmp_path, tile, batch, image, blocks):
    """conv1x1_x6_kernel (fp32 products from exactly split bf16 operands, six bf16 MFMAs per block; opt-in IE_FP32_SPLIT=1) forced onto
    every 128-output-channel 1x1 conv of DenseNet-shaped graphs (K = 64 .. 160: even and odd chunk counts, BN+ReLU prologues, ragged
    pixel tiles), both workgroup tiles, against the float64 oracle at the SAME bound as the native fp32 kernels, and against them."""
    mb = models.densenet(batch, growth=32, blocks=blocks, stem=64, image=image, classes=12, seed=43)
    path = models.write_repo(str(tmp_path), "x6", mb)
    x = models.synthetic_input((batch, 3, image, image), stream="x6")
    ref = O.run(O.load_model(mb), {"data_0": x}, dtype=np.float64)["fc6_1"]

    def go():
        plan = B.DescribeModel(path, batch)["plan"]
        m = B.CreateModel(path, "x6")
        try:
            y = infer(m, "", "data_0", x, "fc6_1", [batch, 12, 1, 1])[0].copy()
            y2 = infer(m, "", "data_0", x, "fc6_1", [batch, 12, 1, 1])[0]
        finally:
            m.Destroy()
        np.testing.assert_array_equal(y, y2)
        return plan, y
    plan, y = _run_with_env(dict(IE_FP32_SPLIT="1", IE_FORCE_ALGO="x6", IE_FORCE_TILE=str(tile)), go)
    nx = [s for s in plan["steps"] if s.get("algo") == "conv1x1_x6"]
    assert len(nx) >= sum(blocks) and all(s["out"]["c"] % 128 == 0 and s["k"] == [1, 1] for s in nx)
    _, y0 = _run_with_env(dict(IE_AUTOTUNE="0"), go)
    e, e0 = rel_err(y, ref), rel_err(y0, ref)
    print(f"bf16x6 tile {tile} B={batch} image={image}: {len(nx)} convs, rel err {e:.2e} (native fp32 kernels {e0:.2e})")
    assert e < RTOL and e < 3 * e0 + 1e-7 and rel_err(y, y0) < 2e-5


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
@pytest.mark.parametrize("batch,image,blocks", [(3, 64, (2, 2)), (2, 112, (3,)), (5, 48, (2, 1))])
def test_winograd_3x3_kernel(tmp_path, tile, batch, image, blocks):
    """conv3x3_wino_kernel (Winograd F(2x2, 3x3): 16 multiplies per 2x2 output tile instead of 36) forced onto every eligible growth
    conv of DenseNet-shaped graphs (32 output channels, even image sizes 16 / 28 / 12 / 8 / 6 with ragged tile blocks), all four
    workgroup tile shapes, against the float64 oracle and against the planner's own choice of kernels."""
    mb = models.densenet(batch, growth=32, blocks=blocks, stem=64, image=image, classes=12, seed=41)
    path = models.write_repo(str(tmp_path), "wino", mb)
    x = models.synthetic_input((batch, 3, image, image), stream="wino")
    ref = O.run(O.load_model(mb), {"data_0": x}, dtype=np.float64)["fc6_1"]

    def go():
        plan = B.DescribeModel(path, batch)["plan"]
        m = B.CreateModel(path, "wino")
        try:
            y = infer(m, "", "data_0", x, "fc6_1", [batch, 12, 1, 1])[0].copy()
            y2 = infer(m, "", "data_0", x, "fc6_1", [batch, 12, 1, 1])[0]
        finally:
            m.Destroy()
        np.testing.assert_array_equal(y, y2)
        return plan, y
    env = dict(IE_FORCE_ALGO="wino", IE_FORCE_TILE=str(tile))
    if tile >= 8:                                   # tiles 8-11: the Winograd-domain products as bf16x6 (opt-in mirror)
        env["IE_FP32_SPLIT"] = "1"
    plan, y = _run_with_env(env, go)
    nw = [s for s in plan["steps"] if s.get("algo") == "wino3x3"]
    assert len(nw) == sum(blocks) and all(s["out"]["c"] == 32 for s in nw)
    _, y0 = _run_with_env(dict(IE_AUTOTUNE="0"), go)
    e = rel_err(y, ref)
    print(f"winograd tile {tile} B={batch} image={image}: {len(nw)} convs, rel err {e:.2e} (default kernels {rel_err(y0, ref):.2e})")
    assert e < RTOL and rel_err(y, y0) < 2e-5
