import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _pkg import load_package  # noqa: E402

load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine_lib():
    """Path of the built shared library (built on demand; the GPU box uses the prebuilt file that travelled)."""
    from gpu_ai_inference_server_amd import build
    if not os.path.exists(build.LIB):
        build.build_library()
    return build.LIB


MINI = {
    # name -> (builder kwargs / callable, input name, input shape)
    "mini_densenet": (lambda m: m.densenet(2, growth=8, blocks=(2, 3), stem=16, image=32, classes=10, seed=5), "data_0", (2, 3, 32, 32)),
    "mini_densenet_scale": (lambda m: m.densenet(3, growth=12, blocks=(2, 2, 2), stem=24, image=64, classes=17, seed=6, caffe_scale=True),
                            "data_0", (3, 3, 64, 64)),
    "mini_gemm_mlp": (lambda m: m.gemm_mlp(4), "x", (4, 64, 1, 1)),
    "mini_resnet_block": (lambda m: m.resnet_block(2), "x", (2, 3, 16, 16)),
    # pre-activation residual blocks: Conv -> Add -> BN(gamma != 1) -> [ReLU] -> GAP (the BN must scale the shortcut too)
    "mini_preact": (lambda m: m.preact_block(2), "x", (2, 3, 16, 16)),
    "mini_preact_norelu": (lambda m: m.preact_block(2, final_relu=False), "x", (2, 3, 16, 16)),
    # the two other export forms of a Caffe Scale layer: [C] constants through Unsqueeze nodes, and opset-6 broadcast=1/axis=1
    "mini_densenet_unsqueeze": (lambda m: m.densenet(3, growth=12, blocks=(2, 2, 2), stem=24, image=64, classes=17, seed=6,
                                                     caffe_scale="unsqueeze"), "data_0", (3, 3, 64, 64)),
    "mini_densenet_legacy": (lambda m: m.densenet(3, growth=12, blocks=(2, 2, 2), stem=24, image=64, classes=17, seed=6,
                                                  caffe_scale="legacy_axis"), "data_0", (3, 3, 64, 64)),
}


@pytest.fixture(scope="session")
def model_repo(tmp_path_factory):
    """A model repository laid out like the reference's ./models (<name>/<version>/model.onnx + config.json)."""
    from gpu_ai_inference_server_amd.modelgen import models
    root = str(tmp_path_factory.mktemp("models"))
    golden = os.path.join(ROOT, "tests", "golden")
    # the reference's own committed fixture (data file): models/test_model/1/{model.onnx,config.json}
    d = os.path.join(root, "test_model", "1")
    os.makedirs(d)
    for f in ("model.onnx", "config.json"):
        with open(os.path.join(golden, "test_model", "1", f), "rb") as src, open(os.path.join(d, f), "wb") as dst:
            dst.write(src.read())
    for name, (mk, _, _) in MINI.items():
        models.write_repo(root, name, mk(models))
    models.write_repo(root, "mini_two_input", models.two_input_graph("N"))
    return root


@pytest.fixture(scope="session")
def densenet_repo(tmp_path_factory):
    """Synthetic DenseNet-121 with a symbolic batch axis (32 MB, generated in ~2 s, never committed)."""
    from gpu_ai_inference_server_amd.modelgen import models
    root = str(tmp_path_factory.mktemp("densenet"))
    cfg = ('{"name":"densenet_onnx","platform":"onnxruntime_onnx","version":"1",'
           '"inputs":[{"name":"data_0","dims":[3,224,224],"shape":[1,3,224,224],"data_type":"FLOAT32"}],'
           '"outputs":[{"name":"fc6_1","dims":[1000],"shape":[1,1000,1,1],"data_type":"FLOAT32"}]}')
    models.write_repo(root, "densenet_onnx", models.densenet121("N"), config_json=cfg)
    return root
