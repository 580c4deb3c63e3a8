"""Import helper: the package directory is named `gpu-ai-inference-server_amd` (hyphens, as the task's
layout rule asks), which `import` cannot spell, so it is loaded under the alias `gpu_ai_inference_server_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "gpu-ai-inference-server_amd")
ALIAS = "gpu_ai_inference_server_amd"


def load_package():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(ALIAS, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
