"""B3, the part that needs /root/reference (CPU, skipped where it does not exist — e.g. on the GPU box): the reference's scripts are
parsed at test time and

* their ORIGINAL import blocks (everything in front of the first function: inference_lora.py:1-35, inference_instantid.py:1-40,
  including the try / except blocks around the optional detectors) execute under ``omg_amd.compat.install()`` and bind the names
  of the hot path to this package's classes;
* the driver functions the GPU tests execute (tests/test_compat_gpu.py, tests/test_compat_instantid_gpu.py: ``sample_image``,
  ``build_model_sd``, ``prepare_text``) have the SAME syntax tree as the reference's (docstrings aside) — "transcribed verbatim" is checked, not claimed.
"""
import ast
import os
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "inference_lora.py")), reason="/root/reference is not available here")
HERE = os.path.dirname(os.path.abspath(__file__))


def _tree(path):
    with open(path) as f:
        return ast.parse(f.read(), filename=path)


def _no_docstring(fn):
    if fn.body and isinstance(fn.body[0], ast.Expr) and isinstance(getattr(fn.body[0], "value", None), ast.Constant) and isinstance(fn.body[0].value.value, str):
        fn.body = fn.body[1:]
    return fn


def _functions(path):
    """top-level functions, docstrings dropped (a transcription need not carry the reference's prose)"""
    return {n.name: _no_docstring(n) for n in _tree(path).body if isinstance(n, ast.FunctionDef)}


def _import_block(path):
    body = []
    for n in _tree(path).body:
        if isinstance(n, ast.FunctionDef):
            break
        body.append(n)
    assert all(isinstance(n, (ast.Import, ast.ImportFrom, ast.Try)) for n in body), [type(n).__name__ for n in body]
    return ast.Module(body=body, type_ignores=[])


@pytest.mark.parametrize("script,ours,names", [
    ("inference_lora.py", "test_compat_gpu.py", ("sample_image", "build_model_sd")),
    ("inference_instantid.py", "test_compat_instantid_gpu.py", ("sample_image", "build_model_sd", "prepare_text")),
])
def test_the_driver_functions_the_gpu_tests_run_are_the_references_own(script, ours, names):
    ref, mine = _functions(os.path.join(REF, script)), _functions(os.path.join(HERE, ours))
    for n in names:
        assert ast.dump(ref[n]) == ast.dump(mine[n]), f"{ours}:{n} differs from {script}:{n}"


@pytest.mark.parametrize("script,hot", [
    ("inference_lora.py", ("LoraMultiConceptPipeline", "AttentionReplace", "ControlNetModel", "StableDiffusionXLPipeline",
                           "revise_regionally_controlnet_forward", "save_image")),
    ("inference_instantid.py", ("InstantidMultiConceptPipeline", "InstantidSingleConceptPipeline", "AttentionReplace", "ControlNetModel",
                                "StableDiffusionXLPipeline", "revise_regionally_controlnet_forward", "FaceAnalysis", "cv2")),
])
def test_the_original_import_block_executes_under_compat_install(script, hot, capsys):
    from omg_amd import compat
    saved_path = list(sys.path)
    try:
        compat.install()
        sys.path.insert(0, REF)                # the scripts run from their checkout: `src` is the reference's own (namespace) package
        ns = {}
        exec(compile(_import_block(os.path.join(REF, script)), script, "exec"), ns)
        for n in hot:
            assert n in ns, n
        mod = lambda o: getattr(o, "__module__", "") or ""
        for n in hot:
            if n in ("save_image", "FaceAnalysis", "cv2"):
                continue                        # third-party, outside the hot path: the real package where installed, a stand-in otherwise
            assert mod(ns[n]).startswith("omg_amd."), (n, mod(ns[n]))
        # the hot-path names resolve to the classes the GPU tests exercise
        assert ns["AttentionReplace"] is __import__("omg_amd.controller", fromlist=["x"]).AttentionReplace
        assert ns["ControlNetModel"] is compat.ControlNetModel
    finally:
        compat.uninstall()
        sys.path[:] = saved_path
    assert "diffusers" not in sys.modules or not vars(sys.modules["diffusers"]).get("__omg_amd_alias__", False)

