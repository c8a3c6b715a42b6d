"""B3 (CPU half): the diffusers-shaped constructors of omg_amd.compat on a synthetic model directory — what from_pretrained /
load_lora_weights / tokenizer / .to() give the reference's build_model_sd (inference_lora.py:152-171).  Compute needs the GPU."""
import os
import sys

import pytest
import torch

from tests import _fake_hub as hub


@pytest.fixture(scope="module")
def dirs(tmp_path_factory):
    root = tmp_path_factory.mktemp("hub")
    return hub.write_sdxl_dir(str(root / "sdxl")), hub.write_controlnet_dir(str(root / "controlnet")), str(root)


def test_from_pretrained_shapes_the_reference_objects(dirs):
    from omg_amd import compat
    from omg_amd._lib import OmgHipError
    model, cn_dir, root = dirs
    compat.clear_component_cache()
    controlnet = compat.ControlNetModel.from_pretrained(cn_dir, torch_dtype=torch.float16).to("cpu")
    pipe = compat.LoraMultiConceptPipeline.from_pretrained(model, controlnet=controlnet, torch_dtype=torch.float16, variant="fp16").to("cpu")
    concept = compat.StableDiffusionXLPipeline.from_pretrained(model, torch_dtype=torch.float16, variant="fp16").to("cpu")
    concept.enable_xformers_memory_efficient_attention()
    assert concept._unet is pipe.unet and concept.vae is pipe.vae, "one set of weights serves both pipelines"
    assert pipe.controlnet is controlnet and pipe.unet.dtype == torch.float16
    assert type(pipe.scheduler).__name__ == "EulerDiscreteScheduler"            # what the checkpoint ships; the scripts never set one
    # the tokenizer calls of inference_lora.py:276-283
    ids = pipe.tokenizer("a man and a woman walking on the street")["input_ids"]
    assert pipe.tokenizer("man")["input_ids"][1] in ids[1:-1] and pipe.tokenizer("dog")["input_ids"][1] not in ids[1:-1]
    # LoRA files: kohya-ss SDXL keys for the concepts (as the reference's shipped files), PEFT keys for the style
    p1 = hub.write_lora_file(os.path.join(root, "loras", "chris-evans.safetensors"), pipe.unet, 11, text_encoders=[pipe.text_encoder, pipe.text_encoder_2])
    p2 = hub.write_lora_file(os.path.join(root, "style", "pytorch_lora_weights.safetensors"), pipe.unet, 12, style="peft", text_encoders=[pipe.text_encoder])
    name = p1.split("/")[-1].split(".")[0]
    concept.load_lora_weights(p1, weight_name="pytorch_lora_weights.safetensors", adapter_name=name)
    pipe.load_lora_weights(os.path.dirname(p2), weight_name="pytorch_lora_weights.safetensors", adapter_name="style")
    concept.load_lora_weights(os.path.dirname(p2), weight_name="pytorch_lora_weights.safetensors", adapter_name="style")
    bank = concept.bank
    assert set(bank.adapters) == {"chris-evans", "style"} and bank is pipe._comp.bank
    assert set(bank.adapters["chris-evans"].text_encoder) == {1, 2} and set(bank.adapters["style"].text_encoder) == {1}
    assert not bank.adapters["chris-evans"].skipped_keys
    concept.set_adapters([name, "style"], adapter_weights=[0.7, 0.5])
    assert concept._active == ((name, 0.7), ("style", 0.5))
    # the controller is built from pipe.tokenizer exactly as the script does
    from omg_amd.controller import AttentionReplace
    from omg_amd.pipeline import revise_regionally_controlnet_forward
    P = "a man and a woman walking on the street"
    ctl = AttentionReplace([P, P], 50, cross_replace_steps={"default_": 1.0}, self_replace_steps=0.4, tokenizer=pipe.tokenizer, device="cpu",
                           dtype=torch.float16, width=4, height=4)
    revise_regionally_controlnet_forward(pipe.unet, ctl)
    assert ctl.is_pure_replacement and ctl.num_att_layers == 2 * 17
    with pytest.raises(OmgHipError):                               # no CPU fallback: the call itself needs the MI355X
        pipe(prompt=[[P, P], [("a man", "", ), ("a woman", "")]], negative_prompt=["", ""], concept_models=concept, controller=ctl, stage=1,
             lora_list=[name, name], styleL=False, num_inference_steps=2, image=None)


def test_instantid_container_surface(dirs):
    from omg_amd import compat
    model, cn_dir, root = dirs
    compat.clear_component_cache()
    c = compat.StableDiffusionXLInstantIDPipeline.from_pretrained(model, torch_dtype=torch.float16, variant="fp16")
    for m in ("load_ip_adapter_instantid", "set_image_proj_model", "set_ip_adapter", "set_ip_adapter_scale", "_encode_prompt_image_emb", "unet", "set_adapters"):
        assert hasattr(c, m)
    # get_face_embedding keeps the reference's selection rule (first of the ascending sort by (x1 - x0) * y1 - y0)
    from PIL import Image
    img = os.path.join(root, "face.png")
    Image.new("RGB", (8, 8), (10, 20, 30)).save(img)

    class FakeApp:
        def get(self, bgr):
            assert bgr.shape == (8, 8, 3) and tuple(bgr[0, 0]) == (30, 20, 10)      # BGR, as cv2.cvtColor(..., COLOR_RGB2BGR) gives
            return [{"bbox": [0, 0, 4, 4], "embedding": "big"}, {"bbox": [0, 0, 1, 1], "embedding": "small"}]

    assert compat.get_face_embedding(FakeApp(), [img]) == ["small"]


def test_the_module_runner_executes_a_script_under_the_aliases(tmp_path, capsys):
    """``python -m omg_amd.run script.py args`` = install() + the script's directory on sys.path + runpy: a stand-in script with the
    reference's own import lines (inference_lora.py:29-32) and an ``if __name__ == "__main__"`` body sees this package's classes and its
    own arguments."""
    import subprocess
    script = tmp_path / "driver.py"
    script.write_text(
        "import sys\n"
        "from src.pipelines.lora_pipeline import LoraMultiConceptPipeline\n"
        "from src.prompt_attention.p2p_attention import AttentionReplace\n"
        "from diffusers import ControlNetModel, StableDiffusionXLPipeline\n"
        "from src.pipelines.lora_pipeline import revise_regionally_controlnet_forward\n"
        "if __name__ == '__main__':\n"
        "    print(LoraMultiConceptPipeline.__module__, AttentionReplace.__module__, ControlNetModel.__module__, sys.argv[1:])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "omg_amd.run", str(script), "--prompt", "a man"], capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "omg_amd.compat omg_amd.controller omg_amd.compat ['--prompt', 'a man']" in r.stdout


def test_peft_default_active_adapter_bookkeeping(dirs):
    """``peft_active_adapters``: what PEFT leaves switched on in a pipe when nobody calls ``set_adapters`` — the FIRST adapter loaded into
    THAT pipe (inference_instantid.py:220-222: one style LoRA per pipe) — per pipe although the weights live in one shared bank; an
    explicit ``set_adapters`` replaces it."""
    from omg_amd import compat
    model, cn_dir, root = dirs
    compat.clear_component_cache()
    idn = compat.ControlNetModel.from_pretrained(cn_dir, torch_dtype=torch.float16)
    pipe = compat.InstantidMultiConceptPipeline.from_pretrained(model, controlnet=idn, torch_dtype=torch.float16, variant="fp16")
    concept = compat.StableDiffusionXLInstantIDPipeline.from_pretrained(model, controlnet=idn, torch_dtype=torch.float16)
    assert pipe.peft_active_adapters() == [] == concept.peft_active_adapters()
    p = hub.write_lora_file(os.path.join(root, "style2", "pytorch_lora_weights.safetensors"), pipe.unet, 13, style="peft", text_encoders=[pipe.text_encoder])
    pipe.load_lora_weights(os.path.dirname(p), weight_name="pytorch_lora_weights.safetensors", adapter_name="style")
    assert pipe.peft_active_adapters() == [("style", 1.0)] and concept.peft_active_adapters() == [], "per pipe, not per bank"
    concept.load_lora_weights(os.path.dirname(p), weight_name="pytorch_lora_weights.safetensors", adapter_name="style")
    p2 = hub.write_lora_file(os.path.join(root, "loras", "other.safetensors"), pipe.unet, 14)
    concept.load_lora_weights(p2, adapter_name="other")
    assert concept.peft_active_adapters() == [("style", 1.0)], "a later load is injected inactive"
    concept.set_adapters(["other", "style"], adapter_weights=[0.7, 0.5])
    assert concept.peft_active_adapters() == [("other", 0.7), ("style", 0.5)]
    assert set(concept.bank.adapters) == {"style", "other"} and concept.bank is pipe._comp.bank


def test_mx8_layer_class_map_on_the_module_tree(dirs):
    """``set_precision_classes``: exactly the named classes get the MX-fp8 flag; everything SURVEY 7.3 item 8 excludes never does; the
    ControlNet's blocks take the same map; presets are consistent."""
    from omg_amd import compat
    from omg_amd.modules import Conv2d, Linear
    from omg_amd.unet import MX8_CLASSES, MX8_PRESETS, mx8_class_of
    model, cn_dir, _ = dirs
    compat.clear_component_cache()
    cn = compat.ControlNetModel.from_pretrained(cn_dir, torch_dtype=torch.float16)
    unet = compat.LoraMultiConceptPipeline.from_pretrained(model, controlnet=cn, torch_dtype=torch.float16, variant="fp16").unet
    assert set(MX8_PRESETS["all"]) == set(MX8_CLASSES) and set(MX8_PRESETS["safe"]) < set(MX8_CLASSES) and MX8_PRESETS["none"] == ()
    by_cls = {}
    for name, m in unet.named_modules():
        if isinstance(m, (Linear, Conv2d)):
            c = mx8_class_of(name, m)
            by_cls.setdefault(c, []).append(name)
    never = by_cls.get(None, [])
    assert any(n == "conv_in" for n in never) and any(n == "conv_out" for n in never)
    assert all(not (n.endswith("attn2.to_k") or n.endswith("attn2.to_v")) or mx8_class_of(n, unet.get_submodule(n)) is None for n in sum(by_cls.values(), []))
    assert any("downsamplers" in n or "upsamplers" in n for n in never) and any("time_emb" in n for n in never)
    # the tiny test topology has channels that are not multiples of 128: only its eligible Linears / convs can be switched on
    on = unet.set_precision_classes(MX8_CLASSES)
    flagged = {n for n, m in unet.named_modules() if isinstance(m, (Linear, Conv2d)) and getattr(m, "mx8", False)}
    assert on == len(flagged) and flagged <= {n for c, ns in by_cls.items() if c is not None for n in ns}
    on_safe = unet.set_precision_classes(MX8_PRESETS["safe"])
    flagged_safe = {n for n, m in unet.named_modules() if isinstance(m, (Linear, Conv2d)) and getattr(m, "mx8", False)}
    assert on_safe == len(flagged_safe) <= on and all(mx8_class_of(n, unet.get_submodule(n)) in MX8_PRESETS["safe"] for n in flagged_safe)
    assert unet.set_precision_classes("safe") == on_safe and unet.set_precision_classes("all") == on      # preset names
    assert unet.set_precision_classes(()) == 0 and unet.linear_precision == "fp16" and unet.conv_precision == "fp16"
    with pytest.raises(ValueError):
        unet.set_precision_classes(["no_such_class"])
    n_cn = cn.set_precision_classes(MX8_CLASSES)
    assert n_cn == sum(1 for _, m in cn.named_modules() if isinstance(m, (Linear, Conv2d)) and getattr(m, "mx8", False))
    assert cn.set_precision_classes(()) == 0


def test_run_module_resolves_the_checkouts_own_src_packages_from_a_foreign_cwd(tmp_path, monkeypatch):
    """`python -m omg_amd.run /path/OMG/script.py` started OUTSIDE the checkout.  The alias package `src` must not shadow the checkout's
    real `src.*` packages (detectors, segmenters): a driver whose sibling `src/efficientvit_stub/...` is importable from its own directory must
    be importable here too, while `src.pipelines.lora_pipeline` resolves to the alias.  A real `diffusers` in sys.modules survives the round trip."""
    import sys
    import types
    from omg_amd import compat, run
    co = tmp_path / "OMG"
    (co / "src" / "segmenter_stub").mkdir(parents=True)
    (co / "src" / "segmenter_stub" / "__init__.py").write_text("ANSWER = 42\n")
    (co / "driver.py").write_text(
        "import json, sys\n"
        "from src.segmenter_stub import ANSWER\n"
        "from src.pipelines.lora_pipeline import LoraMultiConceptPipeline\n"
        "import diffusers\n"
        "json.dump({'answer': ANSWER, 'pipe': LoraMultiConceptPipeline.__module__, 'alias': getattr(diffusers, '__omg_amd_alias__', False)}, open(sys.argv[1], 'w'))\n")
    out = tmp_path / "out.json"
    elsewhere = tmp_path / "elsewhere"
    elsewhere.mkdir()
    monkeypatch.chdir(elsewhere)
    real = types.ModuleType("diffusers"); real.MARK = "real"
    real_sub = types.ModuleType("diffusers.models"); real_sub.MARK = "real.models"
    monkeypatch.setitem(sys.modules, "diffusers", real)
    monkeypatch.setitem(sys.modules, "diffusers.models", real_sub)
    for n in [n for n in sys.modules if n == "src" or n.startswith("src.")]:
        monkeypatch.delitem(sys.modules, n)
    path0 = list(sys.path)
    try:
        run.main([str(co / "driver.py"), str(out)])
    finally:
        sys.path[:] = path0
    import json
    got = json.load(open(out))
    assert got == {"answer": 42, "pipe": "omg_amd.compat", "alias": True}
    assert sys.modules["diffusers"] is real and sys.modules["diffusers.models"] is real_sub and not hasattr(real, "ControlNetModel")
    assert not any(getattr(m, "__dict__", {}).get("__omg_amd_alias__", False) for m in list(sys.modules.values()) if m is not None)


def test_install_twice_then_uninstall_restores_a_real_diffusers(monkeypatch):
    """the second install() of a session (no uninstall() in between) sees only the alias modules; the real `diffusers` the first call set
    aside must survive it and come back with uninstall()."""
    import sys
    import types
    from omg_amd import compat
    real = types.ModuleType("diffusers"); real.MARK = "real"
    real_sub = types.ModuleType("diffusers.utils"); real_sub.MARK = "real.utils"
    monkeypatch.setitem(sys.modules, "diffusers", real)
    monkeypatch.setitem(sys.modules, "diffusers.utils", real_sub)
    saved_src = {n: m for n, m in sys.modules.items() if n == "src" or n.startswith("src.")}
    for n in saved_src:
        monkeypatch.delitem(sys.modules, n)
    try:
        compat.install()
        assert getattr(sys.modules["diffusers"], "__omg_amd_alias__", False)
        compat.install()
        assert getattr(sys.modules["diffusers"], "__omg_amd_alias__", False)
    finally:
        compat.uninstall()
    assert sys.modules["diffusers"] is real and sys.modules["diffusers.utils"] is real_sub
    assert not any(getattr(m, "__dict__", {}).get("__omg_amd_alias__", False) for m in list(sys.modules.values()) if m is not None)
