"""What a packed-weight cache on disk could save (SURVEY §5 "checkpoint" row, VERDICT r5 missing 5 / next 9): the time the derived weight images take to
BUILD at process start — conv [Cout][ky][kx][Cin], fused q|k|v / k|v rows, interleaved GEGLU rows, the merged LoRA slots W + s B A of two rank-64 concepts —
against the time it would take merely to READ the same bytes back from a file.   python tools/pack_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd.synthetic import c2_inputs, make_concept_models
from omg_amd.unet import UNet2DConditionModel, UNetConfig

dev = torch.device("cuda:0")


def sync_t():
    torch.cuda.synchronize()
    return time.perf_counter()


t0 = sync_t()
unet = UNet2DConditionModel(UNetConfig.sdxl(), dtype=torch.float16, device=dev).init_synthetic_(seed=0)
t1 = sync_t()
req = c2_inputs(unet, seed=0)
x = torch.randn(2, 4, 128, 128, device=dev)
kw = dict(encoder_hidden_states=torch.cat([req["negative_prompt_embeds"][:1], req["prompt_embeds"][:1]]),
          added_cond_kwargs={"text_embeds": torch.cat([req["negative_pooled_prompt_embeds"][:1], req["pooled_prompt_embeds"][:1]]),
                             "time_ids": torch.tensor([[1024.0, 1024.0, 0, 0, 1024.0, 1024.0]] * 2, device=dev)})
unet(x, 981, **kw)
t2 = sync_t()
unet(x, 981, **kw)
t3 = sync_t()
concept = make_concept_models(unet, n_concepts=2, rank=64)
concept.bank.build([(("concept0", 1.0),), (("concept1", 1.0),)], scale=[0.8, 0.8], mode="merged")
t4 = sync_t()
packed = 0
for m in unet.modules():
    for v in getattr(m, "_packed", {}).values() if isinstance(getattr(m, "_packed", None), dict) else []:
        if torch.is_tensor(v):
            packed += v.numel() * v.element_size()
    ws = getattr(m, "w_slots", None)
    if torch.is_tensor(ws):
        packed += ws.numel() * ws.element_size()
print(f"random init of the 2.57 B-parameter UNet on the device: {t1 - t0:.2f} s")
print(f"first forward (B = 2, 1024^2) INCLUDING the lazy build of every packed weight image: {t2 - t1:.2f} s;  second forward: {t3 - t2:.3f} s  ->  packing + first-use costs {t2 - t1 - (t3 - t2):.2f} s")
print(f"merged LoRA slots of two rank-64 concepts (W + 0.8 B A for every attention / FF Linear): {t4 - t3:.2f} s")
print(f"derived images + slots resident: {packed / 1e9:.2f} GB  ->  reading them back from a file at 3 GB/s (NVMe) + H2D would take {packed / 3e9:.1f} s, "
      f"i.e. {'MORE' if packed / 3e9 > (t2 - t1 - (t3 - t2)) + (t4 - t3) else 'less'} than rebuilding them ({(t2 - t1 - (t3 - t2)) + (t4 - t3):.2f} s)")
