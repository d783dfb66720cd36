#!/usr/bin/env python
"""The PE-Core-L14-336 tower alone: three encodes of one 250-frame video (random-init weights, fp16 operands), for
`rocprofv3 --kernel-trace --stats`.  usage: python tools/vit_probe.py [frames] [precision]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import preset_config  # noqa: E402
from sam_audio_amd.config import PE_VISION_CONFIGS  # noqa: E402
from sam_audio_amd.synthetic import init_vision_state_dict  # noqa: E402
from sam_audio_amd.vision_encoder import PerceptionEncoder  # noqa: E402
from sam_audio_amd.vision_tower import tower_flops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16"
dev = torch.device("cuda:0")
cfg = preset_config("large*")
pe = PE_VISION_CONFIGS[cfg.vision_encoder.name]
enc = PerceptionEncoder(cfg.vision_encoder, device=dev, precision=prec)
enc.load_state_dict({"model.visual." + k: v for k, v in init_vision_state_dict(pe, seed=5, device=dev).items()})
video = torch.rand(n, 3, 360, 640, device=dev)
frames = enc.transform(video)
enc.encode(frames)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    enc.encode(frames)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
fl = tower_flops(pe, n)
print(f"{cfg.vision_encoder.name} {prec}: {ms:.2f} ms per {n} frames, {fl / ms / 1e9:.1f} TFLOP/s algorithmic ({fl / ms / 1e9 / 2500:.3f} of the 16-bit MFMA peak)")
