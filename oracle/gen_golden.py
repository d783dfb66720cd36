"""Mint tests/golden/*.npz from the REFERENCE's own classes and pin the oracle restatement.

Run in the build container only (needs /root/reference):   python -m oracle.gen_golden

For each case it
  1. builds seeded weights (sam_audio_amd.synthetic.init_state_dict) and seeded inputs,
  2. runs the reference's own SAMAudio.forward / align_inputs / DiT / Patcher / AlignModalities /
     EmbedAnchors code (imported read-only via oracle/ref_import.py; the SAMAudio object is assembled
     without its __init__, which would need dacvae / HF downloads) in eval() + inference_mode,
  3. asserts oracle.samaudio_oracle reproduces every tapped tensor to fp32 round-off,
  4. stores the REFERENCE outputs (not the oracle's) as the fixture.
Integer fixture: Batch.process_anchors (processor.py:78-124) on a table of (start, end) spans that
includes exact multiples of the 0.04 s frame period.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle import samaudio_oracle as O  # noqa: E402
from sam_audio_amd.config import preset_config  # noqa: E402
from sam_audio_amd.synthetic import init_state_dict  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

CASES = {
    # name: (size preset, B, T, Lt, sizes, seed)
    "tiny_b2_t48": ("tiny", 2, 48, 5, [48, 37], 11),
    "mini_b1_t250": ("mini", 1, 250, 8, [250], 12),
}

ANCHOR_TABLE = [
    [("+", 0.0, 0.04), ("-", 0.04, 0.08)],
    [("+", 0.039999, 0.120001), ("+", 1.0, 1.5)],
    [("-", 0.5, 0.5), ("+", 1.9, 1.92)],
    [],
]


def case_inputs(name):
    """Seeded inputs of a golden case - shared with tests (they regenerate, the fixture stores
    outputs only)."""
    size, B, T, Lt, sizes, seed = CASES[name]
    g = torch.Generator().manual_seed(seed)
    cfg = preset_config(size)
    noisy = torch.randn(B, T, 256, generator=g)
    z = torch.randn(B, T, 128, generator=g)
    feats = torch.cat([z, z], dim=2)
    text = torch.randn(B, Lt, cfg.text_encoder.dim, generator=g)
    text_mask = torch.ones(B, Lt, dtype=torch.bool)
    for b in range(B):
        text_mask[b, Lt - b:] = False if b else True
    video = torch.randn(B, cfg.vision_encoder.dim, T, generator=g) * 0.5
    time = torch.rand(B, generator=g)
    sizes_t = torch.tensor(sizes)
    pad_mask = torch.arange(T)[None] < sizes_t[:, None]
    anchors = [[("+", 0.2, 0.6), ("-", 1.0, 1.2)], [("+", 0.1, 0.3)]][:B]
    ids, align = O.anchors_to_ids(anchors, pad_mask, 1920, 48000)
    return dict(cfg=cfg, noisy=noisy, feats=feats, text=text, text_mask=text_mask, video=video,
                time=time, pad_mask=pad_mask, anchor_ids=ids, anchor_alignment=align, seed=seed)


def build_reference_model(cfg, sd):
    ref_import.import_reference()
    from sam_audio.model.align import AlignModalities
    from sam_audio.model.config import TransformerConfig as RefTC
    from sam_audio.model.model import EmbedAnchors, SAMAudio, SinusoidalEmbedding
    from sam_audio.model.transformer import DiT

    t = cfg.transformer
    ref_tc = RefTC(dim=t.dim, n_heads=t.n_heads, n_layers=t.n_layers, context_dim=t.context_dim)
    m = SAMAudio.__new__(SAMAudio)
    torch.nn.Module.__init__(m)
    m.transformer = DiT(ref_tc)
    m.proj = torch.nn.Linear(cfg.in_channels, t.dim)
    m.align_masked_video = AlignModalities(cfg.vision_encoder.dim, t.dim)
    m.embed_anchors = EmbedAnchors(cfg.num_anchors, cfg.anchor_embedding_dim, t.dim)
    m.memory_proj = torch.nn.Linear(cfg.text_encoder.dim, t.dim)
    m.timestep_emb = SinusoidalEmbedding(t.dim)
    own = {k: v for k, v in sd.items() if not k.startswith("audio_codec.")}
    missing, unexpected = torch.nn.Module.load_state_dict(m, own, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    return m.eval()


def run_case(name):
    inp = case_inputs(name)
    cfg = inp["cfg"]
    sd = init_state_dict(cfg, seed=inp["seed"], with_codec=False)
    ref = build_reference_model(cfg, sd)
    taps_ref = {}
    hooks = [
        ref.embed_anchors.register_forward_hook(lambda m, i, o: taps_ref.__setitem__("aligned", o)),
        ref.transformer.x_embedder.register_forward_hook(
            lambda m, i, o: taps_ref.__setitem__("patcher", o.transpose(1, 2))),
    ]
    for i, layer in enumerate(ref.transformer.layers):
        hooks.append(layer.register_forward_hook(
            lambda m, a, o, i=i: taps_ref.__setitem__(f"layer{i}", o)))
    with torch.inference_mode():
        out_ref = ref.forward(
            noisy_audio=inp["noisy"], audio_features=inp["feats"], text_features=inp["text"],
            time=inp["time"], masked_video_features=inp["video"], text_mask=inp["text_mask"],
            anchor_ids=inp["anchor_ids"], anchor_alignment=inp["anchor_alignment"],
            audio_pad_mask=inp["pad_mask"])
    for h in hooks:
        h.remove()
    taps_ref["out"] = out_ref

    taps_or = {}
    with torch.inference_mode():
        out_or = O.samaudio_forward(sd, cfg, inp["noisy"], inp["feats"], inp["text"], inp["time"],
                                    video=inp["video"], text_mask=inp["text_mask"],
                                    anchor_ids=inp["anchor_ids"], anchor_alignment=inp["anchor_alignment"],
                                    pad_mask=inp["pad_mask"], taps=taps_or)
    taps_or["out"] = out_or
    report = {}
    for k, v in taps_ref.items():
        err = (v - taps_or[k]).abs().max().item()
        scale = v.abs().max().item()
        report[k] = (err, scale)
        assert err <= 2e-5 * max(1.0, scale), f"{name}: oracle != reference at {k}: {err} (scale {scale})"
    wanted = ("aligned", "patcher", "layer0", f"layer{cfg.transformer.n_layers - 1}", "out")
    if inp["noisy"].shape[1] > 100:  # keep the long-sequence fixture small
        wanted = ("patcher", "out")
    keep = {k: v.numpy() for k, v in taps_ref.items() if k in wanted}
    np.savez(os.path.join(GOLDEN_DIR, f"forward_{name}.npz"), **keep)
    return report


def run_anchor_table():
    ref_import.import_reference()
    from sam_audio.processor import Batch

    B, T = len(ANCHOR_TABLE), 60
    sizes = torch.tensor([60, 60, 50, 41])
    pad_mask = torch.arange(T)[None] < sizes[:, None]
    batch = Batch(audios=torch.zeros(B, 1, T * 1920), sizes=sizes, wav_sizes=sizes * 1920,
                  descriptions=[""] * B, hop_length=1920, audio_sampling_rate=48000,
                  anchors=ANCHOR_TABLE, audio_pad_mask=pad_mask)
    ids, align = O.anchors_to_ids(ANCHOR_TABLE, pad_mask, 1920, 48000)
    assert torch.equal(ids, batch.anchor_ids) and torch.equal(align, batch.anchor_alignment)
    none_batch = Batch(audios=torch.zeros(B, 1, T * 1920), sizes=sizes, wav_sizes=sizes * 1920,
                       descriptions=[""] * B, hop_length=1920, audio_sampling_rate=48000,
                       anchors=None, audio_pad_mask=pad_mask)
    ids0, align0 = O.anchors_to_ids(None, pad_mask, 1920, 48000)
    assert torch.equal(ids0, none_batch.anchor_ids) and torch.equal(align0, none_batch.anchor_alignment)
    np.savez(os.path.join(GOLDEN_DIR, "anchors.npz"), sizes=sizes.numpy(),
             ids=batch.anchor_ids.numpy(), alignment=batch.anchor_alignment.numpy(),
             ids_none=none_batch.anchor_ids.numpy(), alignment_none=none_batch.anchor_alignment.numpy())


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    for name in CASES:
        rep = run_case(name)
        print(name, {k: f"{e:.2e}/{s:.2f}" for k, (e, s) in rep.items()})
    run_anchor_table()
    print("anchor table OK")


if __name__ == "__main__":
    main()
