"""The oracle restatement against the golden vectors minted from the REFERENCE's own classes
(oracle/gen_golden.py), and the host-side integer work against the reference Batch fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import samaudio_oracle as O
from oracle.gen_golden import ANCHOR_TABLE, CASES, case_inputs
from sam_audio_amd.processor import Batch
from sam_audio_amd.synthetic import init_state_dict

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_forward_matches_reference_fixture(name):
    inp = case_inputs(name)
    sd = init_state_dict(inp["cfg"], seed=inp["seed"], with_codec=False)
    taps = {}
    with torch.inference_mode():
        out = O.samaudio_forward(sd, inp["cfg"], inp["noisy"], inp["feats"], inp["text"], inp["time"],
                                 video=inp["video"], text_mask=inp["text_mask"], anchor_ids=inp["anchor_ids"],
                                 anchor_alignment=inp["anchor_alignment"], pad_mask=inp["pad_mask"], taps=taps)
    taps["out"] = out
    gold = np.load(os.path.join(GOLDEN, f"forward_{name}.npz"))
    for key in gold.files:
        err = (taps[key] - torch.from_numpy(gold[key])).abs().max().item()
        assert err < 2e-5, (key, err)


def _anchor_batch(anchors):
    sizes = torch.tensor([60, 60, 50, 41])
    pad = torch.arange(60)[None] < sizes[:, None]
    return Batch(audios=torch.zeros(4, 1, 60 * 1920), sizes=sizes, wav_sizes=sizes * 1920, descriptions=[""] * 4,
                 hop_length=1920, audio_sampling_rate=48000, anchors=anchors, audio_pad_mask=pad), pad


def test_anchor_indices_bit_exact_vs_reference_fixture():
    """north_star: 'bit-exact frame indices for span prediction' - integer parity, exact equality."""
    gold = np.load(os.path.join(GOLDEN, "anchors.npz"))
    batch, pad = _anchor_batch(ANCHOR_TABLE)
    assert np.array_equal(batch.anchor_ids.numpy(), gold["ids"])
    assert np.array_equal(batch.anchor_alignment.numpy(), gold["alignment"])
    ids, align = O.anchors_to_ids(ANCHOR_TABLE, pad, 1920, 48000)
    assert np.array_equal(ids.numpy(), gold["ids"]) and np.array_equal(align.numpy(), gold["alignment"])
    none_batch, _ = _anchor_batch(None)
    assert np.array_equal(none_batch.anchor_ids.numpy(), gold["ids_none"])
    assert np.array_equal(none_batch.anchor_alignment.numpy(), gold["alignment_none"])


def test_ode_grid_and_midpoint_order():
    """Fixed-grid midpoint on dy/dt = y has the closed form (1 + h + h^2/2)^n; euler (1 + h)^n."""
    y0 = torch.ones(1, 2, 256)
    for method, factor in (("midpoint", lambda h: 1 + h + h * h / 2), ("euler", lambda h: 1 + h)):
        rec = []
        y = O.ode_fixed_grid(lambda t, y: y, y0, method=method, step_size=1 / 16, record=rec)
        assert len(rec) == 16
        assert torch.allclose(y, y0 * factor(1 / 16) ** 16, rtol=1e-5)
    with pytest.raises(ValueError):
        O.ode_fixed_grid(lambda t, y: y, y0, method="dopri5")


def test_codec_shapes_and_reflect_pad():
    from sam_audio_amd.config import preset_config
    cfg = preset_config("tiny")
    sd = init_state_dict(cfg, seed=1)
    wav = torch.randn(1, 1, 1920 + 7)
    with torch.inference_mode():
        z = O.dac_encode(sd, cfg.audio_codec, wav)
        w = O.dac_decode(sd, cfg.audio_codec, z)
    assert z.shape == (1, 128, 2) and w.shape == (1, 1, 2 * 1920)
    assert float(w.abs().max()) <= 1.0
