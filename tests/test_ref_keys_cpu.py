"""State-dict key bookkeeping against the REFERENCE's own module tree (only where /root/reference exists - this
container; the GPU box skips it) plus the prefixes a genuine checkpoint carries that the separate() engine ignores.

The reference SAMAudio cannot be instantiated offline (dacvae / perception_models are absent), but its DiT, AlignModalities,
EmbedAnchors and the two Linear layers can: their state-dict keys under the attribute names of reference model.py:80-93
must be exactly this build's expected non-codec keys."""
import pytest
import torch

from oracle import ref_import
from sam_audio_amd import preset_config
from sam_audio_amd.weights import expected_keys, split_missing_unexpected


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference is not on this box")
def test_expected_keys_equal_the_reference_module_tree():
    ref_import.import_reference()
    from sam_audio.model.align import AlignModalities
    from sam_audio.model.config import TransformerConfig as RefTC
    from sam_audio.model.model import EmbedAnchors
    from sam_audio.model.transformer import DiT
    cfg = preset_config("tiny")
    t = cfg.transformer
    with torch.device("meta"):
        dit = DiT(RefTC(dim=t.dim, n_heads=t.n_heads, n_layers=t.n_layers, context_dim=t.context_dim))
        align = AlignModalities(cfg.vision_encoder.dim, t.dim)
        anchors = EmbedAnchors(cfg.num_anchors, cfg.anchor_embedding_dim, t.dim)
    ref_keys = {"transformer." + k for k in dit.state_dict()}
    ref_keys |= {"align_masked_video." + k for k in align.state_dict()}
    ref_keys |= {"embed_anchors." + k for k in anchors.state_dict()}
    ref_keys |= {"proj.weight", "proj.bias", "memory_proj.weight", "memory_proj.bias"}   # model.py:84,91
    ours = {k for k in expected_keys(cfg) if not k.startswith("audio_codec.")}
    assert ours == ref_keys, (sorted(ours - ref_keys)[:5], sorted(ref_keys - ours)[:5])


def test_a_genuine_checkpoint_key_set_loads_strictly():
    """reference model.py:82-83,346-359: checkpoint.pt carries `vision_encoder.model.*` (PerceptionEncoder is a
    submodule and not in the skip regex) and none of text_encoder / rankers / span_predictor."""
    cfg = preset_config("tiny")
    keys = list(expected_keys(cfg))
    genuine = keys + ["vision_encoder.model.visual.conv1.weight", "vision_encoder.model.visual.proj",
                      "vision_encoder.model.logit_scale"]
    assert split_missing_unexpected(genuine, cfg) == ([], [])
    assert split_missing_unexpected(genuine + ["vision_encoderX.w"], cfg) == ([], ["vision_encoderX.w"])


def test_judge_strict_load_tolerates_the_whole_quantizer_and_a_decoder():
    """reference codec.py:62-63: DACVAEEncoder keeps `model.quantizer` entirely (in_proj AND out_proj); a state dict saved
    from a full DACVAE also holds `decoder.*`.  Both are present-but-unused for the Judge and must not fail strict."""
    import re
    src = open(__import__("sam_audio_amd.judge", fromlist=["x"]).__file__).read()
    m = re.search(r'unused = re\.compile\(r"(.+?)"\)', src)
    assert m, "judge.load_state_dict lost its tolerated-key rule"
    unused = re.compile(m.group(1))
    assert unused.search("audio_codec.quantizer.out_proj.weight_g")
    assert unused.search("audio_codec.quantizer.out_proj.bias")
    assert unused.search("audio_codec.decoder.model.0.weight_v")
    assert not unused.search("audio_codec.quantizer.in_proj.weight_g")
    assert not unused.search("audio_codec.encoder.block.0.weight")
    assert not unused.search("transformer.layers.0.self_attn.q_proj.weight")
