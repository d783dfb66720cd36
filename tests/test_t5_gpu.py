"""Parity of the T5 prompt encoder on the HIP library (sam_audio_amd/csrc/t5.hip, t5_kernels.hip; SURVEY.md section 8
rows a3 / f4) against `transformers.T5EncoderModel` - the reference's own dependency
(reference sam_audio/model/text_encoder.py:11-37), run in fp32 on the CPU - and against oracle/t5_oracle.py (pinned to
the same module, tests/test_t5_oracle_cpu.py).  Everything goes through the C ABI (`samaudio_t5_*`).

Tolerances: fp32 mode = exact-fp32 GEMMs + fp32 streaming kernels -> summation-order noise only.  bf16 mode = bf16 GEMM
operands (weights and activations rounded once per GEMM; q / k / v rounded to bf16), fp32 accumulation / residual stream
/ T5LayerNorm / softmax: bounds are 2x the errors measured on MI355X, printed by the tests.
"""
import pytest
import torch
import transformers

from oracle import t5_oracle as T
from sam_audio_amd import hip
from sam_audio_amd.config import T5EncoderConfig
from sam_audio_amd.t5_encoder import T5Dims, T5EncoderHIP
from sam_audio_amd.text_encoder import T5TextEncoder

pytestmark = pytest.mark.gpu

T5_BASE = dict(vocab_size=32128, d_model=768, d_kv=64, d_ff=3072, num_layers=12, num_heads=12, feed_forward_proj="relu")
SMALL = dict(vocab_size=100, d_model=64, d_kv=32, d_ff=128, num_layers=3, num_heads=2, feed_forward_proj="relu")


def _model(seed, **kw):
    cfg = transformers.T5Config(**kw)
    torch.manual_seed(seed)
    m = transformers.T5EncoderModel(cfg).eval()
    with torch.no_grad():   # T5's default init leaves the relative bias / norms near-trivial: make every term count
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            elif "relative_attention_bias" in n:
                p.copy_(torch.randn_like(p))
    return m, cfg


def _inputs(cfg, B, L, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, cfg.vocab_size, (B, L), generator=g)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    return ids, mask


def _run(kw, B, L, precision, gpu, seed=3):
    m, cfg = _model(seed, **kw)
    ids, mask = _inputs(cfg, B, L, seed + 1)
    with torch.inference_mode():
        want = m(input_ids=ids, attention_mask=mask)["last_hidden_state"]
        oracle = T.t5_encoder(m.state_dict(), cfg, ids, mask)
    enc = T5EncoderHIP(T5Dims.from_hf(cfg), precision=precision, device=str(gpu))
    enc.load_state_dict(m.state_dict())
    got = enc(ids.to(gpu), mask.to(gpu)).cpu()
    scale = want.abs().max().item()
    return (got - want).abs().max().item() / scale, (got - oracle).abs().max().item() / scale, got, want


@pytest.mark.parametrize("kw,B,L", [(SMALL, 4, 40), (dict(SMALL, d_kv=16, num_heads=4, feed_forward_proj="gelu_new"), 2, 150),
                                    (dict(SMALL, d_kv=128, num_heads=1), 3, 1)])
def test_encoder_fp32_matches_transformers(gpu, kw, B, L):
    """ragged masks (padding rows are returned too, as transformers does), one-token prompts, more than 64 keys per row
    (several key chunks per lane), 16 / 32 / 128-wide heads, ReLU and gelu_new"""
    e_hf, e_or, _, _ = _run(kw, B, L, "fp32", gpu)
    print(f"t5 fp32 {kw.get('feed_forward_proj')} B={B} L={L}: vs transformers rel {e_hf:.2e}, vs oracle rel {e_or:.2e}")
    assert e_hf < 2e-5 and e_or < 2e-5


@pytest.mark.parametrize("precision,bound", [("bf16", 1.4e-2), ("fp16", 1.5e-3)])   # measured 6.7e-3 / 7.1e-4
def test_small_encoder_16bit_operands(gpu, precision, bound):
    if precision == "fp16" and gpu.type != "cuda":
        pytest.skip("the simulator build carries the bf16-operand library only")
    e_hf, _, _, _ = _run(SMALL, 4, 40, precision, gpu)
    print(f"t5 small {precision}: vs transformers rel {e_hf:.2e}")
    assert e_hf < bound


def test_t5_base_dims_fp32_and_bf16(gpu):
    """the reference's encoder (t5-base: 12 layers, 768 wide, 12 heads of 64, ReLU feed-forward of 3072), random init:
    32 prompts x 8 tokens as in bench.py --t5"""
    if gpu.type != "cuda":
        pytest.skip("t5-base dims: too slow on the simulator (the small configurations cover the kernels there)")
    e_hf, e_or, _, _ = _run(T5_BASE, 32, 8, "fp32", gpu)
    print(f"t5-base fp32: vs transformers rel {e_hf:.2e}, vs oracle rel {e_or:.2e}")
    assert e_hf < 2e-5 and e_or < 2e-5
    e_hf, _, _, _ = _run(T5_BASE, 32, 8, "bf16", gpu)
    print(f"t5-base bf16: vs transformers rel {e_hf:.2e}")
    assert e_hf < 1.3e-2     # measured 6.35e-3 on MI355X (profiles/r2_call19/gpu_tests_subset.log)
    e_hf, _, _, _ = _run(T5_BASE, 32, 8, "fp16", gpu)
    print(f"t5-base fp16: vs transformers rel {e_hf:.2e}")
    assert e_hf < 1.6e-3     # measured 7.9e-4


def test_all_padding_row_and_argument_errors(gpu):
    """A row with no valid token attends uniformly (transformers: finfo.min + score rounds to finfo.min for every key);
    ids outside the table raise IndexError like nn.Embedding; longer sequences than max_len, a missing weight, a gated
    feed-forward and an unknown key are errors."""
    m, cfg = _model(7, **SMALL)
    ids, mask = _inputs(cfg, 3, 6, 8)
    mask[1] = 0
    with torch.inference_mode():
        want = m(input_ids=ids, attention_mask=mask)["last_hidden_state"]
    dims = T5Dims.from_hf(cfg, max_len=16)
    enc = T5EncoderHIP(dims, precision="fp32", device=str(gpu))
    with pytest.raises(hip.SamAudioHipError, match="no weights"):
        enc(ids.to(gpu), mask.to(gpu))
    sd = m.state_dict()
    with pytest.raises(RuntimeError, match="Missing keys"):
        enc.load_state_dict({k: v for k, v in sd.items() if "block.1.layer.0.SelfAttention.o" not in k})
    with pytest.raises(RuntimeError, match="unexpected_keys"):
        enc.load_state_dict(dict(sd, **{"encoder.bogus": torch.zeros(1)}))
    enc.load_state_dict({"text_encoder.model." + k: v for k, v in sd.items()})
    got = enc(ids.to(gpu), mask.to(gpu)).cpu()
    assert (got - want).abs().max().item() / want.abs().max().item() < 2e-5
    with pytest.raises(IndexError):
        enc(torch.full((1, 2), cfg.vocab_size).to(gpu), torch.ones(1, 2).to(gpu))
    with pytest.raises(ValueError, match="max_len"):
        enc(torch.zeros(1, 17, dtype=torch.long).to(gpu), torch.ones(1, 17).to(gpu))
    assert enc(torch.zeros(0, 4, dtype=torch.long).to(gpu), torch.ones(0, 4).to(gpu)).shape == (0, 4, 64)
    with pytest.raises(NotImplementedError, match="gated"):
        T5EncoderHIP(T5Dims.from_hf(transformers.T5Config(feed_forward_proj="gated-gelu")), device=str(gpu))
    tc = hip.T5Config(precision=hip.F32, vocab=10, d_model=60, d_kv=32, heads=2, d_ff=128, layers=1, max_len=16,
                      act=hip.ACT_RELU, ln_eps=1e-6)
    import ctypes as C
    h = C.c_void_p()
    lib = hip.lib()
    assert lib.samaudio_t5_create(C.byref(tc), C.byref(h)) == 0
    assert lib.samaudio_t5_finalize(h) == hip.ERR_ARG and b"multiples of 64" in lib.samaudio_last_error()
    lib.samaudio_t5_destroy(h)


def test_text_encoder_wrapper_runs_the_hip_stack(gpu):
    """`T5TextEncoder(..., device=gpu)` (the class SAMAudio.separate() calls, reference text_encoder.py:19-37) routes
    through the HIP stack and returns what the transformers module returns for the same tokens, with the bool mask."""
    m, cfg = _model(9, **SMALL)

    class Tok:
        def __call__(self, texts, truncation=True, max_length=512, padding="longest", return_tensors="pt"):
            rows = [[2 + (sum(map(ord, w)) % 90) for w in t.split()][: max_length - 1] + [1] for t in texts]
            width = max(len(r) for r in rows)
            ids = torch.zeros(len(rows), width, dtype=torch.long)
            att = torch.zeros(len(rows), width, dtype=torch.long)
            for i, r in enumerate(rows):
                ids[i, : len(r)] = torch.tensor(r)
                att[i, : len(r)] = 1
            return {"input_ids": ids, "attention_mask": att}

    enc = T5TextEncoder(T5EncoderConfig(name="unused", dim=64), model=m, tokenizer=Tok(), device=gpu)
    assert enc.backend == "hip" and enc._hip is not None
    texts = ["a dog barking loudly", "rain", "two words"]
    feats, mask = enc(texts)
    tok = Tok()(texts)
    with torch.inference_mode():
        want = m(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"])["last_hidden_state"]
    assert mask.dtype == torch.bool and mask.tolist() == tok["attention_mask"].bool().tolist()
    assert feats.shape == want.shape and feats.dtype == torch.float32
    assert (feats.cpu() - want).abs().max().item() / want.abs().max().item() < 2e-5
    cpu_enc = T5TextEncoder(T5EncoderConfig(name="unused", dim=64), model=m, tokenizer=Tok())
    with pytest.raises(hip.SamAudioHipError, match="no CPU fallback"):
        cpu_enc(texts)
