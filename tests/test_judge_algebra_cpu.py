"""The HIP Judge / PE-AV / PE-A-Frame paths evaluate the reference algebra in a re-arranged form (fused q|k|v,
16-row interleaved SwiGLU operand, tap-major conv weights, split concatenations, the mixture branch once per clip, the
bias-free head commuted with the masked mean).  This file replays exactly that decomposition in torch ON THE CONVERTED
ENGINE TENSORS (sam_audio_amd.judge.convert_judge / convert_frame) and checks it against oracle/judge_oracle.py, so
that a wrong split / permutation / broadcast shows up on the CPU, before a GPU is involved.  It mirrors
sam_audio_amd/csrc/peav.hip step by step (same buffer names)."""
import torch
import torch.nn.functional as F

from oracle import gen_golden_judge as G
from oracle import judge_oracle as J
from oracle import samaudio_oracle as O
from sam_audio_amd.config import PEAudioFrameConfig
from sam_audio_amd.judge import convert_frame, convert_judge
from sam_audio_amd.synthetic import init_frame_state_dict, init_judge_state_dict


def _swiglu16(x, w13):
    """GEMM epilogue with swiglu=1: rows of w13 alternate 16 gate rows / 16 up rows (gemm.hip / gemm2.hip)."""
    y = x @ w13.T
    n = y.shape[-1]
    y = y.reshape(*y.shape[:-1], n // 32, 2, 16)
    return (F.silu(y[..., 0, :]) * y[..., 1, :]).reshape(*y.shape[:-3], n // 2)


def _rmsnorm(x, w, eps):
    return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps) * w


def _encoder(t, P, x_act, pad_mask, n_layers, n_heads, eps):
    """PeavEncoder::forward on engine tensors `t` with prefix P.  x_act [rows, T, in_dim]."""
    rows, T, _ = x_act.shape
    D = t[P + "in.w"].shape[0]
    S = T + 1
    h0 = torch.empty(rows, S, D)
    h0[:, 1:] = x_act @ t[P + "in.w"].T + t[P + "in.b"]
    h0[:, 0] = t[P + "cls"]
    mask_s = torch.ones(rows, S, dtype=torch.bool) if pad_mask is None else torch.cat([pad_mask[:, :1], pad_mask], 1)

    def mgn_silu(x, w, b):  # launch_masked_groupnorm_silu into a halo-1 buffer
        m = mask_s[..., None].float()
        n = (m.sum((1, 2), keepdim=True) * D).clamp_min(1)
        mean = (x * m).sum((1, 2), keepdim=True) / n
        var = ((x * x) * m).sum((1, 2), keepdim=True) / n - mean * mean
        out = F.silu((x - mean) * torch.rsqrt(var.clamp_min(0) + 1e-5) * w + b) * m
        return F.pad(out, (0, 0, 1, 1))

    def conv3(buf, w, b):  # 3-tap implicit GEMM: row s of the output reads rows s, s+1, s+2 of the halo buffer
        a = torch.cat([buf[:, 0:S], buf[:, 1:S + 1], buf[:, 2:S + 2]], dim=2)
        return a @ w.T + b

    r1 = conv3(mgn_silu(h0, t[P + "gn1.w"], t[P + "gn1.b"]), t[P + "conv1.w"], t[P + "conv1.b"])
    h = conv3(mgn_silu(r1, t[P + "gn2.w"], t[P + "gn2.b"]), t[P + "conv2.w"], t[P + "conv2.b"]) + h0
    cos, sin = t[P + "rope_cos"][:S], t[P + "rope_sin"][:S]
    for l in range(n_layers):
        L = f"{P}L{l}."
        xn = _rmsnorm(h, t[L + "attn_norm"], eps)
        qkv = xn @ t[L + "wqkv"].T
        q, k, v = [z.reshape(rows, S, n_heads, 128).transpose(1, 2) for z in qkv.split(D, dim=2)]

        def prep(z, w):  # qkv_prep: per-head RMSNorm then RoPE on adjacent pairs, table [pos, 64]
            z = _rmsnorm(z, w, eps)
            z0, z1 = z[..., 0::2], z[..., 1::2]
            return torch.stack([z0 * cos - z1 * sin, z0 * sin + z1 * cos], dim=-1).flatten(-2)

        q, k = prep(q, t[L + "q_norm"]), prep(k, t[L + "k_norm"])
        sc = (q @ k.transpose(-1, -2)) * (128 ** -0.5)
        sc = sc.masked_fill(~mask_s[:, None, None, :], float("-inf"))
        attn = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(rows, S, D)
        h = h + attn @ t[L + "wo"].T
        u = _swiglu16(_rmsnorm(h, t[L + "ffn_norm"], eps), t[L + "w13"])
        h = h + u @ t[L + "w2"].T
    return _rmsnorm(h, t[P + "norm"], eps) @ t[P + "out.w"].T, mask_s


def test_judge_score_decomposition_equals_the_oracle():
    cfg = G.tiny_judge_config()
    sd = init_judge_state_dict(cfg, seed=9)
    t = convert_judge(sd, cfg, torch.float32, "cpu")
    cand, Bi = 3, 2
    g = torch.Generator().manual_seed(2)
    hop = cfg.audio_codec.hop_length
    T = 9
    lengths = torch.tensor([T * hop, 5 * hop])
    wpad = torch.arange(T * hop)[None] < lengths[:, None]
    wav_in = 0.3 * torch.randn(Bi, 1, T * hop, generator=g) * wpad[:, None]
    wav_sep = 0.3 * torch.randn(Bi * cand, 1, T * hop, generator=g) * wpad.repeat_interleave(cand, 0)[:, None]
    pooled = torch.randn(Bi * cand, cfg.text_hidden, generator=g)
    with torch.inference_mode():
        want = J.judge_forward(sd, cfg, pooled, wav_in.repeat_interleave(cand, 0), wav_sep,
                               wpad.repeat_interleave(cand, 0))
        # --- Judge::score -------------------------------------------------------------------------------------
        in_lat = O.dac_encode(sd, cfg.audio_codec, wav_in).transpose(1, 2)
        sep_lat = O.dac_encode(sd, cfg.audio_codec, wav_sep).transpose(1, 2)
        pad = wpad[:, ::hop]
        Bp = Bi * cand
        xa = torch.cat([in_lat, sep_lat], 0)
        mask1 = torch.cat([pad, pad.repeat_interleave(cand, 0)], 0)
        tc, fc = cfg.transformer, cfg.finetune_transformer
        hid, _ = _encoder(t, "t.", xa, mask1, tc.num_hidden_layers, tc.num_attention_heads, tc.rms_norm_eps)
        Bn = cfg.bottleneck_dim
        inp_part = hid[:Bi, 1:] @ t["cat.wi"].T + t["cat.b"]                       # once per clip
        audio = torch.empty(Bp, T, Bn)
        for c in range(cand):                                                      # candidate c of every clip
            rows = torch.arange(Bi) * cand + c
            audio[rows] = hid[Bi + rows, 1:] @ t["cat.wh"].T + inp_part
        t1 = pooled @ t["tp1.w"].T
        t2 = t1 @ t["tp2.w"].T + t["tp2.b"]
        tl = F.layer_norm(t2, (Bn,), t["ln.w"], t["ln.b"], 1e-5)
        tpart = tl @ t["pat.wt"].T + t["pat.b"]
        at = audio @ t["pat.wa"].T + tpart[:, None, :]                             # res_ld = 0 broadcast over frames
        mask2 = mask1[Bi:]
        fout, mask_s = _encoder(t, "ft.", at, mask2, fc.num_hidden_layers, fc.num_attention_heads, fc.rms_norm_eps)
        m = mask_s[:, 1:, None].float()                                            # judge_pool_head_kernel
        pooled_h = (fout[:, 1:] * m).sum(1) / m.sum(1).clamp_min(1)
        got = (pooled_h @ t["head.w"].T) * t["std"] + t["mean"]
    assert (got - want).abs().max() < 2e-5


def test_frame_logits_decomposition_equals_the_oracle():
    cfg = PEAudioFrameConfig(audio=G.TINY_TC, text_model=dict(G.TINY_TEXT, hidden_size=64), codebook_dim=64)
    sd = init_frame_state_dict(cfg, seed=2)
    t = convert_frame(sd, cfg, torch.float32, "cpu")
    g = torch.Generator().manual_seed(6)
    feats, pooled = torch.randn(3, 17, 64, generator=g), torch.randn(3, 64, generator=g)
    pad = torch.arange(17)[None] < torch.tensor([17, 9, 4])[:, None]
    with torch.inference_mode():
        want = J.frame_logits(sd, cfg, pooled, feats, pad)
        ac = cfg.audio
        hid, _ = _encoder(t, "a.", feats, pad, ac.num_hidden_layers, ac.num_attention_heads, ac.rms_norm_eps)
        a_emb = F.layer_norm(hid, (hid.shape[-1],), t["ah.ln_w"], t["ah.ln_b"], 1e-6) @ t["ah.w"].T
        t_emb = F.layer_norm(pooled, (64,), t["th.ln_w"], t["th.ln_b"], 1e-6) @ t["th.w"].T
        got = torch.einsum("bte,be->bt", a_emb[:, 1:], t_emb) * t["logit_scale"] + t["logit_bias"]
    assert ((got - want).abs() * pad).max() < 2e-5
