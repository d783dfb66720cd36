#!/usr/bin/env python
"""CPU numerics lab (TEST INFRASTRUCTURE, runs without a GPU): which GEMM-operand scheme holds 1e-3 on the hostile weights?

    python tools/numerics_lab.py --size 'small*' --steps 4 --schemes fp16,x3,out8

Runs the CPU oracle's solve (oracle/samaudio_oracle.py) with `F.linear` of chosen weight classes replaced by an emulation of a
16-bit operand scheme (products exact, fp32 accumulation as the MFMA does) and reports max-abs latent error against the plain
fp32 oracle.  Schemes:
  fp16     both operands rounded to IEEE half
  bf16     both operands rounded to bfloat16
  x3       compensated: A = A_hi + A_lo, W = W_hi + W_lo (all fp16); A_hi W_hi + A_lo W_hi + A_hi W_lo
  x2a      only A split (A_hi W_hi + A_lo W_hi)
  x2w      only W split
  outN     the N K-columns of largest |A| column-absmax stay exact fp32 on both operands, the rest fp16
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import samaudio_oracle as O  # noqa: E402
from sam_audio_amd import SAMAudioProcessor, preset_config  # noqa: E402
from sam_audio_amd.synthetic import (init_state_dict, make_hostile, synthetic_clip, synthetic_noise,  # noqa: E402
                                     synthetic_text_features)


def classify(key: str):
    if ".layers." in key:
        if ".attention.w" in key:
            return "wo" if key.endswith("wo.weight") else "qkv"
        if ".cross_attention.w" in key:
            if key.endswith("wq.weight"):
                return "cwq"
            if key.endswith("wo.weight"):
                return "cwo"
            return "ckv"
        if ".feed_forward.w2" in key:
            return "w2"
        if ".feed_forward.w" in key:
            return "w13"
    return None


def h(x):
    return x.half().float()


def make_linear(scheme, wclass, classes, stats):
    real = F.linear

    def lin(x, w, b=None):
        c = wclass.get(id(w))
        if c is None or c not in classes or scheme == "fp32":
            return real(x, w, b)
        if scheme == "fp16":
            y = real(h(x), h(w))
        elif scheme == "bf16":
            y = real(x.bfloat16().float(), w.bfloat16().float())
        elif scheme == "x3":
            xh, wh = h(x), h(w)
            xl, wl = h(x - xh), h(w - wh)
            y = real(xh, wh) + real(xl, wh) + real(xh, wl)
        elif scheme == "x2a":
            xh, wh = h(x), h(w)
            y = real(xh, wh) + real(h(x - xh), wh)
        elif scheme == "x2w":
            xh, wh = h(x), h(w)
            y = real(xh, wh) + real(xh, h(w - wh))
        elif scheme.startswith("out"):
            n = int(scheme[3:])
            col = x.reshape(-1, x.shape[-1]).abs().amax(0)
            idx = col.topk(n).indices
            xq, wq = h(x), h(w)
            xq[..., idx] = x[..., idx]
            wq[:, idx] = w[:, idx]
            y = real(xq, wq)
            stats.setdefault(c, []).append(float(col[idx].min() / col.median()))
        else:
            raise ValueError(scheme)
        return y if b is None else y + b

    return lin


def make_attention(round_qkv, round_heads, round_p, round_out):
    """oracle.attention with the HIP pipeline's 16-bit STORAGE roundings switched on one by one (self-attention only)."""
    real = O.attention

    def att(sd, prefix, x, n_heads, eps, *, cross=None, key_mask=None, rope=None):
        if cross is not None:
            return real(sd, prefix, x, n_heads, eps, cross=cross, key_mask=key_mask, rope=rope)
        import math
        r = (lambda t: h(t))
        q, k, v = (O.F.linear(x, sd[prefix + n + ".weight"]) for n in ("wq", "wk", "wv"))
        if round_qkv:
            q, k, v = r(q), r(k), r(v)
        q, k, v = (O.split_heads_interleaved(t, n_heads) for t in (q, k, v))
        q = O.rms_norm(q, sd[prefix + "q_norm.weight"], eps)
        k = O.rms_norm(k, sd[prefix + "k_norm.weight"], eps)
        if rope is not None:
            q, k = O.apply_rope(q, *rope), O.apply_rope(k, *rope)
        if round_heads:
            q, k = r(q), r(k)
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
        if key_mask is not None:
            scores = scores.masked_fill(~key_mask[:, None, None, :], float("-inf"))
        p = torch.softmax(scores, dim=-1)
        if round_p:
            p = r(p)
        out = torch.matmul(p, v)
        B, H, T, hd = out.shape
        out = out.permute(0, 2, 1, 3).reshape(B, T, H * hd)
        if round_out:
            out = r(out)
        return O.F.linear(out, sd[prefix + "wo.weight"])

    return att


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--attn", default="", help="comma list of attention storage roundings to switch on: qkv,heads,p,out")
    ap.add_argument("--size", default="small*")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--clips", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--schemes", default="fp16,x3,out8")
    ap.add_argument("--classes", default="qkv,wo,cwq,cwo,w13,w2")
    ap.add_argument("--benign", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    cfg = preset_config(args.size)
    sd = init_state_dict(cfg, seed=0, device=torch.device("cpu"))
    if not args.benign:
        sd = make_hostile(sd, cfg, seed=0)
    wclass = {id(v): classify(k) for k, v in sd.items() if classify(k)}
    R = args.clips
    hop = cfg.audio_codec.hop_length
    n = int(args.seconds * cfg.audio_codec.sample_rate) // hop * hop
    clips = [synthetic_clip(i, n) for i in range(R)]
    text, tmask = synthetic_text_features(R, 8, seed=7)
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["sound"] * R, audios=clips, text_features=text, text_mask=tmask)
    noise = synthetic_noise(R, n // hop)

    def solve():
        with torch.inference_mode():
            return O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise, step_size=1.0 / args.steps,
                              decode=False)[2]

    t0 = time.time()
    ref = solve()
    print(f"{args.size} {'benign' if args.benign else 'hostile'} {R} clip(s) x {args.seconds} s, {args.steps} midpoint steps: "
          f"|latent| <= {ref.abs().max():.3f} rms {ref.pow(2).mean().sqrt():.3f}  ({time.time() - t0:.1f} s)", flush=True)
    real = F.linear
    for scheme in args.schemes.split(","):
        for classes in [args.classes.split(",")]:
            stats = {}
            O.F.linear = make_linear(scheme, wclass, set(classes), stats)
            real_att = O.attention
            if args.attn:
                a = set(args.attn.split(","))
                O.attention = make_attention("qkv" in a, "heads" in a, "p" in a, "out" in a)
            try:
                t0 = time.time()
                lat = solve()
            finally:
                O.F.linear = real
                O.attention = real_att
            d = (lat - ref)
            extra = "  ".join(f"{k}: min outlier/median {min(v):.1f}" for k, v in stats.items())
            print(f"{scheme:6s} classes={','.join(classes)}: latent max-abs {d.abs().max():.3e} rms {d.pow(2).mean().sqrt():.3e} "
                  f"({time.time() - t0:.1f} s) {extra}", flush=True)


if __name__ == "__main__":
    main()
