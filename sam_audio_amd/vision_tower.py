"""`PEVisionTower` - the PE-Core vision tower behind `PerceptionEncoder.encode` on the HIP library
(reference sam_audio/model/vision_encoder.py:80-89:
`self.model = pe.CLIP.from_config(cfg.name)`; `self.model.encode_image(x, normalize=self.normalize_feature)`).

`core.vision_encoder.pe` is part of the un-vendored perception_models package, so the class is restated from the
published architecture (see oracle/vit_oracle.py for the CPU statement and what it is pinned to).  This module is
host code only: it maps the `visual.*` state_dict keys of `pe.CLIP` onto the engine's tensors (one-time re-layout),
sizes the workspace and calls `samaudio_vit_*`; every arithmetic step of `encode_image` is a HIP kernel
(sam_audio_amd/csrc/vit.hip).  There is no CPU / eager fallback.

reference key (below `visual.`)                         engine tensor
  conv1.weight [W,3,P,P]                                patch.w [W, Kp]  (flattened (c, py, px), K zero-padded to x64)
  class_embedding [W], positional_embedding [S,W]       pos [S, W]       (row 0 = class_embedding + positional_embedding[0])
  ln_pre / ln_post .weight/.bias                        ln_pre.w/.b, ln_post.w/.b
  transformer.resblocks.{i}.ln_1 / ln_2                 L{i}.ln1.w/.b, L{i}.ln2.w/.b
  transformer.resblocks.{i}.attn.in_proj_weight/_bias   L{i}.wqkv [3W, W], L{i}.bqkv
  transformer.resblocks.{i}.attn.out_proj               L{i}.wo, L{i}.bo
  transformer.resblocks.{i}.mlp.c_fc / c_proj           L{i}.w1, L{i}.b1, L{i}.w2, L{i}.b2
  attn_pool.probe, attn_pool.attn.in_proj_*             pool.q [W] (= q-projection of the probe, the same for every frame),
                                                        pool.wkv [2W, W], pool.bkv
  attn_pool.attn.out_proj / layernorm / mlp.*           pool.wo, pool.bo, pool.ln.w/.b, pool.w1, pool.b1, pool.w2, pool.b2
  proj [W, O]                                           proj [O, W]      (transposed: features = pooled @ proj)
  (2-D RoPE has no parameters)                          rope_cos / rope_sin [S, hd/2]
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List, Optional, Tuple

import torch

from . import hip
from .config import PE_VISION_CONFIGS, PEVisionConfig
from .judge import _ensure_ws, _register

POOL_TYPES = {"tok": 0, "avg": 1, "attn": 2}
ACTS = {"gelu": hip.ACT_GELU, "quick_gelu": hip.ACT_QUICK_GELU}


def rope2d_pair_tables(cfg: PEVisionConfig) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos / sin [tokens, head_dim / 2], one entry per rotated pair.  Published Rope2D: a 1-D rotary table of
    `head_dim / 2` channels per axis (theta 10000, frequencies 1 / theta^(2i / (head_dim/2))), x angles on the first
    half of the head and y angles on the second, grid coordinates starting at 1 when a class token occupies (0, 0)."""
    hd = cfg.width // cfg.heads
    per_axis = hd // 4                                  # rotated pairs per axis
    # fp32 throughout, as the published code forms them (fp32 positions x fp32 frequencies)
    inv = 1.0 / (10000.0 ** (torch.arange(per_axis, dtype=torch.float32) * 2.0 / (hd // 2)))
    off = 1 if cfg.use_cls_token else 0
    pos = torch.arange(cfg.grid, dtype=torch.float32) + off
    ang = pos[:, None] * inv[None, :]                   # [G, pairs per axis]
    ax = ang[None, :, :].expand(cfg.grid, cfg.grid, per_axis)   # varies with x (column)
    ay = ang[:, None, :].expand(cfg.grid, cfg.grid, per_axis)   # varies with y (row)
    table = torch.cat([ax, ay], dim=-1).reshape(cfg.grid * cfg.grid, hd // 2)
    if cfg.use_cls_token:
        table = torch.cat([torch.zeros(1, hd // 2), table], dim=0)
    return table.cos().contiguous(), table.sin().contiguous()


def tower_flops(cfg: PEVisionConfig, n_frames: int) -> float:
    """Algorithmic FLOPs (2 x MACs) of `n_frames` tower evaluations at the native resolution: patch GEMM, per block
    q|k|v + attention (QK^T and PV) + out_proj + MLP, pooling head, projection (bench.py's vision_tower roofline)."""
    g = cfg.grid
    S, W, F_ = cfg.tokens, cfg.width, cfg.mlp_width
    per = 2.0 * g * g * 3 * cfg.patch_size ** 2 * W
    per += cfg.layers * (2.0 * S * W * 3 * W + 4.0 * S * S * W + 2.0 * S * W * W + 4.0 * S * W * F_)
    if cfg.pool_type == "attn":
        per += 2.0 * S * W * 2 * W + 4.0 * S * W + 2.0 * W * W + 4.0 * W * F_
    per += 2.0 * W * cfg.output_dim
    return per * n_frames


def expected_keys(cfg: PEVisionConfig) -> List[str]:
    keys = ["conv1.weight", "proj"]
    if cfg.use_cls_token:
        keys.append("class_embedding")
    if cfg.use_abs_posemb:
        keys.append("positional_embedding")
    for name, on in (("ln_pre", cfg.use_ln_pre), ("ln_post", cfg.use_ln_post)):
        if on:
            keys += [f"{name}.weight", f"{name}.bias"]
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        keys += [p + k for k in ("ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias",
                                 "attn.out_proj.weight", "attn.out_proj.bias", "ln_2.weight", "ln_2.bias",
                                 "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")]
    if cfg.pool_type == "attn":
        keys += ["attn_pool." + k for k in ("probe", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                                            "attn.out_proj.bias", "layernorm.weight", "layernorm.bias", "mlp.c_fc.weight",
                                            "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")]
    return keys


def convert_vision(sd: Dict[str, torch.Tensor], cfg: PEVisionConfig, act_dtype: torch.dtype, device) -> Dict[str, torch.Tensor]:
    W, S = cfg.width, cfg.tokens
    f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()   # noqa: E731
    act = lambda t: t.detach().to(device=device, dtype=torch.float32).to(act_dtype).contiguous()  # noqa: E731
    out: Dict[str, torch.Tensor] = {}
    kk = 3 * cfg.patch_size * cfg.patch_size
    kp = (kk + 63) // 64 * 64
    pw = torch.zeros(W, kp, dtype=torch.float32, device=device)
    pw[:, :kk] = f32(sd["conv1.weight"]).reshape(W, kk)
    out["patch.w"] = pw.to(act_dtype)
    pos = f32(sd["positional_embedding"]).clone() if cfg.use_abs_posemb else torch.zeros(S, W, device=device)
    assert pos.shape == (S, W), f"positional_embedding {tuple(pos.shape)} != ({S}, {W}): only the native grid is supported"
    if cfg.use_cls_token:
        pos[0] += f32(sd["class_embedding"])
    out["pos"] = pos.contiguous()
    for name, on in (("ln_pre", cfg.use_ln_pre), ("ln_post", cfg.use_ln_post)):
        if on:
            out[name + ".w"], out[name + ".b"] = f32(sd[name + ".weight"]), f32(sd[name + ".bias"])
    if cfg.use_rope2d:
        cos, sin = rope2d_pair_tables(cfg)
        out["rope_cos"], out["rope_sin"] = cos.to(device), sin.to(device)
    for i in range(cfg.layers):
        s, d = f"transformer.resblocks.{i}.", f"L{i}."
        out[d + "ln1.w"], out[d + "ln1.b"] = f32(sd[s + "ln_1.weight"]), f32(sd[s + "ln_1.bias"])
        out[d + "ln2.w"], out[d + "ln2.b"] = f32(sd[s + "ln_2.weight"]), f32(sd[s + "ln_2.bias"])
        out[d + "wqkv"], out[d + "bqkv"] = act(sd[s + "attn.in_proj_weight"]), f32(sd[s + "attn.in_proj_bias"])
        out[d + "wo"], out[d + "bo"] = act(sd[s + "attn.out_proj.weight"]), f32(sd[s + "attn.out_proj.bias"])
        out[d + "w1"], out[d + "b1"] = act(sd[s + "mlp.c_fc.weight"]), f32(sd[s + "mlp.c_fc.bias"])
        out[d + "w2"], out[d + "b2"] = act(sd[s + "mlp.c_proj.weight"]), f32(sd[s + "mlp.c_proj.bias"])
    if cfg.pool_type == "attn":
        p = "attn_pool."
        w_in, b_in = f32(sd[p + "attn.in_proj_weight"]), f32(sd[p + "attn.in_proj_bias"])
        probe = f32(sd[p + "probe"]).reshape(W)
        # the query is the same for every frame: a load-time constant (fp64 on the host side of the fold)
        out["pool.q"] = (w_in[:W].double() @ probe.double() + b_in[:W].double()).float().contiguous()
        out["pool.wkv"], out["pool.bkv"] = w_in[W:].to(act_dtype).contiguous(), b_in[W:].contiguous()
        out["pool.wo"], out["pool.bo"] = act(sd[p + "attn.out_proj.weight"]), f32(sd[p + "attn.out_proj.bias"])
        out["pool.ln.w"], out["pool.ln.b"] = f32(sd[p + "layernorm.weight"]), f32(sd[p + "layernorm.bias"])
        out["pool.w1"], out["pool.b1"] = act(sd[p + "mlp.c_fc.weight"]), f32(sd[p + "mlp.c_fc.bias"])
        out["pool.w2"], out["pool.b2"] = act(sd[p + "mlp.c_proj.weight"]), f32(sd[p + "mlp.c_proj.bias"])
    out["proj"] = act(sd["proj"]).t().contiguous()
    return out


class PEVisionTower:
    """`pe.CLIP`'s image side: `tower.encode_image(frames [N,3,S,S] float, normalize=bool) -> [N, output_dim]`; also
    callable (`tower(frames, normalize=...)`), which is the signature `PerceptionEncoder(tower=...)` expects."""

    def __init__(self, cfg: Optional[PEVisionConfig] = None, precision: str = "bf16", device: Optional[str] = None,
                 name: str = "PE-Core-L14-336", streams: int = 2):
        hip.check_precision(precision)
        streams = int(os.environ.get("SAMAUDIO_VIT_STREAMS", streams))   # tuning only: A/B of the two-stream encode
        if streams not in (1, 2):
            raise ValueError("streams must be 1 or 2")
        # 2: frame batches of >= 64 frames are encoded as two halves on two HIP streams (a second engine context that
        # borrows the same weights).  The tower's GEMMs are short (K = 1024 = 16 K-tiles per tile) and every CU reaches its
        # epilogue at the same time, so alone the kernel alternates between an MFMA phase and an HBM write burst; two
        # streams interleave the two.  Frames are independent: bitwise equal to one stream (tests/test_vit_gpu.py).
        self.streams = streams
        self._side = None          # (handle, workspace holder, torch stream) of the second context
        if cfg is None:
            if name not in PE_VISION_CONFIGS:
                raise ValueError(f"unknown PE vision config {name!r}; known: {sorted(PE_VISION_CONFIGS)}")
            cfg = PE_VISION_CONFIGS[name]
        if cfg.pool_type not in POOL_TYPES or cfg.act not in ACTS:
            raise ValueError(f"unsupported pool_type / act: {cfg.pool_type!r} / {cfg.act!r}")
        self.cfg = cfg
        self.precision = precision
        self.device = torch.device(device) if device is not None else None
        self._lib = hip.lib(hip.operands_for(precision))
        self._h = C.c_void_p()
        self._tensors: Dict[str, torch.Tensor] = {}
        self._workspace: Optional[torch.Tensor] = None
        self._loaded = False
        vc = hip.VitConfig(
            precision=hip.precision_code(precision), image_size=cfg.image_size, patch_size=cfg.patch_size,
            width=cfg.width, layers=cfg.layers, heads=cfg.heads, mlp_width=cfg.mlp_width, output_dim=cfg.output_dim,
            use_cls_token=int(cfg.use_cls_token), use_rope2d=int(cfg.use_rope2d), use_ln_pre=int(cfg.use_ln_pre),
            use_ln_post=int(cfg.use_ln_post), pool_type=POOL_TYPES[cfg.pool_type], pool_heads=cfg.attn_pooler_heads,
            act=ACTS[cfg.act], ln_eps=cfg.ln_eps)
        self._vc = vc
        hip.check(self._lib.samaudio_vit_create(C.byref(vc), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_side", None):
            self._lib.samaudio_vit_destroy(self._side[0])
            self._side = None
        if getattr(self, "_h", None):
            self._lib.samaudio_vit_destroy(self._h)
            self._h = None

    @property
    def act_dtype(self) -> torch.dtype:
        return hip.act_dtype(self.precision)

    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if self._tensors and self.device != device:
            raise RuntimeError("move the tower before load_state_dict (weights are converted onto the device)")
        self.device = device
        return self

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """`pe.CLIP` keys in: the vision tower's tensors below `visual.` (optionally `model.visual.`, as they appear
        under `vision_encoder.` in a SAMAudio checkpoint); the text tower of the CLIP pair (`token_embedding`,
        `transformer.*`, `ln_final`, `text_projection`, `logit_scale`, ... outside `visual.`) is not on this path and
        is ignored.  A dict whose keys start directly with `conv1.` etc. is taken as the bare tower."""
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        hip.require_gpu(self.device, "PEVisionTower")
        pref = re.compile(r"^(model\.)?visual\.")
        if any(pref.match(k) for k in state_dict):
            sd = {pref.sub("", k): v for k, v in state_dict.items() if pref.match(k)}
        else:
            sd = dict(state_dict)
        want = set(expected_keys(self.cfg))
        missing = sorted(want - set(sd))
        unexpected = sorted(set(sd) - want)
        if strict and (missing or unexpected):
            raise RuntimeError(f"Missing keys: {missing}, unexpected_keys: {unexpected}")
        if not missing:
            with torch.cuda.device(self.device):
                _register(self._lib.samaudio_vit_set_tensor, self._h, self._tensors,
                          convert_vision(sd, self.cfg, self.act_dtype, self.device))
                hip.check(self._lib.samaudio_vit_finalize(self._h))
            self._loaded = True
        return missing, unexpected

    def _side_context(self):
        if self._side is None:
            class _Holder:   # owns the second context's workspace (judge._ensure_ws contract: .device, ._workspace)
                pass
            holder = _Holder()
            holder.device, holder._workspace = self.device, None
            h = C.c_void_p()
            hip.check(self._lib.samaudio_vit_create(C.byref(self._vc), C.byref(h)))
            _register(self._lib.samaudio_vit_set_tensor, h, {}, self._tensors)   # borrowed: the same device tensors
            hip.check(self._lib.samaudio_vit_finalize(h))
            self._side = (h, holder, torch.cuda.Stream(device=self.device))
        return self._side

    def _encode_on(self, handle, owner, x, normalize, feats, tokens):
        n = x.shape[0]
        need = self._lib.samaudio_vit_workspace_bytes(handle, n)
        _ensure_ws(owner, need, lambda p, b: self._lib.samaudio_vit_set_workspace(handle, p, b))
        hip.check(self._lib.samaudio_vit_encode(handle, hip.ptr(x), n, int(bool(normalize)), hip.ptr(feats),
                                                hip.ptr(tokens), hip.current_stream_ptr()))

    @torch.inference_mode()
    def encode_image(self, frames: torch.Tensor, normalize: bool = False, return_tokens: bool = False):
        if not self._loaded:
            raise hip.SamAudioHipError("PEVisionTower: no weights loaded")
        cfg = self.cfg
        assert frames.dim() == 4 and frames.shape[1:] == (3, cfg.image_size, cfg.image_size), \
            f"frames must be [N, 3, {cfg.image_size}, {cfg.image_size}]"
        n = frames.shape[0]
        with torch.cuda.device(self.device):
            x = frames.to(self.device, torch.float32).contiguous()
            feats = torch.empty(n, cfg.output_dim, device=self.device, dtype=torch.float32)
            tokens = torch.empty(n, cfg.tokens, cfg.width, device=self.device, dtype=torch.float32) if return_tokens else None
            if self.streams == 2 and n >= 64:
                import threading
                h2, holder, side = self._side_context()
                k = n // 2
                main = torch.cuda.current_stream(self.device)
                side.wait_stream(main)
                errors = []

                def second():
                    try:
                        with torch.inference_mode(), torch.cuda.device(self.device), torch.cuda.stream(side):
                            self._encode_on(h2, holder, x[k:], normalize, feats[k:], None if tokens is None else tokens[k:])
                    except BaseException as exc:   # re-raised on the caller's thread
                        errors.append(exc)
                th = threading.Thread(target=second)
                th.start()
                self._encode_on(self._h, self, x[:k], normalize, feats[:k], None if tokens is None else tokens[:k])
                th.join()
                main.wait_stream(side)
                if errors:
                    raise errors[0]
            else:
                self._encode_on(self._h, self, x, normalize, feats, tokens)
        return (feats, tokens) if return_tokens else feats

    def __call__(self, frames: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        return self.encode_image(frames, normalize=normalize)
