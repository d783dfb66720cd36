"""ctypes binding of libsamaudio_hip.so (C ABI in include/samaudio.h).

There is deliberately NO fallback: if the shared library is missing or fails to load, importing the
product path raises.  (The CPU restatement under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsamaudio_hip.so")
# the same sources built with -DSA_OPERAND_FP16: the 16-bit GEMM-operand format is IEEE fp16 instead of bfloat16
LIB_PATHS = {"bf16": LIB_PATH, "fp16": os.path.join(_HERE, "libsamaudio_hip_f16.so")}

F32, BF16 = 0, 1             # precision codes of the C ABI: fp32 parity mode | 16-bit GEMM operands (bf16 or fp16 by library)
DT_F32, DT_BF16, DT_I64, DT_U8 = 0, 1, 2, 3   # DT_BF16 = "the library's 16-bit operand format"
PRECISIONS = ("bf16", "fp16", "mixed", "fp32")
# COMPENSATED 16-bit operands under fp32 storage (SAMAudio only; samaudio.h SAMAUDIO_OPT_X3_CLASSES): an fp32 engine context whose six
# big GEMM classes multiply hi/lo-split operands on the 16-bit MFMA - "fp16x3" in libsamaudio_hip_f16.so (22 mantissa bits per
# operand: the mode that holds the 1e-3 parity bound on trained-like weights), "bf16x3" the same mechanism on
# bfloat16 halves in libsamaudio_hip.so (16 bits; what the CPU simulator build exercises).
X3_PRECISIONS = ("fp16x3", "bf16x3")


def check_precision(precision: str, x3_ok: bool = False) -> None:
    if precision not in PRECISIONS + (X3_PRECISIONS if x3_ok else ()):
        raise ValueError("precision must be 'bf16', 'fp16', 'mixed' or 'fp32'" + (", 'fp16x3' or 'bf16x3'" if x3_ok else ""))


def is_x3(precision: str) -> bool:
    return precision in X3_PRECISIONS


def storage_precision(precision: str) -> str:
    """The precision of everything that is not a compensated GEMM (and of the towers beside the DiT): fp32 for the x3 modes."""
    return "fp32" if is_x3(precision) else precision


def tower_precision(precision: str) -> str:
    """The towers beside the DiT (Judge, span predictor, vision tower) have no compensated mode: beside an x3 DiT they run on the
    library's plain 16-bit operands, as they do beside a plain 16-bit one."""
    return {"fp16x3": "fp16", "bf16x3": "bf16"}.get(precision, precision)


def precision_code(precision: str) -> int:
    return F32 if storage_precision(precision) == "fp32" else BF16


def operands_for(precision: str) -> str:
    """Which build of the library a host object of this precision talks to (fp32 mode lives in both; use the default)."""
    return "fp16" if precision in ("fp16", "mixed", "fp16x3") else "bf16"


def act_dtype(precision: str):
    import torch
    return {"bf16": torch.bfloat16, "fp16": torch.float16, "mixed": torch.float16, "fp32": torch.float32,
            "fp16x3": torch.float32, "bf16x3": torch.float32}[precision]


def half_dtype(precision: str):
    """the 16-bit format of the hi / lo halves of an x3 precision"""
    import torch
    return torch.float16 if operands_for(precision) == "fp16" else torch.bfloat16


def dtype_code(dtype, operands: Optional[str] = None, alt_ok: bool = False) -> int:
    """C-ABI dtype code of a torch dtype.  `operands` ("bf16" | "fp16"): the build of the library the tensor is handed to -
    a 16-bit tensor in the other build's format would be reinterpreted silently, so it is refused here (`alt_ok`: the
    tensor is the weight of a SAMAUDIO_OPT_ALT16_CLASSES class of a mixed-precision model, which IS in the other format)."""
    import torch
    if operands is not None and not alt_ok and dtype in (torch.bfloat16, torch.float16):
        want = torch.float16 if operands == "fp16" else torch.bfloat16
        if dtype != want:
            raise TypeError(f"{dtype} tensor handed to the {operands}-operand build of libsamaudio_hip")
    return {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float16: DT_BF16}[dtype]
ODE_EULER, ODE_MIDPOINT = 0, 1
OPT_TAIL_SPLIT, OPT_F32_CLASSES, OPT_QUANT_CLASSES, OPT_QUANT_FORMAT, OPT_ALT16_CLASSES, OPT_PREFETCH_ROWS, OPT_SENTINEL = 1, 2, 3, 4, 5, 6, 7
OPT_X3_CLASSES = 9
SENTINEL_SLOTS = 16
# GEMM classes of the DiT / codec (samaudio.h SAMAUDIO_CLS_*), in bit order
CLASSES = ("time", "out", "in", "prep", "yemb", "ckv", "patch", "qkv", "wo", "cwq", "cwo", "w13", "w2", "codec")
CLS = {name: 1 << i for i, name in enumerate(CLASSES)}
CLS_F32_CAPABLE = CLS["time"] | CLS["out"] | CLS["in"] | CLS["prep"] | CLS["yemb"]
# what precision="bf16" / "fp16" models switch on by default: the classes that touch the ODE state or the hoisted
# conditioning.  Measured on the full solve at large* (profiles/r3_call1/error_budget.log, bf16 / fp16 operands, max-abs on
# |latent| <= 5.2): out 4.3e-3 / 6.0e-4, in 2.5e-3 / 3.5e-4, prep 2.6e-3 / 3.0e-4 against yemb 9.7e-4 / 1.3e-4 and
# time 4.6e-4 / 5.9e-5 - the last two cost 1 ms per evaluation in fp32 and buy nothing measurable; they stay selectable.
CLS_F32_DEFAULT = CLS["out"] | CLS["in"] | CLS["prep"]
# precision="mixed": the five big GEMM classes of the DiT layers (96 % of the flops, <= 2e-4 of the error each on bf16 operands)
# read bfloat16 operands inside the fp16 build; everything else stays fp16 (samaudio.h SAMAUDIO_OPT_ALT16_CLASSES)
CLS_ALT16_MIXED = CLS["qkv"] | CLS["wo"] | CLS["cwq"] | CLS["w13"] | CLS["w2"]
ALT16_WEIGHTS = {"wqkv": "qkv", "wo": "wo", "c_wq": "cwq", "w13": "w13", "w2": "w2"}   # engine tensor L<i>.<name> -> class
# the six big GEMM classes of the DiT layers (97 % of the flops) and the engine weight each reads
X3_ATTENTION = 1 << 14   # samaudio.h SAMAUDIO_X3_ATTENTION: the self-attention's contractions on split operands as well
CLS_X3_GEMMS = CLS["qkv"] | CLS["wo"] | CLS["cwq"] | CLS["cwo"] | CLS["w13"] | CLS["w2"]
CLS_X3_DEFAULT = CLS_X3_GEMMS | CLS["patch"] | CLS["ckv"] | CLS["codec"] | X3_ATTENTION
X3_WEIGHTS = {"wqkv": "qkv", "wo": "wo", "c_wq": "cwq", "c_wo": "cwo", "w13": "w13", "w2": "w2"}
QUANT_FORMATS = {"bf16": 1, "fp16": 2}
SENTINEL_NAMES = CLASSES + ("norm", "attn")


def class_mask(classes) -> int:
    """'time,out' | ['time', 'out'] | int | None -> bit mask of SAMAUDIO_CLS_* ("all" = every class, None = none)."""
    if classes is None:
        return 0
    if isinstance(classes, int):
        return classes
    if isinstance(classes, str):
        classes = [c for c in classes.split(",") if c]
    mask = 0
    for c in classes:
        mask |= (1 << len(CLASSES)) - 1 if c == "all" else (X3_ATTENTION if c == "attn" else CLS[c])
    return mask


ACT_NONE, ACT_SNAKE, ACT_TANH, ACT_SILU, ACT_GELU, ACT_QUICK_GELU, ACT_RELU, ACT_GELU_TANH = 0, 1, 2, 3, 4, 5, 6, 7

ERR_ARG, ERR_WEIGHT, ERR_WORKSPACE, ERR_HIP, ERR_STATE = -1, -2, -3, -4, -5


class SamAudioHipError(RuntimeError):
    pass


class Config(C.Structure):
    """Mirror of `samaudio_config`."""
    _fields_ = [
        ("precision", C.c_int32), ("dim", C.c_int32), ("n_heads", C.c_int32), ("n_layers", C.c_int32),
        ("ffn_hidden", C.c_int32), ("latent_channels", C.c_int32), ("text_dim", C.c_int32),
        ("video_dim", C.c_int32), ("freq_dim", C.c_int32), ("anchor_dim", C.c_int32),
        ("anchor_vocab", C.c_int32), ("max_positions", C.c_int32), ("norm_eps", C.c_float),
        ("codec_dim", C.c_int32), ("codec_latent", C.c_int32), ("enc_dim", C.c_int32),
        ("dec_dim", C.c_int32), ("enc_rates", C.c_int32 * 4), ("dec_rates", C.c_int32 * 4),
    ]


class GemmParams(C.Structure):
    """Mirror of `sa::GemmParams` (sam_audio_amd/csrc/common.h) for the samaudio_op_gemm test hook."""
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p),
        ("a_off", C.c_long), ("a_bstride", C.c_long), ("lda", C.c_long), ("tap_stride", C.c_long),
        ("kc", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("nbatch", C.c_int),
        ("bias", C.c_void_p), ("chan_mod", C.c_int), ("swiglu", C.c_int),
        ("gate_tab", C.c_void_p), ("gate", C.c_void_p), ("gate_ld", C.c_long),
        ("rows_per_gate", C.c_int), ("alpha", C.c_float),
        ("res", C.c_void_p), ("res_bstride", C.c_long), ("res_ld", C.c_long), ("res_off", C.c_long),
        ("out_f32", C.c_void_p), ("f32_bstride", C.c_long), ("f32_ld", C.c_long), ("f32_off", C.c_long),
        ("out_act", C.c_void_p), ("act_bstride", C.c_long), ("act_ld", C.c_long), ("act_off", C.c_long),
        ("act", C.c_int), ("f32_act", C.c_int), ("act_alpha", C.c_void_p),
        ("c_lo", C.c_long), ("c_hi", C.c_long), ("c_ld_rel", C.c_long),
        ("w_bstride", C.c_long), ("raster_gm", C.c_int), ("flags", C.c_int), ("tag", C.c_int),
        ("pf_ptr", C.c_void_p), ("pf_bytes", C.c_long),
    ]


class PeavDims(C.Structure):
    """Mirror of `samaudio_peav_dims`."""
    _fields_ = [("dim", C.c_int32), ("n_heads", C.c_int32), ("n_layers", C.c_int32), ("ffn_hidden", C.c_int32),
                ("in_dim", C.c_int32), ("max_positions", C.c_int32), ("attn_bias", C.c_int32), ("norm_eps", C.c_float)]


class JudgeConfig(C.Structure):
    """Mirror of `samaudio_judge_config`."""
    _fields_ = [("precision", C.c_int32), ("transformer", PeavDims), ("finetune_transformer", PeavDims),
                ("codec_dim", C.c_int32), ("text_hidden", C.c_int32), ("bottleneck_dim", C.c_int32)]


class FrameConfig(C.Structure):
    """Mirror of `samaudio_frame_config`."""
    _fields_ = [("precision", C.c_int32), ("audio", PeavDims), ("codec_dim", C.c_int32), ("embed_dim", C.c_int32)]


class VitConfig(C.Structure):
    """Mirror of `samaudio_vit_config`."""
    _fields_ = [("precision", C.c_int32), ("image_size", C.c_int32), ("patch_size", C.c_int32), ("width", C.c_int32),
                ("layers", C.c_int32), ("heads", C.c_int32), ("mlp_width", C.c_int32), ("output_dim", C.c_int32),
                ("use_cls_token", C.c_int32), ("use_rope2d", C.c_int32), ("use_ln_pre", C.c_int32),
                ("use_ln_post", C.c_int32), ("pool_type", C.c_int32), ("pool_heads", C.c_int32), ("act", C.c_int32),
                ("ln_eps", C.c_float)]


class T5Config(C.Structure):
    """Mirror of `samaudio_t5_config`."""
    _fields_ = [("precision", C.c_int32), ("vocab", C.c_int32), ("d_model", C.c_int32), ("d_kv", C.c_int32),
                ("heads", C.c_int32), ("d_ff", C.c_int32), ("layers", C.c_int32), ("max_len", C.c_int32),
                ("act", C.c_int32), ("ln_eps", C.c_float)]


class MBertConfig(C.Structure):
    """Mirror of `samaudio_mbert_config`."""
    _fields_ = [("precision", C.c_int32), ("vocab", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32),
                ("intermediate", C.c_int32), ("layers", C.c_int32), ("global_every", C.c_int32), ("window", C.c_int32),
                ("max_len", C.c_int32), ("ln_eps", C.c_float)]


class KernelStat(C.Structure):
    """Mirror of `samaudio_kernel_stat`."""
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int64), ("flops", C.c_double), ("bytes", C.c_double),
                ("ms", C.c_double)]


_lib = None     # the default (bf16-operand) library; tests/conftest.py swaps it for its CPU dry-run builds
_libs = {}      # operand format -> loaded library
# library of this THREAD's most recent call: check() reads samaudio_last_error() (itself a thread-local string of the
# library) from the one that reported.  Per thread, because the concurrent row groups and the two-stream vision tower call
# into the bf16 and fp16 builds from several threads at once.
_tls = threading.local()


class _Handle:
    """CDLL proxy that remembers which library was called last (two builds of the library can be loaded side by side)."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)

        def call(*args, _fn=fn, _self=self, _tls=_tls):   # (_tls bound here: a destructor at interpreter exit finds the module's globals gone)
            _tls.last = _self
            return _fn(*args)
        object.__setattr__(self, name, call)
        return call


_PROTOS = {
    "samaudio_last_error": (C.c_char_p, []),
    "samaudio_version": (C.c_char_p, []),
    "samaudio_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "samaudio_destroy": (None, [C.c_void_p]),
    "samaudio_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "samaudio_finalize": (C.c_int, [C.c_void_p, C.c_int]),
    "samaudio_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "samaudio_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64]),
    "samaudio_set_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "samaudio_prepare": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "samaudio_prepare_latent": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "samaudio_codec_decode_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "samaudio_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "samaudio_ode_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "samaudio_codec_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "samaudio_codec_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "samaudio_profile_begin": (C.c_int, [C.c_void_p]),
    "samaudio_sentinel_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p]),
    "samaudio_debug_force_gemm_variant": (None, [C.c_int]),
    "samaudio_debug_set_flag": (None, [C.c_int, C.c_int]),
    "samaudio_debug_poison_lds": (C.c_int, [C.c_void_p]),
    "samaudio_profile_end": (C.c_int, [C.c_void_p, C.POINTER(KernelStat), C.c_int, C.POINTER(C.c_int)]),
    "samaudio_op_gemm": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "samaudio_op_resunit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "samaudio_op_rmsnorm_mod": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                             C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "samaudio_op_groupnorm_silu": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    "samaudio_op_qkv_prep": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    "samaudio_op_self_attention": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]),
    "samaudio_op_cross_attention": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    "samaudio_op_cross_attn_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "samaudio_op_layernorm_accum": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "samaudio_op_masked_groupnorm_silu": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    "samaudio_op_layernorm_rows": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 4 + [C.c_int, C.c_int64, C.c_int,
                                                                                          C.c_float, C.c_void_p]),
    "samaudio_op_split3": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "samaudio_judge_create": (C.c_int, [C.POINTER(JudgeConfig), C.POINTER(C.c_void_p)]),
    "samaudio_judge_destroy": (None, [C.c_void_p]),
    "samaudio_judge_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "samaudio_judge_finalize": (C.c_int, [C.c_void_p]),
    "samaudio_judge_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "samaudio_judge_set_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "samaudio_judge_score": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "samaudio_judge_encode": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "samaudio_frame_create": (C.c_int, [C.POINTER(FrameConfig), C.POINTER(C.c_void_p)]),
    "samaudio_frame_destroy": (None, [C.c_void_p]),
    "samaudio_frame_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "samaudio_frame_finalize": (C.c_int, [C.c_void_p]),
    "samaudio_frame_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "samaudio_frame_set_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "samaudio_frame_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "samaudio_mbert_create": (C.c_int, [C.POINTER(MBertConfig), C.POINTER(C.c_void_p)]),
    "samaudio_mbert_destroy": (None, [C.c_void_p]),
    "samaudio_mbert_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "samaudio_mbert_finalize": (C.c_int, [C.c_void_p]),
    "samaudio_mbert_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "samaudio_mbert_set_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "samaudio_mbert_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "samaudio_t5_create": (C.c_int, [C.POINTER(T5Config), C.POINTER(C.c_void_p)]),
    "samaudio_t5_destroy": (None, [C.c_void_p]),
    "samaudio_t5_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "samaudio_t5_finalize": (C.c_int, [C.c_void_p]),
    "samaudio_t5_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "samaudio_t5_set_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "samaudio_t5_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "samaudio_vit_create": (C.c_int, [C.POINTER(VitConfig), C.POINTER(C.c_void_p)]),
    "samaudio_vit_destroy": (None, [C.c_void_p]),
    "samaudio_vit_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "samaudio_vit_finalize": (C.c_int, [C.c_void_p]),
    "samaudio_vit_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "samaudio_vit_set_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "samaudio_vit_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


def lib(operands: str = "bf16"):
    """Load (once) and return the shared library for a 16-bit operand format; raises if it is not there."""
    global _lib
    if operands == "bf16" and _lib is not None:
        return _lib
    if operands in _libs:
        return _libs[operands]
    path = LIB_PATHS[operands]
    if operands == "bf16" and os.environ.get("SAMAUDIO_LIB_AB"):   # tuning only: A/B an older build of the library
        path = os.environ["SAMAUDIO_LIB_AB"]
    if not os.path.exists(path):
        raise SamAudioHipError(
            f"{path} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or sam_audio_amd/csrc/build.sh.  There is no CPU fallback for the separate() hot path.")
    cdll = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(cdll, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    handle = _Handle(cdll)
    _libs[operands] = handle
    if operands == "bf16":
        _lib = handle
    # tuning only: A/B switches between kernel generations (sam_audio_amd/csrc/kernels.h), e.g. "8=1,9=1"
    for kv in filter(None, os.environ.get("SAMAUDIO_DEBUG_FLAGS", "").split(",")):
        k, v = kv.split("=")
        cdll.samaudio_debug_set_flag(int(k), int(v))
    return handle


def check(code: int) -> None:
    """Map C status codes to the exception types the reference raises at the same boundary
    (SURVEY.md §8b 'Errors')."""
    if code == 0:
        return
    src = getattr(_tls, "last", None) or lib()
    msg = src.samaudio_last_error().decode()
    if code == ERR_ARG:
        raise AssertionError(msg)
    if code == ERR_WEIGHT:
        raise RuntimeError(msg)
    raise SamAudioHipError(f"[{code}] {msg}")


def require_gpu(device, who: str) -> None:
    """The product computes on a ROCm GPU only: weights are converted onto the device and every kernel is HIP."""
    if device is None or device.type != "cuda":
        raise SamAudioHipError(f"{who} needs a ROCm GPU: there is no CPU fallback")


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> C.c_void_p:
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "tensor handed to the HIP library must be contiguous"
    return C.c_void_p(t.data_ptr())


def shape_array(shape: Sequence[int]):
    return (C.c_int64 * len(shape))(*[int(s) for s in shape])
