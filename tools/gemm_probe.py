#!/usr/bin/env python
"""A few launches of chosen GEMM variants on the real DiT shapes, for rocprofv3 --pmc passes
(tools/pmc_gemm.sh).  usage: [PROBE_ROWS=4000] python tools/gemm_probe.py [variant:shape ...]  e.g. 22:w13 22:c_wq
PROBE_ROWS: rows per launch (default 8000 = one row group of 32 clips; 4000 = the two-group launches bench.py times)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import hip  # noqa: E402
from sam_audio_amd.config import preset_config  # noqa: E402
from tests import util  # noqa: E402

t = preset_config("large*").transformer
D, Fh, M = t.dim, t.ffn_hidden, int(os.environ.get("PROBE_ROWS", "8000"))
SHAPES = {"w13": (M, 2 * Fh, D, 1), "c_wq": (M, D, D, 0), "wo": (M, D, D, 0), "qkv": (M, 3 * D, D, 0), "w2": (M, D, Fh, 0)}
dev = torch.device("cuda:0")
for spec in (sys.argv[1:] or ["22:w13", "22:c_wq", "22:qkv", "22:w2"]):
    v, name = spec.split(":")
    m, n, k, sw = SHAPES[name]
    k *= int(os.environ.get("PROBE_KMUL", "1"))   # 3: the K' = 3K launches of the fp16x3 mode (split operands)
    A = torch.randn(m, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    out = torch.empty(m, n // 2 if sw else n, device=dev, dtype=torch.bfloat16)
    hip.lib().samaudio_debug_force_gemm_variant(int(v))
    # PROBE_FLAGS=32768 with PROBE_KMUL=3: the operand-sharing walk of the headline mode (GemmParams.flags bit 15: gemm8x_kernel; random
    # operands are fine for counters - the walk reads the first two thirds of every row)
    for _ in range(3):
        util.gemm("bf16", A, W, m, n, k, swiglu=sw, out_act=out, act_geom=(0, out.shape[1], 0), flags=int(os.environ.get("PROBE_FLAGS", "0")))
    torch.cuda.synchronize()
