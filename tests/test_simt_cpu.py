"""Runs a light subset of the `-m gpu` tests on the functional SIMT simulator (oracle/simt/: every product source,
kernels included, compiled unchanged for the host; one fiber per GPU thread, lane-exact MFMA / DMA / DPP emulation).

This is what stands in for hardware in the CPU suite: the kernels' FUNCTION (indexing, fragment layouts, swizzles,
masks, epilogues) is exercised through the same C ABI and the same test bodies that run on MI355X.  The simulator
itself is calibrated by the GPU-verified kernels: all of tests/test_kernels_gpu.py, test_gemm_gpu.py and
test_gemm2_gpu.py pass on it.  It models no timing and no asynchrony - races and performance stay with the GPU runs.
The full sweep (incl. the codec-heavy end-to-end tests, ~80 min) is run by hand:
    SAMAUDIO_EMU_DRYRUN=simt python -m pytest tests -m gpu -q
"""
import os
import subprocess
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout, **extra):
    env = dict(os.environ, SAMAUDIO_EMU_DRYRUN="simt", **extra)
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + args,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    return tail


def test_every_non_gemm_kernel_on_the_simulator():
    out = _run(["tests/test_kernels_gpu.py"], 600)
    assert " passed" in out and "failed" not in out


def test_no_kernel_consumes_lds_it_never_wrote():
    """SAMAUDIO_SIMT_POISON=1: every LDS array is filled with 0xFF bytes (NaN as fp32 and bf16) before each workgroup,
    as a real CU's LDS holds whatever ran there before - the class of bug behind DESIGN.md section 8 (an MFMA operand
    of padding lanes read past the rows the kernel had written: 0 x NaN).  All non-GEMM kernels, both candidates and
    the new streaming kernels; the GEMM sweep with poison is part of the by-hand run."""
    out = _run(["tests/test_kernels_gpu.py", "tests/test_zz_next_rows_gpu.py", "-k",
                "not (judge or separate or predict or frame_logits or peav_transformer or visual_prompt)"], 900,
               SAMAUDIO_SIMT_POISON="1")
    assert " passed" in out and "failed" not in out


def test_gemm_kernels_on_the_simulator():
    """One pass over every shipped tile variant and epilogue of gemm2.hip plus the fp32 / bf16 kernels of gemm.hip."""
    out = _run(["tests/test_gemm2_gpu.py", "tests/test_gemm_gpu.py", "-k", "not conv_forms"], 900)
    assert " passed" in out and "failed" not in out


def test_pipelined_gemm_kernels_with_dma_landing_as_late_as_the_isa_allows():
    """SAMAUDIO_SIMT_DMA=late: a global_load_lds lands only when the issuing wave's s_waitcnt vmcnt(N) / __syncthreads
    retires it, so a kernel whose counted waits let a wave read a slab too early reads stale LDS here.  (Mutation check
    done by hand: loosening gemm2.hip's counts by 8 fails all 43 tests of this file in this mode and none in the
    default one.)  Covers the ring tiles, conv7h / the fused residual units and the 8-phase kernels."""
    env_extra = {"SAMAUDIO_SIMT_DMA": "late"}
    old = {k: os.environ.get(k) for k in env_extra}
    os.environ.update(env_extra)
    try:
        out = _run(["tests/test_gemm2_gpu.py", "-k", "not conv_forms"], 1200)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert " passed" in out and "failed" not in out


def test_new_streaming_kernels_and_the_peav_transformer_on_the_simulator():
    out = _run(["tests/test_zz_next_rows_gpu.py", "-k",
                "masked_groupnorm or layernorm_rows or peav_transformer or frame_logits"], 900)
    assert "10 passed" in out


def test_round2_policy_tail_split_and_vision_tower_on_the_simulator():
    """The shipped (round-2) tile policy end to end on small shapes: the 8-phase launch split into whole rounds + 128x128
    quadrant tail must be bitwise invisible, and the PE-Core vision tower (tests/test_vit_gpu.py: every structural switch,
    fp32 and bf16, uint8 video -> resize -> tower) must match its oracle with the real kernel code."""
    out = _run(["tests/test_gemm2_gpu.py", "tests/test_vit_gpu.py", "-k",
                "tail_split or 8phase_family or (test_vit_gpu and not checkpoint)"], 900)
    assert " passed" in out and "failed" not in out


def test_t5_prompt_encoder_on_the_simulator():
    """The T5 encoder stack (tests/test_t5_gpu.py: ragged masks, > 64 keys per row, 16 / 32 / 128-wide heads, ReLU and
    gelu_new, error paths, the T5TextEncoder wrapper) against transformers' T5EncoderModel with the real kernel code."""
    out = _run(["tests/test_t5_gpu.py"], 600)
    assert " passed" in out and "failed" not in out


def test_modernbert_text_tower_on_the_simulator():
    """The Judge's / PE-A-Frame's ModernBERT text tower (tests/test_mbert_gpu.py: every hidden state, ragged masks, sequences
    longer than the local window, both layer patterns, error paths) against transformers' ModernBertModel with the real
    kernel code."""
    out = _run(["tests/test_mbert_gpu.py"], 600)
    assert " passed" in out and "failed" not in out


def test_operand_sharing_x3_kernel_with_dma_landing_as_late_as_the_isa_allows():
    """gemm8x_kernel (round 6: the 8-phase kernel that shares operand tiles and fragments between the three products of the compensated
    mode's K-concatenated operands, GemmParams.flags bit 15) against the plain walk and - bit for bit - against gemm8s_kernel's walk of
    the same order, in both weight layouts, with every DMA landing only when its hand-counted wait retires it (vmcnt(4) / vmcnt(8) /
    vmcnt(0) per product: a wait placed behind the wrong barrier in an experimental two-phase form failed exactly here)."""
    env_extra = {"SAMAUDIO_SIMT_DMA": "late"}
    old = {k: os.environ.get(k) for k in env_extra}
    os.environ.update(env_extra)
    try:
        out = _run(["tests/test_x3_gpu.py", "-k", "sharing_the_operand_tiles"], 600)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert "3 passed" in out and "failed" not in out
