"""Static check of generated code (no GPU): the instruction pattern behind round 4's run-to-run difference must not come back.

hipcc turned qkv_prep's written-out bf16 rounding into SDWA word-select instructions directly behind the packed-fp32 instructions
that produce their operands; beside another kernel's waves such a read is sometimes too early (DESIGN.md section 8,
tools/stress_qkv_prep.py).  The shipped kernel rounds with the hardware conversion and has no SDWA instruction at all; the old form
(debug flag 29 = 1, kept as the reproducer) must still show the pattern - otherwise this scan has stopped seeing what it is for."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sam_audio_amd", "csrc")


def _kernels(asm: str):
    """{mangled kernel name: [instruction lines]}"""
    out, cur = {}, None
    for ln in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        t = ln.strip()
        if cur is not None and t and not t.startswith((";", ".")):
            cur.append(t)
    return out


def _sdwa_behind_packed_f32(lines, window=3):
    """SDWA instructions that read a register a packed-fp32 instruction wrote at most `window` instructions earlier"""
    hits, recent = 0, []
    for t in lines:
        op = t.split()[0]
        if "sdwa" in op and "," in t:
            srcs = set(re.findall(r"\bv(\d+)\b", t.split(",", 1)[1]))
            if any(p.startswith("v_pk_") and p.endswith("f32") and (srcs & d) for p, d in recent[-window:]):
                hits += 1
        m = re.match(r"\S+\s+v\[(\d+):(\d+)\]", t)
        dst = {str(i) for i in range(int(m.group(1)), int(m.group(2)) + 1)} if m else set(re.findall(r"^\S+\s+v(\d+)\b", t))
        recent.append((op, dst))
    return hits


@pytest.fixture(scope="module")
def kernels_asm(tmp_path_factory):
    """the bf16 build's code of every file whose kernels round fp32 results to 16 bits outside a GEMM epilogue"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    tmp, out, procs = tmp_path_factory.mktemp("isa"), {}, []
    for name in ("kernels", "attention", "vit_kernels", "peav_kernels", "t5_kernels"):
        dst = tmp / f"{name}.s"
        procs.append((dst, subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
                                             "--cuda-device-only", "-S", "-o", str(dst), os.path.join(CSRC, f"{name}.hip")],
                                            stderr=subprocess.DEVNULL)))
    for dst, p in procs:
        assert p.wait() == 0, dst
        out.update(_kernels(dst.read_text()))
    return out


def test_shipped_qkv_prep_has_no_sdwa_rounding(kernels_asm):
    shipped = [v for k, v in kernels_asm.items() if "qkv_prep_bf16_kernelILb0E" in k]
    old = [v for k, v in kernels_asm.items() if "qkv_prep_bf16_kernelILb1E" in k]
    assert len(shipped) == 1 and len(old) == 1, sorted(k for k in kernels_asm if "qkv_prep" in k)
    assert not any("sdwa" in t.split()[0] for t in shipped[0]), "the shipped qkv_prep rounds with SDWA instructions again"
    assert any(t.split()[0] == "v_cvt_pk_bf16_f32" for t in shipped[0]), "expected the hardware conversion"
    assert _sdwa_behind_packed_f32(old[0]) >= 10, "the reproducer form no longer shows the pattern: is the scan still valid?"


def test_no_kernel_rounds_with_sdwa_behind_packed_fp32(kernels_asm):
    """Round 4 found such pairs in qkv_prep (31: the difference's origin), the two head-norm kernels and the generic cross-attention
    kernel (2 each), the vision tower's RoPE split (1) and its pooling attention (1); all of them now round through pack_h16x2 /
    store2 (the hardware conversion).  Only the reproducer keeps the pattern.  A hit here = a kernel that rounds with the written-out
    f2bf right behind packed-fp32 arithmetic: route it through pack_h16x2."""
    found = {k: _sdwa_behind_packed_f32(v) for k, v in kernels_asm.items()}
    new = {k: n for k, n in found.items() if n and "qkv_prep_bf16_kernelILb1E" not in k}
    assert not new, new


# ---------------------------------------------------------------------------------------------------------------------------
# gemm8.hip: the 8-phase kernels issue their direct-to-LDS loads as inline assembly and order them with hand-counted
# `s_waitcnt vmcnt(N)` - a count the compiler cannot see.  A register spill is a scratch (vector-memory) operation: inside the
# K loop it would both slow the loop and sit in the vmcnt queue the manual counts assume to hold DMA loads only.  Today hipcc spills
# 43 (plain) / 71 (implicit-convolution) VGPRs of gemm8_kernel, all in the per-tile prologue / epilogue; this
# keeps it that way.
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gemm8_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    dst = tmp_path_factory.mktemp("isa8") / "gemm8.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-mllvm",
                           "-pragma-unroll-threshold=262144", "-Wno-inline-asm", "--cuda-device-only", "-S", "-o", str(dst),
                           os.path.join(CSRC, "gemm8.hip")], stderr=subprocess.DEVNULL)
    text = dst.read_text()
    spills = {}
    for blk in text.split("- .agpr_count:")[1:]:
        name, vs = re.search(r"\.name:\s+(\S+)", blk), re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
        if name and vs:
            spills[name.group(1)] = int(vs.group(1))
    return _kernels(text), spills


def test_gemm8_k_loops_are_free_of_scratch_traffic(gemm8_asm):
    kernels, spills = gemm8_asm
    seen = 0
    for name, lines in kernels.items():
        if "gemm8_kernel" not in name and "gemm8s_kernel" not in name and "gemm8x_kernel" not in name:   # (gemm8x: the x3 mode's dominant kernel)
            continue
        mfma = [i for i, t in enumerate(lines) if t.startswith("v_mfma")]
        assert mfma, name
        seen += 1
        region = lines[mfma[0]: mfma[-1] + 1]
        bad = [t for t in region if t.startswith("scratch_") or t.startswith("buffer_load") or t.startswith("buffer_store")]
        assert not bad, f"{name}: {len(bad)} scratch / buffer operations between the first and the last MFMA, e.g. {bad[:3]}"
        # the pipelined / plain 128x128 forms must not spill at all; the 256x256 kernel's prologue / epilogue spills are tracked
        limit = 96 if ("gemm8_kernel" in name or "gemm8x_kernel" in name) else 0
        assert spills.get(name, 0) <= limit, f"{name}: vgpr_spill_count {spills.get(name)} > {limit}"
    assert seen >= 8 and any("gemm8x_kernel" in n for n in kernels)


# ---------------------------------------------------------------------------------------------------------------------------
# gemm.hip (round 6): built WITHOUT the full-unroll budget its epilogue loops need, hipcc indexed the accumulator array dynamically
# and kept it in scratch memory - every 128 x 128 instantiation (bf16 and fp32) stored all accumulators after every K slab, behind
# s_nop waits for the MFMA results (profiles/r6_call14/: the 128-channel DAC-VAE stage 96 -> 49 ms per step with the flag alone).
# The flags are read from csrc/build.sh, so that the scan sees what ships.
# ---------------------------------------------------------------------------------------------------------------------------
def test_gemm_kernels_keep_their_accumulators_in_registers(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    build = open(os.path.join(CSRC, "build.sh")).read()
    m = re.search(r'if \[ \$f = gemm \]; then EXTRA="([^"]*)"', build)
    assert m and "-pragma-unroll-threshold" in m.group(1), "csrc/build.sh: gemm.hip needs the full-unroll budget"
    dst = tmp_path / "gemm.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", *m.group(1).split(),
                           "-DSA_OPERAND_FP16", "--cuda-device-only", "-S", "-o", str(dst), os.path.join(CSRC, "gemm.hip")],
                          stderr=subprocess.DEVNULL)
    seen = 0
    for name, lines in _kernels(dst.read_text()).items():
        if "gemm_kernel" not in name:
            continue
        mfma = [i for i, t in enumerate(lines) if t.startswith("v_mfma")]
        assert mfma, name
        seen += 1
        # the K loop ends with the last MFMA in front of the first epilogue store; nothing in it may touch scratch memory
        bad = [t for t in lines[mfma[0]: mfma[-1] + 1] if t.startswith("scratch_")]
        assert not bad, f"{name}: {len(bad)} scratch operations between the first and the last MFMA, e.g. {bad[:3]}"
    assert seen >= 10   # 3 bf16 + 3 fp32 plain tiles, 4 split-weight tiles
