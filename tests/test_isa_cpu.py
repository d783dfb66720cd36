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
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "kernels.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "--cuda-device-only",
                           "-S", "-o", str(out), os.path.join(CSRC, "kernels.hip")], stderr=subprocess.DEVNULL)
    return _kernels(out.read_text())


def test_shipped_qkv_prep_has_no_sdwa_rounding(kernels_asm):
    shipped = [v for k, v in kernels_asm.items() if "qkv_prep_bf16_kernelILb0E" in k]
    old = [v for k, v in kernels_asm.items() if "qkv_prep_bf16_kernelILb1E" in k]
    assert len(shipped) == 1 and len(old) == 1, sorted(k for k in kernels_asm if "qkv_prep" in k)
    assert not any("sdwa" in t.split()[0] for t in shipped[0]), "the shipped qkv_prep rounds with SDWA instructions again"
    assert any(t.split()[0] == "v_cvt_pk_bf16_f32" for t in shipped[0]), "expected the hardware conversion"
    assert _sdwa_behind_packed_f32(old[0]) >= 10, "the reproducer form no longer shows the pattern: is the scan still valid?"


def test_sdwa_behind_packed_fp32_stays_where_it_is_known(kernels_asm):
    """kernels.hip, bf16 build: besides the reproducer, only the two head-norm kernels have such pairs (2 each; never seen to differ -
    DESIGN.md section 8 lists them with the three of attention.hip / vit_kernels.hip).  A new entry here means a new kernel rounds
    behind packed-fp32 math with the written-out f2bf: use pack_h16x2 there."""
    known = ("qkv_prep_bf16_kernelILb1E", "headnorm_kernelINS_6bf16_tE", "headnorm_layers_kernelINS_6bf16_tE")
    found = {k: _sdwa_behind_packed_f32(v) for k, v in kernels_asm.items()}
    new = {k: n for k, n in found.items() if n and not any(x in k for x in known)}
    assert not new, new
