import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Scratch handed to the engine is filled with 0xFF bytes (NaN in fp32 and bf16) in every test, so a kernel that reads
# scratch nobody wrote fails deterministically instead of depending on what the allocator returned.
os.environ.setdefault("SAMAUDIO_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---------------------------------------------------------------------------------------------------------------
# Dry run of the `-m gpu` tests on the CPU emulation (test infrastructure, see oracle/emu/):
#     SAMAUDIO_EMU_DRYRUN=1 python -m pytest tests/test_zz_next_rows_gpu.py tests/test_path_gpu.py -m gpu -q
# binds the product's host classes to oracle/_emu/libsamaudio_emu.so (the unchanged host orchestration sources linked
# against emulated kernel launchers) and lets "device" tensors live on the CPU, so that the Python host code and the
# GPU tests themselves can be shaken out without hardware.  It proves nothing about the HIP kernels and is never
# active in a normal run: it needs the env switch, and it is set up here - not in the product - by monkeypatching.
# ---------------------------------------------------------------------------------------------------------------
EMU_DRYRUN = os.environ.get("SAMAUDIO_EMU_DRYRUN") == "1"


def _enable_emu_dryrun():
    import contextlib
    import ctypes as C
    import subprocess
    import torch
    from sam_audio_amd import hip
    emu = os.path.join(ROOT, "oracle", "_emu", "libsamaudio_emu.so")
    if not os.path.exists(emu):
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "emu", "build.sh")])
    lib = C.CDLL(emu)
    for name, (res, args) in hip._PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    hip._lib = lib
    hip.require_gpu = lambda device, who: None
    hip.current_stream_ptr = lambda: C.c_void_p(0)
    torch.cuda.device = lambda *a, **k: contextlib.nullcontext()
    torch.cuda.current_stream = lambda *a, **k: None
    os.environ["SAMAUDIO_NO_FOLD"] = "1"  # the folded cross-attention projection is a GPU-only bf16 fast path


if EMU_DRYRUN:
    _enable_emu_dryrun()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if EMU_DRYRUN:
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sam_audio_amd import hip
    hip.lib()  # fail loudly if the HIP library is missing on a GPU box
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _poison_lds(request):
    """Before every GPU test, leave all LDS full of NaN bit patterns (see samaudio_debug_poison_lds)."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            from sam_audio_amd import hip
            hip.check(hip.lib().samaudio_debug_poison_lds(hip.current_stream_ptr()))
    yield
