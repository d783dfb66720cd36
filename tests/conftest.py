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
#     SAMAUDIO_EMU_DRYRUN=simt ... goes one level deeper: oracle/_simt/libsamaudio_simt.so is EVERY product source, kernels
# included, compiled unchanged for the host and executed on a functional SIMT simulator (oracle/simt/: one fiber per GPU
# thread, 64-lane waves, barriers, LDS, MFMA / DMA / DPP builtins emulated lane-exactly).  It checks what a kernel
# computes - indexing, fragment layouts, swizzles, masks, epilogues - not when (no timing, no races).
EMU_MODE = os.environ.get("SAMAUDIO_EMU_DRYRUN", "")
EMU_DRYRUN = EMU_MODE in ("1", "simt")


def _enable_emu_dryrun():
    import contextlib
    import ctypes as C
    import subprocess
    import torch
    from sam_audio_amd import hip
    which = "simt" if EMU_MODE == "simt" else "emu"
    poison = which == "simt" and os.environ.get("SAMAUDIO_SIMT_POISON") == "1"   # LDS poisoned before every workgroup
    asan = which == "simt" and os.environ.get("SAMAUDIO_SIMT_ASAN") == "1"   # AddressSanitizer build (oracle/simt/build.sh)
    emu = os.path.join(ROOT, "oracle", f"_{which}", f"libsamaudio_{which}{'_poison' if poison else '_asan' if asan else ''}.so")
    if os.environ.get("SAMAUDIO_SIMT_LIB"):   # an experiment build of the simulator library (oracle/simt/build.sh), as is
        emu = os.path.abspath(os.environ["SAMAUDIO_SIMT_LIB"])
        os.environ["SAMAUDIO_EMU_NOBUILD"] = "1"
    if not os.environ.get("SAMAUDIO_EMU_NOBUILD"):
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", which, "build.sh")]
                              + (["poison"] if poison else ["asan"] if asan else []))
    lib = C.CDLL(emu)
    for name, (res, args) in hip._PROTOS.items():
        if which == "emu" and name.startswith(("samaudio_vit_", "samaudio_t5_", "samaudio_mbert_")):
            continue   # the launcher emulation predates the vision tower; the SIMT simulator build carries it
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    hip._lib = lib
    for kv in filter(None, os.environ.get("SAMAUDIO_DEBUG_FLAGS", "").split(",")):   # as hip.lib() does for the real library
        k, v = kv.split("=")
        lib.samaudio_debug_set_flag(int(k), int(v))
    hip.require_gpu = lambda device, who: None
    hip.current_stream_ptr = lambda: C.c_void_p(0)
    # `x.to(gpu)` is a copy on a real GPU; tests rely on that (a kernel that works in place must not change the CPU
    # original the reference is computed from), so make it a copy here too
    orig_to = torch.Tensor.to

    def to_copy(self, *a, **k):
        r = orig_to(self, *a, **k)
        names_device = "device" in k or any(isinstance(x, (torch.device, str)) for x in a)
        return r.clone() if (r is self and names_device) else r

    torch.Tensor.to = to_copy
    torch.cuda.device = lambda *a, **k: contextlib.nullcontext()
    torch.cuda.current_stream = lambda *a, **k: None

    class _NoStream:   # SAMAudio(streams=2): the row groups still run on two engine contexts from two host threads
        def __init__(self, *a, **k):
            pass

        def wait_stream(self, other):
            pass

    torch.cuda.Stream = _NoStream
    torch.cuda.stream = lambda *a, **k: contextlib.nullcontext()
    torch.cuda.synchronize = lambda *a, **k: None
    from sam_audio_amd import model as _model_mod
    concurrent = _model_mod.SAMAudio._solve_concurrent

    def one_group_after_the_other(self, *a, **k):   # the simulator runs one launch at a time (one kernel body per process)
        keep, self._serial_groups = self._serial_groups, True
        try:
            return concurrent(self, *a, **k)
        finally:
            self._serial_groups = keep

    _model_mod.SAMAudio._solve_concurrent = one_group_after_the_other
    if which == "emu":
        os.environ["SAMAUDIO_NO_FOLD"] = "1"  # the folded cross-attention projection's kernels are not emulated
        from sam_audio_amd import judge
        from tests.torch_text import TorchTextTower
        judge._TextTower = TorchTextTower   # nor is the ModernBERT text tower (the SIMT simulator build carries it)


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
