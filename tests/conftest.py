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


@pytest.fixture(scope="session")
def gpu():
    import torch
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
