import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available() and os.path.exists(os.path.join(ROOT, "stereo_vo_amd", "libsvo_hip.so"))
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """Plain `pytest tests` on a box without a HIP device skips the gpu-marked tests instead of failing on the first
    one (the product path has no CPU fallback, so they cannot run there)."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if gpu_items and not _have_gpu():
        skip = pytest.mark.skip(reason="no HIP device / libsvo_hip.so: gpu-marked tests need a real MI355X")
        for it in gpu_items:
            it.add_marker(skip)
