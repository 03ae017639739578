import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is a build artefact (git-ignored): compile it once if this checkout has not built it yet
    from mm2d3d_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with -m gpu; without a GPU they are skipped, never run on a fallback.
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _default_formats(request):
    """GPU tests start from, and leave behind, the default storage formats: IEEE fp16 2D maps (the reference's precision: 16),
    fp32 sparse rows - whatever a test (or a failing test) selected."""
    yield
    if "gpu" in request.keywords:
        from mm2d3d_amd import nn2d, scn

        nn2d.set_precision(nn2d.DEFAULT_PRECISION)
        scn.set_activation_dtype(__import__("torch").float32)


@pytest.fixture(params=["fp16", "bf16"])
def half2d(request):
    """Both 16-bit builds of the dense 2D kernels (csrc/h16.h): the test body runs once per storage format; yields the torch dtype."""
    import torch

    from mm2d3d_amd import nn2d

    nn2d.set_precision(request.param)
    yield torch.float16 if request.param == "fp16" else torch.bfloat16
    nn2d.set_precision(nn2d.DEFAULT_PRECISION)


@pytest.fixture
def bf16_mode():
    """bfloat16 2D maps (tests whose bounds were calibrated for that format; the fp16 default has its own tests)."""
    from mm2d3d_amd import nn2d

    nn2d.set_precision("bf16")
    yield
    nn2d.set_precision(nn2d.DEFAULT_PRECISION)
