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
