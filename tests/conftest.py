import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the shared library is a build artefact (git-ignored): make sure it exists before any test imports the ctypes binding
    # (this is __graft_entry__.build()'s job; doing it here too keeps a fresh checkout + `pytest` self-contained)
    from supir_amd import build as _build
    _build.build(force=False, verbose=False)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
