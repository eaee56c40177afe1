"""pytest config: registers the `gpu` marker and puts the in-tree package
(`esm-efficient_amd/esme`) and the repo root (for `oracle`) on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no HIP device is visible and the
    run was not explicitly restricted to `-m gpu`."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no HIP device visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
