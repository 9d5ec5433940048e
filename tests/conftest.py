"""pytest configuration: `gpu` marker + import paths.

`-m "not gpu"` : oracle vs golden fixtures, host logic, C-ABI symbol table (no GPU needed).
`-m gpu`       : parity tests proper — HIP path (through the C ABI) vs the oracle on the same seeded inputs.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "carla-ppo_amd")          # drop-in root: provides `vae.models`, `ppo`, `utils`, `mi355`
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_kept_device_tensors():
    yield
    try:
        import hip_helpers
        if hip_helpers._KEEP:
            hip_helpers.keep_reset()
    except Exception:
        pass
