import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "robotics-rl-srl_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lumped_kuka: GPU test of the lumped-gripper Kuka kernels (the oracle stays in its lumped mode)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # The rounds 1-2 lumped-gripper kernels (cfg.kuka_model = LUMPED) are no longer the default model, but the ABI exposes them and
    # bench.py times them (secondary.kuka_lumped_model): their GPU tests run in the default suite (SRLHIP_SKIP_LUMPED=1 leaves them out).
    if os.environ.get("SRLHIP_SKIP_LUMPED"):
        skip_lumped = pytest.mark.skip(reason="lumped-gripper kernels: SRLHIP_SKIP_LUMPED is set")
        for item in items:
            if "lumped_kuka" in item.keywords:
                item.add_marker(skip_lumped)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _kuka_oracle_model(request):
    """The Kuka oracle integrates the model the product integrates by default: GPU tests run against the FULL 12-DoF gripper model
    (srlhip_default_config: kuka_model = FULL for KukaButton / Moving / RandButton); a GPU test that steps the rounds 1-2
    lumped-gripper kernels says so with @pytest.mark.lumped_kuka (or calls lumped_kuka() itself).  CPU tests keep the oracle's
    default (lumped) unless they switch it themselves."""
    full = "gpu" in request.keywords and "lumped_kuka" not in request.keywords
    if full:
        from oracle import kuka_clib
        kuka_clib.set_full(True)
    yield
    if full:
        from oracle import kuka_clib
        kuka_clib.set_full(False)
