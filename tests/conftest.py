import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "linear_qp_cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu_ctx():
    """A library context on cuda:0.  Fails (does not skip) when the HIP extension cannot be used."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a visible GPU"
    from mrs_uav_trajectory_generation_amd import api
    ctx = api.Context(0)
    ctx.use_torch_stream()
    yield ctx
    ctx.close()
