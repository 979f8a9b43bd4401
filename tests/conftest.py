import json
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
    with open(os.path.join(ROOT, "tests", "golden", "golden_v1.json")) as f:
        return json.load(f)


def rel_err(a, b):
    """max|a-b| / max|b| in float64 (the parity metric used throughout the tests)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def check_digest(t, d, tol):
    """Compare a tensor with a golden digest (sum / abs_sum / l2 / strided samples)."""
    import torch
    from oracle.sif_oracle import digest
    got = digest(t, len(d["samples"]))
    assert got["shape"] == d["shape"], (got["shape"], d["shape"])
    scale = max(d["l2"], 1e-30)
    n = max(1, int(torch.tensor(d["shape"]).prod())) if d["shape"] else 1
    assert abs(got["l2"] - d["l2"]) <= tol * scale, ("l2", got["l2"], d["l2"])
    assert abs(got["abs_sum"] - d["abs_sum"]) <= tol * max(d["abs_sum"], 1e-30), ("abs_sum", got["abs_sum"], d["abs_sum"])
    # plain sum suffers cancellation: bound it by tol * abs_sum
    assert abs(got["sum"] - d["sum"]) <= tol * max(d["abs_sum"], 1e-30), ("sum", got["sum"], d["sum"])
    smax = max(abs(v) for v in d["samples"]) or 1.0
    for g, e in zip(got["samples"], d["samples"]):
        assert abs(g - e) <= tol * smax, ("sample", g, e)
