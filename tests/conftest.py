import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def tiny_cfg():
    with open(os.path.join(GOLDEN, "tiny_cfg.json")) as f:
        c = json.load(f)
    return c["enc"], c["dec"]


@pytest.fixture(scope="session")
def tiny_state():
    return load_npz("tiny_state.npz")


@pytest.fixture(scope="session")
def tiny_train():
    return load_npz("tiny_train.npz")


@pytest.fixture(scope="session")
def tiny_eval():
    return load_npz("tiny_eval.npz")


@pytest.fixture(scope="session")
def tiny_decode():
    return load_npz("tiny_decode.npz")


@pytest.fixture(scope="session")
def utils_golden():
    return load_npz("utils.npz")


def batch_from_golden(g, dec_key="in::dec_input_ids", with_labels=True):
    b = {k[4:]: v.clone() for k, v in g.items() if k.startswith("in::")}
    b["dec_input_ids"] = g[dec_key].clone()
    if not with_labels:
        b["dec_labels"] = None
    return b
