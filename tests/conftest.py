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
    config.addinivalue_line("markers", "isolated: the test body runs in a freshly spawned child pytest process (whole-step / "
                                       "decoder stream captures, process groups): a native abort fails ONE test, not the session")


_ORDER = {"test_ops_gpu.py": 0}      # per-kernel parity first: the cheapest, most diagnostic evidence must never be lost


def pytest_collection_modifyitems(config, items):
    """GPU session order: per-kernel tests, then the in-process integration tests, then the isolated (child-process) ones."""
    def key(it):
        f = os.path.basename(str(it.fspath))
        return (2 if it.get_closest_marker("isolated") else _ORDER.get(f, 1))
    items.sort(key=key)              # stable: file / definition order is kept inside a class


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """`@pytest.mark.isolated`: spawn `python -m pytest <nodeid>` (a child process -- never an exec of this GPU-initialised
    one) and assert on its exit code; inside the child (GSTVD_TEST_CHILD=1) the body runs normally."""
    if pyfuncitem.get_closest_marker("isolated") is None or os.environ.get("GSTVD_TEST_CHILD") == "1":
        return None
    import subprocess
    env = dict(os.environ, GSTVD_TEST_CHILD="1", PYTHONFAULTHANDLER="1")
    cmd = [sys.executable, "-m", "pytest", pyfuncitem.nodeid, "-x", "-q", "-p", "no:cacheprovider"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode("utf-8", "replace")
    log = os.environ.get("GSTVD_TEST_CHILD_LOG")
    if log:
        with open(log, "a") as f:
            f.write("==== %s rc=%d\n%s\n" % (pyfuncitem.nodeid, r.returncode, out))
    assert r.returncode == 0, "isolated test %s: child exit code %d\n%s" % (pyfuncitem.nodeid, r.returncode, out[-6000:])
    return True


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
