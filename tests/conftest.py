import ast
import importlib
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(name=""):
    """The package directory starts with a digit -> import through importlib."""
    return importlib.import_module("1xgpt_amd" + ("." + name if name else ""))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg_kwargs = ast.literal_eval(str(z["cfg"]))
    cfg = pkg("config").GenieConfig(**cfg_kwargs)
    sd = pkg("synthetic").make_state_dict(cfg, seed=int(z["weight_seed"]), law="conditioned")
    return z, cfg, sd


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def record_measure(key, value):
    """Append `key value` to the file named by GENIE_TEST_RECORD (if set): the measured deltas behind the tolerance bars of the bf16-vs-
    reference tests, so that a bar can be kept at ~10x what is measured (tools/gpu_run.sh suite collects them into profiles/)."""
    path = os.environ.get("GENIE_TEST_RECORD")
    if path:
        with open(path, "a") as f:
            f.write(f"{key} {value:.6e}\n")
