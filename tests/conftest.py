import os
import sys
from pathlib import Path

import pytest

HOST_FORMS = "CUSRL_HOST_FORMS"


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_setup(item):
    """The host-logic tests and the gloo workers exercise hook bookkeeping in processes without a GPU: THEY — the tests that are
    not marked ``gpu`` — opt in to the hooks' torch-op (host) forms, which the product refuses otherwise
    (cusrl_amd/utils/misc.py host_form).  A ``gpu`` test runs with the gate shut, exactly like a user's process: a hook that
    is handed a CPU tensor there raises instead of quietly taking the reference's torch-op chain
    (tests/test_auxiliary_rewards.py::test_host_forms_are_refused_in_gpu_tests).  Set before any fixture of the test is built;
    the workers the CPU tests spawn inherit it."""
    if item.get_closest_marker("gpu") is None:
        os.environ[HOST_FORMS] = "1"
    else:
        os.environ.pop(HOST_FORMS, None)

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
        return cache[name]

    return load


_GRADIENT_PARITY: list[dict] = []


@pytest.fixture(scope="session")
def gradient_parity():
    """``check(label, candidate, reference, bound)``: assert that a gradient tensor is within ``bound`` of its reference in
    units of the reference's largest entry (``oracle.gradient_error``) and record the achieved error; the session writes
    every record to ``gpurun_out/gradient_parity.json`` (copied to ``profiles/`` as the evidence behind the bounds)."""
    import oracle

    def check(label, candidate, reference, bound):
        achieved = oracle.gradient_error(candidate, reference)
        _GRADIENT_PARITY.append({"check": label, "achieved": achieved, "bound": bound})
        assert achieved <= bound, f"{label}: gradient error {achieved:.3e} of the largest entry exceeds {bound:.0e}"
        return achieved

    return check


def pytest_sessionfinish(session, exitstatus):
    if not _GRADIENT_PARITY:
        return
    import json

    worst: dict[str, dict] = {}
    for record in _GRADIENT_PARITY:
        family = record["check"].split("[")[0]
        if family not in worst or record["achieved"] > worst[family]["achieved"]:
            worst[family] = record
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "gradient_parity.json").write_text(json.dumps(
        {"measure": "max |candidate - reference| / max |reference| per gradient tensor", "worst_per_family": worst,
         "records": _GRADIENT_PARITY}, indent=1))
