import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (os.path.join(REPO, "hmer-img2latex_amd"), os.path.join(REPO, "oracle"), REPO,
          os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# measured parity errors (max |err| per check), written at session end so that a green run leaves the margins behind:
# gpurun_out/parity_errors.json on the GPU box (copied to profiles/ when a round's numbers are recorded)
MEASURED = {}


def record(name, value):
    MEASURED[name] = max(float(value), MEASURED.get(name, 0.0))


def pytest_sessionfinish(session, exitstatus):
    if not MEASURED:
        return
    import json
    out = os.path.join(REPO, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(dict(sorted(MEASURED.items())), f, indent=1)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter):
    if MEASURED:
        terminalreporter.write_line("measured parity errors (max |err|):")
        for k, v in sorted(MEASURED.items()):
            terminalreporter.write_line(f"  {k}: {v:.3e}")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
