import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    """The native pieces are git-ignored build products: build them in-tree when a fresh checkout lacks
    them (hipcc cross-compiles gfx950 without a GPU).  Building is not a fallback: the tests still fail
    if the HIP library cannot be built or loaded."""
    lib = os.path.join(ROOT, "isonclust2_amd", "libisonclust2_hip.so")
    cli = os.environ.get("IOC_CLI") or os.path.join(ROOT, "isonclust2_amd", "bin", "isONclust2-hip")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(cli) and os.path.exists(orc)):
        import __graft_entry__ as g
        g.build()
    # torch ships its own HIP runtime; in a process that uses both, torch has to initialise first
    # (bench.py does the same), so do it once here on a GPU box
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass


def pytest_sessionfinish(session, exitstatus):
    """`cluster` hands its jobs to resident worker processes (csrc/cli/main.cpp, "serve"): end the idle ones this session
    started instead of leaving them on the card for their idle time."""
    import subprocess
    cli = os.environ.get("IOC_CLI") or os.path.join(ROOT, "isonclust2_amd", "bin", "isONclust2-hip")
    if os.path.exists(cli):
        try:
            subprocess.run([cli, "serve", "stop"], capture_output=True, timeout=60)
        except Exception:
            pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")) as f:
        return json.load(f)
