"""pytest configuration: registers the `gpu` marker, makes the repo root importable and provides the
`be` fixture that runs a parity test on the CPU emulator build (default) and on the HIP library (-m gpu)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    from backends import get_backend

    return get_backend(request.param)
