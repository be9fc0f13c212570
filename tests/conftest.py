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


def pytest_terminal_summary(terminalreporter):
    # MPPO_TEST_JIT=1 (GPU runs): which kernel the robots of this run ended up with
    from backends import JIT_KINDS

    if JIT_KINDS:
        robots = {}
        for dims, kind in JIT_KINDS:
            robots[dims] = kind
        names = ("run-time-sized kernel", "kernel of the library", "kernel compiled at start-up")
        counts = {n: sum(1 for k in robots.values() if k == i) for i, n in enumerate(names)}
        terminalreporter.write_line(f"MPPO_TEST_JIT: {len(robots)} distinct robots opened - " + ", ".join(f"{v} x {k}" for k, v in counts.items()))
        for dims, kind in sorted(robots.items()):
            if kind == 0:
                terminalreporter.write_line(f"  run-time-sized kernel kept for dims {dims}")
