import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "ranked-list-truncation_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, so a plain `pytest tests`
    # works on the CPU container; `-m gpu` on the GPU box runs them for real.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_DURATIONS = []


def pytest_runtest_logreport(report):
    if report.when == "call" and "gpu" in report.keywords:
        _DURATIONS.append((report.duration, report.nodeid))


def pytest_terminal_summary(terminalreporter):
    """Seconds per GPU test, longest first, and their sum against the driver's 900 s step limit - so that the next
    overrun is visible before it is a timeout."""
    if not _DURATIONS:
        return
    total = sum(d for d, _ in _DURATIONS)
    terminalreporter.write_sep("-", f"GPU tests: {total:.0f} s in {len(_DURATIONS)} tests (driver limit 900 s)")
    for d, nodeid in sorted(_DURATIONS, reverse=True)[:25]:
        terminalreporter.write_line(f"{d:7.1f} s  {nodeid.split('::', 1)[-1]}")
