import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """Every iteration-count difference against the oracle that a test declared (or a discovery run recorded)."""
    try:
        import cases
    except Exception:
        return
    if cases.COUNT_DRIFT:
        terminalreporter.write_sep("-", "iteration-count differences against the oracle (declared cases)")
        for label, its, ref in cases.COUNT_DRIFT:
            terminalreporter.write_line("  %d vs oracle %d : %s" % (its, ref, label))
    else:
        terminalreporter.write_line("iteration counts: every compared case identical to the oracle's")


def pytest_sessionfinish(session, exitstatus):
    """GENEO_TEST_COUNT_DRIFT=record is a discovery aid: every mismatch it let through turns the session red."""
    try:
        import cases
    except Exception:
        return
    if cases.RECORD_MODE_HITS and session.exitstatus == 0:
        session.exitstatus = 1
