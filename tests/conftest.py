"""pytest configuration: registers the `gpu` marker and makes tests/ helpers importable."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    """The CPU checkers are compiled on demand (gcc only; a few seconds)."""
    from oracle_lib import build_oracle
    build_oracle(ref=False)
    return True


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """How many histogram cells assert_hist_close excused near the 0.01 fast-exit level (display.cl:237-238)."""
    mod = sys.modules.get("test_gpu_parity")
    ex = getattr(mod, "EXCUSED", None) if mod else None
    if ex is not None:
        terminalreporter.write_line("fast-exit-excused histogram cells this session: %d%s"
                                    % (ex["cells"], (" (" + "; ".join(ex["where"][:8]) + ")") if ex["where"] else ""))
