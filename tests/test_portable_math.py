"""CPU: the pinned math primitives (include/fosphor_portable_math.h) via oracle/pm_check."""
import os
import subprocess

from oracle_lib import ORACLE_DIR


def test_pm_check_quick(oracle_built):
    out = subprocess.run([os.path.join(ORACLE_DIR, "pm_check"), "quick"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PM_CHECK OK" in out.stdout
