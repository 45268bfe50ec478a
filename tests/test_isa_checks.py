"""Checks on the generated gfx950 ISA that need no GPU (hipcc cross-compiles here)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_read_of_an_asm_loaded_register_before_its_wait():
    """pgemm's BatchNorm-backward epilogue issues its global loads through inline asm and waits for them by hand (a compiler-visible load would make the
    K loop wait for the LDS-DMA ring, csrc/pgemm.hip).  The compiler may copy such a register before the wait; the ISA must not contain such a read."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_pending_asm_loads.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
