"""CPU checks of design prototypes for work that is not built yet — no product code involved."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_symbolic_repcode_scan_matches_the_sequential_rule():
    """the block-parallel decoder's repcode scheme (offsets symbolic per block, history maps composed by a scan) against the
    sequential decoder rule on the oracle's sequences of real inputs, at several block sizes"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "repcode_scan_prototype.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
