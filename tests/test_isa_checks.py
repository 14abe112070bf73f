"""Static checks of the compiled gfx950 code (no GPU needed: hipcc cross-compiles).  scripts/isa_lint.py does the work."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


@pytest.fixture(scope="module")
def lint():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "isa_lint.py"), "--json"], timeout=900).decode()
    return json.loads(out)


def test_no_barrier_under_a_narrowed_exec(lint):
    """a workgroup barrier executed under a lane mask means the compiler treats the control flow around it as lane-divergent; how it then
    orders the paths decides whether the kernel works (the GPU-only stalls of rounds 2 and 3, both green on the host emulator).  Branch
    conditions around barriers are made scalar (ZHIP_UNIFORM), leader-only regions are separated by ZHIP_CONVERGE."""
    bad = {k: r["masked_barriers"] for k, r in lint.items() if r["masked_barriers"]}
    assert not bad, bad


def test_decoder_kernel_is_one_piece_of_code(lint):
    assert lint["k_decode"]["calls"] == 0 and lint["k_bf_entropy"]["calls"] == 0


def test_barrier_kernels_are_the_ones_that_passed_on_the_gpu(lint):
    """every kernel that uses s_barrier is pinned (hash of its instruction stream) to the code that last passed `pytest -m gpu` on a real
    MI355X.  A change of such a kernel — intended or a side effect of the compiler's mood — fails HERE until the GPU suite has been re-run
    on it and `python scripts/isa_lint.py --pin` has recorded the new code: no barrier kernel reaches the end of a round untested."""
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "isa_pins.json")))["kernels"]
    now = {k: r["hash"] for k, r in lint.items() if r["barriers"]}
    assert now == pins, {k: (now.get(k), pins.get(k)) for k in set(now) | set(pins) if now.get(k) != pins.get(k)}


def test_kernel_prototypes_are_current():
    """zhip_kernel_decls.h (what the host translation unit launches) is generated from zhip_kernels_*.h"""
    assert subprocess.call([sys.executable, os.path.join(ROOT, "scripts", "gen_kernel_decls.py"), "--check"]) == 0
