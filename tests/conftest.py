import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the real reference built from /root/reference)")


# Strategies greedy / lazy / lazy2: the suite's older tests pin the HASH-CHAIN matcher (reference run with
# ZSTD_c_useRowMatchFinder = ZSTD_ps_disable); the row-hash tests switch the oracle / the device context explicitly.
os.environ.setdefault("ZHIP_ROW_MATCHER", "disable")


@pytest.fixture(autouse=True)
def _oracle_hash_chain_mode_by_default():
    try:
        import ctypes as C
        so = os.path.join(ROOT, "oracle", "libzoracle.so")
        if os.path.exists(so):
            lib = C.CDLL(so)
            lib.zo_set_row_matcher.argtypes = [C.c_int]
            lib.zo_set_row_matcher(0)
    except Exception:
        pass
    yield


def pytest_collection_modifyitems(config, items):
    """a GPU test that does not come back is a failed run, not a stalled one: pytest-timeout (thread method: works while the test sits in a
    HIP call) ends the process after 4 minutes; the whole -m gpu suite takes about one"""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("gpu") is not None and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(240, method="thread"))
