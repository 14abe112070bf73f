"""bench.py's corpus hooks (round-5 verdict, item 6 ii): $ZHIP_SILESIA / $ZHIP_ENWIK9 name a file and the Silesia / text legs then run on the real corpus and
say `"data": "real"`; unset, the synthetic stand-ins are used and the line says "synthetic".  No GPU: make_workload only needs a torch device to put the bytes on."""
import importlib.util
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location("zhip_bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _file(tmp_path, name, n, seed):
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(200)]
    data = b" ".join(words[int(i)] for i in rng.integers(0, 200, n // 5))[:n]
    p = tmp_path / name
    p.write_bytes(data)
    return str(p), np.frombuffer(data, dtype=np.uint8)


def test_silesia_hook_uses_the_file_and_says_real(bench, tmp_path, monkeypatch):
    import torch
    import zstd_amd
    path, data = _file(tmp_path, "silesia.tar", 300_000, 1)
    monkeypatch.setenv("ZHIP_SILESIA", path)
    host, src, n, wdesc, scaling, tile = bench.make_workload(torch, zstd_amd, torch.device("cpu"), "silesia", 0, 1, 1, 3, 0)
    assert bench.make_workload.data_kind == "real" and "ZHIP_SILESIA" in wdesc and scaling == "weak"
    assert n == 3 * len(data)
    got = src[:n].numpy()
    assert got[: len(data)].tobytes() == data.tobytes()                      # copy 0 starts at offset 0
    s1 = 9973 % len(data)                                                     # copy c starts at offset c * 9973 (units of different copies differ)
    assert got[len(data): 2 * len(data)].tobytes() == np.concatenate([data[s1:], data[:s1]]).tobytes()
    assert host.tobytes() == got.tobytes()                                    # the CPU legs see exactly what the device compresses


def test_enwik9_hook_uses_the_file_and_says_real(bench, tmp_path, monkeypatch):
    import torch
    import zstd_amd
    path, data = _file(tmp_path, "enwik9", 500_000, 2)
    monkeypatch.setenv("ZHIP_ENWIK9", path)
    host, src, n, wdesc, scaling, tile = bench.make_workload(torch, zstd_amd, torch.device("cpu"), "text", 0, 2, 1, 1, 400_000)
    assert bench.make_workload.data_kind == "real" and "ZHIP_ENWIK9" in wdesc
    assert scaling == "strong" and n == 200_000                               # frame-per-shard: the fixed total cut into one shard per rank
    assert src[:n].numpy().tobytes() == data[:n].tobytes()


def test_without_the_hooks_the_stand_ins_are_synthetic(bench, monkeypatch):
    import torch
    import zstd_amd
    monkeypatch.delenv("ZHIP_SILESIA", raising=False); monkeypatch.delenv("ZHIP_ENWIK9", raising=False)
    host, src, n, wdesc, scaling, tile = bench.make_workload(torch, zstd_amd, torch.device("cpu"), "text", 0, 1, 1, 1, 0)
    assert bench.make_workload.data_kind == "synthetic" and "stand-in" in wdesc and n == 1 << 20


def test_a_hook_that_names_no_file_is_an_error_not_a_silent_fallback(bench, monkeypatch):
    import torch
    import zstd_amd
    monkeypatch.setenv("ZHIP_SILESIA", "/nonexistent/silesia.tar")
    with pytest.raises(SystemExit):
        bench.make_workload(torch, zstd_amd, torch.device("cpu"), "silesia", 0, 1, 1, 1, 0)


def test_lorem_workload_is_the_references_generator(bench):
    """SURVEY 8(d)'s text stand-in: LOREM_genBuffer (programs/lorem.h:20) through oracle/_ref — generation only"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so")):
        pytest.skip("oracle/_ref is not built here")
    a = bench.lorem_corpus(100_000, 0)
    b = bench.lorem_corpus(100_000, 0)
    assert a.tobytes() == b.tobytes() and len(set(a.tobytes())) > 20
    text = a.tobytes()[:2000].decode("ascii")
    assert " " in text and text[:5].isalpha() or text[0].isalpha()
