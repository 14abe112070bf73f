"""Product stage 1 + stage 2 kernels on the host SIMT emulator: complete frames vs the oracle, byte for byte."""
import numpy as np
import pytest
from _libs import load_oracle, load_emu, corpus_cases, emu_compress_units, _buf, ERR


@pytest.fixture(scope="module")
def libs():
    return load_oracle(), load_emu()


def oracle_unit(lo, a, level):
    cap = lo.zo_compress_bound(len(a)) + 64
    dst = np.zeros(cap, dtype=np.uint8)
    r = lo.zo_compress_unit(_buf(dst), cap, _buf(a), len(a), level)
    assert r != ERR
    return dst[:r].tobytes()


def check(lo, le, cases, level):
    frames = emu_compress_units(le, lo, [c[1] for c in cases], level)
    for (name, a), f in zip(cases, frames):
        o = oracle_unit(lo, a, level)
        if f != o:
            k = next((i for i in range(min(len(f), len(o))) if f[i] != o[i]), min(len(f), len(o)))
            raise AssertionError(f"{name} L{level}: got {len(f)} B, want {len(o)} B, first diff at {k}: "
                                 f"{f[max(0,k-4):k+8].hex()} vs {o[max(0,k-4):k+8].hex()}")


def test_frames_128k(libs):
    lo, le = libs
    check(lo, le, list(corpus_cases(lo, sizes=(131072,), seeds=(0,))), 1)


def test_frames_small_and_ragged(libs):
    lo, le = libs
    cases = []
    for n in (0, 1, 6, 7, 8, 13, 40, 63, 64, 65, 100, 255, 256, 257, 300, 1000, 1023, 1024, 1025, 4096, 16383, 16384, 16385, 70000):
        cases += list(corpus_cases(lo, sizes=(n,), seeds=(1,)))
    check(lo, le, cases, 1)


def test_frames_level3_dfast(libs):
    lo, le = libs
    cases = []
    for n in (9, 10, 100, 1000, 20000, 131072):
        cases += list(corpus_cases(lo, sizes=(n,), seeds=(6,)))
    check(lo, le, cases, 3)


def test_frames_negative_level_and_level2(libs):
    lo, le = libs
    for level in (-1, 2):
        cases = []
        for n in (5000, 50000, 131072):
            cases += list(corpus_cases(lo, sizes=(n,), seeds=(2,)))
        from _libs import make_units
        cases = [c for c in cases if make_units(lo, [len(c[1])], level)["strategy"][0] == 1]
        check(lo, le, cases, level)


@pytest.mark.parametrize("level", [5, 6, 7, 10])
def test_frames_hashchain_levels(libs, level):
    """greedy / lazy / lazy2 frames (cost-based FSE table selection from lazy upwards) vs the oracle"""
    lo, le = libs
    import ctypes as C
    def strat(nn):
        cp = (C.c_uint * 7)()
        return cp[6] if lo.zo_get_cparams(level, nn, cp) == 0 else -1            # small units at level 10 are btlazy2: not ours
    cases = []
    for n in (9, 10, 100, 1000, 20000, 131072):
        cases += list(corpus_cases(lo, sizes=(n,), seeds=(level,)))
    cases = [c for c in cases if 3 <= strat(len(c[1])) <= 5]
    assert cases
    check(lo, le, cases, level)


def test_frames_with_content_checksum(libs):
    """ZSTD_c_checksumFlag: k_xxh64 + the checksum epilogue, vs the oracle's XXH64 restatement (pinned to the reference)"""
    import ctypes as C
    lo, le = libs
    lo.zo_frame_add_checksum.restype = C.c_size_t
    lo.zo_frame_add_checksum.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    cases = []
    for n in (0, 1, 3, 4, 7, 8, 12, 31, 32, 33, 63, 64, 100, 129, 1000, 4099, 70001, 131072):
        cases += list(corpus_cases(lo, sizes=(n,), seeds=(9,)))[:5]
    frames = emu_compress_units(le, lo, [c[1] for c in cases], 1, checksum=True)
    for (name, a), f in zip(cases, frames):
        cap = lo.zo_compress_bound(len(a)) + 64
        dst = np.zeros(cap, dtype=np.uint8)
        r = lo.zo_compress_unit(_buf(dst), cap, _buf(a), len(a), 1)
        r = lo.zo_frame_add_checksum(_buf(dst), r, _buf(a), len(a))
        assert f == dst[:r].tobytes(), name


def _unit_with(nseq, litrun, seed):
    """a unit of about `nseq` sequences with literal runs of about `litrun` skewed bytes between short repeats of an earlier stretch"""
    rng = np.random.default_rng(seed)
    out = [(rng.geometric(0.25, size=64) % 61 + 32).astype(np.uint8)]
    n = 64
    for _ in range(nseq):
        out.append((rng.geometric(0.25, size=litrun + int(rng.integers(0, 3))) % 61 + 32).astype(np.uint8))
        n += len(out[-1])
        flat = np.concatenate(out)
        o = int(rng.integers(8, min(n - 8, 4000)))
        ml = int(rng.integers(5, 24))
        out = [flat, flat[n - o:n - o + ml].copy()]
        n += len(out[-1])
    return np.concatenate(out)[:131072]


def test_staged_packers_at_their_boundaries(libs):
    """round 6: huff0 streams go through LDS images in chunks of 64 x 16 symbols per stream, the sequence bitstream in tiles of 2 x 256 sequences, both carrying
    their partial last word into the next chunk / tile, and a stream's first and last word leave byte by byte.  Units whose sequence count walks across the
    tile size and whose literal count walks across four chunks (and across the 1 KB / 16 KB header sizes), against the oracle byte for byte."""
    lo, le = libs
    cases = []
    for nseq in (250, 255, 256, 257, 510, 511, 512, 513, 514, 767, 768, 769, 1023, 1024, 1025, 1030):
        cases.append((f"seq{nseq}", _unit_with(nseq, 3, nseq)))
    for litrun in (1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33):               # 512 runs: literal counts around 4 x 1 024 at runs of 8, around 16 384 at runs of 32
        cases.append((f"lit{litrun}", _unit_with(512, litrun, 7000 + litrun)))
    check(lo, le, cases, 1)
    check(lo, le, cases[::3], 3)


def _unit_dominant(nseq, seed, rare=0.02):
    """sequences that are nearly all alike (one literal length, one match length, a repeated offset) with a few strays: every FSE table has one dominant
    symbol, i.e. states that forget slowly — the slices of the parallel state chains enter wrong and are redone"""
    rng = np.random.default_rng(seed)
    base = rng.integers(97, 123, size=4096, dtype=np.uint8)
    out = [base]
    n = len(base)
    for _ in range(nseq):
        stray = rng.random() < rare
        ll = int(rng.integers(1, 40)) if stray else 2
        ml = int(rng.integers(4, 60)) if stray else 6
        o = int(rng.integers(64, 4000)) if stray else 1024
        out.append(rng.integers(33, 64, size=ll, dtype=np.uint8))
        n += ll
        flat = np.concatenate(out)
        out = [flat, flat[n - o:n - o + ml].copy()]
        n += ml
        if n >= 131072:
            break
    return np.concatenate(out)[:131072]


def test_state_chains_with_a_dominant_symbol(libs):
    """round 6: a slice of an FSE state chain writes its records on its FIRST walk (the codes put aside) and is redone from the right state only as far as the two
    trajectories differ; tables with one dominant symbol make most slices enter wrong"""
    lo, le = libs
    cases = [(f"dom{k}", _unit_dominant(k, 40 + k)) for k in (600, 2000, 6000, 16000)]
    cases += [(f"dom_rare{k}", _unit_dominant(9000, 50 + k, rare=1.0 / k)) for k in (8, 200, 3000)]
    check(lo, le, cases, 1)
    check(lo, le, cases[1:4], 5)
