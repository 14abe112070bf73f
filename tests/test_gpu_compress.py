"""-m gpu: full-pipeline parity on a real MI355X through the C ABI: frames are byte-identical to the oracle
(oracle/zoracle.c, pinned to the reference), match the committed golden vectors made with the real reference, and
decode bit-exactly with independent decoders (reference build in oracle/_ref when present, system libzstd)."""
import ctypes as C
import ctypes.util
import hashlib, json, os
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, corpus_cases, datagen, _buf, ERR, ROOT

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "units_v1.json")
GOLD_HC = os.path.join(os.path.dirname(__file__), "golden", "units_v2_hashchain.json")


@pytest.fixture(scope="module")
def env():
    import torch
    import zstd_amd
    assert torch.cuda.is_available(), "needs a GPU"
    return load_oracle(), zstd_amd.Context(0, max_units=640), torch


def oracle_chunks(lo, a, level, unit=131072):
    n = len(a)
    cap = lo.zo_compress_bound(unit) * (n // unit + 1) + 64
    dst = np.zeros(cap, dtype=np.uint8)
    nu = max(1, -(-n // unit))
    sizes = np.zeros(nu, dtype=np.uint64)
    r = lo.zo_compress_chunks(level, unit, _buf(a), n, _buf(dst), cap, _buf(sizes), nu)
    assert r != ERR
    return dst[:r].tobytes(), sizes


def system_decompress(blob, n):
    path = ctypes.util.find_library("zstd") or "libzstd.so.1"
    try:
        L = C.CDLL(path)
    except OSError:
        return None
    L.ZSTD_decompress.restype = C.c_size_t
    L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    out = np.zeros(max(n, 1), dtype=np.uint8)
    src = np.frombuffer(blob, dtype=np.uint8)
    r = L.ZSTD_decompress(_buf(out), n, _buf(src), len(blob))
    assert r == n, f"system libzstd decode failed: {r}"
    return out[:n]


def first_diff(a, b):
    k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
    return f"len {len(a)} vs {len(b)}, first diff at {k}: {a[max(0,k-4):k+8].hex()} vs {b[max(0,k-4):k+8].hex()}"


def test_units_128k_match_oracle_bytes(env):
    lo, ctx, torch = env
    cases = list(corpus_cases(lo, sizes=(131072,), seeds=(0, 1)))
    flat = np.concatenate([c[1] for c in cases])
    got, sizes = ctx.compress(flat, level=1, return_sizes=True)
    want, wsizes = oracle_chunks(lo, flat, 1)
    assert np.array_equal(sizes, wsizes), [(c[0], int(a), int(b)) for c, a, b in zip(cases, sizes, wsizes) if a != b]
    assert got == want, first_diff(got, want)
    dec = system_decompress(got, len(flat))
    if dec is not None:
        assert dec.tobytes() == flat.tobytes()


@pytest.mark.parametrize("level", [1, 2, -1, -7, 3, 4, 5, 6, 8])
def test_ragged_units_match_oracle_bytes(env, level):
    lo, ctx, torch = env
    import zstd_amd
    for n in (0, 1, 6, 7, 8, 13, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 4096, 16383, 16384, 16385, 70000, 131071):
        try:
            cp = zstd_amd.get_cparams(level, n)
        except zstd_amd.ZhipError:
            continue
        if cp[6] not in (1, 2, 3, 4, 5):
            continue
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(3,)):
            got = ctx.compress(a, level=level)
            want, _ = oracle_chunks(lo, a, level)
            assert got == want, (name, level, first_diff(got, want))


@pytest.mark.parametrize("level", [1, 3, 5])
def test_staged_packers_at_their_boundaries(env, level):
    """round 6: the entropy stage packs huff0 streams (chunks of 64 x 16 symbols) and the sequence bitstream (tiles of 2 x 256 sequences) through LDS images that
    carry their partial last word; units whose sequence and literal counts walk across those sizes (tests/test_emu_entropy.py has the generator), byte for byte"""
    lo, ctx, torch = env
    from test_emu_entropy import _unit_with
    cases = [(f"seq{k}", _unit_with(k, 3, k)) for k in (250, 255, 256, 257, 510, 511, 512, 513, 514, 767, 768, 769, 1023, 1024, 1025, 1030, 2047, 2048, 2049)]
    cases += [(f"lit{r}", _unit_with(512, r, 7000 + r)) for r in (1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65)]
    cases += [(f"big{r}", _unit_with(3000, r, 9000 + r)) for r in (5, 21, 40)]            # several tiles and chunks per stream, up to the full unit
    for name, a in cases:
        got = ctx.compress(a, level=level)
        want, _ = oracle_chunks(lo, a, level)
        assert got == want, (name, level, first_diff(got, want))


@pytest.mark.parametrize("level", [1, 3, 5])
def test_state_chains_with_a_dominant_symbol(env, level):
    """round 6: one walk per slice of the FSE state chains, redone only as far as a wrong entering state reaches (tests/test_emu_entropy.py has the generator and says why)"""
    lo, ctx, torch = env
    from test_emu_entropy import _unit_dominant
    cases = [(f"dom{k}", _unit_dominant(k, 40 + k)) for k in (600, 2000, 6000, 16000, 30000)]
    cases += [(f"dom_rare{k}", _unit_dominant(9000, 50 + k, rare=1.0 / k)) for k in (8, 200, 3000)]
    flat = np.concatenate([np.concatenate([c[1], np.zeros(-len(c[1]) % 131072, dtype=np.uint8)]) for c in cases])      # one batch of full units (zero tails)
    got = ctx.compress(flat, level=level)
    want, _ = oracle_chunks(lo, flat, level)
    assert got == want, (level, first_diff(got, want))
    for name, a in cases:
        got = ctx.compress(a, level=level)
        want, _ = oracle_chunks(lo, a, level)
        assert got == want, (name, level, first_diff(got, want))


@pytest.mark.parametrize("level,minseen", [(1, 200), (3, 100), (5, 200), (6, 200), (7, 200)])
def test_golden_vectors_from_the_real_reference(env, level, minseen):
    lo, ctx, torch = env
    path = GOLD if level < 5 else GOLD_HC          # levels 5-7: the reference with the row matcher disabled (hash chain)
    gold = {(g["case"], g["level"]): g for g in json.load(open(path))["units"] if g["level"] == level}
    sizes = sorted({g["n"] for g in gold.values()})
    seen = 0
    for n in sizes:
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(0, 5)):
            g = gold.get((name, level))
            if g is None:
                continue
            assert hashlib.sha256(a.tobytes()).hexdigest() == g["src_sha256"]
            got = ctx.compress(a, level=level)
            assert len(got) == g["csize"] and hashlib.sha256(got).hexdigest() == g["dst_sha256"], name
            seen += 1
    assert seen >= minseen


def test_level3_128k_units_match_oracle_bytes(env):
    """strategy dfast (level 3, the reference's default level): full-size units, frames byte-identical to the oracle"""
    lo, ctx, torch = env
    cases = list(corpus_cases(lo, sizes=(131072,), seeds=(0,)))
    flat = np.concatenate([c[1] for c in cases])
    got, sizes = ctx.compress(flat, level=3, return_sizes=True)
    want, wsizes = oracle_chunks(lo, flat, 3)
    assert np.array_equal(sizes, wsizes), [(c[0], int(a), int(b)) for c, a, b in zip(cases, sizes, wsizes) if a != b]
    assert got == want, first_diff(got, want)


@pytest.mark.parametrize("level", [5, 6, 7, 10])
def test_hashchain_128k_units_match_oracle_bytes(env, level):
    """strategies greedy / lazy / lazy2 (hash chain; levels 5-10 of the <= 128 KB row): frames byte-identical to the oracle"""
    lo, ctx, torch = env
    cases = list(corpus_cases(lo, sizes=(131072,), seeds=(level,)))
    rng = np.random.default_rng(level)
    mixed = np.concatenate([rng.integers(0, 256, size=20000, dtype=np.uint8), datagen(lo, 40000, 60, level),
                            rng.integers(0, 256, size=30000, dtype=np.uint8), datagen(lo, 41072, 30, level + 1)])
    cases.append(("mixed_skip_then_match", mixed))          # lazy-skipping stretches followed by compressible data
    flat = np.concatenate([c[1] for c in cases])
    got, sizes = ctx.compress(flat, level=level, return_sizes=True)
    want, wsizes = oracle_chunks(lo, flat, level)
    assert np.array_equal(sizes, wsizes), [(c[0], int(a), int(b)) for c, a, b in zip(cases, sizes, wsizes) if a != b]
    assert got == want, first_diff(got, want)
    dec = system_decompress(got, len(flat))
    if dec is not None:
        assert dec.tobytes() == flat.tobytes()


def test_level4_mixes_dfast_units_with_a_greedy_tail(env):
    lo, ctx, torch = env
    a = datagen(lo, 2 * 131072 + 9000, 50, 4)
    got = ctx.compress(a, level=4)
    want, _ = oracle_chunks(lo, a, 4)
    assert got == want, first_diff(got, want)


def test_frame_checksum_flag(env):
    """ZSTD_c_checksumFlag: every frame ends with XXH64's low 32 bits; bytes = oracle (+ the real reference when present)"""
    lo, ctx, torch = env
    lo.zo_frame_add_checksum.restype = C.c_size_t
    lo.zo_frame_add_checksum.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    a = np.concatenate([datagen(lo, 131072 * 5 + 777, 50, 8)])
    ctx.set_checksum(True)
    try:
        got, sizes = ctx.compress(a, level=3, return_sizes=True)
    finally:
        ctx.set_checksum(False)
    want = b""
    for off in range(0, len(a), 131072):
        u = a[off:off + 131072]
        dst = np.zeros(len(u) + 700, dtype=np.uint8)
        r = lo.zo_compress_unit(_buf(dst), len(dst), _buf(u), len(u), 3)
        r = lo.zo_frame_add_checksum(_buf(dst), r, _buf(u), len(u))
        want += dst[:r].tobytes()
    assert got == want, first_diff(got, want)
    if have_ref():
        lr = load_ref()
        lr.zref_compress_chunks_checksum.restype = C.c_size_t
        lr.zref_compress_chunks_checksum.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        ref = np.zeros(len(a) + 4096, dtype=np.uint8)
        k = lr.zref_compress_chunks_checksum(3, 131072, _buf(a), len(a), _buf(ref), len(ref), None, 0)
        assert got == ref[:k].tobytes()
    dec = system_decompress(got, len(a))          # the decoder verifies the checksums
    if dec is not None:
        assert dec.tobytes() == a.tobytes()
    assert ctx.compress(a[:1000], level=1) == oracle_chunks(lo, a[:1000], 1)[0]      # flag is off again


@pytest.mark.parametrize("checksum", [False, True])
def test_seekable_container_reads_back_through_the_reference_decoder(env, checksum):
    """frames + seek table (contrib/seekable_format): random-access reads with the reference's ZSTD_seekable_decompress"""
    lo, ctx, torch = env
    if not have_ref():
        pytest.skip("reference build (oracle/_ref) not present")
    lr = load_ref()
    lr.zref_seekable_read.restype = C.c_size_t
    lr.zref_seekable_read.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_ulonglong, C.c_void_p]
    a = datagen(lo, 131072 * 7 + 4321, 50, 12)
    ctx.set_checksum(checksum)
    try:
        blob = ctx.compress_seekable(a, level=1)
    finally:
        ctx.set_checksum(False)
    src = np.frombuffer(blob, dtype=np.uint8)
    assert blob[-4:] == (0x8F92EAB1).to_bytes(4, "little")
    nf = C.c_uint(0)
    rng = np.random.default_rng(1)
    for off, ln in [(0, 100), (131072 - 50, 100), (131072 * 3 + 17, 300000), (len(a) - 77, 77)] + [(int(o), 5000) for o in rng.integers(0, len(a) - 5000, size=6)]:
        out = np.zeros(ln, dtype=np.uint8)
        r = lr.zref_seekable_read(_buf(out), ln, _buf(src), len(src), off, C.byref(nf))
        assert r == ln and out.tobytes() == a[off:off + ln].tobytes(), (off, ln, r)
    assert nf.value == 8
    dec = system_decompress(blob, len(a))          # a plain decoder skips the skippable frame
    if dec is not None:
        assert dec.tobytes() == a.tobytes()


def test_multi_unit_stream_device_api_and_roundtrip(env):
    lo, ctx, torch = env
    import zstd_amd
    n = 131072 * 300 + 4321
    a = datagen(lo, n, 50, 21)
    d = torch.from_numpy(np.concatenate([a, np.zeros(64, np.uint8)])).cuda()
    cap = zstd_amd.compress_bound(n)
    dst = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
    usz = torch.zeros(301, dtype=torch.int32, device="cuda")
    r = ctx.compress_device(dst.data_ptr(), cap, d.data_ptr(), n, 1, 131072, usz.data_ptr())
    got = dst[:r].cpu().numpy().tobytes()
    want, wsizes = oracle_chunks(lo, a, 1)
    assert np.array_equal(usz.cpu().numpy().astype(np.uint64), wsizes)
    assert got == want, first_diff(got, want)
    if have_ref():
        lr = load_ref()
        out = np.zeros(n, dtype=np.uint8)
        src = np.frombuffer(got, dtype=np.uint8)
        assert lr.zref_decompress(_buf(out), n, _buf(src), len(got)) == n
        assert out.tobytes() == a.tobytes()
    dec = system_decompress(got, n)
    if dec is not None:
        assert dec.tobytes() == a.tobytes()


def test_dst_too_small_and_unsupported_level_are_errors(env):
    lo, ctx, torch = env
    import zstd_amd
    a = datagen(lo, 100000, 50, 1)
    L = zstd_amd.lib()
    dst = np.zeros(10, dtype=np.uint8)
    r = L.zhip_compress(ctx._h, _buf(dst), 10, _buf(a), len(a), 1, 131072, None)
    assert L.zhip_isError(r) and b"too small" in L.zhip_getErrorName(r)
    with pytest.raises(zstd_amd.ZhipError):
        ctx.compress(a, level=19)
