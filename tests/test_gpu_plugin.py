"""-m gpu: block-level plugin boundary (B1).  The REAL reference library (oracle/_ref/libzstd_ref.so, prebuilt from
/root/reference) is the caller: ZSTD_registerSequenceProducer(cctx, zhip_ctx, zhip_sequence_producer), then
ZSTD_compress2 -> our HIP match finder is invoked per block -> the reference's own entropy stage -> ZSTD_decompress.
Mirrors contrib/externalSequenceProducer/main.c:38-49 and the contract in tests/zstreamtest.c:1921-2064."""
import ctypes as C
import os
import numpy as np
import pytest
from _libs import load_oracle, datagen, _buf, ROOT, have_ref, text_like

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_ref(), reason="needs the prebuilt reference library in oracle/_ref")]

ZSTD_c_compressionLevel, ZSTD_c_validateSequences, ZSTD_c_enableSeqProducerFallback, ZSTD_c_maxBlockSize = 100, 1009, 1014, 1015


@pytest.fixture(scope="module")
def env():
    import torch
    import zstd_amd
    assert torch.cuda.is_available()
    Z = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so"))
    Z.ZSTD_createCCtx.restype = C.c_void_p
    Z.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    Z.ZSTD_CCtx_setParameter.restype = C.c_size_t
    Z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    Z.ZSTD_registerSequenceProducer.restype = None
    Z.ZSTD_registerSequenceProducer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    Z.ZSTD_compress2.restype = C.c_size_t
    Z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    Z.ZSTD_decompress.restype = C.c_size_t
    Z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    Z.ZSTD_isError.restype = C.c_uint
    Z.ZSTD_isError.argtypes = [C.c_size_t]
    Z.ZSTD_compressBound.restype = C.c_size_t
    Z.ZSTD_compressBound.argtypes = [C.c_size_t]
    Z.ZSTD_getErrorCode.restype = C.c_int
    Z.ZSTD_getErrorCode.argtypes = [C.c_size_t]
    return Z, zstd_amd, zstd_amd.Context(0, max_units=64), load_oracle()


def compress_with_plugin(Z, zstd_amd, ctx, a, level, max_block=None, prepare=True, fallback=0, state=None):
    cctx = Z.ZSTD_createCCtx()
    Z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, level)
    Z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_validateSequences, 1)
    Z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_enableSeqProducerFallback, fallback)
    if max_block:
        Z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_maxBlockSize, max_block)
    fn = C.cast(zstd_amd.lib().zhip_sequence_producer, C.c_void_p)
    Z.ZSTD_registerSequenceProducer(cctx, state if state is not None else ctx._h, fn)
    if prepare:
        r = zstd_amd.lib().zhip_prepare_sequences(ctx._h, _buf(a), len(a), max_block or 131072, level)
        assert not zstd_amd.lib().zhip_isError(r)
    cap = Z.ZSTD_compressBound(len(a))
    dst = np.zeros(cap, dtype=np.uint8)
    r = Z.ZSTD_compress2(cctx, _buf(dst), cap, _buf(a), len(a))
    Z.ZSTD_freeCCtx(cctx)
    return r, dst


def roundtrip(Z, a, r, dst):
    out = np.zeros(len(a), dtype=np.uint8)
    d = Z.ZSTD_decompress(_buf(out), len(a), _buf(dst), r)
    assert d == len(a) and out.tobytes() == a.tobytes()


def test_plugin_prepared_fixed_blocks_roundtrip(env):
    Z, zstd_amd, ctx, lo = env
    for a in (datagen(lo, 1_000_003, 50, 3), text_like(700_000, 2)):
        r, dst = compress_with_plugin(Z, zstd_amd, ctx, a, 1, max_block=65536, prepare=True)
        assert not Z.ZSTD_isError(r)
        roundtrip(Z, a, r, dst)
        assert r < len(a) * 0.6


def test_plugin_unprepared_default_partition_roundtrip(env):
    # default partition = 128 KB then 92 KB blocks (zstd_compress.c:4494-4518): served by per-block launches
    Z, zstd_amd, ctx, lo = env
    a = datagen(lo, 600_000, 50, 4)
    r, dst = compress_with_plugin(Z, zstd_amd, ctx, a, 1, prepare=False)
    assert not Z.ZSTD_isError(r)
    roundtrip(Z, a, r, dst)


def test_plugin_single_unit_sequences_equal_internal_parser(env):
    # for one <=128 KB unit the plugin's parse is the reference's own parse (O2 parity); frame bytes may still differ
    # from the internal path because the ingestion path treats repcodes differently -> compare sizes loosely
    Z, zstd_amd, ctx, lo = env
    a = datagen(lo, 131072, 50, 6)
    r, dst = compress_with_plugin(Z, zstd_amd, ctx, a, 1, prepare=True)
    assert not Z.ZSTD_isError(r)
    roundtrip(Z, a, r, dst)
    ref = ctx.compress(a, level=1)
    assert abs(int(r) - len(ref)) < 200


def test_plugin_failure_maps_to_error_or_fallback(env):
    # a producer state that is not a context -> ZHIP_SEQUENCE_PRODUCER_ERROR -> sequenceProducer_failed (106) without
    # fallback, transparent CPU parse with fallback (zstd_compress.c:3338-3356)
    Z, zstd_amd, ctx, lo = env
    a = datagen(lo, 200_000, 50, 8)
    r, dst = compress_with_plugin(Z, zstd_amd, ctx, a, 1, prepare=False, fallback=0, state=C.c_void_p(0))
    assert Z.ZSTD_isError(r) and Z.ZSTD_getErrorCode(r) == 106
    r, dst = compress_with_plugin(Z, zstd_amd, ctx, a, 1, prepare=False, fallback=1, state=C.c_void_p(0))
    assert not Z.ZSTD_isError(r)
    roundtrip(Z, a, r, dst)


@pytest.mark.parametrize("level", [3, 5, 7])
def test_plugin_higher_levels_roundtrip_and_compress_better(env, level):
    """dfast and the hash-chain parsers behind ZSTD_registerSequenceProducer: valid frames, and a better ratio than level 1"""
    Z, zstd_amd, ctx, lo = env
    a = text_like(600_000, 5)
    r1, _ = compress_with_plugin(Z, zstd_amd, ctx, a, 1, max_block=65536, prepare=True)
    r, dst = compress_with_plugin(Z, zstd_amd, ctx, a, level, max_block=65536, prepare=True)
    assert not Z.ZSTD_isError(r)
    roundtrip(Z, a, r, dst)
    assert r < r1


def test_plugin_reused_buffer_with_new_content_is_reparsed(env):
    """the producer's host cache is keyed by address range — a caller that refills the SAME buffer with
    different bytes of the same size must not be served the old parse (zstd does not validate producer sequences by
    default, so a stale parse would silently decode to the wrong data)."""
    Z, zstd_amd, ctx, lo = env
    buf = datagen(lo, 4 * 65536, 50, 11).copy()
    for prepare_first, prepare_second in ((True, False), (True, True), (False, False)):
        r, dst = compress_with_plugin(Z, zstd_amd, ctx, buf, 1, max_block=65536, prepare=prepare_first)
        assert not Z.ZSTD_isError(r)
        roundtrip(Z, buf, r, dst)
        buf[:] = datagen(lo, len(buf), 35, 12 + int(prepare_second))      # same address, same size, new content
        cctx = Z.ZSTD_createCCtx()
        Z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, 1)
        Z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_maxBlockSize, 65536)       # validateSequences stays OFF: nothing else would catch it
        Z.ZSTD_registerSequenceProducer(cctx, ctx._h, C.cast(zstd_amd.lib().zhip_sequence_producer, C.c_void_p))
        if prepare_second:
            assert not zstd_amd.lib().zhip_isError(zstd_amd.lib().zhip_prepare_sequences(ctx._h, _buf(buf), len(buf), 65536, 1))
        cap = Z.ZSTD_compressBound(len(buf))
        dst2 = np.zeros(cap, dtype=np.uint8)
        r2 = Z.ZSTD_compress2(cctx, _buf(dst2), cap, _buf(buf), len(buf))
        Z.ZSTD_freeCCtx(cctx)
        assert not Z.ZSTD_isError(r2)
        roundtrip(Z, buf, r2, dst2)


def test_plugin_many_cctx_on_host_threads_share_one_prepared_context(env):
    """ONE zhip_prepare_sequences, then N reference CCtx on N host threads — each with the producer registered and the SAME zhip context as its
    state — compress shards of whole blocks: prepared blocks are served from the cache without the context lock (the bench's plugin_B1 leg).  Every
    shard must round-trip and equal what one thread alone makes of it; a block the cache does not hold (another size) must not cost the others
    their parse."""
    import threading
    Z, zstd_amd, ctx, lo = env
    blk, nthr = 65536, 4
    a = np.concatenate([datagen(lo, 24 * blk, 50, 21), text_like(8 * blk, 4)])
    per = len(a) // nthr // blk * blk
    shards = [(i * per, per) for i in range(nthr)]
    L = zstd_amd.lib()
    fn = C.cast(L.zhip_sequence_producer, C.c_void_p)
    big = zstd_amd.Context(0, max_units=len(a) // blk + 2)

    def one(o, ln, out, idx):
        cctx = Z.ZSTD_createCCtx()
        Z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, 1)
        Z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_maxBlockSize, blk)
        Z.ZSTD_CCtx_setParameter(cctx, ZSTD_c_validateSequences, 1)
        Z.ZSTD_registerSequenceProducer(cctx, big._h, fn)
        cap = Z.ZSTD_compressBound(ln)
        dst = np.zeros(cap, dtype=np.uint8)
        r = Z.ZSTD_compress2(cctx, _buf(dst), cap, a.ctypes.data + o, ln)
        Z.ZSTD_freeCCtx(cctx)
        out[idx] = (r, dst)

    assert not L.zhip_isError(L.zhip_prepare_sequences(big._h, _buf(a), len(a), blk, 1))
    par = [None] * nthr
    th = [threading.Thread(target=one, args=(o, ln, par, i)) for i, (o, ln) in enumerate(shards)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    # an unprepared block of another size in between (served by its own launch) ...
    odd = datagen(lo, 50000, 50, 33)
    r_odd, d_odd = compress_with_plugin(Z, zstd_amd, big, odd, 1, max_block=50000, prepare=False)
    assert not Z.ZSTD_isError(r_odd)
    roundtrip(Z, odd, r_odd, d_odd)
    # ... must leave the prepared cache in place: the same shards again, one thread, same bytes
    seq = [None] * nthr
    for i, (o, ln) in enumerate(shards):
        one(o, ln, seq, i)
    for (o, ln), (r, dst), (r2, dst2) in zip(shards, par, seq):
        assert not Z.ZSTD_isError(r) and r == r2 and dst[:r].tobytes() == dst2[:r2].tobytes()
        roundtrip(Z, a[o:o + ln], r, dst)
    big.close()
