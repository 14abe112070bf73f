"""Multi-GPU sharding of independent units: contiguous unit ranges per rank, no data-path
collective, host-side ordered gather of the variable-length results.  `torch.distributed` is used only to move the
finished byte strings to rank 0 (gloo on CPU in the tests, nccl=RCCL process groups also provide gather_object)."""
import numpy as np

UNIT = 131072


def unit_range(n_units, world, rank):
    """units [lo, hi) owned by `rank` (same rule as the reference's job split: contiguous, near-equal)"""
    lo = n_units * rank // world
    hi = n_units * (rank + 1) // world
    return lo, hi


def byte_range(n_bytes, unit_size, world, rank):
    n_units = max(1, -(-n_bytes // unit_size))
    lo, hi = unit_range(n_units, world, rank)
    return lo * unit_size, min(n_bytes, hi * unit_size)


def compress_sharded(data, compress_fn, dist=None, level=1, unit_size=UNIT):
    """Every rank holds the whole host buffer `data` (np.uint8) and compresses only its unit range with
    compress_fn(bytes_like, level, unit_size) -> (frames: bytes, sizes: np.ndarray).
    Rank 0 returns (stream, sizes) = frames of all ranks in unit order; other ranks return (None, None)."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    b0, b1 = byte_range(len(data), unit_size, world, rank)
    n_units = max(1, -(-len(data) // unit_size))                 # an empty input is ONE (empty) unit = one empty frame
    u0, u1 = unit_range(n_units, world, rank)
    if u1 > u0:
        frames, sizes = compress_fn(data[b0:b1], level, unit_size)
    else:
        frames, sizes = b"", np.zeros(0, dtype=np.uint64)        # more ranks than units
    if dist is None or world == 1:
        return frames, sizes
    parts = [None] * world if rank == 0 else None
    dist.gather_object((frames, np.asarray(sizes, dtype=np.uint64)), parts, dst=0)
    if rank != 0:
        return None, None
    return b"".join(p[0] for p in parts), np.concatenate([p[1] for p in parts])


def decompress_sharded(stream, decompress_fn, dist=None, find_frames_fn=None):
    """Every rank holds the whole host stream of concatenated frames; frames are independent work items, so rank r decodes the
    contiguous frame range unit_range(nFrames, world, r) with decompress_fn(bytes_like) -> bytes and rank 0 concatenates the
    contents in frame order (no data-path collective).  find_frames_fn(stream) -> dict with 'src_off', 'src_size' (uint64
    arrays) — zstd_amd.find_frames by default.  Rank 0 returns the content; other ranks None."""
    if find_frames_fn is None:
        from zstd_amd import find_frames as find_frames_fn
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    fr = find_frames_fn(stream)
    n = len(fr["src_off"])
    lo, hi = unit_range(n, world, rank)
    if hi > lo:
        b0 = int(fr["src_off"][lo]); b1 = int(fr["src_off"][hi - 1] + fr["src_size"][hi - 1])
        part = decompress_fn(stream[b0:b1])
    else:
        part = b""
    if dist is None or world == 1:
        return part
    parts = [None] * world if rank == 0 else None
    dist.gather_object(part, parts, dst=0)
    return b"".join(parts) if rank == 0 else None
