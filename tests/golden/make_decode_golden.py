"""Generates tests/golden/decode_v1.json — run HERE (needs /root/reference and oracle/_ref), commit the output.

Two kinds of vectors for the DECODE path:
  * the reference's own fixtures: tests/golden-decompression/*.zst (must decode; sha256 of the output taken from the real
    reference) and tests/golden-decompression-errors/*.zst (must be rejected) — stored zlib+base64, they are test DATA;
  * frames made by the real reference (ZSTD_compress2 whole-buffer frames: multi-block frames with repeat modes, high
    levels with the optimal parser, negative levels with raw literals, checksummed frames, dictionary frames) from seeded
    inputs, with the sha256 of what ZSTD_decompress returns for them.
"""
import base64, ctypes as C, glob, hashlib, json, os, sys, zlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import load_oracle, load_ref, _buf, ERR, datagen, text_like

REF = "/root/reference/tests"


def pack(b):
    return base64.b64encode(zlib.compress(bytes(b), 9)).decode()


def main():
    lo, lr = load_oracle(), load_ref()
    out = {"fixtures": [], "errors": [], "frames": []}
    for f in sorted(glob.glob(REF + "/golden-decompression/*.zst")):
        b = open(f, "rb").read()
        src = np.frombuffer(b, dtype=np.uint8)
        dst = np.empty(1 << 20, dtype=np.uint8)
        n = r = lr.zref_decompress(_buf(dst), len(dst), _buf(src), len(src))
        assert r != ERR, f
        out["fixtures"].append({"name": os.path.basename(f), "zst": pack(b), "size": n, "sha256": hashlib.sha256(dst[:n].tobytes()).hexdigest()})
    for f in sorted(glob.glob(REF + "/golden-decompression-errors/*.zst")):
        b = open(f, "rb").read()
        src = np.frombuffer(b, dtype=np.uint8)
        dst = np.empty(1 << 20, dtype=np.uint8)
        assert lr.zref_decompress(_buf(dst), len(dst), _buf(src), len(src)) == ERR, f
        out["errors"].append({"name": os.path.basename(f), "zst": pack(b)})
    cases = []
    for seed, (kind, n, level) in enumerate([("P80", 300000, 1), ("text", 280000, 3), ("text", 100000, 9), ("P50", 60000, 19),
                                             ("text", 70000, 19), ("P95", 262144, -5), ("lowent", 140000, 5), ("P50", 1000, 3),
                                             ("text", 40, 1), ("skew", 30000, 7), ("P95", 500000, 12), ("text", 50000, 16)]):
        if kind == "text":
            a = text_like(n, seed)
        elif kind == "lowent":
            a = np.random.default_rng(seed).integers(0, 4, size=n, dtype=np.uint8)
        elif kind == "skew":
            a = (np.random.default_rng(seed).geometric(0.2, size=n) % 256).astype(np.uint8)
        else:
            a = datagen(lo, n, int(kind[1:]), seed)
        cap = int(lr.zref_compress_bound(n))
        dst = np.empty(cap, dtype=np.uint8)
        r = lr.zref_compress_frame(level, _buf(a), n, _buf(dst), cap)
        assert r != ERR
        cases.append({"name": f"{kind}_n{n}_L{level}", "zst": pack(dst[:r].tobytes()), "size": n, "sha256": hashlib.sha256(a.tobytes()).hexdigest()})
    out["frames"] = cases
    json.dump(out, open(os.path.join(HERE, "decode_v1.json"), "w"), indent=0)
    print("wrote", len(out["fixtures"]), "fixtures,", len(out["errors"]), "error fixtures,", len(cases), "frames,",
          os.path.getsize(os.path.join(HERE, "decode_v1.json")), "bytes")


if __name__ == "__main__":
    main()
