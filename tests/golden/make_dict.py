#!/usr/bin/env python3
"""Regenerates tests/golden/github_like_110k.zdict with the REAL reference's ZDICT_trainFromBuffer (oracle/_ref, built from
/root/reference): a ZDICT-format dictionary (magic, dictID, Huffman + 3 FSE tables, repcodes, content) trained on 4 000
synthetic GitHub-user-shaped JSON records (zstd_amd/workloads.py, seed 99) — the fixture behind the dictionary tests and
bench.py --workload records where /root/reference does not exist.  Run from the repo root: python tests/golden/make_dict.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import load_ref, _buf, ERR
from zstd_amd import workloads as W

lr = load_ref()
lr.zref_train_dict.restype = C.c_size_t
lr.zref_train_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint]
flat, offs = W.github_like_records(4000, seed=99)
sizes = (C.c_size_t * 4000)(*[int(offs[i + 1] - offs[i]) for i in range(4000)])
buf = np.zeros(112640, dtype=np.uint8)           # the CLI's default maxDictSize (programs/zstdcli.c:82)
r = lr.zref_train_dict(_buf(buf), len(buf), _buf(flat), sizes, 4000)
assert r != ERR
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "github_like_110k.zdict")
buf[:r].tofile(path)
print(r, "bytes ->", path)
