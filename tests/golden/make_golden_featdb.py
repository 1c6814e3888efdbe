#!/usr/bin/env python3
"""Golden vectors for the video feature RECORD format, produced by the reference's own writer and reader.

Container-only (needs /root/reference).  Imported and run:
    dumps_npz                       scripts/convert_videodb.py:50-57   (the writer of `..._compressed` databases)
    VideoFeatLmdb.__getitem__       data/data.py:110-122               (the reader the dataset calls per video)
    VideoFeatLmdb.get_dump          data/data.py:98-108
    VideoFeatLmdb._compute_nframe   data/data.py:79-94
The LMDB environment is the only thing replaced: `txn` is a dict (its `.get(key)` is what the reader calls).  Packages
the image lacks are stubbed before the import (lmdb, lz4.frame, msgpack_numpy, toolz / cytoolz, horovod.torch, apex) -
none of them is executed on the compressed-record path.  The uncompressed (msgpack + msgpack_numpy) flavour cannot be
produced here (msgpack_numpy is not installed) and is not part of the fixture.

Writes tests/golden/case_featdb.npz:
    rec.<name>        uint8: the database value the reference's writer produced for video <name>
    out.<name>        float32: what VideoFeatLmdb[<name>] returns for it (max_clip_len = 100)
    dump.<name>       float32: get_dump(<name>)['features'] (not clipped, fp16 widened)
    out_default.<name>  the same read with the constructor's default max_clip_len = -1
    nframe            JSON: _compute_nframe() with max_clip_len = 100

Run:  python tests/golden/make_golden_featdb.py
"""
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("HERO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_stubs  # noqa: E402
from make_golden_collate import install_data_stubs  # noqa: E402


def main():
    install_stubs()
    install_data_stubs()
    import types
    tq = types.ModuleType("tqdm")
    tq.tqdm = lambda it=None, **kw: it
    sys.modules.setdefault("tqdm", tq)
    sys.modules["cytoolz"].curry = lambda f: f                # decorates load_npz only (not called here)
    import msgpack_numpy                                       # the stub: patch() is a no-op
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "scripts"))
    from data.data import VideoFeatLmdb
    import convert_videodb

    rng = np.random.default_rng(7)
    vids = {"tvr_a": rng.standard_normal((130, 48)).astype(np.float32),          # longer than max_clip_len
            "tvr_b": rng.standard_normal((17, 48)).astype(np.float16),           # stored as fp16
            "tvr_c": rng.standard_normal((100, 48)).astype(np.float32),          # exactly max_clip_len
            "tvr_d": rng.standard_normal((1, 48)).astype(np.float32)}
    class Txn(dict):                                       # lmdb.Transaction.get(key) (the reference also passes key=...)
        def get(self, key):
            return dict.get(self, key)

    store = Txn({k.encode("utf-8"): convert_videodb.dumps_npz({"features": v}, compress=True) for k, v in vids.items()})
    store[b"__keys__"] = json.dumps(list(vids)).encode("utf-8")

    class _Env:                                            # __del__ closes the environment
        def close(self):
            pass

    def reader(max_clip_len, name2nframe):
        db = object.__new__(VideoFeatLmdb)                 # __init__ opens the LMDB environment: everything else is set by hand
        db.env = _Env()
        db.compress, db.max_clip_len, db.txn = True, max_clip_len, store
        db.name2nframe = name2nframe
        return db

    out = {}
    db = reader(100, None)
    nframe = db._compute_nframe()
    db.name2nframe = {k: len(v) for k, v in vids.items()}                         # id2nframe.json holds the raw counts
    dflt = reader(-1, {k: len(v) for k, v in vids.items()})
    for k in vids:
        out["rec." + k] = np.frombuffer(store[k.encode("utf-8")], dtype=np.uint8)
        out["out." + k] = db[k].numpy()
        out["dump." + k] = db.get_dump(k)["features"]
        out["out_default." + k] = dflt[k].numpy()
    out["nframe"] = np.frombuffer(json.dumps(nframe).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "case_featdb.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items() if k.startswith("out.")}, nframe)


if __name__ == "__main__":
    main()
