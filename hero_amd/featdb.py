"""Video feature records: the on-disk value format of the reference's video feature database, decoded into the fp32
`[n_frames, 4352]` tensor the collate takes (the data-format row of the batch boundary).

Reference: `VideoFeatLmdb` (data/data.py:52-122) reads one value per video / clip; the writer is
scripts/convert_videodb.py:53-63.  Two flavours:
  * `compress=True` (db name `..._compressed`): `np.savez_compressed(features=...)`, read back with `np.load`;
  * `compress=False`: `msgpack.dumps({"features": ndarray}, use_bin_type=True)` with `msgpack_numpy.patch()` applied,
    read back with `msgpack.loads(raw=False)`.
Stored dtype is float32 or float16 (`_fp16_to_fp32`, data/data.py:27-31); `__getitem__` clips to `max_clip_len` frames.

The key-value store itself (LMDB, `lmdb==0.97` in the reference's image) is a storage engine and outside the hot-path
scope: `VideoFeatReader` takes any object with `get(key: bytes) -> bytes-like` - an `lmdb` read transaction opened with
`buffers=True` is one; a dict is another.

`msgpack_numpy` is an un-vendored, unpinned dependency of the reference (imported at data/data.py:23-24, not in the
Dockerfile's pins) and is not installed here: its published ndarray encoding - a map
`{nd: True, type: dtype.str, kind: b'', shape: [...], data: raw C-order bytes}` (keys are msgpack `bin` under
`use_bin_type=True`, i.e. `bytes` after `loads(raw=False)`; older writers produce `str` keys) - is restated below, both
directions.  No reference-written bytes of that flavour can be produced in this container, so it is checked by round trips
and a hand-assembled record only (parity unpinned for the msgpack flavour).  The npz flavour IS pinned:
tests/golden/case_featdb.npz holds values written by the reference's `dumps_npz` and what the reference's `VideoFeatLmdb`
returns for them (tests/golden/make_golden_featdb.py; tests/test_cpu_featdb.py)."""
import io

import numpy as np
import torch

try:                                   # msgpack is present here and on the GPU box; only the uncompressed flavour needs it
    import msgpack
except ImportError:                    # pragma: no cover
    msgpack = None


def _get(d, key):
    return d[key] if key in d else d[key.encode() if isinstance(key, str) else key.decode()]


def _decode_nd(obj):
    """msgpack object hook: msgpack_numpy's encoding of an ndarray / numpy scalar -> numpy (anything else unchanged)."""
    if not isinstance(obj, dict) or not (b"nd" in obj or "nd" in obj):
        return obj
    descr = _get(obj, "type")
    if isinstance(descr, bytes):
        descr = descr.decode()
    kind = obj.get(b"kind", obj.get("kind", b""))
    if kind in (b"V", "V"):
        raise NotImplementedError("structured dtypes do not occur in HERO's feature records")
    data = _get(obj, "data")
    if _get(obj, "nd"):
        return np.frombuffer(data, dtype=np.dtype(descr)).reshape(_get(obj, "shape"))
    return np.frombuffer(data, dtype=np.dtype(descr))[0]


def _encode_nd(obj):
    """msgpack `default=` hook: the writer side of the same encoding (scripts/convert_videodb.py:59-60)."""
    if isinstance(obj, np.ndarray):
        if obj.dtype.kind == "V":
            raise NotImplementedError("structured dtypes do not occur in HERO's feature records")
        return {b"nd": True, b"type": obj.dtype.str, b"kind": b"", b"shape": list(obj.shape),
                b"data": np.ascontiguousarray(obj).tobytes()}
    if isinstance(obj, (np.bool_, np.number)):
        return {b"nd": False, b"type": obj.dtype.str, b"data": obj.tobytes()}
    raise TypeError("cannot encode %r" % type(obj))


def encode_record(features, compress=True):
    """One database value for `features` ([n_frames, D] float16 / float32), as scripts/convert_videodb.py writes it
    (`dumps_npz(compress=True)` :50-57 / `dumps_msgpack` :59-60)."""
    features = np.asarray(features)
    if compress:
        with io.BytesIO() as w:
            np.savez_compressed(w, features=features, allow_pickle=True)
            return w.getvalue()
    if msgpack is None:
        raise RuntimeError("the uncompressed record flavour needs the msgpack package")
    return msgpack.dumps({"features": features}, use_bin_type=True, default=_encode_nd)


def decode_record(blob, compress=True):
    """Database value -> dict of arrays (`VideoFeatLmdb.get_dump` without the fp16 widening, data/data.py:98-108)."""
    if compress:
        with io.BytesIO(bytes(blob)) as r:
            dump = np.load(r, allow_pickle=True)
            return {k: dump[k] for k in dump.files}       # includes the writer's stray `allow_pickle` entry, as in the reference
    if msgpack is None:
        raise RuntimeError("the uncompressed record flavour needs the msgpack package")
    return msgpack.loads(bytes(blob), raw=False, object_hook=_decode_nd, strict_map_key=False)


def _widen(dump):
    return {k: (a.astype(np.float32) if isinstance(a, np.ndarray) and a.dtype == np.float16 else a) for k, a in dump.items()}


class VideoFeatReader:
    """Read side of `VideoFeatLmdb` (data/data.py:52-122) over any byte store.

    store          object with get(key: bytes) -> bytes-like or None (an lmdb transaction, a dict, ...)
    name2nframe    {video or clip name: frame count} (the reference's id2nframe.json); None = derive it by decoding every
                   record listed under the `__keys__` entry, clipped to max_clip_len (`_compute_nframe`, :79-94)
    pin            return page-locked tensors (what PrefetchLoader / StaticBatchFeeder copy from asynchronously)"""

    def __init__(self, store, name2nframe=None, compress=True, max_clip_len=-1, pin=False):
        self.store, self.compress, self.max_clip_len, self.pin = store, compress, max_clip_len, pin
        self.pad, self.cls_, self.mask = 0, 1, 2                    # data/data.py:62-64
        self.name2nframe = name2nframe if name2nframe is not None else self._compute_nframe()

    def _value(self, name):
        v = self.store.get(name.encode("utf-8"))
        if v is None:
            raise KeyError(name)
        return v

    def _compute_nframe(self):
        import json
        out = {}
        for name in json.loads(bytes(self.store.get(b"__keys__")).decode("utf-8")):
            n = len(decode_record(self._value(name), self.compress)["features"])
            out[name] = self.max_clip_len if n > self.max_clip_len else n      # as written: -1 clips every video to "-1"
        return out

    def get_dump(self, name):
        return _widen(decode_record(self._value(name), self.compress))

    def __getitem__(self, name):
        """float32 [min(n_frames, max_clip_len), D].  The reference's clip is `n if n < max_clip_len else max_clip_len`
        followed by `features[:n]` (:112-121): with the default max_clip_len = -1 that is `features[:-1]` - every
        training configuration sets max_clip_len (100 for TVR, config/train-tvr-8gpu.json:24); kept as is."""
        n = self.name2nframe[name]
        n = n if n < self.max_clip_len else self.max_clip_len
        feat = torch.tensor(decode_record(self._value(name), self.compress)["features"][:n]).float()
        return feat.pin_memory() if self.pin else feat

    def __contains__(self, name):
        return self.store.get(name.encode("utf-8")) is not None
