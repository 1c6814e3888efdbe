"""hero_amd.featdb: the value format of the reference's video feature database (data/data.py:52-122 reader,
scripts/convert_videodb.py:50-60 writer).  The npz flavour is checked against numpy's own reader / writer (what the
reference calls); the msgpack flavour by round trip and against a hand-assembled msgpack_numpy record."""
import io
import json

import numpy as np
import pytest
import torch

from hero_amd import featdb


def _feats(n, d=4352, dtype=np.float32, seed=0):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(dtype)


def test_npz_records_are_what_the_reference_writer_produces():
    f = _feats(7)
    blob = featdb.encode_record(f, compress=True)
    with io.BytesIO(blob) as r:                                   # VideoFeatLmdb.__getitem__'s own decode (data/data.py:114-117)
        ref = np.load(r, allow_pickle=True)["features"]
    assert np.array_equal(ref, f)
    with io.BytesIO() as w:                                       # dumps_npz(dump, compress=True) (scripts/convert_videodb.py:50-57)
        np.savez_compressed(w, **{"features": f}, allow_pickle=True)
        theirs = w.getvalue()
    assert np.array_equal(featdb.decode_record(theirs, compress=True)["features"], f)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("compress", [True, False])
def test_reader_clips_widens_and_round_trips(compress, dtype):
    vids = {"v_a": _feats(130, 64, dtype, 1), "v_b": _feats(12, 64, dtype, 2), "v_c": _feats(100, 64, dtype, 3)}
    store = {k.encode(): featdb.encode_record(v, compress) for k, v in vids.items()}
    store[b"__keys__"] = json.dumps(list(vids)).encode()
    rd = featdb.VideoFeatReader(store, {k: len(v) for k, v in vids.items()}, compress=compress, max_clip_len=100)
    for k, v in vids.items():
        t = rd[k]
        assert t.dtype == torch.float32 and t.shape == (min(len(v), 100), 64)
        assert torch.equal(t, torch.from_numpy(v[:100].astype(np.float32)))
        d = rd.get_dump(k)["features"]
        assert d.dtype == np.float32 and np.array_equal(d, v.astype(np.float32))       # get_dump does not clip
    assert "v_a" in rd and "nope" not in rd
    with pytest.raises(KeyError):
        rd["nope"]
    # frame counts derived from the records (the reference's fallback when id2nframe.json holds null)
    rd2 = featdb.VideoFeatReader(store, None, compress=compress, max_clip_len=100)
    assert rd2.name2nframe == {"v_a": 100, "v_b": 12, "v_c": 100}
    assert torch.equal(rd2["v_a"], rd["v_a"])


def test_default_max_clip_len_keeps_the_reference_quirk():
    """max_clip_len = -1 (the constructor default) makes `n if n < -1 else -1` = -1: the last frame is dropped
    (data/data.py:112-121).  Every shipped config sets max_clip_len; the behaviour is mirrored, not fixed."""
    f = _feats(9, 16)
    rd = featdb.VideoFeatReader({b"v": featdb.encode_record(f)}, {"v": 9})
    assert rd["v"].shape == (8, 16) and torch.equal(rd["v"], torch.from_numpy(f[:-1]))


def test_msgpack_numpy_wire_format():
    """The uncompressed flavour against a record assembled by hand from msgpack_numpy's published encoding (bin keys under
    use_bin_type=True; str keys from older writers): {nd, type, kind, shape, data}."""
    msgpack = pytest.importorskip("msgpack")
    f = _feats(5, 8, np.float16, 4)
    for key in (lambda s: s.encode(), lambda s: s):
        nd = {key("nd"): True, key("type"): "<f2", key("kind"): b"", key("shape"): [5, 8], key("data"): f.tobytes()}
        blob = msgpack.dumps({"features": nd}, use_bin_type=True)
        out = featdb.decode_record(blob, compress=False)["features"]
        assert out.dtype == np.float16 and np.array_equal(out, f)
    ours = msgpack.loads(featdb.encode_record(f, compress=False), raw=False, strict_map_key=False)["features"]
    assert ours[b"nd"] is True and ours[b"type"] == "<f2" and ours[b"shape"] == [5, 8] and ours[b"data"] == f.tobytes()
    rd = featdb.VideoFeatReader({b"v": featdb.encode_record(f, compress=False)}, {"v": 5}, compress=False, max_clip_len=100)
    assert torch.equal(rd["v"], torch.from_numpy(f.astype(np.float32)))


def test_reader_matches_the_reference_reader_on_reference_written_records():
    """tests/golden/case_featdb.npz (make_golden_featdb.py): database values written by the reference's dumps_npz and
    what the reference's VideoFeatLmdb returns for them - clipped read, unclipped get_dump with fp16 widening,
    _compute_nframe, and the default max_clip_len = -1 read."""
    import os
    from tests.util import GOLDEN
    g = np.load(os.path.join(GOLDEN, "case_featdb.npz"))
    names = sorted(k[4:] for k in g.files if k.startswith("rec."))
    assert names == ["tvr_a", "tvr_b", "tvr_c", "tvr_d"]
    store = {n.encode(): g["rec." + n].tobytes() for n in names}
    store[b"__keys__"] = json.dumps(names).encode()
    raw = {n: g["dump." + n].shape[0] for n in names}
    rd = featdb.VideoFeatReader(store, dict(raw), compress=True, max_clip_len=100)
    dflt = featdb.VideoFeatReader(store, dict(raw), compress=True)
    for n in names:
        out = rd[n]
        assert out.dtype == torch.float32 and torch.equal(out, torch.from_numpy(g["out." + n]))
        d = rd.get_dump(n)["features"]
        assert d.dtype == np.float32 and np.array_equal(d, g["dump." + n])
        assert torch.equal(dflt[n], torch.from_numpy(g["out_default." + n]))
    assert g["out_default.tvr_d"].shape[0] == 0 and g["out_default.tvr_a"].shape[0] == 129      # the -1 quirk is the reference's
    assert featdb.VideoFeatReader(store, None, compress=True, max_clip_len=100).name2nframe == json.loads(g["nframe"].tobytes().decode())


@pytest.mark.parametrize("compress", [True, False])
def test_reader_feeds_the_collate_to_the_reference_batch(compress):
    """End of the data path: features stored as database records (fp16 on disk, as the released TVR features are) ->
    VideoFeatReader (clip at max_clip_len) -> video_item / vcmr_collate = the batch the reference's own dataset + collate
    produced for the `narrow` case of tests/golden/case_collate.npz (its features are fp16-exact by construction of this
    test: they are rounded once before both paths would see them - here only the integer tensors and the clip are compared
    bit for bit, the features within fp16 rounding)."""
    import os
    from tests.util import GOLDEN
    from hero_amd import collate as C
    Z = np.load(os.path.join(GOLDEN, "case_collate.npz"))
    case = "narrow"
    desc = json.loads(str(Z[case + ".desc"]))
    want = {k[len(case) + 5:]: Z[k] for k in Z.files if k.startswith(case + ".out.")}
    store, nframe = {}, {}
    for v in desc["videos"]:
        f = Z["%s.feat.%s" % (case, v["vid"])]
        store[v["vid"].encode()] = featdb.encode_record(f.astype(np.float16), compress)
        nframe[v["vid"]] = len(f)
    rd = featdb.VideoFeatReader(store, nframe, compress=compress, max_clip_len=desc["max_clip_len"])
    s2f_all = json.loads(str(want["sub_idx2frame_idx"]))
    items = []
    by_vid = {v["vid"]: (i, v) for i, v in enumerate(desc["videos"])}
    for qid in desc["query_order"]:
        vi, v = by_vid["v" + qid[1:3]]
        feat = rd[v["vid"]]
        assert feat.shape[0] == min(nframe[v["vid"]], desc["max_clip_len"])
        video = C.video_item(feat, [(sid, list(fr)) for sid, fr in s2f_all[vi]], v["sub_tokens"], sep=2)
        q = v["queries"][int(qid.split("_")[1])]
        items.append(C.vcmr_item(video, v["vid"], [(q["tokens"], q["ts"])], cls_=0, frame_interval=desc["frame_interval"]))
    got = C.vcmr_collate(items)
    for k, w in want.items():
        if w.dtype.kind == "U":
            continue
        g = got[k]
        assert tuple(g.shape) == w.shape, k
        if w.dtype.kind == "f":
            torch.testing.assert_close(g, torch.from_numpy(w), rtol=1e-3, atol=1e-3)       # fp16 storage
        else:
            assert torch.equal(g, torch.from_numpy(w)), k
