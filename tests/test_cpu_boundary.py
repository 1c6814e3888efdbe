"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
include/hero_hip.h declares, state-dict schema equals the reference's, host index logic is right,
and the product path refuses to run without a GPU (no silent fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import hero_oracle as O
from tests.util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from hero_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "hero_hip.h")).read()
    declared = set(re.findall(r"\b(hero_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    handle = ctypes.CDLL(built_lib)
    missing = [s for s in sorted(declared) if not hasattr(handle, s)]
    assert not missing, missing
    from hero_amd import _lib
    assert set(_lib.EXPORTS) == declared
    handle.hero_abi_version.restype = ctypes.c_int
    assert handle.hero_abi_version() == 3 == _lib.ABI_VERSION         # include/hero_hip.h HERO_ABI_VERSION (history there)


def test_abi_struct_sizes_are_pinned_per_version(built_lib):
    """a struct of include/hero_hip.h that changes size must come with a new HERO_ABI_VERSION.  The sizes of
    every struct are committed per version (tests/golden/abi_sizes.json); the binding's ctypes mirrors, the header as gcc
    sees it, and the library's own hero_abi_struct_bytes() must all give the table of the CURRENT version, and the table of
    an older version is never edited (a changed struct under an old number fails here: bump the version, add a table)."""
    import json
    import subprocess
    import tempfile
    from hero_amd import _lib
    table = json.load(open(os.path.join(GOLDEN, "abi_sizes.json")))
    header = open(os.path.join(ROOT, "include", "hero_hip.h")).read()
    version = int(re.search(r"#define HERO_ABI_VERSION (\d+)", header).group(1))
    assert version == _lib.ABI_VERSION
    assert str(version) in table, "HERO_ABI_VERSION %d has no committed size table: add it to tests/golden/abi_sizes.json" % version
    want = table[str(version)]
    mine = _lib.abi_struct_sizes()
    assert mine == want, ("struct sizes differ from the table committed for ABI version %d - an incompatible change needs a "
                          "version bump (include/hero_hip.h, hero_amd/_lib.py, INTEGRATION.md) and a new table" % version,
                          {k: (mine.get(k), want.get(k)) for k in set(mine) | set(want) if mine.get(k) != want.get(k)})
    # every typedef'd struct of the header has an id, a mirror and the same size under gcc and inside the library
    names = re.findall(r"^} (Hero\w+);", header, flags=re.M)
    assert ["Hero" + n for n in mine] == names, "ABI_STRUCTS / HERO_STRUCT_* must list the header's structs in declaration order"
    src = '#include <stdio.h>\n#include "hero_hip.h"\nint main(){%s return 0;}\n' % "".join('printf("%%zu ", sizeof(%s));' % n for n in names)
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "s.c"), "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        gcc_sizes = list(map(int, subprocess.check_output([os.path.join(d, "s")]).split()))
    assert gcc_sizes == list(mine.values())
    handle = ctypes.CDLL(built_lib)
    assert handle.hero_abi_struct_count() == len(names)
    assert [handle.hero_abi_struct_bytes(i) for i in range(len(names))] == gcc_sizes
    assert handle.hero_abi_struct_bytes(len(names)) == -1 and handle.hero_abi_struct_bytes(-1) == -1
    # the enum ids are in declaration order too
    ids = re.search(r"enum \{ (HERO_STRUCT_DROPOUT.*?)HERO_STRUCT_COUNT_", header, flags=re.S).group(1)
    assert len(re.findall(r"HERO_STRUCT_[A-Z_]+", ids)) == len(names)


def test_binding_refuses_a_library_with_other_struct_sizes(built_lib, monkeypatch):
    """The load-time guard itself: a mirror that is 8 bytes short of the library's struct raises before any kernel runs."""
    from hero_amd import _lib

    class Short(ctypes.Structure):
        _fields_ = _lib.TensorDesc._fields_[:-2]
    Short.__name__ = "TensorDesc"
    structs = list(_lib.ABI_STRUCTS)
    structs[structs.index(_lib.TensorDesc)] = Short
    monkeypatch.setattr(_lib, "ABI_STRUCTS", structs)
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="struct layouts differ"):
        _lib.lib()
    monkeypatch.undo()
    _lib._lib = None
    _lib.lib()


def test_struct_layouts_match_header_sizes(built_lib):
    """ctypes mirrors must have the C sizes (checked against sizes computed by gcc)."""
    import subprocess
    import tempfile
    from hero_amd import _lib
    src = '#include <stdio.h>\n#include "hero_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(HeroDropout),sizeof(HeroGemmEpilogue),sizeof(HeroLnFwd),sizeof(HeroLnBwd),' \
          'sizeof(HeroAttn),sizeof(HeroAdamW));printf("%zu %zu %zu\\n",sizeof(HeroTensorDesc),' \
          'sizeof(HeroAdamWGroup),sizeof(HeroAdamWMulti));printf("%zu %zu %zu %zu\\n",sizeof(HeroCrossEntropy),' \
          'sizeof(HeroWgradProblem),sizeof(HeroColsum),sizeof(HeroDerive));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "s.c"), "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"),
                               "-o", os.path.join(d, "s")])
        sizes = list(map(int, subprocess.check_output([os.path.join(d, "s")]).split()))
    mine = [ctypes.sizeof(c) for c in (_lib.Dropout, _lib.GemmEpilogue, _lib.LnFwd, _lib.LnBwd,
                                       _lib.Attn, _lib.AdamW, _lib.TensorDesc, _lib.AdamWGroup,
                                       _lib.AdamWMulti, _lib.CrossEntropy, _lib.WgradProblem, _lib.Colsum, _lib.Derive)]
    assert mine == sizes
    # field offsets of the structs that changed in round 3
    for cls, cname in ((_lib.WgradProblem, "HeroWgradProblem"), (_lib.Colsum, "HeroColsum"), (_lib.Derive, "HeroDerive"),
                       (_lib.LnBwd, "HeroLnBwd")):
        fields = [f for f, _ in cls._fields_]
        src = '#include <stdio.h>\n#include <stddef.h>\n#include "hero_hip.h"\nint main(){%s return 0;}\n' % "".join(
            'printf("%%zu ", offsetof(%s, %s));' % (cname, f) for f in fields)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "o.c"), "w") as f:
                f.write(src)
            subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "o.c"), "-o", os.path.join(d, "o")])
            offs = list(map(int, subprocess.check_output([os.path.join(d, "o")]).split()))
        assert [getattr(cls, f).offset for f in fields] == offs, cname


def test_integration_md_ctypes_snippet_matches_header():
    """The ctypes mirrors printed in INTEGRATION.md are what a maintainer copies: every struct defined
    there must have the header's size and field offsets (a stale mirror hands kernels garbage)."""
    import subprocess
    import tempfile
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "class GemmEpilogue" in b)
    classes = "\n".join(m.group(0) for m in re.finditer(r"^class \w+\(C\.Structure\):\n(?:    .*\n?|\s*\n)+?(?=^\S|\Z)",
                                                         code, flags=re.M))
    ns = {"C": ctypes}
    exec(classes, ns)
    mirrors = {k: v for k, v in ns.items() if isinstance(v, type) and issubclass(v, ctypes.Structure)}
    assert {"Dropout", "GemmEpilogue"} <= set(mirrors)
    lines = []
    for name, cls in sorted(mirrors.items()):
        cname = "Hero" + name
        lines.append('printf("%s %%zu", sizeof(%s));' % (name, cname))
        for f, _ in cls._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, f))
        lines.append('printf("\\n");')
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "hero_hip.h"\nint main(){%s return 0;}\n' % "".join(lines)
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "s.c"), "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"),
                               "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().splitlines()
    for line in out:
        name, size, *offs = line.split()
        cls = mirrors[name]
        assert ctypes.sizeof(cls) == int(size), name
        assert [getattr(cls, f).offset for f, _ in cls._fields_] == list(map(int, offs)), name


def test_state_dict_schema_equals_reference():
    from hero_amd.model import HeroForVcmr
    z = np.load(os.path.join(GOLDEN, "tiny_model.npz"))
    ref = {k: tuple(z[k].shape) for k in z.files if not k.startswith("__")}
    m = HeroForVcmr.from_pretrained(os.path.join(GOLDEN, "tiny_config.json"), {}, vfeat_dim=96,
                                    max_frm_seq_len=16, lw_neg_ctx=8.0, lw_neg_q=8.0)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == ref
    # tied decoder / word embedding (model/layers.py:342-345)
    fe = m.v_encoder.f_encoder
    assert fe.lm_head.decoder.weight is fe.embeddings.word_embeddings.weight
    # default LayerNorm eps: 1e-12 in encoder layers, 1e-5 elsewhere (SURVEY §8 a'.3)
    assert fe.encoder.layer[0].output.LayerNorm.eps == 1e-12
    assert fe.embeddings.LayerNorm.eps == 1e-5 and m.v_encoder.frame_transform.LayerNorm.eps == 1e-5


def test_pad_vocab_and_partial_checkpoint():
    from hero_amd.model.modeling_utils import load_partial_checkpoint, pad_tensor_to_mul
    t, n = pad_tensor_to_mul(torch.ones(50265, 4))
    assert t.shape[0] == 50272 and n == 7 and float(t[50265:].abs().sum()) == 0
    t, n = pad_tensor_to_mul(torch.ones(16, 4))
    assert n == 0 and t.shape[0] == 16
    ck = {"roberta.encoder.layer.%d.w" % i: i for i in range(12)}
    ck["roberta.embeddings.x"] = -1
    out = load_partial_checkpoint(ck, 6)
    assert out == {**{"roberta.encoder.layer.%d.w" % j: 2 * j + 1 for j in range(6)}, "roberta.embeddings.x": -1}


def test_frame_map_matches_reference_collect():
    from hero_amd.model.model import build_frame_map
    batch, _ = O.load_npz_case(os.path.join(GOLDEN, "case_ragged.npz"))
    T, L = batch["f_attn_masks"].shape
    B, NF = batch["c_v_feats"].shape[:2]
    offs, ent, inv = build_frame_map(batch["num_subs"], batch["sub_idx2frame_idx"], B, NF, L, "cpu")
    f_seq = torch.randn(T, L, 8)
    want = O.collect_frame_outputs(f_seq, batch["num_subs"], batch["sub_idx2frame_idx"], B, NF)
    flat = f_seq.reshape(T * L, 8)
    got = torch.zeros(B * NF, 8)
    for r in range(B * NF):
        for e in range(int(offs[r]), int(offs[r + 1])):
            got[r] += flat[int(ent[e])]
    torch.testing.assert_close(got.view(B, NF, 8), want)
    for s, d in enumerate(inv.tolist()):
        if d >= 0:
            assert s in ent[int(offs[d]):int(offs[d + 1])].tolist()


def test_flat_gather_index_matches_torch_gather():
    from hero_amd.model.encoder import CrossModalTrm
    batch, _ = O.load_npz_case(os.path.join(GOLDEN, "case_ragged.npz"))
    gi = batch["f_gather_index"]
    T = gi.shape[0]
    max_vl, max_sl = batch["f_v_feats"].shape[1], batch["f_sub_input_ids"].shape[1]
    img, txt = torch.randn(T, max_vl, 4), torch.randn(T, max_sl, 4)
    want = torch.gather(torch.cat([img, txt], 1), 1, gi.unsqueeze(-1).expand(-1, -1, 4))
    flat = CrossModalTrm._flat_gather_index(gi, max_vl, max_sl).long()
    a, b = img.reshape(-1, 4), txt.reshape(-1, 4)
    got = torch.where((flat >= 0)[:, None], a[flat.clamp(min=0)], b[(-flat - 2).clamp(min=0)])
    torch.testing.assert_close(got.view(T, -1, 4), want)


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of silently computing something else."""
    from hero_amd import functional as HF
    from tests.util import load_tiny
    with pytest.raises(RuntimeError, match="no CPU fallback|CUDA"):
        HF.k_cast(torch.zeros(8), torch.bfloat16)
    model, _, _ = load_tiny("cpu")
    batch, _ = O.load_npz_case(os.path.join(GOLDEN, "case_regular.npz"))
    with pytest.raises(RuntimeError):
        model.v_encoder(batch, "repr")


def test_synth_batch_contract():
    """Synthetic batches carry the reference collate's keys/shapes/dtypes (data/data.py:459-470)."""
    from hero_amd.synth import make_batch
    b = make_batch("D1", vfeat_dim=32, vocab=100)
    assert b["f_v_feats"].shape == (16, 4, 32) and b["f_sub_input_ids"].shape == (16, 8)
    assert b["f_attn_masks"].shape == (16, 12) and b["c_v_feats"].shape == (2, 32, 32)
    assert b["f_gather_index"].dtype == torch.int64 and b["f_sub_input_ids"][:, 0].eq(2).all()
    assert b["query_input_ids"].shape == (2, 12) and b["query_input_ids"][:, 0].eq(0).all()
    assert b["num_subs"] == [8, 8] and all(s == i for v in b["sub_idx2frame_idx"] for i, (s, _) in enumerate(v))
    r = make_batch("D2", vfeat_dim=8, vocab=100, ragged=True, videos=4)
    assert r["f_attn_masks"].shape[0] == sum(r["num_subs"])
    # a subtitle with no frame: zero feature row, first mask bit 0 (data/data.py:380-382)
    row = 0
    for v in r["sub_idx2frame_idx"]:
        for _, fr in v:
            if not fr:
                assert r["f_attn_masks"][row, 0] == 0 and r["f_v_feats"][row].abs().sum() == 0
            row += 1


def test_grad_exchange_backend_selection():
    """The exchange is chosen in code (`set_exchange`, `GradArena(backend=...)`), never by the environment: a CPU arena (the
    gloo tests) is always on the process group, host tensors never take the `hero_comm_*` path, the C-ABI exchange needs a
    CUDA arena, and an unknown backend is an error."""
    import pytest
    import torch
    from hero_amd.utils import distributed as D
    from hero_amd import functional as HF
    ps = [torch.nn.Parameter(torch.randn(8, 4))]
    try:
        assert D.exchange() == "torch"
        assert D.GradArena(ps, install=False).backend == "torch"
        assert not D._abi_on(torch.zeros(4))
        D._EXCHANGE[0] = "abi"                      # what set_exchange("abi") leaves behind (it also opens RCCL: needs a GPU)
        assert D.GradArena(ps, install=False).backend == "torch" and not D._abi_on(torch.zeros(4))
        D._EXCHANGE[0] = "torch"
        with pytest.raises(ValueError):
            D.GradArena(ps, install=False, backend="abi")
        with pytest.raises(ValueError):
            D.GradArena(ps, install=False, backend="mpi")
        with pytest.raises(ValueError):
            D.set_exchange("horovod")
    finally:
        D._EXCHANGE[0] = "torch"
        HF.set_grad_sink(None)


def test_grad_arena_groups_are_contiguous():
    """GradArena(groups=...) lays the grouped parameters back to back (in group order) wherever the
    first member falls, keeps 16-byte alignment, and dst_group() returns one stacked view of them."""
    import torch
    from hero_amd.utils import distributed as D
    from hero_amd import functional as HF
    ps = [torch.nn.Parameter(torch.randn(*s)) for s in [(8, 4), (6,), (8, 4), (3,), (8, 4), (8,), (8,), (8,)]]
    wq, odd, wk, odd2, wv, bq, bk, bv = ps
    try:
        arena = D.GradArena(ps, bucket_bytes=64, groups=[(wq, wk, wv), (bq, bk, bv)])
        g = arena.dst_group((wq, wk, wv))
        assert g is not None and g.shape == (24, 4)
        g.copy_(torch.arange(96.0).view(24, 4))
        assert torch.equal(wq.grad, g[:8]) and torch.equal(wk.grad, g[8:16]) and torch.equal(wv.grad, g[16:])
        gb = arena.dst_group((bq, bk, bv))
        assert gb is not None and gb.shape == (24,)
        assert arena.dst_group((wq, wv)) is None and arena.dst_group((odd, odd2)) is None
        for p in ps:
            s, e = arena.slices[p]
            assert s % 4 == 0 and e - s == p.numel()
        spans = sorted(arena.slices[p] for p in ps)
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))          # no overlap
        assert sum(b[2] for b in arena.buckets) == len(ps)
    finally:
        HF.set_grad_sink(None)


def test_pack_plan_maps_are_consistent():
    """BertEncoder._pack_plan (host logic of the packed ragged path): gather / inverse / per-group maps
    and sequence offsets describe the same permutation; dense batches get no plan."""
    import torch
    from hero_amd.model.layers import BertEncoder
    BertEncoder._PLANS.clear()
    m1 = torch.tensor([[1, 1, 1, 0, 0], [0, 1, 1, 1, 1], [1, 0, 0, 0, 0]])       # subtitle rows (one frameless)
    m2 = torch.tensor([[1, 1, 0], [1, 1, 1]])                                      # query rows
    plan = BertEncoder._pack_plan([m1, m2], [15, 6])
    assert plan is not None
    gather, inverse, inv, back, off, n_seq, lmax = plan
    flat = torch.cat([m1.reshape(-1), m2.reshape(-1)]).bool()
    assert gather.tolist() == torch.nonzero(flat).reshape(-1).tolist()
    assert n_seq == 5 and lmax == 4 and off.tolist() == [0, 3, 7, 8, 10, 13]
    assert (inverse >= 0).sum() == flat.sum() and torch.equal(inverse[gather.long()], torch.arange(13, dtype=torch.int32))
    # per group: padded position -> packed row (or -1), packed row -> padded position of that group (or -1)
    assert inv[0].tolist() == [0, 1, 2, -1, -1, -1, 3, 4, 5, 6, 7, -1, -1, -1, -1]
    assert inv[1].tolist() == [8, 9, -1, 10, 11, 12]
    assert back[0].tolist() == [0, 1, 2, 6, 7, 8, 9, 10, -1, -1, -1, -1, -1]
    assert back[1].tolist() == [-1] * 8 + [0, 1, 3, 4, 5]
    dense = torch.ones(4, 6, dtype=torch.long)
    assert BertEncoder._pack_plan([dense], [24]) is None                          # nothing to drop
    BertEncoder._PLANS.clear()


def test_wgrad_split_model_choices():
    """functional._split_for: the reduction split of dW = dY^T X per step shape (fitted on MI355X)."""
    from hero_amd.functional import _split_for
    assert _split_for(3072, 768, 12000, 64) == 3 and _split_for(2304, 768, 12000, 64) == 4
    assert _split_for(768, 768, 12000, 64) in (6, 7) and _split_for(768, 4352, 1920, 64) == 1
    assert _split_for(768, 768, 64, 64) == 1                                       # never more splits than K tiles allow


def test_memo_tracks_identity_and_version():
    """functional.memo caches tensors derived from batch index / mask tensors: hit on the same object,
    miss after an in-place edit (version bump), after a shape-changing view, and for another object."""
    import torch
    from hero_amd import functional as HF
    calls = []
    ids = torch.arange(6).view(2, 3)

    def derive():
        calls.append(1)
        return ids.reshape(-1).to(torch.int32)
    a = HF.memo("t", (ids,), derive)
    b = HF.memo("t", (ids,), derive)
    assert a is b and len(calls) == 1
    ids.add_(1)                                            # in-place edit -> new version -> recomputed
    c = HF.memo("t", (ids,), derive)
    assert c is not a and len(calls) == 2 and c.tolist() == [1, 2, 3, 4, 5, 6]
    HF.memo("t", (ids,), derive, extra=(1,))               # a different `extra` is a different entry
    assert len(calls) == 3
    other = ids.clone()
    HF.memo("t", (other,), derive)
    assert len(calls) == 4


@pytest.mark.parametrize("rows,layers,extra", [(12000, 6, ()), (1920, 3, ((768, 4352),)), (12040, 1, ()), (786432, 2, ()),
                                               (393216, 1, ()), (12000, 1, ((768, 768),) * 3), (4096, 1, ((2304, 768),)), (12000, 5, ()), (8192, 1, ())])
def test_wgrad_batch_plan_covers_every_tile_once(built_lib, rows, layers, extra):
    """hero_wgrad_batch_plan is host code (no GPU): every 192 x 192 tile of every problem appears exactly once per
    k-step; full rounds hold whole tiles; the pieces of a tail tile are contiguous in slice order, share one flag and one
    XCD (workgroup ids w with equal w // (nwg / 8)), no two tail tiles share a flag, a workgroup holds at most one item
    per round, and (round 4) the tail is BALANCED: no workgroup of an XCD that has tail tiles carries more than its
    quota of k-steps (+ rounding), so one BertLayer (192 tiles, 24 per XCD) fills all 256 workgroups."""
    from hero_amd import _lib as L
    lib = L.lib()
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] * layers + list(extra)
    n = len(shapes)
    probs = (L.WgradProblem * n)()
    for i, (m, k) in enumerate(shapes):
        probs[i] = L.WgradProblem(0, 0, 0, m, k, m, k, k, 4)
    buf = np.zeros(8 + 8 * 512 * 16, dtype=np.int32)
    words = lib.hero_wgrad_batch_plan(probs, n, rows, buf.ctypes.data, buf.size)
    assert words > 8, lib.hero_last_error()
    magic, nwg, rounds, items, n_, K, tiles, S = buf[:8]
    assert (n_, K, items, words) == (n, rows, rounds * nwg, 8 + rounds * nwg * 8) and nwg % 8 == 0
    ksteps = -(-rows // 64)
    plan = buf[8:words].reshape(rounds, nwg, 8)
    expect = {(i, a * 192, b * 192) for i, (m, k) in enumerate(shapes) for a in range(-(-m // 192)) for b in range(-(-k // 192))}
    assert tiles == len(expect)
    full = tiles // nwg
    cover, flags = {}, {}
    load = np.zeros(nwg, dtype=np.int64)
    for r in range(rounds):
        for w in range(nwg):
            prob, m0, n0, k0, nk, order, nslices, flag = plan[r, w]
            if nk == 0:
                continue                                      # idle slot of a tail round
            key = (int(prob), int(m0), int(n0))
            assert key in expect
            cover.setdefault(key, []).append((int(order), int(k0), int(nk), int(nslices), r, w, int(flag)))
            if r < full:
                assert (k0, nk, order, nslices) == (0, ksteps, 0, 1)          # a full round: whole tiles, no merge
            else:
                load[w] += nk
    assert set(cover) == expect
    for key, parts in cover.items():
        parts.sort()
        assert [p[0] for p in parts] == list(range(len(parts))) and all(p[3] == len(parts) for p in parts)
        assert parts[0][1] == 0 and sum(p[2] for p in parts) == ksteps
        assert all(a[1] + a[2] == b[1] for a, b in zip(parts, parts[1:]))     # contiguous, in slice order
        if len(parts) > 1:
            assert all(p[4] >= full for p in parts)                           # tail rounds only
            assert len({p[5] // (nwg // 8) for p in parts}) == 1              # one XCD: the merge stays in its L2
            assert len({p[6] for p in parts}) == 1
            assert flags.setdefault(parts[0][6], key) == key                  # a flag per tail tile
            assert min(p[2] for p in parts) >= 2 and len(parts) <= 10
    rem = tiles % nwg
    if rem:
        per_xcd = -(-rem // 8)
        for x in range(8):
            p = min(per_xcd, max(0, rem - x * per_xcd))
            if p == 0:
                continue
            quota = max(4, -(-p * ksteps // (nwg // 8)))
            xl = load[x * (nwg // 8):(x + 1) * (nwg // 8)]
            assert xl.sum() == p * ksteps
            c = nwg // 8
            # equal slices (round 3): S = c // p per tile, at least 4 k-steps each, at most 8 - and (round 6) only as many as
            # pay for their ordered atomics: ~16 us per slice against ~8 us for a whole tile's plain read-add-write, a k-step
            # ~1.17 us (profiles/r06_wgrad_ceiling.txt) - short reductions keep whole tiles
            S_ = max(1, min(c // p, 8, ksteps // 4))
            cost = lambda sl: -(-ksteps // sl) * 1.17 + (8.0 if sl == 1 else 9.0 + 16.0 * sl)      # noqa: E731
            while S_ > 1 and cost(S_ - 1) <= cost(S_):
                S_ -= 1
            if ksteps >= 128 and c % p and c // p <= 8 and (c % p) * 8 >= c and S_ == c // p:   # long reductions, a real share of
                assert xl.max() <= quota + max(8, quota // 8), (x, xl.tolist(), quota)   # the workgroups idle: they carry the remainders
            else:
                assert xl.max() <= -(-ksteps // S_)
            if ksteps <= 32:
                assert S_ == 1                                 # the 1920-row / 480-row groups of the TVR step: no sliced tiles


def test_attention_capability_queries(built_lib):
    """Host-side capability queries the Python layer branches on (no GPU needed): which lengths run packed, and where the
    backward works from the softmax row statistics instead of saved probabilities."""
    from hero_amd import _lib as L
    lib = L.lib()
    assert lib.hero_attention_max_packed_len(L.BF16) == 256 and lib.hero_attention_max_packed_len(L.F32) == 64
    assert [lib.hero_attention_stats_ok(L.BF16, n) for n in (1, 24, 64, 65, 256)] == [1, 1, 1, 0, 0]
    assert lib.hero_attention_stats_ok(L.F32, 24) == 0
    assert lib.hero_attention_max_len(L.BF16, 1) == 256 and lib.hero_attention_max_len(L.F32, 0) >= 256


def test_stack_and_split_rows_are_cat_and_slices_for_autograd():
    """functional.StackRowsFn / SplitRowsFn (the sub + query stack of BertEncoder.forward_multi) are torch.cat and row
    slices as far as autograd is concerned - including a block that receives no gradient."""
    from hero_amd import functional as HF
    torch.manual_seed(0)
    a = torch.randn(5, 8, requires_grad=True)
    b = torch.randn(3, 8, requires_grad=True)
    w0, w1 = torch.randn(5, 8), torch.randn(3, 8)
    x = HF.StackRowsFn.apply(a, b)
    assert torch.equal(x, torch.cat([a, b], 0))
    y0, y1 = HF.SplitRowsFn.apply(x * 2.0, 5, 3)
    ((y0 * w0).sum() + (y1.view(3, 8) * w1).sum()).backward()
    ra, rb = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    rx = torch.cat([ra, rb], 0) * 2.0
    ((rx[:5] * w0).sum() + (rx[5:] * w1).sum()).backward()
    torch.testing.assert_close(a.grad, ra.grad)
    torch.testing.assert_close(b.grad, rb.grad)
    a.grad = b.grad = None
    y0, y1 = HF.SplitRowsFn.apply(HF.StackRowsFn.apply(a, b) * 3.0, 5, 3)
    (y1 * w1).sum().backward()                       # the first block gets no gradient: zeros, not garbage
    assert torch.equal(a.grad, torch.zeros_like(a))
    torch.testing.assert_close(b.grad, 3.0 * w1)


def test_no_memset_nodes_in_the_kernel_library():
    """hipMemsetAsync inside a captured hipGraph is not ordered like the kernel nodes around it on this
    stack (round 1: the ranking-loss buffer was zeroed by a memset node that raced with the kernel that
    accumulates into it whenever the queue was idle -> late-training collapses in graph mode only).
    Buffers are zeroed by kernels; keep it that way."""
    import glob
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hero_amd", "csrc")
    for path in glob.glob(os.path.join(root, "*")):
        if path.endswith((".hip", ".cpp", ".h")):
            code = "\\n".join(line.split("//")[0] for line in open(path).read().splitlines())
            assert "hipMemset" not in code, path


def test_move_to_device_recurses_like_the_reference():
    """data/loader.py:49-66 (move_to_cuda): lists, tuples and dicts are walked recursively - `PrefetchLoader(MetaLoader(...))`
    hands `(task, batch_dict)` tuples to it (pretrain.py:177-180).  Checked with the `meta` device (no GPU needed)."""
    import collections
    import torch
    from hero_amd.loader import move_to_device
    t = lambda: torch.zeros(2, 3)                                              # noqa: E731
    NT = collections.namedtuple("NT", ["a", "b"])
    item = ("tvr", {"x": t(), "nested": [t(), (t(), 7, "s")], "lens": [3, 4], "nt": NT(t(), None)})
    out = move_to_device(item, torch.device("meta"))
    assert out[0] == "tvr" and isinstance(out, tuple)
    b = out[1]
    assert b["x"].device.type == "meta" and b["nested"][0].device.type == "meta" and b["nested"][1][0].device.type == "meta"
    assert b["nested"][1][1:] == (7, "s") and b["lens"] == [3, 4] and isinstance(b["nested"][1], tuple)
    assert isinstance(b["nt"], NT) and b["nt"].a.device.type == "meta" and b["nt"].b is None


def test_wgrad_queue_flushes_at_whole_round_points(monkeypatch):
    """functional._wgrad_queue_full (host logic): under the soft byte cap nothing is flushed; past it the queue waits for
    a tile count that fills whole rounds (or an evenly sliceable tail, priced 1.4x); past the hard cap it flushes anyway."""
    import types
    from hero_amd import functional as HF
    monkeypatch.setenv("HERO_WGRAD_QUEUE_HARD_MB", "100")
    monkeypatch.setattr(HF, "_WQ_HARD", {})
    monkeypatch.setattr(HF, "WGRAD_QUEUE_BYTES", [10 << 20])
    dev = types.SimpleNamespace(index=0)
    layer = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]          # 64 + 64 + 16 + 48 tiles of 192 x 192

    def queue(shapes, nbytes):
        q = [(None, types.SimpleNamespace(shape=(1, k)), None, 0, n, None, None, None) for n, k in shapes]
        monkeypatch.setattr(HF, "_WQ", q)
        monkeypatch.setattr(HF, "_WQ_BYTES", [nbytes])
        return HF._wgrad_queue_full(dev)

    assert not queue(layer, 5 << 20)                       # under the soft cap
    assert not queue(layer, 20 << 20)                      # 192 tiles: 3/4 of a round - wait for more
    assert queue(layer + layer[:1], 20 << 20)              # 256 tiles: one whole round
    assert not queue(layer[:2], 20 << 20)                  # 128 tiles would be two even slices: 0.71 of a whole-tile round
    assert queue(layer[:2], 101 << 20)                     # ... but past the hard cap everything goes
    assert queue(layer * 6, 20 << 20)                      # 1152 tiles = 4.5 rounds (the TVR stack): 4 whole + 128 in two slices


def test_hero_comm_boundary_without_a_gpu(built_lib):
    """hero_comm_*: librccl is dlopen-ed lazily (libhero_hip.so does not link it), rank 0's unique id is 128 bytes, and
    argument errors come back as codes with a message - no collective is attempted without a communicator."""
    import ctypes as C
    import subprocess
    from hero_amd import _lib
    L = _lib.lib()
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in needed
    assert L.hero_comm_available() in (0, 1)
    assert L.hero_comm_rank(None) == -1 and L.hero_comm_world(None) == 0 and L.hero_comm_destroy(None) == 0
    arr = (_lib.CommBucket * 1)()
    assert L.hero_comm_allreduce_buckets(None, arr, 1, None) != 0 and b"hero_comm" in L.hero_last_error()
    assert L.hero_comm_broadcast(None, None, 0, 0, None) != 0
    assert L.hero_comm_allgather(None, None, None, 0, None) != 0
    assert L.hero_comm_allgather_var(None, None, None, None, None) != 0
    if L.hero_comm_available():
        uid = (C.c_char * 128)()
        assert L.hero_comm_unique_id(uid) == 0 and any(uid.raw)
        h = C.c_void_p()
        assert L.hero_comm_init(uid, 2, 2, C.byref(h)) != 0 and b"bad arguments" in L.hero_last_error()


def test_pack_plan_row_maps_are_consistent():
    """BertEncoder._pack_plan (host logic of the ragged path, model/layers.py:299-302 semantics): the padded -> packed and
    packed -> padded row maps of several sequence groups are inverse to each other, sequences stay contiguous and in order,
    and (almost) dense batches / sequences beyond the variable-length kernels' limit get no plan."""
    import torch
    from hero_amd.model.layers import BertEncoder
    g = torch.Generator().manual_seed(5)
    lens_a = torch.randint(3, 25, (7,), generator=g)
    lens_b = torch.randint(1, 16, (4,), generator=g)
    ma = (torch.arange(24)[None, :] < lens_a[:, None]).long()
    mb = (torch.arange(15)[None, :] < lens_b[:, None]).long()
    plan = BertEncoder._pack_plan([ma, mb], [7 * 24, 4 * 15], max_len=64)
    assert plan is not None
    gather, inverse, inv, back, off, n_seq, lmax = plan
    valid = int(ma.sum() + mb.sum())
    assert gather.numel() == valid and n_seq == 11 and lmax == int(max(lens_a.max(), lens_b.max()))
    flat = torch.cat([ma.reshape(-1), mb.reshape(-1)])
    assert torch.equal(flat[gather.long()], torch.ones(valid, dtype=flat.dtype))            # only valid positions are packed
    assert torch.equal(inverse[gather.long()].long(), torch.arange(valid))                 # padded -> packed undoes packed -> padded
    assert int((inverse < 0).sum()) == flat.numel() - valid
    assert torch.equal(off.long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.cat([lens_a, lens_b]).cumsum(0)]))
    assert torch.equal(torch.sort(gather).values, gather)                                   # sequences stay in order, rows contiguous
    # per group: inv = this group's slice of `inverse`; back = packed row -> row inside the group or -1
    assert torch.equal(inv[0], inverse[:7 * 24]) and torch.equal(inv[1], inverse[7 * 24:])
    for gi, (r0, n) in enumerate(((0, 7 * 24), (7 * 24, 4 * 15))):
        own = (gather >= r0) & (gather < r0 + n)
        assert torch.equal(back[gi][own].long(), gather[own].long() - r0) and bool((back[gi][~own] == -1).all())
    # dense batch: nothing to gain; too long for the packed attention kernels: no plan either
    assert BertEncoder._pack_plan([torch.ones(4, 10, dtype=torch.long)], [40], max_len=64) is None
    long_mask = (torch.arange(100)[None, :] < torch.tensor([100, 3, 3, 3])[:, None]).long()
    assert BertEncoder._pack_plan([long_mask], [400], max_len=64) is None


def test_the_package_reads_only_documented_environment_variables():
    """at most ten HERO_* variables, every one of them in README.md's table; the kernel library reads
    none (no getenv in csrc), and gemm_wsd.hip - round 4's measured negative result - is not part of the product library."""
    import glob
    pkg = os.path.join(ROOT, "hero_amd")
    read = set()
    for f in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True):
        src = open(f).read()
        read |= set(re.findall(r"""environ(?:\.get|\.setdefault)?\s*[\(\[]\s*["'](HERO_[A-Z0-9_]+)""", src))
        assert "getenv" not in src, f
    for f in glob.glob(os.path.join(pkg, "csrc", "*.*")):
        if f.endswith((".hip", ".cpp", ".h")):
            assert "getenv" not in open(f).read(), "%s reads the environment" % f
    readme = open(os.path.join(ROOT, "README.md")).read()
    table = readme[readme.index("## Environment variables"):]
    documented = set(re.findall(r"^\| `(HERO_[A-Z0-9_]+)` \|", table, flags=re.M))
    assert read == documented, (sorted(read - documented), sorted(documented - read))
    assert len(read) <= 10
    assert not os.path.exists(os.path.join(pkg, "csrc", "gemm_wsd.hip"))
    from hero_amd import build as hb
    assert "gemm_wsd.hip" not in hb.SOURCES
