"""Kernel-level parity: every C-ABI op against a plain fp32 torch restatement of the same op on
the same seeded inputs.  Needs a real MI355X (-m gpu)."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]
# written tolerances: fp32 path = exact-f32 MFMA / fp32 VALU (only summation order differs);
# bf16 path = bf16 storage of inputs/outputs (2^-9 relative per rounding), fp32 accumulation.
TOL = {torch.float32: dict(rtol=2e-5, atol=2e-5), torch.bfloat16: dict(rtol=2e-2, atol=2e-2)}


@pytest.fixture(scope="module")
def HF():
    from hero_amd import functional
    return functional


@pytest.fixture(scope="module")
def Lb():
    from hero_amd import _lib
    _lib.lib()
    return _lib


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def close(a, b, dtype, scale=1.0):
    t = TOL[dtype]
    torch.testing.assert_close(a.float(), b.float(), rtol=t["rtol"], atol=t["atol"] * scale)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 96), (1, 8, 8), (515, 768, 768),
                                   (480, 3072, 768), (131, 768, 4352)])
def test_gemm_forward_epilogues(HF, Lb, dtype, M, N, K):
    x, w, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=0.05), rnd(N, seed=3)
    res = rnd(M, N, dtype=dtype, seed=4)
    ref = x.float() @ w.float().t() + b
    y = HF.k_linear(x, w, b)
    close(y, ref, dtype, scale=math.sqrt(K) * 0.05)
    aux = torch.empty_like(y)
    y = HF.k_linear(x, w, b, act=Lb.ACT_GELU, aux=aux)
    close(aux, ref, dtype, scale=math.sqrt(K) * 0.05)
    close(y, torch.nn.functional.gelu(ref), dtype, scale=math.sqrt(K) * 0.05)
    aux = torch.empty_like(y)
    y = HF.k_linear(x, w, b, act=Lb.ACT_RELU, aux=aux, residual=res)
    close(aux, torch.relu(ref), dtype, scale=math.sqrt(K) * 0.05)
    close(y, torch.relu(ref) + res.float(), dtype, scale=math.sqrt(K) * 0.05 + 1)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (203, 136, 96), (515, 768, 3072), (4000, 768, 768),
                                   (1024, 200, 328), (11520, 768, 768), (1920, 2304, 768)])
def test_gemm_dgrad_wgrad(HF, Lb, dtype, M, N, K):
    """dgrad: dy[M,N] @ W[N,K]; wgrad: dy^T @ x (fp32 out, split-K atomics for large M)."""
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=0.05)
    dy = rnd(M, N, dtype=dtype, seed=5)
    u = rnd(M, K, dtype=dtype, seed=6)
    r = rnd(M, K, dtype=dtype, seed=7)
    dx = HF.k_dgrad(dy, w)
    close(dx, dy.float() @ w.float(), dtype, scale=math.sqrt(N) * 0.05)
    dxg = HF.k_dgrad(dy, w, act=Lb.ACT_GELU_BWD, aux=u, residual=r)
    uf = u.float()
    gp = 0.5 * (1 + torch.erf(uf / math.sqrt(2))) + uf * torch.exp(-0.5 * uf * uf) / math.sqrt(2 * math.pi)
    close(dxg, (dy.float() @ w.float()) * gp + r.float(), dtype, scale=math.sqrt(N) * 0.05 + 1)
    dW = HF.k_wgrad(dy, x)
    assert dW.dtype == torch.float32
    torch.testing.assert_close(dW, dy.float().t() @ x.float(), rtol=1e-4, atol=1e-3 * math.sqrt(M))
    close(HF.k_colsum(dy), dy.float().sum(0), torch.float32, scale=math.sqrt(M) * (1 if dtype == torch.float32 else 1))


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4])
def test_gemm_forced_geometries(HF, Lb, cfg):
    """Every tile geometry of the K-contiguous path (128x128, 256x256, 64x64) and the register-staged
    loop (bit 2) give the same result on a ragged shape, with the fused epilogues."""
    dtype = torch.bfloat16
    M, N, K = 700, 776, 768
    x, w, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=0.05), rnd(N, seed=3)
    res = rnd(M, N, dtype=dtype, seed=4)
    ref = x.float() @ w.float().t() + b
    Lb.lib().hero_gemm_force_config(cfg)
    try:
        y0 = HF.k_linear(x, w, b)
        aux = torch.empty_like(y0)
        y1 = HF.k_linear(x, w, b, act=Lb.ACT_GELU, aux=aux)
        y2 = HF.k_linear(x, w, b, residual=res)
        dy = rnd(M, N, dtype=dtype, seed=5)
        wt = w.t().contiguous()                      # [K, N]: dgrad through the transposed copy
        dx = HF.k_dgrad_t(dy, wt, act=Lb.ACT_GELU_BWD, aux=x)
    finally:
        Lb.lib().hero_gemm_force_config(-1)
    sc = math.sqrt(K) * 0.05
    close(y0, ref, dtype, scale=sc)
    close(aux, ref, dtype, scale=sc)
    close(y1, torch.nn.functional.gelu(ref), dtype, scale=sc)
    close(y2, ref + res.float(), dtype, scale=sc + 1)
    xf = x.float()
    gp = 0.5 * (1 + torch.erf(xf / math.sqrt(2))) + xf * torch.exp(-0.5 * xf * xf) / math.sqrt(2 * math.pi)
    close(dx, (dy.float() @ w.float()) * gp, dtype, scale=math.sqrt(N) * 0.05 * 2)


@pytest.mark.parametrize("M,N,K", [(12000, 768, 768), (12000, 2304, 768), (12000, 768, 3072), (1000, 200, 128),
                                   (385, 192, 64), (24000, 3072, 768)])
def test_gemm_wave_specialised_matches_4wave_kernels(HF, Lb, M, N, K):
    """gemm_ws.hip (persistent 192x192 tiles, loader / compute waves) against the 4-wave kernels and the fp32
    reference: the six fused epilogues, the SAME dropout mask (index m*N + n), row / column tails, several
    tiles per workgroup (24000 x 3072: 2000 tiles on 256 CUs)."""
    _ws_against_4wave(HF, Lb, M, N, K, 9)


@pytest.mark.parametrize("M,N,K", [(1920, 3072, 768), (1920, 2304, 768), (1000, 200, 128), (385, 192, 64), (5000, 3072, 768)])
def test_gemm_wave_specialised_128x192_tiles(HF, Lb, M, N, K):
    """The 128 x 192 geometry of the same kernel (Geo<2, 3>: four 32-row epilogue passes), which the 1920-row GEMMs of
    the Temporal Transformer take; row / column tails and more than one tile per workgroup (5000 x 3072: 640 tiles)."""
    _ws_against_4wave(HF, Lb, M, N, K, 10)


@pytest.mark.parametrize("M,N,K,cfg", [(1920, 768, 3072, 13), (1920, 768, 768, 13), (2497, 776, 1536, 13), (1920, 768, 4352, 13), (130, 136, 64, 13),
                                       (5000, 768, 768, 13), (1920, 768, 3072, 14), (1000, 392, 640, 14), (3000, 3072, 768, 14)])
def test_gemm_wave_specialised_64_row_tiles(HF, Lb, M, N, K, cfg):
    """The 64-row geometries of the wave-specialised kernel (round 4: Geo<1, 2> = 64 x 128 tiles with a SIX-deep ring,
    Geo<1, 3> = 64 x 192 with four stages) that the Temporal Transformer's 1920-row GEMMs into N = 768 take: the six fused
    epilogues against the 4-wave kernels and fp32 torch, row / column tails, a single k-step (K = 64 - fewer stages than the
    ring is deep), more tiles than workgroups (5000 x 768: 474 tiles), and the ReLU epilogue of `frame_transform`
    (relu(x W^T + b) saved, + residual: model/layers.py:86-93, model/model.py:211-212)."""
    _ws_against_4wave(HF, Lb, M, N, K, cfg, colsum=(M, K) == (1920, 3072))    # + the gelu' column sums (fold in the spare LDS region behind a six-deep ring)
    dtype = torch.bfloat16
    x, w, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=0.05), rnd(N, seed=3)
    res = rnd(M, N, dtype=dtype, seed=4)
    outs = []
    for c in (cfg, 8):
        Lb.lib().hero_gemm_force_config(c)
        try:
            a1, a2 = (torch.empty((M, N), dtype=dtype, device=x.device) for _ in range(2))
            outs.append([HF.k_linear(x, w, b, act=Lb.ACT_RELU, aux=a1), a1, HF.k_linear(x, w, b, act=Lb.ACT_RELU, aux=a2, residual=res), a2])
        finally:
            Lb.lib().hero_gemm_force_config(-1)
    ref = torch.relu(x.float() @ w.float().t() + b)
    sc = math.sqrt(K) * 0.05
    close(outs[0][0], ref, dtype, scale=sc)
    close(outs[0][1], ref, dtype, scale=sc)
    close(outs[0][3], ref, dtype, scale=sc)
    close(outs[0][2], ref + res.float(), dtype, scale=sc + 1)
    for a_, b_ in zip(*outs):
        close(a_, b_, dtype, scale=sc + 1)
    assert ((outs[0][1] == 0) != (outs[1][1] == 0)).float().mean().item() < 1e-3      # the same elements are clipped


@pytest.mark.parametrize("M,N,K", [(32, 768, 768), (32, 1920, 768), (5, 772, 260), (1, 8, 256), (32, 1000, 3072)])
def test_gemm_skinny_f32(HF, Lb, M, N, K):
    """The fp32 Linear layers of the loss head on <= 32 rows (queries): the skinny kernel (one wave per two output columns,
    fixed-order reduction) against fp32 torch and the MFMA tiles, with and without bias, odd column counts, a ragged last
    256-k block; bit-reproducible."""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
    ref = x.double() @ w.double().t()
    y0, y1 = HF.k_linear(x, w), HF.k_linear(x, w, b)
    torch.testing.assert_close(y0.double(), ref, rtol=1e-5, atol=1e-5 * math.sqrt(K))
    torch.testing.assert_close(y1.double(), ref + b.double(), rtol=1e-5, atol=1e-5 * math.sqrt(K))
    assert torch.equal(HF.k_linear(x, w, b), y1)
    Lb.lib().hero_gemm_force_config(3)
    try:
        old = HF.k_linear(x, w, b)
    finally:
        Lb.lib().hero_gemm_force_config(-1)
    torch.testing.assert_close(y1, old, rtol=1e-4, atol=1e-4 * math.sqrt(K))


def test_gemm_small_m_heuristic_takes_the_64_row_tiles(HF, Lb):
    """hero_gemm's own choice for M = 1920, N = 768 (K >= 512) is the 64 x 128 geometry: bit-equal to forcing it
    (`hero_gemm_force_config(8)` is the way back to the 4-wave kernels)."""
    dtype = torch.bfloat16
    for K in (768, 3072):
        x, w, b = rnd(1920, K, dtype=dtype, seed=1), rnd(768, K, dtype=dtype, seed=2, scale=0.05), rnd(768, seed=3)
        y = HF.k_linear(x, w, b)
        Lb.lib().hero_gemm_force_config(13)
        try:
            y13 = HF.k_linear(x, w, b)
        finally:
            Lb.lib().hero_gemm_force_config(-1)
        assert torch.equal(y, y13)


def test_gemm_wave_specialised_large_row_counts(HF, Lb):
    """config 5 sizes (long videos filling the HBM): 1.45 M rows - the activations exceed 2^32 elements / 2^31 bytes.  The
    wave-specialised K,K kernels address everything relative to the tile (panels, residual and saved pre-activation
    reads, stores) and must agree with the 4-wave kernels on the last rows too."""
    dtype = torch.bfloat16
    M = 1_450_000
    g = torch.Generator(device="cuda").manual_seed(5)

    def big(rows, cols, scale=1.0):
        return (torch.randn(rows, cols, device="cuda", generator=g, dtype=torch.float32) * scale).to(dtype) if rows * cols < 2 ** 28 else \
            torch.cat([(torch.randn(rows // 10, cols, device="cuda", generator=g, dtype=torch.float32) * scale).to(dtype) for _ in range(10)], 0)

    for N, K, kind in ((768, 3072, "res"), (3072, 768, "gelu_bwd")):
        x, w = big(M, K), big(N, K, 0.05)
        side = big(M, N)                                             # residual / saved pre-activation
        assert x.numel() > 2 ** 31 or side.numel() > 2 ** 32
        outs = []
        for cfg in (-1, 8):                                          # heuristic (wave-specialised) / never
            Lb.lib().hero_gemm_force_config(cfg)
            try:
                if kind == "res":
                    outs.append(HF.k_linear(x, w, residual=side))
                else:
                    outs.append(HF.k_dgrad_t(x, w, act=Lb.ACT_GELU_BWD, aux=side))
            finally:
                Lb.lib().hero_gemm_force_config(-1)
        for sl in (slice(0, 4096), slice(M // 2, M // 2 + 4096), slice(M - 4096, M)):
            close(outs[0][sl], outs[1][sl], dtype, scale=math.sqrt(K) * 0.05 + 1)
        del x, w, side, outs
        torch.cuda.empty_cache()


def _ws_against_4wave(HF, Lb, M, N, K, ws_cfg, colsum=True):
    dtype = torch.bfloat16
    x, w, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=0.05), rnd(N, seed=3)
    res, u = rnd(M, N, dtype=dtype, seed=4), rnd(M, N, dtype=dtype, seed=6)
    drop = HF.RNG.make(0.1, True, x.device)
    cs = [rnd(N, seed=7), None]
    cs[1] = cs[0].clone()

    def run(cfg, k):
        Lb.lib().hero_gemm_force_config(cfg)
        try:
            aux = torch.empty((M, N), dtype=dtype, device=x.device)
            return [HF.k_linear(x, w, b),
                    HF.k_linear(x, w, b, residual=res, drop=drop),
                    HF.k_linear(x, w, b, act=Lb.ACT_GELU, aux=aux), aux,
                    HF.k_linear(x, w),
                    HF.k_linear(x, w, residual=res),
                    HF.k_dgrad_t(x, w, act=Lb.ACT_GELU_BWD, aux=u, colsum=cs[k] if colsum else None)]
        finally:
            Lb.lib().hero_gemm_force_config(-1)

    got, old = run(ws_cfg, 0), run(8, 1)
    ref = x.float() @ w.float().t()
    sc = math.sqrt(K) * 0.05
    close(got[0], ref + b, dtype, scale=sc)
    close(got[4], ref, dtype, scale=sc)
    close(got[5], ref + res.float(), dtype, scale=sc + 1)
    close(got[3], ref + b, dtype, scale=sc)
    close(got[2], torch.nn.functional.gelu(ref + b), dtype, scale=sc)
    for a_, b_ in zip(got, old):                      # same inputs, same mask: equal up to bf16 rounding of the output
        close(a_, b_, dtype, scale=sc + 1)
    kept = (got[1].float() - res.float()).abs() > 1e-6   # dropout zeroes the same elements in both families
    kept_old = (old[1].float() - res.float()).abs() > 1e-6
    assert (kept != kept_old).float().mean().item() < 1e-3
    if colsum:
        torch.testing.assert_close(cs[0], cs[1], rtol=2e-2, atol=0.05 * math.sqrt(M))


@pytest.mark.parametrize("rows,n_out,n_in", [(12000, 768, 768), (12040, 3072, 768), (4100, 768, 3072), (520, 200, 136),
                                             (12000, 2304, 768)])
def test_gemm_wave_specialised_wgrad(HF, Lb, rows, n_out, n_in):
    """dW += dY^T X on the wave-specialised O,O kernel: transpose reads, reduction split with fp32
    atomics, reduction tails (rows % 64 != 0 -> out-of-range rows read as zeros), accumulate into C."""
    dtype = torch.bfloat16
    dy, x = rnd(rows, n_out, dtype=dtype, seed=1), rnd(rows, n_in, dtype=dtype, seed=2)
    ref = dy.float().t() @ x.float()
    Lb.lib().hero_gemm_force_config(9)
    try:
        dW = HF.k_wgrad(dy, x)
        acc = torch.ones(n_out, n_in, device=x.device)
        HF.k_wgrad(dy, x, out=acc, beta=1.0)
        half = HF.k_wgrad(dy, x, col0=n_out // 2 // 8 * 8, ncols=n_out - n_out // 2 // 8 * 8)
    finally:
        Lb.lib().hero_gemm_force_config(-1)
    tol = dict(rtol=1e-4, atol=1e-3 * math.sqrt(rows))
    torch.testing.assert_close(dW, ref, **tol)
    torch.testing.assert_close(acc, ref + 1.0, **tol)
    torch.testing.assert_close(half, ref[n_out // 2 // 8 * 8:], **tol)


@pytest.mark.parametrize("rows", [12000, 12040, 1920, 520])
def test_wgrad_group_stream_k(HF, Lb, rows):
    """hero_wgrad_group: the four weight gradients of a BertLayer (QKV 2304x768, out 768x768, FFN1 3072x768, FFN2
    768x3072) over the same rows in ONE stream-K launch, accumulated into existing values; the reduction tail
    (rows % 64 != 0), a column-sliced dY, and the small-problem fallback (one hero_gemm per problem)."""
    dtype = torch.bfloat16
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    dys = [rnd(rows, n, dtype=dtype, seed=10 + i) for i, (n, _) in enumerate(shapes)]
    xs = [rnd(rows, k, dtype=dtype, seed=20 + i) for i, (_, k) in enumerate(shapes)]
    outs = [torch.full((n, k), 0.5, device="cuda") for n, k in shapes]
    probs = (Lb.WgradProblem * 4)()
    for i, (n, k) in enumerate(shapes):
        probs[i] = Lb.WgradProblem(dys[i].data_ptr(), xs[i].data_ptr(), outs[i].data_ptr(), n, k, n, k, k, 4)
    Lb.check(Lb.lib().hero_wgrad_group(probs, 4, rows, Lb.BF16, Lb.stream()))
    for i in range(4):
        ref = dys[i].float().t() @ xs[i].float() + 0.5
        torch.testing.assert_close(outs[i], ref, rtol=1e-4, atol=1e-3 * math.sqrt(rows))
    # two problems, the first one a column slice of a wider dY (the per-tensor path of the fused QKV gradient)
    out2 = [torch.zeros(768, 768, device="cuda"), torch.zeros(768, 3072, device="cuda")]
    p2 = (Lb.WgradProblem * 2)()
    p2[0] = Lb.WgradProblem(dys[0].data_ptr() + 768 * 2, xs[0].data_ptr(), out2[0].data_ptr(), 768, 768, 2304, 768, 768, 4)
    p2[1] = Lb.WgradProblem(dys[3].data_ptr(), xs[3].data_ptr(), out2[1].data_ptr(), 768, 3072, 768, 3072, 3072, 4)
    Lb.check(Lb.lib().hero_wgrad_group(p2, 2, rows, Lb.BF16, Lb.stream()))
    torch.testing.assert_close(out2[0], dys[0].float()[:, 768:1536].t() @ xs[0].float(), rtol=1e-4, atol=1e-3 * math.sqrt(rows))
    torch.testing.assert_close(out2[1], dys[3].float().t() @ xs[3].float(), rtol=1e-4, atol=1e-3 * math.sqrt(rows))


def _batch_problems(Lb, rows, layers, shapes, ragged_cols=False):
    dtype = torch.bfloat16
    dys, xs, outs = [], [], []
    for l in range(layers):
        for i, (n, k) in enumerate(shapes):
            dys.append(rnd(rows, n, dtype=dtype, seed=100 + 10 * l + i))
            xs.append(rnd(rows, k, dtype=dtype, seed=200 + 10 * l + i))
            outs.append(torch.full((n, k), 0.25, device="cuda"))
    n = len(dys)
    probs = (Lb.WgradProblem * n)()
    for i in range(n):
        m_, k_ = outs[i].shape
        probs[i] = Lb.WgradProblem(dys[i].data_ptr(), xs[i].data_ptr(), outs[i].data_ptr(), m_, k_, m_, k_, k_, 4)
    return dys, xs, outs, probs


def _run_batch(Lb, probs, n, rows):
    import numpy as np
    buf = np.zeros(8 + 8 * 256 * 16, dtype=np.int32)
    words = Lb.lib().hero_wgrad_batch_plan(probs, n, rows, buf.ctypes.data, buf.size)
    assert words > 8, words
    plan = torch.from_numpy(buf[:words].copy()).cuda()
    Lb.check(Lb.lib().hero_wgrad_batch(probs, n, rows, Lb.BF16, plan.data_ptr(), words, Lb.stream()))
    torch.cuda.synchronize()
    return buf[:words]


@pytest.mark.parametrize("rows,layers", [(1920, 3), (1000, 6), (12000, 6), (8192, 1), (8200, 5)])
def test_wgrad_batch_whole_tiles(HF, Lb, rows, layers):
    """hero_wgrad_batch: the weight gradients of ALL layers of an encoder over the same rows in one launch - whole
    192 x 192 tiles in full rounds (plain fp32 read-add-write), the last partial round cut into k-slices with ordered
    atomics.  (1920, 3) = the Temporal Transformer (576 tiles: 2 rounds + 64 tail tiles - round 6: WHOLE tiles, a 30-step
    reduction does not pay for ordered atomics, profiles/r06_wgrad_ceiling.txt; rounds 3-5: 4 slices), (12000, 6) = the
    cross-modal stack of the benched step (1152 tiles: 4 rounds + 128 tiles x 2 slices), 1000 rows: a reduction tail;
    (8192, 1) / (8200, 5): 192 tail tiles, >= 128 k-steps = ONE BertLayer (what config 5's queue cap flushes at a time) - round 4's balanced
    tail: a 3/4 slice per tile on 192 workgroups, the quarters packed three to a workgroup on the other 64.
    Accumulates into existing values; the result is BIT-REPRODUCIBLE run to run."""
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    dys, xs, outs, probs = _batch_problems(Lb, rows, layers, shapes)
    n = len(dys)
    # bias gradients from the dY panels the kernel streams (every problem but the first, which checks the NULL case)
    dbs = [torch.full((o.shape[0],), 0.125, device="cuda") for o in outs]
    for i in range(1, n):
        probs[i].dbias = dbs[i].data_ptr()
    plan = _run_batch(Lb, probs, n, rows)
    for i in range(1, n):
        torch.testing.assert_close(dbs[i], dys[i].float().sum(0) + 0.125, rtol=1e-4, atol=2e-4 * rows)
    assert float((dbs[0] - 0.125).abs().max()) == 0.0
    first_db = [d.clone() for d in dbs]
    assert plan[6] == layers * 192
    if (rows, layers) in ((1920, 3), (1000, 6)):
        assert plan[7] == 1                     # short reductions: the tail round keeps whole tiles (no atomics at all)
    if (rows, layers) == (12000, 6):
        assert plan[7] == 2                     # the long one is sliced: 128 tail tiles x 2 slices, ordered atomics
    if layers in (1, 5):                        # 192 tail tiles (24 per XCD on 32 workgroups): big slices + packed remainders
        assert plan[2] > layers * 192 // 256 + 1
    first = [o.clone() for o in outs]
    for i in range(n):
        ref = dys[i].float().t() @ xs[i].float() + 0.25
        torch.testing.assert_close(outs[i], ref, rtol=1e-4, atol=1e-3 * math.sqrt(rows))
    for rep in range(2):
        for o in outs:
            o.fill_(0.25)
        for d in dbs:
            d.fill_(0.125)
        _run_batch(Lb, probs, n, rows)
        for o, f in zip(outs, first):
            assert torch.equal(o, f), "hero_wgrad_batch is not bit-reproducible"
        for d, f in zip(dbs, first_db):
            assert torch.equal(d, f), "hero_wgrad_batch bias gradients are not bit-reproducible"
    # the slice-order flags are back to zero: a second launch right behind the first one accumulates once more
    _run_batch(Lb, probs, n, rows)
    for i in (0, n - 1):
        ref = 2 * (dys[i].float().t() @ xs[i].float()) + 0.25
        torch.testing.assert_close(outs[i], ref, rtol=1e-4, atol=2e-3 * math.sqrt(rows))


def test_wgrad_batch_ragged_shapes_and_small_groups(HF, Lb):
    """Output shapes that are not multiples of the 192 x 192 tile (4352-wide projections, a 1000 x 776 weight), a
    column-sliced dY, and groups smaller than one round of the chip (sliced tails only)."""
    import numpy as np
    rows = 1920
    shapes = [(768, 4352), (1000, 776), (768, 4352), (2304, 768), (768, 3072), (3072, 768)]
    dys, xs, outs, probs = _batch_problems(Lb, rows, 1, shapes)
    wide = rnd(rows, 2304, dtype=torch.bfloat16, seed=77)
    probs[3] = Lb.WgradProblem(wide.data_ptr() + 768 * 2, xs[3].data_ptr(), outs[3].data_ptr(), 768, 768, 2304, 768, 768, 4)
    outs[3].fill_(0.25)
    _run_batch(Lb, probs, len(shapes), rows)
    for i in range(len(shapes)):
        dy = wide[:, 768:1536] if i == 3 else dys[i]
        ref = dy.float().t() @ xs[i].float() + 0.25
        got = outs[i][:768] if i == 3 else outs[i]
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-3 * math.sqrt(rows))
    assert float((outs[3][768:] - 0.25).abs().max()) == 0.0          # rows of dW outside the problem are untouched
    buf = np.zeros(8 + 8 * 256 * 16, dtype=np.int32)
    # a handful of tiles (16): round 3 left such groups to the stream-K group kernel (fp32 atomics in any order); round 4
    # runs them as a sliced tail too, ordered atomics - bit-reproducible like the large groups
    small = (Lb.WgradProblem * 1)(probs[3])
    assert Lb.lib().hero_wgrad_batch_plan(small, 1, rows, buf.ctypes.data, buf.size) > 8
    again = []
    for rep in range(3):
        outs[3].fill_(0.25)
        plan = _run_batch(Lb, small, 1, rows)
        assert plan[6] == 16
        again.append(outs[3][:768].clone())
    torch.testing.assert_close(again[0], wide[:, 768:1536].float().t() @ xs[3].float() + 0.25, rtol=1e-4, atol=1e-3 * math.sqrt(rows))
    assert torch.equal(again[0], again[1]) and torch.equal(again[0], again[2])
    # a group smaller than one round of the chip (two 4352-wide projections: 184 tiles) runs as a sliced tail only
    two = (Lb.WgradProblem * 2)(probs[0], probs[2])
    for o in (outs[0], outs[2]):
        o.fill_(0.25)
    plan = _run_batch(Lb, two, 2, rows)
    assert plan[2] == 1 and plan[6] == 184
    for i in (0, 2):
        torch.testing.assert_close(outs[i], dys[i].float().t() @ xs[i].float() + 0.25, rtol=1e-4, atol=1e-3 * math.sqrt(rows))


def test_deferred_weight_gradients_are_flushed_with_the_backward_pass(HF, Lb):
    """Sink-accumulated weight gradients are queued during backward and launched in groups; whoever looks at a
    .grad after backward() sees the complete sum, whatever the number of queued problems."""
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(n, 768, device="cuda") * 0.02) for n in (768, 1536, 768, 768, 2304)]
    x = rnd(4096, 768, dtype=torch.bfloat16, seed=1)
    y = x
    for w in ws:                                       # five chained Linear layers: 4 + 1 queued problems
        y = HF.linear(y, w)
        y = y[:, :768].contiguous()
    y.float().pow(2).sum().backward()
    ref_ws = [w.detach().clone().requires_grad_(True) for w in ws]
    yr = x.float()
    for w in ref_ws:
        yr = (yr @ w.to(torch.bfloat16).float().t())[:, :768]
        yr = yr.to(torch.bfloat16).float()
    yr.pow(2).sum().backward()
    for w, r in zip(ws, ref_ws):
        assert w.grad is not None
        torch.testing.assert_close(w.grad, r.grad, rtol=5e-2, atol=5e-2 * float(r.grad.abs().max()))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(12000, 3072, 768), (1920, 3072, 768), (300, 136, 64)])
def test_gemm_gelu_saved_derivative(HF, Lb, dtype, M, N, K):
    """HERO_ACT_GELU_DG / HERO_ACT_MUL_AUX (round 4): BertIntermediate's GEMM saves gelu'(pre-activation) instead of the
    pre-activation, the gradient GEMM multiplies by the saved tensor - on the wave-specialised, the 4-wave and the
    generic epilogues, against fp32 torch (erf form, model/layers.py:16-25)."""
    if dtype == torch.float32 and M > 2000:
        pytest.skip("large shape only in the product dtype")
    x, w, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=0.05), rnd(N, seed=3)
    aux = torch.empty((M, N), dtype=dtype, device="cuda")
    y = HF.k_linear(x, w, b, act=Lb.ACT_GELU_DG, aux=aux)
    pre = x.float() @ w.float().t() + b
    pre.requires_grad_(True)
    ref = torch.nn.functional.gelu(pre)
    (dref,) = torch.autograd.grad(ref.sum(), pre)
    sc = math.sqrt(K) * 0.05
    close(y, ref.detach(), dtype, scale=sc)
    close(aux, dref, dtype, scale=1.0)
    # the same tensors through the old pair give the same forward output bit for bit
    aux2 = torch.empty_like(aux)
    y2 = HF.k_linear(x, w, b, act=Lb.ACT_GELU, aux=aux2)
    assert torch.equal(y, y2)
    dy, wt = rnd(M, K, dtype=dtype, seed=5), rnd(N, K, dtype=dtype, seed=6, scale=0.05)     # [M, K] @ [N, K]^T -> [M, N]
    du = HF.k_dgrad_t(dy, wt, act=Lb.ACT_MUL_AUX, aux=aux)
    want = (dy.float() @ wt.float().t()) * aux.float()
    close(du, want, dtype, scale=sc)


@pytest.mark.parametrize("M,N,K,cfgs", [(12000, 3072, 768, (-1, 9, 10, 8)), (1920, 3072, 768, (-1, 10, 8)), (1000, 768, 3072, (-1, 13, 0, 1, 3)),
                                          (300, 136, 64, (-1,))])
def test_gemm_column_sums_as_per_tile_partials(HF, Lb, M, N, K, cfgs):
    """HeroGemmEpilogue.colsum_partial (round 5): the gelu' epilogue writes each output tile's column sums to row (tile row / 64)
    of a [ceil(M / 64), N] table (zeros in the rows of the tile's other 64-row blocks) instead of fp32 atomics into [N] - on the
    wave-specialised geometries and the 4-wave ones; the table's column sums are the bias gradient (fp32 torch), the output is
    the bits of the atomic variant's, and two runs give identical tables."""
    dtype = torch.bfloat16
    dy, wt = rnd(M, K, dtype=dtype, seed=5), rnd(N, K, dtype=dtype, seed=6, scale=0.05)
    aux = rnd(M, N, dtype=dtype, seed=7)
    want = ((dy.float() @ wt.float().t()) * aux.float())
    nb = -(-M // 64)
    for cfg in cfgs:
        Lb.lib().hero_gemm_force_config(cfg)
        try:
            part = torch.full((nb, N), float("nan"), device="cuda")
            du = HF.k_dgrad_t(dy, wt, act=Lb.ACT_MUL_AUX, aux=aux, colsum=part, colsum_partial=True)
            part2 = torch.full((nb, N), float("nan"), device="cuda")
            HF.k_dgrad_t(dy, wt, act=Lb.ACT_MUL_AUX, aux=aux, colsum=part2, colsum_partial=True)
            atom = torch.zeros(N, device="cuda")
            du3 = HF.k_dgrad_t(dy, wt, act=Lb.ACT_MUL_AUX, aux=aux, colsum=atom)
        finally:
            Lb.lib().hero_gemm_force_config(-1)
        assert torch.isfinite(part).all(), cfg                       # every row of the table was written
        assert torch.equal(part, part2) and torch.equal(du, du3), cfg
        got = HF.k_colsum(part)
        ref = du.float().sum(0) if False else want.sum(0)
        torch.testing.assert_close(got, ref, rtol=2e-2, atol=2e-2 * float(want.abs().max()) * math.sqrt(M)), cfg
        torch.testing.assert_close(got, atom, rtol=1e-4, atol=1e-4 * float(atom.abs().max()) + 1e-6)   # same values, other order
        assert (part != 0).any(1).sum() <= nb and (part[0] != 0).any()


@pytest.mark.parametrize("M,N,K", [(1440, 50272, 768), (360, 8200, 768), (1440, 16384, 136)])
def test_dgrad_long_reduction_small_output(HF, Lb, M, N, K):
    """functional._dgrad_long_reduction: dx = dy @ Wt^T with a vocabulary-long reduction (configs[3]: the MLM decoder's
    input gradient, 50272 % 64 = 32) - split-K on the direct-to-LDS kernels + the ragged rest + a cast, against fp32."""
    dy = rnd(M, N, dtype=torch.bfloat16, seed=1, scale=0.05)
    wt = rnd(K, N, dtype=torch.bfloat16, seed=2, scale=0.05)
    dx = HF.k_dgrad_t(dy, wt)
    assert dx.dtype == torch.bfloat16 and dx.shape == (M, K)
    ref = dy.float() @ wt.float().t()
    close(dx, ref, torch.bfloat16, scale=float(ref.abs().max()))
    # round 5: the splits write slabs (HeroGemmEpilogue.split_stride) that ONE fold adds in slab order - no
    # atomics, so the result is the same bits run after run
    for _ in range(3):
        assert torch.equal(HF.k_dgrad_t(dy, wt), dx)


def test_gemm_split_slabs_and_fold(HF, Lb):
    """split_k with split_stride: split s writes its own fp32 slab, hero_gemm_splits says how many there are,
    hero_fold_slabs sums them in slab order (fp32 and bf16 outputs) - equal to the unsplit GEMM up to summation order,
    bit-reproducible; bad strides are refused."""
    import ctypes as C
    M, N, K = 256, 512, 64 * 37                     # 37 k-tiles: 8 asked -> ceil(37 / 5) = 8 ranges of 5 (last: 2)
    a = rnd(M, K, dtype=torch.bfloat16, seed=1)
    b = rnd(N, K, dtype=torch.bfloat16, seed=2, scale=0.05)
    n = Lb.lib().hero_gemm_splits(K, 8, Lb.BF16)
    assert n == 8 and Lb.lib().hero_gemm_splits(K, 1, Lb.BF16) == 1 and Lb.lib().hero_gemm_splits(64, 8, Lb.BF16) == 1
    slabs = torch.full((n, M, N), float("nan"), device="cuda")
    HF.k_gemm(a, b, slabs, M, N, K, K, K, N, Lb.LAYOUT_K, Lb.LAYOUT_K, Lb.BF16, out_f32=True, split_k=8, split_stride=M * N)
    ref = a.float() @ b.float().t()
    assert torch.isfinite(slabs).all()
    torch.testing.assert_close(slabs.sum(0), ref, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(slabs[0], a[:, :320].float() @ b[:, :320].float().t(), rtol=1e-4, atol=1e-3)    # slab 0 = k-tiles 0..4
    for dt_ in (torch.float32, torch.bfloat16):
        out = torch.empty((M, N), dtype=dt_, device="cuda")
        Lb.check(Lb.lib().hero_fold_slabs(Lb.ptr(slabs), n, M * N, Lb.ptr(out), M * N, Lb.dt(out), Lb.stream()))
        want = slabs[0].clone()
        for k in range(1, n):
            want += slabs[k]
        assert torch.equal(out, want.to(dt_))
    epi = Lb.GemmEpilogue(act=Lb.ACT_NONE, out_f32=1, split_k=8, dropout=Lb.no_dropout(), split_stride=M * N - 8)      # slab too small
    assert Lb.lib().hero_gemm(Lb.ptr(a), Lb.ptr(b), Lb.ptr(slabs), M, N, K, K, K, N, 0, 0, Lb.BF16, C.byref(epi), Lb.stream()) != 0
    epi = Lb.GemmEpilogue(act=Lb.ACT_NONE, out_f32=1, split_k=1, dropout=Lb.no_dropout(), split_stride=M * N)          # stride without a split
    assert Lb.lib().hero_gemm(Lb.ptr(a), Lb.ptr(b), Lb.ptr(slabs), M, N, K, K, K, N, 0, 0, Lb.BF16, C.byref(epi), Lb.stream()) != 0


def test_box_probes_report_sane_peaks(Lb):
    """hero_probe_mfma / hero_probe_hbm (bench.py's `box` block): an MI355X delivers ~2.3-2.5 PFLOP/s of dense bf16 MFMA at
    ~2.1-2.4 GHz and 4-6.5 TB/s of streaming copy; generous brackets, the point is that the numbers are physical."""
    import ctypes as C
    n = 1 << 28
    a = torch.zeros(n // 4, device="cuda")
    b = torch.empty_like(a)
    tf, ghz, cp, rd = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    Lb.check(Lb.lib().hero_probe_mfma(Lb.ptr(b), n, C.byref(tf), C.byref(ghz), Lb.stream()))
    Lb.check(Lb.lib().hero_probe_hbm(Lb.ptr(a), Lb.ptr(b), n, C.byref(cp), C.byref(rd), Lb.stream()))
    print("box: %.0f TFLOP/s at %.2f GHz, copy %.0f GB/s, read %.0f GB/s" % (tf.value, ghz.value, cp.value, rd.value))
    assert 1200 < tf.value < 2700 and 1.1 < ghz.value < 2.6 and abs(ghz.value * 1024 * 1024 / 1000 - tf.value) < 1.0   # TF/s = 1024 SIMDs x 1024 flop/clk x GHz
    assert 2000 < cp.value < 8200 and 1500 < rd.value < 8200
    assert torch.equal(b[4:], a[4:])                        # the copy pass copied (the read pass may touch dst[0..3])
    assert Lb.lib().hero_probe_hbm(Lb.ptr(a), Lb.ptr(b), 1 << 20, C.byref(cp), C.byref(rd), Lb.stream()) != 0     # too small: inside the Infinity Cache
    assert Lb.lib().hero_probe_mfma(Lb.ptr(b), 64, C.byref(tf), C.byref(ghz), Lb.stream()) != 0


def test_deferred_weight_gradients_same_destination_twice(HF, Lb):
    """a parameter used twice in one backward pass with equal row counts (tied / shared nn.Linear) queues two
    problems with the SAME dW and dbias.  Inside one hero_wgrad_batch launch both would sit in the same round on different
    workgroups and the plain read-add-write of the full-round tiles would lose one of them; wgrad_flush cuts the batch."""
    rows, n_out, n_in = 4096, 1536, 3072                 # 128 tiles per use: both uses would share one round of 256
    dys = [rnd(rows, n_out, dtype=torch.bfloat16, seed=1 + k) for k in range(2)]
    xs = [rnd(rows, n_in, dtype=torch.bfloat16, seed=5 + k) for k in range(2)]
    dW = torch.ones(n_out, n_in, device="cuda")
    db = torch.ones(n_out, device="cuda")

    class Use(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t, k):
            ctx.k = k
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            HF.k_wgrad(dys[ctx.k], xs[ctx.k], out=dW, beta=1.0, dbias=db, dbias_done=lambda: None)
            return g, None

    t = torch.zeros(4, device="cuda", requires_grad=True)
    Use.apply(Use.apply(t, 0), 1).sum().backward()
    torch.cuda.synchronize()
    ref = 1.0 + sum(d.float().t() @ x.float() for d, x in zip(dys, xs))
    refb = 1.0 + sum(d.float().sum(0) for d in dys)
    torch.testing.assert_close(dW, ref, rtol=1e-4, atol=1e-3 * math.sqrt(rows))
    torch.testing.assert_close(db, refb, rtol=1e-4, atol=1e-3 * math.sqrt(rows))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(700, 776, 768), (12000, 3072, 768), (130, 64, 64)])
def test_gemm_output_column_sums(HF, Lb, dtype, M, N, K):
    """epilogue.colsum: the column sums of the stored output (here dH = (dY W) * gelu'(u), i.e. the FFN1
    bias gradient) come out of the GEMM that produces it, accumulated into the existing values."""
    if dtype == torch.float32 and M > 1000:
        pytest.skip("large shape only in the product dtype")
    dy = rnd(M, K, dtype=dtype, seed=1)
    wt = rnd(N, K, dtype=dtype, seed=2, scale=0.05)            # [out features of dx, reduction]
    u = rnd(M, N, dtype=dtype, seed=3)
    cs = rnd(N, seed=4)
    cs0 = cs.clone()
    dx = HF.k_dgrad_t(dy, wt, act=Lb.ACT_GELU_BWD, aux=u, colsum=cs)
    ref_dx = HF.k_dgrad_t(dy, wt, act=Lb.ACT_GELU_BWD, aux=u)
    torch.testing.assert_close(dx, ref_dx, rtol=0, atol=0)
    want = cs0 + ref_dx.float().sum(0)
    torch.testing.assert_close(cs, want, rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=0.05 * math.sqrt(M) if dtype == torch.bfloat16 else 1e-3)


def test_gemm_transpose_detecting(HF, Lb):
    """A = I with an asymmetric B catches row/column swaps in the MFMA fragment maps."""
    for dtype in DT:
        n = 128
        eye = torch.eye(n, dtype=dtype).cuda()
        b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(dtype).cuda()
        y = HF.k_linear(eye, b)          # I @ b^T
        torch.testing.assert_close(y.float(), b.float().t())
        y2 = HF.k_dgrad(eye, b)          # I @ b
        torch.testing.assert_close(y2.float(), b.float())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,cols", [(37, 128), (1000, 768), (64, 4352), (5, 1536)])
def test_layernorm(HF, dtype, rows, cols):
    x = rnd(rows, cols, dtype=dtype, seed=1) * 2 + 0.5
    g, b = rnd(cols, seed=2) * 0.1 + 1, rnd(cols, seed=3) * 0.1
    dy = rnd(rows, cols, dtype=dtype, seed=4)
    for eps in (1e-12, 1e-5):
        y, mean, rstd, _ = HF.k_ln_fwd(x, g, b, eps, dtype, rows, cols)
        xf = x.float().clone().requires_grad_(True)
        gf, bf = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref = torch.nn.functional.layer_norm(xf, (cols,), gf, bf, eps)
        close(y, ref, dtype, scale=3)
        ref.backward(dy.float())
        dx, _, dg, db = HF.k_ln_bwd(x, dy, g, mean, rstd)
        close(dx, xf.grad, dtype, scale=3)
        torch.testing.assert_close(dg, gf.grad, rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=0.3 if dtype == torch.bfloat16 else 1e-3)
        torch.testing.assert_close(db, bf.grad, rtol=1e-4, atol=1e-3)


def test_layernorm_fp32_in_bf16_out_and_tables(HF):
    rows, cols = 300, 768
    x = rnd(rows, cols, seed=1)
    g, b = rnd(cols, seed=2) * 0.1 + 1, rnd(cols, seed=3) * 0.1
    t0, t1 = rnd(50, cols, seed=4), rnd(7, cols, seed=5)
    i0 = torch.randint(0, 50, (rows,), generator=torch.Generator().manual_seed(6)).int().cuda()
    i1 = torch.randint(0, 7, (rows,), generator=torch.Generator().manual_seed(7)).int().cuda()
    for od in DT:
        y, mean, rstd, pre = HF.k_ln_fwd(x, g, b, 1e-5, od, rows, cols, tabs=[t0, t1, t1[3:4]],
                                         idxs=[i0, i1, None], want_pre=True)
        s = x + t0[i0.long()] + t1[i1.long()] + t1[3]
        close(pre, s, od)
        close(y, torch.nn.functional.layer_norm(s, (cols,), g, b, 1e-5), od, scale=3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("S,L,H", [(3, 9, 2), (5, 24, 12), (2, 60, 12), (2, 100, 3), (1, 130, 2), (4, 15, 12),
                                   (3, 32, 4), (2, 33, 2), (2, 64, 3), (41, 24, 12), (1, 1, 1), (2, 65, 2),
                                   (3, 128, 2), (2, 129, 3), (2, 200, 4), (3, 256, 12)])
def test_attention_fwd_bwd(HF, Lb, dtype, S, L, H):
    """fp32: exact-f32 VALU kernels (forward L <= 256; backward up to hero_attention_max_len, 148 rows);
    bf16: matrix-core kernels for every L <= 256."""
    D = H * 64
    qkv = rnd(S * L, 3 * D, dtype=dtype, seed=1)
    lens = [max(1, L - 3 * i) for i in range(S)]
    m = torch.zeros(S, L)
    for i, n in enumerate(lens):
        m[i, :n] = 1
    if S > 1:
        m[1, 0] = 0                                   # subtitle without frames: first key masked
    madd = ((1 - m) * -10000.0).cuda()
    dctx = rnd(S * L, D, dtype=dtype, seed=2)
    ctx, saved = HF.k_attn_fwd(qkv, madd, S, L, H)
    q = qkv.float().requires_grad_(True)
    qq, kk, vv = [t.reshape(S, L, H, 64).permute(0, 2, 1, 3) for t in q.split(D, dim=1)]
    sc = qq @ kk.transpose(-1, -2) / 8.0 + madd[:, None, None, :]
    pr = torch.softmax(sc, -1)
    ref = (pr @ vv).permute(0, 2, 1, 3).reshape(S * L, D)
    ptol = dict(rtol=1e-4 if dtype == torch.float32 else 2e-2, atol=1e-5 if dtype == torch.float32 else 2e-3)
    stats_mode = saved.dim() == 1
    assert stats_mode == bool(Lb.lib().hero_attention_stats_ok(Lb.dt(qkv), L))
    if stats_mode:
        # bf16, L <= 64: the forward keeps (row max, 1 / row sum) instead of the probabilities ...
        st = saved.view(S, H, L, 2)
        mx = sc.max(-1).values
        torch.testing.assert_close(st[..., 0], mx.detach(), rtol=2e-2, atol=5e-2)
        torch.testing.assert_close(1.0 / st[..., 1], torch.exp(sc - mx[..., None]).sum(-1).detach(), rtol=2e-2, atol=2e-3)
        HF.ATTN_SAVE_PROBS = True                     # ... the same kernels with the probabilities saved
        try:
            ctx_p, probs = HF.k_attn_fwd(qkv, madd, S, L, H)
        finally:
            HF.ATTN_SAVE_PROBS = False
        assert probs.dim() == 4 and torch.equal(ctx_p, ctx)
    else:
        probs = saved
    torch.testing.assert_close(probs, pr, **ptol)
    close(ctx, ref, dtype)
    if L > Lb.lib().hero_attention_max_len(Lb.dt(qkv), 1):
        return                                                    # forward-only length in this dtype
    ref.backward(dctx.float())
    dqkv = HF.k_attn_bwd(qkv, probs, dctx, S, L, H, ctx=ctx)     # ctx: 64 < L <= 256 in bf16 runs on the matrix cores
    close(dqkv, q.grad, dtype, scale=2)
    if stats_mode:                                                # rebuilt from q, k + statistics: the SAME bits
        assert torch.equal(HF.k_attn_bwd(qkv, saved, dctx, S, L, H, ctx=ctx, mask_add=madd), dqkv)
    if dtype == torch.bfloat16 and L > 64:                        # ... and agrees with the fp32-VALU kernels (no ctx)
        close(dqkv, HF.k_attn_bwd(qkv, probs, dctx, S, L, H), dtype, scale=2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("lens,H", [([24, 9, 1, 17, 24], 3), ([60, 33, 5], 2), ([15, 15], 12), ([7], 1),
                                    ([100, 37, 64, 1], 2), ([256, 130, 31], 3)])
def test_attention_packed_sequences(HF, dtype, lens, H):
    """Variable-length (packed) batches: sequence s = rows [off[s], off[s+1]) - same result as running
    every sequence on its own, forward and backward, with and without dropout (self-consistent)."""
    if dtype == torch.float32 and max(lens) > 64:
        pytest.skip("packed batches beyond 64 rows exist in the bf16 matrix-core kernels only")
    D = H * 64
    S, Lmax, M = len(lens), max(lens), sum(lens)
    off = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32).cuda()
    qkv = rnd(M, 3 * D, dtype=dtype, seed=1)
    dctx = rnd(M, D, dtype=dtype, seed=2)
    ctx, probs = HF.k_attn_fwd(qkv, None, S, Lmax, H, seq_off=off)
    dqkv = HF.k_attn_bwd(qkv, probs, dctx, S, Lmax, H, seq_off=off, ctx=ctx)
    r0 = 0
    for s_, n in enumerate(lens):
        HF.ATTN_SAVE_PROBS = probs.dim() == 4          # compare like with like (a long packed batch keeps its probabilities)
        try:
            c1, p1 = HF.k_attn_fwd(qkv[r0:r0 + n].contiguous(), None, 1, n, H)
        finally:
            HF.ATTN_SAVE_PROBS = False
        d1 = HF.k_attn_bwd(qkv[r0:r0 + n].contiguous(), p1, dctx[r0:r0 + n].contiguous(), 1, n, H, ctx=c1)
        close(ctx[r0:r0 + n], c1, dtype)
        if probs.dim() == 4:
            torch.testing.assert_close(probs[s_, :, :n, :n], p1[0], rtol=1e-5, atol=1e-6)
        else:                                          # bf16, Lmax <= 64: softmax row statistics instead of probabilities
            torch.testing.assert_close(probs.view(S, H, Lmax, 2)[s_, :, :n], p1.view(1, H, n, 2)[0], rtol=1e-5, atol=1e-6)
        close(dqkv[r0:r0 + n], d1, dtype, scale=2)
        r0 += n
    drop = HF.RNG.make(0.2, True, qkv.device)
    cd, pd = HF.k_attn_fwd(qkv, None, S, Lmax, H, drop=drop, seq_off=off)
    dd = HF.k_attn_bwd(qkv, pd, dctx, S, Lmax, H, drop=drop, seq_off=off, ctx=cd)
    lhs = (dctx.float() * cd.float()).sum()
    rhs = (dd[:, 2 * D:].float() * qkv[:, 2 * D:].float()).sum()
    torch.testing.assert_close(lhs, rhs, rtol=2e-2 if dtype == torch.bfloat16 else 1e-3, atol=0.5 if dtype == torch.bfloat16 else 1e-2)


@pytest.mark.parametrize("S,L,H,packed", [(301, 24, 12, False), (505, 24, 12, False), (410, 31, 12, True), (200, 20, 4, False), (420, 56, 12, True),
                                          (130, 50, 3, False), (300, 22, 12, False)])
def test_attention_several_pairs_per_wave(HF, Lb, S, L, H, packed):
    """Round 5: launches with more (sequence, head) pairs than one round of resident waves give every wave 2 or 3 pairs and
    prefetch the next pair's operands (attention_mfma.hip).  The arithmetic of a pair is unchanged: the big launch must give
    the BITS of the same sequences run one pair per wave in launches of other sizes (hero_attention_force_ppw), forward and backward, with dropout, with the
    saved row statistics and with saved probabilities, padded and packed (ragged lengths, incl. pairs past the last wave)."""
    D = H * 64
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(S)
    lens = [int(x) for x in torch.randint(1, L + 1, (S,), generator=g)] if packed else [L] * S
    M = sum(lens)
    qkv, dctx = rnd(M, 3 * D, dtype=dtype, seed=1), rnd(M, D, dtype=dtype, seed=2)
    off = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32).cuda() if packed else None
    m = (torch.rand(S, L, generator=g) > 0.1).float()
    m[:, 0] = 1
    madd = ((1 - m) * -10000.0).cuda()
    drop = HF.RNG.make(0.1, True, qkv.device)
    for save_probs, ppw in ((False, 0), (True, 0), (False, 2), (False, 3), (True, 3)):
        HF.ATTN_SAVE_PROBS = save_probs
        try:
            Lb.check(Lb.lib().hero_attention_force_ppw(ppw))     # 0: the launcher's choice; 2 / 3 pairs per wave, forward and backward
            try:
                ctx, saved = HF.k_attn_fwd(qkv, madd, S, L, H, drop=drop, seq_off=off)
                dq = HF.k_attn_bwd(qkv, saved, dctx, S, L, H, drop=drop, seq_off=off, ctx=ctx, mask_add=madd)
            finally:
                Lb.check(Lb.lib().hero_attention_force_ppw(1))   # the reference launches below: one pair per wave
            # the same sequences in chunks of 64 - the dropout index of an element depends on
            # its sequence number, so every chunk is run as sequences [s0, s0 + n) of a launch of s0 + n sequences whose first s0 are
            # empty (packed form: seq_off = 0 ... 0, then the chunk's bounds)
            r0 = 0
            for s0 in range(0, S, 64):
                n = min(64, S - s0)
                rows = sum(lens[s0:s0 + n])
                o = torch.tensor([0] * (s0 + 1) + list(torch.tensor(lens[s0:s0 + n]).cumsum(0)), dtype=torch.int32).cuda()
                mk = torch.zeros(s0 + n, L, device="cuda")
                mk[s0:] = madd[s0:s0 + n]
                c1, sv1 = HF.k_attn_fwd(qkv[r0:r0 + rows].contiguous(), mk, s0 + n, L, H, drop=drop, seq_off=o)
                d1 = HF.k_attn_bwd(qkv[r0:r0 + rows].contiguous(), sv1, dctx[r0:r0 + rows].contiguous(), s0 + n, L, H, drop=drop, seq_off=o,
                                   ctx=c1, mask_add=mk)
                assert torch.equal(ctx[r0:r0 + rows], c1), (save_probs, ppw, s0)
                assert torch.equal(dq[r0:r0 + rows], d1), (save_probs, ppw, s0)
                r0 += rows
        finally:
            HF.ATTN_SAVE_PROBS = False
            Lb.check(Lb.lib().hero_attention_force_ppw(0))
    # and against fp32 torch (padded case only: one reference for the whole launch)
    if not packed:
        ctx, saved = HF.k_attn_fwd(qkv, madd, S, L, H)
        q = qkv.float()
        qq, kk, vv = [t.reshape(S, L, H, 64).permute(0, 2, 1, 3) for t in q.split(D, dim=1)]
        pr = torch.softmax(qq @ kk.transpose(-1, -2) / 8.0 + madd[:, None, None, :], -1)
        close(ctx, (pr @ vv).permute(0, 2, 1, 3).reshape(S * L, D), dtype)


def test_row_stack_and_split_in_place(HF, Lb):
    """Round 5: the fused query pass stacks the subtitle rows and the query rows.  hero_gather_rows allocates the first (large)
    block with spare rows behind it, StackRowsFn copies only the small block in; in the backward CsrGatherSumFn does the same
    for the gradient and SplitRowsFn stacks in place.  Same values as cat / separate gradients; a block that merely sits at the
    start of a larger tensor is NOT extended (its neighbours are live data)."""
    dtype = torch.bfloat16
    a = rnd(300, 64, dtype=dtype, seed=1).requires_grad_(True)
    b = rnd(37, 64, dtype=dtype, seed=2).requires_grad_(True)
    idx = torch.randperm(300, generator=torch.Generator().manual_seed(3))[:200].to(torch.int32).cuda()
    first = HF.GatherRowsFn.apply(a, None, idx, 37)
    stacked = HF.StackRowsFn.apply(first, b)
    assert stacked.shape == (237, 64) and stacked.data_ptr() == first.data_ptr()          # the large block did not move
    assert torch.equal(stacked[:200], a.detach()[idx.long()]) and torch.equal(stacked[200:], b.detach())
    w = rnd(237, 64, dtype=dtype, seed=4)
    (stacked.float() * w.float()).sum().backward()
    ga = torch.zeros(300, 64, device="cuda")
    ga[idx.long()] = w[:200].float()
    torch.testing.assert_close(a.grad.float(), ga, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(b.grad.float(), w[200:].float(), rtol=1e-2, atol=1e-2)
    # no spare rows / a view into a larger live tensor: plain cat, nothing behind the block is touched
    big = rnd(400, 64, dtype=dtype, seed=5)
    keep = big.clone()
    st2 = HF.StackRowsFn.apply(big[:200], b.detach())
    assert st2.data_ptr() != big.data_ptr() and torch.equal(big, keep)
    assert torch.equal(st2[:200], big[:200]) and torch.equal(st2[200:], b.detach())
    # the split's backward: the first block's gradient comes from CsrGatherSumFn with spare rows, the second is copied behind it
    x = rnd(237, 64, dtype=dtype, seed=6).requires_grad_(True)
    blocks = HF.SplitRowsFn.apply(x, 200, 37)
    v0 = blocks[0].view(50, 4, 64)
    v0._hero_tail_rows = 37
    offs = torch.arange(0, 101, 2, dtype=torch.int32).cuda()[:51]          # 50 outputs of 2 source rows each (rows 0..99)
    ent = torch.arange(100, dtype=torch.int32).cuda()
    inv = torch.full((200,), -1, dtype=torch.int32)
    inv[:100] = torch.arange(100, dtype=torch.int32) // 2
    out = HF.CsrGatherSumFn.apply(v0, offs, ent, inv.cuda(), 50)
    wo, wq = rnd(50, 64, dtype=dtype, seed=7), rnd(37, 64, dtype=dtype, seed=8)
    ((out.float() * wo.float()).sum() + (blocks[1].float() * wq.float()).sum()).backward()
    want = torch.zeros(237, 64, device="cuda")
    want[:100] = wo.float().repeat_interleave(2, 0)
    want[200:] = wq.float()
    torch.testing.assert_close(x.grad.float(), want, rtol=1e-2, atol=1e-2)


def test_gather_backward_through_the_first_occurrence_map(HF, Lb):
    """Round 5: hero_inverse_first + GatherRowsFn(valid=mask).  The reference's gather index (data/data.py:504-512) names a
    source row a second time only from a padded position BEHIND its valid one, and padded positions receive exactly zero
    gradient - then the backward is one gather through the first-occurrence map.  Checked: the map itself, the gradients
    against the scatter-add backward when the repeated references carry zero gradient, and that a reference's reference-built
    index (tests/golden/case_collate.npz) has its repeats at padded positions only."""
    import numpy as np
    dtype = torch.bfloat16
    na, nb, n = 40, 100, 160
    g = torch.Generator().manual_seed(11)
    first = torch.randperm(na + nb, generator=g)[:120]                       # 120 distinct sources, valid part
    again = first[torch.randint(0, 120, (n - 120,), generator=g)]            # repeats, all behind the valid part
    src = torch.cat([first, again])
    idx = torch.where(src < na, src, -(src - na) - 2).to(torch.int32)
    idx[7] = -1                                                              # a position that reads nothing
    idx_d = idx.cuda()
    inv = torch.empty(na + nb, dtype=torch.int32, device="cuda")
    Lb.check(Lb.lib().hero_inverse_first(Lb.ptr(idx_d), n, Lb.ptr(inv), na, nb, Lb.stream()))
    want = torch.full((na + nb,), -1, dtype=torch.int32)
    for j in range(n - 1, -1, -1):
        v = int(idx[j])
        if v >= 0:
            want[v] = j
        elif v <= -2:
            want[na - v - 2] = j
    assert torch.equal(inv.cpu(), want)
    assert Lb.lib().hero_inverse_first(Lb.ptr(idx_d), n, Lb.ptr(inv), 30000, 30000, Lb.stream()) != 0      # beyond the LDS table: refused
    a0, b0 = rnd(na, 64, dtype=dtype, seed=1), rnd(nb, 64, dtype=dtype, seed=2)
    w = rnd(n, 64, dtype=dtype, seed=3)
    w[120:] = 0                                                              # the repeats (padded positions) carry no gradient
    grads = []
    valid = torch.zeros(n, dtype=torch.int64, device="cuda")
    valid[:120] = 1                                                          # the repeats sit at masked positions
    for fg in (None, valid):
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        out = HF.GatherRowsFn.apply(a, b, idx_d, 0, fg)
        if fg is not None:                                                   # the device-side check agreed: one-gather backward
            assert out.grad_fn.first is not None and out.grad_fn.first[1].ok()
        (out.float() * w.float()).sum().backward()
        grads.append((a.grad.clone(), b.grad.clone(), out.detach()))
    assert torch.equal(grads[0][2], grads[1][2])
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    # the reference's own index tensors: every repeated reference sits at a masked position
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "case_collate.npz"), allow_pickle=True)
    keys = [k for k in z.files if k.endswith("f_gather_index")]
    assert keys
    for k in keys:
        gi, m = z[k], z[k.replace("f_gather_index", "f_attn_masks")]
        for row_g, row_m in zip(gi, m):
            first_pos = {}
            for j, v in enumerate(row_g.tolist()):
                first_pos.setdefault(v, j)
            for j, (v, ok) in enumerate(zip(row_g.tolist(), row_m.tolist())):
                if ok:
                    assert first_pos[v] == j, (k, j, v)      # a VALID position is always the first reference to its source row


def test_gather_backward_falls_back_when_a_valid_position_repeats_a_source(HF, Lb):
    """VERDICT r5 #1c: the one-gather backward was hard-wired on a property of the REFERENCE's collate output.  It is checked
    at run time now (functional.first_reference_map): an index that references a source row from two VALID positions fails
    the check and gets the scatter-add backward, i.e. the sum of both gradients."""
    dtype = torch.float32
    na, nb, n = 6, 10, 16
    src = torch.arange(16)
    src[5] = 2                               # position 5 (valid) names source row 2 again; source row 5 is never referenced
    src[12] = 9                              # ... and a txt row twice as well
    idx = torch.where(src < na, src, -(src - na) - 2).to(torch.int32).cuda()
    valid = torch.ones(n, dtype=torch.int64, device="cuda")
    a0, b0, w = rnd(na, 64, dtype=dtype, seed=1), rnd(nb, 64, dtype=dtype, seed=2), rnd(n, 64, dtype=dtype, seed=3)
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    out = HF.GatherRowsFn.apply(a, b, idx, 0, valid)
    assert out.grad_fn.first is not None and not out.grad_fn.first[1].ok()
    (out * w).sum().backward()
    ar, br = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    both = torch.cat([ar, br], 0)
    (both[src.cuda()] * w).sum().backward()
    torch.testing.assert_close(a.grad, ar.grad)
    torch.testing.assert_close(b.grad, br.grad)
    assert float(a.grad[2].abs().sum()) > 0 and torch.equal(a.grad[2], (w[2] + w[5]))
    assert float(a.grad[5].abs().sum()) == 0
    # the same index with the repeats MASKED passes the check (and then drops what the masked positions were sent)
    valid2 = valid.clone()
    valid2[5] = 0
    valid2[12] = 0
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    out = HF.GatherRowsFn.apply(a, b, idx, 0, valid2)
    assert out.grad_fn.first[1].ok()
    w2 = w.clone()
    w2[5] = 0
    w2[12] = 0
    (out * w2).sum().backward()
    ar.grad = br.grad = None
    (torch.cat([ar, br], 0)[src.cuda()] * w2).sum().backward()
    torch.testing.assert_close(a.grad, ar.grad)
    torch.testing.assert_close(b.grad, br.grad)
    # without a mask nothing is assumed
    out = HF.GatherRowsFn.apply(a0.clone().requires_grad_(True), b0.clone().requires_grad_(True), idx, 0, None)
    assert out.grad_fn.first is None


def test_model_with_a_caller_built_gather_index_gets_the_reference_gradients():
    """The same at model level (tiny golden model, fp32): a batch whose f_gather_index names one frame row from two valid
    positions.  torch.gather's backward (the reference, model/encoder.py:271-279) adds both gradients; so does the HIP path."""
    import hero_amd
    from oracle import hero_oracle as O
    from tests.util import GOLDEN, load_tiny, rel_err, to_dev
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.float32)
    batch, _ = O.load_npz_case(os.path.join(GOLDEN, "case_train.npz"))
    gi, am = batch["f_gather_index"].clone(), batch["f_attn_masks"]
    row = int((am.sum(1) >= 3).nonzero()[0])
    assert int(am[row, 1]) == 1 and int(am[row, 0]) == 1
    gi[row, 1] = gi[row, 0]                                   # valid position 1 re-reads position 0's source row
    batch = dict(batch, f_gather_index=gi)
    model, P, cfg = load_tiny("cuda")
    model.train()
    set_dropout(model, 0.0)
    HF_ = hero_amd.functional
    HF_.set_grad_sink(None)
    losses = model(to_dev(batch, "cuda"), task="tvr", compute_loss=True)
    sum(l.sum() for l in losses).backward()
    Pq = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in P.items()}
    ref = O.vsm_losses(batch, Pq, cfg)
    sum(ref).backward()
    for got, want in zip(losses, ref):
        assert rel_err(got, want) < 1e-3
    grads = dict(model.named_parameters())
    for name in ("v_encoder.f_encoder.img_embeddings.img_linear.weight", "v_encoder.f_encoder.embeddings.word_embeddings.weight",
                 "v_encoder.f_encoder.img_embeddings.position_embeddings.weight"):
        assert rel_err(grads[name].grad, Pq[name].grad) < 1e-3, name


def test_attention_dropout_adjoint(HF, Lb):
    """With dropout on, forward is linear in V for a fixed mask and backward must use the SAME mask:
    <dctx, ctx(V)> == <dV, V>.  Also the keep rate is 1-p."""
    S, L, H, D = 4, 40, 2, 128
    qkv = rnd(S * L, 3 * D, seed=1)
    drop = HF.RNG.make(0.25, True, qkv.device)
    ctx, probs = HF.k_attn_fwd(qkv, None, S, L, H, drop=drop)
    ctx2, _ = HF.k_attn_fwd(qkv, None, S, L, H, drop=drop)
    torch.testing.assert_close(ctx, ctx2)                       # same site -> same mask
    ctx0, _ = HF.k_attn_fwd(qkv, None, S, L, H)
    assert (ctx - ctx0).abs().max() > 1e-3
    dctx = rnd(S * L, D, seed=2)
    dqkv = HF.k_attn_bwd(qkv, probs, dctx, S, L, H, drop=drop)
    v = qkv[:, 2 * D:]
    lhs = (dctx * ctx).sum()
    rhs = (dqkv[:, 2 * D:] * v).sum()
    torch.testing.assert_close(lhs, rhs, rtol=1e-3, atol=1e-2)
    # keep-rate through a GEMM epilogue: ones @ I with dropout -> entries are 0 or 1/(1-p)
    n = 256
    y = torch.empty(n, n, device="cuda")
    HF.k_gemm(torch.eye(n).cuda(), torch.ones(n, n).cuda(), y, n, n, n, n, n, n, Lb.LAYOUT_K, Lb.LAYOUT_K,
              Lb.F32, drop=HF.RNG.make(0.1, True, y.device))
    kept = (y > 0).float().mean().item()
    assert abs(kept - 0.9) < 0.01, kept
    vals = torch.unique(y)
    assert len(vals) == 2 and abs(vals.max().item() - 1 / 0.9) < 1e-5


@pytest.mark.parametrize("S,L,H", [(5, 24, 3), (3, 15, 2), (2, 60, 2), (2, 37, 1), (2, 100, 2), (2, 200, 3), (1, 256, 2)])
def test_attention_mfma_dropout_matches_f32_kernels(HF, Lb, S, L, H):
    """The bf16 matrix-core attention kernels (L <= 64 and 64 < L <= 256) and the fp32 VALU kernels draw the SAME dropout
    mask for the same site (index = ((s*H+h)*L + q)*round_up(L,4) + k): forward and backward agree on
    bf16-representable inputs within the bf16 tolerance, and the adjoint identity holds."""
    D = H * 64
    qkv16 = rnd(S * L, 3 * D, dtype=torch.bfloat16, seed=1)
    dctx16 = rnd(S * L, D, dtype=torch.bfloat16, seed=2)
    m = torch.ones(S, L)
    m[0, L - 1] = 0
    madd = ((1 - m) * -10000.0).cuda()
    drop = HF.RNG.make(0.3, True, qkv16.device)
    HF.ATTN_SAVE_PROBS = True
    try:
        ctx16, probs16 = HF.k_attn_fwd(qkv16, madd, S, L, H, drop=drop)
    finally:
        HF.ATTN_SAVE_PROBS = False
    ctx32, probs32 = HF.k_attn_fwd(qkv16.float(), madd, S, L, H, drop=drop)
    torch.testing.assert_close(probs16, probs32, rtol=2e-2, atol=2e-3)
    close(ctx16, ctx32, torch.bfloat16)
    d16 = HF.k_attn_bwd(qkv16, probs32, dctx16, S, L, H, drop=drop, ctx=ctx16)
    if Lb.lib().hero_attention_stats_ok(Lb.BF16, L):             # with dropout too: statistics alone give the same bits
        ctx_s, stats = HF.k_attn_fwd(qkv16, madd, S, L, H, drop=drop)
        assert stats.dim() == 1 and torch.equal(ctx_s, ctx16)
        assert torch.equal(HF.k_attn_bwd(qkv16, stats, dctx16, S, L, H, drop=drop, ctx=ctx16, mask_add=madd),
                           HF.k_attn_bwd(qkv16, probs16, dctx16, S, L, H, drop=drop, ctx=ctx16))
    if L <= 128:                                  # the fp32 backward is LDS-resident up to hero_attention_max_len(F32, 1)
        d32 = HF.k_attn_bwd(qkv16.float(), probs32, dctx16.float(), S, L, H, drop=drop)
        close(d16, d32, torch.bfloat16, scale=2)
    lhs = (dctx16.float() * ctx16.float()).sum()
    rhs = (d16[:, 2 * D:].float() * qkv16[:, 2 * D:].float()).sum()
    torch.testing.assert_close(lhs, rhs, rtol=2e-2, atol=0.5)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,cols,ld", [(37, 50265, 50272), (5, 100, 100), (64, 1931, 1936), (3, 7, 7)])
def test_cross_entropy_matches_torch(HF, dtype, rows, cols, ld):
    """hero_cross_entropy_fwd/bwd (MLM vocabulary rows with padding columns, NCE with a temperature, FOM with
    ignore_index) against F.cross_entropy on the same logits."""
    x = rnd(rows, ld, dtype=dtype, seed=1, scale=3.0)
    g = torch.Generator().manual_seed(2)
    y = torch.randint(0, cols, (rows,), generator=g).cuda()
    y[0] = -1
    w = rnd(rows, seed=3).abs() + 0.5
    for temp in (1.0, 0.7):
        xr = x.float()[:, :cols].clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(xr / temp, y, ignore_index=-1, reduction="none")
        (ref * w).sum().backward()
        xh = x.clone().requires_grad_(True)
        got = HF.cross_entropy(xh, y, ncols=cols, ignore_index=-1, inv_temp=1.0 / temp)
        (got * w).sum().backward()
        torch.testing.assert_close(got, ref.detach(), rtol=1e-4, atol=1e-4)
        tol = dict(rtol=1e-4, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-3)
        torch.testing.assert_close(xh.grad.float()[:, :cols], xr.grad, **tol)
        assert float(xh.grad.float()[:, cols:].abs().sum()) == 0.0 and float(xh.grad.float()[0].abs().sum()) == 0.0


@pytest.mark.parametrize("dtype", DT)
def test_matmul_nt_activations(HF, dtype):
    a = rnd(130, 4352, dtype=dtype, seed=1, scale=0.1).requires_grad_(True)
    b = rnd(1936, 4352, dtype=dtype, seed=2, scale=0.1).requires_grad_(True)
    y = HF.matmul_nt(a, b)
    af, bf = a.detach().float().requires_grad_(True), b.detach().float().requires_grad_(True)
    ref = af @ bf.t()
    close(y, ref, dtype, scale=1.0)
    dy = rnd(130, 1936, dtype=dtype, seed=3)
    y.backward(dy)
    ref.backward(dy.float())
    close(a.grad, af.grad, dtype, scale=4.0)
    close(b.grad, bf.grad, dtype, scale=2.0)


def test_ln_bwd_dropout_consistency(HF, Lb):
    """ProjResLn: dx_dropped must equal dx * (the GEMM epilogue's mask)."""
    M, N, K = 300, 256, 128
    x, w, b, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    drop = HF.RNG.make(0.3, True, x.device)
    y = HF.k_linear(x, w, b, residual=res, drop=drop)
    z = x @ w.t() + b
    mask = ((y - res).abs() > 1e-6).float()          # kept positions (z != 0 almost surely)
    torch.testing.assert_close(y, z * mask / 0.7 + res, rtol=1e-4, atol=1e-4)
    g, bt = rnd(N, seed=5) * 0.1 + 1, rnd(N, seed=6)
    out, mean, rstd, _ = HF.k_ln_fwd(y, g, bt, 1e-12, torch.float32, M, N)
    dy = rnd(M, N, seed=7)
    dx, dxd, _, _ = HF.k_ln_bwd(y, dy, g, mean, rstd, drop_in=drop)
    torch.testing.assert_close(dxd, dx * mask / 0.7, rtol=1e-5, atol=1e-6)
    # fused pass: parameter gradients + the feeding linear's bias gradient, accumulated (beta = 1)
    dg = torch.ones(N, device="cuda"); db = torch.full((N,), 2.0, device="cuda"); dbias = torch.full((N,), 3.0, device="cuda")
    dx2, dxd2, _, _ = HF.k_ln_bwd(y, dy, g, mean, rstd, drop_in=drop, dgamma=dg, dbeta=db, grad_beta=1.0,
                                  want_params=False, dbias_in=dbias)
    torch.testing.assert_close(dx2, dx)
    xh = (y - mean[:, None]) * rstd[:, None]
    torch.testing.assert_close(dg, 1 + (dy * xh).sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db, 2 + dy.sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dbias, 3 + dxd.sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("p_drop", [0.1, 0.3])
def test_dropout_mask_statistics(HF, Lb, p_drop):
    """The counter-based mask (csrc/common.h DropCtx: two keyed 32-bit hashes per 4-element group): keep rate within
    5 sigma of 1 - p, no correlation between neighbours along a row, down a column or between the four elements of a
    group, a different mask per site, the same mask when a site is replayed."""
    M, N, K = 2048, 768, 64
    x, w, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(M, N, seed=4)
    masks = []
    for _ in range(2):
        drop = HF.RNG.make(p_drop, True, x.device)
        y = HF.k_linear(x, w, None, residual=res, drop=drop)
        y2 = HF.k_linear(x, w, None, residual=res, drop=drop)
        torch.testing.assert_close(y, y2, rtol=0, atol=0)                   # same site + seed word: same mask
        masks.append(((y - res).abs() > 1e-7).double())
    m = masks[0]
    n = m.numel()
    sigma = math.sqrt(p_drop * (1 - p_drop) / n)
    assert abs(m.mean().item() - (1 - p_drop)) < 5 * sigma + 1.0 / 65536      # the threshold is a 16-bit fraction

    def corr(a, b):
        a, b = a.reshape(-1) - a.mean(), b.reshape(-1) - b.mean()
        return (a * b).mean().item() / math.sqrt((a * a).mean().item() * (b * b).mean().item())
    lim = 6.0 / math.sqrt(n)
    assert abs(corr(m[:, :-1], m[:, 1:])) < lim                               # along a row (inside and across groups)
    assert abs(corr(m[:-1], m[1:])) < lim                                     # down a column
    g4 = m.reshape(M, N // 4, 4)
    for i in range(4):
        for j in range(i + 1, 4):
            assert abs(corr(g4[..., i], g4[..., j])) < lim                    # the four decisions of one hash pair
    assert abs(corr(masks[0], masks[1])) < lim                                # another site: another mask


@pytest.mark.parametrize("dtype", DT)
def test_gather_scatter(HF, dtype):
    a, b = rnd(50, 128, dtype=dtype, seed=1), rnd(30, 128, dtype=dtype, seed=2)
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(-31, 50, (200,), generator=g).int()
    idx[idx == -1] = 5
    idx[::17] = -1
    idx = idx.cuda()
    out = HF.k_gather_rows(a, b, idx, 200, 128)
    li = idx.long()
    ref = torch.where((li >= 0)[:, None], a[li.clamp(min=0)].float(),
                      torch.where((li == -1)[:, None], torch.zeros(1, 128, device="cuda"),
                                  b[(-li - 2).clamp(min=0)].float()))
    torch.testing.assert_close(out.float(), ref)
    da, db = torch.zeros_like(a), torch.zeros_like(b)
    src = rnd(200, 128, dtype=dtype, seed=4)
    HF.k_scatter_add(src, idx, da, db)
    ra, rb = torch.zeros(50, 128, device="cuda"), torch.zeros(30, 128, device="cuda")
    ra.index_add_(0, li[li >= 0], src.float()[li >= 0])
    rb.index_add_(0, (-li - 2)[li <= -2], src.float()[li <= -2])
    close(da, ra, dtype, scale=4)
    close(db, rb, dtype, scale=4)
    tab = torch.zeros(50, 128, device="cuda")
    HF.k_scatter_add(src, idx.clamp(min=0), tab, None, skip=5)
    rt = torch.zeros(50, 128, device="cuda")
    keep = li.clamp(min=0) != 5
    rt.index_add_(0, li.clamp(min=0)[keep], src.float()[keep])
    torch.testing.assert_close(tab, rt, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rows,n_dst,cols,dtype", [(9600, 50272, 768, torch.bfloat16), (1300, 37, 768, torch.float32),
                                                   (70000, 50272, 64, torch.bfloat16), (257, 2, 4352, torch.float32)])
def test_scatter_add_sorted_is_exact_and_reproducible(HF, Lb, rows, n_dst, cols, dtype):
    """hero_segment_sort + hero_scatter_add_sorted (round 4): embedding-table gradients without atomics.  The order is
    the stable sort of the rows by destination (dropped rows last); the sums equal index_add_ of the same rows and are
    bit-identical run to run, also with destinations that receive hundreds of rows (which the atomic version is not)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    idx = torch.randint(0, n_dst, (rows,), device="cuda", generator=g, dtype=torch.int32)
    idx[::7] = 5 % n_dst                                   # a frequent token
    idx[1::11] = 1 % n_dst                                 # the padding id: skipped
    idx[2::13] = -1                                        # dropped rows
    skip = 1 % n_dst
    src = rnd(rows, cols, dtype=dtype, seed=4)
    order = HF.segment_order(idx, n_dst, skip)
    key = torch.where((idx < 0) | (idx == skip), torch.full_like(idx, 1 << 30), idx).long()
    want = torch.sort(key * rows + torch.arange(rows, device="cuda"), stable=True)[1].int()
    assert torch.equal(order, want)
    outs = []
    for rep in range(3):
        dst = torch.full((n_dst, cols), 0.5, device="cuda")
        HF.k_scatter_add_sorted(src, idx, dst, skip)
        outs.append(dst)
    keep = (idx >= 0) & (idx != skip)
    ref = torch.full((n_dst, cols), 0.5, device="cuda", dtype=torch.float64)
    ref.index_add_(0, idx[keep].long(), src[keep].double())
    torch.testing.assert_close(outs[0].double(), ref, rtol=1e-5, atol=1e-4 * (rows / max(n_dst, 1) + 8))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_csr_gather_and_elementwise(HF, Lb):
    from hero_amd.model.model import build_frame_map
    num_subs = [2, 1]
    sub2frm = [[(0, [0, 1]), (1, [3])], [(0, [2, 0])]]
    offs, ent, inv = build_frame_map(num_subs, sub2frm, 2, 4, 5, torch.device("cuda"))
    src = rnd(3 * 5, 64, seed=1).requires_grad_(True)
    out = HF.CsrGatherSumFn.apply(src, offs, ent, inv, 8)
    ref = torch.zeros(8, 64, device="cuda")
    ref[0], ref[1], ref[3] = src[0], src[1], src[5]
    ref[4 + 2], ref[4 + 0] = src[10], src[11]
    torch.testing.assert_close(out, ref)
    out.backward(torch.ones_like(out) * 2)
    expect = torch.zeros(15, 64, device="cuda")
    expect[[0, 1, 5, 10, 11]] = 2
    torch.testing.assert_close(src.grad, expect)
    x = rnd(1001, seed=2)
    torch.testing.assert_close(HF.k_cast(HF.k_cast(x, torch.bfloat16), torch.float32), x.bfloat16().float())


def test_adamw_matches_oracle(HF, Lb):
    from oracle import hero_oracle as O
    from hero_amd.optim import AdamW
    p0, g = rnd(1000, 33, seed=1), rnd(1000, 33, seed=2)
    p = torch.nn.Parameter(p0.clone())
    opt = AdamW([{"params": [p], "weight_decay": 0.01}], lr=1e-3, betas=(0.9, 0.98))
    P = {"w": p0.clone().cpu()}
    state = {}
    for step in (1, 2, 3):
        p.grad = g * step
        opt.step()
        O.adamw_step(P, {"w": (g * step).cpu()}, state, lr=1e-3, step=step)
    torch.testing.assert_close(p.detach().cpu(), P["w"], rtol=1e-5, atol=1e-6)
    # fused clip: norm > max -> grads scaled by max/(norm+1e-6)
    q = torch.nn.Parameter(p0.clone())
    opt2 = AdamW([{"params": [q], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.98))
    q.grad = g.clone()
    opt2.step(grad_sumsq=opt2.grad_sumsq(), max_grad_norm=1.0)
    P2, st2 = {"w": p0.clone().cpu()}, {}
    gn = g.norm().cpu()
    O.adamw_step(P2, {"w": g.cpu() * (1.0 / (gn + 1e-6))}, st2, lr=1e-3, step=1, wd=0.0)
    torch.testing.assert_close(q.detach().cpu(), P2["w"], rtol=1e-5, atol=1e-6)


def test_colsum_multi_and_deferred_layernorm_fold(HF, Lb):
    """hero_colsum_multi: many column sums (bf16 activations with a column offset, fp32 partial matrices, accumulate /
    overwrite) in two launches, bit-reproducible; hero_layernorm_bwd(defer_fold) leaves the partials the multi-fold
    then turns into the same dgamma / dbeta / dbias the immediate fold gives."""
    import ctypes as C
    srcs = [rnd(12000, 2304, dtype=torch.bfloat16, seed=1), rnd(1920, 768, dtype=torch.bfloat16, seed=2),
            rnd(1000, 3 * 768, dtype=torch.float32, seed=3), rnd(37, 16, dtype=torch.float32, seed=4)]
    specs = [(srcs[0], 768, 1536, 1.0), (srcs[1], 0, 768, 0.0), (srcs[2], 768, 768, 1.0), (srcs[3], 0, 16, 1.0)]
    outs = [torch.full((n,), 0.5, device="cuda") for _, _, n, _ in specs]
    probs = (Lb.Colsum * len(specs))()
    for i, (t, c0, n, beta) in enumerate(specs):
        probs[i] = Lb.Colsum(t.data_ptr() + c0 * t.element_size(), outs[i].data_ptr(), t.shape[0], n, t.shape[1], Lb.dt(t), beta)
    ws = torch.empty(Lb.lib().hero_colsum_multi_workspace_bytes(probs, len(specs)) // 4 + 1, device="cuda")
    first = None
    for rep in range(3):
        for o in outs:
            o.fill_(0.5)
        Lb.check(Lb.lib().hero_colsum_multi(probs, len(specs), ws.data_ptr(), Lb.stream()))
        torch.cuda.synchronize()
        if first is None:
            first = [o.clone() for o in outs]
        assert all(torch.equal(a, b) for a, b in zip(outs, first))
    for (t, c0, n, beta), o in zip(specs, outs):
        ref = t.float()[:, c0:c0 + n].sum(0) + 0.5 * beta
        torch.testing.assert_close(o, ref, rtol=2e-5, atol=2e-3)
    # deferred LayerNorm fold == immediate fold
    rows, cols = 4000, 768
    x, dy = rnd(rows, cols, dtype=torch.bfloat16, seed=5), rnd(rows, cols, dtype=torch.bfloat16, seed=6)
    gamma = rnd(cols, seed=7)
    mean, rstd = x.float().mean(1), 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5)
    res = []
    for defer in (0, 1):
        dg, db, dbi = (torch.zeros(cols, device="cuda") for _ in range(3))
        dx = torch.empty_like(dy)
        nblk = Lb.lib().hero_layernorm_bwd_blocks(rows)
        part = torch.empty(max(nblk, 1024) * 3 * cols, device="cuda")
        a = Lb.LnBwd()
        a.x, a.dy, a.gamma, a.mean, a.rstd = x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        a.dx, a.dgamma, a.dbeta, a.dbias_in = dx.data_ptr(), dg.data_ptr(), db.data_ptr(), dbi.data_ptr()
        a.grad_beta, a.workspace, a.rows, a.cols, a.x_dtype, a.dtype, a.defer_fold = 1.0, part.data_ptr(), rows, cols, Lb.BF16, Lb.BF16, defer
        a.dropout_out.scale = a.dropout_in.scale = 1.0
        Lb.check(Lb.lib().hero_layernorm_bwd(C.byref(a), Lb.stream()))
        if defer:
            p3 = (Lb.Colsum * 3)(*[Lb.Colsum(part.data_ptr() + 4 * k * cols, t.data_ptr(), nblk, cols, 3 * cols, Lb.F32, 1.0)
                                   for k, t in enumerate((dg, db, dbi))])
            w3 = torch.empty(Lb.lib().hero_colsum_multi_workspace_bytes(p3, 3) // 4 + 1, device="cuda")
            Lb.check(Lb.lib().hero_colsum_multi(p3, 3, w3.data_ptr(), Lb.stream()))
        torch.cuda.synchronize()
        res.append((dg.clone(), db.clone(), dbi.clone()))
    for a_, b_ in zip(res[0], res[1]):
        torch.testing.assert_close(a_, b_, rtol=1e-5, atol=1e-3)
