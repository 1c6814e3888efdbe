"""Autograd functions over the libhero_hip.so kernels.

Everything arithmetic on the encoder path runs in hand-written HIP kernels reached through the
C ABI (hero_amd/_lib.py); torch is used here for device memory, stream handles, autograd
bookkeeping and a few index tensors.  No function in this file has a CPU or torch-eager
fallback.

Block structure (one autograd node each, residual fan-in fused into GEMM epilogues):
  attn_block : LN(drop(ctx Wo^T + bo) + x),  ctx = MHA(x Wqkv^T + bqkv)     model/layers.py:217-222
  ffn_block  : LN(drop(gelu(a W1^T + b1) W2^T + b2) + a)                    model/layers.py:264-272
"""
import ctypes as C
import math
import os
import weakref

import numpy as np
import torch

from . import _lib as L

# --------------------------------------------------------------------------------------------- #
# global switches
# --------------------------------------------------------------------------------------------- #
_COMPUTE_DTYPE = torch.bfloat16


def set_compute_dtype(dtype):
    """torch.bfloat16 (default: bf16 storage + bf16 MFMA, fp32 accumulate) or torch.float32
    (exact-f32 MFMA path used for the parity gate)."""
    global _COMPUTE_DTYPE
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _COMPUTE_DTYPE = dtype


def compute_dtype():
    return _COMPUTE_DTYPE


class _Rng:
    """Counter-based dropout state: one device uint64 seed word per device + a site counter."""

    def __init__(self):
        self.seeds = {}
        self.site = 0

    def seed_tensor(self, device):
        idx = device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else None)
        key = (device.type, idx)            # "cuda" and "cuda:0" are the same seed word
        t = self.seeds.get(key)
        if t is None:
            t = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64,
                             device=device)
            self.seeds[key] = t
        return t

    def make(self, p, training, device):
        if not training or p <= 0.0:
            return None
        if p >= 1.0:
            raise ValueError("dropout p must be < 1")
        self.site += 1
        return L.Dropout(self.seed_tensor(device).data_ptr(), self.site,
                         max(1, int(round(p * 65536.0))), 1.0 / (1.0 - p))


RNG = _Rng()


def manual_seed(seed, device=None):
    """Reset the dropout stream (kept on the device so that hipGraph replays can advance it)."""
    RNG.site = 0
    for t in RNG.seeds.values():
        t.fill_(seed & 0x7FFFFFFFFFFFFFFF)
    if device is not None:
        RNG.seed_tensor(torch.device(device)).fill_(seed & 0x7FFFFFFFFFFFFFFF)


def advance_seed():
    """Advance every device seed word by one (a device-side op: safe inside graph capture)."""
    for t in RNG.seeds.values():
        t.add_(1)


def _d(drop):
    return drop if drop is not None else L.no_dropout()


# --------------------------------------------------------------------------------------------- #
# compute copies of the fp32 master weights
# --------------------------------------------------------------------------------------------- #
_WCACHE = {}
_WEPOCH = [0]
_WGEN = [0]            # grows with every NEW compute copy: a captured refresh_weight_cache() covers the copies of its time only


def weight_cache_generation():
    return _WGEN[0]


def notify_weights_updated():
    """Call after parameters were changed in a way autograd's version counters cannot see
    (raw-pointer optimisers such as hero_amd.optim.AdamW, or `p.data` updates)."""
    _WEPOCH[0] += 1


def clear_weight_cache():
    _WGEN[0] += 1
    _WCACHE.clear()
    for k in [k for k in _REFRESH if k not in _REFRESH_PINNED]:
        del _REFRESH[k]


_TRUST = [False]


class weights_frozen:
    """Inside this context the caller guarantees that parameters change only through
    `notify_weights_updated()`-announcing code (hero_amd.optim.AdamW): cache hits then skip the
    per-call version / data_ptr signature (1.5 ms of Python per micro-step on HERO-base)."""

    def __enter__(self):
        self.prev = _TRUST[0]
        _TRUST[0] = True

    def __exit__(self, *exc):
        _TRUST[0] = self.prev


_REFRESH = {}              # descriptor-table signature -> device tables of hero_copy_multi (several: the tasks of a multi-task run
_REFRESH_PINNED = set()    # update different parameter sets); tables used under stream capture are never dropped (a graph holds their addresses)


def refresh_weight_cache():
    """Re-derive every cached compute copy from the current fp32 master weights NOW, in ONE launch
    (hero_copy_multi over a device-resident descriptor table: ~160 cast / transpose launches per
    optimiser step otherwise).  Called right after the optimiser step - eagerly, and inside the
    captured hipGraph - so the next forward finds the cache valid.  (Round 5 also let the AdamW kernel write the straight
    copies from its registers; measured +17 us per optimiser step in the kernels and no change of the step - removed in
    round 6, DESIGN changelog.)"""
    if not _WCACHE:
        return
    entries = list(_WCACHE.items())
    tsig = tuple((key, val[1].data_ptr(), tuple(p.data_ptr() for p in val[2])) for key, val in entries)
    if tsig not in _REFRESH:
        if len(_REFRESH) > 16:
            for k in [k for k in _REFRESH if k not in _REFRESH_PINNED][:8]:
                del _REFRESH[k]
        descs, tdesc, tidx = [], [], []
        for key, (_, out, params) in entries:
            transposed = len(key) == 3
            esz, code = out.element_size(), L.dt(out)
            off = 0
            for p in params:
                if not p.is_contiguous() or p.dtype != torch.float32:
                    raise RuntimeError("hero_amd: master weights must be contiguous fp32")
                rows = p.shape[0] if p.dim() > 1 else 1
                cols = p.numel() // rows
                if transposed:          # out [K, sum N]: this tensor's columns start at `off`
                    d = L.CopyDesc(p.data_ptr(), out.data_ptr() + off * esz, rows, cols, out.shape[1], 1, code, 0)
                    off += rows
                else:                   # out [sum N, K] (or a flat concatenation of 1-D tensors)
                    d = L.CopyDesc(p.data_ptr(), out.data_ptr() + off * esz, rows, cols, cols, 0, code, 0)
                    off += rows * cols
                nt = (-(-rows // 64)) * (-(-cols // 64))
                tdesc += [len(descs)] * nt
                tidx += list(range(nt))
                descs.append(d)
        dev = entries[0][1][1].device
        arr = (L.CopyDesc * len(descs))(*descs)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        _REFRESH[tsig] = (raw, torch.tensor(tdesc, dtype=torch.int32, device=dev),
                          torch.tensor(tidx, dtype=torch.int32, device=dev), len(tdesc))
    if torch.cuda.is_current_stream_capturing():
        _REFRESH_PINNED.add(tsig)
    raw, tdesc, tidx, n = _REFRESH[tsig]
    L.check(L.lib().hero_copy_multi(raw.data_ptr(), tdesc.data_ptr(), tidx.data_ptr(), n, L.stream()))
    ep = _WEPOCH[0]
    for key, (_, out, params) in entries:
        _WCACHE[key] = ((tuple(p._version for p in params), tuple(p.data_ptr() for p in params), ep), out, params)


def packed(params, dtype):
    """Row-concatenate fp32 parameters into one contiguous tensor of `dtype` (cached)."""
    params = tuple(params)
    if len(params) == 1 and dtype == torch.float32 and params[0].is_contiguous():
        return params[0].detach()
    key = (tuple(id(p) for p in params), dtype)
    hit = _WCACHE.get(key)
    if _TRUST[0] and hit is not None and hit[0] is not None and hit[0][2] == _WEPOCH[0]:
        return hit[1]
    sig = (tuple(p._version for p in params), tuple(p.data_ptr() for p in params), _WEPOCH[0])
    if hit is not None and hit[0] == sig:
        return hit[1]
    rows = sum(p.shape[0] for p in params)
    shape = (rows,) + tuple(params[0].shape[1:])
    reuse = hit is not None and tuple(hit[1].shape) == shape and hit[1].device == params[0].device
    out = hit[1] if reuse else torch.empty(shape, dtype=dtype, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        dst = out[off:off + p.shape[0]]
        L.check(L.lib().hero_cast(L.ptr(p.detach().contiguous()), dst.data_ptr(), n, L.F32,
                                  L.dt(out), L.stream()))
        off += p.shape[0]
    if hit is None or not reuse:
        _WGEN[0] += 1
    _WCACHE[key] = (sig, out, params)
    return out


def packed_t(params, dtype):
    """Transposed compute copy: [K_in, sum_i N_i] = concat_i(p_i^T) (cached like `packed`), used as
    the K-contiguous B operand of dgrad: dX = dY @ W  ==  dY @ (W^T)^T."""
    params = tuple(params)
    key = (tuple(id(p) for p in params), dtype, "T")
    hit = _WCACHE.get(key)
    if _TRUST[0] and hit is not None and hit[0] is not None and hit[0][2] == _WEPOCH[0]:
        return hit[1]
    sig = (tuple(p._version for p in params), tuple(p.data_ptr() for p in params), _WEPOCH[0])
    if hit is not None and hit[0] == sig:
        return hit[1]
    kin = params[0].shape[1]
    cols = sum(p.shape[0] for p in params)
    reuse = hit is not None and tuple(hit[1].shape) == (kin, cols) and hit[1].device == params[0].device
    out = hit[1] if reuse else torch.empty((kin, cols), dtype=dtype, device=params[0].device)
    off = 0
    for p in params:
        L.check(L.lib().hero_transpose_cast(L.ptr(p.detach().contiguous()),
                                            out.data_ptr() + off * out.element_size(), p.shape[0], kin,
                                            cols, L.dt(out), L.stream()))
        off += p.shape[0]
    if hit is None or not reuse:
        _WGEN[0] += 1
    _WCACHE[key] = (sig, out, params)
    return out


# --------------------------------------------------------------------------------------------- #
# parameter-gradient sink
# --------------------------------------------------------------------------------------------- #
class GradSink:
    """Where the HIP backward kernels ACCUMULATE parameter gradients (dW += ..., fp32).

    The kernels add straight into `p.grad` (allocated zeroed on first use), so there are no
    per-parameter temporaries and no autograd AccumulateGrad launches; the functions return None
    for parameter inputs.  hero_amd.utils.distributed.GradArena subclasses this to make every
    `p.grad` a view of one flat buffer and to launch bucketed all-reduces as gradients become
    final (`use` is called once per forward use of a parameter, `done` once per backward use)."""

    def dst(self, p):
        if p.grad is None:
            p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return p.grad

    def dst_group(self, params):
        """A single contiguous view over several parameters' gradients, or None (separate tensors)."""
        return None

    def use(self, p):
        pass

    def done(self, p):
        pass

    def wants_overlap(self):
        """True when gradients should become final layer by layer during this backward pass (a bucketed all-reduce
        is waiting for them): weight gradients are then launched per layer instead of per encoder."""
        return False


SINK = GradSink()


def set_grad_sink(sink):
    global SINK
    SINK = sink if sink is not None else GradSink()


def _use(*params):
    """Count one forward use of each parameter (called from inside autograd.Function.forward, where
    grad mode is always off - so no is_grad_enabled() test here; the sink resets its counts at the
    start of every micro-step)."""
    for p in params:
        if p is not None and p.requires_grad:
            SINK.use(p)


def _is_param(t):
    return t is not None and t.is_leaf and t.requires_grad


# --------------------------------------------------------------------------------------------- #
# raw kernel wrappers (no autograd)
# --------------------------------------------------------------------------------------------- #
def _p(t):
    return t if isinstance(t, int) else L.ptr(t)


def k_gemm(A, B, Cm, M, N, K, lda, ldb, ldc, al, bl, dtype_code, bias=None, residual=None,
           aux=None, act=L.ACT_NONE, out_f32=False, beta=0.0, split_k=1, drop=None, colsum=None, split_stride=0,
           colsum_partial=False):
    """A/B/Cm may be tensors or raw device pointers (int) for column-sliced operands."""
    epi = L.GemmEpilogue(_p(bias), _p(residual), _p(aux), act, 1 if out_f32 else 0,
                         beta, split_k, _d(drop), _p(colsum), 1 if colsum_partial else 0, 0, split_stride)
    L.check(L.lib().hero_gemm(_p(A), _p(B), _p(Cm), M, N, K, lda, ldb, ldc, al, bl,
                              dtype_code, C.byref(epi), L.stream()))


def k_linear(x2, Wc, bias=None, act=L.ACT_NONE, aux=None, residual=None, drop=None):
    """y[M,N] = epilogue(x2[M,K] @ Wc[N,K]^T)."""
    M, K = x2.shape
    N = Wc.shape[0]
    y = torch.empty((M, N), dtype=x2.dtype, device=x2.device)
    k_gemm(x2, Wc, y, M, N, K, K, K, N, L.LAYOUT_K, L.LAYOUT_K, L.dt(x2), bias=bias,
           residual=residual, aux=aux, act=act, drop=drop)
    return y


def k_dgrad(dy2, Wc, act=L.ACT_NONE, aux=None, residual=None):
    """dx[M,K] = epilogue(dy2[M,N] @ Wc[N,K])   (W in its nn.Linear layout, outer-contiguous B)."""
    M, N = dy2.shape
    K = Wc.shape[1]
    dx = torch.empty((M, K), dtype=dy2.dtype, device=dy2.device)
    k_gemm(dy2, Wc, dx, M, K, N, N, K, K, L.LAYOUT_K, L.LAYOUT_O, L.dt(dy2), act=act, aux=aux,
           residual=residual)
    return dx


def k_dgrad_t(dy2, Wt, act=L.ACT_NONE, aux=None, residual=None, colsum=None, colsum_partial=False):
    """dx[M,K] = epilogue(dy2[M,N] @ Wt[K,N]^T) with the transposed weight copy: both operands
    reduction-contiguous -> the direct-to-LDS GEMM path.  colsum (fp32 [K]) += column sums of dx (fp32 atomics), or with
    colsum_partial a [ceil(M / 64), K] table of per-tile sums that is overwritten (deterministic, HeroGemmEpilogue)."""
    M, N = dy2.shape
    K = Wt.shape[0]
    if (dy2.dtype == torch.bfloat16 and act == L.ACT_NONE and aux is None and residual is None and colsum is None
            and N >= 8192 and M * K <= (1 << 21) and (M * K) % 4 == 0 and (N - N // 64 * 64) % 8 == 0
            and dy2.is_contiguous() and Wt.is_contiguous() and Wt.shape[1] == N):
        return _dgrad_long_reduction(dy2, Wt)
    dx = torch.empty((M, K), dtype=dy2.dtype, device=dy2.device)
    k_gemm(dy2, Wt, dx, M, K, N, N, N, K, L.LAYOUT_K, L.LAYOUT_K, L.dt(dy2), act=act, aux=aux,
           residual=residual, colsum=colsum, colsum_partial=colsum_partial)
    return dx


def _dgrad_long_reduction(dy2, Wt):
    """dx = dy2 @ Wt^T where the reduction is tens of thousands long and the output small: the gradient of the MLM
    vocabulary projection w.r.t. its input, (1440, 50272) x (50272, 768) (model/layers.py:330-354).  One tile per
    workgroup leaves 72 workgroups walking 786 k-steps each, and 50272 % 64 != 0 sent the whole GEMM to the
    register-staged fallback (982 us, profiles/r03_kernel_stats_D3.csv).  Here: the 64-aligned part of the reduction
    split across the chip on the direct-to-LDS kernels, every split writing its fp32 partial sum to a slab of its own
    (HeroGemmEpilogue.split_stride), the ragged rest (< 64 columns) as one more slab, and ONE fold in slab order that also
    rounds to bf16 (hero_fold_slabs) - bit-reproducible (round 4 merged the splits with fp32 atomics: ADVICE r4)."""
    M, N = dy2.shape
    K = Wt.shape[0]
    n1 = N // 64 * 64
    tiles = -(-M // 128) * -(-K // 128)
    split = max(1, min(n1 // 64 // 8, 16, -(-512 // tiles)))
    n_main = L.lib().hero_gemm_splits(n1, split, L.BF16)
    stride = M * K
    slabs = torch.empty((n_main + (1 if n1 < N else 0), M, K), dtype=torch.float32, device=dy2.device)
    if n_main > 1:
        k_gemm(dy2, Wt, slabs, M, K, n1, N, N, K, L.LAYOUT_K, L.LAYOUT_K, L.BF16, out_f32=True, split_k=split, split_stride=stride)
    else:
        k_gemm(dy2, Wt, slabs, M, K, n1, N, N, K, L.LAYOUT_K, L.LAYOUT_K, L.BF16, out_f32=True)
    if n1 < N:
        k_gemm(L.ptr(dy2) + 2 * n1, L.ptr(Wt) + 2 * n1, L.ptr(slabs) + 4 * n_main * stride, M, K, N - n1, N, N, K,
               L.LAYOUT_K, L.LAYOUT_K, L.BF16, out_f32=True)
    out = torch.empty((M, K), dtype=torch.bfloat16, device=dy2.device)
    L.check(L.lib().hero_fold_slabs(L.ptr(slabs), slabs.shape[0], stride, L.ptr(out), stride, L.BF16, L.stream()))
    return out


def _split_for(n_out, n_in, rows, bk):
    """Split of the reduction (the M rows) for dW = dY^T X.  The [n_out, n_in] output has few
    128x128 tiles, so the row range is cut across workgroups and fp32 atomics merge the partial
    sums.  The split minimises a cost model fitted on MI355X (tools/wgrad_sweep.py): a workgroup
    runs at ~2.7 TFLOP/s alone on a CU and ~1.83 when two share it (512 resident slots on 256
    CUs), and the atomic merge moves split x output bytes at ~2 TB/s."""
    tiles = ((n_out + 127) // 128) * ((n_in + 127) // 128)
    ktiles = max(1, -(-rows // bk))
    flops = 2.0 * n_out * n_in * rows
    out_bytes = 4.0 * n_out * n_in
    best, best_t = 1, None
    for s in range(1, 17):
        if s > 1 and ktiles // s < 4:
            break
        blocks = tiles * s
        rounds = -(-blocks // 512)
        last = blocks - (rounds - 1) * 512
        t = flops / blocks * ((rounds - 1) / 1.83e12 + (1 / 2.7e12 if last <= 256 else 1 / 1.83e12))
        if s > 1:
            t += out_bytes * s / 2.0e12
        if best_t is None or t < best_t:
            best, best_t = s, t
    return best


# Weight gradients that ACCUMULATE into the gradient sink are not launched one by one: inside a backward pass they
# are queued and launched together.  Round 3: the queue holds up to WGRAD_BATCH[0] problems - the weight gradients of
# ALL the BertLayers of an encoder reduce over the same rows - and goes out as ONE whole-tile launch
# (hero_wgrad_batch: 6 x 192 = 1152 tiles = 4.5 rounds of the chip, no atomics in the full rounds, bit-reproducible);
# groups with fewer tiles than the chip has CUs, and the boundary micro-steps of a data-parallel run (where a layer's
# gradients must become final - and their bucket's all-reduce start - while backward is still running), go four at a
# time through the stream-K launch of round 2 (hero_wgrad_group).  The queue is flushed when it is full, when the row
# count changes, and by an autograd-engine callback at the end of the backward pass, so nothing outside ever sees a
# pending gradient.  `on_done` (gradient-sink finality for the bucketed all-reduce) runs when the launch is issued.
# The queue keeps dY / X alive until then (~1 GB for HERO-base at 12000 rows).
# Module-level switches below are for lab scripts and tests (set from Python); none of them is read from the environment.
# The package's environment variables are listed in README.md ("Environment variables") and checked by
# tests/test_cpu_boundary.py::test_the_package_reads_only_documented_environment_variables.
_WQ = []
_WQ_TASK = [-1]          # autograd graph task the queued problems belong to
GROUP_WGRADS = [True]       # False: one launch per weight gradient (lab A/B)
WGRAD_BATCH = [32]          # 4: the per-layer stream-K launches of round 2 (lab A/B)
WGRAD_DBIAS_RIDE = [True]   # False: bias gradients as (deferred) column sums of their own instead of riding on hero_wgrad_batch (lab A/B)
# ... and only for reductions of at most this many rows (lab A/B: 0 = never ride).  History: rounds 4-5 summed the dY panels with the
# kernel's LOADER waves (24 KB more LDS reads per k-step on an LDS-bound loop): the tiles that did it ran ~20 % slower and gated
# their round - config 5's 397056-row launches 8.0-8.6 ms against 6.9-7.0 without, so round 6 first took the sums out (-4.2 %
# on config 5, +-0.3 % on the TVR batch; profiles/r06_d4_ride_ab.txt).  Round 6, late: the sums come from the COMPUTE waves - one
# extra MFMA per row block against a constant selector operand, no LDS traffic (csrc/gemm_ws.hip) - and the ride wins at every
# size: D2 6.070 -> 6.005 ms, ragged 7.92 -> 7.86, config 5 (256 videos) 150.3 -> 148.8 (tools/lab/ab_ride.py, same process,
# alternating; profiles/r06_mfma_ride_ab.txt): the 55 MB re-read of dqkv per layer is gone.
WGRAD_RIDE_MAX_ROWS = [1 << 30]
B1_EPILOGUE = [False]       # True: the FFN1 bias gradient from the gelu' GEMM epilogue's fp32 atomics, as in rounds 1-3 (lab A/B)
B1_PARTIALS = [True]        # False: round 4's ride on hero_wgrad_batch (lab A/B); True: per-tile partial sums from the gelu' epilogue
WGRAD_QUEUE_BYTES = [int(os.environ.get("HERO_WGRAD_QUEUE_MB", "4096")) << 20]   # dY bytes the queue may keep alive (config 5
_WQ_BYTES = [0]                                                  # sizes its batch to 90 % of HBM: there it flushes at once)
_WPLANS = {}             # (rows, ((M, N), ...)) -> (device int32 plan, words) or None when the group is too small
_WPLANS_PINNED = set()   # keys whose plan tensor a captured graph references by address


def _wgrad_queue_full(device):
    """Byte budget of the queue (it keeps dY / X alive).  Past the soft cap (HERO_WGRAD_QUEUE_MB, 4 GB) the queue is
    flushed as soon as its tiles fill whole rounds of the chip - or a tail the plan can slice evenly; past the hard cap
    (15 % of the device memory) in any case.  Config 5 is what this is for: one BertLayer there is 5-20 GB of dY and 192
    tiles = 3/4 of a round; round 3 flushed wherever the byte cap fell (192 tiles at the 256-video profiling size), round 4
    waits for the next whole round (256 tiles = a layer and the first weight of the next one)."""
    if _WQ_BYTES[0] <= WGRAD_QUEUE_BYTES[0]:
        return False
    hard = _WQ_HARD.get(device.index)
    if hard is None:
        env = os.environ.get("HERO_WGRAD_QUEUE_HARD_MB")
        hard = (int(env) << 20) if env else max(2 * WGRAD_QUEUE_BYTES[0], int(0.15 * torch.cuda.get_device_properties(device).total_memory))
        _WQ_HARD[device.index] = hard
    if _WQ_BYTES[0] > hard:
        return True
    tiles = sum(-(-e[4] // 192) * -(-e[1].shape[1] // 192) for e in _WQ)
    full, rem = divmod(tiles, 256)
    if rem == 0:
        return True
    p = -(-rem // 8)
    slices = max(1, min(32 // p, 8))
    # a sliced tail runs ~1.4x slower per flop than whole tiles (several k-streams per XCD share less in its L2, every
    # slice adds a tile of atomics: 0.30 vs 0.43 of peak, profiles/r03_kernel_stats_D4.csv vs r03_kernel_stats.csv)
    return tiles / 256.0 >= 0.9 * (full + (1.4 if slices > 1 else 1.0) / slices)


_WQ_HARD = {}


def _wgrad_limit():
    return 4 if (WGRAD_BATCH[0] <= 4 or SINK.wants_overlap()) else min(WGRAD_BATCH[0], 32)


def _wgrad_plan(probs, n, rows, device):
    key = (rows, device.index, tuple((probs[i].M, probs[i].N) for i in range(n)))
    hit = _WPLANS.get(key, False)
    if hit is False:
        buf = np.zeros(8 + 8 * 512 * 16, dtype=np.int32)
        words = L.lib().hero_wgrad_batch_plan(probs, n, rows, buf.ctypes.data, buf.size)
        if words < 0:
            L.check(words)
        hit = (torch.from_numpy(buf[:words].copy()).to(device), words) if words > 0 else None
        # Plans are never dropped while a captured hipGraph may hold their device address (as optim.adamw pins its
        # tables): entries made or used under stream capture are pinned; the rest of the cache (ragged eager batches make
        # a key per distinct row count) is bounded.
        if len(_WPLANS) > 256:
            for k in [k for k in _WPLANS if k not in _WPLANS_PINNED][:128]:
                del _WPLANS[k]
        _WPLANS[key] = hit
    if hit is not None and torch.cuda.is_current_stream_capturing():
        _WPLANS_PINNED.add(key)
    return hit


_CB_TASK = [-1]


def _ensure_flush_callback(task):
    """One engine callback per backward pass flushes whatever is still queued (weight gradients, column sums)."""
    if _CB_TASK[0] != task:
        _CB_TASK[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(deferred_flush)


def deferred_flush():
    _CB_TASK[0] = -1
    wgrad_flush()
    colsum_flush()


# Column sums that ACCUMULATE into the gradient sink (bias gradients of the nn.Linear layers, LayerNorm gamma / beta /
# fused-bias partial folds) are queued the same way and go out as hero_colsum_multi: two launches for up to 64 sums
# instead of two or three tiny launches each (72 launches, 0.39 ms per micro-step in round 2), in a fixed summation
# order (the per-call folds used 8-way fp32 atomics).  Entries keep their source tensors alive.
_CQ = []
_CQ_TASK = [-1]


def _colsum_defer_ok():
    return GROUP_WGRADS[0] and not SINK.wants_overlap() and torch._C._current_graph_task_id() != -1


def _colsum_queue(keep, src_ptr, dst, rows, cols, ld, dtype, on_done=None, dst_rows=None, row_cols=0):
    """dst_rows (int32 [cols / row_cols]) + row_cols: the column sums are added to dst's ROWS dst_rows[j] (hero_hip.h)."""
    task = torch._C._current_graph_task_id()
    if _CQ and _CQ_TASK[0] != task:
        del _CQ[:]                      # left behind by a backward pass that raised
    _CQ_TASK[0] = task
    _ensure_flush_callback(task)
    _CQ.append(((keep, dst_rows), L.Colsum(src_ptr, L.ptr(dst), rows, cols, ld, dtype, 1.0, row_cols, L.ptr(dst_rows)), dst, on_done))
    if len(_CQ) >= 64:
        colsum_flush()


def colsum_flush():
    while _CQ:
        # one launch must not hold the same destination twice (the fold's read-add-write of dst is not atomic): a
        # parameter used twice in the forward pass (the sub-token embedding LayerNorm: subtitles and queries) goes into
        # consecutive launches, which the stream orders
        # ... and "the same destination" means overlapping ADDRESS RANGES, not equal pointers: a problem with
        # indexed destination rows (HeroColsum.dst_rows) covers its whole table from the table base, a constant-row
        # problem on the same table starts at base + row * D - one launch would mix the former's atomicAdd with the
        # latter's plain read-add-write on the same addresses
        part, seen = [], []
        while _CQ and len(part) < 64:
            lo = _CQ[0][1].dst
            hi = lo + 4 * _CQ[0][2].numel()
            if any(lo < b and a < hi for a, b in seen):
                break
            seen.append((lo, hi))
            part.append(_CQ.pop(0))
        n = len(part)
        probs = (L.Colsum * n)(*[e[1] for e in part])
        need = L.lib().hero_colsum_multi_workspace_bytes(probs, n) // 4
        ws = _workspace(max(need, 1), part[0][2].device, slot="colsum_multi")
        L.check(L.lib().hero_colsum_multi(probs, n, L.ptr(ws), L.stream()))
        for e in part:
            if e[3] is not None:
                e[3]()


def wgrad_flush():
    _WQ_BYTES[0] = 0
    while _WQ:
        rows, dtype = _WQ[0][0].shape[0], _WQ[0][0].dtype
        # One launch never holds the same destination twice: full-round tiles of hero_wgrad_batch do a plain (non-atomic)
        # read-add-write of dW and of the riding bias sums, so a parameter used twice in one backward pass with equal row
        # counts (a shared / tied nn.Linear, fuse_query_pass=False with coinciding row counts) goes into consecutive
        # launches, which the stream orders - as colsum_flush does.
        dst_key = lambda e: (e[2].data_ptr(), e[6].data_ptr() if e[6] is not None else None)     # noqa: E731
        seen_w, seen_b = {_WQ[0][2].data_ptr()}, {dst_key(_WQ[0])[1]} - {None}
        n = 1
        while n < len(_WQ) and n < 32 and _WQ[n][0].shape[0] == rows and _WQ[n][0].dtype == dtype:
            kw, kb = dst_key(_WQ[n])
            if kw in seen_w or (kb is not None and kb in seen_b):
                break
            seen_w.add(kw)
            if kb is not None:
                seen_b.add(kb)
            n += 1
        group, rest = _WQ[:n], _WQ[n:]
        del _WQ[:]
        _WQ.extend(rest)
        probs = (L.WgradProblem * n)()
        for i, (dy2, x2, out, col0, N, _, _, _) in enumerate(group):
            K = x2.shape[1]
            probs[i] = L.WgradProblem(L.ptr(dy2) + col0 * dy2.element_size(), L.ptr(x2), L.ptr(out), N, K,
                                      dy2.shape[1], K, K, _split_for(N, K, rows, 64 if dtype == torch.bfloat16 else 32), None)
        plan = _wgrad_plan(probs, n, rows, group[0][0].device) if dtype == torch.bfloat16 else None
        if plan is not None:
            ride = WGRAD_DBIAS_RIDE[0] and rows <= WGRAD_RIDE_MAX_ROWS[0]
            for i, e in enumerate(group):                   # bias gradients ride on the dY panels of the batch kernel
                probs[i].dbias = L.ptr(e[6]) if ride else None
                if not ride and e[6] is not None:
                    k_colsum(e[0], out=e[6], beta=1.0, col0=e[3], ncols=e[4], on_done=lambda: None)
            L.check(L.lib().hero_wgrad_batch(probs, n, rows, L.BF16, L.ptr(plan[0]), plan[1], L.stream()))
            for e in group:
                if e[6] is not None and e[7] is not None:
                    e[7]()
        else:
            for g0 in range(0, n, 4):
                sub = (L.WgradProblem * min(4, n - g0))(*[probs[i] for i in range(g0, min(g0 + 4, n))])
                L.check(L.lib().hero_wgrad_group(sub, len(sub), rows, L.dt(group[0][0]), L.stream()))
            for dy2, _, _, col0, N, _, dbias, dbias_done in group:
                if dbias is not None:
                    k_colsum(dy2, out=dbias, beta=1.0, col0=col0, ncols=N, on_done=dbias_done or (lambda: None))
        for e in group:
            if e[5] is not None:
                e[5]()


def k_wgrad(dy2, x2, out=None, beta=0.0, col0=0, ncols=None, on_done=None, dbias=None, dbias_done=None):
    """dW[N,K] (fp32) = beta*dW + dy2[:, col0:col0+N]^T @ x2[M,K].
    dbias ([N] fp32, accumulated): the bias gradient of the same layer = column sums of the same dY columns.  When the
    problem goes out through hero_wgrad_batch they are taken from the dY panels that kernel streams anyway; otherwise
    they become a (deferred) hero_colsum of their own."""
    M, ld = dy2.shape
    N = ld - col0 if ncols is None else ncols
    K = x2.shape[1]
    if (out is not None and beta == 1.0 and GROUP_WGRADS[0] and out.is_contiguous() and dy2.is_contiguous()
            and x2.is_contiguous() and torch._C._current_graph_task_id() != -1):
        task = torch._C._current_graph_task_id()
        if _WQ and _WQ_TASK[0] != task:
            del _WQ[:]                  # left behind by a backward pass that raised: never launch them into this one
            _WQ_BYTES[0] = 0
        if _WQ and (_WQ[0][0].shape[0] != M or _WQ[0][0].dtype != dy2.dtype):
            wgrad_flush()
        if not _WQ:
            _WQ_TASK[0] = task
        _ensure_flush_callback(task)
        _WQ.append((dy2, x2, out, col0, N, on_done, dbias, dbias_done))
        _WQ_BYTES[0] += dy2.numel() * dy2.element_size()
        if len(_WQ) >= _wgrad_limit() or _wgrad_queue_full(dy2.device):
            wgrad_flush()
        return out
    if dbias is not None:
        k_colsum(dy2, out=dbias, beta=1.0, col0=col0, ncols=N, on_done=dbias_done or (lambda: None))
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=dy2.device)
    split = _split_for(N, K, M, 64 if dy2.dtype == torch.bfloat16 else 32)
    a = L.ptr(dy2) + col0 * dy2.element_size()
    k_gemm(a, x2, out, N, K, M, ld, K, K, L.LAYOUT_O, L.LAYOUT_O, L.dt(dy2), out_f32=True,
           beta=beta, split_k=split)
    if on_done is not None:
        on_done()
    return out


_WS = {}


def _workspace(n_floats, device, slot=""):
    key = (device.type, device.index, slot)
    t = _WS.get(key)
    if t is None or t.numel() < n_floats:
        t = torch.empty((max(n_floats, 1 << 20),), dtype=torch.float32, device=device)
        _WS[key] = t
    return t


def k_colsum(dy2, out=None, beta=0.0, col0=0, ncols=None, on_done=None):
    """on_done given: the sum may be DEFERRED to the end of the backward pass (see colsum_flush)."""
    M, ld = dy2.shape
    N = ld - col0 if ncols is None else ncols
    if (on_done is not None and out is not None and beta == 1.0 and dy2.is_contiguous() and N % 4 == 0 and ld % 4 == 0
            and (col0 * dy2.element_size()) % 8 == 0 and _colsum_defer_ok()):
        _colsum_queue(dy2, L.ptr(dy2) + col0 * dy2.element_size(), out, M, N, ld, L.dt(dy2), on_done)
        return out
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=dy2.device)
    ws = _workspace(256 * N, dy2.device)      # stream-ordered reuse
    L.check(L.lib().hero_colsum(L.ptr(dy2) + col0 * dy2.element_size(), L.ptr(out), M, N, ld,
                                L.dt(dy2), beta, L.ptr(ws), L.stream()))
    if on_done is not None:
        on_done()
    return out


def acc_linear_grads(dy2, x2, weight, bias, col0=0, ncols=None):
    """weight.grad += dy^T x ; bias.grad += colsum(dy) straight into the gradient sink."""
    N = ncols if ncols is not None else weight.shape[0]
    want_b = bias is not None and bias.requires_grad
    if weight is not None and weight.requires_grad:
        k_wgrad(dy2, x2, out=SINK.dst(weight), beta=1.0, col0=col0, ncols=N, on_done=lambda: SINK.done(weight),
                dbias=SINK.dst(bias) if want_b else None, dbias_done=(lambda: SINK.done(bias)) if want_b else None)
    elif want_b:
        k_colsum(dy2, out=SINK.dst(bias), beta=1.0, col0=col0, ncols=N, on_done=lambda: SINK.done(bias))


def k_ln_fwd(x2, gamma, beta, eps, out_dtype, rows, cols, tabs=(), idxs=(), want_pre=False,
             drop=None, device=None):
    device = device or (x2.device if x2 is not None else gamma.device)
    y = torch.empty((rows, cols), dtype=out_dtype, device=device)
    mean = torch.empty((rows,), dtype=torch.float32, device=device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=device)
    pre = torch.empty((rows, cols), dtype=out_dtype, device=device) if want_pre else None
    a = L.LnFwd()
    a.x = L.ptr(x2)
    for k in range(3):
        a.tab[k] = L.ptr(tabs[k]) if k < len(tabs) else None
        a.idx[k] = L.ptr(idxs[k]) if k < len(idxs) and idxs[k] is not None else None
    a.gamma, a.beta, a.y, a.pre = L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(pre)
    a.mean, a.rstd = L.ptr(mean), L.ptr(rstd)
    a.rows, a.cols, a.eps = rows, cols, eps
    a.x_dtype = L.dt(x2) if x2 is not None else L.dt(y)
    a.y_dtype = L.dt(y)
    a.dropout = _d(drop)
    L.check(L.lib().hero_layernorm_fwd(C.byref(a), L.stream()))
    return y, mean, rstd, pre


def ln_param_dsts(gamma_p, beta_p):
    """(dgamma dst, dbeta dst) in the gradient sink for LayerNorm parameters (None if frozen)."""
    dg = SINK.dst(gamma_p) if gamma_p.requires_grad else None
    db = SINK.dst(beta_p) if beta_p.requires_grad else None
    return dg, db


def ln_params_done(gamma_p, beta_p):
    if gamma_p.requires_grad:
        SINK.done(gamma_p)
    if beta_p.requires_grad:
        SINK.done(beta_p)


def k_ln_bwd(x2, dy2, gamma, mean, rstd, want_dx=True, want_params=True, drop_out=None,
             drop_in=None, dgamma=None, dbeta=None, grad_beta=0.0, dbias_in=None):
    """dgamma/dbeta given: accumulate into them with grad_beta (the gradient sink path)."""
    rows, cols = dy2.shape
    dev = dy2.device
    dx = torch.empty_like(dy2) if want_dx else None
    dxd = torch.empty_like(dy2) if (want_dx and drop_in is not None) else None
    dg, db = dgamma, dbeta
    if want_params and dg is None and db is None:
        dg = torch.empty((cols,), dtype=torch.float32, device=dev)
        db = torch.empty((cols,), dtype=torch.float32, device=dev)
    ws = _workspace(3072 * cols, dev) if (dg is not None or db is not None or dbias_in is not None) else None
    defer = (grad_beta == 1.0 and ws is not None and cols <= 1024 and (want_dx or dbias_in is not None) and _colsum_defer_ok())
    if defer:           # the fused kernel's per-block partial sums stay in a buffer of their own until the multi-fold
        nblk = L.lib().hero_layernorm_bwd_blocks(rows)
        ws = torch.empty((nblk, 3 * cols), dtype=torch.float32, device=dev)
    a = L.LnBwd()
    a.x, a.dy, a.gamma, a.mean, a.rstd = L.ptr(x2), L.ptr(dy2), L.ptr(gamma), L.ptr(mean), L.ptr(rstd)
    a.dx, a.dx_dropped, a.dgamma, a.dbeta = L.ptr(dx), L.ptr(dxd), L.ptr(dg), L.ptr(db)
    a.grad_beta, a.workspace = grad_beta, L.ptr(ws)
    a.rows, a.cols, a.x_dtype, a.dtype = rows, cols, L.dt(x2), L.dt(dy2)
    a.dropout_out, a.dropout_in = _d(drop_out), _d(drop_in)
    a.dbias_in = L.ptr(dbias_in)
    a.defer_fold = 1 if defer else 0
    L.check(L.lib().hero_layernorm_bwd(C.byref(a), L.stream()))
    if defer:
        for k, dst in enumerate((dg, db, dbias_in)):
            if dst is not None:
                _colsum_queue(ws, L.ptr(ws) + 4 * k * cols, dst, ws.shape[0], cols, 3 * cols, L.F32)
    return dx, (dxd if dxd is not None else dx), dg, db


ATTN_SAVE_PROBS = False     # tests: keep the fp32 probabilities for the backward instead of the softmax row statistics


def k_attn_fwd(qkv, mask_add, S, Lq, H, drop=None, want_probs=True, out=None, seq_off=None):
    """seq_off (int32 [S+1]): packed batch - sequence s is rows [seq_off[s], seq_off[s+1]) of qkv,
    Lq is the maximum length (<= 64; bf16: <= 256).
    Returns (ctx, saved): `saved` is what k_attn_bwd needs - the fp32 probabilities [S, H, Lq, Lq], or, where the
    kernels can rebuild them (hero_attention_stats_ok: bf16, Lq <= 64), the softmax row statistics as a FLAT fp32
    tensor [S*H*Lq*2] (13.8 MB -> 1.2 MB per layer on the bench batch)."""
    D = H * 64
    ctx = out if out is not None else torch.empty((qkv.shape[0], D), dtype=qkv.dtype, device=qkv.device)
    probs = stats = None
    if want_probs:
        if not ATTN_SAVE_PROBS and L.lib().hero_attention_stats_ok(L.dt(qkv), Lq):
            stats = torch.empty((S * H * Lq * 2,), dtype=torch.float32, device=qkv.device)
        else:
            probs = torch.empty((S, H, Lq, Lq), dtype=torch.float32, device=qkv.device)
    a = L.Attn(L.ptr(qkv), L.ptr(mask_add), L.ptr(ctx), L.ptr(probs), None, None, S, Lq, H,
               1.0 / math.sqrt(64.0), L.dt(qkv), _d(drop), L.ptr(seq_off), L.ptr(stats))
    L.check(L.lib().hero_attention_fwd(C.byref(a), L.stream()))
    return ctx, (stats if stats is not None else probs)


def k_attn_bwd(qkv, saved, dctx, S, Lq, H, drop=None, out=None, seq_off=None, ctx=None, mask_add=None):
    """saved: k_attn_fwd's second result.  ctx: the forward output (optional).  With it the bf16 backward of
    64 < Lq <= 256 runs on the matrix-core kernels (delta_i = dO_i . ctx_i); without it those lengths take the
    fp32-VALU path.  mask_add: the forward's additive mask - needed when `saved` holds row statistics."""
    dqkv = out if out is not None else torch.empty_like(qkv)
    is_stats = saved.dim() == 1
    a = L.Attn(L.ptr(qkv), L.ptr(mask_add) if is_stats else None, L.ptr(ctx), None if is_stats else L.ptr(saved), L.ptr(dctx),
               L.ptr(dqkv), S, Lq, H, 1.0 / math.sqrt(64.0), L.dt(qkv), _d(drop), L.ptr(seq_off), L.ptr(saved) if is_stats else None)
    L.check(L.lib().hero_attention_bwd(C.byref(a), L.stream()))
    return dqkv


_SLACK = {}        # storage address -> weak reference to a row buffer allocated with spare rows behind its result


def k_gather_rows(a, b, idx, rows, cols, tail_rows=0):
    """tail_rows: allocate that many spare rows BEHIND the result (same storage): a following row stack / gradient
    stack writes its small remaining blocks there instead of copying this (large) block (StackRowsFn, SplitRowsFn)."""
    ref = a if a is not None else b
    if tail_rows > 0:
        base = torch.empty((rows + tail_rows, cols), dtype=ref.dtype, device=ref.device)
        _SLACK[base.untyped_storage().data_ptr()] = weakref.ref(base)      # alive as long as any view of it is
        if len(_SLACK) > 64:
            for k in [k for k, r in _SLACK.items() if r() is None]:
                del _SLACK[k]
        out = base[:rows]
    else:
        out = torch.empty((rows, cols), dtype=ref.dtype, device=ref.device)
    L.check(L.lib().hero_gather_rows(L.ptr(a), L.ptr(b), L.ptr(idx), L.ptr(out), rows, cols,
                                     L.dt(ref), L.stream()))
    return out


def k_scatter_add(src2, idx, dst_a, dst_b=None, skip=-1):
    rows, cols = src2.shape
    L.check(L.lib().hero_scatter_add_rows(L.ptr(src2), L.ptr(idx), L.ptr(dst_a), L.ptr(dst_b), rows,
                                          cols, L.dt(src2), L.dt(dst_a), skip, L.stream()))


def segment_order(idx, n_dst, skip=-1):
    """Rows sorted by destination (hero_segment_sort), cached per index tensor like every derived batch tensor (memo:
    the same batch object runs several micro-steps; refresh_memo redoes it when new ids are written into the buffers)."""
    def build():
        order = torch.empty(idx.numel(), dtype=torch.int32, device=idx.device)
        ws = torch.empty(max(L.lib().hero_segment_sort_workspace_bytes(idx.numel()) // 4, 1), dtype=torch.int32, device=idx.device)
        L.check(L.lib().hero_segment_sort(L.ptr(idx), idx.numel(), n_dst, skip, L.ptr(order), L.ptr(ws), L.stream()))
        return order
    return memo("seg_order", (idx,), build, (n_dst, skip))


def k_scatter_add_sorted(src2, idx, dst, skip=-1):
    """dst[idx[r]] += src2[r] without atomics, bit-reproducible (embedding-table gradients); dst fp32 [n_dst, cols]."""
    rows, cols = src2.shape
    order = segment_order(idx, dst.shape[0], skip)
    ws = _workspace(L.lib().hero_scatter_add_sorted_workspace_bytes(rows, cols) // 4, src2.device, slot="scatter_sorted")
    L.check(L.lib().hero_scatter_add_sorted(L.ptr(src2), L.ptr(idx), L.ptr(order), L.ptr(dst), rows, cols, L.dt(src2), skip,
                                            L.ptr(ws), L.stream()))


def k_cast(x, dtype):
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    L.check(L.lib().hero_cast(L.ptr(x), L.ptr(y), x.numel(), L.dt(x), L.dt(y), L.stream()))
    return y


def k_act_bwd(dy, aux, act):
    dx = torch.empty_like(dy)
    fn = L.lib().hero_relu_bwd if act == L.ACT_RELU else L.lib().hero_gelu_bwd
    L.check(fn(L.ptr(dy), L.ptr(aux), L.ptr(dx), dy.numel(), L.dt(dy), L.stream()))
    return dx


def _as2d(x):
    if not x.is_contiguous():
        x = x.contiguous()
    return x.view(-1, x.shape[-1])


_MEMO = {}


def memo(tag, tensors, fn, extra=(), spec=None):
    """Cache a tensor DERIVED from batch index / mask tensors (int32 row indices, additive masks,
    flat gather maps) on the identity + version of its sources: the same batch object is fed for
    several micro-steps (gradient accumulation, hipGraph warm-up) and these conversions are a few
    dozen tiny launches each time.  The sources are kept alive so their addresses stay unique.
    spec = (L.DERIVE_* mode, p0, p1, p2): the entry is an elementwise function of ONE contiguous int64 source that
    hero_derive_multi computes - refresh_memo then redoes all such entries in one launch."""
    key = (tag, extra) + tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype) for t in tensors)
    hit = _MEMO.get(key)
    if hit is not None:
        return hit[0]
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        # A miss while a hipGraph is being captured is computed INSIDE that graph and not remembered: a remembered entry would
        # live in the graph's private pool and be found - not recomputed - by the next capture (TrainStep records a plain
        # and a boundary graph back to back), which would then depend on the other graph's last replay (round 6: found by
        # the bucketed feeder, whose commits move the version counters between a bucket's first eager step and its capture).
        return fn()
    if len(_MEMO) > 512:
        _MEMO.clear()
    out = fn()
    if spec is not None and not (len(tensors) == 1 and tensors[0].dtype == torch.int64 and tensors[0].is_contiguous()
                                 and tensors[0].is_cuda and out.is_contiguous() and out.numel() == tensors[0].numel()):
        spec = None
    _MEMO[key] = (out, tensors, fn, spec)
    return out


def reset_caches():
    """Forget everything this module caches about models and batches that are no longer in use: compute copies of weights
    (they keep their parameters - whole models - alive, and refresh_weight_cache() re-derives ALL of them after every
    optimiser step), memoised batch tensors, workspaces, weight-gradient plans, the gradient sink.  For processes that build
    several models in a row (bench.py's secondary workloads: each would otherwise pay for its predecessors' copies).
    Not while a captured hipGraph that references those buffers is still going to be replayed."""
    clear_weight_cache()
    _REFRESH.clear()
    _REFRESH_PINNED.clear()
    _MEMO.clear()
    _WS.clear()
    _WPLANS.clear()
    _WPLANS_PINNED.clear()
    _SEQ_OFF.clear()
    _SLACK.clear()
    del _WQ[:]
    del _CQ[:]
    set_grad_sink(None)
    from .model.layers import BertEncoder          # pack plans keep their mask tensors (whole batches) alive
    BertEncoder._PLANS.clear()
    BertEncoder._STATIC_PLANS.clear()


def host_sortable_orders(batch):
    """The memoised segment orders (`segment_order`: rows sorted by destination for the embedding-table gradients) that are a
    function of ONE int64 id tensor of `batch`: [(batch key, order tensor, skip index)].  A feeder that gets its ids from the
    host can compute such an order there (numpy stable argsort, hidden behind the GPU's step) and copy it in, instead of
    re-sorting on the device inside every commit (hero_segment_sort is one workgroup: 69 us for 9600 ids)."""
    # by address (callers pass views, idx.view(-1)) - and ONLY the plain int32 casts of a batch tensor (`row_index`, spec
    # DERIVE_I32): an index derived any other way (a flat gather map ...) must keep its device sort
    by_out = {v[0].data_ptr(): v for k_, v in _MEMO.items()
              if isinstance(v[0], torch.Tensor) and k_[0] == "row_index" and v[3] is not None and v[3][0] == L.DERIVE_I32}
    keys = {t.data_ptr(): k for k, t in batch.items() if torch.is_tensor(t) and t.dtype == torch.int64}
    found = []
    for mk, (out, srcs, fn, spec) in _MEMO.items():
        if mk[0] != "seg_order" or len(srcs) != 1:
            continue
        ridx = by_out.get(srcs[0].data_ptr())                  # the int32 row index the order was sorted from ...
        if ridx is None or len(ridx[1]) != 1:
            continue
        ids = ridx[1][0]                                       # ... itself derived from this int64 tensor
        k = keys.get(ids.data_ptr())
        if k is not None and batch[k].numel() == out.numel():
            found.append((k, out, int(mk[1][1])))
    return found


def host_segment_order(ids, skip):
    """numpy twin of hero_segment_sort: row numbers sorted by (id, row), rows with id < 0 or == skip last (stable)."""
    v = np.asarray(ids, dtype=np.int64).reshape(-1)
    key = np.where((v < 0) | (v == skip), np.int64(1) << 40, v)
    return np.argsort(key, kind="stable").astype(np.int32)


def refresh_memo(sources=None, skip_outputs=()):
    """Recompute memoised derived tensors IN PLACE from the current contents of their sources.  For callers that
    rewrite batch buffers in place (hero_amd.collate.DeviceCollate, or new data copied into a captured batch):
    captured graphs and cached maps hold the derived tensors by address.  sources: only the entries derived - directly or
    through other entries - from one of these tensors (default: every entry).  Entries with a `spec` go out together as
    hero_derive_multi launches.  Order: an entry is redone once every entry it is derived from has been (waves of a
    topological order over the output addresses; round 6: a 0/1 mask gathered per (query, video) pair and THEN cast to fp32
    is a spec entry behind a builder entry - the old "spec entries first" order left it one batch behind)."""
    skip_ids = {id(t) for t in skip_outputs}                  # entries the caller refreshes itself (host-computed orders)
    entries = [e for e in _MEMO.values() if id(e[0]) not in skip_ids]
    if sources is not None:                                    # what this call keeps current (the caller may restamp_memo them)
        src_ptrs = {t.data_ptr() for t in sources}
        _LAST_REFRESHED[:] = [k for k, e in _MEMO.items() if any(t.data_ptr() in src_ptrs for t in e[1])]
        dirty = set(src_ptrs)
        hit = [False] * len(entries)
        grew = True
        while grew:                                            # everything downstream of the sources
            grew = False
            for n, (out, srcs, fn, spec) in enumerate(entries):
                if not hit[n] and any(t.data_ptr() in dirty for t in srcs):
                    hit[n] = grew = True
                    if isinstance(out, torch.Tensor):
                        dirty.add(out.data_ptr())
        todo = [e for n, e in enumerate(entries) if hit[n]]
    else:
        todo = list(entries)
    waiting = {e[0].data_ptr() for e in todo if isinstance(e[0], torch.Tensor)}      # outputs not yet redone
    while todo:
        ready = [e for e in todo if not any(t.data_ptr() in waiting and t is not e[0] for t in e[1])]
        if not ready:
            raise RuntimeError("refresh_memo: memo entries derive from each other in a cycle")
        batch = [L.Derive(L.ptr(srcs[0]), L.ptr(out), out.numel(), spec[0], spec[1], spec[2], spec[3])
                 for out, srcs, fn, spec in ready if spec is not None]
        for i in range(0, len(batch), 16):
            part = batch[i:i + 16]
            L.check(L.lib().hero_derive_multi((L.Derive * len(part))(*part), len(part), L.stream()))
        for out, srcs, fn, spec in ready:
            if spec is not None:
                continue
            new = fn()
            if isinstance(out, torch.Tensor):
                if new.shape != out.shape:
                    raise RuntimeError("refresh_memo: a derived tensor changed shape %s -> %s; the batch structure is "
                                       "different, not just its contents" % (tuple(out.shape), tuple(new.shape)))
                out.copy_(new)
        ids = {id(e) for e in ready}
        for e in ready:
            if isinstance(e[0], torch.Tensor):
                waiting.discard(e[0].data_ptr())
        todo = [e for e in todo if id(e) not in ids]


_LAST_REFRESHED = []


def last_refreshed_keys():
    """Keys of the memo entries derived DIRECTLY from the `sources` of the last refresh_memo(sources=...) call."""
    return list(_LAST_REFRESHED)


def restamp_memo(keys):
    """Re-key memo entries to the CURRENT version counters of their sources and return the new keys.  For a caller that has
    just refreshed exactly these entries in place by other means - a replayed hipGraph of refresh_memo - and then moved the
    sources' version counters (hero_amd.loader.StaticBatchFeeder.commit): the entries ARE current, and a later lookup - an
    eager step, or the capture of another step graph - must find them instead of rebuilding beside them."""
    out = []
    for k in keys:
        e = _MEMO.pop(k, None)
        if e is None:
            continue
        nk = (k[0], k[1]) + tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype) for t in e[1])
        _MEMO[nk] = e
        out.append(nk)
    return out


def as_mask_add(mask, S, Lq):
    """Reference mask convention -> [S, L] additive fp32.  Accepts the (S,L) 0/1 mask of
    BertEncoder.forward or the extended (S,1,1,L) additive mask BertLayer receives."""
    if mask is None:
        return None
    if mask.dim() == 4:
        if mask.shape[1] != 1 or mask.shape[2] != 1:
            raise NotImplementedError("hero_amd attention supports key masks of shape (S,1,1,L) only")
        return mask.reshape(S, Lq).to(torch.float32).contiguous()
    if mask.requires_grad:
        return ((1.0 - mask.reshape(S, Lq).to(torch.float32)) * -10000.0).contiguous()
    return memo("mask_add", (mask,), lambda: ((1.0 - mask.reshape(S, Lq).to(torch.float32)) * -10000.0).contiguous(),
                (S, Lq), spec=(L.DERIVE_MASK_ADD, 0, 0, 0))


# --------------------------------------------------------------------------------------------- #
# autograd functions
# --------------------------------------------------------------------------------------------- #
class CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return k_cast(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return k_cast(dy.contiguous(), ctx.src), None


def cast(x, dtype):
    return x if x.dtype == dtype else CastFn.apply(x, dtype)


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) [+ residual]; act in {none, relu, gelu}.  (nn.Linear call sites outside
    the fused blocks: img_linear, frame_transform, query_input_proj, standalone sub-modules.)
    Leaf-parameter gradients go to the gradient sink; other weight tensors get normal gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, residual):
        x2 = _as2d(x)
        Wc = packed((weight,), x2.dtype)
        aux = torch.empty((x2.shape[0], Wc.shape[0]), dtype=x2.dtype, device=x2.device) \
            if act != L.ACT_NONE else None
        r2 = _as2d(residual) if residual is not None else None
        y = k_linear(x2, Wc, bias.detach() if bias is not None else None, act=act, aux=aux,
                     residual=r2)
        ctx.act = act
        ctx.has_res = residual is not None
        ctx.xshape = x.shape
        ctx.params = (weight, bias)
        ctx.sink = _is_param(weight) and (bias is None or _is_param(bias))
        if ctx.sink:
            _use(weight, bias)
        Wt = packed_t((weight,), x2.dtype) if ctx.needs_input_grad[0] else None
        ctx.save_for_backward(x2, Wt, aux)
        return y.view(*x.shape[:-1], Wc.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, Wt, aux = ctx.saved_tensors
        weight, bias = ctx.params
        dy2 = _as2d(dy)
        dz = dy2 if ctx.act == L.ACT_NONE else k_act_bwd(dy2, aux, ctx.act)
        dx = k_dgrad_t(dz, Wt).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.sink:
            acc_linear_grads(dz, x2, weight, bias)
        else:
            dW = k_wgrad(dz, x2) if ctx.needs_input_grad[1] else None
            db = k_colsum(dz) if (bias is not None and ctx.needs_input_grad[2]) else None
        return dx, dW, db, None, (dy if ctx.has_res else None)


def linear(x, weight, bias=None, act=L.ACT_NONE, residual=None):
    if x.dtype == torch.bfloat16 and weight.shape[0] % 8 != 0:
        # a bf16 GEMM operand row must be a multiple of 16 bytes; narrow heads whose width is not (the 100
        # frame-order classes of fom_output, model/model.py:118) run in fp32 - they are tiny
        x = cast(x, torch.float32)
        residual = cast(residual, torch.float32) if residual is not None else None
    return LinearFn.apply(x, weight, bias, act, residual)


class MatmulNTFn(torch.autograd.Function):
    """y = a @ b^T for two ACTIVATION tensors in the compute dtype (no weight cache): the MFM-NCE logits
    masked_output @ [pos; neg]^T (model/model.py:271-291).  b's rows are padded to a multiple of 8 by the caller."""

    @staticmethod
    def forward(ctx, a, b):
        a2, b2 = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a2, b2)
        return k_linear(a2, b2)

    @staticmethod
    def backward(ctx, dy):
        a2, b2 = ctx.saved_tensors
        dy2 = dy.contiguous()
        da = k_dgrad(dy2, b2) if ctx.needs_input_grad[0] else None
        db = k_wgrad(dy2, a2).to(b2.dtype) if ctx.needs_input_grad[1] else None
        return da, db


def matmul_nt(a, b):
    return MatmulNTFn.apply(a, b)


class CrossEntropyFn(torch.autograd.Function):
    """Per-row softmax cross-entropy of [rows, ld] logits over their first `ncols` columns (hero_cross_entropy_*);
    reduction 'none', rows labelled `ignore_index` give 0."""

    @staticmethod
    def forward(ctx, logits, labels, ncols, ignore_index, inv_temp):
        x2 = logits.contiguous()
        rows, ld = x2.shape
        lab = labels.contiguous().to(torch.int64)
        loss = torch.empty((rows,), dtype=torch.float32, device=x2.device)
        lse = torch.empty((rows,), dtype=torch.float32, device=x2.device)
        a = L.CrossEntropy(L.ptr(x2), L.ptr(lab), L.ptr(loss), L.ptr(lse), None, None, rows, ncols, ld, L.dt(x2),
                           inv_temp, ignore_index)
        L.check(L.lib().hero_cross_entropy_fwd(C.byref(a), L.stream()))
        ctx.save_for_backward(x2, lab, lse)
        ctx.meta = (ncols, ignore_index, inv_temp)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        x2, lab, lse = ctx.saved_tensors
        ncols, ignore_index, inv_temp = ctx.meta
        rows, ld = x2.shape
        g = dloss.contiguous().float()
        dx = torch.empty_like(x2)
        a = L.CrossEntropy(L.ptr(x2), L.ptr(lab), None, L.ptr(lse), L.ptr(g), L.ptr(dx), rows, ncols, ld, L.dt(x2),
                           inv_temp, ignore_index)
        L.check(L.lib().hero_cross_entropy_bwd(C.byref(a), L.stream()))
        return dx, None, None, None, None


def cross_entropy(logits, labels, ncols=None, ignore_index=-100, inv_temp=1.0):
    return CrossEntropyFn.apply(logits, labels, logits.shape[1] if ncols is None else ncols, ignore_index,
                                float(inv_temp))


class EmbedLnFn(torch.autograd.Function):
    """y = dropout(LN(x + sum_k table_k[idx_k])) — embedding sums of model/embed.py fused with their
    LayerNorm.  idx_k None = row 0 of the given (fp32) table for every row, an int = that fixed row
    (the token-type row: slicing the parameter instead would cost a zeros + copy + add in autograd).
    Gradients of leaf tables are scatter-added straight into the gradient sink (no dense temporary)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, drop, out_dtype, skip_idx, idxs, *tables):
        x2 = _as2d(x) if x is not None else None
        cols = gamma.shape[0]
        rows = x2.shape[0] if x2 is not None else idxs[0].numel()
        tabs = [t.detach()[i:i + 1] if isinstance(i, int) else t.detach()
                for t, i in zip(tables, tuple(idxs) + (None,) * len(tables))]
        y, mean, rstd, pre = k_ln_fwd(x2, gamma.detach(), beta.detach(), eps, out_dtype, rows, cols,
                                      tabs=tabs, idxs=[None if isinstance(i, int) else i for i in idxs],
                                      want_pre=len(tables) > 0, drop=drop, device=gamma.device)
        ctx.drop, ctx.idxs, ctx.skip = drop, idxs, skip_idx
        ctx.tables = tables
        ctx.ln = (gamma, beta)
        ctx.ln_sink = _is_param(gamma) and _is_param(beta)
        _use(*[t for t in tables if _is_param(t)])
        if ctx.ln_sink:
            _use(gamma, beta)
        ctx.has_x = x is not None
        ctx.xshape = x.shape if x is not None else None
        ctx.save_for_backward(pre if pre is not None else x2, gamma.detach(), mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        pre, gamma, mean, rstd = ctx.saved_tensors
        dy2 = _as2d(dy)
        gp, bp = ctx.ln
        want_dx = (ctx.has_x and ctx.needs_input_grad[0]) or len(ctx.tables) > 0
        if ctx.ln_sink:
            dgd, dbd = ln_param_dsts(gp, bp)
            dx, _, _, _ = k_ln_bwd(pre, dy2, gamma, mean, rstd, want_dx=want_dx, drop_out=ctx.drop,
                                   dgamma=dgd, dbeta=dbd, grad_beta=1.0, want_params=False)
            ln_params_done(gp, bp)
            dg = db = None
        else:
            dx, _, dg, db = k_ln_bwd(pre, dy2, gamma, mean, rstd, want_dx=want_dx, drop_out=ctx.drop)
        grads_t = []
        for k, tab in enumerate(ctx.tables):
            idx = ctx.idxs[k] if k < len(ctx.idxs) else None
            skip = ctx.skip[k] if ctx.skip else -1
            row, idx = (idx, None) if isinstance(idx, int) else (0, idx)
            if _is_param(tab):
                per = getattr(idx, "_hero_period", 0) if idx is not None else 0
                if idx is None:
                    k_colsum(dx, out=SINK.dst(tab).view(-1, tab.shape[-1])[row], beta=1.0, on_done=lambda tab=tab: SINK.done(tab))
                    grads_t.append(None)
                    continue
                elif per and dx.shape[0] % per == 0 and dx.shape[0] // per >= 8:
                    # periodic index (position ids broadcast over the sequences): fold the repeats with a column sum
                    # over [S, period*D] and add the `period` rows to the table - not S-way contended atomics.  Round 4:
                    # the fold joins the deferred multi-sum of the backward pass (its destination rows are indexed,
                    # HeroColsum.dst_rows) instead of two reduction launches + a scatter of its own per table
                    D_ = dx.shape[1]
                    if (skip is None or skip < 0) and dx.is_contiguous() and (per * D_) % 4 == 0 and _colsum_defer_ok():
                        _colsum_queue(dx, L.ptr(dx), SINK.dst(tab), dx.shape[0] // per, per * D_, per * D_, L.dt(dx),
                                      on_done=lambda tab=tab: SINK.done(tab), dst_rows=idx[:per].contiguous(), row_cols=D_)
                        grads_t.append(None)
                        continue
                    folded = k_colsum(dx.view(dx.shape[0] // per, per * dx.shape[1]))
                    k_scatter_add(folded.view(per, dx.shape[1]), idx[:per], SINK.dst(tab), None, skip)
                elif (idx.dtype == torch.int32 and idx.is_contiguous() and dx.shape[1] % 4 == 0
                        and dx.shape[0] <= (1 << 17)):      # (beyond: a 6 KB-per-16-rows workspace; config 5 keeps the atomics)
                    k_scatter_add_sorted(dx, idx.view(-1), SINK.dst(tab).view(-1, tab.shape[-1]), skip)
                else:
                    k_scatter_add(dx, idx, SINK.dst(tab), None, skip)
                SINK.done(tab)
                grads_t.append(None)
            elif not ctx.needs_input_grad[8 + k]:
                grads_t.append(None)
            else:
                g = torch.zeros(tab.shape, dtype=torch.float32, device=dy.device)
                if idx is None:
                    k_colsum(dx, out=g.view(-1, tab.shape[-1])[row])
                else:
                    k_scatter_add(dx, idx, g, None, skip)
                grads_t.append(g)
        gx = dx.view(ctx.xshape) if (ctx.has_x and ctx.needs_input_grad[0]) else None
        return (gx, dg, db, None, None, None, None, None) + tuple(grads_t)


def embed_ln(x, gamma, beta, eps, drop, out_dtype, tables=(), idxs=(), skip_idx=None):
    return EmbedLnFn.apply(x, gamma, beta, eps, drop, out_dtype, skip_idx, tuple(idxs), *tables)


def _stack_in_place(blocks, rows, cols):
    """If the FIRST block was allocated with enough spare rows behind it (k_gather_rows(tail_rows=...)), the stacked
    tensor is that storage: the other (small) blocks are copied into the spare rows and a [sum rows, cols] view over the
    first block's storage is returned - the large block is not moved.  None if the first block has no such room."""
    first = blocks[0]
    total = sum(rows)
    ref = _SLACK.get(first.untyped_storage().data_ptr()) if first.is_cuda else None
    base = ref() if ref is not None else None
    # only a storage that k_gather_rows allocated WITH spare rows - and that is still that allocation (the weak reference
    # is alive) - qualifies: any other tensor at offset 0 of a larger storage may have live data behind it
    if not (base is not None and first.is_contiguous() and first.dim() == 2 and first.storage_offset() == 0
            and base.dtype == first.dtype and base.shape[1] == cols and first.shape[0] == rows[0] and base.shape[0] >= total):
        return None
    full = base[:total]
    r0 = rows[0]
    for blk, n in zip(blocks[1:], rows[1:]):
        full[r0:r0 + n].copy_(blk.reshape(n, cols))
        r0 += n
    return full


class StackRowsFn(torch.autograd.Function):
    """cat of 2-D row blocks; the backward hands out views of dy.  Round 5: when the first block has spare rows behind it
    (the subtitle embeddings: hero_gather_rows allocates the 480 query rows with them, model/encoder.py) only the
    remaining blocks are copied - 0.7 MB instead of an 18 MB cat per micro-step."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.rows = [x.shape[0] for x in xs]
        full = _stack_in_place(xs, ctx.rows, xs[0].shape[1]) if all(x.dtype == xs[0].dtype for x in xs) else None
        return full if full is not None else torch.cat(xs, 0)

    @staticmethod
    def backward(ctx, dy):
        out, r0 = [], 0
        for i, n in enumerate(ctx.rows):
            out.append(dy[r0:r0 + n] if ctx.needs_input_grad[i] else None)
            r0 += n
        return tuple(out)


class SplitRowsFn(torch.autograd.Function):
    """x -> row blocks of the given sizes (views).  backward = ONE stacked gradient; aten's slices would each materialise
    a zero tensor of the full shape, copy their block in and add the results (5 passes over the stacked rows instead of
    1).  The first output carries `_hero_tail_rows` = the rows of the other blocks: a consumer whose backward produces the
    first block's gradient with hero_gather_rows (CsrGatherSumFn) allocates it with that many spare rows, and this
    backward then copies only the small blocks in (round 5) instead of a cat of everything."""

    @staticmethod
    def forward(ctx, x, *sizes):
        ctx.sizes, ctx.meta = sizes, (x.shape[1], x.dtype, x.device)
        outs, r0 = [], 0
        for n in sizes:
            outs.append(x[r0:r0 + n])
            r0 += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        cols, dtype, dev = ctx.meta
        parts = [g.reshape(n, cols) if g is not None else torch.zeros((n, cols), dtype=dtype, device=dev)
                 for g, n in zip(grads, ctx.sizes)]
        full = _stack_in_place(parts, list(ctx.sizes), cols) if all(p_.dtype == dtype for p_ in parts) else None
        return (full if full is not None else torch.cat(parts, 0),) + (None,) * len(ctx.sizes)


class _FirstRefCheck:
    """Outcome of the run-time precondition check of GatherRowsFn's one-gather backward: did every VALID output position
    turn out to be the first reference to its source row?  The device computes one flag when the first-occurrence map is
    built (once per index tensor), a non-blocking copy brings it to pinned host memory, and the BACKWARD reads it - several
    milliseconds of queued work later, so the wait is already over and no forward ever stalls on it."""
    __slots__ = ("flag", "event", "value")

    def __init__(self, flag, event):
        self.flag, self.event, self.value = flag, event, None

    def ok(self):
        if self.value is None:
            self.event.synchronize()
            self.value = int(self.flag[0]) == 0
        return self.value


def first_reference_map(idx, valid, na, nb):
    """(inv, check) for GatherRowsFn's backward: inv = hero_inverse_first(idx) - source row -> its first referencing
    position - and check.ok() == "no valid position (valid != 0) is a LATER reference to its source row", which is what
    makes `d source = d out[inv]` the exact adjoint of the gather when masked positions receive no gradient.  HERO's own
    f_gather_index (data/data.py:504-512) always passes; a caller-built index that repeats a source at a valid position does
    not, and the backward then falls back to the scatter-add.  Both are memoised per index tensor; a memo refreshed IN PLACE
    (functional.refresh_memo: the feeder's static buffers, whose index hero_amd.collate.DeviceCollate builds in the
    reference's layout) keeps the verdict of the batch it was built for.  None where it cannot be decided (too many rows
    for the map's LDS table, no mask, shapes that do not line up, or a stream that is being captured)."""
    n = idx.numel()
    if (valid is None or valid.numel() != n or na + nb > 38400 or idx.dtype != torch.int32 or not idx.is_contiguous()
            or not idx.is_cuda):
        return None

    def build():
        inv = torch.empty(na + nb, dtype=torch.int32, device=idx.device)
        L.check(L.lib().hero_inverse_first(L.ptr(idx), n, L.ptr(inv), na, nb, L.stream()))
        return inv
    inv = memo("gather_inverse", (idx, valid), build, (na, nb))
    chk = getattr(inv, "_hero_first_check", None)
    if chk is None:
        if torch.cuda.is_current_stream_capturing():       # the flag has to reach the host: not inside a capture
            return None
        i64 = idx.to(torch.int64)
        src = torch.where(i64 >= 0, i64, na - i64 - 2).clamp_(0, na + nb - 1)
        pos = torch.arange(n, device=idx.device, dtype=torch.int32)
        later = (inv[src] != pos) & (idx != -1) & (valid.reshape(-1) != 0)
        flag = torch.empty(1, dtype=torch.int32).pin_memory()
        flag.copy_(later.any().to(torch.int32).reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        chk = inv._hero_first_check = _FirstRefCheck(flag, ev)
    return inv, chk


class GatherRowsFn(torch.autograd.Function):
    """out[r] = a[idx[r]] (idx>=0) | 0 (idx==-1) | b[-idx-2]."""

    @staticmethod
    def forward(ctx, a, b, idx, tail_rows=0, valid=None):
        """valid: 0/1 mask over the output positions (HERO: f_attn_masks).  With it the caller states that positions whose bit
        is 0 receive no gradient; IF, in addition, every valid position is the first reference to its source row - checked on
        the device, `first_reference_map` - the backward is one gather through the first-occurrence map instead of two zero
        fills and a scatter-add.  Without it, or when the check fails, the backward is the scatter-add."""
        a2, b2 = _as2d(a), (_as2d(b) if b is not None else None)
        ctx.save_for_backward(idx)
        ctx.ashape, ctx.bshape = a.shape, (b.shape if b is not None else None)
        na, nb = a2.shape[0], (b2.shape[0] if b2 is not None else 0)
        ctx.first = first_reference_map(idx, valid, na, nb) if valid is not None else None
        return k_gather_rows(a2, b2, idx, idx.numel(), a2.shape[1], tail_rows=tail_rows)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy2 = _as2d(dy)
        if ctx.first is not None and ctx.first[1].ok():
            inv = ctx.first[0]
            na = inv.numel() if ctx.bshape is None else ctx.ashape.numel() // ctx.ashape[-1]
            both = k_gather_rows(dy2.contiguous(), None, inv, inv.numel(), dy2.shape[1])
            return both[:na].view(ctx.ashape), (both[na:].view(ctx.bshape) if ctx.bshape is not None else None), None, None, None
        da = torch.zeros(ctx.ashape, dtype=dy.dtype, device=dy.device)
        db = torch.zeros(ctx.bshape, dtype=dy.dtype, device=dy.device) if ctx.bshape is not None else None
        k_scatter_add(dy2, idx, da, db)
        return da, db, None, None, None


class PermuteRowsFn(torch.autograd.Function):
    """out[r] = a[idx[r]] (idx >= 0) | 0 (idx == -1) for an INJECTIVE idx; `inv` is the reverse map
    (source row -> output row or -1), so the backward is a gather too (no atomics)."""

    @staticmethod
    def forward(ctx, a, idx, inv):
        a2 = _as2d(a)
        ctx.save_for_backward(inv)
        ctx.ashape = a.shape
        return k_gather_rows(a2, None, idx, idx.numel(), a2.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (inv,) = ctx.saved_tensors
        dy2 = _as2d(dy)
        return k_gather_rows(dy2, None, inv, inv.numel(), dy2.shape[1]).view(ctx.ashape), None, None


class CsrGatherSumFn(torch.autograd.Function):
    """out[r] = sum_e src[entries[e]], e in [offsets[r], offsets[r+1]); inverse = src row -> out row
    (each source row feeds at most one output row: collect_frame_outputs, model/model.py:156-187)."""

    @staticmethod
    def forward(ctx, src, offsets, entries, inverse, n_out):
        s2 = _as2d(src)
        out = torch.empty((n_out, s2.shape[1]), dtype=s2.dtype, device=s2.device)
        L.check(L.lib().hero_csr_gather_sum(L.ptr(s2), L.ptr(offsets), L.ptr(entries), L.ptr(out),
                                            n_out, s2.shape[1], L.dt(s2), L.stream()))
        ctx.save_for_backward(inverse)
        ctx.sshape = src.shape
        ctx.tail = int(getattr(src, "_hero_tail_rows", 0))      # SplitRowsFn: rows stacked behind this block
        return out

    @staticmethod
    def backward(ctx, dy):
        (inverse,) = ctx.saved_tensors
        dy2 = _as2d(dy)
        ds = k_gather_rows(dy2, None, inverse, inverse.numel(), dy2.shape[1], tail_rows=ctx.tail)
        return ds.view(ctx.sshape), None, None, None, None


class ExpandRowsFn(torch.autograd.Function):
    """out[r] = x[idx[r]] for an index that names rows SEVERAL times (the videos of a VSM batch with query_per_video queries each,
    data/vsm.py:105-145: pair r = (query r, video q_vidx[r])); backward = for every source row the sum of its pairs' gradients in
    pair order (hero_csr_gather_sum over the index sorted by source: no atomics, the same bits every run).  x: [n, ...]."""

    @staticmethod
    def forward(ctx, x, idx):
        x2 = x.reshape(x.shape[0], -1).contiguous()
        n = x2.shape[0]

        def i32():
            return idx.reshape(-1).to(torch.int32).contiguous()

        def csr():                                    # on the device, no host read: also rebuilt inside a feeder's commit graph
            srt, order = torch.sort(idx.reshape(-1), stable=True)
            off = torch.searchsorted(srt, torch.arange(n + 1, device=idx.device, dtype=srt.dtype))    # rows below each source
            return torch.cat([off.to(torch.int32), order.to(torch.int32)])
        i = memo("expand_idx", (idx,), i32, spec=(L.DERIVE_I32, 0, 0, 0))
        ctx.csr = memo("expand_csr", (idx,), csr, (n,))
        ctx.meta = (x.shape, n)
        out = k_gather_rows(x2, None, i, i.numel(), x2.shape[1])
        return out.view((i.numel(),) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, dy):
        shape, n = ctx.meta
        d2 = dy.reshape(dy.shape[0], -1).contiguous()
        out = torch.empty((n, d2.shape[1]), dtype=d2.dtype, device=d2.device)
        L.check(L.lib().hero_csr_gather_sum(L.ptr(d2), L.ptr(ctx.csr[:n + 1]), L.ptr(ctx.csr[n + 1:]), L.ptr(out), n, d2.shape[1],
                                            L.dt(d2), L.stream()))
        return out.view(shape), None


# ---- transformer blocks ------------------------------------------------------------------------
# Parameters of the blocks are always leaf nn.Parameters; their gradients go to the gradient sink
# (accumulated in place by the kernels) and the functions return None for them.
def _qkv_bwd(dqkv, x2, qkv_params, D):
    """Parameter gradients of the fused QKV projection.  When the gradient sink keeps the three
    weights (and the three biases) back to back, d[Wq;Wk;Wv] is ONE [3D, D] GEMM and d[bq;bk;bv] one
    column sum; otherwise three of each."""
    wq, bq, wk, bk, wv, bv = qkv_params
    ws, bs = (wq, wk, wv), (bq, bk, bv)
    gw = SINK.dst_group(ws) if all(p.requires_grad for p in ws) else None
    gb = SINK.dst_group(bs) if all(p.requires_grad for p in bs) else None
    if gw is not None:
        k_wgrad(dqkv, x2, out=gw, beta=1.0, on_done=lambda: [SINK.done(p) for p in ws],
                dbias=gb.view(-1) if gb is not None else None, dbias_done=lambda: [SINK.done(p) for p in bs])
    elif gb is not None:
        k_colsum(dqkv, out=gb.view(-1), beta=1.0, on_done=lambda: [SINK.done(p) for p in bs])
    for i, (w, b) in enumerate(zip(ws, bs)):
        acc_linear_grads(dqkv, x2, None if gw is not None else w, None if gb is not None else b, col0=i * D,
                         ncols=D)


class SelfAttentionFn(torch.autograd.Function):
    """BertSelfAttention (model/layers.py:124-164): ctx = MHA(x)."""

    @staticmethod
    def forward(ctx, x, mask_add, H, drop_p, wq, bq, wk, bk, wv, bv):
        S, Lq, D = x.shape
        x2 = _as2d(x)
        Wqkv = packed((wq, wk, wv), x2.dtype)
        bqkv = packed((bq, bk, bv), torch.float32)
        qkv = k_linear(x2, Wqkv, bqkv)
        ctxt, probs = k_attn_fwd(qkv, mask_add, S, Lq, H, drop=drop_p)
        ctx.dims, ctx.drop, ctx.mask_add = (S, Lq, H, D), drop_p, mask_add
        ctx.params = (wq, bq, wk, bk, wv, bv)
        _use(*ctx.params)
        ctx.save_for_backward(x2, packed_t((wq, wk, wv), x2.dtype), qkv, probs, ctxt)
        return ctxt.view(S, Lq, D)

    @staticmethod
    def backward(ctx, dctx):
        x2, Wqkv_t, qkv, probs, ctxt = ctx.saved_tensors
        S, Lq, H, D = ctx.dims
        dqkv = k_attn_bwd(qkv, probs, _as2d(dctx), S, Lq, H, drop=ctx.drop, ctx=ctxt, mask_add=ctx.mask_add)
        dx = k_dgrad_t(dqkv, Wqkv_t).view(S, Lq, D) if ctx.needs_input_grad[0] else None
        _qkv_bwd(dqkv, x2, ctx.params, D)
        return (dx,) + (None,) * 9


class ProjResLnFn(torch.autograd.Function):
    """BertSelfOutput / BertOutput (model/layers.py:175-179, 250-254): LN(drop(h W^T + b) + res)."""

    @staticmethod
    def forward(ctx, h, res, eps, drop, w, b, gamma, beta):
        h2, r2 = _as2d(h), _as2d(res)
        Wc = packed((w,), h2.dtype)
        y = k_linear(h2, Wc, b.detach(), residual=r2, drop=drop)
        out, mean, rstd, _ = k_ln_fwd(y, gamma.detach(), beta.detach(), eps, y.dtype, y.shape[0], y.shape[1])
        ctx.drop = drop
        ctx.hshape, ctx.rshape = h.shape, res.shape
        ctx.params = (w, b, gamma, beta)
        _use(*ctx.params)
        ctx.save_for_backward(h2, packed_t((w,), h2.dtype), y, mean, rstd, gamma.detach())
        return out.view(res.shape)

    @staticmethod
    def backward(ctx, dout):
        h2, Wt, y, mean, rstd, gamma = ctx.saved_tensors
        w, b, gp, bp = ctx.params
        dgd, dbd = ln_param_dsts(gp, bp)
        fuse_b = b.requires_grad and y.shape[1] <= 1024
        dy, dyd, _, _ = k_ln_bwd(y, _as2d(dout), gamma, mean, rstd, drop_in=ctx.drop, dgamma=dgd,
                                 dbeta=dbd, grad_beta=1.0, want_params=False,
                                 dbias_in=SINK.dst(b) if fuse_b else None)
        ln_params_done(gp, bp)
        if fuse_b:
            SINK.done(b)
        acc_linear_grads(dyd, h2, w, None if fuse_b else b)
        dh = k_dgrad_t(dyd, Wt).view(ctx.hshape)
        return (dh, dy.view(ctx.rshape)) + (None,) * 6


ONE_ATTN_LAUNCH = [True]    # False: one attention launch per sequence group (lab A/B)
_SEQ_OFF = {}


def _one_attention_launch(segs, masks, drops, x2):
    """Several padded sequence groups stacked along the rows (HERO's subtitle rows, 480 x 24, and its query rows, 32 x 15)
    as ONE variable-length attention launch: `seq_off` lists every sequence of every group (rows back to back, nothing is
    moved), the additive masks are stacked into one [sum S, max L] tensor (HeroAttn.mask is indexed [s, max L] with a
    seq_off too).  The query group's launch - 384 waves, ~5 us forward / ~7 us backward plus two kernel boundaries per layer -
    disappears into the subtitle one.  bf16 matrix-core kernels with row statistics only (L <= 64)."""
    if not (ONE_ATTN_LAUNCH[0] and len(segs) > 1 and all(len(sg) == 2 for sg in segs) and x2.dtype == torch.bfloat16 and x2.is_cuda):
        return segs, masks, drops
    lmax = max(sg[1] for sg in segs)
    if ATTN_SAVE_PROBS or not L.lib().hero_attention_stats_ok(L.BF16, lmax) or any(m is not None and m.requires_grad for m in masks):
        return segs, masks, drops
    key = (tuple(segs), x2.device.index)
    off = _SEQ_OFF.get(key)
    if off is None:
        o, r = [0], 0
        for S_, L_ in segs:
            for _ in range(S_):
                r += L_
                o.append(r)
        off = _SEQ_OFF[key] = torch.tensor(o, dtype=torch.int32, device=x2.device)
    if all(m is None for m in masks):
        mcat = None
    else:
        def build():
            parts = []
            for (S_, L_), m in zip(segs, masks):
                mm = m.reshape(S_, L_).float() if m is not None else torch.zeros((S_, L_), dtype=torch.float32, device=x2.device)
                parts.append(torch.nn.functional.pad(mm, (0, lmax - L_)) if L_ < lmax else mm)
            return torch.cat(parts, 0).contiguous()
        mcat = memo("attn_mask_cat", tuple(m for m in masks if m is not None), build, (tuple(segs),))
    return (("packed", sum(sg[0] for sg in segs), lmax, off),), (mcat,), (drops[0],)


class AttnBlockFn(torch.autograd.Function):
    """BertAttention (model/layers.py:217-222) as ONE node: a = LN(drop(MHA(x) Wo^T + bo) + x).

    `x` is [rows, D]; `segs` = ((S, L), ...) lists the sequence groups stacked along the rows
    (sum S*L == rows).  Everything except the attention itself is row-wise, so several groups with
    different lengths — e.g. HERO's subtitle rows (L = frames+tokens) and its query rows — share the
    GEMM / LayerNorm launches and only the attention kernel is called per group."""

    @staticmethod
    def forward(ctx, x, segs, masks, H, eps, drops_attn, drop_hid, wq, bq, wk, bk, wv, bv, wo, bo, g1, b1):
        x2 = _as2d(x)
        D = x2.shape[1]
        Wqkv = packed((wq, wk, wv), x2.dtype)
        bqkv = packed((bq, bk, bv), torch.float32)
        Wo = packed((wo,), x2.dtype)
        qkv = k_linear(x2, Wqkv, bqkv)
        ctxt = torch.empty((x2.shape[0], D), dtype=x2.dtype, device=x2.device)
        segs, masks, drops_attn = _one_attention_launch(segs, masks, drops_attn, x2)
        probs = []
        r0 = 0
        for seg, m, dr in zip(segs, masks, drops_attn):          # each group writes its row slice
            if len(seg) == 4:                                    # ("packed", S, Lmax, seq_off): all rows
                _, S, Lq, off = seg
                _, p = k_attn_fwd(qkv, m, S, Lq, H, drop=dr, out=ctxt, seq_off=off)
                probs.append(p)
                continue
            S, Lq = seg
            _, p = k_attn_fwd(qkv[r0:r0 + S * Lq], m, S, Lq, H, drop=dr, out=ctxt[r0:r0 + S * Lq])
            probs.append(p)
            r0 += S * Lq
        y1 = k_linear(ctxt, Wo, bo.detach(), residual=x2, drop=drop_hid)
        a, mean, rstd, _ = k_ln_fwd(y1, g1.detach(), b1.detach(), eps, y1.dtype, x2.shape[0], D)
        ctx.meta = (segs, H, D, drops_attn, drop_hid, x.shape)
        ctx.masks = masks                 # additive masks (derived, no grad): the backward rebuilds P from q, k with them
        ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo, g1, b1)
        _use(*ctx.params)
        ctx.save_for_backward(x2, packed_t((wq, wk, wv), x2.dtype), packed_t((wo,), x2.dtype), qkv, ctxt, y1,
                              mean, rstd, g1.detach(), *probs)
        return a.view(x.shape)

    @staticmethod
    def backward(ctx, da):
        x2, Wqkv_t, Wo_t, qkv, ctxt, y1, mean, rstd, g1 = ctx.saved_tensors[:9]
        probs = ctx.saved_tensors[9:]
        segs, H, D, drops_attn, drop_hid, xshape = ctx.meta
        wq, bq, wk, bk, wv, bv, wo, bo, g1p, b1p = ctx.params
        dgd, dbd = ln_param_dsts(g1p, b1p)
        fuse_b = bo.requires_grad and D <= 1024
        dy1, dy1d, _, _ = k_ln_bwd(y1, _as2d(da), g1, mean, rstd, drop_in=drop_hid, dgamma=dgd,
                                   dbeta=dbd, grad_beta=1.0, want_params=False,
                                   dbias_in=SINK.dst(bo) if fuse_b else None)
        ln_params_done(g1p, b1p)
        if fuse_b:
            SINK.done(bo)
        acc_linear_grads(dy1d, ctxt, wo, None if fuse_b else bo)
        dctx = k_dgrad_t(dy1d, Wo_t)
        dqkv = torch.empty_like(qkv)
        r0 = 0
        for seg, p, dr, m in zip(segs, probs, drops_attn, ctx.masks):
            if len(seg) == 4:
                _, S, Lq, off = seg
                k_attn_bwd(qkv, p, dctx, S, Lq, H, drop=dr, out=dqkv, seq_off=off, ctx=ctxt, mask_add=m)
                continue
            S, Lq = seg
            k_attn_bwd(qkv[r0:r0 + S * Lq], p, dctx[r0:r0 + S * Lq], S, Lq, H, drop=dr,
                       out=dqkv[r0:r0 + S * Lq], ctx=ctxt[r0:r0 + S * Lq], mask_add=m)
            r0 += S * Lq
        _qkv_bwd(dqkv, x2, (wq, bq, wk, bk, wv, bv), D)
        dx = k_dgrad_t(dqkv, Wqkv_t, residual=dy1).view(xshape)    # + residual-path gradient, fused
        return (dx,) + (None,) * 16


class FfnBlockFn(torch.autograd.Function):
    """BertIntermediate + BertOutput (model/layers.py:236-254) as ONE node:
    out = LN(drop(gelu(a W1^T + b1) W2^T + b2) + a)."""

    @staticmethod
    def forward(ctx, a, eps, drop_hid, w1, b1, w2, b2, g2, bt2):
        shp = a.shape
        a2 = _as2d(a)
        W1, W2 = packed((w1,), a2.dtype), packed((w2,), a2.dtype)
        # `u` receives gelu'(a W1^T + b1), not the pre-activation (round 4, HERO_ACT_GELU_DG): nothing else reads it, both
        # come out of one evaluation of Phi, and the backward epilogue becomes a multiply (HERO_ACT_MUL_AUX: ~7 us of exp /
        # rcp / polynomial per 12000 x 3072 launch, tools/lab/gelu_ab.py)
        u = torch.empty((a2.shape[0], W1.shape[0]), dtype=a2.dtype, device=a2.device)
        hg = k_linear(a2, W1, b1.detach(), act=L.ACT_GELU_DG, aux=u)
        y2 = k_linear(hg, W2, b2.detach(), residual=a2, drop=drop_hid)
        out, mean, rstd, _ = k_ln_fwd(y2, g2.detach(), bt2.detach(), eps, y2.dtype, y2.shape[0], y2.shape[1])
        ctx.drop, ctx.shp = drop_hid, shp
        ctx.params = (w1, b1, w2, b2, g2, bt2)
        _use(*ctx.params)
        ctx.save_for_backward(a2, packed_t((w1,), a2.dtype), packed_t((w2,), a2.dtype), u, hg, y2, mean,
                              rstd, g2.detach())
        return out.view(shp)

    @staticmethod
    def backward(ctx, dout):
        a2, W1_t, W2_t, u, hg, y2, mean, rstd, g2 = ctx.saved_tensors
        w1, b1, w2, b2, g2p, bt2p = ctx.params
        dgd, dbd = ln_param_dsts(g2p, bt2p)
        fuse_b = b2.requires_grad and y2.shape[1] <= 1024
        dy2, dy2d, _, _ = k_ln_bwd(y2, _as2d(dout), g2, mean, rstd, drop_in=ctx.drop, dgamma=dgd,
                                   dbeta=dbd, grad_beta=1.0, want_params=False,
                                   dbias_in=SINK.dst(b2) if fuse_b else None)
        ln_params_done(g2p, bt2p)
        if fuse_b:
            SINK.done(b2)
        acc_linear_grads(dy2d, hg, w2, None if fuse_b else b2)
        # db1 = column sums of du.  Rounds 1-3: fp32 atomics from this GEMM's gelu' epilogue (order-dependent).  Round 4: they
        # rode on the batched weight-gradient launch (deterministic, but 3072 more bias columns through hero_wgrad_batch's
        # loader waves cost that launch +0.09 ms per micro-step: tools/lab/ab_wgrad.py, profiles/r05_wgrad_ab.txt).  Round 5:
        # the epilogue writes each tile's column sums to a [rows / 64, 3072] table (HeroGemmEpilogue.colsum_partial: plain
        # stores, no atomics) and the table is folded in fixed order - by the deferred multi-sum of the backward pass, or at
        # once where gradients must become final layer by layer (boundary micro-steps of a data-parallel run).
        part = None
        if b1.requires_grad and B1_PARTIALS[0] and dy2d.dtype == torch.bfloat16:
            part = torch.empty((-(-dy2d.shape[0] // 64), W2_t.shape[0]), dtype=torch.float32, device=dy2d.device)
        fuse_b1 = part is None and b1.requires_grad and (SINK.wants_overlap() or not GROUP_WGRADS[0] or B1_EPILOGUE[0])
        du = k_dgrad_t(dy2d, W2_t, act=L.ACT_MUL_AUX, aux=u,    # * gelu'(pre-activation) = the saved tensor, fused
                       colsum=part if part is not None else (SINK.dst(b1) if fuse_b1 else None), colsum_partial=part is not None)
        if part is not None:
            k_colsum(part, out=SINK.dst(b1), beta=1.0, on_done=lambda: SINK.done(b1))
        acc_linear_grads(du, a2, w1, None if (fuse_b1 or part is not None) else b1)
        if fuse_b1:
            SINK.done(b1)
        da = k_dgrad_t(du, W1_t, residual=dy2).view(ctx.shp)        # + residual-path gradient, fused
        return (da,) + (None,) * 8
