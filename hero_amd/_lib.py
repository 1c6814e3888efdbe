"""ctypes binding of libhero_hip.so (C ABI declared in include/hero_hip.h).

The product path has no CPU or PyTorch-eager fallback: if the shared library is missing, or a
tensor is not a contiguous CUDA tensor, the call raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HERO_HIP_LIB") or os.path.join(_HERE, "libhero_hip.so")   # HERO_HIP_LIB: alternative build (A/B runs)

ABI_VERSION = 3          # include/hero_hip.h HERO_ABI_VERSION this binding's struct layouts were written against
F32, BF16 = 0, 1
LAYOUT_K, LAYOUT_O = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_BWD, ACT_RELU_BWD, ACT_GELU_DG, ACT_MUL_AUX = 0, 1, 2, 3, 4, 5, 6

EXPORTS = [
    "hero_last_error", "hero_abi_version", "hero_abi_struct_count", "hero_abi_struct_bytes", "hero_gemm", "hero_gemm_splits", "hero_fold_slabs", "hero_wgrad_group", "hero_wgrad_batch_plan", "hero_wgrad_batch", "hero_prof_enable", "hero_prof_read", "hero_probe_mfma", "hero_probe_hbm", "hero_gemm_force_config", "hero_layernorm_fwd",
    "hero_layernorm_bwd_workspace_bytes", "hero_layernorm_bwd", "hero_colsum_workspace_bytes",
    "hero_colsum", "hero_colsum_multi", "hero_colsum_multi_workspace_bytes", "hero_layernorm_bwd_blocks", "hero_attention_fwd", "hero_attention_bwd", "hero_attention_max_len", "hero_attention_max_packed_len", "hero_attention_stats_ok", "hero_attention_force_ppw",
    "hero_gather_rows", "hero_inverse_first", "hero_csr_gather_sum", "hero_scatter_add_rows", "hero_segment_sort_workspace_bytes", "hero_scatter_add_sorted_workspace_bytes", "hero_segment_sort", "hero_scatter_add_sorted", "hero_cast", "hero_transpose_cast", "hero_copy_multi",
    "hero_relu_bwd", "hero_gelu_bwd", "hero_add", "hero_sumsq", "hero_adamw", "hero_adamw_multi", "hero_adamw_multi_chunk",
    "hero_query_pool_fwd", "hero_query_pool_bwd", "hero_rownorm_fwd", "hero_rownorm_bwd", "hero_score_max_fwd",
    "hero_score_max_bwd", "hero_rank_loss", "hero_sums_scaled", "hero_st_ed_fwd", "hero_st_ed_bwd", "hero_st_ed_bwd_workspace_bytes",
    "hero_cross_entropy_fwd", "hero_cross_entropy_bwd",
    "hero_collate_subs", "hero_collate_clip_mask", "hero_collate_frame_map", "hero_collate_gather_feats", "hero_derive_multi",
    "hero_comm_available", "hero_comm_unique_id", "hero_comm_init", "hero_comm_destroy", "hero_comm_rank", "hero_comm_world",
    "hero_comm_allreduce_buckets", "hero_comm_broadcast", "hero_comm_allgather", "hero_comm_allgather_var",
]


class Dropout(C.Structure):
    _fields_ = [("seed_ptr", C.c_void_p), ("site", C.c_uint64),
                ("threshold16", C.c_uint32), ("scale", C.c_float)]


class GemmEpilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("residual", C.c_void_p), ("aux", C.c_void_p),
                ("act", C.c_int), ("out_f32", C.c_int), ("beta", C.c_float),
                ("split_k", C.c_int), ("dropout", Dropout), ("colsum", C.c_void_p), ("colsum_partial", C.c_int), ("pad_", C.c_int), ("split_stride", C.c_longlong)]


class CrossEntropy(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("labels", C.c_void_p), ("loss", C.c_void_p), ("lse", C.c_void_p),
                ("dloss", C.c_void_p), ("dlogits", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int),
                ("ld", C.c_int), ("dtype", C.c_int), ("inv_temp", C.c_float), ("ignore_index", C.c_int64)]


class Derive(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64), ("mode", C.c_int), ("p0", C.c_int),
                ("p1", C.c_int), ("p2", C.c_int)]


DERIVE_MASK_ADD, DERIVE_F32, DERIVE_I32, DERIVE_FLAT_GATHER = 0, 1, 2, 3


class WgradProblem(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("M", C.c_int), ("N", C.c_int),
                ("ld_dy", C.c_int), ("ld_x", C.c_int), ("ld_dw", C.c_int), ("split_hint", C.c_int), ("dbias", C.c_void_p)]


class LnFwd(C.Structure):
    _fields_ = [("x", C.c_void_p), ("tab", C.c_void_p * 3), ("idx", C.c_void_p * 3),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("y", C.c_void_p),
                ("pre", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("rows", C.c_int), ("cols", C.c_int), ("eps", C.c_float),
                ("x_dtype", C.c_int), ("y_dtype", C.c_int), ("dropout", Dropout)]


class LnBwd(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("gamma", C.c_void_p),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("dx", C.c_void_p),
                ("dx_dropped", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("grad_beta", C.c_float), ("workspace", C.c_void_p),
                ("rows", C.c_int), ("cols", C.c_int), ("x_dtype", C.c_int), ("dtype", C.c_int),
                ("dropout_out", Dropout), ("dropout_in", Dropout), ("dbias_in", C.c_void_p), ("defer_fold", C.c_int)]


class Colsum(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int),
                ("dtype", C.c_int), ("beta", C.c_float), ("row_cols", C.c_int), ("dst_rows", C.c_void_p)]


class Attn(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("mask", C.c_void_p), ("ctx", C.c_void_p),
                ("probs", C.c_void_p), ("dctx", C.c_void_p), ("dqkv", C.c_void_p),
                ("S", C.c_int), ("L", C.c_int), ("H", C.c_int), ("scale", C.c_float),
                ("dtype", C.c_int), ("dropout", Dropout), ("seq_off", C.c_void_p), ("stats", C.c_void_p)]


class AdamW(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("n", C.c_size_t), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("weight_decay", C.c_float), ("step", C.c_int),
                ("grad_sumsq", C.c_void_p), ("max_grad_norm", C.c_float),
                ("grad_scale", C.c_float), ("shadow", C.c_void_p)]


class TensorDesc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("n", C.c_uint64), ("group", C.c_int32), ("step_lag", C.c_int32)]


class AdamWGroup(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float)]


class AdamWMulti(C.Structure):
    _fields_ = [("descs", C.c_void_p), ("chunk_tensor", C.c_void_p), ("chunk_index", C.c_void_p),
                ("n_chunks", C.c_int), ("groups", AdamWGroup * 8), ("step", C.c_int),
                ("grad_sumsq", C.c_void_p), ("max_grad_norm", C.c_float), ("grad_scale", C.c_float),
                ("step_ptr", C.c_void_p), ("lr_ptr", C.c_void_p), ("tensor_steps", C.c_void_p), ("n_tensors", C.c_int)]


class CopyDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32),
                ("ldd", C.c_int32), ("transpose", C.c_int32), ("dst_dtype", C.c_int32), ("pad_", C.c_int32)]


class QueryPool(C.Structure):
    _fields_ = [("q", C.c_void_p), ("mask", C.c_void_p), ("w", C.c_void_p), ("pooled", C.c_void_p),
                ("att", C.c_void_p), ("dpooled", C.c_void_p), ("dq", C.c_void_p), ("dw", C.c_void_p),
                ("B", C.c_int), ("L", C.c_int), ("D", C.c_int), ("dtype", C.c_int)]


class RowNorm(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("rnorm", C.c_void_p), ("dy", C.c_void_p),
                ("dx", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("x_dtype", C.c_int),
                ("eps", C.c_float)]


class ScoreMax(C.Structure):
    _fields_ = [("s", C.c_void_p), ("mask", C.c_void_p), ("out", C.c_void_p), ("arg", C.c_void_p),
                ("ds_ctx", C.c_void_p), ("ds_q", C.c_void_p), ("gc", C.c_void_p), ("gq", C.c_void_p),
                ("qn", C.c_void_p), ("cn", C.c_void_p), ("dqn", C.c_void_p), ("dcn", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("L", C.c_int), ("D", C.c_int), ("n0", C.c_int),
                ("n_own", C.c_int), ("ld_s", C.c_int), ("gc_scale", C.c_float), ("gq_scale", C.c_float)]


class RankLoss(C.Structure):
    _fields_ = [("s", C.c_void_p), ("loss_ctx_rows", C.c_void_p), ("loss_q_rows", C.c_void_p),
                ("ds_ctx", C.c_void_p), ("ds_q", C.c_void_p), ("nq", C.c_int), ("nv", C.c_int),
                ("margin", C.c_float), ("lse", C.c_int), ("hard", C.c_int), ("pool", C.c_int),
                ("hard_w", C.c_float), ("easy_w", C.c_float)]


class StEd(C.Structure):
    _fields_ = [("q2", C.c_void_p), ("ctx", C.c_void_p), ("mask", C.c_void_p), ("w_st", C.c_void_p),
                ("w_ed", C.c_void_p), ("targets", C.c_void_p), ("loss_rows", C.c_void_p),
                ("p_st", C.c_void_p), ("p_ed", C.c_void_p), ("sim", C.c_void_p), ("g", C.c_void_p),
                ("dq2", C.c_void_p), ("dctx", C.c_void_p), ("dw_st", C.c_void_p), ("dw_ed", C.c_void_p),
                ("B", C.c_int), ("L", C.c_int), ("D", C.c_int), ("K", C.c_int), ("dtype", C.c_int), ("g_scale", C.c_float),
                ("ws", C.c_void_p)]


class CommBucket(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("count", C.c_size_t), ("dtype", C.c_int), ("pad_", C.c_int)]


# HERO_STRUCT_* id (include/hero_hip.h) -> the ctypes mirror of that struct; lib() holds every sizeof against the library's
ABI_STRUCTS = [Dropout, GemmEpilogue, WgradProblem, LnFwd, LnBwd, Colsum, Attn, AdamW, TensorDesc, AdamWGroup, AdamWMulti,
               CopyDesc, QueryPool, RowNorm, ScoreMax, RankLoss, StEd, CrossEntropy, Derive, CommBucket]


def abi_struct_sizes():
    """{struct name: ctypes.sizeof} of this binding, in HERO_STRUCT_* order (tests/golden/abi_sizes.json pins it per version)."""
    return {st.__name__: C.sizeof(st) for st in ABI_STRUCTS}


_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "hero_amd: %s is missing. Build it with `python -m hero_amd.build` "
                "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no fallback "
                "path." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.hero_last_error.restype = C.c_char_p
        have = L.hero_abi_version()
        if have != ABI_VERSION:
            raise RuntimeError("hero_amd: %s reports ABI version %d, this binding was written against %d (struct layouts "
                               "differ: rebuild with `python -m hero_amd.build --force`)" % (LIB_PATH, have, ABI_VERSION))
        L.hero_abi_struct_bytes.argtypes = [C.c_int]
        if L.hero_abi_struct_count() != len(ABI_STRUCTS):
            raise RuntimeError("hero_amd: %s declares %d ABI structs, this binding mirrors %d" % (LIB_PATH, L.hero_abi_struct_count(), len(ABI_STRUCTS)))
        bad = [(st.__name__, C.sizeof(st), L.hero_abi_struct_bytes(i)) for i, st in enumerate(ABI_STRUCTS)
               if C.sizeof(st) != L.hero_abi_struct_bytes(i)]
        if bad:
            raise RuntimeError("hero_amd: struct layouts differ from %s (name, binding bytes, library bytes): %s" % (LIB_PATH, bad))
        L.hero_layernorm_bwd_workspace_bytes.restype = C.c_size_t
        L.hero_colsum_workspace_bytes.restype = C.c_size_t
        L.hero_colsum_multi_workspace_bytes.restype = C.c_size_t
        L.hero_colsum_multi_workspace_bytes.argtypes = [C.POINTER(Colsum), C.c_int]
        L.hero_colsum_multi.argtypes = [C.POINTER(Colsum), C.c_int, C.c_void_p, C.c_void_p]
        L.hero_layernorm_bwd_blocks.argtypes = [C.c_int]
        L.hero_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 9 + [
            C.POINTER(GemmEpilogue), C.c_void_p]
        L.hero_gemm_splits.argtypes = [C.c_int, C.c_int, C.c_int]
        L.hero_fold_slabs.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.hero_wgrad_group.argtypes = [C.POINTER(WgradProblem), C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.hero_wgrad_batch_plan.argtypes = [C.POINTER(WgradProblem), C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.hero_wgrad_batch.argtypes = [C.POINTER(WgradProblem), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.hero_prof_enable.argtypes = [C.c_int]
        L.hero_prof_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_longlong)]
        L.hero_probe_mfma.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]
        L.hero_probe_hbm.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]
        L.hero_layernorm_fwd.argtypes = [C.POINTER(LnFwd), C.c_void_p]
        L.hero_layernorm_bwd.argtypes = [C.POINTER(LnBwd), C.c_void_p]
        L.hero_layernorm_bwd_workspace_bytes.argtypes = [C.c_int, C.c_int]
        L.hero_colsum_workspace_bytes.argtypes = [C.c_int, C.c_int]
        L.hero_colsum.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_float, C.c_void_p, C.c_void_p]
        L.hero_attention_fwd.argtypes = [C.POINTER(Attn), C.c_void_p]
        L.hero_attention_bwd.argtypes = [C.POINTER(Attn), C.c_void_p]
        L.hero_attention_stats_ok.argtypes = [C.c_int, C.c_int]
        L.hero_attention_force_ppw.argtypes = [C.c_int]
        L.hero_attention_max_len.argtypes = [C.c_int, C.c_int]
        L.hero_attention_max_packed_len.argtypes = [C.c_int]
        L.hero_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.hero_inverse_first.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.hero_csr_gather_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.hero_scatter_add_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.hero_segment_sort_workspace_bytes.argtypes = [C.c_int]
        L.hero_segment_sort_workspace_bytes.restype = C.c_size_t
        L.hero_segment_sort.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hero_scatter_add_sorted_workspace_bytes.argtypes = [C.c_int, C.c_int]
        L.hero_scatter_add_sorted_workspace_bytes.restype = C.c_size_t
        L.hero_scatter_add_sorted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p]
        L.hero_st_ed_bwd_workspace_bytes.argtypes = [C.c_int]
        L.hero_st_ed_bwd_workspace_bytes.restype = C.c_size_t
        L.hero_cast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.hero_transpose_cast.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]
        L.hero_relu_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                    C.c_void_p]
        L.hero_gelu_bwd.argtypes = L.hero_relu_bwd.argtypes
        L.hero_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.hero_sumsq.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hero_adamw.argtypes = [C.POINTER(AdamW), C.c_void_p]
        L.hero_adamw_multi.argtypes = [C.POINTER(AdamWMulti), C.c_void_p]
        L.hero_copy_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        for fn, st in (("hero_query_pool_fwd", QueryPool), ("hero_query_pool_bwd", QueryPool),
                       ("hero_rownorm_fwd", RowNorm), ("hero_rownorm_bwd", RowNorm),
                       ("hero_score_max_fwd", ScoreMax), ("hero_score_max_bwd", ScoreMax),
                       ("hero_rank_loss", RankLoss), ("hero_st_ed_fwd", StEd), ("hero_st_ed_bwd", StEd)):
            getattr(L, fn).argtypes = [C.POINTER(st), C.c_void_p]
        L.hero_sums_scaled.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        L.hero_collate_subs.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]
        L.hero_collate_gather_feats.argtypes = [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p]
        L.hero_derive_multi.argtypes = [C.POINTER(Derive), C.c_int, C.c_void_p]
        L.hero_collate_clip_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.hero_collate_frame_map.argtypes = [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p]
        L.hero_cross_entropy_fwd.argtypes = [C.POINTER(CrossEntropy), C.c_void_p]
        L.hero_cross_entropy_bwd.argtypes = [C.POINTER(CrossEntropy), C.c_void_p]
        L.hero_comm_unique_id.argtypes = [C.c_void_p]
        L.hero_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.hero_comm_destroy.argtypes = [C.c_void_p]
        L.hero_comm_rank.argtypes = [C.c_void_p]
        L.hero_comm_world.argtypes = [C.c_void_p]
        L.hero_comm_allreduce_buckets.argtypes = [C.c_void_p, C.POINTER(CommBucket), C.c_int, C.c_void_p]
        L.hero_comm_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.hero_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.hero_comm_allgather_var.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libhero_hip: %s (code %d)" % (lib().hero_last_error().decode(), rc))


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("hero_amd: unsupported dtype %s (float32 / bfloat16 only)" % t.dtype)


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("hero_amd: tensor is on %s; the HIP path needs CUDA/ROCm tensors and has "
                           "no CPU fallback" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("hero_amd: non-contiguous tensor passed to a HIP kernel")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Raw hipStream_t of torch's current stream (the private fast accessors cost ~0.3 us; the
    public `torch.cuda.current_stream().cuda_stream` ~10 us, x165 launches per micro-step)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def no_dropout():
    return Dropout(None, 0, 0, 1.0)
