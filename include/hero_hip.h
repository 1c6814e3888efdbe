/* hero_hip.h — C ABI of libhero_hip.so: hand-written HIP (gfx950 / MI355X) kernels for the
 * HERO hierarchical-encoder hot path (linjieli222/HERO: model/layers.py, model/encoder.py,
 * model/embed.py, model/model.py, optim/adamw.py).
 *
 * Conventions
 *   - plain pointers + sizes, no framework types; every pointer is DEVICE memory unless noted;
 *   - every entry point returns 0 on success or a negative HERO_ERR_* code; the message is
 *     available from hero_last_error() (thread-local, valid until the next failing call);
 *   - nothing allocates, nothing synchronises: work is enqueued on `stream` (a hipStream_t);
 *     scratch memory is passed in by the caller (see the *_workspace_bytes queries);
 *   - `dtype` is the ACTIVATION element type: HERO_F32 (exact-f32 MFMA path used for the parity
 *     gate) or HERO_BF16 (bf16 storage + bf16 MFMA, fp32 accumulation). Parameters that are
 *     small (bias, LayerNorm affine, embedding tables) are always fp32; GEMM weight operands are
 *     in the activation dtype; parameter gradients are always fp32;
 *   - row-major everywhere, leading dimensions in ELEMENTS.
 *
 * Each declaration cites the reference code (file:line in linjieli222/HERO) it replaces.
 */
#ifndef HERO_HIP_H_
#define HERO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hero_stream_t; /* hipStream_t */

enum { HERO_F32 = 0, HERO_BF16 = 1 };
enum { HERO_OK = 0, HERO_ERR_ARG = -1, HERO_ERR_LAUNCH = -2, HERO_ERR_UNSUPPORTED = -3 };

const char* hero_last_error(void);
/* ABI version of the structs and entry points below; a binding compiled against another version must refuse to run
 * (hero_amd/_lib.py does).  History: 1 = rounds 1-3.  2 = INCOMPATIBLE struct changes of round 4 - HeroQueryPool.dw is
 * [B, D] and OVERWRITTEN (was [D], accumulated), HeroStEd gained the required zero-initialised `ws`, HeroColsum gained
 * row_cols / dst_rows.  3 = the round-5 struct changes that went out under "2" by mistake - HeroTensorDesc 48 -> 64 bytes
 * (shadow, shadow_dtype; it is passed as an ARRAY, so the stride changed), HeroGemmEpilogue + 16 bytes (colsum_partial,
 * split_stride), HeroScoreMax + 8 bytes (gc_scale, gq_scale), HeroStEd.pad_ became g_scale - plus round 6: the three
 * scales are plain multipliers (the "0 means 1" sentinel is gone: pass 1.0f for "no weight"), hero_abi_struct_bytes().
 * INTEGRATION.md section 2 lists the breaks per version. */
#define HERO_ABI_VERSION 3
int hero_abi_version(void);
/* sizeof() of every struct of this header as the LIBRARY was compiled, by HERO_STRUCT_* id (-1 for an unknown id): a
 * binding asserts its own layout against these at load time (hero_amd/_lib.py), so a struct that changes size without a
 * version bump is caught structurally and not by a crash.  hero_abi_struct_count() = number of ids. */
enum { HERO_STRUCT_DROPOUT = 0, HERO_STRUCT_GEMM_EPILOGUE, HERO_STRUCT_WGRAD_PROBLEM, HERO_STRUCT_LN_FWD, HERO_STRUCT_LN_BWD,
       HERO_STRUCT_COLSUM, HERO_STRUCT_ATTN, HERO_STRUCT_ADAMW, HERO_STRUCT_TENSOR_DESC, HERO_STRUCT_ADAMW_GROUP,
       HERO_STRUCT_ADAMW_MULTI, HERO_STRUCT_COPY_DESC, HERO_STRUCT_QUERY_POOL, HERO_STRUCT_ROW_NORM, HERO_STRUCT_SCORE_MAX,
       HERO_STRUCT_RANK_LOSS, HERO_STRUCT_ST_ED, HERO_STRUCT_CROSS_ENTROPY, HERO_STRUCT_DERIVE, HERO_STRUCT_COMM_BUCKET,
       HERO_STRUCT_COUNT_ };
int hero_abi_struct_count(void);
int hero_abi_struct_bytes(int which);

/* Counter-based dropout. keep(element) is a pure function of (*seed_ptr, site, element index), so
 * the backward kernels regenerate the forward mask instead of loading one. threshold16 = round(p *
 * 65536) (0 disables), scale = 1/(1-p). seed_ptr points to ONE device uint64 that the host
 * advances per step (it stays valid under hipGraph replay); site distinguishes call sites.
 * Replaces torch.nn.Dropout at model/layers.py:149,177,252,80-84 and model/embed.py:57,116,160. */
typedef struct HeroDropout {
  const uint64_t* seed_ptr;
  uint64_t site;
  uint32_t threshold16;
  float scale;
} HeroDropout;

/* ------------------------------------------------------------------------------------------ */
/* GEMM with fused epilogue — nn.Linear forward / dgrad / wgrad                                 */
/*   model/layers.py:125-127 (Q,K,V), :176 (attention out), :237 (FFN1), :251 (FFN2), :90       */
/*   (LinearLayer), model/embed.py:110 (img_linear) and their autograd transposes.              */
/* ------------------------------------------------------------------------------------------ */
enum { HERO_LAYOUT_K = 0, /* operand is [outer, reduction], reduction contiguous          */
       HERO_LAYOUT_O = 1  /* operand is [reduction, outer], outer (M or N) contiguous     */ };
enum { HERO_ACT_NONE = 0,
       HERO_ACT_GELU = 1,     /* fwd: aux <- acc+bias (pre-activation), out <- gelu_erf(.)     */
       HERO_ACT_RELU = 2,     /* fwd: out <- relu(acc+bias); aux <- that value (pre-residual) */
       HERO_ACT_GELU_BWD = 3, /* bwd: out <- acc * gelu_erf'(aux)                              */
       HERO_ACT_RELU_BWD = 4, /* bwd: out <- acc * (aux > 0)                                   */
       HERO_ACT_GELU_DG = 5,  /* fwd: out <- gelu_erf(acc+bias); aux <- gelu_erf'(acc+bias): the   */
                              /*      derivative is saved instead of the pre-activation (BertIntermediate, */
                              /*      model/layers.py:236-239: nothing else reads it), so that ...  */
       HERO_ACT_MUL_AUX = 6   /* bwd: out <- acc * aux   ... the backward epilogue is one multiply  */ };

typedef struct HeroGemmEpilogue {
  const float* bias;    /* [N] fp32 or NULL                                                  */
  const void* residual; /* [M, ldc] activation dtype or NULL; added last                     */
  void* aux;            /* [M, ldc] activation dtype; meaning depends on `act`               */
  int act;              /* HERO_ACT_*                                                        */
  int out_f32;          /* 1: C is fp32 regardless of dtype (wgrad)                          */
  float beta;           /* out_f32 only: C <- result + beta * C                              */
  int split_k;          /* >1: reduction split over blocks, fp32 atomics into C (needs       */
                        /*     out_f32, act NONE, no bias/residual/dropout; C pre-scaled)    */
  HeroDropout dropout;  /* applied after bias/act, before residual; index = m*N + n          */
  float* colsum;        /* optional [N] fp32, ACCUMULATED: colsum[n] += sum_m out[m, n] (the values  */
                        /* written to C, before rounding).  K,K operands, no split_k.  Gives the     */
                        /* bias gradient of the layer whose output gradient this GEMM produces        */
                        /* (dH = (dY W2) * gelu'(u) -> db1) without another pass over dH.             */
  int colsum_partial;   /* 0: `colsum` is [N], accumulated with fp32 atomics (order-dependent).  1: `colsum` is a   */
                        /* [ceil(M / 64), N] fp32 table that is OVERWRITTEN - every output tile writes its column     */
                        /* sums to row (first tile row / 64) and zeros to the rows of its other 64-row blocks; the     */
                        /* table's column sums (hero_colsum / hero_colsum_multi, fixed order) are the result:          */
                        /* bit-reproducible, no atomics.  Needs tile heights that are multiples of 64 (all kernels).    */
  int pad_;
  long long split_stride; /* split_k > 1 only.  0: the splits merge with fp32 atomics into C (order-dependent).      */
                        /* != 0 (elements, >= (M-1) ldc + N): split s WRITES its partial sum to the fp32 slab        */
                        /* C + s * split_stride (no pre-scale, beta ignored); the number of slabs written is the      */
                        /* return value of hero_gemm_splits(); hero_fold_slabs adds them in slab order (deterministic) */
} HeroGemmEpilogue;

/* C[M,N] = op(A)[M,K] * op(B)[K,N] (+ epilogue).
 *   a_layout K: A is [M, lda] ; O: A is [K, lda] (wgrad: dY^T)
 *   b_layout K: B is [N, ldb] (nn.Linear weight, x @ W^T) ; O: B is [K, ldb] (dgrad / wgrad)
 * Constraints: bf16: K-layout operands need K % 8 == 0, O-layout operands need their outer dim % 8
 * == 0; f32: % 4. N % 4 == 0. All base pointers 16-byte aligned, lda/ldb/ldc multiples of that
 * vector width. */
int hero_gemm(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
              int a_layout, int b_layout, int dtype, const HeroGemmEpilogue* epi, hero_stream_t stream);
/* The number of k-ranges hero_gemm actually cuts a reduction of length K into when asked for `split_k` (whole k-tiles
 * of 64 (bf16) / 32 (f32) per range, no empty range): the slab count of a split_stride launch. */
int hero_gemm_splits(int K, int split_k, int dtype);
/* out[i] = sum_{s < n_slabs} slabs[s * stride + i], s ascending (fixed order), i < n (n % 4 == 0); out_dtype F32 / BF16. */
int hero_fold_slabs(const float* slabs, int n_slabs, size_t stride, void* out, size_t n, int out_dtype, hero_stream_t stream);

/* Per-launch timing of hero_gemm with HIP events recorded on the launch stream (bench.py's
 * roofline leg; off by default, never enable inside graph capture).
 * slot = (dtype == HERO_BF16 ? 4 : 0) + a_layout * 2 + b_layout for the 4-wave kernels, 8 / 9 for the
 * wave-specialised 192x192 K,K / O,O kernels, 10 for the 128x192 K,K tiles (gemm_ws.hip). hero_prof_read synchronises. */
/* Grouped weight gradient: dw_p[M_p, N_p] (fp32) += dy_p[K, :M_p]^T x_p[K, :N_p] for 1..4 problems that reduce over
 * the same K rows - the four nn.Linear weights of a BertLayer (model/layers.py:125-127, 176, 237, 251) - in ONE
 * launch (stream-K over the (tile, k) space of the whole group, fp32 atomics into dw).  Small / unaligned / fp32
 * groups run as one hero_gemm per problem (split_hint = its split_k). */
typedef struct HeroWgradProblem {
  const void* dy;   /* [K, ld_dy] dtype, the M columns starting at this pointer */
  const void* x;    /* [K, ld_x] dtype                                          */
  float* dw;        /* [M, ld_dw] fp32, accumulated                            */
  int M, N, ld_dy, ld_x, ld_dw;
  int split_hint;
  float* dbias;     /* optional [M] fp32 (hero_wgrad_batch only; must be NULL for hero_wgrad_group): += column sums of */
                    /* dy = the bias gradient of the same nn.Linear, taken from the dY panels the kernel streams anyway */
} HeroWgradProblem;
int hero_wgrad_group(const HeroWgradProblem* probs, int n, int K, int dtype, hero_stream_t stream);
/* Batched weight gradients, whole tiles: up to HERO_WGRAD_BATCH_MAX problems over the same K rows - the weight
 * gradients of ALL BertLayers of an encoder, queued during the backward pass (model/layers.py:112-114, 173, 231, 245 x
 * num_hidden_layers) - in ONE launch.  Whole 192 x 192 output tiles per workgroup in full rounds of the chip (plain
 * fp32 read-add-write, no atomics), the remaining tiles cut into k-slices whose atomics are applied in slice order:
 * dw is bit-reproducible for a fixed problem list.  Two steps:
 *   hero_wgrad_batch_plan  (host, no device work): writes the schedule for (shapes of probs, K) into `plan`
 *                          (int32 words, host memory); returns the number of words, 0 if the group is too small
 *                          for this kernel (fewer tiles than workgroups: use hero_wgrad_group), < 0 on error /
 *                          insufficient capacity.  Depends only on M, N of the problems and K: cache it.
 *   hero_wgrad_batch       launches with the plan copied to device memory by the caller (plan_dev, plan_words).
 * bf16 only; M, N multiples of 8; pointers 16-byte aligned; one launch at a time per device (the slice-order flags
 * are library state). */
#define HERO_WGRAD_BATCH_MAX 32
int hero_wgrad_batch_plan(const HeroWgradProblem* probs, int n, int K, int32_t* plan, int capacity_words);
int hero_wgrad_batch(const HeroWgradProblem* probs, int n, int K, int dtype, const int32_t* plan_dev, int plan_words,
                     hero_stream_t stream);
int hero_gemm_force_config(int cfg); /* tuning hook, bits 0-1 tile geometry: 0 128x128, 1 192x128, 2 256x256,
                                      * 3 64x64 (1 and 3: direct-to-LDS path only); bit 2: register staging;
                                      * bits 8+: M-tiles per L2 locality group; 8 / 9 / 10: the wave-specialised
                                      * kernels never / always with 192x192 / always with 128x192 tiles (11 / 12 were
                                      * round 4's deferred-epilogue lab variant: now plain 4-wave); 13 / 14: 64x128 (six-deep ring) / 64x192
                                      * (four-deep) tiles - what small-M GEMMs (the Temporal Transformer's 1920 rows into
                                      * N = 768, the 480 query rows) take by default;
                                      * -1 heuristic (fp32 GEMMs with M <= 32 then run a skinny VALU kernel) */
int hero_prof_enable(int on);
int hero_prof_read(int slot, double* total_ms, double* total_flops, long long* launches);

/* Box probes (bench.py: `roofline.peak_measured`, `.hbm_peak_measured`, `.clock_ghz`): what this box's matrix pipes and HBM
 * deliver right now.  These two entries TIME themselves with HIP events and synchronise `stream` (like hero_prof_read);
 * never call them under stream capture.
 * hero_probe_mfma: ~0.2 s of back-to-back v_mfma_f32_32x32x16_bf16 on every CU (8 waves per CU, 9 accumulators per wave,
 *   no memory traffic; the first ~70 ms after idle run at ramping clocks and are not timed): dense bf16 TFLOP/s and the
 *   sustained matrix clock (from the MFMA occupancy: 32 pipe cycles per 32x32x16 bf16 MFMA).  scratch: device memory, >= CUs * 2048 + 16 bytes.
 * hero_probe_hbm: a streaming copy src -> dst and a streaming read of src over `bytes` (>= 256 MiB each, beyond the
 *   Infinity Cache): copy_gbps = (bytes read + bytes written) / time, read_gbps = bytes / time.  dst[0..3] may be written
 *   by the read pass. */
int hero_probe_mfma(void* scratch, size_t scratch_bytes, double* tflops, double* ghz, hero_stream_t stream);
int hero_probe_hbm(const void* src, void* dst, size_t bytes, double* copy_gbps, double* read_gbps, hero_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* LayerNorm (apex FusedLayerNorm call sites: model/layers.py:52,79,171,246,338,               */
/* model/embed.py:25,93,99,143,171) fused with the embedding sums in front of it               */
/* (model/embed.py:28-58 SubEmbeddings, :102-117 ImageEmbeddings, :146-161 FrameEmbeddings)     */
/* and the dropout behind it.                                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct HeroLnFwd {
  const void* x;        /* [rows, cols] x_dtype, or NULL                                       */
  const float* tab[3];  /* fp32 tables [*, cols] gathered and added to x (NULL = unused)        */
  const int32_t* idx[3];/* per-row table row; NULL = row 0 for every row                       */
  const float* gamma;   /* [cols] */
  const float* beta;    /* [cols] */
  void* y;              /* [rows, cols] y_dtype                                                */
  void* pre;            /* optional [rows, cols] y_dtype: the summed LN input (for backward)   */
  float* mean;          /* optional [rows] */
  float* rstd;          /* optional [rows] */
  int rows, cols;
  float eps;
  int x_dtype, y_dtype;
  HeroDropout dropout;  /* on y; index = row*cols + col                                         */
} HeroLnFwd;
int hero_layernorm_fwd(const HeroLnFwd* a, hero_stream_t stream);

typedef struct HeroLnBwd {
  const void* x;        /* [rows, cols] x_dtype: the LN input (or `pre` saved by forward)       */
  const void* dy;       /* [rows, cols] dtype                                                   */
  const float* gamma;
  const float* mean;
  const float* rstd;
  void* dx;             /* optional [rows, cols] dtype                                          */
  void* dx_dropped;     /* optional [rows, cols] dtype: dx * mask(dropout_in) — the gradient    */
                        /* of the pre-dropout branch feeding this LN's residual sum             */
  float* dgamma;        /* optional [cols]: dgamma <- grad_beta*dgamma + sum                    */
  float* dbeta;         /* optional [cols]                                                      */
  float grad_beta;      /* 0 overwrite, 1 accumulate                                            */
  void* workspace;      /* hero_layernorm_bwd_workspace_bytes(rows, cols) bytes                 */
  int rows, cols;
  int x_dtype, dtype;
  HeroDropout dropout_out; /* the forward's output dropout (applied to dy first)                */
  HeroDropout dropout_in;  /* dropout of the GEMM epilogue that produced x (for dx_dropped)     */
  float* dbias_in;         /* optional [cols] (cols <= 1024): += sum_rows dx*mask(dropout_in) =     */
                           /* bias gradient of the linear layer feeding this LN, same pass         */
  int defer_fold;          /* 1 (fused path only: cols <= 1024 and dx wanted): leave the per-block   */
                           /* partial sums in `workspace` as fp32 [hero_layernorm_bwd_blocks(rows)]  */
                           /* [3][cols] (dgamma | dbeta | dbias_in) and do not touch the outputs -   */
                           /* the caller folds them later, e.g. many at once with hero_colsum_multi  */
} HeroLnBwd;
size_t hero_layernorm_bwd_workspace_bytes(int rows, int cols);
int hero_layernorm_bwd_blocks(int rows);   /* rows of the partial-sum matrix the fused backward writes */
int hero_layernorm_bwd(const HeroLnBwd* a, hero_stream_t stream);

/* Column sum: out[c] <- beta*out[c] + sum_r x[r, c]  (bias gradients of every nn.Linear).      */
size_t hero_colsum_workspace_bytes(int rows, int cols);
int hero_colsum(const void* x, float* out, int rows, int cols, int ld, int dtype, float beta,
                void* workspace, hero_stream_t stream);
/* Many column sums in TWO launches (row-chunk partials, then a fixed-order fold: deterministic, no atomics): the   */
/* bias gradients of all nn.Linear layers of a backward pass (model/layers.py:125-127, 175-179, 236-239, 250-254)   */
/* and the deferred LayerNorm partial folds, queued and flushed together.  dst[c] <- beta*dst[c] + sum_r src[r,c]. */
#define HERO_COLSUM_MULTI_MAX 64
typedef struct HeroColsum {
  const void* src;   /* [rows, ld] dtype, the `cols` columns starting at this pointer (16-byte aligned) */
  float* dst;        /* [cols]                                                                         */
  int rows, cols, ld, dtype;   /* cols, ld multiples of 4                                              */
  float beta;
  int row_cols;      /* with dst_rows: width of a destination row (cols = n_rows * row_cols)           */
  const int32_t* dst_rows;     /* optional (hero_colsum_multi only; beta must be 1): the sum of columns  */
                     /* [j * row_cols, (j + 1) * row_cols) is ADDED to dst + dst_rows[j] * row_cols - the fold   */
                     /* of a periodic position-id gradient ([S, L * D] view of [S * L, D]) goes straight into  */
                     /* the embedding table's rows (model/embed.py:28-58, 146-161); rows < 0 are dropped      */
} HeroColsum;
size_t hero_colsum_multi_workspace_bytes(const HeroColsum* p, int n);
int hero_colsum_multi(const HeroColsum* p, int n, void* workspace, hero_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Masked multi-head self-attention, head size 64 — model/layers.py:129-160                     */
/*   scores = Q K^T * scale + mask[s, key]; P = softmax(scores); ctx = dropout(P) V             */
/* qkv is the fused projection output [S*L, 3*H*64] (Q | K | V, head h at columns h*64).        */
/* ------------------------------------------------------------------------------------------ */
typedef struct HeroAttn {
  const void* qkv;    /* [S*L, 3*H*64] dtype                                                   */
  const float* mask;  /* [S, L] additive fp32 ((1-m)*-10000, model/layers.py:299-302) or NULL   */
  void* ctx;          /* fwd out [S*L, H*64] dtype; bwd in (optional): with it the bf16 backward of    */
                      /* 64 < L <= 256 runs on the matrix cores (delta_i = dO_i . ctx_i)               */
  float* probs;       /* [S, H, L, L] fp32 softmax output (pre-dropout); fwd out (may be NULL   */
                      /* for inference), bwd in                                                 */
  const void* dctx;   /* bwd in  [S*L, H*64] dtype                                              */
  void* dqkv;         /* bwd out [S*L, 3*H*64] dtype                                            */
  int S, L, H;
  float scale;        /* 1/sqrt(64)                                                             */
  int dtype;
  HeroDropout dropout; /* on P; index = ((s*H+h)*L + q)*round_up(L,4) + k                       */
  const int32_t* seq_off; /* optional [S+1] row offsets of a PACKED batch (L <= 64; bf16: 256): sequence s is */
                       /* rows [seq_off[s], seq_off[s+1]) of qkv/ctx/dctx/dqkv, at most L long;   */
                       /* probs keeps its [S, H, L, L] layout, dropout indices use L; mask NULL   */
  float* stats;        /* optional [S, H, L, 2] fp32 (row maximum, 1 / row sum) of the softmax: fwd out, bwd in.  Where */
                       /* hero_attention_stats_ok() says so, the backward takes stats INSTEAD of probs (probs NULL; it  */
                       /* needs `mask` again) and recomputes the probabilities bit-identically from q, k: 24x less      */
                       /* saved state at L = 24 (model/layers.py:129-160 keeps attention_probs for autograd)            */
} HeroAttn;
int hero_attention_fwd(const HeroAttn* a, hero_stream_t stream);
int hero_attention_bwd(const HeroAttn* a, hero_stream_t stream);
int hero_attention_max_len(int dtype, int backward);
/* longest sequence of a PACKED (seq_off) batch the kernels take for this dtype: 256 on the bf16 matrix-core */
/* kernels, 64 otherwise (fp32, or HERO_ATTN_MFMA=0) - callers fall back to the padded layout beyond it      */
int hero_attention_max_packed_len(int dtype);
int hero_attention_force_ppw(int ppw); /* tuning hook: (sequence, head) pairs per wave of the L <= 32 bf16 kernels, 1..3; 0 = heuristic */
/* 1 when forward + backward of this dtype / length run from `stats` without saved probabilities (bf16 matrix-core */
/* kernels, L <= 64)                                                                                               */
int hero_attention_stats_ok(int dtype, int L);

/* ------------------------------------------------------------------------------------------ */
/* Row gathers / scatters                                                                       */
/* ------------------------------------------------------------------------------------------ */
/* out[r] = idx[r] >= 0 ? a[idx[r]] : (idx[r] == -1 ? 0 : b[-idx[r]-2]).                         */
/* Replaces torch.gather(cat([img_emb, txt_emb]), gather_index) (model/encoder.py:271-279) and  */
/* serves as the backward of hero_csr_gather_sum.                                               */
int hero_gather_rows(const void* a, const void* b, const int32_t* idx, void* out, int rows, int cols,
                     int dtype, hero_stream_t stream);
/* out[r] = sum_{e in [offsets[r], offsets[r+1])} src[entries[e]].                               */
/* Replaces HierarchicalVlModel.collect_frame_outputs (model/model.py:156-187).                 */
/* inv[r] (r < na: rows of a) / inv[na + r] (rows of b) = the FIRST output row of a hero_gather_rows index that reads that
 * source row, -1 if none (na + nb <= 38400: one workgroup, LDS-resident, deterministic).  With it the backward of a gather
 * whose repeated references are known to carry no gradient (HERO's f_gather_index: a source row is re-referenced only from
 * padded positions behind its valid one, data/data.py:504-512, whose gradients are exactly zero) is ONE gather - no zero
 * fills, no scatter. */
int hero_inverse_first(const int32_t* idx, int n, int32_t* inv, int na, int nb, hero_stream_t stream);
int hero_csr_gather_sum(const void* src, const int32_t* offsets, const int32_t* entries, void* out,
                        int rows, int cols, int dtype, hero_stream_t stream);
/* dst[idx[r]] += src[r] (rows with idx[r] < 0 or == skip_idx are dropped).                      */
/* dst_dtype HERO_F32: embedding-table gradients (nn.Embedding backward, padding_idx = skip).   */
/* Two destinations as in hero_gather_rows when b != NULL.                                      */
int hero_scatter_add_rows(const void* src, const int32_t* idx, void* dst_a, void* dst_b, int rows,
                          int cols, int src_dtype, int dst_dtype, int skip_idx, hero_stream_t stream);
/* The same sum WITHOUT atomics, bit-reproducible (nn.Embedding backward into the fp32 gradient of word_embeddings,    */
/* model/embed.py:15-18, 28-58: destinations that receive three or more rows make the atomic version order-dependent): */
/*   hero_segment_sort        order[rows] = the row numbers sorted by (idx[row], row), rows with idx < 0 or == skip_idx */
/*                            last; one workgroup, stable radix sort; n_dst = rows of the table (bounds the key width); */
/*                            workspace of hero_segment_sort_workspace_bytes(rows) bytes.  Depends on idx only: cache it. */
/*   hero_scatter_add_sorted  blocks of 16 sorted rows, one wave each: a run of equal destinations inside a block is   */
/*                            added to dst[idx] by that wave (row order, fp32; the row has no other writer); a run that */
/*                            crosses blocks (a token at the head of every subtitle) leaves one partial sum per block   */
/*                            and the wave of its first block folds them in block order.  dst is fp32; cols % 4 == 0;   */
/*                            workspace of hero_scatter_add_sorted_workspace_bytes(rows, cols) bytes.                    */
size_t hero_segment_sort_workspace_bytes(int rows);
size_t hero_scatter_add_sorted_workspace_bytes(int rows, int cols);
int hero_segment_sort(const int32_t* idx, int rows, int n_dst, int skip_idx, int32_t* order, void* workspace,
                      hero_stream_t stream);
int hero_scatter_add_sorted(const void* src, const int32_t* idx, const int32_t* order, float* dst, int rows, int cols,
                            int src_dtype, int skip_idx, void* workspace, hero_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Elementwise                                                                                  */
/* ------------------------------------------------------------------------------------------ */
/* dst <- (dst_dtype) src, n elements (fp32 master weights -> compute copies, outputs -> fp32). */
int hero_cast(const void* src, void* dst, size_t n, int src_dtype, int dst_dtype, hero_stream_t stream);
/* dst[c * ldd + r] <- (dst_dtype) src[r * cols + c]: transposed compute copies of the fp32 master
 * weights, so that dgrad (dY W) also runs with both operands reduction-contiguous. */
int hero_transpose_cast(const float* src, void* dst, int rows, int cols, int ldd, int dst_dtype, hero_stream_t stream);
/* dx <- dy * (y > 0) — F.relu backward (model/layers.py:92).                                   */
int hero_relu_bwd(const void* dy, const void* y, void* dx, size_t n, int dtype, hero_stream_t stream);
/* dx <- dy * gelu_erf'(u) — backward of model/layers.py:16-25 outside the fused FFN block.     */
int hero_gelu_bwd(const void* dy, const void* u, void* dx, size_t n, int dtype, hero_stream_t stream);
/* y <- a + b (gradient fan-in where two consumers read one activation).                        */
int hero_add(const void* a, const void* b, void* y, size_t n, int dtype, hero_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Optimiser — optim/adamw.py:43-106 + clip_grad_norm_ (train_vcmr.py:257-260) over flat fp32   */
/* parameter / gradient arenas.                                                                 */
/* ------------------------------------------------------------------------------------------ */
/* sumsq[0] += sum g^2 (fp32 device scalar; caller zeroes it).  Deterministic (fixed summation  */
/* order): data-parallel replicas must derive the SAME clipping factor from the same gradients.  */
/* workspace: >= 2048 floats of device scratch.                                                  */
int hero_sumsq(const float* g, size_t n, float* sumsq, float* workspace, hero_stream_t stream);
typedef struct HeroAdamW {
  float* p;
  const float* g;
  float* m;
  float* v;
  size_t n;
  float lr, beta1, beta2, eps, weight_decay;
  int step;                /* 1-based, for bias correction                                     */
  const float* grad_sumsq; /* optional device scalar: global sum g^2 for clipping               */
  float max_grad_norm;     /* used when grad_sumsq != NULL: g *= min(1, max/(norm+1e-6))        */
  float grad_scale;        /* extra multiplier on g (e.g. 1/world_size); 1 = none               */
  void* shadow;            /* optional bf16 copy of p refreshed in the same pass               */
} HeroAdamW;
int hero_adamw(const HeroAdamW* a, hero_stream_t stream);

/* Multi-tensor form: ONE launch updates every tensor in a device-resident descriptor table (one
 * workgroup per hero_adamw_multi_chunk()-element chunk). `chunk_tensor[c]` is the descriptor a chunk
 * belongs to, `chunk_index[c]` its index within that tensor. A tensor's own step count is
 * step - step_lag (parameters that received no gradient in some steps lag behind, as with the
 * reference's per-parameter state['step']). */
typedef struct HeroTensorDesc {
  float* p;
  const float* g;
  float* m;
  float* v;
  uint64_t n;
  int32_t group;
  int32_t step_lag;
} HeroTensorDesc;
typedef struct HeroAdamWGroup {
  float lr, beta1, beta2, eps, weight_decay;
} HeroAdamWGroup;
typedef struct HeroAdamWMulti {
  const HeroTensorDesc* descs; /* device */
  const int32_t* chunk_tensor; /* device [n_chunks] */
  const int32_t* chunk_index;  /* device [n_chunks] */
  int n_chunks;
  HeroAdamWGroup groups[8];
  int step;
  const float* grad_sumsq;
  float max_grad_norm;
  float grad_scale;
  const int32_t* step_ptr; /* optional device scalar overriding `step` (hipGraph replays)      */
  const float* lr_ptr;     /* optional device array [8] overriding groups[].lr                 */
  int32_t* tensor_steps;   /* optional device array of PER-TENSOR step counts (round 3): a tensor's count lives  */
                           /* in slot descs[i].step_lag, is incremented by this call (a pre-pass over the        */
                           /* n_tensors descriptors) and is the step of its bias correction - the reference's    */
                           /* state['step'], which only advances when the parameter is updated                   */
                           /* (optim/adamw.py:71-72), kept on the device so that captured graphs of different    */
                           /* tasks do not advance the counters of parameters they skip                          */
  int n_tensors;           /* number of descriptors (needed with tensor_steps)                                   */
} HeroAdamWMulti;
int hero_adamw_multi(const HeroAdamWMulti* a, hero_stream_t stream);
int hero_adamw_multi_chunk(void);

/* Multi-tensor refresh of the compute copies of the fp32 master weights (after an optimiser     */
/* step): ONE launch casts / transposes every tensor of a device-resident descriptor table.     */
/*   transpose == 0: dst[r * ldd + c]        = (dst_dtype) src[r * cols + c]                      */
/*   transpose == 1: dst[c * ldd + r]        = (dst_dtype) src[r * cols + c]                      */
/* (dst already points at the tensor's slot inside a packed [sum N, K] / [K, sum N] buffer.)     */
/* Work is cut into tiles of 64 x 64 elements: tile_desc[t] = descriptor, tile_index[t] = tile   */
/* number inside it (row-major over ceil(rows/64) x ceil(cols/64)).                              */
typedef struct HeroCopyDesc {
  const float* src;
  void* dst;
  int32_t rows, cols, ldd;
  int32_t transpose;
  int32_t dst_dtype;
  int32_t pad_;
} HeroCopyDesc;
int hero_copy_multi(const HeroCopyDesc* descs, const int32_t* tile_desc, const int32_t* tile_index,
                    int n_tiles, hero_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* VSM / VCMR task head - model/pretrain.py:62-116,128-201,203-292,  */
/* model/encoder.py:460-471.  The few hundred tiny fp32 tensor ops between the encoders and the  */
/* three loss scalars as a dozen kernels.  Every op has its forward and backward here; the two   */
/* dense contractions in between (video_query_linear, normalised query x context scores) are    */
/* hero_gemm calls.  All row-major, fp32 unless a dtype is given.                                */
/* ------------------------------------------------------------------------------------------ */
/* Query pooling (QueryFeatEncoder.get_modularized_queries, model/encoder.py:460-471):           */
/*   sc[b,l] = <q[b,l,:], w>; att = softmax_l(sc*mask + (1-mask)*-1e4); pooled = sum_l att*q.    */
typedef struct HeroQueryPool {
  const void* q;          /* [B, L, D] dtype                                                    */
  const float* mask;      /* [B, L] 0/1                                                         */
  const float* w;         /* [D] modular_vector_mapping.weight                                  */
  float* pooled;          /* fwd out [B, D]                                                     */
  float* att;             /* fwd out / bwd in [B, L]                                            */
  const float* dpooled;   /* bwd in [B, D]                                                      */
  void* dq;               /* bwd out [B, L, D] dtype                                            */
  float* dw;              /* bwd out [B, D]: per-query shares of dw, OVERWRITTEN; the caller sums the  */
                          /* rows (hero_colsum): a fixed-order sum instead of B-way fp32 atomics       */
  int B, L, D, dtype;
} HeroQueryPool;
int hero_query_pool_fwd(const HeroQueryPool* a, hero_stream_t stream);
int hero_query_pool_bwd(const HeroQueryPool* a, hero_stream_t stream);

/* F.normalize(x, dim=-1, eps) (model/pretrain.py:370-371): y = x / max(||x||_2, eps).           */
/* rnorm[r] = 1/max(||x_r||, eps), negated when the clamp was active (backward then skips the   */
/* projection term).                                                                            */
typedef struct HeroRowNorm {
  const void* x;          /* [rows, cols] x_dtype                                               */
  float* y;               /* fwd out [rows, cols] fp32                                          */
  float* rnorm;           /* fwd out / bwd in [rows]                                            */
  const float* dy;        /* bwd in [rows, cols] fp32                                           */
  void* dx;               /* bwd out [rows, cols] x_dtype                                       */
  int rows, cols, x_dtype;
  float eps;
} HeroRowNorm;
int hero_rownorm_fwd(const HeroRowNorm* a, hero_stream_t stream);
int hero_rownorm_bwd(const HeroRowNorm* a, hero_stream_t stream);

/* Video-level scores (model/pretrain.py:364-382): s[m, n*L + l] = <qn[m], cn[n, l]> comes from  */
/* hero_gemm; this is mask_logits + max over l:  out[m,n] = max_l (s*mask[n,l] + (1-mask)*-1e4). */
/* Backward: g[m,n] = gc[0]*ds_ctx[m,n] + gq[0]*ds_q[m,n] (the ranking loss' two gradient        */
/* matrices and the upstream gradients of its two scalars) flows to the arg-max entry only:     */
/*   dqn[m,:]  = sum_n g*mask[n,arg]*cn[n,arg,:];   dcn[n,arg,:] += g*mask*qn[m,:]  for          */
/* n in [n0, n0+n_own) (a data-parallel rank only needs its own videos' rows; dcn is            */
/* [n_own*L, D] and is fully written).                                                          */
typedef struct HeroScoreMax {
  const float* s;         /* [M, N*L], row stride ld_s >= N*L                                   */
  const float* mask;      /* [N, L] 0/1                                                         */
  float* out;             /* fwd out [M, N]                                                     */
  int32_t* arg;           /* fwd out / bwd in [M, N]                                            */
  const float* ds_ctx;    /* bwd in [M, N]                                                      */
  const float* ds_q;      /* bwd in [M, N]                                                      */
  const float* gc;        /* bwd in: device scalar                                              */
  const float* gq;        /* bwd in: device scalar                                              */
  const float* qn;        /* bwd in [M, D]                                                      */
  const float* cn;        /* bwd in [N*L, D]                                                    */
  float* dqn;             /* bwd out [M, D]                                                     */
  float* dcn;             /* bwd out [n_own*L, D]                                               */
  int M, N, L, D, n0, n_own, ld_s;
  float gc_scale, gq_scale; /* bwd: *gc and *gq are multiplied by these (the loss weights, folded in; 1.0f = none) */
} HeroScoreMax;
/* out[s] = scales[s] * sum(src[s * seg_len .. (s + 1) * seg_len)), 1..4 segments, fixed summation order: the final
 * reductions of the loss head (sum of the start / end rows, means of the ranking-loss rows) with the loss weights
 * (model/pretrain.py:283-290) folded in - one launch instead of sum / mean / mul per loss.  `scales` is a HOST array. */
int hero_sums_scaled(const float* src, int n_segs, int seg_len, const float* scales, float* out, hero_stream_t stream);
int hero_score_max_fwd(const HeroScoreMax* a, hero_stream_t stream);
int hero_score_max_bwd(const HeroScoreMax* a, hero_stream_t stream);

/* In-batch ranking loss over ALL negatives (get_video_level_loss, use_all_neg=True,            */
/* model/pretrain.py:203-264): query m belongs to video m / (nq/nv).  loss_ctx_rows[m] = mean    */
/* over the other videos n of w*rl(s[m,own], s[m,n]); loss_q_rows[m] = mean over the queries m2  */
/* of other videos of w*rl(s[m,own], s[m2,own]); rl = hinge(margin) or log1p(exp(neg-pos));      */
/* w = 1, or with hard negatives hard_w for the `pool` largest negatives of that row and easy_w  */
/* for the rest.  ds_ctx / ds_q = gradients of mean(loss_ctx_rows) / mean(loss_q_rows) w.r.t. s  */
/* (both fully written).                                                                        */
typedef struct HeroRankLoss {
  const float* s;         /* [nq, nv]                                                           */
  float* loss_ctx_rows;   /* [nq]                                                               */
  float* loss_q_rows;     /* [nq]                                                               */
  float* ds_ctx;          /* [nq, nv]                                                           */
  float* ds_q;            /* [nq, nv]                                                           */
  int nq, nv;
  float margin;
  int lse;                /* 0 hinge, 1 lse                                                     */
  int hard;               /* use_hard_negative                                                  */
  int pool;               /* hard_pool_size                                                     */
  float hard_w, easy_w;   /* hard_neg_weight, 0.1                                               */
} HeroRankLoss;
int hero_rank_loss(const HeroRankLoss* a, hero_stream_t stream);

/* Start / end localisation (model/pretrain.py:128-166, 96-110): sim[b,l] = <q2[b], ctx[b,l]>,    */
/* st/ed = Conv1d(1,1,K,pad=K/2,no bias)(sim), mask_logits, cross-entropy against targets[b,0/1] */
/* (ignore_index -1, mean over the valid rows of each).  loss_rows[b] = ce_st[b]/n_st +           */
/* ce_ed[b]/n_ed (sum over b = the reference's loss_st_ed).                                      */
typedef struct HeroStEd {
  const float* q2;        /* [B, D] video_query_linear(modularized query)                       */
  const void* ctx;        /* [B, L, D] dtype: frame embeddings                                  */
  const float* mask;      /* [B, L] 0/1                                                         */
  const float* w_st;      /* [K]                                                                */
  const float* w_ed;      /* [K]                                                                */
  const int64_t* targets; /* [B, 2]                                                             */
  float* loss_rows;       /* fwd out [B]                                                        */
  float* p_st;            /* fwd out / bwd in [B, L] softmax                                    */
  float* p_ed;            /* fwd out / bwd in [B, L]                                            */
  float* sim;             /* fwd out / bwd in [B, L]                                            */
  const float* g;         /* bwd in: device scalar, upstream gradient of sum(loss_rows)         */
  float* dq2;             /* bwd out [B, D]                                                     */
  void* dctx;             /* bwd out [B, L, D] dtype                                            */
  float* dw_st;           /* bwd out [K], ACCUMULATED (+=)                                      */
  float* dw_ed;           /* bwd out [K], ACCUMULATED (+=)                                      */
  int B, L, D, K, dtype;
  float g_scale;          /* bwd: *g is multiplied by this (the loss weight, folded in; 1.0f = none)    */
  float* ws;              /* bwd scratch, hero_st_ed_bwd_workspace_bytes(B) bytes, ZERO before the first */
                          /* use (the kernel leaves its arrival counter at zero): per-pair shares of     */
                          /* dw_st / dw_ed, folded in pair order by the last workgroup to arrive - the   */
                          /* same sum every run (round 3: B-way fp32 atomics)                            */
} HeroStEd;
size_t hero_st_ed_bwd_workspace_bytes(int B);
int hero_st_ed_fwd(const HeroStEd* a, hero_stream_t stream);
int hero_st_ed_bwd(const HeroStEd* a, hero_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Softmax cross-entropy over wide logit rows (pre-training heads, BASELINE configs[3]):        */
/*   MLM  model/encoder.py:355-389, model/layers.py:330-354 (vocabulary GEMM output; `cols`     */
/*        excludes the vocabulary padding of model/encoder.py:232-233, ld includes it)          */
/*   MFM-NCE model/model.py:271-291 (logits / nce_temp)   FOM model/model.py:293-336            */
/* x = logits * inv_temp;  loss[r] = lse(x[r, :cols]) - x[r, labels[r]]  (0 when labels[r] ==   */
/* ignore_index);  dlogits[r, c] = (softmax(x[r])[c] - [c == labels[r]]) * dloss[r] * inv_temp, */
/* zero rows for ignored labels, zero in columns >= cols.  dlogits may alias logits.            */
/* ------------------------------------------------------------------------------------------ */
typedef struct HeroCrossEntropy {
  const void* logits;    /* [rows, ld] dtype                                                    */
  const int64_t* labels; /* [rows]                                                              */
  float* loss;           /* fwd out [rows]                                                      */
  float* lse;            /* fwd out / bwd in [rows]: log-sum-exp of x[r, :cols]                 */
  const float* dloss;    /* bwd in [rows]                                                       */
  void* dlogits;         /* bwd out [rows, ld] dtype                                            */
  int rows, cols, ld, dtype;
  float inv_temp;
  int64_t ignore_index;
} HeroCrossEntropy;
int hero_cross_entropy_fwd(const HeroCrossEntropy* a, hero_stream_t stream);
int hero_cross_entropy_bwd(const HeroCrossEntropy* a, hero_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Collate on the device: the index tensors of a batch from per-subtitle / per-video length      */
/* arrays (data/data.py:406-512 video_collate + get_gather_index; the python walk of             */
/* sub_idx2frame_idx in model/model.py:156-187).  All arrays int32 on the device.                */
/* ------------------------------------------------------------------------------------------ */
/* gather_index / attn_mask [T, out_size] int64 (data/data.py:504-512, 380-382); out_size is the reference's  */
/* f_attn_masks width = max over rows of (frame slots + tokens) (data/data.py:433-436), <= max_vl + max_sl   */
int hero_collate_subs(const int32_t* sub_nfrm, const int32_t* sub_ntok, int64_t* gather_index, int64_t* attn_mask, int T,
                      int max_vl, int out_size, hero_stream_t stream);
/* f_v_feats [T, max_vl, D] fp32 <- c_v_feats [B, NF, D]: the per-subtitle frame feature copies of            */
/* VideoFeatSubTokDataset.__getitem__ (data/data.py:371-379: frames outside [0, vid_nfrm) dropped, zero rows   */
/* after the kept ones) made on the device; row_vid [T] is scratch (video of each subtitle row).  D % 4 == 0.  */
int hero_collate_gather_feats(const float* c_v_feats, float* f_v_feats, const int32_t* vid_sub_off, const int32_t* vid_nfrm,
                              const int32_t* sub_frm_off, const int32_t* sub_frm, int32_t* row_vid, int T, int max_vl, int B, int NF,
                              int D, hero_stream_t stream);
/* attn_mask [B, NF] int64 = f < vid_nfrm[b] */
int hero_collate_clip_mask(const int32_t* vid_nfrm, int64_t* attn_mask, int B, int NF, hero_stream_t stream);
/* Frame map of collect_frame_outputs, two passes around an exclusive scan done by the caller:       */
/* fill = 0: counts[b * NF + f] = number of (subtitle, slot) pairs matched to frame f of video b;     */
/* fill = 1: entries[offsets[bf] ...] = flat source rows (subtitle row * Lf + slot) in subtitle /     */
/* slot order, inverse[source row] = bf (inverse must be pre-filled with -1).                         */
int hero_collate_frame_map(const int32_t* vid_sub_off, const int32_t* sub_frm_off, const int32_t* sub_frm, const int32_t* offsets,
                           int32_t* counts, int32_t* entries, int32_t* inverse, int B, int NF, int Lf, int fill,
                           hero_stream_t stream);

/* Everything the model derives from the int64 index / mask tensors of a batch, in one launch: additive attention   */
/* masks (1 - m) * -10000 (model/layers.py:299-302), fp32 masks, int32 row indices, and f_gather_index as flat rows   */
/* of hero_gather_rows (>= 0: image row, <= -2: text row; model/encoder.py:271-279).  src int64, n elements.          */
enum { HERO_DERIVE_MASK_ADD = 0, HERO_DERIVE_F32 = 1, HERO_DERIVE_I32 = 2, HERO_DERIVE_FLAT_GATHER = 3 };
#define HERO_DERIVE_MAX 16
typedef struct HeroDerive {
  const int64_t* src;
  void* dst;        /* fp32 (MASK_ADD, F32) or int32 (I32, FLAT_GATHER), n elements */
  int64_t n;
  int mode;
  int p0, p1, p2;   /* FLAT_GATHER: row width of src, max_vl, max_sl */
} HeroDerive;
int hero_derive_multi(const HeroDerive* d, int n, hero_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Gradient exchange over RCCL (round 4) - replaces Horovod's allreduce_ / broadcast_ */
/* (utils/distributed.py:19-46, 103-151) and the negatives' allgather (model/pretrain.py:427-451).  */
/* Every collective is ENQUEUED on the caller's stream (no stream, thread or synchronisation of its  */
/* own: a captured step simply contains it).  librccl.so is dlopen-ed by the first call;             */
/* hero_comm_available() = 0 where it is missing.  One communicator per process = per GPU.          */
/* ------------------------------------------------------------------------------------------ */
typedef struct HeroCommBucket {
  void* buf;        /* device buffer, reduced in place                                          */
  size_t count;     /* elements                                                                 */
  int dtype;        /* HERO_F32 or HERO_BF16 (the wire format of hero_amd.utils.distributed.GradArena) */
  int pad_;
} HeroCommBucket;
int hero_comm_available(void);
int hero_comm_unique_id(void* id128);                          /* rank 0: 128 bytes to hand to every rank       */
int hero_comm_init(const void* id128, int rank, int world, void** comm_out);   /* collective; current device  */
int hero_comm_destroy(void* comm);
int hero_comm_rank(void* comm);
int hero_comm_world(void* comm);
int hero_comm_allreduce_buckets(void* comm, const HeroCommBucket* buckets, int n, hero_stream_t stream);  /* SUM, <= 64 buckets, one group */
int hero_comm_broadcast(void* comm, void* buf, size_t bytes, int root, hero_stream_t stream);
int hero_comm_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, hero_stream_t stream);
/* recv = the ranks' buffers back to back; bytes_per_rank[world] is a HOST array, equal on every rank (the padded gather   */
/* + slice of model/pretrain.py:383-401 as one group of broadcasts)                                                       */
int hero_comm_allgather_var(void* comm, const void* send, void* recv, const size_t* bytes_per_rank, hero_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HERO_HIP_H_ */
