// Batch index tensors built ON THE DEVICE from per-subtitle / per-video length arrays.
//
// The reference's collate (data/data.py:406-512 video_collate / get_gather_index) builds the interleave
// index, the attention masks and - per forward, in HierarchicalVlModel.collect_frame_outputs
// (model/model.py:156-187) - walks python lists of (subtitle, frame list) pairs on the host.  Here the
// loader hands over five small int32 arrays and everything else is derived by kernels, in place, into
// buffers of fixed capacity (so a captured hipGraph sees new batches of the same shape):
//   sub_nfrm[T], sub_ntok[T]           frames matched to / tokens (incl. SEP) of each subtitle row
//   sub_frm_off[T + 1], sub_frm[...]   the frame indices (inside its video) of each subtitle
//   vid_sub_off[B + 1], vid_nfrm[B]    first subtitle row / frame count of each video
// Outputs, identical bit for bit to hero_amd/synth.py + hero_amd.model.model.build_frame_map:
//   f_gather_index, f_attn_masks [T, out_size] int64 (out_size = max over rows of frame slots + tokens: the width
//   the reference's pad_sequence gives f_attn_masks, data/data.py:433-436 - NOT max_vl + max_sl), c_attn_masks [B, NF] int64,
//   f_v_feats [T, max_vl, D] gathered from c_v_feats [B, NF, D] (data/data.py:374-379: index_select per subtitle),
//   frame map CSR: counts -> (exclusive scan by the caller) -> entries [nnz], inverse [T * Lf] int32
#include "common.h"

namespace hero {
namespace {

// data/data.py:504-512 (get_gather_index) and :380-382 (a subtitle without frames keeps one zero frame slot
// whose mask bit is 0); one thread per (row, position)
__global__ void collate_subs_kernel(const int32_t* __restrict__ nfrm, const int32_t* __restrict__ ntok, int64_t* __restrict__ gidx,
                                    int64_t* __restrict__ mask, int T, int max_vl, int Lf) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < (size_t)T * Lf; q += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(q / Lf), p = (int)(q - (size_t)r * Lf);
    const int nf = nfrm[r], nt = ntok[r], eff = nf > 0 ? nf : 1;
    gidx[q] = (p >= eff && p < eff + nt) ? (int64_t)(max_vl + p - eff) : (int64_t)p;
    mask[q] = nf > 0 ? (p < nf + nt ? 1 : 0) : ((p >= 1 && p < 1 + nt) ? 1 : 0);
  }
}

__global__ void collate_clip_mask_kernel(const int32_t* __restrict__ vid_nfrm, int64_t* __restrict__ mask, int B, int NF) {
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < B * NF; q += gridDim.x * blockDim.x) mask[q] = (q % NF) < vid_nfrm[q / NF] ? 1 : 0;
}

// f_v_feats[r, k, :] = c_v_feats[video(r), k-th frame of subtitle r that lies inside its video, :], zero rows after them - the
// per-subtitle copies of the frame features (data/data.py:371-379: frames outside the clipped video are dropped, then
// index_select) made on the device, so that the host hands over c_v_feats only.  One workgroup per (row, slot),
// 16-byte accesses.  `row_vid[r]` = the video of subtitle row r.
__global__ void gather_feats_kernel(const float4* __restrict__ c_feats, float4* __restrict__ f_feats, const int32_t* __restrict__ row_vid,
                                    const int32_t* __restrict__ vid_nfrm, const int32_t* __restrict__ frm_off, const int32_t* __restrict__ frm,
                                    int max_vl, int NF, int D4) {
  const int r = blockIdx.x / max_vl, k = blockIdx.x - r * max_vl;
  float4* dst = f_feats + (size_t)blockIdx.x * D4;
  const int nfv = vid_nfrm[row_vid[r]];
  int f = -1, seen = 0;
  for (int j = frm_off[r]; j < frm_off[r + 1]; ++j) {          // wave-uniform walk of a short list
    const int c = frm[j];
    if (c >= 0 && c < nfv) {
      if (seen == k) { f = c; break; }
      ++seen;
    }
  }
  if (f < 0 || f >= NF) {
    for (int c = threadIdx.x; c < D4; c += blockDim.x) dst[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float4* src = c_feats + ((size_t)row_vid[r] * NF + f) * D4;
  for (int c = threadIdx.x; c < D4; c += blockDim.x) dst[c] = src[c];
}

// row_vid[r] = video of subtitle row r (vid_sub_off is the exclusive scan of num_subs)
__global__ void row_video_kernel(const int32_t* __restrict__ vid_sub_off, int32_t* __restrict__ row_vid, int B) {
  for (int b = blockIdx.x; b < B; b += gridDim.x)
    for (int r = vid_sub_off[b] + threadIdx.x; r < vid_sub_off[b + 1]; r += blockDim.x) row_vid[r] = b;
}

// one thread per output frame (video b, frame f): walk the video's subtitles in row order, their frame lists in slot
// order - the order of the host builder's stable sort - and count (FILL = false) or record (FILL = true) the matches
// Tensors DERIVED from the int64 index / mask tensors of a batch (additive attention masks, fp32 masks, int32 row
// indices, the flat form of f_gather_index), all of a batch in ONE launch: blockIdx.y = descriptor.
struct DeriveArgs { HeroDerive d[HERO_DERIVE_MAX]; };
__global__ void derive_multi_kernel(DeriveArgs a) {
  const HeroDerive d = a.d[blockIdx.y];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = d.src[i];
    switch (d.mode) {
      case HERO_DERIVE_MASK_ADD: static_cast<float*>(d.dst)[i] = (1.0f - (float)x) * -10000.0f; break;      // model/layers.py:299-302
      case HERO_DERIVE_F32: static_cast<float*>(d.dst)[i] = (float)x; break;
      case HERO_DERIVE_I32: static_cast<int32_t*>(d.dst)[i] = (int32_t)x; break;
      default: {                                                     // HERO_DERIVE_FLAT_GATHER: p0 = row width, p1 = max_vl, p2 = max_sl
        const int64_t row = i / d.p0;
        static_cast<int32_t*>(d.dst)[i] = (int32_t)(x < d.p1 ? row * d.p1 + x : -(row * d.p2 + (x - d.p1)) - 2);
      }
    }
  }
}

template <bool FILL>
__global__ void frame_map_kernel(const int32_t* __restrict__ vid_sub_off, const int32_t* __restrict__ sub_frm_off,
                                 const int32_t* __restrict__ sub_frm, const int32_t* __restrict__ offsets, int32_t* __restrict__ counts,
                                 int32_t* __restrict__ entries, int32_t* __restrict__ inverse, int B, int NF, int Lf) {
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < B * NF; q += gridDim.x * blockDim.x) {
    const int b = q / NF, f = q - b * NF;
    int n = 0;
    int e = FILL ? offsets[q] : 0;
    for (int s = vid_sub_off[b]; s < vid_sub_off[b + 1]; ++s) {
      const int o0 = sub_frm_off[s], o1 = sub_frm_off[s + 1];
      for (int j = o0; j < o1; ++j)
        if (sub_frm[j] == f) {
          if (FILL) {
            const int src = s * Lf + (j - o0);
            entries[e++] = src;
            inverse[src] = q;
          }
          ++n;
        }
    }
    if (!FILL) counts[q] = n;
  }
}

}  // namespace
}  // namespace hero

using namespace hero;

extern "C" int hero_collate_subs(const int32_t* sub_nfrm, const int32_t* sub_ntok, int64_t* gather_index, int64_t* attn_mask, int T,
                                 int max_vl, int out_size, hero_stream_t stream) {
  HERO_REQUIRE(sub_nfrm && sub_ntok && gather_index && attn_mask, "hero_collate_subs: null pointer");
  HERO_REQUIRE(T >= 0 && max_vl > 0 && out_size > 0, "hero_collate_subs: bad dims");
  if (T == 0) return HERO_OK;
  const size_t n = (size_t)T * out_size;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(collate_subs_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), sub_nfrm, sub_ntok, gather_index,
                     attn_mask, T, max_vl, out_size);
  return check_launch("hero_collate_subs");
}

extern "C" int hero_collate_gather_feats(const float* c_v_feats, float* f_v_feats, const int32_t* vid_sub_off, const int32_t* vid_nfrm,
                                         const int32_t* sub_frm_off, const int32_t* sub_frm, int32_t* row_vid, int T, int max_vl, int B,
                                         int NF, int D, hero_stream_t stream) {
  HERO_REQUIRE(c_v_feats && f_v_feats && vid_sub_off && vid_nfrm && sub_frm_off && sub_frm && row_vid, "hero_collate_gather_feats: null pointer");
  HERO_REQUIRE(T > 0 && max_vl > 0 && B > 0 && NF > 0 && D > 0 && D % 4 == 0, "hero_collate_gather_feats: bad dims (D must be a multiple of 4)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(row_video_kernel, dim3(B < 1024 ? B : 1024), dim3(64), 0, s, vid_sub_off, row_vid, B);
  hipLaunchKernelGGL(gather_feats_kernel, dim3(T * max_vl), dim3(256), 0, s, reinterpret_cast<const float4*>(c_v_feats),
                     reinterpret_cast<float4*>(f_v_feats), row_vid, vid_nfrm, sub_frm_off, sub_frm, max_vl, NF, D / 4);
  return check_launch("hero_collate_gather_feats");
}

extern "C" int hero_derive_multi(const HeroDerive* d, int n, hero_stream_t stream) {
  HERO_REQUIRE(d && n >= 1 && n <= HERO_DERIVE_MAX, "hero_derive_multi: 1..%d descriptors", HERO_DERIVE_MAX);
  DeriveArgs a;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    HERO_REQUIRE(d[i].src && d[i].dst && d[i].n >= 0 && d[i].mode >= HERO_DERIVE_MASK_ADD && d[i].mode <= HERO_DERIVE_FLAT_GATHER,
                 "hero_derive_multi: bad descriptor %d", i);
    HERO_REQUIRE(d[i].mode != HERO_DERIVE_FLAT_GATHER || (d[i].p0 > 0 && d[i].p1 > 0 && d[i].p2 > 0), "hero_derive_multi: flat gather needs widths");
    a.d[i] = d[i];
    most = d[i].n > most ? d[i].n : most;
  }
  if (most == 0) return HERO_OK;
  const int gx = (int)((most + 255) / 256 < 1024 ? (most + 255) / 256 : 1024);
  hipLaunchKernelGGL(derive_multi_kernel, dim3(gx, n), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("hero_derive_multi");
}

extern "C" int hero_collate_clip_mask(const int32_t* vid_nfrm, int64_t* attn_mask, int B, int NF, hero_stream_t stream) {
  HERO_REQUIRE(vid_nfrm && attn_mask && B > 0 && NF > 0, "hero_collate_clip_mask: bad arguments");
  hipLaunchKernelGGL(collate_clip_mask_kernel, dim3((B * NF + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), vid_nfrm,
                     attn_mask, B, NF);
  return check_launch("hero_collate_clip_mask");
}

extern "C" int hero_collate_frame_map(const int32_t* vid_sub_off, const int32_t* sub_frm_off, const int32_t* sub_frm,
                                      const int32_t* offsets, int32_t* counts, int32_t* entries, int32_t* inverse, int B, int NF,
                                      int Lf, int fill, hero_stream_t stream) {
  HERO_REQUIRE(vid_sub_off && sub_frm_off && sub_frm && B > 0 && NF > 0 && Lf > 0, "hero_collate_frame_map: bad arguments");
  HERO_REQUIRE(fill ? (offsets && entries && inverse) : (counts != nullptr), "hero_collate_frame_map: missing output for this pass");
  const int grid = (B * NF + 127) / 128;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (fill) hipLaunchKernelGGL(frame_map_kernel<true>, dim3(grid), dim3(128), 0, s, vid_sub_off, sub_frm_off, sub_frm, offsets, counts, entries, inverse, B, NF, Lf);
  else hipLaunchKernelGGL(frame_map_kernel<false>, dim3(grid), dim3(128), 0, s, vid_sub_off, sub_frm_off, sub_frm, offsets, counts, entries, inverse, B, NF, Lf);
  return check_launch("hero_collate_frame_map");
}
