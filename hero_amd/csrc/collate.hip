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
//   f_gather_index, f_attn_masks [T, max_vl + max_sl] int64, c_attn_masks [B, NF] int64,
//   frame map CSR: counts -> (exclusive scan by the caller) -> entries [nnz], inverse [T * Lf] int32
#include "common.h"

namespace hero {
namespace {

// data/data.py:504-512 (get_gather_index) and :380-382 (a subtitle without frames keeps one zero frame slot
// whose mask bit is 0); one thread per (row, position)
__global__ void collate_subs_kernel(const int32_t* __restrict__ nfrm, const int32_t* __restrict__ ntok, int64_t* __restrict__ gidx,
                                    int64_t* __restrict__ mask, int T, int max_vl, int max_sl) {
  const int Lf = max_vl + max_sl;
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

// one thread per output frame (video b, frame f): walk the video's subtitles in row order, their frame lists in slot
// order - the order of the host builder's stable sort - and count (FILL = false) or record (FILL = true) the matches
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
                                 int max_vl, int max_sl, hero_stream_t stream) {
  HERO_REQUIRE(sub_nfrm && sub_ntok && gather_index && attn_mask, "hero_collate_subs: null pointer");
  HERO_REQUIRE(T >= 0 && max_vl > 0 && max_sl > 0, "hero_collate_subs: bad dims");
  if (T == 0) return HERO_OK;
  const size_t n = (size_t)T * (max_vl + max_sl);
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(collate_subs_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), sub_nfrm, sub_ntok, gather_index,
                     attn_mask, T, max_vl, max_sl);
  return check_launch("hero_collate_subs");
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
