// Soft-NMS on the device, batched over segments (classes): the reference runs it on the CPU, class by class, after
// copying boxes and scores to the host (pet/lib/ops/boxlist_ops.py:70-91 -> csrc/NMS/soft_nms.cpp:5-110); with
// `labels` it is the multi-label variant (boxlist_ops.py:94-117 -> csrc/NMS/ml_soft_nms.cpp:5-122): only boxes with
// the label of the selected box decay, and the selection stops after `topk` picks.
//
// The algorithm is inherently sequential -- pick the best remaining box, decay the others by their overlap with it,
// drop the ones that fall under min_score by swapping them with the last element -- and its OUTPUT ORDER and its
// tie-breaking depend on that exact array shuffling.  One wavefront per segment reproduces it step for step with the
// segment resident in LDS (n <= 2048 boxes):
//   * argmax with "first position wins" over the live range (wave reduction of (score, position));
//   * the swap to the front;
//   * the decay of the rest in parallel (same fp32 expression tree; built -ffp-contract=off);
//   * the removal pass.  Sequentially a dead element is overwritten by the current last one, which is then examined
//     in turn; the net effect is a two-pointer compaction: dead slots below the new length are filled, lowest first,
//     by the live elements above it, highest first.  That is evaluated in parallel from two prefix counts.
// Linear and hard decay are bit-identical to the reference; gaussian decay uses the device expf (ulp-level
// differences).
#include "common.h"

namespace {

constexpr int SN_MAX = 2048;
constexpr int SN_PER = SN_MAX / 64;

struct SoftSeg { int32_t off[65]; int P; };

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int* total) {
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  *total = __shfl(incl, 63, 64);
  return incl - v;
}

// areas are recomputed from the corners where the reference keeps an array (same fp32 expression, same value): the
// freed LDS holds the labels
#define SN_AREA(q) ((x2[q] - x1[q]) * (y2[q] - y1[q]))

__global__ void __launch_bounds__(64) soft_nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                      const int64_t* __restrict__ labels, SoftSeg T, float thr,
                                                      int method, float sigma, float min_score, int topk,
                                                      float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                      int64_t* __restrict__ out_labels, int64_t* __restrict__ out_idx,
                                                      int32_t* __restrict__ out_count) {
  __shared__ float x1[SN_MAX], y1[SN_MAX], x2[SN_MAX], y2[SN_MAX], sc[SN_MAX];
  __shared__ int lb[SN_MAX];
  __shared__ int id[SN_MAX];
  __shared__ int holes[SN_MAX];
  const int p = blockIdx.x, lane = threadIdx.x;
  const int base = T.off[p];
  int nd = T.off[p + 1] - base;
  for (int i = lane; i < nd; i += 64) {
    const float4 b = *(const float4*)(boxes + 4 * (size_t)(base + i));
    x1[i] = b.x; y1[i] = b.y; x2[i] = b.z; y2[i] = b.w;
    sc[i] = scores[base + i];
    lb[i] = labels ? (int)labels[base + i] : 0;
    id[i] = i;
  }
  __builtin_amdgcn_wave_barrier();
  for (int i = 0; i < nd; ++i) {
    if (labels && topk == i) { nd = topk; break; }          // ml_soft_nms.cpp:31-35 (0 keeps nothing, < 0 never stops)
    // 1. first position of the maximum score in [i, nd)
    float best = -INFINITY;
    int bpos = 0x7fffffff;
    for (int q = i + lane; q < nd; q += 64) {
      const float s = sc[q];
      if (s > best || bpos == 0x7fffffff) { if (s > best || bpos == 0x7fffffff) { best = s; bpos = q; } }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const float ob = __shfl_xor(best, d, 64);
      const int op = __shfl_xor(bpos, d, 64);
      if (op != 0x7fffffff && (bpos == 0x7fffffff || ob > best || (ob == best && op < bpos))) { best = ob; bpos = op; }
    }
    // 2. swap it to position i (every lane holds the same bpos)
    const float ix1 = x1[bpos], iy1 = y1[bpos], ix2 = x2[bpos], iy2 = y2[bpos], isc = sc[bpos];
    const float iar = (ix2 - ix1) * (iy2 - iy1);
    const int iid = id[bpos], ilb = lb[bpos];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      x1[bpos] = x1[i]; y1[bpos] = y1[i]; x2[bpos] = x2[i]; y2[bpos] = y2[i]; sc[bpos] = sc[i]; lb[bpos] = lb[i];
      id[bpos] = id[i];
      x1[i] = ix1; y1[i] = iy1; x2[i] = ix2; y2[i] = iy2; sc[i] = isc; lb[i] = ilb; id[i] = iid;
    }
    __builtin_amdgcn_wave_barrier();
    // 3. decay (i, nd); each lane owns a contiguous run so that the prefix counts below follow positions
    const int rest = nd - (i + 1);
    const int per = (rest + 63) / 64;
    const int lo = i + 1 + lane * per, hi = min(lo + per, nd);
    int dead_cnt = 0;
    for (int q = lo; q < hi; ++q) {
      float s = sc[q];
      if (lb[q] == ilb) {
        const float inter = fmaxf(0.f, fminf(ix2, x2[q]) - fmaxf(ix1, x1[q])) *
                            fmaxf(0.f, fminf(iy2, y2[q]) - fmaxf(iy1, y1[q]));
        const float ovr = inter / (iar + SN_AREA(q) - inter);
        if (method == 1) {
          if (ovr > thr) s = (1.f - ovr) * s;
        } else if (method == 2) {
          s = expf(-(ovr * ovr) / sigma) * s;
        } else {
          if (ovr > thr) s = 0.f;
        }
        sc[q] = s;
      }
      dead_cnt += (s < min_score) ? 1 : 0;
    }
    int total_dead;
    const int dead_before = wave_excl_scan(dead_cnt, lane, &total_dead);
    if (total_dead == 0) continue;                          // the common case: nothing to remove
    // 4. removal == two-pointer compaction
    const int new_nd = nd - total_dead;
    // holes: dead positions below new_nd, ranked upward
    {
      int r = dead_before;
      for (int q = lo; q < hi; ++q)
        if (sc[q] < min_score) { if (q < new_nd) holes[r] = q; ++r; }
    }
    // fillers: live positions >= new_nd, ranked downward.  live_after(q) = live elements in (q, nd)
    const int live_cnt = (hi > lo ? hi - lo : 0) - dead_cnt;
    int total_live;
    const int live_before = wave_excl_scan(live_cnt, lane, &total_live);
    __builtin_amdgcn_wave_barrier();
    {
      int seen = live_before;                                // live elements in (i, q)
      for (int q = lo; q < hi; ++q) {
        if (!(sc[q] < min_score)) {
          if (q >= new_nd) {
            const int rank = total_live - seen - 1;          // live elements above q
            const int h = holes[rank];
            x1[h] = x1[q]; y1[h] = y1[q]; x2[h] = x2[q]; y2[h] = y2[q]; sc[h] = sc[q]; lb[h] = lb[q]; id[h] = id[q];
          }
          ++seen;
        }
      }
    }
    nd = new_nd;
    __builtin_amdgcn_wave_barrier();
  }
  for (int i = lane; i < nd; i += 64) {
    *(float4*)(out_boxes + 4 * (size_t)(base + i)) = make_float4(x1[i], y1[i], x2[i], y2[i]);
    out_scores[base + i] = sc[i];
    if (out_labels) out_labels[base + i] = (int64_t)lb[i];
    out_idx[base + i] = (int64_t)id[i];
  }
  if (lane == 0) out_count[p] = nd;
}

}  // namespace

CPM_EXPORT int cpm_soft_nms_batched(const float* boxes, const float* scores, const int64_t* labels,
                                    const int32_t* h_offsets, int P, float iou_threshold, int method, float sigma,
                                    float min_score, int topk, float* out_boxes, float* out_scores,
                                    int64_t* out_labels, int64_t* out_idx, int32_t* out_counts, void* stream) {
  CPM_REQUIRE(P >= 0 && P <= 64, "1..64 segments per call");
  if (P == 0) return CPM_OK;
  CPM_REQUIRE(h_offsets && out_counts, "null pointer");
  CPM_REQUIRE(method >= 0 && method <= 2, "method: 0 hard, 1 linear, 2 gaussian");
  SoftSeg T = {};
  T.P = P;
  for (int i = 0; i <= P; ++i) {
    T.off[i] = h_offsets[i];
    if (i) {
      CPM_REQUIRE(h_offsets[i] >= h_offsets[i - 1], "offsets must not decrease");
      CPM_REQUIRE(h_offsets[i] - h_offsets[i - 1] <= SN_MAX, "more than 2048 boxes in a segment");
    }
  }
  CPM_REQUIRE(h_offsets[0] == 0, "offsets start at 0");
  if (h_offsets[P] > 0) {
    CPM_REQUIRE(boxes && scores && out_boxes && out_scores && out_idx, "null pointer");
    CPM_REQUIRE((((uintptr_t)boxes) & 15) == 0 && (((uintptr_t)out_boxes) & 15) == 0, "boxes must be 16-byte aligned");
  }
  hipLaunchKernelGGL(soft_nms_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, boxes, scores, labels, T,
                     iou_threshold, method, sigma, min_score, topk, out_boxes, out_scores, out_labels, out_idx,
                     out_counts);
  return cpm::check_launch("soft_nms_batched");
}
