// Bounding-box voting (https://arxiv.org/abs/1505.01749) with the reference's optional score re-estimation:
// pet/lib/ops/csrc/Box_ops/box_voting.cu:24-210, and with labels box_ml_voting.cu (pairs of different labels never vote).  The reference materialises a [N, K, 7] tensor (weighted corners,
// score weight, score, box weight per (top box, candidate) pair) and sums it over K with a framework reduction; here
// one wavefront owns a top box, its lanes sweep the K candidates accumulating the seven sums in registers, and a wave
// reduction finishes -- no intermediate tensor, one launch.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__global__ void __launch_bounds__(256) box_voting_kernel(const float* __restrict__ boxes,
                                                         const float* __restrict__ scores,
                                                         const int64_t* __restrict__ labels, int N,
                                                         const float* __restrict__ qboxes,
                                                         const float* __restrict__ qscores,
                                                         const int64_t* __restrict__ qlabels, int K, int method,
                                                         float beta, float thr, float* __restrict__ out_boxes,
                                                         float* __restrict__ out_scores) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;
  const float4 a = *(const float4*)(boxes + 4 * (size_t)i);
  const float sa = (a.z - a.x) * (a.w - a.y);
  const int64_t la = labels ? labels[i] : 0;
  float sx1 = 0.f, sy1 = 0.f, sx2 = 0.f, sy2 = 0.f, snum = 0.f, ssc = 0.f, sbw = 0.f;
  for (int j = lane; j < K; j += 64) {
    const float4 b = *(const float4*)(qboxes + 4 * (size_t)j);
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z), top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    float iou = inter / (sa + (b.z - b.x) * (b.w - b.y) - inter);                // box_voting.cu:15-22
    if (labels && qlabels[j] != la) iou = 0.f;                                    // box_ml_voting.cu:17
    if (!(iou >= thr)) continue;
    const float wt = qscores[j];
    sx1 += b.x * wt; sy1 += b.y * wt; sx2 += b.z * wt; sy2 += b.w * wt;
    float sw = 1.f, sc = wt;                                                      // :97-124
    if (method == 1) {
      if (wt != 0.f) sc = 1.f / (1.f + powf(1.f / wt - 1.f, 1.f / beta));
    } else if (method == 3) {
      sw = iou;
      sc = iou * wt;
    } else if (method == 4) {
      sc = powf(wt, beta);
    }
    snum += sw; ssc += sc; sbw += wt;
  }
  sx1 = wave_sum(sx1); sy1 = wave_sum(sy1); sx2 = wave_sum(sx2); sy2 = wave_sum(sy2);
  snum = wave_sum(snum); ssc = wave_sum(ssc); sbw = wave_sum(sbw);
  if (lane == 0) {
    *(float4*)(out_boxes + 4 * (size_t)i) = make_float4(sx1 / sbw, sy1 / sbw, sx2 / sbw, sy2 / sbw);   // :181-185
    float s = scores[i];                                                          // :187-204
    if (method >= 1 && method <= 3) s = ssc / snum;
    else if (method == 4) s = powf(ssc / snum, 1.f / beta);
    else if (method == 5) s = ssc / powf(snum, beta);
    out_scores[i] = s;
  }
}

}  // namespace

CPM_EXPORT int cpm_box_voting(const float* boxes, const float* scores, const int64_t* labels, int N,
                              const float* query_boxes, const float* query_scores, const int64_t* query_labels, int K,
                              int scoring_method, float beta, float threshold, float* out_boxes, float* out_scores,
                              void* stream) {
  CPM_REQUIRE(N >= 0 && K >= 0, "bad shape");
  CPM_REQUIRE(scoring_method >= 0 && scoring_method <= 5, "scoring method 0..5 (BOX_VOTING_METHODS)");
  if (N == 0) return CPM_OK;
  CPM_REQUIRE(boxes && scores && out_boxes && out_scores && (K == 0 || (query_boxes && query_scores)), "null pointer");
  CPM_REQUIRE((labels == nullptr) == (query_labels == nullptr) || K == 0, "labels and query_labels come together");
  CPM_REQUIRE((((uintptr_t)boxes | (uintptr_t)query_boxes | (uintptr_t)out_boxes) & 15) == 0,
              "boxes must be 16-byte aligned");
  hipLaunchKernelGGL(box_voting_kernel, dim3(cpm::cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, boxes, scores, labels, N,
                     query_boxes, query_scores, query_labels, K, scoring_method, beta, threshold, out_boxes, out_scores);
  return cpm::check_launch("box_voting");
}
