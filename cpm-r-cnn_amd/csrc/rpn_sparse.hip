// Sparse backward of the RPN head (rpn/rpn.py:34-41 under rpn/loss.py:88-126).
//
// The RPN loss is a sum over the SAMPLED anchors only (256 per image: balanced_positive_negative_sampler.py:27-67), so
// the gradient it sends into the head's outputs -- objectness and box deltas of all 2 x 268 569 anchors of a batch --
// is exactly zero everywhere except at <= images x 256 anchors.  Differentiating the head densely (what autograd does
// by default, and what this package did until round 4) runs the data and weight gradients of the shared 3x3 conv over
// every pixel of P2..P6 -- ~1.4 ms of MFMA kernels per step, 0.8 of it on P2 alone -- to multiply zeros.  Here the
// backward pass of the head is restricted to the sampled anchors, as rows of small dense matrices:
//
//   row r = sampled anchor (image n, level l, pixel (h, w), anchor a):
//     DT[r][c]   = [t > 0] * (dlogit * Wcls[a][c] + sum_j dbox[j] * Wbox[4a + j][c])      gradient at the 3x3 conv's output
//     Gc[r][a']  = dlogit if a' == a else 0;  Gb[r][k] likewise                           the predictors' output gradients
//     T[r][c]    = t[pixel][c]                                                             the predictors' input
//     X[r][(tap, c)] = feature[pixel + tap][c] (zero outside the map)                      the 3x3 conv's input patch
//
//   dWconv = DT^T X, dbconv = sum DT, dWcls = Gc^T T, dWbox = Gb^T T  (the package's own weight-gradient kernels on these
//   [P x ..] matrices), dX = DT Wconv (its data-gradient kernel) and dfeature[pixel + tap] += dX[r][(tap, .)] (here).
//
// Rows past the number of sampled anchors are all-zero, so every matrix has the CAPACITY images x 256 rows and nothing
// is read back to the host.  The result is the dense gradient with its zero terms left out (tests/test_gpu_rpn_sparse.py
// holds them together).
#include "common.h"

namespace {

constexpr int MAXL = 8;

struct RpnLevels {
  int n_levels, A, C, per_image;
  int H[MAXL], W[MAXL];
  int off[MAXL + 1];            // first anchor of the level inside an image's run (per_image = off[n_levels])
  const float* dlog[MAXL];      // [N][H][W][A]
  const float* dbox[MAXL];      // [N][H][W][4A]
  long long dlog_ns[MAXL], dbox_ns[MAXL];   // floats between two images of dlog / dbox (H * W * A | 4A when dense)
  const float* t[MAXL];         // [N][H][W][C]   relu(conv(feature))
  const float* feat[MAXL];      // [N][H][W][C]
  float* dfeat[MAXL];           // [N][H][W][C]   (scatter kernel)
};

// pos | neg -> ascending list of the flagged positions (the rows of the matrices above are then in a fixed order: sums
// over them are reproducible for a given sample).  Two launches over 4096-element blocks: per-block counts, then every
// block sums the counts in front of it and writes its flagged positions in order; block 0 also pads the list with -1.
// (A single-workgroup scan of the 537 138 anchors of a batch took 345 us on the forward pass's critical path.)
constexpr int MC_PER_THREAD = 16, MC_THREADS = 256, MC_BLOCK = MC_PER_THREAD * MC_THREADS;

__device__ __forceinline__ unsigned mc_flags(const uint8_t* __restrict__ pos, const uint8_t* __restrict__ neg, int64_t i0,
                                             int64_t total) {
  unsigned f = 0;
  if (i0 + MC_PER_THREAD <= total) {
    const uint4 a = *(const uint4*)(pos + i0), b = *(const uint4*)(neg + i0);
    const unsigned w[4] = {a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) f |= ((w[k >> 2] >> (8 * (k & 3))) & 0xFFu) ? (1u << k) : 0u;
  } else {
    for (int k = 0; k < MC_PER_THREAD; ++k)
      if (i0 + k < total && (pos[i0 + k] | neg[i0 + k])) f |= 1u << k;
  }
  return f;
}

__global__ __launch_bounds__(MC_THREADS) void mask_count_kernel(const uint8_t* __restrict__ pos,
                                                                const uint8_t* __restrict__ neg, int64_t total,
                                                                int* __restrict__ block_count) {
  __shared__ int ws[MC_THREADS / 64];
  const int tid = threadIdx.x;
  const int64_t i0 = ((int64_t)blockIdx.x * MC_THREADS + tid) * MC_PER_THREAD;
  int c = __popc(mc_flags(pos, neg, i0, total));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((tid & 63) == 0) ws[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) block_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(MC_THREADS) void mask_write_kernel(const uint8_t* __restrict__ pos,
                                                                const uint8_t* __restrict__ neg, int64_t total, int cap,
                                                                const int* __restrict__ block_count, int nblocks,
                                                                int* __restrict__ idx, int* __restrict__ count) {
  __shared__ int ws[MC_THREADS / 64];
  __shared__ int s_base, s_all;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // counts of the blocks in front of this one (and, for block 0, of all blocks)
  int before = 0, all = 0;
  for (int b = tid; b < nblocks; b += MC_THREADS) {
    const int v = block_count[b];
    all += v;
    if (b < (int)blockIdx.x) before += v;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { before += __shfl_down(before, d, 64); all += __shfl_down(all, d, 64); }
  if (lane == 0) ws[wave] = before;
  __syncthreads();
  if (tid == 0) s_base = ws[0] + ws[1] + ws[2] + ws[3];
  __syncthreads();
  if (lane == 0) ws[wave] = all;
  __syncthreads();
  if (tid == 0) s_all = ws[0] + ws[1] + ws[2] + ws[3];
  __syncthreads();
  const int64_t i0 = ((int64_t)blockIdx.x * MC_THREADS + tid) * MC_PER_THREAD;
  const unsigned f = mc_flags(pos, neg, i0, total);
  const int c = __popc(f);
  int incl = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __syncthreads();
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  int o = s_base + incl - c;
  for (int w = 0; w < wave; ++w) o += ws[w];
  for (int k = 0; k < MC_PER_THREAD; ++k)
    if (f & (1u << k)) {
      if (o < cap) idx[o] = (int)(i0 + k);
      ++o;
    }
  if (blockIdx.x == 0) {
    const int n = s_all;
    for (int q = n + tid; q < cap; q += MC_THREADS) idx[q] = -1;
    if (tid == 0) count[0] = n;           // (> cap: the caller sized the list too small; positions beyond it were dropped)
  }
}

__global__ __launch_bounds__(256) void rpn_rows_kernel(RpnLevels L, const int* __restrict__ idx, int n_img,
                                                       const float* __restrict__ wc, const float* __restrict__ wb,
                                                       float* __restrict__ DT, float* __restrict__ Gc, float* __restrict__ Gb,
                                                       float* __restrict__ T, float* __restrict__ X, int4* __restrict__ pix) {
  const int r = blockIdx.x, tid = threadIdx.x;
  const int C = L.C, A = L.A;
  const int id = idx[r];
  float* dt = DT + (size_t)r * C;
  float* tt = T + (size_t)r * C;
  float* xx = X + (size_t)r * 9 * C;
  if (id < 0 || id >= n_img * L.per_image) {            // a row past the sample: contributes nothing anywhere
    for (int c = tid; c < C; c += 256) { dt[c] = 0.f; tt[c] = 0.f; }
    for (int c = tid; c < 9 * C; c += 256) xx[c] = 0.f;
    for (int k = tid; k < A; k += 256) Gc[(size_t)r * A + k] = 0.f;
    for (int k = tid; k < 4 * A; k += 256) Gb[(size_t)r * 4 * A + k] = 0.f;
    if (tid == 0) pix[r] = make_int4(-1, 0, 0, 0);
    return;
  }
  const int n = id / L.per_image, rem = id - n * L.per_image;
  int l = 0;
  while (l + 1 < L.n_levels && rem >= L.off[l + 1]) ++l;
  const int q = rem - L.off[l];
  const int a = q % A, p = q / A;
  const int Wl = L.W[l], Hl = L.H[l];
  const int w = p % Wl, h = p / Wl;
  const size_t pb = ((size_t)n * Hl + h) * Wl + w;
  const size_t pin = (size_t)h * Wl + w;                // pixel inside its image
  const float dl = L.dlog[l][(size_t)n * L.dlog_ns[l] + pin * A + a];
  float db[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) db[j] = L.dbox[l][(size_t)n * L.dbox_ns[l] + pin * 4 * A + 4 * a + j];
  const float* tp = L.t[l] + pb * C;
  for (int c = tid; c < C; c += 256) {
    const float tv = tp[c];
    float v = dl * wc[(size_t)a * C + c];
#pragma unroll
    for (int j = 0; j < 4; ++j) v += db[j] * wb[(size_t)(4 * a + j) * C + c];
    dt[c] = tv > 0.f ? v : 0.f;
    tt[c] = tv;
  }
  for (int k = tid; k < A; k += 256) Gc[(size_t)r * A + k] = k == a ? dl : 0.f;
  for (int k = tid; k < 4 * A; k += 256) Gb[(size_t)r * 4 * A + k] = (k >> 2) == a ? db[k & 3] : 0.f;
  const float* fp = L.feat[l];
  for (int tap = 0; tap < 9; ++tap) {
    const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
    const bool in = (unsigned)hh < (unsigned)Hl && (unsigned)ww < (unsigned)Wl;
    const float* src = fp + (((size_t)n * Hl + hh) * Wl + ww) * C;
    for (int c = tid; c < C; c += 256) xx[tap * C + c] = in ? src[c] : 0.f;
  }
  if (tid == 0) pix[r] = make_int4(l, n, h, w);
}

// dfeature[pixel + tap][c] += dX[r][(tap, c)]: float atomics, 1 KB-contiguous per (row, tap) at C = 256
__global__ __launch_bounds__(256) void rpn_scatter_kernel(RpnLevels L, const int4* __restrict__ pix,
                                                          const float* __restrict__ dX) {
  const int r = blockIdx.x, tid = threadIdx.x;
  const int4 px = pix[r];
  if (px.x < 0) return;
  const int l = px.x, n = px.y, h = px.z, w = px.w, C = L.C;
  float* dst = L.dfeat[l];
  if (!dst) return;
  const int Hl = L.H[l], Wl = L.W[l];
  const float* src = dX + (size_t)r * 9 * C;
  for (int tap = 0; tap < 9; ++tap) {
    // the conv reads input pixel (h + dr, w + ds) with weight tap (dr + 1, ds + 1): that pixel gets the tap's product
    const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
    if ((unsigned)hh >= (unsigned)Hl || (unsigned)ww >= (unsigned)Wl) continue;
    float* d = dst + (((size_t)n * Hl + hh) * Wl + ww) * C;
    for (int c = tid; c < C; c += 256) atomicAdd(d + c, src[tap * C + c]);
  }
}

}  // namespace

CPM_EXPORT int cpm_mask_compact(const uint8_t* pos, const uint8_t* neg, int64_t total, int cap, int32_t* idx,
                                int32_t* count, int32_t* workspace, void* stream) {
  CPM_REQUIRE(total >= 0 && cap >= 1 && total < (1ll << 31), "bad sizes");
  CPM_REQUIRE(pos && neg && idx && count && workspace, "null pointer");
  CPM_REQUIRE((((uintptr_t)pos | (uintptr_t)neg) & 15) == 0, "masks must be 16-byte aligned");
  const int nblocks = (int)((total + MC_BLOCK - 1) / MC_BLOCK) > 0 ? (int)((total + MC_BLOCK - 1) / MC_BLOCK) : 1;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)nblocks), dim3(MC_THREADS), 0, s, pos, neg, total, workspace);
  hipLaunchKernelGGL(mask_write_kernel, dim3((unsigned)nblocks), dim3(MC_THREADS), 0, s, pos, neg, total, cap, workspace,
                     nblocks, idx, count);
  return cpm::check_launch("mask_compact");
}

static int fill_levels(RpnLevels& L, int n_levels, const int* hs, const int* ws, int A, int C) {
  if (n_levels < 1 || n_levels > MAXL || A < 1 || C < 1) return CPM_EINVAL;
  L.n_levels = n_levels; L.A = A; L.C = C;
  int off = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (hs[l] < 1 || ws[l] < 1) return CPM_EINVAL;
    L.H[l] = hs[l]; L.W[l] = ws[l]; L.off[l] = off;
    off += hs[l] * ws[l] * A;
  }
  L.off[n_levels] = off;
  L.per_image = off;
  return CPM_OK;
}

CPM_EXPORT int cpm_rpn_sparse_rows(const int32_t* idx, int cap, int n_img, int n_levels, const int* hs, const int* ws, int A,
                                   int C, const float* const* dlog, const float* const* dbox, const float* const* t,
                                   const float* const* feat, const float* w_cls, const float* w_box, float* DT, float* Gc,
                                   float* Gb, float* T, float* X, int32_t* pix4, const int64_t* dlog_image_stride,
                                   const int64_t* dbox_image_stride, void* stream) {
  CPM_REQUIRE(idx && dlog && dbox && t && feat && w_cls && w_box && DT && Gc && Gb && T && X && pix4, "null pointer");
  CPM_REQUIRE(cap >= 1 && n_img >= 1, "bad sizes");
  RpnLevels L = {};
  CPM_REQUIRE(fill_levels(L, n_levels, hs, ws, A, C) == CPM_OK, "bad level table (1..8 levels)");
  CPM_REQUIRE((int64_t)n_img * L.per_image < (1ll << 31), "too many anchors");
  for (int l = 0; l < n_levels; ++l) {
    CPM_REQUIRE(dlog[l] && dbox[l] && t[l] && feat[l], "null level pointer");
    L.dlog[l] = dlog[l]; L.dbox[l] = dbox[l]; L.t[l] = t[l]; L.feat[l] = feat[l];
    L.dlog_ns[l] = dlog_image_stride ? dlog_image_stride[l] : (long long)hs[l] * ws[l] * A;
    L.dbox_ns[l] = dbox_image_stride ? dbox_image_stride[l] : (long long)hs[l] * ws[l] * 4 * A;
    CPM_REQUIRE(L.dlog_ns[l] >= (long long)hs[l] * ws[l] * A && L.dbox_ns[l] >= (long long)hs[l] * ws[l] * 4 * A,
                "image stride below the image's size");
  }
  hipLaunchKernelGGL(rpn_rows_kernel, dim3((unsigned)cap), dim3(256), 0, (hipStream_t)stream, L, idx, n_img, w_cls, w_box,
                     DT, Gc, Gb, T, X, (int4*)pix4);
  return cpm::check_launch("rpn_sparse_rows");
}

CPM_EXPORT int cpm_rpn_sparse_scatter(const int32_t* pix4, int cap, int n_levels, const int* hs, const int* ws, int C,
                                      const float* dX, float* const* dfeat, void* stream) {
  CPM_REQUIRE(pix4 && dX && dfeat && cap >= 1, "null pointer / bad size");
  RpnLevels L = {};
  CPM_REQUIRE(fill_levels(L, n_levels, hs, ws, 1, C) == CPM_OK, "bad level table (1..8 levels)");
  for (int l = 0; l < n_levels; ++l) L.dfeat[l] = dfeat[l];          // a null level takes no gradient
  hipLaunchKernelGGL(rpn_scatter_kernel, dim3((unsigned)cap), dim3(256), 0, (hipStream_t)stream, L, (const int4*)pix4, dX);
  return cpm::check_launch("rpn_sparse_scatter");
}
