// ResNet / ResNeXt stem: y = relu?( conv7x7 / stride 2 / pad 3 (x) * scale + shift ), 3 -> 64 channels, as ONE kernel.
//
// Reference: pet/models/imagenet/resnet.py:175-181 (conv1 + bn1 + relu; the max-pool stays cpm_maxpool3x3s2_forward).
// The reduction of this layer is 7 x 7 x 3 = 147 long with 3 channels per tap: the implicit-GEMM kernels step over taps
// in 32-channel blocks (29 of 32 lanes idle), so the stem used to run as cpm_im2col (344 MB of columns written and
// read back for 2 x 800 x 1344, 222 us) + a 1x1 GEMM over them (136 us).  Here a workgroup owns 96 output pixels of
// one output row:
//   * the 7 input rows x 197 input pixels x 3 channels its taps reach (16.5 KB of the NHWC image, contiguous per row)
//     go to LDS once, split into bf16 hi / lo (bf16x3 arithmetic, see conv_igemm.hip);
//   * along a filter row the 21 values (7 taps x 3 channels) of output pixel m are the contiguous run [6 m, 6 m + 21)
//     of that LDS row, so the MFMA A operand (8 consecutive reduction elements of a pixel) is read straight out of the
//     patch -- no column image at all.  The reduction is ordered (filter row r, 32 slots: 21 taps + 11 zero weights);
//     the slots behind a run read the next pixels' values, which meet zero weights;
//   * the weights (64 x 7 x 32 slots, hi / lo) sit in LDS for the lifetime of the workgroup, which loops over tiles;
//   * three waves, each 32 pixels x 64 output channels: 14 k-halves x 6 MFMAs per tile (six waves of 32 x 32 each:
//     155 vs 140 us for 2 x 800 x 1344 incl. the max-pool).
// bf16x3 only (the exact-f32 arithmetic keeps the im2col path).
#include "common.h"
#include "igemm_common.h"
#include "../../include/cpmrcnn_hip.h"

using namespace cpmconv;

namespace {

constexpr int BM = 96;            // output pixels per tile (one output row segment)
constexpr int PR = 608;           // patch row pitch in bf16 elements (>= 6 * 95 + 32, multiple of 8)
constexpr int PVALID = (2 * BM + 5) * 3;   // 591 image values per patch row
constexpr int WR = 232;           // weight row pitch in bf16 elements (464 B: ds_read_b128 of 16 consecutive rows is conflict-free)
constexpr int KH = 14;            // k-halves (16 reduction slots each): 7 filter rows x 32 slots

__device__ __forceinline__ unsigned short bf16_rne(float v) {
  return (unsigned short)(cvt_pk_bf16(v, 0.f) & 0xFFFFu);
}

__global__ __launch_bounds__(192) void stem7x7_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      int relu, int N, int H, int W, int P, int Q, float* __restrict__ y,
                                                      int tiles_per_row, int total_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned short p_hi[7 * PR];
  __shared__ __attribute__((aligned(16))) unsigned short p_lo[7 * PR];
  __shared__ __attribute__((aligned(16))) unsigned short w_hi[64 * WR];
  __shared__ __attribute__((aligned(16))) unsigned short w_lo[64 * WR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // weights: [oc][r][slot], slot = s * 3 + c for the 21 taps of a filter row, zero behind them
  for (int i = tid; i < 64 * 7 * 32; i += 192) {
    const int oc = i / 224, rem = i - oc * 224, r = rem >> 5, t = rem & 31;
    const float v = t < 21 ? w[oc * 147 + r * 21 + t] : 0.f;
    const unsigned short hi = bf16_rne(v);
    const unsigned short lo = bf16_rne(v - __uint_as_float((unsigned)hi << 16));
    w_hi[oc * WR + rem] = hi;
    w_lo[oc * WR + rem] = lo;
  }
  for (int i = tid; i < 7 * PR; i += 192) { p_hi[i] = 0; p_lo[i] = 0; }      // the slots behind the image values stay zero
  __syncthreads();

  const int m_l = lane & 31, kh = lane >> 5;
  const int m = 32 * wave + m_l;                                   // this lane's pixel of the tile (A operand row)
  float e_sc[2], e_sh[2];
#pragma unroll
  for (int jn = 0; jn < 2; ++jn) {
    e_sc[jn] = scale ? scale[32 * jn + m_l] : 1.f;
    e_sh[jn] = shift ? shift[32 * jn + m_l] : 0.f;
  }

  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int tx = tile % tiles_per_row, row = tile / tiles_per_row;
    const int oh = row % P, n = row / P;
    const int ow0 = tx * BM;
    const int iw0 = 2 * ow0 - 3;                                   // first input pixel of the patch
    // values [e_lo, e_hi) of a patch row lie inside the image
    const int e_lo = iw0 < 0 ? -iw0 * 3 : 0;
    const int e_hi = min(W - iw0, 2 * BM + 5) * 3;
    // ---- the patch: 7 rows x 296 pairs of values; all loads of a tile are issued before the first is used
    float pv[7][2][2];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int ih = 2 * oh - 3 + r;
      const bool row_ok = (unsigned)ih < (unsigned)H;
      const float* src = x + ((int64_t)(n * H + (row_ok ? ih : 0)) * W + iw0) * 3;   // (iw0 may be -3: only masked reads there)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int pe = 2 * tid + 384 * k2;
        pv[r][k2][0] = (row_ok && pe >= e_lo && pe < e_hi) ? src[pe] : 0.f;
        pv[r][k2][1] = (row_ok && pe + 1 >= e_lo && pe + 1 < e_hi) ? src[pe + 1] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int pe = 2 * tid + 384 * k2;
        if (pe > PVALID) continue;
        const float v0 = pv[r][k2][0], v1 = pv[r][k2][1];
        const unsigned hi = cvt_pk_bf16(v0, v1);
        const unsigned lo = cvt_pk_bf16(v0 - __uint_as_float(hi << 16), v1 - __uint_as_float(hi & 0xFFFF0000u));
        *(unsigned*)&p_hi[r * PR + pe] = hi;
        *(unsigned*)&p_lo[r * PR + pe] = lo;
      }
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[jn][e] = 0.f;
#pragma unroll
    for (int j = 0; j < KH; ++j) {
      const int a_idx = (j >> 1) * PR + 6 * m + (j & 1) * 16 + 8 * kh;            // even: 4-byte aligned dwords
      const unsigned* ah_p = (const unsigned*)&p_hi[a_idx];
      const unsigned* al_p = (const unsigned*)&p_lo[a_idx];
      const u32x4 ahv = {ah_p[0], ah_p[1], ah_p[2], ah_p[3]};
      const u32x4 alv = {al_p[0], al_p[1], al_p[2], al_p[3]};
      const bf16x8 ah = __builtin_bit_cast(bf16x8, ahv), al = __builtin_bit_cast(bf16x8, alv);
#pragma unroll
      for (int jn = 0; jn < 2; ++jn) {
        const int b_idx = (32 * jn + m_l) * WR + j * 16 + 8 * kh;
        const bf16x8 bh = __builtin_bit_cast(bf16x8, *(const uint4*)&w_hi[b_idx]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, *(const uint4*)&w_lo[b_idx]);
        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[jn], 0, 0, 0);
        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[jn], 0, 0, 0);
        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[jn], 0, 0, 0);
      }
    }
    // C/D map of the 32x32 MFMA: col = lane & 31 (output channel), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (pixel)
    float* const yrow = y + ((size_t)(n * P + oh) * Q) * 64;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ow = ow0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * kh;
      if (ow >= Q) continue;
#pragma unroll
      for (int jn = 0; jn < 2; ++jn) {
        float v = acc[jn][e] * e_sc[jn] + e_sh[jn];
        if (relu) v = fmaxf(v, 0.f);
        yrow[(size_t)ow * 64 + 32 * jn + m_l] = v;
      }
    }
    __syncthreads();                                               // the patch is free for the next tile
  }
}

}  // namespace

CPM_EXPORT int cpm_stem7x7_forward(const float* x, const float* w, const float* scale, const float* shift, int relu,
                                   int N, int H, int W, float* y, void* stream) {
  CPM_REQUIRE(x && w && y, "null pointer");
  CPM_REQUIRE(N > 0 && H >= 7 && W >= 7, "bad shape");
  CPM_REQUIRE(cpm_get_conv_math() == CPM_MATH_BF16X3, "cpm_stem7x7_forward serves the bf16x3 arithmetic (cpm_set_conv_math)");
  const int P = (H + 6 - 7) / 2 + 1, Q = (W + 6 - 7) / 2 + 1;
  CPM_REQUIRE((int64_t)N * H * W * 3 < (1ll << 31) && (int64_t)N * P * Q * 64 < (1ll << 33), "image batch too large");
  const int tiles_per_row = cpm::cdiv(Q, BM);
  const int64_t total = (int64_t)N * P * tiles_per_row;
  CPM_REQUIRE(total < (1ll << 31), "too many tiles");
  const int grid = (int)(total < 512 ? total : 512);              // two workgroups per CU, each looping over its tiles
  hipLaunchKernelGGL(stem7x7_kernel, dim3(grid), dim3(192), 0, (hipStream_t)stream, x, w, scale, shift, relu, N, H, W, P, Q,
                     y, tiles_per_row, (int)total);
  return cpm::check_launch("stem7x7_forward");
}
