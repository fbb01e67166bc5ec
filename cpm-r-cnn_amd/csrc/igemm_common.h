// Pieces shared by the implicit-GEMM convolution kernels (conv_igemm.hip: fp32 operands split in flight, the weight
// side optionally given as a pre-split image).
#pragma once
#include "common.h"

namespace cpmconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;
constexpr int LDP = 36;  // LDS row pitch in floats

// 3-term split-bf16 arithmetic (SURVEY "numerics": fp32-accumulate with a split-bf16 scheme): x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi) (x - hi is exact in fp32), and a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The dropped terms are <= 2^-16 |a b| (measured <= 1e-5 relative
// on the conv outputs, the parity bar is 1e-3), bf16 keeps fp32's exponent range so no scaling is involved, and
// three bf16 MFMAs cost 3/16 of the fp32 MFMA they replace.
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  // one v_cvt_pk_bf16_f32 (round to nearest even), pinned: written as casts the compiler re-derives the hi values
  // through extra single-element conversions (6 instead of 4 per float4)
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  hi.x = cvt_pk_bf16(v.x, v.y);
  hi.y = cvt_pk_bf16(v.z, v.w);
  const float r0 = v.x - __uint_as_float(hi.x << 16), r1 = v.y - __uint_as_float(hi.x & 0xFFFF0000u);
  const float r2 = v.z - __uint_as_float(hi.y << 16), r3 = v.w - __uint_as_float(hi.y & 0xFFFF0000u);
  lo.x = cvt_pk_bf16(r0, r1);
  lo.y = cvt_pk_bf16(r2, r3);
}

// 16-byte load through a buffer descriptor: an offset at or beyond num_records returns zeros WITHOUT touching
// memory.  Masked gather lanes (padding, row/channel tails) are simply given OOB_OFF: no branch around the load
// (hipcc drains vmcnt(0) inside such branches), no select, and no hot cache line as with a dummy address.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB_OFF = 0xFFFFFFF0u;
// base offset of a weight row that does not exist: weights stay below 2 GiB (validate()), so base + tap offset
// is still beyond num_records and the load returns zeros
constexpr unsigned B_INVALID = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// One gather description serves both directions.  GEMM row m = (n, i, j) on a (OHp x OWp) row grid:
//   input tap  : ih = i*ihmul + ihadd + t_r*hstep,  iw = j*iwmul + iwadd + t_s*wstep   (zero outside the image)
//   weight tap : r = r0 + t_r*rstep,                s = s0 + t_s*sstep,  t_r < nr, t_s < ns
//   output pos : oh = i*osh + oah,                  ow = j*osw + oaw     on the full (OH x OW) output
// forward conv   : ihmul = stride, ihadd = -pad, hstep = dil, r0 = 0, rstep = 1, nr = R, osh = 1, oah = 0
// data gradient  : one launch per output phase (a, b) in stride x stride: rows are the outputs with
//                  oh = i*stride + a; only the taps r = r0 + t*stride with r0 = (a + pad) % stride reach them, and
//                  th = oh + pad - r = stride * (i + (a + pad - r0)/stride - t)  =>  ihmul = 1, hstep = -1.
//                  No masked taps and no div/mod in the loop (a stride-2 3x3 gradient does 9/4 of the taps per
//                  row instead of 9, a 1x1 stride-2 one runs a quarter of the rows).
struct IgemmArgs {
  const float* in;     // [N][IH][IW][Ctot]
  const float* wm;     // [OCtot][R][S][CgR]
  float* out;          // [N][OH][OW][OCtot]
  const float* scale;  // [OCtot] or null
  const float* shift;  // [OCtot] or null
  const float* res;    // residual or null
  const float* mask;   // [rows][OCtot] or null: output kept only where mask > 0 (ReLU gate of the consumer-side fusion)
  int N, IH, IW, Ctot;
  int OH, OW, OCtot;   // full output grid
  int OHp, OWp;        // row grid of this launch
  int R, S;            // weight window (for the B operand pitch)
  int ihmul, ihadd, hstep, iwmul, iwadd, wstep;
  int r0, rstep, nr, s0, sstep, ns;
  int osh, oah, osw, oaw;
  int groups, CgR, OCg;
  int M;               // N*OHp*OWp
  int ksteps_per_tap;  // ceil(CgR/32)
  int ksteps;          // nr*ns*ksteps_per_tap
  int split_k;         // >= 1
  int res_mode, relu;
  int atomic_out;      // 1: atomicAdd raw accumulators (split-K / accumulate)
  unsigned in_bytes, wm_bytes;
  int xcd_swizzle;
  int m_base;          // first GEMM row of this launch (rows [m_base, M) are tiled)
  float* slab;         // split-K: partial sums go to plane `split` of this [split_k][rows][OCtot] buffer with plain stores
  size_t slab_stride;  //   (floats per plane) and splitk_reduce_kernel folds the planes in split order + runs the epilogue;
                       //   null: float atomics into the zeroed output (no workspace)
  int staged_epi;      // finish every non-atomic epilogue row-wise through LDS (16-byte accesses), not only residual ones
  int b_presplit;      // wm is a pre-split weight image (cpm_split_w4): bf16x3 arithmetic, vector path only
  int tail;            // reduction channels % 4 != 0 on the vector path: igemm_kernel<.., TAIL> masks the rows' last loads
  int dbg;             // timing-only experiments (CPM_IGEMM_DBG): 8 zero-record descriptors (nothing fetched), 16 no epilogue
};

struct Plan { int bm, bn, wm, wn, split; };

}  // namespace cpmconv
