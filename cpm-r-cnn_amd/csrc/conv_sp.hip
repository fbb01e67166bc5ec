// Implicit-GEMM convolution over SPLIT-PLANE operands: 3-term split-bf16 arithmetic without any conversion in the loop.
//
// conv_igemm.hip splits every fp32 operand into hi = bf16(x), lo = bf16(x - hi) while it moves from registers to LDS:
// per 128x128x32 k-step a wave issues ~200 VALU instructions and 16 LDS stores next to its 24 MFMAs, and with two
// waves per SIMD those phases do not overlap (profiles/round1_kernel_stats.md: MFMA pipe ~30 % busy).  Here the
// operands already LIE in memory as bf16 hi / lo planes -- written once by whoever produced them (the producing
// convolution's epilogue, or cpm_split_planes) -- so a k-step is:
//
//   8 x buffer_load_dwordx4 ... lds per wave   global -> LDS directly (no VGPRs, no VALU, no ds_write), masked gather
//                                               lanes get an out-of-range offset and the DMA writes zeros
//   16 x ds_read_b128 per wave                 conflict-free through an XOR swizzle applied on the SOURCE side (the
//                                               DMA writes lane-linear: base + 16 * lane)
//   24 x v_mfma_f32_32x32x16_bf16 per wave     a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulation
//
// Split-plane format ("SP") of a [rows][C] fp32 matrix (rows = pixels of an NHWC activation, or (oc, tap) rows of a
// KRSC weight): [rows][2][C] bf16 -- per row C hi values, then C lo values: the same 4*C bytes as the fp32 row.
//
// Two LDS buffers of 4 planes (A_hi, A_lo, B_hi, B_lo; rows of 32 bf16 = 64 B); the DMA of k-step t+1 is issued
// before the MFMAs of k-step t and is waited for (s_waitcnt vmcnt(0) + s_barrier) after them; two workgroups per CU.
// Same gather description (IgemmArgs), tile order, split-K and fused epilogue as conv_igemm.hip; the epilogue can
// write the result a second time as SP for the next convolution.
#include "common.h"
#include "igemm_common.h"

using namespace cpmconv;

namespace {

constexpr unsigned OOB_V = 0x80000000u;      // voffset of a masked lane: beyond every tensor (all are < 2 GiB here)
typedef __attribute__((address_space(3))) void* lds_void_p;

// 16 bytes per lane, global -> LDS at (wave-uniform lds) + 16 * lane
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_p)lds, 16, (int)voff, (int)soff, 0, 0);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2))) void igemm_sp_kernel(IgemmArgs a) {
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int AR = BM / NW, BR = BN / NW;          // rows a wave stages per plane
  constexpr int AI = AR / 16, BI = BR / 16;          // DMA instructions per wave and plane (16 rows of 64 B each)
  static_assert(AR % 16 == 0 && BR % 16 == 0 && WTM % 32 == 0 && WTN % 32 == 0, "tile shape");
  constexpr int PA_HI = 0, PA_LO = BM * 16, PB_HI = 2 * BM * 16, PB_LO = 2 * BM * 16 + BN * 16;   // dwords in a stage
  constexpr int STAGE = (2 * BM + 2 * BN) * 16;
  constexpr int CP = BN + 4;
  constexpr int CROWS = 32 * WM;                     // the epilogue stages one 32-row slab of every wave at a time
  constexpr int LDS_DW = 2 * STAGE > CROWS * CP ? 2 * STAGE : CROWS * CP;
  __shared__ __attribute__((aligned(16))) unsigned sm[LDS_DW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.OCg + BN - 1) / BN;
  int bid = blockIdx.x;
  if (a.xcd_swizzle) {                                // see igemm_kernel: contiguous logical tiles per XCD
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int g = blockIdx.y, split = blockIdx.z;
  const int m0 = a.m_base + tile_m * BM, n0 = tile_n * BN;

  // ---- DMA geometry: lane -> (row, 16-byte slot); the slot holds source chunk slot ^ ((row >> 2) & 3) ----------
  const int drow = lane >> 2, dslot = lane & 3;
  const unsigned in_pitch = (unsigned)a.Ctot * 4u, w_pitch = (unsigned)a.CgR * 4u;       // bytes per SP row
  unsigned a_off[AI];
  int a_h[AI], a_w[AI], a_cmax[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = wave * AR + i * 16 + drow;
    const int chunk = dslot ^ ((row >> 2) & 3);
    const int m = m0 + row;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int jj = mm % a.OWp, t = mm / a.OWp;
    const int ii = t % a.OHp, n = t / a.OHp;
    a_h[i] = ok ? ii * a.ihmul + a.ihadd : -(1 << 28);
    a_w[i] = jj * a.iwmul + a.iwadd;
    a_off[i] = (unsigned)((n * a.IH + (ok ? a_h[i] : 0)) * a.IW + a_w[i]) * in_pitch + (unsigned)(g * a.CgR + chunk * 8) * 2u;
    a_cmax[i] = a.CgR - chunk * 8;                    // the chunk holds channels of this group while cb < a_cmax
    asm volatile("" : "+v"(a_off[i]), "+v"(a_h[i]), "+v"(a_w[i]), "+v"(a_cmax[i]));
  }
  unsigned b_off[BI];
  int b_cmax[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = wave * BR + i * 16 + drow;
    const int chunk = dslot ^ ((row >> 2) & 3);
    const int oc = n0 + row;
    b_off[i] = oc < a.OCg ? (unsigned)((g * a.OCg + oc) * a.R * a.S) * w_pitch + (unsigned)(chunk * 8) * 2u : OOB_V;
    b_cmax[i] = oc < a.OCg ? a.CgR - chunk * 8 : -(1 << 28);
    asm volatile("" : "+v"(b_off[i]), "+v"(b_cmax[i]));
  }
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc((const float*)a.in_sp, a.in_bytes),
                               rs_wm = make_rsrc((const float*)a.wm_sp, a.wm_bytes);
  const unsigned a_lo = (unsigned)a.Ctot * 2u, b_lo = (unsigned)a.CgR * 2u;   // lo plane of a row (scalar offset)

  const int per = (a.ksteps + a.split_k - 1) / a.split_k;
  const int k_begin = split * per;
  const int k_end = min(a.ksteps, k_begin + per);
  const int nk = k_end - k_begin;

  // position of the NEXT k-step to stage: (tap row, tap column, channel block), advanced without divisions
  int s_tr, s_ts, s_cb;
  {
    const int tap = k_begin / a.ksteps_per_tap;
    s_cb = (k_begin - tap * a.ksteps_per_tap) * BK;
    s_tr = tap / a.ns;
    s_ts = tap - s_tr * a.ns;
  }
  auto stage_tile = [&](int buf) {
    const int dh = s_tr * a.hstep, dw = s_ts * a.wstep;
    const unsigned wtap = (unsigned)((a.r0 + s_tr * a.rstep) * a.S + a.s0 + s_ts * a.sstep) * w_pitch + (unsigned)s_cb * 2u;
    const unsigned aoff = (unsigned)(dh * a.IW + dw) * in_pitch + (unsigned)s_cb * 2u;     // wraps for negative taps
    unsigned* const base = sm + buf * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const bool ok = (s_cb < a_cmax[i]) & ((unsigned)(a_h[i] + dh) < (unsigned)a.IH) &
                      ((unsigned)(a_w[i] + dw) < (unsigned)a.IW);
      const unsigned vo = ok ? a_off[i] + aoff : OOB_V;
      unsigned* const dst = base + (wave * AR + i * 16) * 16;
      dma16(rs_in, dst + PA_HI, vo, 0);
      dma16(rs_in, dst + PA_LO, vo, a_lo);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const unsigned vo = s_cb < b_cmax[i] ? b_off[i] + wtap : OOB_V;
      unsigned* const dst = base + (wave * BR + i * 16) * 16;
      dma16(rs_wm, dst + PB_HI, vo, 0);
      dma16(rs_wm, dst + PB_LO, vo, b_lo);
    }
    s_cb += BK;
    if (s_cb >= a.ksteps_per_tap * BK) {
      s_cb = 0;
      if (++s_ts == a.ns) { s_ts = 0; ++s_tr; }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31, khalf = lane >> 5;
  struct Frag { bf16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
  auto fetch = [&](int buf, int sub, Frag& f) {
    const int r_sw = (((sub * 2 + khalf) ^ ((frow >> 2) & 3)) << 2);
    const unsigned* const base = sm + buf * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int o = (wm * WTM + i * 32 + frow) * 16 + r_sw;
      f.ah[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(base + PA_HI + o));
      f.al[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(base + PA_LO + o));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int o = (wn * WTN + j * 32 + frow) * 16 + r_sw;
      f.bh[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(base + PB_HI + o));
      f.bl[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(base + PB_LO + o));
    }
  };
  auto mfma3 = [&](const Frag& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
      }
  };
  auto step = [&](int it, int cur) {
    if (it + 1 < nk) stage_tile(cur ^ 1);             // lands under this step's MFMAs
    Frag f0, f1;
    fetch(cur, 0, f0);
    fetch(cur, 1, f1);
    mfma3(f0);
    mfma3(f1);
    __syncthreads();                                  // s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier
  };

  if (nk > 0) stage_tile(0);
  __syncthreads();
  for (int it = 0; it < nk; it += 2) {
    step(it, 0);
    if (it + 1 < nk) step(it + 1, 1);
  }

  // ---- epilogue ------------------------------------------------------------------------------------------
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  const bool dense_rows = a.osh == 1 && a.osw == 1 && a.OHp == a.OH && a.OWp == a.OW;
  if (a.atomic_out) {                                 // split-K / accumulate: raw sums, the epilogue runs as a pass
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0;
        if (m >= a.M) continue;
        int orow = m;
        if (!dense_rows) {
          const int jj = m % a.OWp, t = m / a.OWp;
          orow = ((t / a.OHp) * a.OH + (t % a.OHp) * a.osh + a.oah) * a.OW + jj * a.osw + a.oaw;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int ocl = n0 + wn * WTN + j * 32 + ecol;
          if (ocl >= a.OCg) continue;
          const size_t o = (size_t)orow * a.OCtot + g * a.OCg + ocl;
          if (a.slab) a.slab[(size_t)split * a.slab_stride + o] = acc[i][j][e];
          else atomicAdd(a.out + o, acc[i][j][e]);
        }
      }
    return;
  }
  // Stage the tile through LDS (free after the loop's last barrier), one 32-row slab of every wave at a time, and
  // finish it row-wise with 16-byte accesses: residual / gate reads and the fp32 + SP stores are whole rows instead of
  // 4-byte column slices.  Slab row wm * 32 + r is tile row wm * WTM + i * 32 + r.
  float (*Cs)[CP] = reinterpret_cast<float (*)[CP]>(sm);
  constexpr int CV = BN / 4, RPS = NT / CV;
  static_assert(NT % CV == 0 && CROWS % RPS == 0, "epilogue layout");
  const int cv = (tid % CV) * 4, r0 = tid / CV;
  const int ocl = n0 + cv;
  const bool col_ok = ocl < a.OCg;
  const bool vec_out = (a.OCg & 3) == 0 && (a.OCtot & 3) == 0 && ocl + 3 < a.OCg;
  const int oc = g * a.OCg + ocl;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok) {
    float* scp = &sc.x; float* shp = &sh.x;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ocl + k < a.OCg) {
        if (a.scale) scp[k] = a.scale[oc + k];
        if (a.shift) shp[k] = a.shift[oc + k];
      }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i) __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        Cs[wm * 32 + (e & 3) + 8 * (e >> 2) + erow0][wn * WTN + j * 32 + ecol] = acc[i][j][e];
    __syncthreads();
    if (!col_ok) continue;
    for (int r = r0; r < CROWS; r += RPS) {
      const int m = m0 + (r >> 5) * WTM + i * 32 + (r & 31);
      if (m >= a.M) continue;
      int orow = m, oh = 0, ow = 0, n = 0;
      if (!dense_rows || (a.res && a.res_mode == 1)) {
        const int jj = m % a.OWp, t = m / a.OWp;
        const int ii = t % a.OHp;
        n = t / a.OHp;
        oh = ii * a.osh + a.oah;
        ow = jj * a.osw + a.oaw;
        orow = (n * a.OH + oh) * a.OW + ow;
      }
      float4 v = *(const float4*)&Cs[r][cv];
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      const size_t o = (size_t)orow * a.OCtot + oc;
      const float* rp = !a.res ? nullptr
                        : a.res_mode == 0
                            ? a.res + o
                            : a.res + ((size_t)(n * ((a.OH + 1) / 2) + oh / 2) * ((a.OW + 1) / 2) + ow / 2) * a.OCtot + oc;
      if (vec_out) {
        if (rp) {
          const float4 rv = *(const float4*)rp;
          v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.mask) {
          const float4 gt = *(const float4*)(a.mask + o);
          v.x = gt.x > 0.f ? v.x : 0.f; v.y = gt.y > 0.f ? v.y : 0.f;
          v.z = gt.z > 0.f ? v.z : 0.f; v.w = gt.w > 0.f ? v.w : 0.f;
        }
        *(float4*)(a.out + o) = v;
        if (a.out_sp) {
          uint2 hi, lo;
          split4(v, hi, lo);
          char* const row = (char*)a.out_sp + (size_t)orow * a.OCtot * 4;
          *(uint2*)(row + (size_t)oc * 2) = hi;
          *(uint2*)(row + (size_t)(a.OCtot + oc) * 2) = lo;
        }
      } else {
        float* vp = &v.x;
        for (int k = 0; k < 4 && ocl + k < a.OCg; ++k) {
          float t = vp[k] + (rp ? rp[k] : 0.f);
          if (a.relu) t = fmaxf(t, 0.f);
          if (a.mask) t = a.mask[o + k] > 0.f ? t : 0.f;
          a.out[o + k] = t;
          if (a.out_sp) {
            __bf16* const row = (__bf16*)((char*)a.out_sp + (size_t)orow * a.OCtot * 4);
            const __bf16 h = (__bf16)t;
            row[oc + k] = h;
            row[a.OCtot + oc + k] = (__bf16)(t - (float)h);
          }
        }
      }
    }
  }
}

// fp32 [rows][C] -> SP [rows][2][C]
__global__ __launch_bounds__(256) void split_planes_kernel(const float4* __restrict__ x, int64_t total4, int c4,
                                                           uint2* __restrict__ sp) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / c4;
    const int c = (int)(i - row * c4);
    uint2 hi, lo;
    split4(x[i], hi, lo);
    sp[row * 2 * c4 + c] = hi;
    sp[(row * 2 + 1) * c4 + c] = lo;
  }
}

}  // namespace

namespace cpmconv {

bool sp_eligible(const IgemmArgs& a) {
  static const int on = [] { const char* v = getenv("CPM_CONV_SP"); return v ? atoi(v) : 1; }();
  return on && a.in_sp && a.wm_sp && (a.CgR % 8 == 0) && (a.Ctot % 8 == 0) && a.OCg > 32 &&
         a.in_bytes < OOB_V && a.wm_bytes < OOB_V && (((uintptr_t)a.in_sp | (uintptr_t)a.wm_sp) & 15) == 0 &&
         (!a.out_sp || ((a.OCtot & 3) == 0 && (a.OCg & 3) == 0 && ((uintptr_t)a.out_sp & 15) == 0));
}

int launch_igemm_sp(const IgemmArgs& a, int bm, int bn, hipStream_t s) {
  const int rows = a.M - a.m_base;
#define LAUNCH_SP(BM, BN, WM, WN)                                                             \
  do {                                                                                        \
    dim3 grid((unsigned)(cpm::cdiv(rows, BM) * cpm::cdiv(a.OCg, BN)), a.groups, a.split_k);   \
    hipLaunchKernelGGL((igemm_sp_kernel<BM, BN, WM, WN>), grid, dim3(64 * WM * WN), 0, s, a); \
  } while (0)
  static const int big = [] { const char* v = getenv("CPM_SP_BIG"); return v ? atoi(v) : 0; }();
  if (bm == 128 && bn == 128 && big == 1) LAUNCH_SP(256, 128, 4, 2);
  else if (bm == 128 && bn == 128 && big == 2) LAUNCH_SP(256, 256, 2, 4);
  else if (bm == 128 && bn == 128 && big == 3) LAUNCH_SP(256, 256, 4, 2);
  else if (bm == 128 && bn == 128) LAUNCH_SP(128, 128, 2, 2);
  else if (bm == 128 && bn == 64) LAUNCH_SP(128, 64, 2, 2);
  else LAUNCH_SP(64, 64, 2, 2);
#undef LAUNCH_SP
  return cpm::check_launch("conv igemm (split planes)");
}

}  // namespace cpmconv

CPM_EXPORT int cpm_split_planes(const float* x, int64_t rows, int channels, void* sp, void* stream) {
  CPM_REQUIRE(rows >= 0 && channels > 0 && channels % 4 == 0, "channels must be a positive multiple of 4");
  if (rows == 0) return CPM_OK;
  CPM_REQUIRE(x && sp, "null pointer");
  CPM_REQUIRE((((uintptr_t)x | (uintptr_t)sp) & 15) == 0, "pointers must be 16-byte aligned");
  const int64_t total4 = rows * (channels / 4);
  const int64_t b = (total4 + 255) / 256;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)(b > 16384 ? 16384 : b)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)x, total4, channels / 4, (uint2*)sp);
  return cpm::check_launch("split_planes");
}
