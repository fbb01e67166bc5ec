// Implicit-GEMM convolution, bf16x3 arithmetic, with the two halves of a k-step on DIFFERENT waves.
//
// What igemm_kernel's k-step costs a wave (profiles/round5_fb_ablation.txt): its 24 MFMAs are 768
// cycles of the matrix pipe, but the wave's own in-order stream also carries 8 buffer_load_dwordx4 (each ~50-70 cycles
// of issue), the bf16 split of the tile it stages (~100 VALU), 16 ds_write_b64 and the waits in between -- ~2 900 cycles
// per k-step whether or not the wave shares its SIMD, so the pipe idles half the time and two waves per SIMD overlap
// only 1.84x.  Here a workgroup is EIGHT waves for the same 128 x 128 tile:
//
//   waves 0-3  (one per SIMD)  MFMA waves: fragment reads from the current LDS buffer, 24 MFMAs, barrier.  Nothing else
//                              is in their stream.
//   waves 4-7  (one per SIMD)  loader waves: wait for the tile they requested two steps ago, split it, store it to the
//                              OTHER LDS buffer, request the tile three steps ahead, barrier.
//
// One s_barrier per k-step orders both (the loaders write the buffer the MFMA waves read in the NEXT step: the double
// buffering of igemm_kernel, unchanged).  Two workgroups per CU = two MFMA waves + two loader waves per SIMD at <= 128
// VGPRs each; the matrix pipe sees two lean MFMA streams, vector-memory issue and conversion work run beside them on the
// VALU / VMEM ports.  Same LDS image, same operand order, same accumulation order as igemm_kernel<128,128,2,2,true,SPLIT>:
// results are bit-identical.
#include <stdlib.h>

#include "common.h"
#include "igemm_common.h"

using namespace cpmconv;

namespace {

template <int BM, int BN, bool BPRE, int LW = 4, int WTN_ = 64>
__global__ __launch_bounds__(64 * ((BM / 64) * (BN / WTN_) + LW))
    __attribute__((amdgpu_waves_per_eu((((BM / 64) * (BN / WTN_) + LW) == 12 ? 3 : (((BM / 64) * (BN / WTN_) + LW) == 10 ? 5 : 4)),
                                       (((BM / 64) * (BN / WTN_) + LW) == 12 ? 3 : (((BM / 64) * (BN / WTN_) + LW) == 10 ? 5 : 4))))) void igemm_ws_kernel(IgemmArgs a) {
  constexpr int WM = BM / 64, WN = BN / WTN_;   // MFMA waves: one 64 x WTN_ tile each
  constexpr int NW = WM * WN;
  constexpr int NT = 64 * (NW + LW);
  constexpr int NL = 64 * LW;                   // loader threads (the last LW waves)
  constexpr int RPP = NL / 8;                   // rows per load pass
  constexpr int WTM = 64, WTN = WTN_;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int AP = BM / RPP, BP = BN / RPP;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % RPP == 0 && BN % RPP == 0 && RPP % 16 == 0, "tile shape");
  constexpr int CP = BN + 4;
  constexpr int LDS_AB = 2 * (BM + BN) * 32, LDS_C = BM * CP;
  __shared__ __attribute__((aligned(16))) float smem[LDS_AB > LDS_C ? LDS_AB : LDS_C];
  unsigned* const sm = reinterpret_cast<unsigned*>(smem);
  constexpr int PA_HI = 0, PA_LO = 2 * BM * 16, PB_HI = 4 * BM * 16, PB_LO = 4 * BM * 16 + 2 * BN * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= NW;
  const int tiles_n = (a.OCg + BN - 1) / BN;
  int bid = blockIdx.x;
  if (a.xcd_swizzle) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int g = blockIdx.y;
  const int split = blockIdx.z;
  const int m0 = a.m_base + tile_m * BM, n0 = tile_n * BN;

  const int per = (a.ksteps + a.split_k - 1) / a.split_k;
  const int k_begin = split * per;
  const int k_end = min(a.ksteps, k_begin + per);
  const int nk = k_end - k_begin;

  const int wq = wave % NW;                     // MFMA wave position in the WM x WN grid (loader waves: unused)
  const int wm = wq / WN, wn = wq % WN;
  f32x16 acc[TM][TN];

  if (loader) {
    // ---- loader waves ------------------------------------------------------------------------------------------------
    const int lt = tid - 64 * NW;
    const int lrow = lt >> 3, lcol = (lt & 7) * 4;
    unsigned a_off[AP];
    int a_h[AP], a_w[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = m0 + i * RPP + lrow;
      const bool ok = m < a.M;
      const int mm = ok ? m : 0;
      const int jj = mm % a.OWp, t = mm / a.OWp;
      const int ii = t % a.OHp, n = t / a.OHp;
      a_h[i] = ok ? ii * a.ihmul + a.ihadd : -(1 << 28);
      a_w[i] = jj * a.iwmul + a.iwadd;
      a_off[i] = (unsigned)(((n * a.IH + (ok ? a_h[i] : 0)) * a.IW + a_w[i]) * a.Ctot + g * a.CgR + lcol) * 4u;
      asm volatile("" : "+v"(a_off[i]), "+v"(a_h[i]), "+v"(a_w[i]));
    }
    unsigned b_off[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int oc = n0 + i * RPP + lrow;
      b_off[i] = oc < a.OCg ? (unsigned)((g * a.OCg + oc) * a.R * a.S * a.CgR + lcol) * 4u : B_INVALID;
      asm volatile("" : "+v"(b_off[i]));
    }
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, a.in_bytes), rs_wm = make_rsrc(a.wm, a.wm_bytes);
    int l_tr, l_ts, l_cb;
    {
      const int tap = k_begin / a.ksteps_per_tap;
      l_cb = (k_begin - tap * a.ksteps_per_tap) * BK;
      l_tr = tap / a.ns;
      l_ts = tap - l_tr * a.ns;
    }
    const int cb_end = a.ksteps_per_tap * BK;
    auto load_tile = [&](bool live, float4 (&ra)[AP], float4 (&rb)[BP]) {
      const int cb = l_cb;
      const int dh = l_tr * a.hstep, dw = l_ts * a.wstep;
      const unsigned wtap = (unsigned)(((a.r0 + l_tr * a.rstep) * a.S + a.s0 + l_ts * a.sstep) * a.CgR + cb) * 4u;
      const unsigned aoff = (unsigned)((dh * a.IW + dw) * a.Ctot + cb) * 4u;
      {
        const int ncb = l_cb + BK;
        const bool wrap_c = ncb >= cb_end;
        const int nts = l_ts + (wrap_c ? 1 : 0);
        const bool wrap_s = nts == a.ns;
        l_cb = wrap_c ? 0 : ncb;
        l_ts = wrap_s ? 0 : nts;
        l_tr += wrap_s ? 1 : 0;
      }
      const bool c_ok = live & (cb + lcol < a.CgR);
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const bool ok = c_ok & ((unsigned)(a_h[i] + dh) < (unsigned)a.IH) & ((unsigned)(a_w[i] + dw) < (unsigned)a.IW);
        ra[i] = bload4(rs_in, ok ? a_off[i] + aoff : OOB_OFF);
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) rb[i] = bload4(rs_wm, c_ok ? b_off[i] + wtap : OOB_OFF);
    };
    const int w_sw = ((((lcol >> 3) ^ ((lrow >> 2) & 3)) << 2) | ((lcol >> 1) & 2));
    auto store_tile = [&](int buf, const float4 (&ra)[AP], const float4 (&rb)[BP]) {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        uint2 hi, lo;
        split4(ra[i], hi, lo);
        const int o = (buf * BM + i * RPP + lrow) * 16 + w_sw;
        *(uint2*)(sm + PA_HI + o) = hi;
        *(uint2*)(sm + PA_LO + o) = lo;
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        uint2 hi, lo;
        if (BPRE) {
          hi = make_uint2(__float_as_uint(rb[i].x), __float_as_uint(rb[i].y));
          lo = make_uint2(__float_as_uint(rb[i].z), __float_as_uint(rb[i].w));
        } else {
          split4(rb[i], hi, lo);
        }
        const int o = (buf * BN + i * RPP + lrow) * 16 + w_sw;
        *(uint2*)(sm + PB_HI + o) = hi;
        *(uint2*)(sm + PB_LO + o) = lo;
      }
    };
    float4 ra0[AP], rb0[BP], ra1[AP], rb1[BP];
    // tile t lives in register set t & 1 from its request until its store, one step before the MFMA waves read it
    load_tile(nk > 0, ra0, rb0);
    load_tile(nk > 1, ra1, rb1);
    store_tile(0, ra0, rb0);
    load_tile(nk > 2, ra0, rb0);
    __syncthreads();
    for (int it = 0; it < nk; it += 2) {
      // step `it` (the MFMA waves read buffer 0): tile it + 1 -> buffer 1, request tile it + 3
      store_tile(1, ra1, rb1);
      load_tile(it + 3 < nk, ra1, rb1);
      __syncthreads();
      if (it + 1 < nk) {
        store_tile(0, ra0, rb0);
        load_tile(it + 4 < nk, ra0, rb0);
        __syncthreads();
      }
    }
  } else {
    // ---- MFMA waves --------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31;
    struct Frag { bf16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
    auto fetch = [&](int cur, int sub, Frag& f) {
      const int r_sw = (((sub * 2 + (lane >> 5)) ^ ((frow >> 2) & 3)) << 2);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = (cur * BM + wm * WTM + i * 32 + frow) * 16 + r_sw;
        f.ah[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PA_HI + o));
        f.al[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PA_LO + o));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = (cur * BN + wn * WTN + j * 32 + frow) * 16 + r_sw;
        f.bh[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PB_HI + o));
        f.bl[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PB_LO + o));
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
    __syncthreads();                             // tile 0 is in buffer 0
    for (int it = 0; it < nk; ++it) {
      const int cur = it & 1;
      Frag f0, f1;
      fetch(cur, 0, f0);
      mfma3(f0);
      fetch(cur, 1, f1);
      mfma3(f1);
      __syncthreads();
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------------------------------
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  const bool dense_rows = a.osh == 1 && a.osw == 1 && a.OHp == a.OH && a.OWp == a.OW;
  if ((a.res || a.staged_epi) && !a.atomic_out) {
    // residual epilogue: the tile through LDS, finished row-wise with 16-byte accesses by all eight waves
    float (*Cs)[CP] = reinterpret_cast<float (*)[CP]>(smem);
    if (!loader) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            Cs[wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0][wn * WTN + j * 32 + ecol] = acc[i][j][e];
    }
    __syncthreads();
    constexpr int CV = BN / 4;
    constexpr int RPS = NT / CV;
    const int cv = (tid % CV) * 4, r0 = tid / CV;
    const int ocl = n0 + cv;
    const bool vec_out = (a.OCg & 3) == 0 && (a.OCtot & 3) == 0 && ocl + 3 < a.OCg;
    const int oc = g * a.OCg + ocl;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ocl < a.OCg) {
      float* scp = &sc.x; float* shp = &sh.x;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ocl + k < a.OCg) {
          if (a.scale) scp[k] = a.scale[oc + k];
          if (a.shift) shp[k] = a.shift[oc + k];
        }
    }
    if (vec_out) {
      constexpr int UNR = (BM / RPS) >= 4 ? 4 : (BM / RPS);
      for (int rb = r0; rb < BM; rb += RPS * UNR) {
        float4 rv[UNR], gv[UNR];
        size_t oo[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int r = rb + u * RPS, m = m0 + r;
          ok[u] = r < BM && m < a.M;
          const int mm = ok[u] ? m : m0;
          int orow = mm, oh = 0, ow = 0, n = 0;
          if (!dense_rows || a.res_mode == 1) {
            const int jj = mm % a.OWp, t = mm / a.OWp;
            const int ii = t % a.OHp;
            n = t / a.OHp;
            oh = ii * a.osh + a.oah;
            ow = jj * a.osw + a.oaw;
            orow = (n * a.OH + oh) * a.OW + ow;
          }
          oo[u] = (size_t)orow * a.OCtot + oc;
          rv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          gv[u] = make_float4(1.f, 1.f, 1.f, 1.f);
          if (a.res && ok[u]) {
            const float* rp = a.res_mode == 0
                                  ? a.res + oo[u]
                                  : a.res + ((size_t)(n * ((a.OH + 1) / 2) + oh / 2) * ((a.OW + 1) / 2) + ow / 2) * a.OCtot + oc;
            rv[u] = *(const float4*)rp;
          }
          if (a.mask && ok[u]) gv[u] = *(const float4*)(a.mask + oo[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (!ok[u]) continue;
          float4 v = *(const float4*)&Cs[rb + u * RPS][cv];
          v.x = v.x * sc.x + sh.x + rv[u].x; v.y = v.y * sc.y + sh.y + rv[u].y;
          v.z = v.z * sc.z + sh.z + rv[u].z; v.w = v.w * sc.w + sh.w + rv[u].w;
          if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (a.mask) {
            v.x = gv[u].x > 0.f ? v.x : 0.f; v.y = gv[u].y > 0.f ? v.y : 0.f;
            v.z = gv[u].z > 0.f ? v.z : 0.f; v.w = gv[u].w > 0.f ? v.w : 0.f;
          }
          *(float4*)(a.out + oo[u]) = v;
        }
      }
      return;
    }
    for (int r = r0; r < BM; r += RPS) {
      const int m = m0 + r;
      if (m >= a.M || ocl >= a.OCg) continue;
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
      float* dst = a.out + (size_t)orow * a.OCtot + oc;
      const float* rp = !a.res ? nullptr
                        : a.res_mode == 0
                            ? a.res + (size_t)orow * a.OCtot + oc
                            : a.res + ((size_t)(n * ((a.OH + 1) / 2) + oh / 2) * ((a.OW + 1) / 2) + ow / 2) * a.OCtot + oc;
      float* vp = &v.x;
      for (int k = 0; k < 4 && ocl + k < a.OCg; ++k) {
        float o = vp[k] + (rp ? rp[k] : 0.f);
        if (a.relu) o = fmaxf(o, 0.f);
        if (a.mask) o = a.mask[(size_t)orow * a.OCtot + oc + k] > 0.f ? o : 0.f;
        dst[k] = o;
      }
    }
    return;
  }
  if (loader) return;
  float e_sc[TN], e_sh[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ocl = n0 + wn * WTN + j * 32 + ecol;
    const bool ok = ocl < a.OCg && !a.atomic_out;
    e_sc[j] = (a.scale && ok) ? a.scale[g * a.OCg + ocl] : 1.f;
    e_sh[j] = (a.shift && ok) ? a.shift[g * a.OCg + ocl] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0;
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
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int ocl = n0 + wn * WTN + j * 32 + ecol;
        if (ocl >= a.OCg) continue;
        const int oc = g * a.OCg + ocl;
        float v = acc[i][j][e];
        float* dst = a.out + (size_t)orow * a.OCtot + oc;
        if (a.atomic_out) {
          if (a.slab) a.slab[(size_t)split * a.slab_stride + (size_t)orow * a.OCtot + oc] = v;
          else atomicAdd(dst, v);
        } else {
          if (a.scale) v *= e_sc[j];
          if (a.shift) v += e_sh[j];
          if (a.res) {
            if (a.res_mode == 0) {
              v += a.res[(size_t)orow * a.OCtot + oc];
            } else {
              const int rh = (a.OH + 1) / 2, rw = (a.OW + 1) / 2;
              v += a.res[((size_t)(n * rh + oh / 2) * rw + ow / 2) * a.OCtot + oc];
            }
          }
          if (a.relu) v = fmaxf(v, 0.f);
          if (a.mask) v = a.mask[(size_t)orow * a.OCtot + oc] > 0.f ? v : 0.f;
          *dst = v;
        }
      }
    }
  }
}

}  // namespace

namespace cpmconv {

int launch_ws(const IgemmArgs& a, int bm, int bn, hipStream_t s) {
  const int rows = a.M - a.m_base;
  dim3 grid((unsigned)(cpm::cdiv(rows, bm) * cpm::cdiv(a.OCg, bn)), a.groups, a.split_k);
  if (bm == 128 && bn == 128) {
    if (a.b_presplit) hipLaunchKernelGGL((igemm_ws_kernel<128, 128, true>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((igemm_ws_kernel<128, 128, false>), grid, dim3(512), 0, s, a);
  } else if (bm == 256 && bn == 128) {
    static const char* lwv = getenv("CPM_IGEMM_WS_LOADERS");
    const int lw = lwv ? atoi(lwv) : 8;
    if (lw == 8) {
      if (a.b_presplit) hipLaunchKernelGGL((igemm_ws_kernel<256, 128, true, 8>), grid, dim3(1024), 0, s, a);
      else hipLaunchKernelGGL((igemm_ws_kernel<256, 128, false, 8>), grid, dim3(1024), 0, s, a);
    } else {
      if (a.b_presplit) hipLaunchKernelGGL((igemm_ws_kernel<256, 128, true>), grid, dim3(768), 0, s, a);
      else hipLaunchKernelGGL((igemm_ws_kernel<256, 128, false>), grid, dim3(768), 0, s, a);
    }
  } else if (bm == 128 && bn == 96) {
    if (a.b_presplit) hipLaunchKernelGGL((igemm_ws_kernel<128, 96, true, 4, 32>), grid, dim3(640), 0, s, a);
    else hipLaunchKernelGGL((igemm_ws_kernel<128, 96, false, 4, 32>), grid, dim3(640), 0, s, a);
  } else {
    return CPM_EINVAL;
  }
  return cpm::check_launch("conv igemm (loader + MFMA waves)");
}

}  // namespace cpmconv
