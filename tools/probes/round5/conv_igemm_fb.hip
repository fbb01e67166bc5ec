// Implicit-GEMM convolution, bf16x3 arithmetic, with the WEIGHT operand read straight into MFMA fragments from a
// fragment-major image -- no LDS, no conversion work and no barrier on that side (the k-step of
// igemm_kernel moves both operands global -> registers -> split -> LDS -> fragments; the LDS pipe of a CU then carries
// 64 KB of stores and 128 KB of reads per k-step round of its eight waves, collided behind one barrier).
//
// Here only the ACTIVATION operand takes that road (it must: it arrives as fp32 rows and is split on the way).  The
// weight operand of a conv is the same for every pixel tile and every step until the optimizer moves it, so it is kept
// as the operand registers of v_mfma_f32_32x32x16_bf16 themselves:
//
//   image[col block cb32][k-step ks][half sub][plane hi|lo][lane 0..63] = 16 bytes = 8 bf16:
//       column  oc = cb32 * 32 + (lane & 31)
//       k       c  = (ks % ksteps_per_tap) * 32 + sub * 16 + (lane >> 5) * 8 + 0..7  of tap ks / ksteps_per_tap
//
// so one buffer_load_dwordx4 per (column block, half, plane) is a whole wave's B operand: 1 KB contiguous, shared
// through the L1 / L2 by every workgroup of the launch.  A wave holds the fragments of the CURRENT k-step in one register
// set and requests the next step's into the other (weights are L2-resident: one step of ~3 000 cycles covers that
// latency); the activation tiles keep igemm_kernel's two-steps-ahead register sets and its LDS image.
// Same products, same order of accumulation per output as igemm_kernel<.., SPLIT=2>: results are bit-identical.
#include <stdlib.h>

#include "common.h"
#include "igemm_common.h"

using namespace cpmconv;

namespace {

// ---- the image ---------------------------------------------------------------------------------------------------------
// src: the weight as the igemm kernels read it, [rows][taps][CgR] (forward: KRSC; data gradient: the re-laid image), as
// f32 or as its pre-split image (w4: per 4 consecutive elements 8 B of hi, 8 B of lo).  One thread per 16-byte entry.
__global__ __launch_bounds__(256) void wfrag_kernel(const float* __restrict__ src, int presplit, int rows, int taps,
                                                    int cgr, int kpt, uint4* __restrict__ dst, int64_t entries) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < entries; e += (int64_t)gridDim.x * 256) {
    const int lane = (int)(e & 63);
    const int plane = (int)((e >> 6) & 1), sub = (int)((e >> 7) & 1);
    const int64_t t = e >> 8;                                  // cb32 * ksteps + ks
    const int ksteps = taps * kpt;
    const int ks = (int)(t % ksteps), cb32 = (int)(t / ksteps);
    const int tap = ks / kpt, cblk = ks - tap * kpt;
    const int oc = cb32 * 32 + (lane & 31);
    const int c0 = cblk * 32 + sub * 16 + (lane >> 5) * 8;
    unsigned o[4] = {0u, 0u, 0u, 0u};
    if (oc < rows) {
      const size_t base = ((size_t)oc * taps + tap) * cgr;
#pragma unroll
      for (int h = 0; h < 2; ++h) {                            // two groups of four consecutive channels
        const int c = c0 + 4 * h;
        if (c < cgr) {                                         // cgr % 4 == 0: a group is inside or outside as a whole
          uint2 hi, lo;
          if (presplit) {
            const uint4 v = *(const uint4*)(src + base + c);
            hi = make_uint2(v.x, v.y); lo = make_uint2(v.z, v.w);
          } else {
            split4(*(const float4*)(src + base + c), hi, lo);
          }
          o[2 * h] = plane ? lo.x : hi.x;
          o[2 * h + 1] = plane ? lo.y : hi.y;
        }
      }
    }
    dst[e] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ---- the kernel --------------------------------------------------------------------------------------------------------
// ABL: timing-only ablations (CPM_FB_ABL; results are wrong): 1 no activation loads, 2 no split / LDS stores, 4 no LDS
// fragment reads, 8 no weight-fragment loads, 16 no barrier, 32 no MFMAs
template <int BM, int BN, int WM, int WN, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN)
    __attribute__((amdgpu_waves_per_eu(2, (BM * BN >= 128 * 128 ? 2 : (BM * BN >= 128 * 64 ? 3 : 4))))) void igemm_fb_kernel(
    IgemmArgs a, const uint4* __restrict__ fb, unsigned fb_bytes, int fb_ksteps) {
  constexpr int NT = 64 * WM * WN;
  constexpr int RPP = NT / 8;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int AP = BM / RPP;
  constexpr int NB = TN * 4;                    // B fragment loads per k-step and wave: (column block, half, plane)
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % RPP == 0 && RPP % 16 == 0, "tile shape");
  constexpr int CP = BN + 4;
  constexpr int LDS_A = 2 * BM * 32, LDS_C = BM * CP;       // floats: A hi|lo planes x 2 buffers; epilogue staging
  __shared__ __attribute__((aligned(16))) float smem[LDS_A > LDS_C ? LDS_A : LDS_C];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.OCg + BN - 1) / BN;
  int bid = blockIdx.x;
  if (a.xcd_swizzle) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int split = blockIdx.z;
  const int m0 = a.m_base + tile_m * BM, n0 = tile_n * BN;

  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
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
    a_off[i] = (unsigned)(((n * a.IH + (ok ? a_h[i] : 0)) * a.IW + a_w[i]) * a.Ctot + lcol) * 4u;
    asm volatile("" : "+v"(a_off[i]), "+v"(a_h[i]), "+v"(a_w[i]));
  }
  // B: byte offset of this lane's entry of (column block j, k-step 0, half 0, plane hi); a column block past the
  // last output channel gets an out-of-range offset (zeros, nothing fetched)
  unsigned b_base[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c0 = n0 + wn * WTN + j * 32;
    b_base[j] = c0 < a.OCg ? (unsigned)((((size_t)(c0 >> 5) * fb_ksteps) * 4) * 64 + lane) * 16u : B_INVALID;
    asm volatile("" : "+v"(b_base[j]));
  }

  const int per = (a.ksteps + a.split_k - 1) / a.split_k;
  const int k_begin = split * per;
  const int k_end = min(a.ksteps, k_begin + per);
  const int nk = k_end - k_begin;

  float4 ra0[AP], ra1[AP];
  u32x4 bq0[NB], bq1[NB];
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, a.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_fb = __builtin_amdgcn_make_buffer_rsrc((void*)fb, 0, (int)fb_bytes, 0x00020000);

  // two cursors over (tap row, tap column, channel block): the activation loads run two k-steps ahead, the fragment
  // loads one
  struct Cur { int tr, ts, cb; };
  Cur ca, cbq;
  {
    const int tap = k_begin / a.ksteps_per_tap;
    ca.cb = (k_begin - tap * a.ksteps_per_tap) * BK;
    ca.tr = tap / a.ns;
    ca.ts = tap - ca.tr * a.ns;
    cbq = ca;
  }
  const int cb_end = a.ksteps_per_tap * BK;
  auto advance = [&](Cur& c) {
    const int ncb = c.cb + BK;
    const bool wrap_c = ncb >= cb_end;
    const int nts = c.ts + (wrap_c ? 1 : 0);
    const bool wrap_s = nts == a.ns;
    c.cb = wrap_c ? 0 : ncb;
    c.ts = wrap_s ? 0 : nts;
    c.tr += wrap_s ? 1 : 0;
  };
  auto load_a = [&](bool live, float4 (&ra)[AP]) {
    const int cb = ca.cb;
    const int dh = ca.tr * a.hstep, dw = ca.ts * a.wstep;
    const unsigned aoff = (unsigned)((dh * a.IW + dw) * a.Ctot + cb) * 4u;
    advance(ca);
    const bool c_ok = live & (cb + lcol < a.CgR);
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const bool ok = c_ok & ((unsigned)(a_h[i] + dh) < (unsigned)a.IH) & ((unsigned)(a_w[i] + dw) < (unsigned)a.IW);
      ra[i] = bload4(rs_in, ok ? a_off[i] + aoff : OOB_OFF);
    }
  };
  auto load_b = [&](bool live, u32x4 (&bq)[NB]) {
    // k-step of the image: taps in their natural (r, s) order, whatever subset / order this launch walks
    const int tap = (a.r0 + cbq.tr * a.rstep) * a.S + a.s0 + cbq.ts * a.sstep;
    const unsigned koff = (unsigned)(tap * a.ksteps_per_tap + cbq.cb / BK) * 4096u;      // 4 entries x 64 lanes x 16 B
    advance(cbq);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)                                  // e = half * 2 + plane
        bq[j * 4 + e] = __builtin_amdgcn_raw_buffer_load_b128(rs_fb, (int)(live ? b_base[j] + koff + e * 1024u : OOB_OFF),
                                                              0, 0);
  };

  unsigned* const sm = reinterpret_cast<unsigned*>(smem);
  constexpr int PA_HI = 0, PA_LO = 2 * BM * 16;
  const int w_sw = ((((lcol >> 3) ^ ((lrow >> 2) & 3)) << 2) | ((lcol >> 1) & 2));
  auto store_a = [&](int buf, const float4 (&ra)[AP]) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      uint2 hi, lo;
      split4(ra[i], hi, lo);
      const int o = (buf * BM + i * RPP + lrow) * 16 + w_sw;
      *(uint2*)(sm + PA_HI + o) = hi;
      *(uint2*)(sm + PA_LO + o) = lo;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31;
  struct FragA { bf16x8 ah[TM], al[TM]; };
  auto fetch_a = [&](int cur, int sub, FragA& f) {
    const int r_sw = (((sub * 2 + (lane >> 5)) ^ ((frow >> 2) & 3)) << 2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int o = (cur * BM + wm * WTM + i * 32 + frow) * 16 + r_sw;
      f.ah[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PA_HI + o));
      f.al[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PA_LO + o));
    }
  };
  auto mfma3 = [&](const FragA& f, const u32x4 (&bq)[NB], int sub) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[j * 4 + sub * 2]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, bq[j * 4 + sub * 2 + 1]);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], bh, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], bl, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], bh, acc[i][j], 0, 0, 0);
      }
  };
  // One k-step: request the NEXT step's weight fragments and the activation tile two steps ahead, multiply the first
  // half, move the activation tile of the next step (loaded a step ago) to the other LDS buffer, multiply the second
  // half, barrier.
  FragA fc0, fc1;
  auto eat = [&](const FragA& f, const u32x4 (&bq)[NB]) {        // ABL & 32: keep the operands live without multiplying
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(f.ah[i]), "v"(f.al[i]));
#pragma unroll
    for (int j = 0; j < NB; ++j) asm volatile("" ::"v"(bq[j]));
  };
  auto step = [&](int it, int cur, float4 (&la)[AP], const float4 (&sa)[AP], const u32x4 (&bcur)[NB], u32x4 (&bnext)[NB]) {
    if (!(ABL & 8)) load_b(it + 1 < nk, bnext);
    if (!(ABL & 1)) load_a(it + 2 < nk, la);
    FragA f0, f1;
    if (!(ABL & 4)) {
      fetch_a(cur, 0, f0);
      fetch_a(cur, 1, f1);
    } else {
      f0 = fc0; f1 = fc1;
    }
    if (!(ABL & 32)) mfma3(f0, (ABL & 8) ? bq0 : bcur, 0); else eat(f0, (ABL & 8) ? bq0 : bcur);
    if (!(ABL & 2)) store_a(cur ^ 1, (ABL & 1) ? ra0 : sa);
    if (!(ABL & 32)) mfma3(f1, (ABL & 8) ? bq0 : bcur, 1); else eat(f1, (ABL & 8) ? bq0 : bcur);
    constexpr int NM = TM * TN * 3 * 2;
    constexpr int VPM = (AP * 12 + (AP + NB) * 3 + NM - 1) / NM;
    constexpr int WEVERY = NM / AP > 0 ? NM / AP : 1;
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * TM, 0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      if (m >= 1 && m <= AP + NB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (m % WEVERY == WEVERY - 1) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
    }
    if (!(ABL & 16)) __syncthreads();
  };

  if (nk > 0) {
    load_b(true, bq0);
    load_a(true, ra0);
    store_a(0, ra0);
    load_a(nk > 1, ra1);
  }
  __syncthreads();
  if (ABL & 4) { fetch_a(0, 0, fc0); fetch_a(0, 1, fc1); }
  for (int it = 0; it < nk; it += 2) {
    step(it, 0, ra0, ra1, bq0, bq1);
    if (it + 1 < nk) step(it + 1, 1, ra1, ra0, bq1, bq0);
  }

  // ---- epilogue (igemm_kernel's two forms) ---------------------------------------------------------------------------
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  const bool dense_rows = a.osh == 1 && a.osw == 1 && a.OHp == a.OH && a.OWp == a.OW;
  const int g = 0;
  if ((a.res || a.staged_epi) && !a.atomic_out) {
    float (*Cs)[CP] = reinterpret_cast<float (*)[CP]>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          Cs[wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0][wn * WTN + j * 32 + ecol] = acc[i][j][e];
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
    float gate[16][TN];
    if (a.mask && !a.atomic_out) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0;
        int orow = m;
        if (!dense_rows && m < a.M) {
          const int jj = m % a.OWp, t = m / a.OWp;
          orow = ((t / a.OHp) * a.OH + (t % a.OHp) * a.osh + a.oah) * a.OW + jj * a.osw + a.oaw;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int ocl = n0 + wn * WTN + j * 32 + ecol;
          gate[e][j] = (m < a.M && ocl < a.OCg) ? a.mask[(size_t)orow * a.OCtot + g * a.OCg + ocl] : 1.f;
        }
      }
    }
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
          if (a.mask) v = gate[e][j] > 0.f ? v : 0.f;
          *dst = v;
        }
      }
    }
  }
}

}  // namespace

namespace cpmconv {

size_t wfrag_bytes(int rows, int taps, int cgr) {
  return (size_t)cpm::cdiv(rows, 32) * taps * cpm::cdiv(cgr, BK) * 4096;
}

int build_wfrag(const float* src, int presplit, int rows, int taps, int cgr, void* dst, hipStream_t s) {
  const int64_t entries = (int64_t)(wfrag_bytes(rows, taps, cgr) / 16);
  const int64_t b = (entries + 255) / 256;
  hipLaunchKernelGGL(wfrag_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(256), 0, s, src, presplit, rows, taps, cgr,
                     cpm::cdiv(cgr, BK), (uint4*)dst, entries);
  return cpm::check_launch("weight fragment image");
}

bool fb_supported(const IgemmArgs& a, int bm, int bn) {
  return a.groups == 1 && a.CgR == a.Ctot && (a.CgR & 3) == 0 && (bm == 128 || bm == 64) && (bn == 128 || bn == 64) &&
         wfrag_bytes(a.OCg, a.R * a.S, a.CgR) < 0x7FFFF000ull;
}

int launch_fb(const IgemmArgs& a, int bm, int bn, const void* fb, hipStream_t s) {
  const int rows = a.M - a.m_base;
  const unsigned bytes = (unsigned)wfrag_bytes(a.OCg, a.R * a.S, a.CgR);
  const int fbk = a.R * a.S * a.ksteps_per_tap;
  dim3 grid((unsigned)(cpm::cdiv(rows, bm) * cpm::cdiv(a.OCg, bn)), 1, a.split_k);
  static const char* ablv = getenv("CPM_FB_ABL");
  const int abl = ablv ? atoi(ablv) : 0;
#define ABL_CASE(V) if (abl == V) { hipLaunchKernelGGL((igemm_fb_kernel<128, 128, 2, 2, V>), grid, dim3(256), 0, s, a, (const uint4*)fb, bytes, fbk); return cpm::check_launch("fb ablation"); }
  if (bm == 128 && bn == 128 && abl) {
    ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(7) ABL_CASE(8) ABL_CASE(15) ABL_CASE(16) ABL_CASE(31) ABL_CASE(32)
    ABL_CASE(47) ABL_CASE(23) ABL_CASE(6) ABL_CASE(9) ABL_CASE(11)
  }
#undef ABL_CASE
  if (bm == 128 && bn == 128)
    hipLaunchKernelGGL((igemm_fb_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, a, (const uint4*)fb, bytes, fbk);
  else if (bm == 128 && bn == 64)
    hipLaunchKernelGGL((igemm_fb_kernel<128, 64, 2, 2>), grid, dim3(256), 0, s, a, (const uint4*)fb, bytes, fbk);
  else if (bm == 64 && bn == 64)
    hipLaunchKernelGGL((igemm_fb_kernel<64, 64, 2, 2>), grid, dim3(256), 0, s, a, (const uint4*)fb, bytes, fbk);
  else
    return CPM_EINVAL;
  return cpm::check_launch("conv igemm (fragment-major weights)");
}

}  // namespace cpmconv
