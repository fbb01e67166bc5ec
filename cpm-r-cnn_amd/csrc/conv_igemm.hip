// Implicit-GEMM convolution on the gfx950 matrix cores, NHWC: exact fp32 (v_mfma_f32_32x32x2_f32) or 3-term
// split-bf16 with fp32 accumulation (v_mfma_f32_32x32x16_bf16), selected by cpm_set_conv_math.
//
// One kernel family serves conv2d forward, its data gradient (also = ConvTranspose2d forward) and
// nn.Linear (a 1x1 conv on a 1x1 image); a second one serves the weight gradient.
//
//   out[m][oc] = sum_{tap=(r,s)} sum_{c} in[gather(m, tap)][c] * Wm[oc][tap][c]
//
//   * GEMM rows m = (n, oh, ow) output pixels, columns oc output channels, reduction (tap, c).
//   * gather: one parametrised description for forward and data-gradient (see IgemmArgs); the data gradient
//     runs one launch per stride phase and uses the weight re-laid as Wm[c][r][s][k] (transform kernel below),
//     so that both directions read the B operand as "row = output channel, 32 consecutive reduction elements".
//   * workgroup = 256 threads = 4 waves (one per SIMD), tile BM x BN x 32, waves in a WM x WN grid,
//     each wave owns (BM/WM) x (BN/WN) as 32x32 MFMA tiles.
//   * f32 mode: LDS A[2][BM][36], B[2][BN][36] floats; the 36-float row pitch (144 B) makes every
//     ds_read_b128 of a 16-lane group hit 16 distinct 16-B slots (bank = dword mod 64).  A lane reads 4
//     consecutive k for its row with ONE ds_read_b128 and feeds 4 MFMAs with them: the MFMA's two k-slots
//     (lane>>5) therefore hold k = {j, 4+j} -- a permutation of the reduction order applied identically to A and
//     B, which leaves the sum unchanged.  Results are an exact-fp32 fmaf chain per output.
//   * bf16x3 mode: every fp32 value is split while it moves from registers to LDS (hi = bf16(x), lo = bf16(x - hi));
//     LDS holds hi and lo planes of 64-byte rows with XOR-swizzled 16-byte chunks; a product is
//     a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on the bf16 MFMA (see split4 / store_tile / fetch / mfma3).
//   * global->LDS through registers, issued two k-steps ahead (two register sets), one barrier per k-step,
//     2 workgroups per CU for the 128x128 tile, 3-4 for the smaller ones.
//   * epilogue fused: per-channel scale/shift (bias or frozen AffineChannel2d), residual add
//     (same-shape or nearest-2x-upsampled, the FPN top-down path; also used to accumulate a data gradient in
//     place), ReLU, and the ReLU gate / scale of the layer that produced the input (gated data gradient).
//   * split-K (grid.z) for skinny problems (FC layers, small RoI counts): partial sums are added with
//     float atomics into a zeroed output and the epilogue runs as a second tiny kernel.
//
// Reference call sites replaced: see include/cpmrcnn_hip.h (conv section).
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "common.h"
#include "igemm_common.h"

using namespace cpmconv;

namespace {

// SPLIT: 0 = exact f32 MFMA, 1 = bf16x3 with both operands split in the kernel, 2 = bf16x3 with the WEIGHT operand given
// as a pre-split image (a.wm points at it; cpm_split_w4: per four consecutive reduction elements 8 bytes of hi and 8
// bytes of lo in place of their 16 bytes of f32 -- the same offsets, no conversion work on that side).
// TAIL (vector path): the reduction channel count is not a multiple of 4 (the data gradient of an 18-channel conv -- the
// offset predictors of DeformConvPack): rows are then 4- or 8-byte aligned only, which buffer_load_dwordx4 serves at
// full width (tools/probes/round5/unaligned_probe.hip), and the components of a 16-byte load that reach past the row's
// channels -- the next tap's / pixel's first values -- are zeroed in registers (past the tensor's end the descriptor's
// per-dword range check returns zeros by itself: range_probe.hip).  Before: the scalar path, 40.8 us against 26.3 for
// the same layer padded to 20 channels.
template <int BM, int BN, int WM, int WN, bool VEC, int SPLIT, bool TAIL = false>
__global__ __launch_bounds__(64 * WM * WN)
    __attribute__((amdgpu_waves_per_eu(2, (BM * BN >= 128 * 128 ? 2 : (BM * BN >= 128 * 64 ? 3 : 4))))) void igemm_kernel(
        IgemmArgs a) {
  constexpr int NT = 64 * WM * WN;              // threads
  constexpr int RPP = NT / 8;                   // rows per load pass (8 threads x float4 cover a 32-float row)
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;   // MFMA tiles per wave
  constexpr int AP = BM / RPP, BP = BN / RPP;   // load passes
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % RPP == 0 && BN % RPP == 0, "tile shape");

  constexpr int CP = BN + 4;                    // epilogue staging pitch (floats)
  constexpr int LDS_AB = SPLIT ? 2 * (BM + BN) * 32 : 2 * (BM + BN) * LDP, LDS_C = BM * CP;
  __shared__ __attribute__((aligned(16))) float smem[LDS_AB > LDS_C ? LDS_AB : LDS_C];
  float (*As)[BM][LDP] = reinterpret_cast<float (*)[BM][LDP]>(smem);
  float (*Bs)[BN][LDP] = reinterpret_cast<float (*)[BN][LDP]>(smem + 2 * BM * LDP);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.OCg + BN - 1) / BN;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an L2), so give
  // each XCD a contiguous run of logical tiles -- the N-tiles of one row block and neighbouring row blocks (which
  // share 3x3 halo rows) then hit the same L2.  Bijective for any grid size; a speed hint only.
  int bid = blockIdx.x;
  if (a.xcd_swizzle) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int g = blockIdx.y;
  const int split = blockIdx.z;
  const int m0 = a.m_base + tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread load geometry --------------------------------------------------------------
  // Everything a load needs per k-step is (row byte offset + one uniform tap offset) and two range checks; the
  // offsets are pinned in registers (the empty asm) -- otherwise the compiler re-derives them from n/h/w with
  // integer multiplies inside every k-step and sinks each load under its own exec-masked branch.
  const int lrow = tid >> 3;        // 0..RPP-1
  const int lcol = (tid & 7) * 4;   // 0,4,..,28
  unsigned a_off[AP];
  int a_h[AP], a_w[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + i * RPP + lrow;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int jj = mm % a.OWp, t = mm / a.OWp;
    const int ii = t % a.OHp, n = t / a.OHp;
    a_h[i] = ok ? ii * a.ihmul + a.ihadd : -(1 << 28);      // a row past M never passes the range check
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

  // k-step range of this split
  const int per = (a.ksteps + a.split_k - 1) / a.split_k;
  const int k_begin = split * per;
  const int k_end = min(a.ksteps, k_begin + per);
  const int nk = k_end - k_begin;

  // two register sets: the loads of tile t+2 are in flight while tile t is multiplied and tile t+1 moves to LDS
  float4 ra0[AP], rb0[BP], ra1[AP], rb1[BP];
  // (a.dbg, CPM_IGEMM_DBG, timing only: 8 = descriptors of zero records, nothing is fetched; 16 = no epilogue)
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, (a.dbg & 8) ? 0u : a.in_bytes),
                               rs_wm = make_rsrc(a.wm, (a.dbg & 8) ? 0u : a.wm_bytes);

  // position (tap row, tap column, channel block) of the NEXT tile to load, advanced without divisions: the loads of
  // a k-step are then a handful of scalar instructions and sit in the same straight-line block as its MFMAs
  int l_tr, l_ts, l_cb;
  {
    const int tap = k_begin / a.ksteps_per_tap;
    l_cb = (k_begin - tap * a.ksteps_per_tap) * BK;
    l_tr = tap / a.ns;
    l_ts = tap - l_tr * a.ns;
  }
  const int cb_end = a.ksteps_per_tap * BK;
  auto load_tile = [&](bool live, float4 (&ra)[AP], float4 (&rb)[BP]) {     // live == false: every lane masked
    const int cb = l_cb;                                    // first reduction channel of this k-step (uniform)
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
    if (VEC) {
      // branch-free: all the step's loads issue back to back; a masked lane gets an out-of-range offset
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const bool ok = c_ok & ((unsigned)(a_h[i] + dh) < (unsigned)a.IH) & ((unsigned)(a_w[i] + dw) < (unsigned)a.IW);
        ra[i] = bload4(rs_in, ok ? a_off[i] + aoff : OOB_OFF);
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) rb[i] = bload4(rs_wm, c_ok ? b_off[i] + wtap : OOB_OFF);
      if (TAIL) {
        const int rem = a.CgR - (cb + lcol);                  // channels of this lane's 4 that exist (<= 0: all masked above)
        if (rem < 4) {
#pragma unroll
          for (int i = 0; i < AP; ++i) { if (rem < 2) ra[i].y = 0.f; if (rem < 3) ra[i].z = 0.f; ra[i].w = 0.f; }
#pragma unroll
          for (int i = 0; i < BP; ++i) { if (rem < 2) rb[i].y = 0.f; if (rem < 3) rb[i].z = 0.f; rb[i].w = 0.f; }
        }
      }
      return;
    }
    const int c0 = cb + lcol;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const bool ok = c_ok && (unsigned)(a_h[i] + dh) < (unsigned)a.IH && (unsigned)(a_w[i] + dw) < (unsigned)a.IW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        const float* src = a.in + ((a_off[i] + aoff) >> 2);
        v.x = src[0];
        if (c0 + 1 < a.CgR) v.y = src[1];
        if (c0 + 2 < a.CgR) v.z = src[2];
        if (c0 + 3 < a.CgR) v.w = src[3];
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c_ok && b_off[i] != B_INVALID) {
        const float* src = a.wm + ((b_off[i] + wtap) >> 2);
        v.x = src[0];
        if (c0 + 1 < a.CgR) v.y = src[1];
        if (c0 + 2 < a.CgR) v.z = src[2];
        if (c0 + 3 < a.CgR) v.w = src[3];
      }
      rb[i] = v;
    }
  };
  // SPLIT LDS image: four planes A_hi | A_lo | B_hi | B_lo, each [2 buffers][rows][32 bf16 = 64 B], no padding.
  // A row's four 16-byte chunks are stored at chunk ^ ((row >> 2) & 3): the 16 lanes of a ds_read_b128 group (16
  // consecutive rows, same chunk) then hit 16 distinct 4-bank groups, and so do the ds_write_b64 of two rows.
  // hi and lo live in different planes (kilobytes apart), so the compiler cannot fuse them into one ds_write2.
  unsigned* const sm = reinterpret_cast<unsigned*>(smem);
  constexpr int PA_HI = 0, PA_LO = 2 * BM * 16, PB_HI = 4 * BM * 16, PB_LO = 4 * BM * 16 + 2 * BN * 16;
  static_assert(!SPLIT || (RPP % 16 == 0 && WTM % 32 == 0), "swizzle assumes row blocks of 16");
  const int w_sw = ((((lcol >> 3) ^ ((lrow >> 2) & 3)) << 2) | ((lcol >> 1) & 2));   // dword offset inside the row
  auto store_tile = [&](int buf, const float4 (&ra)[AP], const float4 (&rb)[BP]) {
    if (SPLIT) {
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
        if (SPLIT == 2) {
          hi = make_uint2(__float_as_uint(rb[i].x), __float_as_uint(rb[i].y));
          lo = make_uint2(__float_as_uint(rb[i].z), __float_as_uint(rb[i].w));
        } else {
          split4(rb[i], hi, lo);
        }
        const int o = (buf * BN + i * RPP + lrow) * 16 + w_sw;
        *(uint2*)(sm + PB_HI + o) = hi;
        *(uint2*)(sm + PB_LO + o) = lo;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) *(float4*)&As[buf][i * RPP + lrow][lcol] = ra[i];
#pragma unroll
    for (int i = 0; i < BP; ++i) *(float4*)&Bs[buf][i * RPP + lrow][lcol] = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31, fk = (lane >> 5) * 4;
  auto mma_half = [&](int cur, int kb0) {          // exact-f32 arithmetic (the bf16x3 path uses fetch / mfma3 below)
#pragma unroll
    for (int kb = kb0; kb < kb0 + BK / 16; ++kb) {
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const float4*)&As[cur][wm * WTM + i * 32 + frow][kb * 8 + fk];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *(const float4*)&Bs[cur][wn * WTN + j * 32 + frow][kb * 8 + fk];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
  };
  // SPLIT: operand fetch and MFMAs as separate pieces, so that a k-step can fetch BOTH 16-deep halves first and
  // then run its 24 MFMAs with the next tile's bf16 split + LDS stores laid into their issue gaps
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
  // One k-step: issue the loads of tile t+2, multiply the first half of tile t, move tile t+1 (loaded a whole
  // step ago, so its wait is short) from registers to the other LDS buffer, multiply the second half, barrier.
  // The store sits in the MIDDLE of the MFMA stream so that the wait + ds_writes of this wave overlap the
  // co-resident waves' MFMAs and the step ends with the barrier alone.
  auto step = [&](int it, int cur, float4 (&la)[AP], float4 (&lb)[BP], const float4 (&sa)[AP],
                  const float4 (&sb)[BP]) {
    load_tile(it + 2 < nk, la, lb);
    if (SPLIT) {
      Frag f0, f1;
      fetch(cur, 0, f0);
      fetch(cur, 1, f1);
      mfma3(f0);
      // unconditional (the last step re-stores a stale tile into the idle buffer): one straight-line region
      store_tile(cur ^ 1, sa, sb);
      mfma3(f1);
      constexpr int NM = TM * TN * 3 * 2;                      // MFMAs of the step
      // VALU ops per MFMA gap: 12 per float4 for the split + ~4 per load for its offset and mask
      constexpr int VPM = ((AP + (SPLIT == 2 ? 0 : BP)) * 12 + (AP + BP) * 4 + NM - 1) / NM;
      constexpr int WEVERY = NM / (AP + BP) > 0 ? NM / (AP + BP) : 1;
      __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        if (m >= 1 && m <= AP + BP) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // the loads early in the step
        if (m % WEVERY == WEVERY - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    } else {
      mma_half(cur, 0);
      if (it + 1 < nk) store_tile(cur ^ 1, sa, sb);
      mma_half(cur, BK / 16);
    }
    __syncthreads();
  };

  if (nk > 0) {
    load_tile(true, ra0, rb0);
    store_tile(0, ra0, rb0);
    load_tile(nk > 1, ra1, rb1);
  }
  __syncthreads();
  for (int it = 0; it < nk; it += 2) {
    step(it, 0, ra0, rb0, ra1, rb1);                    // tile it in LDS[0]; tile it+1 waits in set 1
    if (it + 1 < nk) step(it + 1, 1, ra1, rb1, ra0, rb0);
  }

  // ---- epilogue ----------------------------------------------------------------------------------
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  const bool dense_rows = a.osh == 1 && a.osw == 1 && a.OHp == a.OH && a.OWp == a.OW;
  if (a.dbg & 16) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) t += acc[i][j][0] + acc[i][j][15];
    if (t == 12345.678f) a.out[0] = t;
    return;
  }
  if ((a.res || a.staged_epi) && !a.atomic_out) {
    // Residual epilogue (bottleneck conv3, FPN laterals): memory bound on thin reductions.  Stage the tile through
    // LDS (the operand buffers are free after the last barrier) and finish it row-wise with 16-byte accesses, so
    // that the residual reads and the stores are whole 512-byte rows instead of 4-byte column slices.
    float (*Cs)[CP] = reinterpret_cast<float (*)[CP]>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          Cs[wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0][wn * WTN + j * 32 + ecol] = acc[i][j][e];
    __syncthreads();
    constexpr int CV = BN / 4;                  // float4 per tile row
    constexpr int RPS = NT / CV;                // rows per store pass
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
      // Rows in groups of UNR: the group's residual reads are issued together, THEN the tile values are finished and
      // stored.  One row at a time every residual read sat behind the previous row's store (the compiler cannot move a
      // load above a store that may alias it -- and in the accumulating data gradient res IS out), so a 128-row tile
      // paid 16 dependent memory round trips; on the thin 1x1 layers (2-8 k-steps) that was most of the workgroup's life.
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
      if (vec_out) {
        if (rp) {
          const float4 rv = *(const float4*)rp;
          v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(float4*)dst = v;
      } else {
        float* vp = &v.x;
        for (int k = 0; k < 4 && ocl + k < a.OCg; ++k) {
          float o = vp[k] + (rp ? rp[k] : 0.f);
          if (a.relu) o = fmaxf(o, 0.f);
          if (a.mask) o = a.mask[(size_t)orow * a.OCtot + oc + k] > 0.f ? o : 0.f;
          dst[k] = o;
        }
      }
    }
    return;
  }
  // per-column epilogue constants once per lane (inside the store loop every one of these loads would sit behind
  // the previous store: the compiler cannot hoist a load over a store that may alias it)
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
    // gate operands of the 32-row slab first (independent loads in flight), see igemm3x3_kernel's epilogue
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

// ---- the same kernel with ONE LDS stage and one register set of loads (bf16x3, vector path, direct epilogues) --------
// Measured with in-kernel stamps: a k-step of igemm_kernel costs a wave ~2 900 cycles for 768 cycles of
// MFMA work whether or not it shares its SIMD, and two waves per SIMD overlap 1.84x -- what bounds the kernel is the
// number of instruction streams per SIMD, which its 72 KB of LDS and ~240 VGPRs hold at two.  Here a k-step is
//   barrier (every wave has its operands of tile t-1 in registers) -> split + store tile t into THE stage, issue the
//   loads of tile t+1 into the same registers -> barrier -> all operand reads of tile t -> its 24 MFMAs,
// which run in the pipe while the wave does the next step's split and stores.  32 KB of LDS and <= 168 VGPRs: three
// workgroups per CU.  Epilogues that stage the tile through LDS (residual / staged) stay on igemm_kernel.
template <int BM, int BN, int WM, int WN, bool BPRE = false>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(3, 3))) void igemm_s1_kernel(IgemmArgs a) {
  constexpr bool VEC = true, SPLIT = true;
  constexpr int NT = 64 * WM * WN;              // threads
  constexpr int RPP = NT / 8;                   // rows per load pass (8 threads x float4 cover a 32-float row)
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;   // MFMA tiles per wave
  constexpr int AP = BM / RPP, BP = BN / RPP;   // load passes
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % RPP == 0 && BN % RPP == 0, "tile shape");

  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * 32];      // hi / lo planes of A and B, one stage

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.OCg + BN - 1) / BN;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an L2), so give
  // each XCD a contiguous run of logical tiles -- the N-tiles of one row block and neighbouring row blocks (which
  // share 3x3 halo rows) then hit the same L2.  Bijective for any grid size; a speed hint only.
  int bid = blockIdx.x;
  if (a.xcd_swizzle) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int g = blockIdx.y;
  const int split = blockIdx.z;
  const int m0 = a.m_base + tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread load geometry --------------------------------------------------------------
  // Everything a load needs per k-step is (row byte offset + one uniform tap offset) and two range checks; the
  // offsets are pinned in registers (the empty asm) -- otherwise the compiler re-derives them from n/h/w with
  // integer multiplies inside every k-step and sinks each load under its own exec-masked branch.
  const int lrow = tid >> 3;        // 0..RPP-1
  const int lcol = (tid & 7) * 4;   // 0,4,..,28
  unsigned a_off[AP];
  int a_h[AP], a_w[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + i * RPP + lrow;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int jj = mm % a.OWp, t = mm / a.OWp;
    const int ii = t % a.OHp, n = t / a.OHp;
    a_h[i] = ok ? ii * a.ihmul + a.ihadd : -(1 << 28);      // a row past M never passes the range check
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

  // k-step range of this split
  const int per = (a.ksteps + a.split_k - 1) / a.split_k;
  const int k_begin = split * per;
  const int k_end = min(a.ksteps, k_begin + per);
  const int nk = k_end - k_begin;

  // one register set: the loads of tile t+1 are issued right behind the stores of tile t
  float4 ra0[AP], rb0[BP];
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, a.in_bytes), rs_wm = make_rsrc(a.wm, a.wm_bytes);

  // position (tap row, tap column, channel block) of the NEXT tile to load, advanced without divisions: the loads of
  // a k-step are then a handful of scalar instructions and sit in the same straight-line block as its MFMAs
  int l_tr, l_ts, l_cb;
  {
    const int tap = k_begin / a.ksteps_per_tap;
    l_cb = (k_begin - tap * a.ksteps_per_tap) * BK;
    l_tr = tap / a.ns;
    l_ts = tap - l_tr * a.ns;
  }
  const int cb_end = a.ksteps_per_tap * BK;
  auto load_tile = [&](bool live, float4 (&ra)[AP], float4 (&rb)[BP]) {     // live == false: every lane masked
    const int cb = l_cb;                                    // first reduction channel of this k-step (uniform)
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
    if (VEC) {
      // branch-free: all the step's loads issue back to back; a masked lane gets an out-of-range offset
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const bool ok = c_ok & ((unsigned)(a_h[i] + dh) < (unsigned)a.IH) & ((unsigned)(a_w[i] + dw) < (unsigned)a.IW);
        ra[i] = bload4(rs_in, ok ? a_off[i] + aoff : OOB_OFF);
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) rb[i] = bload4(rs_wm, c_ok ? b_off[i] + wtap : OOB_OFF);
      return;
    }
    const int c0 = cb + lcol;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const bool ok = c_ok && (unsigned)(a_h[i] + dh) < (unsigned)a.IH && (unsigned)(a_w[i] + dw) < (unsigned)a.IW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        const float* src = a.in + ((a_off[i] + aoff) >> 2);
        v.x = src[0];
        if (c0 + 1 < a.CgR) v.y = src[1];
        if (c0 + 2 < a.CgR) v.z = src[2];
        if (c0 + 3 < a.CgR) v.w = src[3];
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c_ok && b_off[i] != B_INVALID) {
        const float* src = a.wm + ((b_off[i] + wtap) >> 2);
        v.x = src[0];
        if (c0 + 1 < a.CgR) v.y = src[1];
        if (c0 + 2 < a.CgR) v.z = src[2];
        if (c0 + 3 < a.CgR) v.w = src[3];
      }
      rb[i] = v;
    }
  };
  // SPLIT LDS image: four planes A_hi | A_lo | B_hi | B_lo, each [2 buffers][rows][32 bf16 = 64 B], no padding.
  // A row's four 16-byte chunks are stored at chunk ^ ((row >> 2) & 3): the 16 lanes of a ds_read_b128 group (16
  // consecutive rows, same chunk) then hit 16 distinct 4-bank groups, and so do the ds_write_b64 of two rows.
  // hi and lo live in different planes (kilobytes apart), so the compiler cannot fuse them into one ds_write2.
  unsigned* const sm = reinterpret_cast<unsigned*>(smem);
  constexpr int PA_HI = 0, PA_LO = BM * 16, PB_HI = 2 * BM * 16, PB_LO = 2 * BM * 16 + BN * 16;
  static_assert(!SPLIT || (RPP % 16 == 0 && WTM % 32 == 0), "swizzle assumes row blocks of 16");
  const int w_sw = ((((lcol >> 3) ^ ((lrow >> 2) & 3)) << 2) | ((lcol >> 1) & 2));   // dword offset inside the row
  auto store_tile = [&](int buf, const float4 (&ra)[AP], const float4 (&rb)[BP]) {
    if (SPLIT) {
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
      return;
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
  // SPLIT: operand fetch and MFMAs as separate pieces, so that a k-step can fetch BOTH 16-deep halves first and
  // then run its 24 MFMAs with the next tile's bf16 split + LDS stores laid into their issue gaps
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
  load_tile(nk > 0, ra0, rb0);
  for (int it = 0; it < nk; ++it) {
    if (it) __syncthreads();                            // every wave holds its operands of tile it-1 in registers
    store_tile(0, ra0, rb0);                            // (waits for the loads issued a step ago)
    load_tile(it + 1 < nk, ra0, rb0);
    __syncthreads();                                    // tile `it` complete in LDS
    Frag f0, f1;
    fetch(0, 0, f0);
    fetch(0, 1, f1);
    mfma3(f0);
    mfma3(f1);
  }

  // ---- epilogue ----------------------------------------------------------------------------------
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  const bool dense_rows = a.osh == 1 && a.osw == 1 && a.OHp == a.OH && a.OWp == a.OW;
  // per-column epilogue constants once per lane (inside the store loop every one of these loads would sit behind
  // the previous store: the compiler cannot hoist a load over a store that may alias it)
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
    // gate operands of the 32-row slab first (independent loads in flight), see igemm3x3_kernel's epilogue
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

// ---- persistent tiles: the bf16x3 / vector / pre-split-weight kernel as ONE pipeline over a run of output tiles ---------
// The thin layers of the step (1x1 convolutions of layer2-4 and their data gradients, 3x3 convolutions on the stride-16 /
// stride-32 maps, the RoI heads' small GEMMs) give a workgroup 2-16 k-steps between a prologue (address arithmetic with
// integer divisions, the first operand fetch: one exposed memory round trip) and an epilogue (residual / gate reads:
// another one, then the stores), and the grid arrives in two or three residency rounds that each pay both again.  Here a
// workgroup owns a contiguous RUN of tiles (tile order: output-channel tile fastest, so a run re-reads the same input
// rows out of L2) and treats (tile, k-step) as one sequence: the operand loads run two steps ahead ACROSS tile
// boundaries (the load cursor carries its own tile geometry), so a tile's epilogue executes with the next tile's first
// two operand tiles already in flight, and nothing is re-launched.  The epilogue goes straight from the accumulators
// (no LDS staging: the operand buffers are live): per 32-row slab every residual / gate value is requested first, then
// the slab is finished and stored.  Arithmetic and reduction order are those of igemm_kernel<.., true, 2>: results are
// bit-identical (tests/test_gpu_conv.py::test_persistent_tiles_equal_one_tile_per_workgroup).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN)
    __attribute__((amdgpu_waves_per_eu(2, (BM * BN >= 128 * 128 ? 2 : (BM * BN >= 128 * 64 ? 3 : 4))))) void igemm_pt_kernel(
        IgemmArgs a, int ntiles, int run) {
  constexpr int NT = 64 * WM * WN;              // threads
  constexpr int RPP = NT / 8;                   // rows per load pass (8 threads x float4 cover a 32-float row)
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;   // MFMA tiles per wave
  constexpr int AP = BM / RPP, BP = BN / RPP;   // load passes
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % RPP == 0 && BN % RPP == 0 && RPP % 16 == 0, "tile shape");

  // hi / lo planes of A and B, two stages each: [plane][stage][row][32 bf16], 16-byte chunks XOR-swizzled (igemm_kernel)
  __shared__ __attribute__((aligned(16))) unsigned sm[2 * (BM + BN) * 32];
  constexpr int PA_HI = 0, PA_LO = 2 * BM * 16, PB_HI = 4 * BM * 16, PB_LO = 4 * BM * 16 + 2 * BN * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.OCg + BN - 1) / BN;
  const int tiles_m = (a.M - a.m_base + BM - 1) / BM;
  const int tpg = tiles_m * tiles_n;            // tiles per group
  // workgroups on one XCD (ids b, b + 8, ..) take neighbouring runs: the weight tiles and halo rows they share hit one L2
  int wg = blockIdx.x;
  if (a.xcd_swizzle) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int t_begin = wg * run, t_end = min(ntiles, t_begin + run);
  if (t_begin >= t_end) return;
  const int nk = a.ksteps;

  // rows of this launch are the input pixels themselves (1x1, stride 1, no padding: forward and data gradient): no
  // division anywhere in the tile geometry
  const bool pw = a.nr == 1 && a.ns == 1 && a.ihmul == 1 && a.iwmul == 1 && a.ihadd == 0 && a.iwadd == 0 &&
                  a.OHp == a.IH && a.OWp == a.IW;

  // ---- the load cursor: tile, k-step inside it, tap position, per-thread row geometry of THAT tile -----------------
  const int lrow = tid >> 3;        // 0..RPP-1
  const int lcol = (tid & 7) * 4;   // 0,4,..,28
  unsigned a_off[AP], b_off[BP];
  int a_h[AP], a_w[AP];
  int lt = t_begin, lstep = 0, l_tr = 0, l_ts = 0, l_cb = 0;
  const int cb_end = a.ksteps_per_tap * BK;
  auto geom = [&](int t) {
    const int g = t / tpg, r = t - g * tpg, tm = r / tiles_n, tn = r - tm * tiles_n;
    const int m0 = a.m_base + tm * BM, n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = m0 + i * RPP + lrow;
      const bool ok = m < a.M;
      const int mm = ok ? m : 0;
      if (pw) {
        a_h[i] = ok ? 0 : -(1 << 28);                          // a row past M never passes the range check
        a_w[i] = 0;
        a_off[i] = (unsigned)(mm * a.Ctot + g * a.CgR + lcol) * 4u;
      } else {
        const int jj = mm % a.OWp, t2 = mm / a.OWp;
        const int ii = t2 % a.OHp, n = t2 / a.OHp;
        a_h[i] = ok ? ii * a.ihmul + a.ihadd : -(1 << 28);
        a_w[i] = jj * a.iwmul + a.iwadd;
        a_off[i] = (unsigned)(((n * a.IH + (ok ? a_h[i] : 0)) * a.IW + a_w[i]) * a.Ctot + g * a.CgR + lcol) * 4u;
      }
      asm volatile("" : "+v"(a_off[i]), "+v"(a_h[i]), "+v"(a_w[i]));
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int oc = n0 + i * RPP + lrow;
      b_off[i] = oc < a.OCg ? (unsigned)((g * a.OCg + oc) * a.R * a.S * a.CgR + lcol) * 4u : B_INVALID;
      asm volatile("" : "+v"(b_off[i]));
    }
    l_tr = 0; l_ts = 0; l_cb = 0;
  };

  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, a.in_bytes), rs_wm = make_rsrc(a.wm, a.wm_bytes);
  auto load_tile = [&](float4 (&ra)[AP], float4 (&rb)[BP]) {
    const bool live = lt < t_end;                            // the cursor ran past the run: every lane masked
    const int cb = l_cb;                                     // first reduction channel of this k-step (uniform)
    const int dh = l_tr * a.hstep, dw = l_ts * a.wstep;
    const unsigned wtap = (unsigned)(((a.r0 + l_tr * a.rstep) * a.S + a.s0 + l_ts * a.sstep) * a.CgR + cb) * 4u;
    const unsigned aoff = (unsigned)((dh * a.IW + dw) * a.Ctot + cb) * 4u;
    const bool c_ok = live & (cb + lcol < a.CgR);
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const bool ok = c_ok & ((unsigned)(a_h[i] + dh) < (unsigned)a.IH) & ((unsigned)(a_w[i] + dw) < (unsigned)a.IW);
      ra[i] = bload4(rs_in, ok ? a_off[i] + aoff : OOB_OFF);
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) rb[i] = bload4(rs_wm, c_ok ? b_off[i] + wtap : OOB_OFF);
    // advance: channel block, tap column, tap row, and -- behind the tile's last k-step -- the next tile of the run
    {
      const int ncb = l_cb + BK;
      const bool wrap_c = ncb >= cb_end;
      const int nts = l_ts + (wrap_c ? 1 : 0);
      const bool wrap_s = nts == a.ns;
      l_cb = wrap_c ? 0 : ncb;
      l_ts = wrap_s ? 0 : nts;
      l_tr += wrap_s ? 1 : 0;
    }
    if (++lstep == nk) {
      lstep = 0;
      ++lt;
      if (lt < t_end) geom(lt);
    }
  };
  const int w_sw = ((((lcol >> 3) ^ ((lrow >> 2) & 3)) << 2) | ((lcol >> 1) & 2));   // dword offset inside the row
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
      const int o = (buf * BN + i * RPP + lrow) * 16 + w_sw;
      *(uint2*)(sm + PB_HI + o) = make_uint2(__float_as_uint(rb[i].x), __float_as_uint(rb[i].y));
      *(uint2*)(sm + PB_LO + o) = make_uint2(__float_as_uint(rb[i].z), __float_as_uint(rb[i].w));
    }
  };

  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  zero_acc();

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

  // ---- epilogue of one tile, straight from the accumulators --------------------------------------------------------
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  const bool dense_rows = a.osh == 1 && a.osw == 1 && a.OHp == a.OH && a.OWp == a.OW;
  const bool need_pos = !dense_rows || (a.res && a.res_mode == 1);
  // Branch-free: every access goes through a buffer descriptor and a lane outside the tensor (row past M, channel past
  // the group) gets an out-of-range offset -- its loads return zeros and its store is dropped.
  const unsigned out_bytes = (unsigned)((size_t)a.N * a.OH * a.OW * a.OCtot * 4);
  const unsigned res_bytes = a.res_mode == 0 ? out_bytes
                                             : (unsigned)((size_t)a.N * ((a.OH + 1) / 2) * ((a.OW + 1) / 2) * a.OCtot * 4);
  const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(a.out, out_bytes), rs_res = make_rsrc(a.res, a.res ? res_bytes : 0u),
                               rs_mask = make_rsrc(a.mask, a.mask ? out_bytes : 0u);
  auto epilogue = [&](int t) {
    const int g = t / tpg, r = t - g * tpg, tm = r / tiles_n, tn = r - tm * tiles_n;
    const int m0 = a.m_base + tm * BM + wm * WTM, n0 = tn * BN + wn * WTN;
    int oc[TN];
    bool okc[TN];
    float e_sc[TN], e_sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ocl = n0 + j * 32 + ecol;
      okc[j] = ocl < a.OCg;
      oc[j] = g * a.OCg + ocl;
      e_sc[j] = (a.scale && okc[j]) ? a.scale[oc[j]] : 1.f;
      e_sh[j] = (a.shift && okc[j]) ? a.shift[oc[j]] : 0.f;
    }
    // groups of 8 accumulator rows (half a 32-row slab): every residual / gate value of a group is requested before its
    // first store -- the compiler cannot move a load above a store that may alias it (in the accumulating data gradient
    // the residual IS the output) -- and a group's temporaries (8 x TN offsets, residuals, gates) stay small enough to
    // live beside the accumulators AND both operand register sets, whose loads are in flight right now
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned bo[8][TN];                    // byte offset of the output element, OOB_OFF outside the tensor
        float rv[8][TN], gv[8][TN];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int e = h * 8 + q;
          const int m = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + erow0;
          const bool ok = m < a.M;
          int orow = m, rrow = 0;
          if (need_pos) {
            const int mm = ok ? m : a.m_base;
            const int jj = mm % a.OWp, t2 = mm / a.OWp;
            const int ii = t2 % a.OHp, n = t2 / a.OHp;
            const int oh = ii * a.osh + a.oah, ow = jj * a.osw + a.oaw;
            orow = (n * a.OH + oh) * a.OW + ow;
            rrow = (n * ((a.OH + 1) / 2) + oh / 2) * ((a.OW + 1) / 2) + ow / 2;
          }
          const int rbase = a.res_mode == 0 ? orow * a.OCtot : rrow * a.OCtot;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const bool live = ok & okc[j];
            bo[q][j] = live ? (unsigned)(orow * a.OCtot + oc[j]) * 4u : OOB_OFF;
            if (a.res)
              rv[q][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                  rs_res, (int)(live ? (unsigned)(rbase + oc[j]) * 4u : OOB_OFF), 0, 0));
          }
        }
        if (a.mask) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              gv[q][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_mask, (int)bo[q][j], 0, 0));
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v = acc[i][j][h * 8 + q];
            // (the two forms round like igemm_kernel's two epilogues: its residual path contracts scale and shift
            // into one fma, its direct path applies them under their flags)
            if (a.res) {
              v = __builtin_fmaf(v, e_sc[j], e_sh[j]) + rv[q][j];
            } else {
              if (a.scale) v *= e_sc[j];
              if (a.shift) v += e_sh[j];
            }
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.mask) v = gv[q][j] > 0.f ? v : 0.f;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, (int)bo[q][j], 0, 0);
          }
        }
      }
    }
  };

  // ---- the pipeline ---------------------------------------------------------------------------------------------
  // One step: issue the loads of the tile two steps ahead, multiply the first half of the current one, move the tile one
  // step ahead (loaded a whole step ago) from registers to the other LDS stage, multiply the second half, barrier; behind
  // a tile's last k-step its epilogue runs -- the operand tiles of the next tile's first two steps are in flight or in LDS.
  int ct = t_begin, cstep = 0;
  auto step = [&](int cur, float4 (&la)[AP], float4 (&lb)[BP], const float4 (&sa)[AP], const float4 (&sb)[BP]) {
    load_tile(la, lb);
    Frag f0, f1;
    fetch(cur, 0, f0);
    fetch(cur, 1, f1);
    mfma3(f0);
    store_tile(cur ^ 1, sa, sb);          // unconditional (the last step re-stores a stale tile into the idle stage)
    mfma3(f1);
    constexpr int NM = TM * TN * 3 * 2;                      // MFMAs of the step
    constexpr int VPM = (AP * 12 + (AP + BP) * 4 + NM - 1) / NM;
    constexpr int WEVERY = NM / (AP + BP) > 0 ? NM / (AP + BP) : 1;
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      if (m % WEVERY == WEVERY - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __syncthreads();
    if (++cstep == nk) {
      epilogue(ct);
      zero_acc();
      ++ct;
      cstep = 0;
    }
  };

  float4 ra0[AP], rb0[BP], ra1[AP], rb1[BP];
  geom(t_begin);
  load_tile(ra0, rb0);
  store_tile(0, ra0, rb0);
  load_tile(ra1, rb1);
  __syncthreads();
  const int total_steps = (t_end - t_begin) * nk;
  for (int s = 0; s < total_steps; s += 2) {
    step(0, ra0, rb0, ra1, rb1);                         // tile s in stage 0; tile s + 1 waits in register set 1
    if (s + 1 < total_steps) step(1, ra1, rb1, ra0, rb0);
  }
}

// ---- 3x3 stride-1 convolution with the input patch staged ONCE per channel block (bf16x3 arithmetic) ----------
// The generic kernel gathers (and splits, and stores to LDS) an A tile per tap: a 3x3 layer moves every input
// element 9 x (K / BN) times through that path.  Here a workgroup owns an 8 x 16 patch of output pixels (128 GEMM
// rows); per 32-channel block it stages the 10 x 18 input halo once and runs the 9 taps as shifted views of it:
// A-side global loads, bf16 splits and LDS stores drop by 6.4x, the B (weight) path is unchanged.
// Serves forward and the stride-1 data gradient (both are "output (y,x) reads input (y+dy, x+dx)", the gather
// description supplies dy/dx per tap and the weight tap).  Conditions in halo_eligible().
constexpr int HT_H = 8, HT_W = 16, HALO_W = HT_W + 2, HALO_ROWS = (HT_H + 2) * (HT_W + 2);     // 180 halo pixels

template <int BN, bool BPRE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void igemm3x3_kernel(IgemmArgs a) {
  constexpr int BM = HT_H * HT_W;                 // 128
  constexpr int WM = 2, WN = 2, WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int RPP = 32, BP = BN / RPP;          // B load passes (8 threads x float4 per 32-float row)
  constexpr int AL = (HALO_ROWS * 8 + 255) / 256; // A halo loads per thread per channel block (6)
  constexpr int LDS_DW = 2 * HALO_ROWS * 32 + 2 * BN * 32;
  __shared__ __attribute__((aligned(16))) unsigned sm[LDS_DW];
  constexpr int PA_HI = 0, PA_LO = 2 * HALO_ROWS * 16, PB_HI = 4 * HALO_ROWS * 16, PB_LO = PB_HI + 2 * BN * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.OCg + BN - 1) / BN;
  const int patches_x = (a.OW + HT_W - 1) / HT_W, patches_y = (a.OH + HT_H - 1) / HT_H;
  int bid = blockIdx.x;
  if (a.xcd_swizzle) {                                // see igemm_kernel: contiguous logical ids per XCD, so that the
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;     // N-tiles of a patch and neighbouring
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);   // patches (shared halo) share an L2
  }
  const int tile_n = bid % tiles_n; bid /= tiles_n;
  const int pxi = bid % patches_x; bid /= patches_x;
  const int pyi = bid % patches_y;
  const int n = bid / patches_y;
  const int oy0 = pyi * HT_H, ox0 = pxi * HT_W, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, a.in_bytes), rs_wm = make_rsrc(a.wm, a.wm_bytes);
  const int CB = a.Ctot / 32, T = CB * 9;

  // ---- A: halo slots of this thread (halo pixel hp, channel quad cq), fixed over the channel blocks
  unsigned ha_off[AL];
  int ha_lds[AL];
#pragma unroll
  for (int j = 0; j < AL; ++j) {
    const int id = tid + 256 * j;
    const int hp = id >> 3, cq = id & 7;
    const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
    const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
    const bool ok = id < HALO_ROWS * 8 && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
    ha_off[j] = ok ? (unsigned)(((n * a.IH + iy) * a.IW + ix) * a.Ctot + cq * 4) * 4u : B_INVALID;
    ha_lds[j] = id < HALO_ROWS * 8 ? hp * 16 + ((((cq >> 1) ^ ((hp >> 2) & 3)) << 2) | ((cq & 1) << 1)) : -1;
    asm volatile("" : "+v"(ha_off[j]), "+v"(ha_lds[j]));
  }
  // ---- B: as in igemm_kernel
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  unsigned b_off[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int oc = n0 + i * RPP + lrow;
    b_off[i] = oc < a.OCg ? (unsigned)(oc * 9 * a.CgR + lcol) * 4u : B_INVALID;
    asm volatile("" : "+v"(b_off[i]));
  }
  const int w_sw = ((((lcol >> 3) ^ ((lrow >> 2) & 3)) << 2) | ((lcol >> 1) & 2));

  float4 ha[AL], rb0[BP], rb1[BP];
  auto load_a = [&](int cb) {
#pragma unroll
    for (int j = 0; j < AL; ++j) ha[j] = bload4(rs_in, ha_off[j] + (unsigned)cb * 128u);   // invalid base stays >= 2 GiB
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < AL; ++j) {
      if (ha_lds[j] < 0) continue;
      uint2 hi, lo;
      split4(ha[j], hi, lo);
      const int o = buf * HALO_ROWS * 16 + ha_lds[j];
      *(uint2*)(sm + PA_HI + o) = hi;
      *(uint2*)(sm + PA_LO + o) = lo;
    }
  };
  auto load_b = [&](int t, float4 (&rb)[BP]) {
    const int cb = t / 9, tap = t - cb * 9;
    const int tr = tap / 3, ts = tap - tr * 3;
    const unsigned wtap = (unsigned)(((a.r0 + tr * a.rstep) * a.S + a.s0 + ts * a.sstep) * a.CgR + cb * 32) * 4u;
#pragma unroll
    for (int i = 0; i < BP; ++i) rb[i] = bload4(rs_wm, b_off[i] + wtap);
  };
  auto store_b = [&](int buf, const float4 (&rb)[BP]) {
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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // The lane's A rows as halo indices of the centre tap.  GEMM row ml of the tile is output pixel (y = ml >> 4,
  // x = col_of(ml)): on odd patch rows the columns are rotated by 2.  A ds_read_b128 is served in lane groups
  // {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (MI355X_MICROARCH.md, LDS): 8 pixels of patch row y and 8 of row y+1,
  // whose halo rows are 18 apart -- with x = ml & 15 on both rows two of the 16 land on the same banks (27 % of this
  // kernel's LDS cycles were conflict cycles); 18 + the rotation = 16 puts them on complementary residues mod 16.
  auto col_of = [](int ml) { return (ml & 16) ? ((ml & 15) - 2) & 15 : (ml & 15); };
  const int frow = lane & 31, khalf = lane >> 5;
  int mh[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ml = wm * WTM + i * 32 + frow;
    mh[i] = ((ml >> 4) + 1) * HALO_W + col_of(ml) + 1;
  }
  struct Frag { bf16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
  auto fetch = [&](int abuf, int bbuf, int hoff, int sub, Frag& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int h = mh[i] + hoff;
      const int o = (abuf * HALO_ROWS + h) * 16 + ((((sub * 2 + khalf) ^ ((h >> 2) & 3))) << 2);
      f.ah[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PA_HI + o));
      f.al[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PA_LO + o));
    }
    const int r_sw = (((sub * 2 + khalf) ^ ((frow >> 2) & 3)) << 2);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int o = (bbuf * BN + wn * WTN + j * 32 + frow) * 16 + r_sw;
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

  // prologue: halo of channel block 0, B tiles of steps 0 and 1
  load_a(0);
  load_b(0, rb0);
  store_a(0);
  store_b(0, rb0);
  if (T > 1) load_b(1, rb1);
  __syncthreads();

  auto step = [&](int t, float4 (&lb)[BP], const float4 (&sb)[BP]) {
    const int cb = t / 9, tap = t - cb * 9;
    const int tr = tap / 3, ts = tap - tr * 3;
    const int hoff = (a.ihadd + tr * a.hstep) * HALO_W + (a.iwadd + ts * a.wstep);
    if (t + 2 < T) load_b(t + 2, lb);
    if (tap == 0 && cb + 1 < CB) load_a(cb + 1);            // lands while this block's first taps run
    Frag f0, f1;
    fetch(cb & 1, t & 1, hoff, 0, f0);
    fetch(cb & 1, t & 1, hoff, 1, f1);
    mfma3(f0);
    store_b((t + 1) & 1, sb);                               // B tile of step t+1 (stale re-store on the last step)
    if (tap == 4 && cb + 1 < CB) store_a((cb + 1) & 1);     // the other halo buffer: last read in block cb-1
    mfma3(f1);
    // lay the B split (48 VALU ops) and its stores into the issue gaps of the 24 MFMAs, as in igemm_kernel
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
#pragma unroll
    for (int m = 0; m < TM * TN * 6; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      if (m % 6 == 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __syncthreads();
  };
  for (int t = 0; t < T; t += 2) {
    step(t, rb0, rb1);
    if (t + 1 < T) step(t + 1, rb1, rb0);
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  // The gate / residual operands of a 32-row slab are fetched FIRST (16 x TN independent loads in flight), then the
  // slab is finished and stored: interleaved with the stores, every load waited out its full latency (the compiler
  // cannot move a load across a store that may alias it) -- the gated data gradient ran 1.5x slower than the forward.
  float e_sc[TN], e_sh[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int oc = n0 + wn * WTN + j * 32 + ecol;
    e_sc[j] = (a.scale && oc < a.OCg) ? a.scale[oc] : 1.f;
    e_sh[j] = (a.shift && oc < a.OCg) ? a.shift[oc] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float gate[16][TN], resv[16][TN];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ml = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0;
      const int oy = oy0 + (ml >> 4), ox = ox0 + col_of(ml);
      const bool row_ok = oy < a.OH && ox < a.OW;
      const size_t orow = ((size_t)n * a.OH + oy) * a.OW + ox;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int oc = n0 + wn * WTN + j * 32 + ecol;
        const bool ok = row_ok && oc < a.OCg;
        gate[e][j] = (a.mask && ok) ? a.mask[orow * a.OCtot + oc] : 1.f;
        float r = 0.f;
        if (a.res && ok) {
          if (a.res_mode == 0) {
            r = a.res[orow * a.OCtot + oc];
          } else {
            const int rh = (a.OH + 1) / 2, rw = (a.OW + 1) / 2;
            r = a.res[((size_t)(n * rh + oy / 2) * rw + ox / 2) * a.OCtot + oc];
          }
        }
        resv[e][j] = r;
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ml = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0;
      const int oy = oy0 + (ml >> 4), ox = ox0 + col_of(ml);
      if (oy >= a.OH || ox >= a.OW) continue;
      const size_t orow = ((size_t)n * a.OH + oy) * a.OW + ox;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int oc = n0 + wn * WTN + j * 32 + ecol;
        if (oc >= a.OCg) continue;
        float v = acc[i][j][e];
        if (a.scale) v *= e_sc[j];
        if (a.shift) v += e_sh[j];
        v += resv[e][j];
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.mask) v = gate[e][j] > 0.f ? v : 0.f;
        a.out[orow * a.OCtot + oc] = v;
      }
    }
  }
}

// ---- the same for the RoI heads' 7x7 maps: igemm3x3_roi_kernel -----------------------------------------------------
// The grid head is eight 3x3 convolutions over 576 channels on R x 7 x 7 maps: GEMM rows are dense (M = 49 R), but a
// map's neighbours are its own zero border, never the next RoI.  A workgroup owns 128 consecutive GEMM rows -- pixels
// of at most FOUR RoIs -- and stages their 9x9 halos (324 halo rows, border cells and RoIs beyond the batch read as
// zeros) once per 32-channel block; the nine taps are shifted views of it: the A side moves 324 rows per nine k-steps
// instead of 9 x 128 (3.5x fewer loads, splits and LDS stores), the B side is igemm_kernel's.  ONE halo stage (41 KB) +
// two B stages (32 KB): two workgroups per CU; the next block's halo waits in registers and is stored between two
// barriers behind the block's last tap.  Forward and the stride-1 data gradient; conditions in roi_halo_eligible().
constexpr int RH_S = 7, RH_HW = RH_S + 2, RH_CELLS = RH_HW * RH_HW, RH_PX = RH_S * RH_S;       // 9, 81, 49
constexpr int RH_ROIS = 4, RH_ROWS = RH_ROIS * RH_CELLS;                                         // 324 halo rows

template <int BN, bool BPRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void igemm3x3_roi_kernel(IgemmArgs a) {
  constexpr int BM = 128;
  constexpr int WM = 2, WN = 2, WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int RPP = 32, BP = BN / RPP;
  constexpr int AL = (RH_ROWS * 8 + 255) / 256;   // halo loads per thread per channel block (11)
  constexpr int LDS_DW = RH_ROWS * 32 + 2 * BN * 32;
  __shared__ __attribute__((aligned(16))) unsigned sm[LDS_DW];
  constexpr int PA_HI = 0, PA_LO = RH_ROWS * 16, PB_HI = RH_ROWS * 32, PB_LO = PB_HI + 2 * BN * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.OCg + BN - 1) / BN;
  int bid = blockIdx.x;
  if (a.xcd_swizzle) {                                // contiguous logical ids per XCD: the N-tiles of a row block share an L2
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int r_first = m0 / RH_PX;

  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, a.in_bytes), rs_wm = make_rsrc(a.wm, a.wm_bytes);
  const int CB = a.Ctot / 32, T = CB * 9;

  // ---- A: halo slots of this thread (halo row hp = RoI * 81 + cell, channel quad cq), fixed over the channel blocks
  unsigned ha_off[AL];
  int ha_lds[AL];
#pragma unroll
  for (int j = 0; j < AL; ++j) {
    const int id = tid + 256 * j;
    const int hp = id >> 3, cq = id & 7;
    const int r = hp / RH_CELLS, c = hp - r * RH_CELLS;
    const int hy = c / RH_HW, hx = c - hy * RH_HW;
    const int n = r_first + r, iy = hy - 1, ix = hx - 1;
    const bool ok = id < RH_ROWS * 8 && n < a.N && (unsigned)iy < (unsigned)RH_S && (unsigned)ix < (unsigned)RH_S;
    ha_off[j] = ok ? (unsigned)(((n * RH_S + iy) * RH_S + ix) * a.Ctot + cq * 4) * 4u : B_INVALID;
    ha_lds[j] = id < RH_ROWS * 8 ? hp * 16 + ((((cq >> 1) ^ ((hp >> 2) & 3)) << 2) | ((cq & 1) << 1)) : -1;
    asm volatile("" : "+v"(ha_off[j]), "+v"(ha_lds[j]));
  }
  // ---- B: as in igemm_kernel
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  unsigned b_off[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int oc = n0 + i * RPP + lrow;
    b_off[i] = oc < a.OCg ? (unsigned)(oc * 9 * a.CgR + lcol) * 4u : B_INVALID;
    asm volatile("" : "+v"(b_off[i]));
  }
  const int w_sw = ((((lcol >> 3) ^ ((lrow >> 2) & 3)) << 2) | ((lcol >> 1) & 2));

  float4 ha[AL], rb0[BP], rb1[BP];
  auto load_a = [&](int cb) {
#pragma unroll
    for (int j = 0; j < AL; ++j) ha[j] = bload4(rs_in, ha_off[j] + (unsigned)cb * 128u);   // invalid base stays >= 2 GiB
  };
  auto store_a = [&]() {
#pragma unroll
    for (int j = 0; j < AL; ++j) {
      if (ha_lds[j] < 0) continue;
      uint2 hi, lo;
      split4(ha[j], hi, lo);
      *(uint2*)(sm + PA_HI + ha_lds[j]) = hi;
      *(uint2*)(sm + PA_LO + ha_lds[j]) = lo;
    }
  };
  auto load_b = [&](int t, float4 (&rb)[BP]) {
    const int cb = t / 9, tap = t - cb * 9;
    const int tr = tap / 3, ts = tap - tr * 3;
    const unsigned wtap = (unsigned)(((a.r0 + tr * a.rstep) * a.S + a.s0 + ts * a.sstep) * a.CgR + cb * 32) * 4u;
#pragma unroll
    for (int i = 0; i < BP; ++i) rb[i] = bload4(rs_wm, b_off[i] + wtap);
  };
  auto store_b = [&](int buf, const float4 (&rb)[BP]) {
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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // the lane's A rows as halo rows of the centre tap; GEMM rows beyond M read an interior cell of the first halo
  const int frow = lane & 31, khalf = lane >> 5;
  int mh[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 32 + frow;
    const int r = m / RH_PX, px = m - r * RH_PX;
    const int y = px / RH_S, x = px - y * RH_S;
    mh[i] = m < a.M ? (r - r_first) * RH_CELLS + (y + 1) * RH_HW + x + 1 : RH_HW + 1;
  }
  struct Frag { bf16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
  auto fetch = [&](int bbuf, int hoff, int sub, Frag& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int h = mh[i] + hoff;
      const int o = h * 16 + ((((sub * 2 + khalf) ^ ((h >> 2) & 3))) << 2);
      f.ah[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PA_HI + o));
      f.al[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(sm + PA_LO + o));
    }
    const int r_sw = (((sub * 2 + khalf) ^ ((frow >> 2) & 3)) << 2);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int o = (bbuf * BN + wn * WTN + j * 32 + frow) * 16 + r_sw;
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

  // prologue: halo of channel block 0, B tiles of steps 0 and 1
  load_a(0);
  load_b(0, rb0);
  store_a();
  store_b(0, rb0);
  if (T > 1) load_b(1, rb1);
  __syncthreads();

  auto step = [&](int t, float4 (&lb)[BP], const float4 (&sb)[BP]) {
    const int cb = t / 9, tap = t - cb * 9;
    const int tr = tap / 3, ts = tap - tr * 3;
    const int hoff = (a.ihadd + tr * a.hstep) * RH_HW + (a.iwadd + ts * a.wstep);
    if (t + 2 < T) load_b(t + 2, lb);
    if (tap == 4 && cb + 1 < CB) load_a(cb + 1);            // lands while the block's last taps run
    Frag f0, f1;
    fetch(t & 1, hoff, 0, f0);
    fetch(t & 1, hoff, 1, f1);
    mfma3(f0);
    store_b((t + 1) & 1, sb);                               // B tile of step t+1 (stale re-store on the last step)
    mfma3(f1);
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
#pragma unroll
    for (int m = 0; m < TM * TN * 6; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      if (m % 6 == 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __syncthreads();
    if (tap == 8 && cb + 1 < CB) {                          // every wave is done with this block's halo
      store_a();
      __syncthreads();
    }
  };
  for (int t = 0; t < T; t += 2) {
    step(t, rb0, rb1);
    if (t + 1 < T) step(t + 1, rb1, rb0);
  }

  // ---- epilogue (rows are dense: GEMM row m is output row m)
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  float e_sc[TN], e_sh[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int oc = n0 + wn * WTN + j * 32 + ecol;
    e_sc[j] = (a.scale && oc < a.OCg) ? a.scale[oc] : 1.f;
    e_sh[j] = (a.shift && oc < a.OCg) ? a.shift[oc] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float gate[16][TN], resv[16][TN];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const size_t orow = (size_t)(m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int oc = n0 + wn * WTN + j * 32 + ecol;
        const bool ok = orow < (size_t)a.M && oc < a.OCg;
        gate[e][j] = (a.mask && ok) ? a.mask[orow * a.OCtot + oc] : 1.f;
        resv[e][j] = (a.res && ok) ? a.res[orow * a.OCtot + oc] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const size_t orow = (size_t)(m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0);
      if (orow >= (size_t)a.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int oc = n0 + wn * WTN + j * 32 + ecol;
        if (oc >= a.OCg) continue;
        float v = acc[i][j][e];
        if (a.scale) v *= e_sc[j];
        if (a.shift) v += e_sh[j];
        v += resv[e][j];
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.mask) v = gate[e][j] > 0.f ? v : 0.f;
        a.out[orow * a.OCtot + oc] = v;
      }
    }
  }
}

// epilogue as a separate pass (after split-K atomics)
__global__ void epilogue_kernel(float* __restrict__ out, const float* __restrict__ scale,
                                const float* __restrict__ shift, const float* __restrict__ res, int64_t M, int OC,
                                int OH, int OW, int res_mode, int relu, const float* __restrict__ mask = nullptr) {
  const int64_t total = M * OC;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int oc = idx % OC;
    const int64_t m = idx / OC;
    float v = out[idx];
    v = v * (scale ? scale[oc] : 1.f) + (shift ? shift[oc] : 0.f);
    if (res) {
      if (res_mode == 0) {
        v += res[idx];
      } else {
        const int ow = m % OW;
        const int64_t t = m / OW;
        const int oh = t % OH;
        const int64_t n = t / OH;
        const int rh = (OH + 1) / 2, rw = (OW + 1) / 2;
        v += res[((n * rh + oh / 2) * rw + ow / 2) * OC + oc];
      }
    }
    if (relu) v = fmaxf(v, 0.f);
    if (mask) v = mask[idx] > 0.f ? v : 0.f;
    out[idx] = v;
  }
}

// split-K, slab form: out[m][oc] = epilogue( (acc ? out : 0) + slab[0] + slab[1] + ... ) in split order -- one pass
// instead of zero-fill + float atomics + epilogue pass, and bit-reproducible
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, size_t stride,
                                                            float* __restrict__ out, int accumulate,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ res, int64_t M, int OC, int OH,
                                                            int OW, int res_mode, int relu,
                                                            const float* __restrict__ mask) {
  const int64_t total4 = M * OC / 4;                      // OC % 4 == 0 (host)
  for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = i4 * 4;
    float4 v = *(const float4*)(slab + idx);
    for (int s = 1; s < splits; ++s) {
      const float4 t = *(const float4*)(slab + (size_t)s * stride + idx);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (accumulate) { const float4 o = *(const float4*)(out + idx); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    const int oc = (int)(idx % OC);
    const int64_t m = idx / OC;
    if (scale) { const float4 t = *(const float4*)(scale + oc); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
    if (shift) { const float4 t = *(const float4*)(shift + oc); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (res) {
      int64_t ri = idx;
      if (res_mode == 1) {
        const int ow = (int)(m % OW);
        const int64_t t = m / OW;
        const int oh = (int)(t % OH);
        const int64_t n = t / OH;
        ri = ((n * ((OH + 1) / 2) + oh / 2) * ((OW + 1) / 2) + ow / 2) * OC + oc;
      }
      const float4 t = *(const float4*)(res + ri);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (mask) {
      const float4 t = *(const float4*)(mask + idx);
      v.x = t.x > 0.f ? v.x : 0.f; v.y = t.y > 0.f ? v.y : 0.f; v.z = t.z > 0.f ? v.z : 0.f; v.w = t.w > 0.f ? v.w : 0.f;
    }
    *(float4*)(out + idx) = v;
  }
}

// the same for outputs whose channel count is not a multiple of 4 (cls_score: 81, iou_pred: 2, RPN predictors: 3 / 12)
__global__ __launch_bounds__(256) void splitk_reduce_scalar_kernel(const float* __restrict__ slab, int splits, size_t stride,
                                                                   float* __restrict__ out, int accumulate,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   const float* __restrict__ res, int64_t M, int OC,
                                                                   int OH, int OW, int res_mode, int relu,
                                                                   const float* __restrict__ mask) {
  const int64_t total = M * OC;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    float v = slab[idx];
    for (int s = 1; s < splits; ++s) v += slab[(size_t)s * stride + idx];
    if (accumulate) v += out[idx];
    const int oc = (int)(idx % OC);
    const int64_t m = idx / OC;
    v = v * (scale ? scale[oc] : 1.f) + (shift ? shift[oc] : 0.f);
    if (res) {
      if (res_mode == 0) {
        v += res[idx];
      } else {
        const int ow = (int)(m % OW);
        const int64_t t = m / OW;
        const int oh = (int)(t % OH);
        const int64_t n = t / OH;
        v += res[((n * ((OH + 1) / 2) + oh / 2) * ((OW + 1) / 2) + ow / 2) * OC + oc];
      }
    }
    if (relu) v = fmaxf(v, 0.f);
    if (mask) v = mask[idx] > 0.f ? v : 0.f;
    out[idx] = v;
  }
}

// split-K accumulator seeded with the bias: out[m][oc] = shift[oc] -- when the epilogue is the bias alone (conv + bias
// feeding a GroupNorm: the grid head), this replaces BOTH the zero fill before the atomics and the epilogue pass after
__global__ void __launch_bounds__(256) seed_rows_kernel(float* __restrict__ out, const float* __restrict__ shift,
                                                        int64_t total4, int oc4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x)
    ((float4*)out)[i] = ((const float4*)shift)[i % oc4];
}

// KRSC [K][R][S][Cg] (K = groups*Kg)  ->  DGRAD operand [groups*Cg][R][S][Kg]
// Per (group, tap) this is a Kg x Cg transpose.  32x32 tiles through LDS: the reads run along c and the writes along
// k, both as contiguous 128-byte rows (a thread-per-destination-element version reads with a stride of R*S*Cg floats
// and ran at a fraction of the copy rate on the 12-29 M element FC weights).
// k_scale ([groups*Kg] or null): the image of diag(k_scale) * W -- a frozen per-output-channel factor behind the conv
// (AffineChannel2d) is then applied by the data gradient's reduction itself: dx = W^T (scale * g).
// w4: the image is written pre-split (cpm_split_w4's format along k: Kg % 4 == 0), for the bf16x3 kernels.
__device__ __forceinline__ void wt_store_tile(const float (&tile)[32][33], float* __restrict__ wt, int g, int Cg, int RS,
                                              int t, int Kg, int c0, int k0, int w4) {
  if (w4) {
    const int cl = threadIdx.x >> 3, kq = (threadIdx.x & 7) * 4;       // one k-quad of one c row per thread
    const int c = c0 + cl, k = k0 + kq;
    if (c < Cg && k < Kg) {
      uint2 hi, lo;
      split4(make_float4(tile[kq][cl], tile[kq + 1][cl], tile[kq + 2][cl], tile[kq + 3][cl]), hi, lo);
      *(uint4*)(wt + (((int64_t)(g * Cg + c)) * RS + t) * Kg + k) = make_uint4(hi.x, hi.y, lo.x, lo.y);
    }
    return;
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int c = c0 + ty + j, k = k0 + tx;
    if (c < Cg && k < Kg) wt[(((int64_t)(g * Cg + c)) * RS + t) * Kg + k] = tile[tx][ty + j];
  }
}

__global__ __launch_bounds__(256) void weight_to_dgrad(const float* __restrict__ w, int groups, int Kg, int RS, int Cg,
                                                       float* __restrict__ wt, const float* __restrict__ k_scale, int w4) {
  __shared__ float tile[32][33];
  const int tiles_c = (Cg + 31) / 32, tiles_k = (Kg + 31) / 32;
  const int64_t per_gt = (int64_t)tiles_c * tiles_k;
  const int64_t total = per_gt * RS * groups;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  for (int64_t b = blockIdx.x; b < total; b += gridDim.x) {
    const int tcx = (int)(b % tiles_c);
    int64_t r = b / tiles_c;
    const int tkx = (int)(r % tiles_k); r /= tiles_k;
    const int t = (int)(r % RS);
    const int g = (int)(r / RS);
    const int k0 = tkx * 32, c0 = tcx * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
      const int k = k0 + ty + j, c = c0 + tx;
      float v = (k < Kg && c < Cg) ? w[(((int64_t)(g * Kg + k)) * RS + t) * Cg + c] : 0.f;
      if (k_scale && k < Kg) v *= k_scale[g * Kg + k];
      tile[ty + j][tx] = v;
    }
    __syncthreads();
    wt_store_tile(tile, wt, g, Cg, RS, t, Kg, c0, k0, w4);
    __syncthreads();
  }
}

// The same transform for MANY weights in one launch (all conv weights of the flat parameter buffer, once per optimizer
// step instead of once per data-gradient call): a device table gives every weight's offsets, shape and the index of
// its first 32x32 tile; a workgroup finds its weight by binary search over those tile starts.
__global__ __launch_bounds__(256) void weights_to_dgrad_batched(const cpm_wt_desc* __restrict__ descs, int n,
                                                                int64_t total_tiles, const float* __restrict__ src,
                                                                float* __restrict__ dst, int w4_all) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int64_t b = blockIdx.x; b < total_tiles; b += gridDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                          // last descriptor with tile_start <= b
      const int mid = (lo + hi + 1) >> 1;
      if (descs[mid].tile_start <= b) lo = mid; else hi = mid - 1;
    }
    const cpm_wt_desc d = descs[lo];
    const float* w = src + d.src_off;
    float* wt = dst + d.dst_off;
    const int Kg = d.Kg, Cg = d.Cg, RS = d.RS;
    const int w4 = w4_all && (Kg & 3) == 0;                  // (a weight with Kg % 4 != 0 keeps the f32 image)
    const int tiles_c = (Cg + 31) / 32, tiles_k = (Kg + 31) / 32;
    int64_t r = b - d.tile_start;
    const int tcx = (int)(r % tiles_c); r /= tiles_c;
    const int tkx = (int)(r % tiles_k); r /= tiles_k;
    const int t = (int)(r % RS);
    const int g = (int)(r / RS);
    const int k0 = tkx * 32, c0 = tcx * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
      const int k = k0 + ty + j, c = c0 + tx;
      float v = (k < Kg && c < Cg) ? w[(((int64_t)(g * Kg + k)) * RS + t) * Cg + c] : 0.f;
      if (d.k_scale && k < Kg) v *= d.k_scale[g * Kg + k];
      tile[ty + j][tx] = v;
    }
    __syncthreads();
    wt_store_tile(tile, wt, g, Cg, RS, t, Kg, c0, k0, w4);
    __syncthreads();
  }
}

// ---- weight gradient ---------------------------------------------------------------------------------
// dw[oc][tap][c] += sum_m dy[m][oc] * x[gather(m, tap)][c]
// tile: BM output channels x BN input channels for ONE tap; reduction over pixels in chunks of 32.
// LDS holds the two operands pixel-major ([32][BM+4], [32][BN+4]); a lane's MFMA operand is one
// ds_read_b32 at [pixel = 2*s + (lane>>5)][channel = lane & 31] (conflict-free: consecutive lanes,
// consecutive dwords).
struct WgradArgs {
  const float* x;   // [N][IH][IW][Ctot]
  const float* dy;  // [M][OCtot]
  float* dw;        // [OCtot][R][S][Cg]
  int N, IH, IW, Ctot, OH, OW, OCtot;
  int R, S, stride, pad, dil, groups, Cg, OCg, M;
  int split_k, chunks;  // chunks = ceil(M/32)
  unsigned x_bytes, dy_bytes;
  int debug_nostore;
  int xcd_swizzle;  // wgrad_split_kernel: give every XCD whole pixel splits (see the kernel)
  float* slab;      // split_k > 1: [split][OCtot][R][S][Cg] partial sums, one plane per reduction split, written with
                    // plain stores and folded into dw in split order by wgrad_reduce_kernel (deterministic; the
                    // float-atomic epilogue it replaces cost 23 % of the weight-gradient time).  null: see the epilogues
  size_t slab_stride;   // floats per plane
  float* dshift;    // [OCtot] or null: += sum over pixels of dy (the bias gradient), folded into the dy reads of the
                    // workgroups that own tap 0 / input-channel tile 0 (every dy element passes exactly one of them)
  const float* row_scale;   // [OCtot] or null: dw[oc] += row_scale[oc] * (the sums) -- the frozen per-channel factor
                    // behind the conv (y = conv * scale + shift): dy arrives as the gradient at y, dw = scale * (dy^T x)
};

template <int BM, int BN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int PA = BM + 4, PB = BN + 4;
  constexpr int AV = BM / 4, BV = BN / 4;             // float4 per pixel row
  constexpr int APASS = (32 * AV + 255) / 256, BPASS = (32 * BV + 255) / 256;
  __shared__ __attribute__((aligned(16))) float As[2][32][PA];
  __shared__ __attribute__((aligned(16))) float Bs[2][32][PB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.Cg + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int tap = blockIdx.y;
  const int r = tap / a.S, s = tap - r * a.S;
  const int g = blockIdx.z / a.split_k, split = blockIdx.z % a.split_k;
  const int oc0 = tile_m * BM, c0 = tile_n * BN;

  const int per = (a.chunks + a.split_k - 1) / a.split_k;
  const int ch_begin = split * per, ch_end = min(a.chunks, ch_begin + per);
  const int nk = ch_end - ch_begin;

  float4 ra[APASS], rb[BPASS];
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.x, a.x_bytes), rs_dy = make_rsrc(a.dy, a.dy_bytes);
  const bool vec_a = (a.OCtot & 3) == 0 && (a.OCg & 3) == 0;
  const bool vec_b = (a.Ctot & 3) == 0 && (a.Cg & 3) == 0;

  // Each thread owns fixed (pixel-row, channel-vector) slots of the 32-pixel chunk; its pixel index advances by
  // 32 per chunk, so (n, oh, ow) is decomposed ONCE here and then stepped with precomputed carries -- no
  // integer division in the reduction loop.
  const int hw = a.OH * a.OW;
  const int dn = 32 / hw, rem = 32 % hw;
  const int dh = rem / a.OW, dwid = rem % a.OW;
  int b_n[BPASS], b_oh[BPASS], b_ow[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int pr = (tid + i * 256) / BV;
    const int m = ch_begin * 32 + pr;
    b_ow[i] = m % a.OW;
    const int t = m / a.OW;
    b_oh[i] = t % a.OH;
    b_n[i] = t / a.OH;
  }

  auto load_chunk = [&](int ch) {
    const int mbase = ch * 32;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int id = tid + i * 256;
      const int pr = id / AV, cv = (id % AV) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int m = mbase + pr;
      if (VEC) {
        const int ocl = oc0 + cv;
        const bool ok = pr < 32 && m < a.M && ocl < a.OCg;
        v = bload4(rs_dy, ok ? (unsigned)(m * a.OCtot + g * a.OCg + ocl) * 4u : OOB_OFF);
        // an output channel count that is not a multiple of 4 (18: DeformConvPack's offset predictor): the row's last
        // 16-byte load reaches into the next pixel's row (4-byte aligned loads are served at full width) -- zero those
        // components; never taken otherwise
        if (ocl + 3 >= a.OCg) {
          if (ocl + 1 >= a.OCg) v.y = 0.f;
          if (ocl + 2 >= a.OCg) v.z = 0.f;
          v.w = 0.f;
        }
      } else if (pr < 32 && m < a.M) {
        const int ocl = oc0 + cv;
        const size_t off = (size_t)m * a.OCtot + g * a.OCg + ocl;
        if (vec_a && ocl + 3 < a.OCg) {
          v = *(const float4*)(a.dy + off);
        } else {
          if (ocl < a.OCg) v.x = a.dy[off];
          if (ocl + 1 < a.OCg) v.y = a.dy[off + 1];
          if (ocl + 2 < a.OCg) v.z = a.dy[off + 2];
          if (ocl + 3 < a.OCg) v.w = a.dy[off + 3];
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int id = tid + i * 256;
      const int pr = id / BV, cv = (id % BV) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (VEC) {
        const int ih = b_oh[i] * a.stride - a.pad + r * a.dil, iw = b_ow[i] * a.stride - a.pad + s * a.dil;
        const int cl = c0 + cv;
        const bool ok = pr < 32 && b_n[i] < a.N && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW &&
                        cl < a.Cg;
        v = bload4(rs_x, ok ? (unsigned)(((b_n[i] * a.IH + ih) * a.IW + iw) * a.Ctot + g * a.Cg + cl) * 4u : OOB_OFF);
      } else if (pr < 32 && b_n[i] < a.N) {
        const int ih = b_oh[i] * a.stride - a.pad + r * a.dil, iw = b_ow[i] * a.stride - a.pad + s * a.dil;
        if ((unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW) {
          const int cl = c0 + cv;
          const size_t off = ((size_t)(b_n[i] * a.IH + ih) * a.IW + iw) * a.Ctot + g * a.Cg + cl;
          if (vec_b && cl + 3 < a.Cg) {
            v = *(const float4*)(a.x + off);
          } else {
            if (cl < a.Cg) v.x = a.x[off];
            if (cl + 1 < a.Cg) v.y = a.x[off + 1];
            if (cl + 2 < a.Cg) v.z = a.x[off + 2];
            if (cl + 3 < a.Cg) v.w = a.x[off + 3];
          }
        }
      }
      rb[i] = v;
      // advance this slot to the next chunk
      b_n[i] += dn; b_oh[i] += dh; b_ow[i] += dwid;
      if (b_ow[i] >= a.OW) { b_ow[i] -= a.OW; ++b_oh[i]; }
      if (b_oh[i] >= a.OH) { b_oh[i] -= a.OH; ++b_n[i]; }
    }
  };
  const bool do_bias = a.dshift != nullptr && tap == 0 && tile_n == 0;      // block-uniform
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);        // this thread's channel vector (256 % AV == 0: same for all i)
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int id = tid + i * 256;
      const int pr = id / AV, cv = (id % AV) * 4;
      if (pr < 32) *(float4*)&As[buf][pr][cv] = ra[i];
      if (do_bias && pr < 32) { bsum.x += ra[i].x; bsum.y += ra[i].y; bsum.z += ra[i].z; bsum.w += ra[i].w; }
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int id = tid + i * 256;
      const int pr = id / BV, cv = (id % BV) * 4;
      if (pr < 32) *(float4*)&Bs[buf][pr][cv] = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nk > 0) {
    load_chunk(ch_begin);
    store_chunk(0);
  }
  __syncthreads();
  const int fc = lane & 31, fh = lane >> 5;
  int cur = 0;
  for (int it = 0; it < nk; ++it) {
    if (it + 1 < nk) load_chunk(ch_begin + it + 1);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      float fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = As[cur][2 * ks + fh][wm * WTM + i * 32 + fc];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = Bs[cur][2 * ks + fh][wn * WTN + j * 32 + fc];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (it + 1 < nk) store_chunk(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  if (do_bias) {
    // threads tid % AV share a channel vector: fold the 256 / AV partial sums through LDS, one atomic per channel
    static_assert(256 % AV == 0 && (256 / AV) * BM <= 2 * 32 * PA, "bias reduction layout");
    float* sb = &As[0][0][0];
    *(float4*)&sb[(tid / AV) * BM + (tid % AV) * 4] = bsum;
    __syncthreads();
    if (tid < BM && oc0 + tid < a.OCg) {
      float t = 0.f;
#pragma unroll
      for (int p = 0; p < 256 / AV; ++p) t += sb[p * BM + tid];
      atomicAdd(a.dshift + g * a.OCg + oc0 + tid, t);
    }
  }

  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cl = c0 + wn * WTN + j * 32 + ecol;
    if (cl >= a.Cg) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ocl = oc0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0;
        if (ocl >= a.OCg) continue;
        const size_t o = ((size_t)(g * a.OCg + ocl) * a.R * a.S + tap) * a.Cg + cl;
        const float v = a.row_scale ? acc[i][j][e] * a.row_scale[g * a.OCg + ocl] : acc[i][j][e];
        if (a.slab) a.slab[(size_t)split * a.slab_stride + o] = v;                  // this split's plane
        else if (a.split_k == 1) a.dw[o] += v;                                      // the element's only writer
        else atomicAdd(a.dw + o, v);
      }
  }
}

// ---- weight gradient, 3-term split-bf16 arithmetic -----------------------------------------------------
// Same tiling as wgrad_kernel (BM output channels x BN input channels for ONE tap, pixels reduced in chunks of
// 32), but the bf16 MFMA wants a lane's 8 reduction elements (pixels) CONTIGUOUS for its channel, i.e. a
// channel-major LDS image, while memory is pixel-major (NHWC).  The transposition happens in registers: a thread
// owns one 4-channel vector x 4 consecutive pixels (four 16-byte loads, each 16-lane group reading 256 contiguous
// bytes), splits every value into hi/lo bf16 and writes, per channel, the 4 pixels as ONE 8-byte store.
// Channel c of the tile lives in LDS row (c % 4) * (BM/4) + c / 4, so that the 16 lanes of a store group (16
// consecutive channel vectors, same component) write 16 consecutive rows -- with the igemm chunk swizzle that is
// conflict-free for the stores and for the ds_read_b128 operand fetches alike.  The accumulator tile comes out
// row/column-permuted accordingly; it is un-permuted through LDS at the end so that the float atomics into dw
// are 256-byte contiguous per wave instruction.
// DBG (timing-only ablations, tools/wgrad_dbg.sh; results are wrong): 1 = every load masked (issued, nothing fetched),
// 2 = also no bf16 split and no LDS stores, 3 = also the operand reads hoisted out of the loop (MFMAs + barriers only)
// KS = 2: a workgroup of eight waves, waves 0-3 and 4-7 reducing the two halves of the workgroup's pixel range into
// their own accumulators (own LDS buffers, nothing shared in the loop), folded through LDS before the tile leaves: the
// same waves per CU as two independent workgroups, half the partial tiles -- the float atomics / slab planes a launch
// pays for (1.3 TB/s chip-wide: 32 MB for a 512-workgroup launch = 25 us) halve.
template <int BM, int BN, int WM, int WN, int DBG = 0, int KS = 1>
__global__ __launch_bounds__(256 * KS)
    __attribute__((amdgpu_waves_per_eu(2, (BM * BN >= 128 * 128 ? 2 : (BM * BN >= 128 * 64 ? 3 : 4))))) void wgrad_split_kernel(
        WgradArgs a) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int QA = BM / 4, QB = BN / 4;              // channel vectors per tile
  constexpr int CP = BN + 4;
  constexpr int LDS_AB = 2 * (BM + BN) * 32, LDS_C = BM * CP;
  static_assert(QA % 16 == 0 && QB % 16 == 0 && QA * 8 <= 256 && QB * 8 <= 256, "tile shape");
  __shared__ __attribute__((aligned(16))) float smem_all[KS * LDS_AB > LDS_C ? KS * LDS_AB : LDS_C];
  const int slice = KS == 1 ? 0 : (int)(threadIdx.x >> 8);
  float* const smem = smem_all + slice * LDS_AB;
  unsigned* const sm = reinterpret_cast<unsigned*>(smem);
  constexpr int PA_HI = 0, PA_LO = 2 * BM * 16, PB_HI = 4 * BM * 16, PB_LO = 4 * BM * 16 + 2 * BN * 16;

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.Cg + BN - 1) / BN;
  // XCD-aware order.  The (channel tile, tap) workgroups of one pixel split all stream the same dy and x chunks; in
  // launch order (x fastest, workgroups dealt round-robin over the 8 XCDs) they land on all eight L2s, each of which
  // then fetches those chunks from the Infinity Cache for itself (46 % of the kernel's L2 requests missed).  Give each
  // XCD a contiguous run of logical ids instead, i.e. whole splits: the sharers sit behind one L2.  Bijective for
  // any grid (the igemm kernels' formula); a speed hint only.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd_swizzle) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int lin = bx + gx * (by + gy * bz), nwg = gx * gy * gridDim.z;
    const int q = nwg >> 3, rr = nwg & 7, xcd = lin & 7;
    const int l2 = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (lin >> 3);
    bx = l2 % gx;
    by = (l2 / gx) % gy;
    bz = l2 / (gx * gy);
  }
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int tap = by;
  const int r = tap / a.S, s = tap - r * a.S;
  const int g = bz / a.split_k, split = bz % a.split_k;
  const int oc0 = tile_m * BM, c0 = tile_n * BN;

  // (KS = 2: split_k counts workgroups; the pixel range is cut into KS * split_k runs, two neighbours per workgroup)
  const int per = (a.chunks + KS * a.split_k - 1) / (KS * a.split_k);
  const int ch_begin = min(a.chunks, (split * KS + slice) * per), ch_end = min(a.chunks, ch_begin + per);
  const int nk = ch_end - ch_begin;
  const int nk_loop = KS == 1 ? nk : min(per, a.chunks - min(a.chunks, split * KS * per));   // slice 0's count: the longer one

  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.x, a.x_bytes), rs_dy = make_rsrc(a.dy, a.dy_bytes);

  // slot of this thread: channel vector q (16 consecutive per 16-lane group) and pixel run (4 pixels)
  // lane bits: [3:0] channel vector inside a block of 16, [4] low bit of the pixel run, [5..] vector block and the
  // run's high bits.  The 32 lanes one ds_write_b64 pass serves are then 16 consecutive rows x the two 8-byte
  // halves of one chunk: with the chunk swizzle that is every bank exactly once (pixel run in bit 4 of the row
  // index instead would put rows r and r+16 on the same banks: a 2-way conflict on every store).
  const int l16 = tid & 15, half_run = (tid >> 4) & 1, hi = tid >> 5;
  const int qa = (hi % (QA / 16)) * 16 + l16, pra = (hi / (QA / 16)) * 2 + half_run;   // pra < 8 <=> thread has an A slot
  const int qb = (hi % (QB / 16)) * 16 + l16, prb = (hi / (QB / 16)) * 2 + half_run;
  const bool a_act = pra < 8 && oc0 + 4 * qa < a.OCg;
  const bool b_act = prb < 8 && c0 + 4 * qb < a.Cg;
  unsigned a_off = (unsigned)((ch_begin * 32 + 4 * pra) * a.OCtot + g * a.OCg + oc0 + 4 * qa) * 4u;
  const unsigned a_step = (unsigned)(32 * a.OCtot) * 4u, a_pix = (unsigned)a.OCtot * 4u;
  const unsigned b_chan = (unsigned)(g * a.Cg + c0 + 4 * qb) * 4u;
  asm volatile("" : "+v"(a_off));

  // The thread's 4 pixels, stepped by 32 pixels per chunk without divisions AND without multiplications: per pixel
  // (oh, ow) and the byte offset of its input element for this tap; a row / image carry adds a constant to the
  // offset (v_mul_lo_u32 runs at quarter rate: the three of ((n*IH + ih)*IW + iw)*C cost more than the whole walk).
  const int hw = a.OH * a.OW;
  const int dn = 32 / hw, rem = 32 % hw;
  const int dh = rem / a.OW, dwid = rem % a.OW;
  const int tap_h = r * a.dil - a.pad, tap_w = s * a.dil - a.pad;
  const unsigned pixb = (unsigned)a.Ctot * 4u;                                  // bytes per input pixel
  const unsigned off_step = (unsigned)((dn * a.IH + dh * a.stride) * a.IW + dwid * a.stride) * pixb;
  const unsigned off_cw = (unsigned)((a.IW - a.OW) * a.stride) * pixb;          // ow wrapped: next output row
  const unsigned off_ch = (unsigned)((a.IH - a.OH * a.stride) * a.IW) * pixb;   // oh wrapped: next image
  int b_m = ch_begin * 32 + 4 * prb;                                            // first pixel of the run
  int b_oh[4], b_ow[4];
  unsigned b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = b_m + i;
    b_ow[i] = m % a.OW;
    const int t = m / a.OW;
    b_oh[i] = t % a.OH;
    const int n = t / a.OH;
    b_off[i] = (unsigned)((n * a.IH + b_oh[i] * a.stride + tap_h) * a.IW + b_ow[i] * a.stride + tap_w) * pixb + b_chan;
  }

  // two register sets: the loads of chunk t+2 are in flight while chunk t is multiplied and chunk t+1 (loaded a whole
  // step ago) is split and moved to the other LDS buffer.  Everything of a step is ONE straight-line block (dead
  // loads are masked by their offset, the last step re-stores a chunk of zeros into the idle buffer), so that the
  // address arithmetic, the loads, the bf16 split and the LDS stores all sit in the issue gaps of the step's 24 MFMAs
  // instead of in front of and behind them.
  float4 ra0[4], rb0[4], ra1[4], rb1[4];
  constexpr bool ALL_A = QA == 32, ALL_B = QB == 32;     // 256 threads = QA/16 x 8 runs: every thread has a slot
  auto load_chunk = [&](bool live, float4 (&ra)[4], float4 (&rb)[4]) {
    if (DBG >= 1) live = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = bload4(rs_dy, (live & a_act) ? a_off + i * a_pix : OOB_OFF);   // rows >= M: beyond dy
    a_off += a_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ih = __mul24(b_oh[i], a.stride) + tap_h, iw = __mul24(b_ow[i], a.stride) + tap_w;
      const bool ok = live & b_act & (b_m + i < a.M) & ((unsigned)ih < (unsigned)a.IH) & ((unsigned)iw < (unsigned)a.IW);
      unsigned off = ok ? b_off[i] : OOB_OFF;
      asm volatile("" : "+v"(off));                         // a select, not a branch around the load
      rb[i] = bload4(rs_x, off);
      b_ow[i] += dwid;
      const bool cw = b_ow[i] >= a.OW;
      b_ow[i] -= cw ? a.OW : 0;
      b_oh[i] += dh + (cw ? 1 : 0);
      const bool chh = b_oh[i] >= a.OH;
      b_oh[i] -= chh ? a.OH : 0;
      b_off[i] += off_step + (cw ? off_cw : 0u) + (chh ? off_ch : 0u);
    }
    b_m += 32;
  };
  // the 4 pixels of one channel -> one 8-byte hi store and one 8-byte lo store in the channel's row
  const int wa_sw = (((pra >> 1) ^ ((qa >> 2) & 3)) << 2) | ((pra & 1) << 1);
  const int wb_sw = (((prb >> 1) ^ ((qb >> 2) & 3)) << 2) | ((prb & 1) << 1);
  auto store_chunk = [&](int buf, const float4 (&ra)[4], const float4 (&rb)[4]) {
    if (DBG >= 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ra[i].x), "v"(ra[i].w), "v"(rb[i].x), "v"(rb[i].w));
      return;
    }
    if (ALL_A || pra < 8) {
      const float4 ch[4] = {make_float4(ra[0].x, ra[1].x, ra[2].x, ra[3].x), make_float4(ra[0].y, ra[1].y, ra[2].y, ra[3].y),
                            make_float4(ra[0].z, ra[1].z, ra[2].z, ra[3].z), make_float4(ra[0].w, ra[1].w, ra[2].w, ra[3].w)};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint2 hi, lo;
        split4(ch[j], hi, lo);
        const int o = (buf * BM + j * QA + qa) * 16 + wa_sw;        // row j*QA + qa: (row >> 2) & 3 == (qa >> 2) & 3
        *(uint2*)(sm + PA_HI + o) = hi;
        *(uint2*)(sm + PA_LO + o) = lo;
      }
    }
    if (ALL_B || prb < 8) {
      const float4 ch[4] = {make_float4(rb[0].x, rb[1].x, rb[2].x, rb[3].x), make_float4(rb[0].y, rb[1].y, rb[2].y, rb[3].y),
                            make_float4(rb[0].z, rb[1].z, rb[2].z, rb[3].z), make_float4(rb[0].w, rb[1].w, rb[2].w, rb[3].w)};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint2 hi, lo;
        split4(ch[j], hi, lo);
        const int o = (buf * BN + j * QB + qb) * 16 + wb_sw;
        *(uint2*)(sm + PB_HI + o) = hi;
        *(uint2*)(sm + PB_LO + o) = lo;
      }
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
  auto step = [&](int it, int cur, float4 (&la)[4], float4 (&lb)[4], const float4 (&sa)[4], const float4 (&sb)[4]) {
    load_chunk(it + 2 < nk, la, lb);
    Frag f0, f1;
    fetch(DBG >= 3 ? 0 : cur, 0, f0);
    if (DBG >= 3) asm volatile("" : "+v"(f0.ah[0]), "+v"(f0.bh[0]));
    mfma3(f0);
    store_chunk(cur ^ 1, sa, sb);
    fetch(DBG >= 3 ? 0 : cur, 1, f1);
    if (DBG >= 3) asm volatile("" : "+v"(f1.ah[0]), "+v"(f1.bh[0]));
    mfma3(f1);
    // issue order: the first half's operand reads, then one MFMA per gap with its share of the VALU work (pixel walk,
    // offsets, bf16 split), the loads in the first gaps, the LDS stores spread evenly, and the second half's operand
    // reads two per gap from the last third of the first half on (their registers free up as the first half drains:
    // both halves resident at once cost 32 VGPRs more and spilled)
    constexpr int NM = TM * TN * 6, NH = NM / 2;           // MFMAs of the step / of a half
    constexpr int NR = 2 * (TM + TN);                      // operand reads of a half (even)
    constexpr int NW = (ALL_A || BM == 64 ? 8 : 0) + (ALL_B || BN == 64 ? 8 : 0);       // LDS stores of a thread with slots
    constexpr int VPM = (8 * 12 + 8 * 10 + NM - 1) / NM;   // split VALU + ~10 per load for its pixel walk, offset, mask
    constexpr int R0 = NH - NR / 2;                        // first gap with second-half reads
    static_assert(R0 >= 0, "two reads per gap");
    __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      if (m >= 1 && m <= 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (m >= R0 && m < NH) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      const int nw = (m + 1) * NW / NM - m * NW / NM;     // the LDS stores spread evenly over the MFMAs
      if (nw >= 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      if (nw >= 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      if (nw >= 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __syncthreads();
  };

  // The bias gradient (sum of dy over the pixels) belongs to the workgroups of tap 0 / input-channel tile 0: they walk
  // their dy rows once more in front of the reduction loop (the rows then come out of L2 for the loop itself); kept
  // out of the loop, whose registers are all spoken for.
  const bool do_bias = a.dshift != nullptr && tap == 0 && tile_n == 0;      // block-uniform
  if (do_bias) {
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);      // channels oc0 + 4*qa .. +3 over this thread's pixel runs
    unsigned o = a_off;
    for (int it = 0; it < nk; ++it, o += a_step) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = bload4(rs_dy, a_act ? o + i * a_pix : OOB_OFF);
      bsum.x += (v[0].x + v[1].x) + (v[2].x + v[3].x);
      bsum.y += (v[0].y + v[1].y) + (v[2].y + v[3].y);
      bsum.z += (v[0].z + v[1].z) + (v[2].z + v[3].z);
      bsum.w += (v[0].w + v[1].w) + (v[2].w + v[3].w);
    }
    // the 8 pixel runs of a channel vector sit in 8 threads: fold through LDS
    static_assert(8 * BM <= LDS_AB, "bias reduction layout");
    if (pra < 8) *(float4*)&smem[pra * BM + 4 * qa] = bsum;
    __syncthreads();
    if (tid < BM && oc0 + tid < a.OCg) {
      float t = 0.f;
#pragma unroll
      for (int p = 0; p < 8; ++p) t += smem[p * BM + tid];
      atomicAdd(a.dshift + g * a.OCg + oc0 + tid, t);
    }
    __syncthreads();
  }

  load_chunk(nk > 0, ra0, rb0);
  store_chunk(0, ra0, rb0);
  load_chunk(nk > 1, ra1, rb1);
  __syncthreads();
  // (a step past the slice's own range multiplies a chunk of zeros: both slices run slice 0's trip count, barriers match)
  for (int it = 0; it < nk_loop; it += 2) {
    step(it, 0, ra0, rb0, ra1, rb1);
    if (it + 1 < nk_loop) step(it + 1, 1, ra1, rb1, ra0, rb0);
  }

  // un-permute through LDS: accumulator row R holds output channel (R % QA) * 4 + R / QA, column C input channel
  // (C % QB) * 4 + C / QB
  float (*Cs)[CP] = reinterpret_cast<float (*)[CP]>(smem_all);
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  if (KS > 1) __syncthreads();                    // the tile image overlaps the other slice's operand buffers
#pragma unroll
  for (int pass = KS - 1; pass >= 0; --pass) {    // KS = 2: slice 1 lays its tile down, slice 0 adds its own onto it
    if (slice == pass) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int C = wn * WTN + j * 32 + ecol;
          const int cl = (C % QB) * 4 + C / QB;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int R = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0;
            float* const q = &Cs[(R % QA) * 4 + R / QA][cl];
            *q = pass == KS - 1 ? acc[i][j][e] : *q + acc[i][j][e];
          }
        }
    }
    __syncthreads();
  }
  if (a.debug_nostore) return;
  constexpr int NT = 256 * KS;
  const int t_all = threadIdx.x;
  if (a.slab || a.split_k == 1) {
    // plain 16-byte accesses: into this split's slab plane, or (a single split: the tile's only writer) added to dw
    float* const base = a.slab ? a.slab + (size_t)split * a.slab_stride : a.dw;
    for (int idx = t_all; idx < BM * (BN / 4); idx += NT) {
      const int row = idx / (BN / 4), col = (idx - row * (BN / 4)) * 4;
      const int ocl = oc0 + row, cl = c0 + col;
      if (ocl >= a.OCg || cl >= a.Cg) continue;
      float* p = base + ((size_t)(g * a.OCg + ocl) * a.R * a.S + tap) * a.Cg + cl;
      float4 v = *(const float4*)&Cs[row][col];
      if (a.row_scale) { const float rs = a.row_scale[g * a.OCg + ocl]; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
      if (cl + 3 < a.Cg) {                                   // Cg % 4 == 0 on this kernel (wvec)
        if (!a.slab) { const float4 o = *(const float4*)p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *(float4*)p = v;
      }
    }
    return;
  }
  for (int idx = t_all; idx < BM * BN; idx += NT) {
    const int row = idx / BN, col = idx - row * BN;
    const int ocl = oc0 + row, cl = c0 + col;
    if (ocl < a.OCg && cl < a.Cg)
      atomicAdd(a.dw + ((size_t)(g * a.OCg + ocl) * a.R * a.S + tap) * a.Cg + cl,
                a.row_scale ? Cs[row][col] * a.row_scale[g * a.OCg + ocl] : Cs[row][col]);
  }
}

// ---- weight gradient of a 3x3 / stride-1 / pad-1 convolution: the three taps of one filter row per workgroup ---------
// wgrad_split_kernel gives every (channel tile, tap) its own workgroup: the nine taps of a tile stream the same dy rows
// and (shifted) x rows, and split every value they load into bf16 hi/lo again.  Timing-only ablations of that kernel
// (tools/wgrad_dbg.sh) put 29 % of a long reduction into that traffic and 23 % into the split + LDS stores, with the
// MFMA-only loop at 0.77 of the bf16x3 peak.  Here a workgroup (128 output channels x 64 input channels) takes the three
// taps s = 0, 1, 2 of ONE filter row r: per 32-pixel chunk it loads the dy rows once and 32 NEW x pixels (the window of
// the three taps is 34 pixels; 32 of them were loaded by the previous chunk), for 36 MFMAs per wave instead of 24 --
// loaded bytes and split work per MFMA halve.
//   * dy: the channel-major image of wgrad_split_kernel (transposed in registers, ds_read_b128 operands), shared by
//     the three taps.
//   * x: a RING of pixel rows in LDS ([pixel][hi 64 ch | lo 64 ch | pad], 320-byte pitch: four consecutive rows x 64 B
//     fall on disjoint banks, no swizzle, so a row's address is linear and a tap shift is an offset immediate), read as
//     MFMA B operands by ds_read_b64_tr_b16 (four rows x 16 channels per 16-lane group, delivered channel-major).
//     The ring holds the PADDED pixel stream: one zero row behind every image row, so that tap s = 0 of a row's first
//     pixel and tap s = 2 of its last one read zeros without any masking in the loop; x rows that the filter row r
//     takes outside the image are masked where they are loaded (they are then zero for all three taps).  Rows 128,
//     129 mirror rows 0, 1 (written twice) so that a read never wraps.
//   * the accumulators (3 taps x 64 x 32 per wave = 96 registers) leave through LDS one tap after the other.
// Requires groups == 1 layout rules checked by taps_eligible(); everything else stays on wgrad_split_kernel.
typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t* lds_v4i16_p;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_taps_kernel(WgradArgs a) {
  constexpr int BM = 128, BN = 64, WN = 2, WTM = 64, WTN = 32, TM = 2;
  constexpr int QA = BM / 4;
  constexpr int NR = 128, ROWB = 320;                     // ring rows, bytes per ring row
  constexpr int A_BYTES = 4 * BM * 16 * 4;                // hi + lo planes, two buffers: 32 KB
  constexpr int RING_BYTES = (NR + 2) * ROWB;
  constexpr int CP = BN + 4;
  static_assert(RING_BYTES >= BM * CP * 4, "epilogue staging fits into the ring");
  // three separate objects: the compiler then knows that the stores into the idle dy buffer cannot alias the operand
  // reads of the live one, and schedules those reads freely around them
  __shared__ __attribute__((aligned(16))) unsigned smA0[A_BYTES / 8];     // [hi | lo][128 rows][16 dwords]
  __shared__ __attribute__((aligned(16))) unsigned smA1[A_BYTES / 8];
  __shared__ __attribute__((aligned(16))) unsigned char ring[RING_BYTES];
  constexpr int PA_LO = BM * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = a.Cg / BN;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd_swizzle) {                                     // whole splits per XCD, see wgrad_split_kernel
    const int gx = gridDim.x, gy = gridDim.y;
    const int lin = bx + gx * (by + gy * bz), nwg = gx * gy * gridDim.z;
    const int q = nwg >> 3, rr = nwg & 7, xcd = lin & 7;
    const int l2 = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (lin >> 3);
    bx = l2 % gx;
    by = (l2 / gx) % gy;
    bz = l2 / (gx * gy);
  }
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int r = by;                                        // filter row of this workgroup
  const int g = bz / a.split_k, split = bz % a.split_k;
  const int oc0 = tile_m * BM, c0 = tile_n * BN;
  const int W = a.OW, H = a.OH;

  const int per = (a.chunks + a.split_k - 1) / a.split_k;
  const int ch_begin = min(a.chunks, split * per), ch_end = min(a.chunks, ch_begin + per);
  const int nk = ch_end - ch_begin;
  const int m_begin = ch_begin * 32;

  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.x, a.x_bytes), rs_dy = make_rsrc(a.dy, a.dy_bytes);

  // ---- dy (A operand): as wgrad_split_kernel -- a thread owns one 4-channel vector x 4 consecutive pixels
  const int l16 = tid & 15, half_run = (tid >> 4) & 1, hi8 = tid >> 5;
  const int qa = (hi8 % (QA / 16)) * 16 + l16, pra = (hi8 / (QA / 16)) * 2 + half_run;
  const bool a_act = oc0 + 4 * qa < a.OCg;
  unsigned a_off = (unsigned)((m_begin + 4 * pra) * a.OCtot + g * a.OCg + oc0 + 4 * qa) * 4u;
  const unsigned a_step = (unsigned)(32 * a.OCtot) * 4u, a_pix = (unsigned)a.OCtot * 4u;
  asm volatile("" : "+v"(a_off));
  const int wa_sw = (((pra >> 1) ^ ((qa >> 2) & 3)) << 2) | ((pra & 1) << 1);

  // ---- x (B operand): the pixel stream.  Stream index t <-> centre pixel mc = m_begin - 1 + t (an output pixel whose
  // tap s = 1 reads this x pixel), x pixel f = mc + (r - 1) W, ring row (t + image rows crossed since t = 0) mod NR.
  // A thread loads ONE pixel per step, t = 2 + (tid >> 3) + 32 j, and eight of its channels (two 16-byte loads: one
  // walk of the pixel's position per 32 bytes).
  const int dwid = 32 % W, dh = 32 / W;
  const int b0 = m_begin - 1 + W;                          // (b0 + t) / W - 1 = image row (over all images) of mc
  const int row0 = b0 / W;
  const int t_lim = a.M - m_begin + 1;                     // mc < M  <=>  t < t_lim
  const unsigned x_step = (unsigned)(32 * a.Ctot) * 4u;
  const int prow = tid >> 3, cq = tid & 7;                 // channels 8 cq .. 8 cq + 7
  int l_iw, l_oh, l_R, l_t;
  unsigned l_off;
  {
    const int t = 2 + prow, z = b0 + t;
    const int rowz = z / W;
    l_t = t;
    l_iw = z - rowz * W;
    l_oh = (rowz - 1 + H) % H;
    l_R = (t + rowz - row0) & (NR - 1);
    l_off = (unsigned)((m_begin - 1 + t + (r - 1) * W) * a.Ctot + g * a.Cg + c0 + 8 * cq) * 4u;
  }
  auto x_valid = [&](int t, int oh) {                      // the filter row stays inside the image, the pixel exists
    return (t < t_lim) & ((unsigned)(oh + r - 1) < (unsigned)H);
  };
  if (!a_act) a_off = 0x80000000u;                         // beyond dy for the whole walk (dy < 2 GiB: validate())

  // two register sets: chunk t+2 in flight, chunk t+1 (loaded a whole step ago) waiting for its LDS store; the ring row
  // and last-column flag travel with the values (the walk has moved on by the time they are stored).  Loads past the
  // workgroup's last chunk are not masked: what they fetch (zeros beyond the tensors, other rows inside) is stored but
  // never multiplied.  (Measured and dropped: one register set re-loaded behind its store, and a sched_group_barrier
  // issue pattern -- both within 5 % of the compiler's own order, the pattern with two sets 4-10 % slower.)
  float4 ra[4], rb[2], ra2[4], rb2[2];
  int mt, mt2;
  auto load_chunk = [&](float4 (&ra)[4], float4 (&rb)[2], int& mt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = bload4(rs_dy, a_off + i * a_pix);       // rows >= M: beyond dy
    a_off += a_step;
    unsigned off = x_valid(l_t, l_oh) ? l_off : OOB_OFF - 16u;
    asm volatile("" : "+v"(off));
    rb[0] = bload4(rs_x, off);
    rb[1] = bload4(rs_x, off + 16u);
    mt = l_R | (l_iw == W - 1 ? 256 : 0);
    l_t += 32;
    l_off += x_step;
    l_iw += dwid;
    const bool cw = l_iw >= W;
    l_iw -= cw ? W : 0;
    const int adv = dh + (cw ? 1 : 0);
    l_oh += adv;
    l_oh -= l_oh >= H ? H : 0;
    l_R = (l_R + 32 + adv) & (NR - 1);
  };
  auto store_x = [&](const float4 v0, const float4 v1, int meta, uint4& hi, uint4& lo) {   // 8 channels -> the ring row
    uint2 h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
    lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
    unsigned char* const p = ring + (meta & 255) * ROWB + cq * 16;
    *(uint4*)p = hi;
    *(uint4*)(p + 128) = lo;
  };
  // the rare stores of a pixel, kept out of the step's straight-line block (one wave-uniform branch at its end): the
  // mirror of ring rows 0 / 1 and the zero row behind the last pixel of an image row
  auto store_x_rare = [&](const uint4 hi, const uint4 lo, int meta) {
    const int R = meta & 255;
    const bool need = (R < 2) | ((meta & 256) != 0);
    if (!__any(need)) return;
    if (R < 2) {
      unsigned char* const p = ring + (R + NR) * ROWB + cq * 16;
      *(uint4*)p = hi;
      *(uint4*)(p + 128) = lo;
    }
    if (meta & 256) {
      const int Rz = (R + 1) & (NR - 1);
      unsigned char* const z = ring + Rz * ROWB + cq * 16;
      const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
      *(uint4*)z = zero;
      *(uint4*)(z + 128) = zero;
      if (Rz < 2) {
        *(uint4*)(z + NR * ROWB) = zero;
        *(uint4*)(z + NR * ROWB + 128) = zero;
      }
    }
  };
  auto store_chunk = [&](int buf, const float4 (&ra)[4]) {
    const float4 ch[4] = {make_float4(ra[0].x, ra[1].x, ra[2].x, ra[3].x), make_float4(ra[0].y, ra[1].y, ra[2].y, ra[3].y),
                          make_float4(ra[0].z, ra[1].z, ra[2].z, ra[3].z), make_float4(ra[0].w, ra[1].w, ra[2].w, ra[3].w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint2 hi, lo;
      split4(ch[j], hi, lo);
      const int o = (j * QA + qa) * 16 + wa_sw;
      *(uint2*)((buf ? smA1 : smA0) + o) = hi;
      *(uint2*)((buf ? smA1 : smA0) + PA_LO + o) = lo;
    }
  };

  // ---- MFMA side: a lane's four reduction pixels k = 16 h + 8 (lane >> 5) + 4 rd + q of the chunk (the rows whose
  // addresses it supplies to the transposed reads), their column in the image row and their ring row minus one
  const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int lanecol = wn * 64 + ((g16 & 1) * 16 + 4 * p4) * 2;
  int k_iw[2][2], k_Rb[2][2];
  unsigned k_addr[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int k = 16 * h + 8 * (lane >> 5) + 4 * rd + q4;
      const int t = k + 1, z = b0 + t, rowz = z / W;
      k_iw[h][rd] = z - rowz * W;
      k_Rb[h][rd] = (t - 1 + rowz - row0) & (NR - 1);
      k_addr[h][rd] = (unsigned)(k_Rb[h][rd] * ROWB + lanecol);
    }
  auto advance_k = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        k_iw[h][rd] += dwid;
        const bool cw = k_iw[h][rd] >= W;
        k_iw[h][rd] -= cw ? W : 0;
        k_Rb[h][rd] = (k_Rb[h][rd] + 32 + dh + (cw ? 1 : 0)) & (NR - 1);
        k_addr[h][rd] = (unsigned)(k_Rb[h][rd] * ROWB + lanecol);
      }
  };

  f32x16 acc[3][TM];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[s3][i][e] = 0.f;

  const int frow = lane & 31;
  auto fetch_a = [&](int cur, int sub, bf16x8 (&ah)[TM], bf16x8 (&al)[TM]) {
    const int r_sw = (((sub * 2 + (lane >> 5)) ^ ((frow >> 2) & 3)) << 2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int o = (wm * WTM + i * 32 + frow) * 16 + r_sw;
      ah[i] = __builtin_bit_cast(bf16x8, *(const uint4*)((cur ? smA1 : smA0) + o));
      al[i] = __builtin_bit_cast(bf16x8, *(const uint4*)((cur ? smA1 : smA0) + PA_LO + o));
    }
  };
  auto fetch_b = [&](int h, int s3, bf16x8& bh, bf16x8& bl) {
    const unsigned char* const p0 = ring + k_addr[h][0] + s3 * ROWB;
    const unsigned char* const p1 = ring + k_addr[h][1] + s3 * ROWB;
    const v4i16_t h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_p)p0);
    const v4i16_t h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_p)p1);
    const v4i16_t l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_p)(p0 + 128));
    const v4i16_t l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_p)(p1 + 128));
    typedef short v8i16_t __attribute__((ext_vector_type(8)));
    const v8i16_t hh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    const v8i16_t ll = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    bh = __builtin_bit_cast(bf16x8, hh);
    bl = __builtin_bit_cast(bf16x8, ll);
  };
  auto half = [&](int cur, int h) {
    bf16x8 ah[TM], al[TM];
    fetch_a(cur, h, ah, al);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      bf16x8 bh, bl;
      fetch_b(h, s3, bh, bl);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        acc[s3][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[s3][i], 0, 0, 0);
        acc[s3][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[s3][i], 0, 0, 0);
        acc[s3][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[s3][i], 0, 0, 0);
      }
    }
  };
  auto step = [&](int cur, float4 (&la)[4], float4 (&lb)[2], int& lm, const float4 (&sa)[4], const float4 (&sb)[2], int smt) {
    uint4 khi, klo;
    load_chunk(la, lb, lm);                        // chunk t+2 goes out first, into the idle set
    half(cur, 0);
    store_x(sb[0], sb[1], smt, khi, klo);          // chunk t+1: its x pixel first (the second half's ring reads must stay
    store_chunk(cur ^ 1, sa);                      // behind this store), then dy into the idle buffer
    half(cur, 1);
    advance_k();
    asm volatile("" : "+v"(k_addr[0][0]), "+v"(k_addr[0][1]), "+v"(k_addr[1][0]), "+v"(k_addr[1][1]), "+v"(l_off), "+v"(l_R),
                 "+v"(l_oh), "+v"(l_iw));          // the walks belong into this block's MFMA gaps, not behind the branch
    store_x_rare(khi, klo, smt);
    __syncthreads();
  };

  // bias gradient: the workgroups of filter row 0 / input-channel tile 0 (every dy element passes exactly one of them)
  const bool do_bias = a.dshift != nullptr && r == 0 && tile_n == 0;
  if (do_bias) {
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned o = a_off;
    for (int it = 0; it < nk; ++it, o += a_step) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = bload4(rs_dy, a_act ? o + i * a_pix : OOB_OFF);
      bsum.x += (v[0].x + v[1].x) + (v[2].x + v[3].x);
      bsum.y += (v[0].y + v[1].y) + (v[2].y + v[3].y);
      bsum.z += (v[0].z + v[1].z) + (v[2].z + v[3].z);
      bsum.w += (v[0].w + v[1].w) + (v[2].w + v[3].w);
    }
    float* const sb = reinterpret_cast<float*>(ring);
    *(float4*)&sb[pra * BM + 4 * qa] = bsum;
    __syncthreads();
    if (tid < BM && oc0 + tid < a.OCg) {
      float t = 0.f;
#pragma unroll
      for (int p = 0; p < 8; ++p) t += sb[p * BM + tid];
      atomicAdd(a.dshift + g * a.OCg + oc0 + tid, t);
    }
    __syncthreads();
  }

  // prologue: stream pixels t = 0, 1 (16 threads), chunk 0 into buffer 0, chunk 1 into the second register set
  if (tid < 16) {
    const int t = tid >> 3, z = b0 + t, rowz = z / W;
    const int iw = z - rowz * W, mc = m_begin - 1 + t;
    const int oh = (rowz - 1 + H) % H;
    const bool ok = mc >= 0 && x_valid(t, oh);
    const unsigned off = ok ? (unsigned)((mc + (r - 1) * W) * a.Ctot + g * a.Cg + c0 + 8 * cq) * 4u : OOB_OFF - 16u;
    const float4 v0 = bload4(rs_x, off), v1 = bload4(rs_x, off + 16u);
    const int meta = ((t + rowz - row0) & (NR - 1)) | (iw == W - 1 ? 256 : 0);
    uint4 hi, lo;
    store_x(v0, v1, meta, hi, lo);
    store_x_rare(hi, lo, meta);
  }
  load_chunk(ra, rb, mt);
  {
    uint4 khi, klo;
    store_x(rb[0], rb[1], mt, khi, klo);
    store_x_rare(khi, klo, mt);
    store_chunk(0, ra);
  }
  load_chunk(ra2, rb2, mt2);
  __syncthreads();
  for (int it = 0; it < nk; it += 2) {
    step(0, ra, rb, mt, ra2, rb2, mt2);             // chunk it in LDS; chunk it+1 waits in set 2, it+2 is loaded into set 1
    if (it + 1 < nk) step(1, ra2, rb2, mt2, ra, rb, mt);
  }

  // the three taps' tiles leave through LDS one after the other; accumulator row R holds output channel
  // (R % QA) * 4 + R / QA (the dy image's row order), columns are input channels in order
  float (*Cs)[CP] = reinterpret_cast<float (*)[CP]>(ring);
  const int ecol = lane & 31, erow0 = 4 * (lane >> 5);
  if (a.debug_nostore) {
    float t = 0.f;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) t += acc[s3][0][0] + acc[s3][1][15];
    if (t == 12345.678f) a.dw[0] = t;
    return;
  }
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int tap = r * 3 + s3;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int R = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + erow0;
        Cs[(R % QA) * 4 + R / QA][wn * WTN + ecol] = acc[s3][i][e];
      }
    __syncthreads();
    if (a.slab || a.split_k == 1) {
      float* const base = a.slab ? a.slab + (size_t)split * a.slab_stride : a.dw;
      for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
        const int row = idx / (BN / 4), col = (idx - row * (BN / 4)) * 4;
        const int ocl = oc0 + row, cl = c0 + col;
        if (ocl >= a.OCg) continue;
        float* p = base + ((size_t)(g * a.OCg + ocl) * 9 + tap) * a.Cg + cl;
        float4 v = *(const float4*)&Cs[row][col];
        if (a.row_scale) { const float rs = a.row_scale[g * a.OCg + ocl]; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
        if (!a.slab) { const float4 o = *(const float4*)p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *(float4*)p = v;
      }
    } else {
      for (int idx = tid; idx < BM * BN; idx += 256) {
        const int row = idx / BN, col = idx - row * BN;
        const int ocl = oc0 + row;
        if (ocl < a.OCg)
          atomicAdd(a.dw + ((size_t)(g * a.OCg + ocl) * 9 + tap) * a.Cg + c0 + col,
                    a.row_scale ? Cs[row][col] * a.row_scale[g * a.OCg + ocl] : Cs[row][col]);
      }
    }
    __syncthreads();
  }
}

// dw += slab[0] + slab[1] + ... in split order (fixed association: bit-reproducible weight gradients)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, size_t stride,
                                                           int64_t n, float* __restrict__ dw) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 acc = ((const float4*)slab)[i];
    for (int s = 1; s < splits; ++s) {
      const float4 v = *(const float4*)(slab + (size_t)s * stride + 4 * i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4 o = ((float4*)dw)[i];
    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
    ((float4*)dw)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += slab[(size_t)s * stride + i];
    dw[i] += acc;
  }
}

// ---- weight gradient, one input channel per group -------------------------------------------------------
// dw[oc][tap] += sum_m dy[m][oc] * x[gather(m, tap)][group(oc)]  for Cg == 1 (the grouped ConvTranspose2d 576 -> 9 of
// Grid_output seen from its weight gradient: 9 groups x 64 outputs x 16 taps x ONE input channel).  The MFMA tile
// kernels would pad the single input channel to 32; here a wave owns a run of pixels, lanes are the group's output
// channels (coalesced dy rows), the tap's input value is a wave-uniform scalar, and each lane keeps R*S <= 16 sums.
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_cg1_kernel(WgradArgs a, int pix_per_block) {
  __shared__ float red[4][TAPS][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // pixel index math and x loads go scalar
  const int g = blockIdx.y;
  const int taps = a.R * a.S;
  const int m_begin = blockIdx.x * pix_per_block, m_end = min(a.M, m_begin + pix_per_block);
  for (int oc0 = 0; oc0 < a.OCg; oc0 += 64) {
    const int ocl = oc0 + lane;
    const bool live = ocl < a.OCg;
    float acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = 0.f;
    // lane t < taps fetches the pixel's tap-t input value (one vector load for all 16 taps instead of 16 dependent
    // scalar loads); the values are then broadcast lane by lane
    const int tr = lane / a.S, ts = lane - tr * a.S;
    // a wave takes runs of 4 consecutive pixels: the 8 loads of a run are issued together and one run AHEAD of the
    // multiply-adds that consume them (one pixel per iteration left every iteration waiting out its own two
    // loads); (n, oh, ow) is decoded once per run
    float dyv[4], xt[4];
    auto load_run = [&](int mb, float (&dv)[4], float (&xv)[4]) {
      int ow = mb % a.OW, t2 = mb / a.OW;
      int oh = t2 % a.OH, n = t2 / a.OH;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = mb + j;
        const bool in = m < m_end;
        dv[j] = (live && in) ? a.dy[(size_t)m * a.OCtot + g * a.OCg + ocl] : 0.f;
        const int ih = oh * a.stride - a.pad + tr * a.dil, iw = ow * a.stride - a.pad + ts * a.dil;
        xv[j] = 0.f;
        if (in && lane < taps && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW)
          xv[j] = a.x[((size_t)(n * a.IH + ih) * a.IW + iw) * a.Ctot + g];
        if (++ow == a.OW) { ow = 0; if (++oh == a.OH) { oh = 0; ++n; } }
      }
    };
    int mb = m_begin + 4 * wave;
    if (mb < m_end) load_run(mb, dyv, xt);
    for (; mb < m_end; mb += 16) {
      float ndy[4] = {0.f, 0.f, 0.f, 0.f}, nx[4] = {0.f, 0.f, 0.f, 0.f};
      if (mb + 16 < m_end) load_run(mb + 16, ndy, nx);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < TAPS; ++t)                    // v_readlane (the index is a constant): __shfl is a ds_bpermute
          acc[t] += dyv[j] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xt[j]), t));
#pragma unroll
      for (int j = 0; j < 4; ++j) { dyv[j] = ndy[j]; xt[j] = nx[j]; }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) red[wave][t][lane] = acc[t];
    __syncthreads();
    if (wave == 0 && live) {
      // the frozen per-channel factor behind the conv (cpm_conv2d_backward_weight_scaled) multiplies the finished sum
      const float rs = a.row_scale ? a.row_scale[g * a.OCg + ocl] : 1.f;
      for (int t = 0; t < taps; ++t) {
        const float v = (red[0][t][lane] + red[1][t][lane] + red[2][t][lane] + red[3][t][lane]) * rs;
        const size_t o = (size_t)(g * a.OCg + ocl) * taps + t;
        if (a.slab) a.slab[(size_t)blockIdx.x * a.slab_stride + o] = v;      // this pixel block's plane (deterministic mode)
        else atomicAdd(a.dw + o, v);
      }
    }
    __syncthreads();
  }
}

// ---- host side ---------------------------------------------------------------------------------------
int num_cus() { return 256; }

// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).  Off by default:
// the hot path pays one predictable branch.
struct ProfRec { hipEvent_t a, b; double flops; int kind; int dims[10]; double epi_bytes; int split; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static double g_flops_next = 0.0;
static int g_dims_next[10] = {0};
// operand bytes of the call's EPILOGUE (residual / gate / accumulator reads): algorithmic traffic that the in + out +
// weight count of a convolution leaves out; and the reduction split of the launch (its partial sums are not)
static double g_epi_bytes_next = 0.0;
static int g_split_next = 1;

struct ProfScope {
  hipStream_t s; int kind; bool on; ProfRec r;
  ProfScope(hipStream_t s_, int kind_) : s(s_), kind(kind_), on(g_prof_on) {
    if (on) {
      r.kind = kind; r.flops = g_flops_next;
      r.epi_bytes = g_epi_bytes_next; r.split = g_split_next;
      for (int i = 0; i < 10; ++i) r.dims[i] = g_dims_next[i];
      on = hipEventCreate(&r.a) == hipSuccess && hipEventCreate(&r.b) == hipSuccess &&
           hipEventRecord(r.a, s) == hipSuccess;
    }
  }
  ~ProfScope() {
    if (on && hipEventRecord(r.b, s) == hipSuccess) g_prof.push_back(r);
  }
};

// cpm_set_deterministic: split reductions of forward / data gradient go through slab planes folded in split order
// instead of float atomics (and the weight gradient)
static int g_deterministic = 0;

// conv arithmetic: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = 3-term split-bf16 MFMA (fp32 accumulate)
static int g_conv_split = 0;


int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// igemm3x3_roi_kernel: a dense 3x3 / stride 1 / pad 1 over 7x7 maps (forward, or the data gradient in gather form) whose
// row blocks give the chip at least CPM_IGEMM_ROI_MIN workgroups without a reduction split
static int64_t roi_halo_tiles(const IgemmArgs& a) {
  if (!g_conv_split) return 0;
  if (!(a.R == 3 && a.S == 3 && a.nr == 3 && a.ns == 3 && a.rstep == 1 && a.sstep == 1 && a.ihmul == 1 && a.iwmul == 1 &&
        (a.hstep == 1 || a.hstep == -1) && (a.wstep == 1 || a.wstep == -1) && a.osh == 1 && a.osw == 1 && a.oah == 0 &&
        a.oaw == 0 && a.groups == 1 && a.CgR == a.Ctot && a.Ctot % 32 == 0 && a.OHp == a.OH && a.OWp == a.OW &&
        a.IH == RH_S && a.IW == RH_S && a.OH == RH_S && a.OW == RH_S && a.ihadd * a.hstep == -1 &&
        a.iwadd * a.wstep == -1 && a.OCg >= 128 && a.in_bytes < 0x80000000u && (!a.res || a.res_mode == 0) && !a.slab))
    return 0;
  return (int64_t)cpm::cdiv(a.M, 128) * cpm::cdiv(a.OCg, 128);
}
// Measured (MI355X, tools/bench_conv.py --math w4, 576 -> 576 on R RoIs, us forward / data gradient; generic plan vs this
// kernel): R = 105: 121 / 129 vs 162 / 151; 128: 128 / 136 vs 160 / 134; 192: 199 / 214 vs 216 / 200; 256: 233 / 240 vs
// 227 / 235.  3.5x fewer A-side loads, splits and LDS stores buy nothing below two full residency rounds: what a k-step
// waits for is not its A operand.  OFF by default (CPM_IGEMM_ROI_MIN=0); the tests switch it on.
static int roi_halo_min() {          // read per call: the tests switch it (0 = off)
  const int v = env_int("CPM_IGEMM_ROI_MIN", 0);
  return v > 0 ? v : 0x7fffffff;
}

// tile + split-K choice: the biggest tile that still gives every CU two workgroups; thin problems take the small
// tile and split the reduction until there are ~3 workgroups per CU (their K loops are latency bound otherwise)
Plan plan_igemm(const IgemmArgs& a) {
  auto tiles = [&](int bm, int bn) { return (int64_t)cpm::cdiv(a.M, bm) * cpm::cdiv(a.OCg, bn) * a.groups; };
  static const int big_waves = env_int("CPM_IGEMM_BIG_WAVES", 4);      // 4: 2x2 waves, 8: 2x4 waves on 128x128
  Plan p;
  p.wm = 2; p.wn = 2; p.split = 1;
  if (roi_halo_tiles(a) >= roi_halo_min()) {          // launch_igemm: igemm3x3_roi_kernel (no reduction split)
    p.bm = 128; p.bn = 128;
    return p;
  }
  if (a.OCg <= 32) {
    // narrow outputs (RPN heads, DCN offset predictors, grouped columns): 128x32 tile, 3 workgroups per CU (LDS).
    // A thin grid with a long reduction (1024->18 3x3 on a stride-16 map: 33 tiles x 288 k-steps) splits the
    // reduction until the chip is ~3 workgroups per CU deep, keeping >= 8 k-steps per workgroup.
    p.bm = 128; p.bn = 32; p.wm = 4; p.wn = 1;
    const int64_t t = tiles(128, 32), want = 3ll * num_cus();
    if (t < want && a.ksteps >= 32) {
      int64_t sp = (want + t - 1) / t;
      if (sp > a.ksteps / 8) sp = a.ksteps / 8;
      if (sp > 64) sp = 64;
      if (sp > 1) p.split = (int)sp;
    }
    return p;
  }
  if (const char* f = getenv("CPM_IGEMM_FORCE")) {          // experiments: "bm,bn,split"
    int bm, bn, sp;
    if (sscanf(f, "%d,%d,%d", &bm, &bn, &sp) == 3) {
      p.bm = bm; p.bn = bn; p.wm = bn == 32 ? 4 : 2; p.wn = bn == 32 ? 1 : 2; p.split = sp < 1 ? 1 : sp;
      if (p.split > a.ksteps / 2) p.split = a.ksteps / 2 > 0 ? a.ksteps / 2 : 1;
      return p;
    }
  }
  // Cost model fitted to tools/sweep_igemm.sh on MI355X.  A candidate = (tile, reduction split).  Its grid runs in
  // residency rounds of `per_cu` workgroups per CU (LDS-limited: 2 for the 128-tiles, 4 for 64x64); a workgroup
  // costs its MFMA work / tile efficiency plus a fixed prologue+epilogue, and a partly filled round is cheaper
  // than a full one (a workgroup alone on a CU is latency bound, not 4x faster).  Thin problems therefore get the
  // small tile with the reduction split until the chip holds ~4 workgroups per CU, big ones the 128x128 tile
  // with the remainder of the last round re-tiled (launch_igemm).
  struct Cand { int bm, bn, per_cu; double eff; };
  // bf16x3: the MFMA part of a k-step is ~2.5x shorter, so the fixed per-tile work weighs more and the small tiles
  // (3-4 workgroups per CU) lose less against the big one (tools/sweep_igemm.sh, MATH=bf16x3)
  const Cand cands_f32[3] = {{128, 128, 2, 1.00}, {128, 64, 2, 0.86}, {64, 64, 4, 0.78}};
  const Cand cands_split_short[3] = {{128, 128, 2, 1.00}, {128, 64, 3, 0.90}, {64, 64, 4, 0.95}};
  // 2-4 k-steps (1x1 layers over 64 or 128 channels at full resolution): the tile is all prologue and epilogue, and
  // four 64x64 workgroups per CU interleave those phases better than two big ones (measured in the training step, with
  // the real residual / gate epilogues: 64->256 on 200x336 124 -> 97 us, 128->512 on 100x168 72 -> 67 us)
  if (g_conv_split && a.ksteps <= 4 && a.OCg >= 64 && tiles(64, 64) >= 2ll * num_cus()) {
    p.bm = 64; p.bn = 64; p.split = 1;
    return p;
  }
  // 24-40 k-steps on between one and two residency rounds of 128x128 tiles (fc6's data gradient: 1024 RoIs x 12 544
  // columns over 1 024 channels = 784 tiles): the round model below prefers 3 136 64x64 tiles (204 us); as persistent
  // runs of two 128x128 tiles on 392 workgroups the launch takes 143 us (launch_one: igemm_pt_kernel)
  if (g_conv_split && a.ksteps >= 24 && a.ksteps <= 40 && a.OCg >= 128 && a.groups == 1) {
    const int64_t t = tiles(128, 128);
    if (t > 2ll * num_cus() && t <= 4ll * num_cus()) {
      p.bm = 128; p.bn = 128; p.split = 1;
      return p;
    }
  }
  const Cand* cands = (g_conv_split && a.ksteps <= 40) ? cands_split_short : cands_f32;   // short reductions (1x1)
  const double fixed = (g_conv_split && a.ksteps <= 40) ? 10.0 : 5.0;
  double best = 1e300;
  for (int ci = 0; ci < 3; ++ci) {
    const Cand& c = cands[ci];
    if (c.bn == 128 && a.OCg < 128) continue;
    const int64_t t = tiles(c.bm, c.bn);
    const int64_t slots = (int64_t)c.per_cu * num_cus();
    const int max_split = a.ksteps >= 32 ? a.ksteps / 16 : 1;
    for (int sp = 1; sp <= max_split && sp <= 64; ++sp) {
      const int64_t blocks = t * sp;
      const double per_block = ((double)c.bm * c.bn * ((double)a.ksteps / sp + fixed)) / c.eff;
      const int64_t full = blocks / slots, tail = blocks % slots;
      double rounds = (double)full;
      if (tail) {
        const double occ = (double)((tail + num_cus() - 1) / num_cus()) / c.per_cu;     // fraction of a full round
        rounds += 0.42 + 0.58 * occ;
      }
      // the 128x128 grid drops its partial round onto 64x64 tiles when that round is less than half full
      if (c.bm == 128 && c.bn == 128 && sp == 1 && full >= 1 && tail && tail * 2 < slots)
        rounds = (double)full + 0.30 + 0.7 * (double)tail / slots;
      double cost = rounds * per_block * c.per_cu;     // a round of per_cu workgroups shares the CU's MFMA pipes
      // A short reduction on big tiles that fill the chip about once: every workgroup is in its prologue, then in its
      // k-steps, then in its epilogue at the same time -- nothing overlaps (256->1024 1x1 on 8 400 pixels: 528 tiles,
      // 42.4 us; as 2 112 64x64 tiles 34.6 us; tools/sweep_short_k.sh)
      if (g_conv_split && a.ksteps <= 16 && c.bm == 128 && c.bn == 128 && sp == 1 && full <= 1) cost *= 1.15;
      if (sp > 1) cost += 3.0 * (double)a.M * a.OCg * a.groups / num_cus();            // memset + atomics + epilogue pass
      if (cost < best) { best = cost; p.bm = c.bm; p.bn = c.bn; p.split = sp; }
      if (blocks >= 4 * slots) break;
    }
  }
  if (p.bm == 128 && p.bn == 128 && big_waves == 8) p.wn = 4;
  static const int dbg = env_int("CPM_IGEMM_DEBUG", 0);
  if (dbg) fprintf(stderr, "[igemm plan] M=%d OCg=%d ksteps=%d groups=%d -> %dx%d split %d (tiles %lld)\n", a.M, a.OCg,
                   a.ksteps, a.groups, p.bm, p.bn, p.split, (long long)tiles(p.bm, p.bn));
  return p;
}

// igemm_pt_kernel: a run of tiles per workgroup, as many workgroups as the chip holds at once (or fewer)
template <int BM, int BN, int WM, int WN>
int launch_pt(const IgemmArgs& a, int per_cu, hipStream_t s) {
  const int64_t tiles = (int64_t)cpm::cdiv(a.M - a.m_base, BM) * cpm::cdiv(a.OCg, BN) * a.groups;
  const int cap = env_int("CPM_IGEMM_PT_WGS", 0);              // tests: a small grid, so that small problems get runs too
  const int64_t slots = cap > 0 ? cap : (int64_t)per_cu * num_cus();
  const int run = (int)((tiles + slots - 1) / slots);
  const int64_t wgs = (tiles + run - 1) / run;
  hipLaunchKernelGGL((igemm_pt_kernel<BM, BN, WM, WN>), dim3((unsigned)wgs), dim3(64 * WM * WN), 0, s, a, (int)tiles, run);
  return cpm::check_launch("conv igemm (persistent tiles)");
}

int launch_one(const IgemmArgs& a, int bm, int bn, int wn, bool vec, hipStream_t s) {
  const int rows = a.M - a.m_base;
  // persistent tiles (igemm_pt_kernel): whenever the grid would not fit the chip at once.  CPM_IGEMM_PT: 0 off, 1 wherever
  // the kernel applies (tests), 2 by grid size.  Read per call: the tests switch it.
  const int pt = env_int("CPM_IGEMM_PT", 2);
  if (pt && vec && g_conv_split && a.b_presplit && a.split_k == 1 && !a.atomic_out && !a.slab && a.ksteps >= 1 &&
      bn >= 64 && (wn == 2)) {
    const int per_cu = bm * bn >= 128 * 128 ? 2 : (bm * bn >= 128 * 64 ? 3 : 4);
    const int64_t tiles = (int64_t)cpm::cdiv(rows, bm) * cpm::cdiv(a.OCg, bn) * a.groups;
    // by grid size (2): grids of several residency rounds on dense output rows.  Measured per layer (MI355X, tools/
    // bench_conv.py): the 1x1 layers of layer1 / layer2 and the P2 lateral (8-16 tiles per workgroup) 10-20 % faster
    // forward and in the plain data gradient (64->256 on 200x336: 64.6 -> 52.7 us, 256->64: 41.1 -> 35.6, lateral 98.8 ->
    // 90.5); NO change on the 2-3-tile runs of layer3 / layer4 (30 us either way, whatever the tile: those are not bound
    // by their prologues); the accumulating data gradient (residual == output) and strided output rows are faster
    // through igemm_kernel's LDS-staged row-wise epilogue (256->128 stride 2: 30 vs 41 us) and stay there.
    const bool dense_out = a.osh == 1 && a.osw == 1 && a.OHp == a.OH && a.OWp == a.OW;
    // ... and at least two full rounds of them: a run length is a whole number, so 1056 tiles on 1024 slots would go to
    // 528 workgroups of two tiles -- half the chip's slots -- where the plain grid runs one full round and a tail
    // (1024 -> 1024 1x1 on 50x84, X-101 at bs=1: 59 us persistent, 51 plain)
    static const int min_rounds = env_int("CPM_IGEMM_PT_MIN_ROUNDS", 2);
    const int64_t slots = (int64_t)per_cu * num_cus();
    if (pt == 1 || (tiles > slots && tiles >= min_rounds * slots && dense_out && !(a.res && a.res == a.out))) {
      if (bm == 128 && bn == 128) return launch_pt<128, 128, 2, 2>(a, per_cu, s);
      if (bm == 128 && bn == 64) return launch_pt<128, 64, 2, 2>(a, per_cu, s);
      if (bm == 64 && bn == 64) return launch_pt<64, 64, 2, 2>(a, per_cu, s);
    }
  }
  // one-LDS-stage variant (three workgroups per CU): pays once the grid holds three rounds' worth of workgroups per CU
  // (256->256 1x1 on 200x336: 125 -> 109 us; 576x576x3x3 on 192 RoIs 209 -> 200 us), loses 5-10 % on grids that do not
  // even fill two per CU (its workgroups are slower one by one).  CPM_IGEMM_S1: 0 off, 1 always (tests), 2 by grid size.
  const int s1 = env_int("CPM_IGEMM_S1", 2);                  // read per call: the tests switch it
  const int64_t s1_wgs = (int64_t)cpm::cdiv(rows, 128) * cpm::cdiv(a.OCg, 128) * a.groups * a.split_k;
  if ((s1 == 1 || (s1 == 2 && s1_wgs >= 3ll * num_cus())) && vec && g_conv_split && bm == 128 && bn == 128 && wn == 2 &&
      (a.atomic_out || !(a.res || a.staged_epi))) {
    dim3 grid((unsigned)(cpm::cdiv(rows, 128) * cpm::cdiv(a.OCg, 128)), a.groups, a.split_k);
    if (a.b_presplit) hipLaunchKernelGGL((igemm_s1_kernel<128, 128, 2, 2, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((igemm_s1_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, a);
    return cpm::check_launch("conv igemm (one LDS stage)");
  }
#define LAUNCH(BM, BN, WM, WN)                                                                       \
  do {                                                                                               \
    dim3 grid((unsigned)(cpm::cdiv(rows, BM) * cpm::cdiv(a.OCg, BN)), a.groups, a.split_k);          \
    if (vec && g_conv_split && a.tail)                                                                     \
      hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, 1, true>), grid, dim3(64 * WM * WN), 0, s, a); \
    else if (vec && g_conv_split && a.b_presplit)                                                          \
      hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, 2>), grid, dim3(64 * WM * WN), 0, s, a);      \
    else if (vec && g_conv_split)                                                                          \
      hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, 1>), grid, dim3(64 * WM * WN), 0, s, a);      \
    else if (vec)                                                                                          \
      hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, 0>), grid, dim3(64 * WM * WN), 0, s, a);      \
    else if (g_conv_split)                                                                                 \
      hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, false, 1>), grid, dim3(64 * WM * WN), 0, s, a);     \
    else                                                                                                   \
      hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, false, 0>), grid, dim3(64 * WM * WN), 0, s, a);     \
  } while (0)
  if (bm == 128 && bn == 128 && wn == 4) LAUNCH(128, 128, 2, 4);
  else if (bm == 128 && bn == 128) LAUNCH(128, 128, 2, 2);
  else if (bm == 128 && bn == 64) LAUNCH(128, 64, 2, 2);
  else if (bm == 128 && bn == 32) LAUNCH(128, 32, 4, 1);
  else LAUNCH(64, 64, 2, 2);
#undef LAUNCH
  return cpm::check_launch("conv igemm");
}

int launch_igemm(IgemmArgs a, const Plan& p, hipStream_t s, int prof_kind) {
  bool vec = (a.CgR % 4 == 0) && (a.Ctot % 4 == 0) && (((uintptr_t)a.in & 15) == 0) &&
             (((uintptr_t)a.wm & 15) == 0);
  // ragged reduction channels (18): the vector path with masked tails (igemm_kernel<.., TAIL>), bf16x3 only
  static const int tail_on = env_int("CPM_IGEMM_TAIL_VEC", 1);
  a.tail = 0;
  if (!vec && tail_on && g_conv_split && !a.b_presplit && a.groups == 1 && a.CgR == a.Ctot && a.CgR >= 4 &&
      (((uintptr_t)a.in & 15) == 0) && (((uintptr_t)a.wm & 15) == 0)) {
    vec = true;
    a.tail = 1;
  }
  if (a.b_presplit && !(vec && g_conv_split)) {
    cpm::set_error("conv igemm: a pre-split weight image outside the bf16x3 vector path");
    return CPM_EINVAL;
  }
  static const int swz = env_int("CPM_IGEMM_XCD", 1);
  static const int idbg = env_int("CPM_IGEMM_DBG", 0);
  a.dbg = idbg;
  static const int tail_split = env_int("CPM_IGEMM_TAIL", 1);
  a.xcd_swizzle = swz;
  a.m_base = 0;
  static const int staged = env_int("CPM_IGEMM_STAGED_EPI", 0);   // measured: no gain without a residual (direct stores)
  a.staged_epi = staged;
  ProfScope prof_scope(s, prof_kind);
  static const int use_halo = env_int("CPM_IGEMM_HALO", 1);
  if (use_halo && g_conv_split && vec && a.R == 3 && a.S == 3 && a.nr == 3 && a.ns == 3 && a.rstep == 1 && a.sstep == 1 &&
      a.ihmul == 1 && a.iwmul == 1 && (a.hstep == 1 || a.hstep == -1) && (a.wstep == 1 || a.wstep == -1) &&
      a.osh == 1 && a.osw == 1 && a.oah == 0 && a.oaw == 0 && a.groups == 1 && a.CgR == a.Ctot && a.Ctot % 32 == 0 &&
      a.OHp == a.OH && a.OWp == a.OW && a.IH == a.OH && a.IW == a.OW && a.split_k == 1 && !a.atomic_out &&
      a.ihadd * a.hstep == -1 && a.iwadd * a.wstep == -1 && a.OCg >= 128 && a.OH >= 7 && a.OW >= 14 &&
      a.in_bytes < 0x80000000u) {
    const int64_t patches = (int64_t)a.N * cpm::cdiv(a.OH, HT_H) * cpm::cdiv(a.OW, HT_W);
    const int64_t blocks = patches * cpm::cdiv(a.OCg, 128);
    static const int halo_min = env_int("CPM_IGEMM_HALO_MIN", 320);    // below ~1.25 workgroups per CU the generic kernel's finer tiles win (128 ch on 100x168: 64 -> 60 us)
    if (blocks >= halo_min) {
      if (a.b_presplit) hipLaunchKernelGGL((igemm3x3_kernel<128, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((igemm3x3_kernel<128>), dim3((unsigned)blocks), dim3(256), 0, s, a);
      return cpm::check_launch("conv igemm 3x3 (halo)");
    }
  }
  if (vec && a.split_k == 1 && !a.atomic_out && roi_halo_tiles(a) >= roi_halo_min()) {
    const int64_t blocks = roi_halo_tiles(a);
    if (a.b_presplit) hipLaunchKernelGGL((igemm3x3_roi_kernel<128, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((igemm3x3_roi_kernel<128, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    return cpm::check_launch("conv igemm 3x3 (RoI halo)");
  }
  // Wave quantisation: the big tiles run 2 workgroups per CU (LDS), i.e. 512 at a time.  When the tile count is a
  // little over a multiple of 512 the last round would keep a few CUs busy for a whole tile time while the rest
  // idle (2100 tiles = 4.1 rounds cost 5).  So the rows of the full rounds go to the big tile and the remaining
  // rows are re-tiled 64x64 (4 workgroups per CU, a quarter of the tile time) in a second launch.
  if (tail_split && p.bm == 128 && p.bn >= 64 && a.split_k == 1 && a.groups == 1) {
    const int tiles_n = cpm::cdiv(a.OCg, p.bn);
    const int64_t tiles_m = cpm::cdiv(a.M, p.bm), total = tiles_m * tiles_n, slots = 2 * num_cus();
    const int64_t rem = total % slots;
    if (total > slots && rem != 0 && rem * 2 < slots) {
      const int64_t main_m = (total - rem) / tiles_n;      // whole row blocks in the full rounds
      if (main_m > 0 && main_m < tiles_m) {
        IgemmArgs b = a;
        b.M = (int)(main_m * p.bm);
        int rc = launch_one(b, p.bm, p.bn, p.wn, vec, s);
        if (rc != CPM_OK) return rc;
        a.m_base = (int)(main_m * p.bm);
        return launch_one(a, 64, 64, 2, vec, s);
      }
    }
  }
  return launch_one(a, p.bm, p.bn, p.wn, vec, s);
}

int validate(const cpm_conv_desc* d) {
  if (!d) return CPM_EINVAL;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->R <= 0 || d->S <= 0) return CPM_EINVAL;
  if (d->stride <= 0 || d->pad < 0 || d->dilation <= 0 || d->groups <= 0) return CPM_EINVAL;
  if (d->C % d->groups || d->K % d->groups) return CPM_EINVAL;
  const int P = (d->H + 2 * d->pad - d->dilation * (d->R - 1) - 1) / d->stride + 1;
  const int Q = (d->W + 2 * d->pad - d->dilation * (d->S - 1) - 1) / d->stride + 1;
  if (P != d->P || Q != d->Q) return CPM_EINVAL;
  // byte offsets travel as 32-bit buffer offsets: every tensor must stay below 2^30 floats (4 GiB - 16)
  if ((int64_t)d->N * d->H * d->W * d->C >= (1ll << 30) - (1 << 17) || (int64_t)d->N * d->P * d->Q * d->K >= (1ll << 30) - (1 << 17) ||
      (int64_t)d->K * d->R * d->S * (d->C / d->groups) >= (1ll << 29) - 4)     // weights < 2 GiB: B_INVALID
    return CPM_EINVAL;
  return CPM_OK;
}

size_t dgrad_weight_bytes(const cpm_conv_desc* d) {
  return ((size_t)d->K * d->R * d->S * (d->C / d->groups) * sizeof(float) + 255) / 256 * 256;
}

}  // namespace

static size_t wgrad_slab_bytes(const cpm_conv_desc* d);

// bytes of split-K slab planes the planner would use for the forward (or the data gradient) of `d`
static size_t splitk_slab_bytes(const cpm_conv_desc* d, bool dgrad) {
  IgemmArgs a = {};
  a.groups = d->groups;
  if (!dgrad) {
    a.M = d->N * d->P * d->Q; a.OCg = d->K / d->groups; a.OCtot = d->K;
    a.ksteps = d->R * d->S * cpm::cdiv(d->C / d->groups, BK);
  } else {
    const int st = d->stride;
    a.M = d->N * d->H * d->W; a.OCg = d->C / d->groups; a.OCtot = d->C;
    a.ksteps = cpm::cdiv(d->R, st) * cpm::cdiv(d->S, st) * cpm::cdiv(d->K / d->groups, BK);
  }
  const Plan p = plan_igemm(a);
  if (p.split <= 1) return 0;
  return (size_t)p.split * (((size_t)a.M * a.OCtot + 63) / 64 * 64) * sizeof(float);
}

CPM_EXPORT size_t cpm_conv2d_workspace_bytes(const cpm_conv_desc* d) {
  if (validate(d) != CPM_OK) return 0;
  // the larger of: data-gradient weight image + its split-K slab, weight-gradient slab, forward split-K slab
  const size_t dg = dgrad_weight_bytes(d) + 256 + splitk_slab_bytes(d, true), wg = wgrad_slab_bytes(d),
               fw = splitk_slab_bytes(d, false);
  size_t m = dg > wg ? dg : wg;
  m = m > fw ? m : fw;
  return m + 256;
}

// a split-K slab of `split` planes fits the workspace region [off, bytes)?
static float* slab_in(void* workspace, size_t bytes, size_t off, int split, size_t plane_floats, int oc_tot,
                      bool needs_pass = false) {
  // Measured slower than the float-atomic form on the forward / data-gradient splits (grid head 3x3 at 64 RoIs: 87 vs
  // 83 us; iou_fc1: 46 vs 36 us -- the bias-seeded accumulator needs no epilogue pass, the planes cost split x the
  // output in stores and again in loads), so it is the DETERMINISTIC mode's path, not the default one.  (The weight
  // gradient's planes ARE the default: there the atomics cost 23 % of the kernel.)
  // CPM_SPLITK_SLAB=2: planes only where the atomic form needs a zero fill AND an epilogue pass anyway (a frozen
  // affine / ReLU / gate behind the sum): fill + atomics + pass become plain stores + one reduce-and-epilogue pass.
  // In the training step that is the default (2): 24.16 -> 23.77 ms/step over five alternating runs.  1 = planes for
  // every split (the deterministic mode's path), 0 = atomics everywhere.
  static const int on = env_int("CPM_SPLITK_SLAB", env_int("CPM_DETERMINISTIC", 0) ? 1 : 2);
  if (!(on == 1 || g_deterministic || (on == 2 && needs_pass)) || split <= 1 || !workspace) return nullptr;
  off = (off + 255) / 256 * 256;
  if (bytes < off + (size_t)split * plane_floats * sizeof(float)) return nullptr;
  return (float*)((char*)workspace + off);
}

// cpm_conv_next_output_prepared: the caller (a layer chain) has had the previous kernel fill this call's output
static thread_local int g_out_prepared = 0;
CPM_EXPORT void cpm_conv_next_output_prepared(int on) { g_out_prepared = on ? 1 : 0; }

static int conv_forward_impl(const cpm_conv_desc* d, const float* x, const float* w,
                             const float* scale, const float* shift, const float* residual, int res_mode, int relu,
                             float* y, hipStream_t s, void* workspace = nullptr, size_t workspace_bytes = 0,
                             int w_presplit = 0) {
  const bool prepared = g_out_prepared != 0;
  g_out_prepared = 0;
  IgemmArgs a = {};
  a.b_presplit = w_presplit;
  a.in = x; a.wm = w; a.out = y; a.scale = scale; a.shift = shift; a.res = residual;
  a.N = d->N; a.IH = d->H; a.IW = d->W; a.Ctot = d->C;
  a.OH = d->P; a.OW = d->Q; a.OCtot = d->K; a.OHp = d->P; a.OWp = d->Q;
  a.R = d->R; a.S = d->S;
  a.ihmul = a.iwmul = d->stride; a.ihadd = a.iwadd = -d->pad; a.hstep = a.wstep = d->dilation;
  a.r0 = a.s0 = 0; a.rstep = a.sstep = 1; a.nr = d->R; a.ns = d->S;
  a.osh = a.osw = 1; a.oah = a.oaw = 0;
  a.groups = d->groups; a.CgR = d->C / d->groups; a.OCg = d->K / d->groups;
  a.M = d->N * d->P * d->Q;
  a.ksteps_per_tap = cpm::cdiv(a.CgR, BK);
  a.ksteps = d->R * d->S * a.ksteps_per_tap;
  a.res_mode = res_mode; a.relu = relu;
  a.in_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->C * 4);
  a.wm_bytes = (unsigned)((size_t)d->K * d->R * d->S * (d->C / d->groups) * 4);
  Plan p = plan_igemm(a);
  a.split_k = p.split;
  a.atomic_out = a.split_k > 1;
  bool seeded = false;
  const size_t plane = ((size_t)a.M * a.OCtot + 63) / 64 * 64;
  const bool bias_seed = shift && !scale && !residual && !relu && (a.OCtot & 3) == 0 &&
                         (((uintptr_t)shift | (uintptr_t)y) & 15) == 0;
  a.slab = slab_in(workspace, workspace_bytes, 0, a.split_k, plane, a.OCtot,
                   !bias_seed && (scale || shift || residual || relu));
  a.slab_stride = plane;
  if (a.slab) {
    // partial sums in slab planes: no zero fill, no atomics; the reduce pass below also runs the epilogue
  } else if (a.atomic_out) {
    if (shift && !scale && !residual && !relu && (a.OCtot & 3) == 0 && (((uintptr_t)shift | (uintptr_t)y) & 15) == 0) {
      if (!prepared) {                      // (prepared: the previous kernel of the chain left the bias in every pixel)
        const int64_t total4 = (int64_t)a.M * a.OCtot / 4;
        const int64_t b = (total4 + 255) / 256;
        hipLaunchKernelGGL(seed_rows_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, s, y, shift, total4,
                           a.OCtot / 4);
      }
      seeded = true;
    } else if (prepared && !shift && !scale && !residual && !relu) {
      // cleared by the previous kernel of the chain
    } else if (hipMemsetAsync(y, 0, (size_t)a.M * a.OCtot * sizeof(float), s) != hipSuccess) {
      return CPM_ELAUNCH;
    }
  }
  g_flops_next = 2.0 * d->N * d->P * d->Q * (double)d->K * d->R * d->S * (d->C / d->groups);
  g_epi_bytes_next = !residual ? 0.0
                     : 4.0 * d->K * (res_mode == 0 ? (double)d->N * d->P * d->Q
                                                   : (double)d->N * ((d->P + 1) / 2) * ((d->Q + 1) / 2));
  g_split_next = a.split_k;
  { const int dd[10] = {d->N, d->H, d->W, d->C, d->K, d->R, d->stride, d->groups, d->P, d->Q}; for (int i = 0; i < 10; ++i) g_dims_next[i] = dd[i]; }
  int rc = launch_igemm(a, p, s, 0);
  if (rc != CPM_OK) return rc;
  if (a.slab) {
    const int64_t b = ((int64_t)a.M * a.OCtot / 4 + 255) / 256;
    const dim3 rg((unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b)));
    // (the scalar form handles ONE element per thread and pass: a grid sized for float4s walked four dependent rounds
    // of loads -- 27 us for the 4200 x 18 output of X-101's offset predictors, 31 times per step)
    const int64_t bs = ((int64_t)a.M * a.OCtot + 255) / 256;
    const dim3 rgs((unsigned)(bs > 4096 ? 4096 : (bs < 1 ? 1 : bs)));
    if (a.OCtot & 3)
      hipLaunchKernelGGL(splitk_reduce_scalar_kernel, rgs, dim3(256), 0, s, a.slab, a.split_k, a.slab_stride, y, 0, scale,
                         shift, residual, (int64_t)a.M, a.OCtot, a.OH, a.OW, res_mode, relu, (const float*)nullptr);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, rg, dim3(256), 0, s, a.slab, a.split_k, a.slab_stride, y, 0, scale, shift,
                         residual, (int64_t)a.M, a.OCtot, a.OH, a.OW, res_mode, relu, (const float*)nullptr);
    rc = cpm::check_launch("conv split-K reduce");
  } else if (a.atomic_out && !seeded && (scale || shift || residual || relu)) {
    const int64_t total = (int64_t)a.M * a.OCtot;
    int64_t b = (total + 255) / 256;
    hipLaunchKernelGGL(epilogue_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, s, y, scale, shift,
                       residual, (int64_t)a.M, a.OCtot, a.OH, a.OW, res_mode, relu);
    rc = cpm::check_launch("conv epilogue");
  }
  return rc;
}

CPM_EXPORT int cpm_conv2d_forward(const cpm_conv_desc* d, const float* x, const float* w, const float* scale,
                                  const float* shift, const float* residual, int res_mode, int relu, float* y,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(x && w && y, "null pointer");
  CPM_REQUIRE(res_mode == 0 || res_mode == 1, "bad res_mode");
  return conv_forward_impl(d, x, w, scale, shift, residual, res_mode, relu, y, (hipStream_t)stream, workspace,
                           workspace_bytes);
}

// y = epilogue(conv(x, W)) with W given as its pre-split image (cpm_split_w4 of the KRSC weight): bf16x3 arithmetic only
CPM_EXPORT int cpm_conv2d_forward_w4(const cpm_conv_desc* d, const float* x, const void* w4, const float* scale,
                                     const float* shift, const float* residual, int res_mode, int relu, float* y,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(x && w4 && y, "null pointer");
  CPM_REQUIRE(res_mode == 0 || res_mode == 1, "bad res_mode");
  CPM_REQUIRE(g_conv_split, "a pre-split weight image serves the bf16x3 arithmetic only (cpm_set_conv_math)");
  CPM_REQUIRE((d->C / d->groups) % 4 == 0 && d->C % 4 == 0 && (((uintptr_t)x | (uintptr_t)w4) & 15) == 0,
              "a pre-split weight image needs the vector path: channels per group % 4 == 0, 16-byte aligned operands");
  return conv_forward_impl(d, x, (const float*)w4, scale, shift, residual, res_mode, relu, y, (hipStream_t)stream,
                           workspace, workspace_bytes, 1);
}

// hi / lo image of a float array for the weight side of the bf16x3 kernels: elements 4i .. 4i+3 -> 8 bytes of bf16 hi,
// 8 bytes of bf16 lo (lo = bf16(x - hi)) at byte offset 16 i -- the offsets of the float array itself.  n % 4 == 0.
__global__ __launch_bounds__(256) void split_w4_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    uint2 hi, lo;
    split4(src[i], hi, lo);
    dst[i] = make_uint4(hi.x, hi.y, lo.x, lo.y);
  }
}

CPM_EXPORT int cpm_split_w4(const float* src, void* dst, int64_t n, void* stream) {
  CPM_REQUIRE(src && dst && n >= 0 && n % 4 == 0, "cpm_split_w4: n must be a multiple of 4");
  CPM_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "cpm_split_w4: 16-byte aligned buffers");
  if (n == 0) return CPM_OK;
  const int64_t n4 = n / 4;
  int64_t b = (n4 + 255) / 256;
  b = b > 8192 ? 8192 : b;
  hipLaunchKernelGGL(split_w4_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (uint4*)dst, n4);
  return cpm::check_launch("cpm_split_w4");
}

static int run_dgrad(const cpm_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate,
                     const float* shift, int relu, void* workspace, size_t workspace_bytes, hipStream_t s,
                     const char* who, const float* out_scale = nullptr, const float* out_mask = nullptr,
                     bool prepared = false, const float* k_scale = nullptr, int prepared_w4 = 0) {
  const bool out_ready = g_out_prepared != 0;       // cpm_conv_next_output_prepared: consumed by this call
  g_out_prepared = 0;
  const size_t need = dgrad_weight_bytes(d);
  if (!prepared && (!workspace || workspace_bytes < need)) {
    cpm::set_error("%s: workspace %zu < %zu", who, workspace_bytes, need);
    return CPM_EWORKSPACE;
  }
  // prepared: `w` already is the [g][c][tap][k] image (cpm_weights_to_dgrad_batched)
  const float* wt = prepared ? w : (const float*)workspace;
  const int Cg = d->C / d->groups, Kg = d->K / d->groups;
  // the weight side pre-split (bf16x3, vector path): an image made here is written that way, a prepared one says so
  const bool vec_ok = Kg % 4 == 0 && d->K % 4 == 0 && (((uintptr_t)dy | (uintptr_t)wt) & 15) == 0;
  static const int env_w4 = env_int("CPM_W4", 1);
  const int w4 = prepared ? prepared_w4 : (env_w4 && g_conv_split && vec_ok ? 1 : 0);
  if (prepared_w4 && !(g_conv_split && vec_ok)) {
    cpm::set_error("%s: a pre-split weight image needs the bf16x3 arithmetic and the vector path (K / groups %% 4 == 0, "
                   "16-byte aligned operands)", who);
    return CPM_EINVAL;
  }
  if (!prepared) {
    const int64_t b = (int64_t)cpm::cdiv(Cg, 32) * cpm::cdiv(Kg, 32) * d->R * d->S * d->groups;
    hipLaunchKernelGGL(weight_to_dgrad, dim3((unsigned)(b > 16384 ? 16384 : b)), dim3(256), 0, s, w, d->groups, Kg,
                       d->R * d->S, Cg, (float*)workspace, k_scale, w4);
  }
  IgemmArgs a = {};
  a.b_presplit = w4;
  a.in = dy; a.wm = wt; a.out = dx; a.shift = shift; a.relu = relu;
  a.N = d->N; a.IH = d->P; a.IW = d->Q; a.Ctot = d->K;      // the GEMM's "input" is dy
  a.OH = d->H; a.OW = d->W; a.OCtot = d->C;                 // its "output" is dx
  a.R = d->R; a.S = d->S;
  a.groups = d->groups; a.CgR = Kg; a.OCg = Cg;
  a.ksteps_per_tap = cpm::cdiv(a.CgR, BK);
  a.in_bytes = (unsigned)((size_t)d->N * d->P * d->Q * d->K * 4);
  a.wm_bytes = (unsigned)((size_t)d->K * d->R * d->S * Cg * 4);
  const int st = d->stride;
  // split-K / accumulate decision on the whole problem so that every phase agrees on atomics
  IgemmArgs whole = a;
  whole.M = d->N * d->H * d->W;
  whole.ksteps = cpm::cdiv(d->R, st) * cpm::cdiv(d->S, st) * a.ksteps_per_tap;
  Plan p = plan_igemm(whole);
  a.split_k = p.split;
  a.atomic_out = (a.split_k > 1) || accumulate;
  // accumulate without split-K: read-add-write through the (LDS-staged, 16-byte) residual epilogue with dx as its
  // own residual -- two streaming passes over dx instead of one float atomic per element
  a.scale = out_scale;
  a.mask = out_mask;
  const bool acc_via_res = accumulate && a.split_k == 1 && !shift && !relu;
  if (acc_via_res) {
    a.atomic_out = 0;
    a.res = dx;
    a.res_mode = 0;
  }
  // split-K partial sums in slab planes (plain stores, folded in split order with the epilogue in one pass) when the
  // workspace has room behind the weight image and every stride phase has taps (a phase without taps writes nothing)
  bool all_phases = true;
  for (int pa = 0; pa < st; ++pa)
    for (int pb = 0; pb < st; ++pb) {
      const int r0 = (pa + d->pad) % st, s0 = (pb + d->pad) % st;
      if ((d->H - pa + st - 1) / st <= 0 || (d->W - pb + st - 1) / st <= 0) continue;
      const int nr = r0 < d->R ? (d->R - r0 + st - 1) / st : 0, ns = s0 < d->S ? (d->S - s0 + st - 1) / st : 0;
      if (nr * ns * a.ksteps_per_tap < a.split_k) all_phases = false;   // (such a phase runs unsplit: one plane only)
    }
  const size_t plane = ((size_t)whole.M * a.OCtot + 63) / 64 * 64;
  a.slab = a.split_k > 1 ? slab_in(workspace, workspace_bytes, prepared ? 0 : need, a.split_k, plane, a.OCtot,
                                   !accumulate && (shift || relu || out_scale || out_mask))
                         : nullptr;
  a.slab_stride = plane;
  if (a.slab) {
    a.atomic_out = 1;
    // a stride phase without taps (or one that runs unsplit) leaves plane rows unwritten: they must read as zero
    if (!all_phases && hipMemsetAsync(a.slab, 0, (size_t)a.split_k * plane * sizeof(float), s) != hipSuccess) return CPM_ELAUNCH;
  }
  if (a.atomic_out && !accumulate && !a.slab && !out_ready) {     // (out_ready: cleared by the chain's previous kernel)
    if (hipMemsetAsync(dx, 0, (size_t)whole.M * a.OCtot * sizeof(float), s) != hipSuccess) return CPM_ELAUNCH;
  }
  g_flops_next = 2.0 * d->N * d->P * d->Q * (double)d->K * d->R * d->S * (d->C / d->groups);
  // the gate (the conv's own input, read where the gradient is masked) and the running sum of an accumulating call
  g_epi_bytes_next = 4.0 * d->N * d->H * d->W * (double)d->C * ((out_mask ? 1 : 0) + (accumulate ? 1 : 0));
  g_split_next = a.split_k;
  { const int dd[10] = {d->N, d->H, d->W, d->C, d->K, d->R, d->stride, d->groups, d->P, d->Q}; for (int i = 0; i < 10; ++i) g_dims_next[i] = dd[i]; }
  int rc = CPM_OK;
  for (int pa = 0; pa < st && rc == CPM_OK; ++pa) {
    for (int pb = 0; pb < st && rc == CPM_OK; ++pb) {
      a.OHp = (d->H - pa + st - 1) / st;
      a.OWp = (d->W - pb + st - 1) / st;
      if (a.OHp <= 0 || a.OWp <= 0) continue;
      a.r0 = (pa + d->pad) % st; a.s0 = (pb + d->pad) % st;
      a.rstep = a.sstep = st;
      a.nr = a.r0 < d->R ? (d->R - a.r0 + st - 1) / st : 0;
      a.ns = a.s0 < d->S ? (d->S - a.s0 + st - 1) / st : 0;
      a.ihmul = a.iwmul = 1;
      a.ihadd = (pa + d->pad - a.r0) / st; a.iwadd = (pb + d->pad - a.s0) / st;
      a.hstep = a.wstep = -1;
      a.osh = a.osw = st; a.oah = pa; a.oaw = pb;
      a.M = d->N * a.OHp * a.OWp;
      a.ksteps = a.nr * a.ns * a.ksteps_per_tap;
      if (a.ksteps == 0 && (a.atomic_out || acc_via_res)) continue;      // nothing to add
      Plan pp = plan_igemm(a);
      pp.split = a.split_k;
      if (a.ksteps < a.split_k) { a.split_k = 1; pp.split = 1; }
      rc = launch_igemm(a, pp, s, 1);
      a.split_k = p.split;
      g_flops_next = 0.0;                               // the call's flops are booked on its first launch
      g_epi_bytes_next = 0.0;
    }
  }
  if (rc != CPM_OK) return rc;
  if (a.slab) {
    const int64_t b = ((int64_t)whole.M * a.OCtot / 4 + 255) / 256;
    const dim3 rg((unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b)));
    const int64_t bs = ((int64_t)whole.M * a.OCtot + 255) / 256;
    const dim3 rgs((unsigned)(bs > 4096 ? 4096 : (bs < 1 ? 1 : bs)));
    if (a.OCtot & 3)
      hipLaunchKernelGGL(splitk_reduce_scalar_kernel, rgs, dim3(256), 0, s, a.slab, a.split_k, a.slab_stride, dx,
                         accumulate ? 1 : 0, out_scale, shift, (const float*)nullptr, (int64_t)whole.M, a.OCtot, a.OH,
                         a.OW, 0, relu, out_mask);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, rg, dim3(256), 0, s, a.slab, a.split_k, a.slab_stride, dx,
                         accumulate ? 1 : 0, out_scale, shift, (const float*)nullptr, (int64_t)whole.M, a.OCtot, a.OH,
                         a.OW, 0, relu, out_mask);
    rc = cpm::check_launch("dgrad split-K reduce");
  } else if (a.atomic_out && (shift || relu || out_scale || out_mask)) {
    const int64_t total = (int64_t)whole.M * a.OCtot;
    int64_t b = (total + 255) / 256;
    hipLaunchKernelGGL(epilogue_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, s, dx,
                       out_scale, shift, (const float*)nullptr, (int64_t)whole.M, a.OCtot, a.OH, a.OW, 0,
                       relu, out_mask);
    rc = cpm::check_launch("dgrad epilogue");
  }
  return rc;
}

CPM_EXPORT int cpm_conv2d_backward_data(const cpm_conv_desc* d, const float* dy, const float* w, float* dx,
                                        int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(dy && w && dx, "null pointer");
  CPM_REQUIRE(d->dilation == 1, "dilated dgrad not implemented");
  return run_dgrad(d, dy, w, dx, accumulate, nullptr, 0, workspace, workspace_bytes, (hipStream_t)stream,
                   "cpm_conv2d_backward_data");
}

CPM_EXPORT int cpm_conv2d_backward_data_gated(const cpm_conv_desc* d, const float* dy, const float* w, float* dx,
                                              const float* in_scale, const float* in_act, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(dy && w && dx, "null pointer");
  CPM_REQUIRE(d->dilation == 1, "dilated dgrad not implemented");
  return run_dgrad(d, dy, w, dx, 0, nullptr, 0, workspace, workspace_bytes, (hipStream_t)stream,
                   "cpm_conv2d_backward_data_gated", in_scale, in_act);
}

CPM_EXPORT int cpm_conv2d_backward_data_fused(const cpm_conv_desc* d, const float* dy, const float* w,
                                              const float* k_scale, float* dx, int accumulate, const float* in_scale,
                                              const float* in_act, void* workspace, size_t workspace_bytes,
                                              void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(dy && w && dx, "null pointer");
  CPM_REQUIRE(d->dilation == 1, "dilated dgrad not implemented");
  CPM_REQUIRE(!(accumulate && in_scale), "an accumulated data gradient takes a gate only: dx = (dx + W^T dy) * [act > 0]");
  return run_dgrad(d, dy, w, dx, accumulate, nullptr, 0, workspace, workspace_bytes, (hipStream_t)stream,
                   "cpm_conv2d_backward_data_fused", in_scale, in_act, false, k_scale);
}

CPM_EXPORT int cpm_conv2d_backward_data_prepared(const cpm_conv_desc* d, const float* dy, const float* wt, float* dx,
                                                 int accumulate, const float* in_scale, const float* in_act,
                                                 void* workspace, size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(dy && wt && dx, "null pointer");
  CPM_REQUIRE(d->dilation == 1, "dilated dgrad not implemented");
  CPM_REQUIRE(!(accumulate && in_scale), "an accumulated data gradient takes a gate only: dx = (dx + W^T dy) * [act > 0]");
  return run_dgrad(d, dy, wt, dx, accumulate, nullptr, 0, workspace, workspace_bytes, (hipStream_t)stream,
                   "cpm_conv2d_backward_data_prepared", in_scale, in_act, true);
}

// the same with `wt4` = the pre-split image (cpm_weights_to_dgrad_batched_w4; K / groups % 4 == 0): bf16x3 arithmetic only
CPM_EXPORT int cpm_conv2d_backward_data_prepared_w4(const cpm_conv_desc* d, const float* dy, const void* wt4, float* dx,
                                                    int accumulate, const float* in_scale, const float* in_act,
                                                    void* workspace, size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(dy && wt4 && dx, "null pointer");
  CPM_REQUIRE(d->dilation == 1, "dilated dgrad not implemented");
  CPM_REQUIRE(!(accumulate && in_scale), "an accumulated data gradient takes a gate only: dx = (dx + W^T dy) * [act > 0]");
  return run_dgrad(d, dy, (const float*)wt4, dx, accumulate, nullptr, 0, workspace, workspace_bytes, (hipStream_t)stream,
                   "cpm_conv2d_backward_data_prepared_w4", in_scale, in_act, true, nullptr, 1);
}

CPM_EXPORT int cpm_weights_to_dgrad_batched(const cpm_wt_desc* d_descs, int n, int64_t total_tiles, const float* src_base,
                                            float* dst_base, void* stream) {
  CPM_REQUIRE(n >= 0 && total_tiles >= 0, "bad counts");
  if (n == 0 || total_tiles == 0) return CPM_OK;
  CPM_REQUIRE(d_descs && src_base && dst_base, "null pointer");
  const int64_t b = total_tiles > 65536 ? 65536 : total_tiles;
  hipLaunchKernelGGL(weights_to_dgrad_batched, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, d_descs, n,
                     total_tiles, src_base, dst_base, 0);
  return cpm::check_launch("weights_to_dgrad_batched");
}

// the same with the images written pre-split (weights with K / groups % 4 == 0; the others keep their f32 image):
// what cpm_conv2d_backward_data_prepared_w4 takes
CPM_EXPORT int cpm_weights_to_dgrad_batched_w4(const cpm_wt_desc* d_descs, int n, int64_t total_tiles,
                                               const float* src_base, float* dst_base, void* stream) {
  CPM_REQUIRE(n >= 0 && total_tiles >= 0, "bad counts");
  if (n == 0 || total_tiles == 0) return CPM_OK;
  CPM_REQUIRE(d_descs && src_base && dst_base && ((uintptr_t)dst_base & 15) == 0, "null / unaligned pointer");
  const int64_t b = total_tiles > 65536 ? 65536 : total_tiles;
  hipLaunchKernelGGL(weights_to_dgrad_batched, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, d_descs, n,
                     total_tiles, src_base, dst_base, 1);
  return cpm::check_launch("weights_to_dgrad_batched_w4");
}

CPM_EXPORT int cpm_conv_transpose2d_forward(const cpm_conv_desc* d, const float* x, const float* w,
                                            const float* bias, int relu, float* y, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(x && w && y, "null pointer");
  CPM_REQUIRE(d->dilation == 1, "dilated transposed conv not implemented");
  return run_dgrad(d, x, w, y, 0, bias, relu, workspace, workspace_bytes, (hipStream_t)stream,
                   "cpm_conv_transpose2d_forward");
}

// tile and reduction split of a weight-gradient problem (shared by the launcher and the workspace query)
struct WgradPlan { int bm, bn, wm, wn, split; bool bf16; int ks = 1; bool taps = false; };   // ks: pixel runs per workgroup (wgrad_split_kernel)

static WgradArgs wgrad_args(const cpm_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias) {
  WgradArgs a = {};
  a.x = x; a.dy = dy; a.dw = dw; a.dshift = dbias;
  a.N = d->N; a.IH = d->H; a.IW = d->W; a.Ctot = d->C; a.OH = d->P; a.OW = d->Q; a.OCtot = d->K;
  a.R = d->R; a.S = d->S; a.stride = d->stride; a.pad = d->pad; a.dil = d->dilation;
  a.groups = d->groups; a.Cg = d->C / d->groups; a.OCg = d->K / d->groups;
  a.M = d->N * d->P * d->Q;
  a.chunks = cpm::cdiv(a.M, 32);
  a.x_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->C * 4);
  a.dy_bytes = (unsigned)((size_t)d->N * d->P * d->Q * d->K * 4);
  return a;
}

static WgradPlan plan_wgrad(const WgradArgs& a, bool wvec) {
  const int taps = a.R * a.S;
  auto blocks = [&](int bm, int bn) {
    return (int64_t)cpm::cdiv(a.OCg, bm) * cpm::cdiv(a.Cg, bn) * taps * a.groups;
  };
  // Split of the pixel reduction.  Fitted to tools/sweep_wgrad_split.sh on MI355X: a workgroup costs its chunks PLUS a
  // fixed ~9 chunk-times (prologue latency, the un-permuted 64 KB tile it writes, its share of the slab reduction), and
  // the grid runs in residency rounds of per_cu workgroups per CU, a round that is at most half full running at
  // 0.78 of a full one's time (a workgroup alone on a CU has the MFMA pipes to itself).  The earlier model (rounds / sk,
  // i.e. no fixed cost) split the RoI-head and layer4 gradients 3-4x too deep: 576x576x3x3 over 64 RoIs 155 -> 107 us,
  // 512x512x3x3 88 -> 60 us, fc6 232 -> 169 us.
  auto split_for = [&](int64_t nb, int per_cu, int ks = 1) {
    static const int fill = env_int("CPM_WGRAD_FILL", 100);      // percent of the chip's slots a launch plans for
    const int64_t slots = (int64_t)per_cu * num_cus() * fill / 100;
    const int maxs = a.chunks / 8 > 0 ? (a.chunks / 8 > 256 ? 256 : a.chunks / 8) : 1;
    const double fixed = (per_cu == 2 || ks == 2) ? 9.0 : 5.0;
    int best = 1;
    double best_cost = 1e30;
    for (int sk = 1; sk <= maxs; ++sk) {
      const int64_t nblk = nb * sk;
      const int64_t full = nblk / slots, tail = nblk % slots;
      const double rounds = (double)full + (tail == 0 ? 0.0 : (tail * 2 <= slots ? 0.78 : 1.0));
      const double cost = rounds * ((double)cpm::cdiv(a.chunks, sk * ks) + fixed);
      if (cost < best_cost - 1e-9) { best_cost = cost; best = sk; }
    }
    return best;
  };
  WgradPlan p;
  // three taps per workgroup (wgrad_taps_kernel): 3x3 / stride 1 / pad 1, whole 64-channel input tiles
  const int env_taps = env_int("CPM_WGRAD_TAPS", 1);          // read per call: 0 off, 1 by the rule below, 2 wherever eligible (tests)
  if (env_taps && wvec && g_conv_split && a.R == 3 && a.S == 3 && a.stride == 1 && a.dil == 1 && a.pad == 1 &&
      a.groups == 1 && a.Cg % 64 == 0 && a.OCg >= 128 && a.IH == a.OH && a.IW == a.OW && a.OW >= 3 &&
      a.OH >= 32 / a.OW + 2 && a.chunks >= 8) {
    // ... where a workgroup keeps a reduction of >= CPM_WGRAD_TAPS_MIN chunks: its three tiles (96 KB of partial sums,
    // against 64 KB per tap workgroup) make thin launches atomics-bound sooner than wgrad_split_kernel's
    static const int taps_min = env_int("CPM_WGRAD_TAPS_MIN", 16);
    const int sk = split_for((int64_t)cpm::cdiv(a.OCg, 128) * (a.Cg / 64) * 3, 2);
    if (env_taps == 2 || a.chunks / sk >= taps_min) {
      p = {128, 64, 2, 2, 1, true};
      p.taps = true;
      p.split = sk;
      if (const int forced = env_int("CPM_WGRAD_SPLIT", 0)) p.split = forced < a.chunks ? forced : a.chunks;
      return p;
    }
  }
  if (a.OCg <= 32 || a.Cg <= 32) {
    if (a.OCg <= 32 && a.Cg > 32) p = {32, 128, 1, 4, 1, false};
    else if (a.Cg <= 32 && a.OCg > 32) p = {128, 32, 4, 1, 1, false};
    else p = {64, 64, 2, 2, 1, false};
  } else if (a.OCg % 128 == 0 && a.Cg % 128 == 0 && a.chunks >= 64) {
    p = {128, 128, 2, 2, 1, false};     // the reduction (pixels) is split until the grid fills the chip
  } else if (g_conv_split && wvec && a.OCg >= 96 && a.Cg >= 96 && a.chunks >= 16) {
    // the split-bf16 kernel only pays off on the 128x128 tile (the 64x64 one is LDS/convert bound): ragged channel
    // counts (576 = 4.5 tiles) and short reductions with many tiles (FC layers) take it with masked edges
    p = {128, 128, 2, 2, 1, false};
  } else {
    p = {64, 64, 2, 2, 1, false};
  }
  // (64x64 tiles on the 1x1 layers, to quarter the atomic bytes of their 32-way splits: measured 15 % slower)
  p.bf16 = wvec && g_conv_split && p.bm % 64 == 0 && p.bn % 64 == 0;
  // Eight waves in two pixel runs per workgroup (KS = 2) where the launch is made of partial tiles: few output tiles,
  // deep splits.  tools/sweep_wgrad.py on MI355X, best split of either form: 256->1024 1x1 on 50x84 37.8 -> 34.4 us,
  // 128->512 1x1 on 100x168 43.9 -> 36.2, 512->2048 1x1 on 25x42 36.3 -> 33.5, 256->256 3x3 on 50x84 54.4 -> 49.9;
  // from ~90 tiles on the two forms tie, from 144 on (512->512 3x3 on 25x42 56.8 vs 64.5; 576->576 3x3 on 105 RoIs
  // 141 vs 153; fc6 124 vs 147) and on the long 36-tile reductions of P2 (523 vs 529) four-wave workgroups win.
  static const int env_ks = env_int("CPM_WGRAD_KS", 2);
  const int64_t nb = blocks(p.bm, p.bn);
  if (p.bf16 && p.bm == 128 && p.bn == 128 && env_ks == 2 && a.chunks >= 32 && nb <= 64 && !(nb >= 16 && a.chunks >= 2048)) {
    p.ks = 2;                           // the same waves per CU, half the partial tiles
    p.split = split_for(nb, 1, 2);
  } else {
    p.split = split_for(nb, p.bm * p.bn >= 128 * 128 ? 2 : 4);
  }
  if (const int forced = env_int("CPM_WGRAD_SPLIT", 0)) p.split = forced < a.chunks ? forced : a.chunks;   // sweeps
  return p;
}

static size_t wgrad_slab_bytes(const cpm_conv_desc* d) {
  if (d->C / d->groups == 1 && d->R * d->S <= 16) {                    // wgrad_cg1_kernel: one plane per pixel block
    const int64_t M = (int64_t)d->N * d->P * d->Q;
    int ppb = (int)((M * d->groups / (2 * num_cus()) + 63) / 64 * 64);
    ppb = ppb < 128 ? 128 : (ppb > 2048 ? 2048 : ppb);
    return (size_t)cpm::cdiv(M, ppb) * (((size_t)d->K * d->R * d->S + 63) / 64 * 64) * sizeof(float);
  }
  const WgradArgs a = wgrad_args(d, nullptr, nullptr, nullptr, nullptr);
  const bool wvec = (a.OCtot % 4 == 0) && (a.OCg % 4 == 0) && (a.Ctot % 4 == 0) && (a.Cg % 4 == 0);
  const WgradPlan p = plan_wgrad(a, wvec);
  return p.split > 1 ? (size_t)p.split * dgrad_weight_bytes(d) : 0;
}

static int run_wgrad(const cpm_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                     void* workspace, size_t workspace_bytes, hipStream_t s, const float* row_scale);
static int run_wgrad(const cpm_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                     void* workspace, size_t workspace_bytes, hipStream_t s) {
  return run_wgrad(d, x, dy, dw, dbias, workspace, workspace_bytes, s, nullptr);
}
static int run_wgrad(const cpm_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                     void* workspace, size_t workspace_bytes, hipStream_t s, const float* row_scale) {
  WgradArgs a = wgrad_args(d, x, dy, dw, dbias);
  a.row_scale = row_scale;
  const int taps = d->R * d->S;
  g_flops_next = 2.0 * d->N * d->P * d->Q * (double)d->K * d->R * d->S * (d->C / d->groups);
  g_epi_bytes_next = 0.0; g_split_next = 1;
  { const int dd[10] = {d->N, d->H, d->W, d->C, d->K, d->R, d->stride, d->groups, d->P, d->Q}; for (int i = 0; i < 10; ++i) g_dims_next[i] = dd[i]; }
  static const int nostore = env_int("CPM_WGRAD_NOSTORE", 0);
  a.debug_nostore = nostore;
  static const int wxcd = env_int("CPM_WGRAD_XCD", 1);
  a.xcd_swizzle = wxcd;
  ProfScope prof_scope(s, 2);
  if (a.Cg == 1 && taps <= 16) {
    if (dbias) {
      cpm::set_error("cpm_conv2d_backward_weight_bias: one input channel per group is not covered, use cpm_epilogue_backward");
      return CPM_EINVAL;
    }
    // pixels per workgroup: ~2 workgroups per CU -- every workgroup ends with OCg * taps float atomics on the same
    // few addresses, so fewer, longer workgroups (128 pixels each: 1215 of them for 88 RoIs) were atomics bound
    int ppb = (int)(((int64_t)a.M * a.groups / (2 * num_cus()) + 63) / 64 * 64);
    ppb = ppb < 128 ? 128 : (ppb > 2048 ? 2048 : ppb);
    dim3 grid((unsigned)cpm::cdiv(a.M, ppb), (unsigned)a.groups);
    // deterministic mode: every pixel block writes its own plane, folded in block order (instead of float atomics)
    const size_t dw_n = (size_t)d->K * taps, pl = (dw_n + 63) / 64 * 64;
    static const int det_env = env_int("CPM_DETERMINISTIC", 0);
    if ((g_deterministic || det_env) && workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= (size_t)grid.x * pl * sizeof(float)) {
      a.slab = (float*)workspace;
      a.slab_stride = pl;
    }
    hipLaunchKernelGGL((wgrad_cg1_kernel<16>), grid, dim3(256), 0, s, a, ppb);
    int rc1 = cpm::check_launch("conv wgrad (one channel per group)");
    if (rc1 == CPM_OK && a.slab) {
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cpm::cdiv((int64_t)dw_n / 4 + 1, 256)), dim3(256), 0, s, a.slab,
                         (int)grid.x, a.slab_stride, (int64_t)dw_n, dw);
      rc1 = cpm::check_launch("conv wgrad reduce (one channel per group)");
    }
    return rc1;
  }
  // wvec_x: the input side is vector-loadable; wvec: the gradient side too (the bf16x3 kernels need both).  The f32
  // tile kernel takes ragged output channel counts on its vector path by masking the tail of a row's last load.
  static const int tail_on = env_int("CPM_WGRAD_TAIL_VEC", 1);
  const bool wvec_x = (a.Ctot % 4 == 0) && (a.Cg % 4 == 0) && (((uintptr_t)a.x & 15) == 0) && (((uintptr_t)a.dy & 15) == 0);
  const bool wvec = wvec_x && (a.OCtot % 4 == 0) && (a.OCg % 4 == 0);
  const bool wvec_f32 = wvec || (wvec_x && tail_on && a.groups == 1 && a.OCg >= 4);
  const WgradPlan p = plan_wgrad(a, wvec);
  a.split_k = p.split;
  // Deterministic mode (cpm_set_deterministic / CPM_DETERMINISTIC / CPM_WGRAD_SLAB=1): a split reduction lands in one
  // slab plane per split (plain stores) and is folded into dw in split order -- bit-reproducible.  Default: the splits
  // add into dw with float atomics.  Measured both ways in the training step: in isolation the atomic epilogue costs
  // 23 % of the kernel (grid-head layer 130 -> 106 us with planes incl. the fold), but inside the step the fold pass
  // (98 launches, 1.36 ms for R-50; each reads split x dw) gives the gain back: R-50 30.0 vs 30.0 ms/step, R-101
  // 38.2-40.2 vs 36.8 ms/step (17 more blocks of thin 1x1 layers whose 32-way split planes cost more than their atomics).
  static const int env_slab = env_int("CPM_WGRAD_SLAB", env_int("CPM_DETERMINISTIC", 0));
  const bool use_slab = env_slab || g_deterministic;
  const size_t dw_elems = (size_t)d->K * d->R * d->S * (d->C / d->groups);
  const size_t plane = (dw_elems + 63) / 64 * 64;                     // 256-byte aligned planes
  if (use_slab && a.split_k > 1 && workspace && workspace_bytes >= (size_t)a.split_k * plane * sizeof(float) &&
      ((uintptr_t)workspace & 15) == 0) {
    a.slab = (float*)workspace;
    a.slab_stride = plane;
  }
  if (p.taps) {
    dim3 tgrid((unsigned)(cpm::cdiv(a.OCg, 128) * (a.Cg / 64)), 3, a.groups * a.split_k);
    hipLaunchKernelGGL(wgrad_taps_kernel, tgrid, dim3(256), 0, s, a);
    int rc = cpm::check_launch("conv wgrad (three taps per workgroup)");
    if (rc == CPM_OK && a.slab) {
      const int64_t n = (int64_t)dw_elems;
      int64_t b = (n / 4 + 255) / 256;
      b = b < 1 ? 1 : (b > 4096 ? 4096 : b);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)b), dim3(256), 0, s, a.slab, a.split_k, a.slab_stride, n, dw);
      rc = cpm::check_launch("conv wgrad reduce");
    }
    return rc;
  }
  dim3 grid((unsigned)(cpm::cdiv(a.OCg, p.bm) * cpm::cdiv(a.Cg, p.bn)), taps, a.groups * a.split_k);
  static const int wdbg = env_int("CPM_WGRAD_DBG", 0);
#define WCASE(BM, BN, WM, WN)                                                                        \
  if (p.bm == BM && p.bn == BN) {                                                                    \
    if (p.bf16 && wdbg && BM == 128 && BN == 128) {                                                  \
      if (wdbg == 1) hipLaunchKernelGGL((wgrad_split_kernel<128, 128, 2, 2, 1>), grid, dim3(256), 0, s, a);      \
      else if (wdbg == 2) hipLaunchKernelGGL((wgrad_split_kernel<128, 128, 2, 2, 2>), grid, dim3(256), 0, s, a); \
      else hipLaunchKernelGGL((wgrad_split_kernel<128, 128, 2, 2, 3>), grid, dim3(256), 0, s, a);                \
    } else if (p.bf16 && p.ks == 2 && BM == 128 && BN == 128)                                        \
      hipLaunchKernelGGL((wgrad_split_kernel<128, 128, 2, 2, 0, 2>), grid, dim3(512), 0, s, a);      \
    else if (p.bf16)                                                                                 \
      hipLaunchKernelGGL((wgrad_split_kernel<(BM) % 64 == 0 ? BM : 64, (BN) % 64 == 0 ? BN : 64, 2, 2>), grid, \
                         dim3(256), 0, s, a);                                                        \
    else if (wvec_f32)                                                                               \
      hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, true>), grid, dim3(256), 0, s, a);            \
    else                                                                                             \
      hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, false>), grid, dim3(256), 0, s, a);           \
  }
  WCASE(32, 128, 1, 4) else WCASE(128, 32, 4, 1) else WCASE(64, 64, 2, 2) else WCASE(128, 128, 2, 2)
#undef WCASE
  int rc = cpm::check_launch("conv wgrad");
  if (rc == CPM_OK && a.slab) {
    const int64_t n = (int64_t)dw_elems;
    int64_t b = (n / 4 + 255) / 256;
    b = b < 1 ? 1 : (b > 4096 ? 4096 : b);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)b), dim3(256), 0, s, a.slab, a.split_k, a.slab_stride, n, dw);
    rc = cpm::check_launch("conv wgrad reduce");
  }
  return rc;
}

CPM_EXPORT int cpm_conv2d_backward_weight(const cpm_conv_desc* d, const float* x, const float* dy, float* dw,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(x && dy && dw, "null pointer");
  return run_wgrad(d, x, dy, dw, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

CPM_EXPORT int cpm_conv2d_backward_weight_bias(const cpm_conv_desc* d, const float* x, const float* dy, float* dw,
                                               float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(x && dy && dw && dbias, "null pointer");
  return run_wgrad(d, x, dy, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream);
}

CPM_EXPORT int cpm_conv2d_backward_weight_scaled(const cpm_conv_desc* d, const float* x, const float* dy,
                                                 const float* k_scale, float* dw, float* dbias, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(validate(d) == CPM_OK, "bad conv descriptor");
  CPM_REQUIRE(x && dy && dw, "null pointer");
  return run_wgrad(d, x, dy, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream, k_scale);
}

// ---- profiling hooks (bench.py roofline leg) -----------------------------------------------------------
CPM_EXPORT int cpm_set_conv_math(int mode) {
  CPM_REQUIRE(mode == CPM_MATH_F32 || mode == CPM_MATH_BF16X3, "unknown conv math mode");
  g_conv_split = mode == CPM_MATH_BF16X3;
  return CPM_OK;
}

CPM_EXPORT int cpm_get_conv_math(void) { return g_conv_split ? CPM_MATH_BF16X3 : CPM_MATH_F32; }

CPM_EXPORT int cpm_set_deterministic(int on) {
  g_deterministic = on != 0;
  return CPM_OK;
}

CPM_EXPORT int cpm_get_deterministic(void) { return g_deterministic; }

CPM_EXPORT int cpm_prof_enable(int on) {
  for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof_on = on != 0;
  return CPM_OK;
}

CPM_EXPORT int cpm_prof_summary(int kind, double* total_ms, double* total_flops, int64_t* launches) {
  CPM_REQUIRE(total_ms && total_flops && launches, "null pointer");
  double ms = 0.0, fl = 0.0;
  int64_t n = 0;
  for (auto& r : g_prof) {
    if (r.kind != kind) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) return CPM_ELAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return CPM_ELAUNCH;
    ms += t; fl += r.flops; ++n;
  }
  *total_ms = ms; *total_flops = fl; *launches = n;
  return CPM_OK;
}

CPM_EXPORT int cpm_prof_dump(const char* path) {
  CPM_REQUIRE(path, "null path");
  FILE* f = fopen(path, "w");
  CPM_REQUIRE(f, "cannot open file");
  fprintf(f, "kind,N,H,W,C,K,R,stride,groups,P,Q,gflop,ms,epi_bytes,split\n");
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) { fclose(f); return CPM_ELAUNCH; }
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.a, r.b);
    fprintf(f, "%d", r.kind);
    for (int i = 0; i < 10; ++i) fprintf(f, ",%d", r.dims[i]);
    fprintf(f, ",%.4f,%.5f,%.0f,%d\n", r.flops / 1e9, t, r.epi_bytes, r.split);
  }
  fclose(f);
  return CPM_OK;
}
