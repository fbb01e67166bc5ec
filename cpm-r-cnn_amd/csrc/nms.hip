// Batched (multi-label) NMS entirely on the device, for gfx950.
//
// Semantics: pet/lib/ops/csrc/NMS/ml_nms.cu (IoU :19-25 without +1, same-label test :16,
// descending-score order :92-94, greedy sweep with topk early-out :127-140, indices into the
// caller's order :143-145); with labels == NULL it is torchvision.ops.nms as bound at
// pet/lib/ops/nms.py:2,10.  Ties in the score sort go to the lower input index.
//
// Structure (differs from the reference, which copies the whole N x N/64 bit mask to the host
// and sweeps it on one CPU core, ml_nms.cu:117-140):
//   1. per-segment bitonic sort of 64-bit keys {~orderable(score), index} in LDS (<= 16384 boxes;
//      larger segments use a global-memory bitonic network);
//   2. 64x64 suppression tiles: one wavefront per tile, one lane per row box, the 64-bit row word
//      is exactly one wave64 lane mask;
//   3. a single-wave sweep per segment that keeps the "removed" bitmap in LDS, resolves each
//      64-row diagonal tile serially with v_readlane and ORs the kept rows' words lane-parallel.
// All P segments (image x FPN level for the RPN, rpn/inference.py:67-114) share each launch.
// Built with -ffp-contract=off so the IoU expression rounds like the reference's.
#include "common.h"

namespace {

constexpr int MAX_SEG = 32;       // segments per launch (kernel-argument table)
constexpr int LDS_SORT_MAX = 16384;
constexpr int SWEEP_MAX_BLOCKS = 2048;  // 131072 boxes per segment

struct SegTable {
  int32_t off[MAX_SEG + 1];       // row offsets of the segments
  int64_t mask_off[MAX_SEG + 1];  // uint64 offsets of each segment's mask
  int P;
};

__device__ __forceinline__ uint64_t make_key(float score, uint32_t idx) {
  uint32_t u = __float_as_uint(score);
  if (u == 0x80000000u) u = 0;                            // -0 == +0
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);         // ascending orderable
  return ((uint64_t)(~u) << 32) | idx;                    // descending score, ascending index
}

// ---- 1a. LDS bitonic sort (n <= 16384), one workgroup per segment -------------------------------------
__global__ __launch_bounds__(1024) void sort_segments_lds(const float* __restrict__ scores, SegTable T,
                                                          int32_t* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = (uint64_t*)smem;
  const int p = blockIdx.x;
  const int base = T.off[p], n = T.off[p + 1] - base;
  if (n <= 0 || n > LDS_SORT_MAX) return;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = threadIdx.x; i < np2; i += blockDim.x)
    keys[i] = i < n ? make_key(scores[base + i], (uint32_t)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t a = keys[i], b = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) order[base + i] = (int32_t)(keys[i] & 0xffffffffu);
}

// ---- 1b. global bitonic network for big segments ----------------------------------------------------
__global__ void big_sort_init(const float* __restrict__ scores, int base, int n, int np2,
                              uint64_t* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np2) keys[i] = i < n ? make_key(scores[base + i], (uint32_t)i) : ~0ull;
}
__global__ void big_sort_step(uint64_t* __restrict__ keys, int np2, int j, int k) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np2) return;
  const int ixj = i ^ j;
  if (ixj > i) {
    const uint64_t a = keys[i], b = keys[ixj];
    const bool up = (i & k) == 0;
    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
  }
}
__global__ void big_sort_finish(const uint64_t* __restrict__ keys, int base, int n, int32_t* __restrict__ order) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) order[base + i] = (int32_t)(keys[i] & 0xffffffffu);
}

// ---- gather boxes / labels into sorted order ---------------------------------------------------------
__global__ void gather_sorted(const float* __restrict__ boxes, const int64_t* __restrict__ labels, SegTable T,
                              const int32_t* __restrict__ order, float4* __restrict__ sboxes,
                              int32_t* __restrict__ slabels) {
  const int p = blockIdx.y;
  const int base = T.off[p], n = T.off[p + 1] - base;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int src = base + order[base + i];
    sboxes[base + i] = *(const float4*)(boxes + 4 * (size_t)src);
    slabels[base + i] = labels ? (int32_t)labels[src] : 0;
  }
}

__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  const float inter = w * h;
  const float sa = (a.z - a.x) * (a.w - a.y);
  const float sb = (b.z - b.x) * (b.w - b.y);
  return (inter / (sa + sb - inter)) > thr;
}

// ---- 2. suppression tiles: grid (col_blocks, row_blocks, P), one wavefront per tile ------------------
__global__ __launch_bounds__(64) void mask_tiles(const float4* __restrict__ sboxes,
                                                 const int32_t* __restrict__ slabels, SegTable T, float thr,
                                                 uint64_t* __restrict__ mask) {
  const int p = blockIdx.z;
  const int base = T.off[p], n = T.off[p + 1] - base;
  const int nblk = (n + 63) >> 6;
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (rb >= nblk || cb >= nblk || rb > cb) return;
  __shared__ float4 cbox[64];
  __shared__ int32_t clab[64];
  const int lane = threadIdx.x;
  const int col_size = min(n - cb * 64, 64), row_size = min(n - rb * 64, 64);
  if (lane < col_size) {
    cbox[lane] = sboxes[base + cb * 64 + lane];
    clab[lane] = slabels[base + cb * 64 + lane];
  }
  __syncthreads();
  if (lane < row_size) {
    const int row = rb * 64 + lane;
    const float4 a = sboxes[base + row];
    const int32_t al = slabels[base + row];
    uint64_t t = 0;
    const int start = (rb == cb) ? lane + 1 : 0;
    for (int i = start; i < col_size; ++i)
      if (al == clab[i] && iou_gt(a, cbox[i], thr)) t |= 1ull << i;
    mask[T.mask_off[p] + (int64_t)row * nblk + cb] = t;
  }
}

// ---- 3. greedy sweep: one wavefront per segment -------------------------------------------------------
__global__ __launch_bounds__(64) void sweep_segments(const uint64_t* __restrict__ mask,
                                                     const int32_t* __restrict__ order, SegTable T, int topk,
                                                     int64_t* __restrict__ keep, int32_t* __restrict__ keep_count) {
  __shared__ uint64_t remv[SWEEP_MAX_BLOCKS];
  const int p = blockIdx.x;
  const int base = T.off[p], n = T.off[p + 1] - base;
  const int nblk = (n + 63) >> 6;
  const int lane = threadIdx.x;
  const uint64_t* m = mask + T.mask_off[p];
  for (int j = lane; j < nblk; j += 64) remv[j] = 0;
  __syncthreads();
  int nkeep = 0;
  bool done = false;
  // the diagonal words do not depend on the sweep: the next block's are fetched while this block is resolved
  uint64_t diag_next = lane < n ? m[(int64_t)lane * nblk] : 0;
  for (int blk = 0; blk < nblk && !done; ++blk) {
    const int row_l = blk * 64 + lane;
    // diagonal tile words, one per lane
    const uint64_t diag = diag_next;
    {
      const int rn = row_l + 64;
      diag_next = (blk + 1 < nblk && rn < n) ? m[(int64_t)rn * nblk + blk + 1] : 0;
    }
    uint64_t cur = remv[blk];
    const int rows_here = min(n - blk * 64, 64);
    uint64_t kept = 0;
    // wave-uniform control flow (`cur`, `alive`, `kept` are identical in every lane).  Only the rows that are still
    // alive are visited -- find-first-set on the alive mask -- instead of all 64: a kept row removes its victims from
    // the mask before they are reached.
    uint64_t alive = (rows_here == 64 ? ~0ull : ((1ull << rows_here) - 1ull)) & ~cur;
    while (alive) {
      const int b = __builtin_amdgcn_readfirstlane(__builtin_ctzll(alive));      // wave-uniform -> v_readlane below
      kept |= 1ull << b;
      cur |= ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(diag >> 32), b) << 32) |
             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)diag, b);
      alive &= ~cur;
      alive &= (b == 63) ? 0ull : ~((2ull << b) - 1ull);          // rows up to b are done
      if (topk > 0 && nkeep + __popcll(kept) >= topk) { done = true; break; }
    }
    // emit kept indices in order
    if ((kept >> lane) & 1ull) {
      const int pos = nkeep + __popcll(kept & ((1ull << lane) - 1ull));
      keep[base + pos] = (int64_t)order[base + row_l];
    }
    nkeep += __popcll(kept);
    if (done) break;
    // OR the kept rows into the words of the following blocks: lane b owns kept row b and streams that row's
    // remaining words (independent loads, all in flight together) into the LDS words with ds_or_b64 -- instead
    // of one lane per word walking up to 64 dependent loads
    if ((kept >> lane) & 1ull) {
      const uint64_t* rowp = m + (int64_t)row_l * nblk;
      // eight words per trip: the loads of a trip are issued together (one word per trip leaves every load waiting out
      // its full latency in front of the branch that consumes it)
      for (int j0 = blk + 1; j0 < nblk; j0 += 8) {
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (j0 + u < nblk) ? rowp[j0 + u] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (v[u]) atomicOr(reinterpret_cast<unsigned long long*>(&remv[j0 + u]), (unsigned long long)v[u]);
      }
    }
    __syncthreads();
  }
  if (lane == 0) keep_count[p] = nkeep;
}

// ---- 3b. segments of at most 2048 boxes (every RPN / RoI-head call): transposed tiles, no cross-lane reductions ------
// The sweep above asks "which later boxes does kept row r remove" and has to OR 64-bit words ACROSS the lanes that hold
// the kept rows: ~500 wave OR-reductions (DPP chains, ~0.2 us each) per 2000-box segment, 137-150 us on one
// wavefront.  The transposed question -- "is box c removed by an earlier kept box" -- needs no reduction at all: lane
// c holds, per earlier 64-box block rb, the word T[c][rb] of the boxes of rb that would suppress c (IoU is symmetric,
// so mask_tiles_t computes the same predicate calls as mask_tiles, stored the other way round), and
//     removed(c) = OR over rb of (T[c][rb] & kept[rb]) != 0
// is per-lane ANDs against the wave-uniform kept words.  Inside a block the rounds of the greedy resolution turn into
// two ballots each.  The rows of block b+1 are fetched while block b is resolved.
constexpr int SMALL_NB = 32;

__global__ __launch_bounds__(64) void mask_tiles_t(const float4* __restrict__ sboxes,
                                                   const int32_t* __restrict__ slabels, SegTable T, float thr,
                                                   uint64_t* __restrict__ mask) {
  const int p = blockIdx.z;
  const int base = T.off[p], n = T.off[p + 1] - base;
  const int nblk = (n + 63) >> 6;
  const int rb = blockIdx.y, cb = blockIdx.x;                  // earlier block rb, this box's block cb
  if (rb >= nblk || cb >= nblk || rb > cb) return;
  __shared__ float4 rbox[64];
  __shared__ int32_t rlab[64];
  const int lane = threadIdx.x;
  const int r_size = min(n - rb * 64, 64), c_size = min(n - cb * 64, 64);
  if (lane < r_size) {
    rbox[lane] = sboxes[base + rb * 64 + lane];
    rlab[lane] = slabels ? slabels[base + rb * 64 + lane] : 0;
  }
  __syncthreads();
  if (lane < c_size) {
    const int c = cb * 64 + lane;
    const float4 b = sboxes[base + c];
    const int32_t bl = slabels ? slabels[base + c] : 0;
    uint64_t t = 0;
    const int limit = (rb == cb) ? lane : r_size;              // only EARLIER boxes can suppress
    for (int i = 0; i < limit; ++i)
      if (bl == rlab[i] && iou_gt(rbox[i], b, thr)) t |= 1ull << i;     // the call mask_tiles makes for (row i, col c)
    mask[T.mask_off[p] + (int64_t)rb * (nblk * 64) + c] = t;   // [rb][box]: a block's 64 words are contiguous
  }
}

__global__ __launch_bounds__(64) void sweep_segments_small(const uint64_t* __restrict__ mask,
                                                           const int32_t* __restrict__ order, SegTable T, int topk,
                                                           int64_t* __restrict__ keep,
                                                           int32_t* __restrict__ keep_count) {
  const int p = blockIdx.x;
  const int base = T.off[p], n = T.off[p + 1] - base;
  const int nblk = (n + 63) >> 6;
  const int lane = threadIdx.x;
  const uint64_t* m = mask + T.mask_off[p];
  uint64_t kw[SMALL_NB];                                       // kept words of the finished blocks (wave-uniform)
#pragma unroll
  for (int j = 0; j < SMALL_NB; ++j) kw[j] = 0ull;
  // words rb <= blk of this lane's box in block blk (the others are never written by mask_tiles_t)
  // (the box's original index travels with its row: fetched a block ahead, not between the resolution and the store)
  auto load_row = [&](int blk, uint64_t (&w)[SMALL_NB], int32_t& ord) {
    const int row = blk * 64 + lane;
#pragma unroll
    for (int j = 0; j < SMALL_NB; ++j) {
      // unconditional loads (the mask is padded to whole blocks; words never written are discarded below): a per-lane
      // predicate here puts every load under its own exec-masked branch and the loads stop overlapping
      uint64_t v = 0ull;
      if (j <= blk) v = m[(int64_t)j * (nblk * 64) + row];       // wave-uniform condition (blk < nblk)
      w[j] = (row < n && j <= blk) ? v : 0ull;
    }
    ord = order ? order[base + (row < n ? row : n - 1)] : row;   // (no order list: the segment came sorted)
  };
  uint64_t wa[SMALL_NB], wb[SMALL_NB];
  int32_t oa = 0, ob = 0;
  load_row(0, wa, oa);
  int nkeep = 0;
  bool done = false;
  auto process = [&](int blk, const uint64_t (&w)[SMALL_NB], int32_t ord) {
    const int row_l = blk * 64 + lane;
    bool dead = false;
    uint64_t dt = 0;                                           // same-block boxes (earlier lanes) that suppress this one
#pragma unroll
    for (int j = 0; j < SMALL_NB; ++j) {
      if (j < blk) dead |= (w[j] & kw[j]) != 0ull;
      if (j == blk) dt = w[j];
    }
    uint64_t alive = __ballot(row_l < n && !dead);
    uint64_t kept = 0;
    // rounds: an alive box that no alive earlier box suppresses is kept; what the kept ones suppress is dead; repeat
    // (the lowest alive box always qualifies).  Two ballots per round.
    while (alive) {
      const uint64_t k = alive & ~__ballot((dt & alive) != 0ull);
      const uint64_t dead_now = __ballot((dt & k) != 0ull);
      kept |= k;
      alive &= ~(k | dead_now);
    }
    if (topk > 0 && nkeep + __popcll(kept) >= topk) {               // keep the first (topk - nkeep) of them
      int extra = nkeep + __popcll(kept) - topk;
      while (extra-- > 0) kept &= ~(1ull << (63 - __builtin_clzll(kept)));
      done = true;
    }
    if ((kept >> lane) & 1ull) {
      const int pos = nkeep + __popcll(kept & ((1ull << lane) - 1ull));
      keep[base + pos] = (int64_t)ord;
    }
    nkeep += __popcll(kept);
#pragma unroll
    for (int j = 0; j < SMALL_NB; ++j)
      if (j == blk) kw[j] = kept;
  };
  for (int blk = 0; blk < nblk && !done; blk += 2) {
    if (blk + 1 < nblk) load_row(blk + 1, wb, ob);
    process(blk, wa, oa);
    if (done || blk + 1 >= nblk) break;
    if (blk + 2 < nblk) load_row(blk + 2, wa, oa);
    process(blk + 1, wb, ob);
  }
  if (lane == 0) keep_count[p] = nkeep;
}

__global__ void box_iou_kernel(const float* __restrict__ boxes, int N, const float* __restrict__ query, int K,
                               float* __restrict__ out) {
  const int64_t total = (int64_t)N * K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = idx % K, i = idx / K;
    const float4 a = *(const float4*)(boxes + 4 * (size_t)i), b = *(const float4*)(query + 4 * (size_t)j);
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    const float sa = (a.z - a.x) * (a.w - a.y), sb = (b.z - b.x) * (b.w - b.y);
    out[idx] = inter / (sa + sb - inter);
  }
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct WsLayout {
  size_t order, sboxes, slabels, bigkeys, mask, total;
};

WsLayout ws_layout(const int32_t* off, int P) {
  WsLayout w;
  const size_t N = (size_t)off[P];
  size_t mask_words = 0, bigmax = 0;
  for (int p = 0; p < P; ++p) {
    const size_t n = (size_t)(off[p + 1] - off[p]);
    mask_words += ((n + 63) / 64) * 64 * ((n + 63) / 64);       // rows padded to whole 64-box blocks
    if (n > (size_t)LDS_SORT_MAX) {
      size_t np2 = 1;
      while (np2 < n) np2 <<= 1;
      if (np2 > bigmax) bigmax = np2;
    }
  }
  size_t o = 0;
  w.order = o;   o = align_up(o + N * sizeof(int32_t), 256);
  w.sboxes = o;  o = align_up(o + N * sizeof(float4), 256);
  w.slabels = o; o = align_up(o + N * sizeof(int32_t), 256);
  w.bigkeys = o; o = align_up(o + bigmax * sizeof(uint64_t), 256);
  w.mask = o;    o = align_up(o + mask_words * sizeof(uint64_t), 256);
  w.total = o + 256;
  return w;
}

}  // namespace

CPM_EXPORT size_t cpm_nms_workspace_bytes(const int32_t* h_offsets, int P) {
  if (!h_offsets || P <= 0) return 0;
  return ws_layout(h_offsets, P).total;
}

static int nms_batched_impl(const float* boxes, const float* scores, const int64_t* labels,
                            const int32_t* h_offsets, int P, float iou_threshold, int topk, int64_t* keep,
                            int32_t* keep_count, void* workspace, size_t workspace_bytes, void* stream, bool presorted) {
  CPM_REQUIRE(h_offsets && P > 0, "segment table");
  CPM_REQUIRE(keep_count, "null keep_count");
  hipStream_t s = (hipStream_t)stream;
  for (int p = 0; p < P; ++p) {
    CPM_REQUIRE(h_offsets[p + 1] >= h_offsets[p], "offsets must be non-decreasing");
    CPM_REQUIRE(h_offsets[p + 1] - h_offsets[p] <= SWEEP_MAX_BLOCKS * 64, "segment larger than 131072 boxes");
  }
  CPM_REQUIRE(h_offsets[0] == 0, "offsets must start at 0");
  const int N = h_offsets[P];
  if (N == 0) {
    (void)hipMemsetAsync(keep_count, 0, sizeof(int32_t) * P, s);
    return CPM_OK;
  }
  CPM_REQUIRE(boxes && scores && keep, "null pointer");
  CPM_REQUIRE(((uintptr_t)boxes & 15) == 0, "boxes must be 16-byte aligned");
  WsLayout w = ws_layout(h_offsets, P);
  if (!workspace || workspace_bytes < w.total) {
    cpm::set_error("cpm_nms_batched: workspace %zu < %zu", workspace_bytes, w.total);
    return CPM_EWORKSPACE;
  }
  char* ws = (char*)workspace;
  int32_t* order = (int32_t*)(ws + w.order);
  float4* sboxes = (float4*)(ws + w.sboxes);
  int32_t* slabels = (int32_t*)(ws + w.slabels);
  uint64_t* bigkeys = (uint64_t*)(ws + w.bigkeys);
  uint64_t* mask = (uint64_t*)(ws + w.mask);

  int64_t mask_base = 0;
  for (int p0 = 0; p0 < P; p0 += MAX_SEG) {
    const int np = (P - p0) < MAX_SEG ? (P - p0) : MAX_SEG;
    SegTable T;
    T.P = np;
    int maxn = 0;
    for (int i = 0; i <= np; ++i) T.off[i] = h_offsets[p0 + i];
    for (int i = 0; i < np; ++i) {
      const int64_t n = T.off[i + 1] - T.off[i];
      T.mask_off[i] = mask_base;
      mask_base += ((n + 63) / 64) * 64 * ((n + 63) / 64);
      if (n > maxn) maxn = (int)n;
    }
    T.mask_off[np] = mask_base;
    if (maxn == 0) {
      (void)hipMemsetAsync(keep_count + p0, 0, sizeof(int32_t) * np, s);
      continue;
    }
    // segments that arrive in descending score order (the RPN's: the pre-NMS top-k leaves them sorted), unlabelled and
    // small enough for the transposed tiles: the stable sort is the identity -- no sort, no gather, the tiles read the
    // caller's boxes
    if (presorted && !labels && maxn <= SMALL_NB * 64 && (((uintptr_t)boxes) & 15) == 0) {
      const int nblk = (maxn + 63) / 64;
      hipLaunchKernelGGL(mask_tiles_t, dim3(nblk, nblk, np), dim3(64), 0, s, (const float4*)boxes,
                         (const int32_t*)nullptr, T, iou_threshold, mask);
      hipLaunchKernelGGL(sweep_segments_small, dim3(np), dim3(64), 0, s, mask, (const int32_t*)nullptr, T, topk, keep,
                         keep_count + p0);
      continue;
    }
    // 1. sort
    {
      int lds_n = maxn < LDS_SORT_MAX ? maxn : LDS_SORT_MAX;
      int np2 = 1;
      while (np2 < lds_n) np2 <<= 1;
      const size_t lds_bytes = (size_t)np2 * sizeof(uint64_t);
      if (lds_bytes > 65536)
        (void)hipFuncSetAttribute((const void*)sort_segments_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes);
      hipLaunchKernelGGL(sort_segments_lds, dim3(np), dim3(1024), lds_bytes, s, scores, T, order);
      for (int i = 0; i < np; ++i) {
        const int n = T.off[i + 1] - T.off[i];
        if (n <= LDS_SORT_MAX) continue;
        int b2 = 1;
        while (b2 < n) b2 <<= 1;
        const int blocks = b2 / 256;
        hipLaunchKernelGGL(big_sort_init, dim3(blocks), dim3(256), 0, s, scores, T.off[i], n, b2, bigkeys);
        for (int k = 2; k <= b2; k <<= 1)
          for (int j = k >> 1; j > 0; j >>= 1)
            hipLaunchKernelGGL(big_sort_step, dim3(blocks), dim3(256), 0, s, bigkeys, b2, j, k);
        hipLaunchKernelGGL(big_sort_finish, dim3(cpm::cdiv(n, 256)), dim3(256), 0, s, bigkeys, T.off[i], n, order);
      }
    }
    hipLaunchKernelGGL(gather_sorted, dim3(cpm::cdiv(maxn, 256) > 64 ? 64 : cpm::cdiv(maxn, 256), np), dim3(256), 0,
                       s, boxes, labels, T, order, sboxes, slabels);
    // 2. tiles + 3. sweep
    const int nblk = (maxn + 63) / 64;
    if (maxn <= SMALL_NB * 64) {
      hipLaunchKernelGGL(mask_tiles_t, dim3(nblk, nblk, np), dim3(64), 0, s, sboxes, slabels, T, iou_threshold, mask);
      hipLaunchKernelGGL(sweep_segments_small, dim3(np), dim3(64), 0, s, mask, order, T, topk, keep, keep_count + p0);
    } else {
      hipLaunchKernelGGL(mask_tiles, dim3(nblk, nblk, np), dim3(64), 0, s, sboxes, slabels, T, iou_threshold, mask);
      hipLaunchKernelGGL(sweep_segments, dim3(np), dim3(64), 0, s, mask, order, T, topk, keep, keep_count + p0);
    }
  }
  return cpm::check_launch("nms_batched");
}

CPM_EXPORT int cpm_nms_batched(const float* boxes, const float* scores, const int64_t* labels,
                               const int32_t* h_offsets, int P, float iou_threshold, int topk, int64_t* keep,
                               int32_t* keep_count, void* workspace, size_t workspace_bytes, void* stream) {
  return nms_batched_impl(boxes, scores, labels, h_offsets, P, iou_threshold, topk, keep, keep_count, workspace,
                          workspace_bytes, stream, false);
}

CPM_EXPORT int cpm_nms_batched_presorted(const float* boxes, const float* scores, const int32_t* h_offsets, int P,
                                         float iou_threshold, int topk, int64_t* keep, int32_t* keep_count,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return nms_batched_impl(boxes, scores, nullptr, h_offsets, P, iou_threshold, topk, keep, keep_count, workspace,
                          workspace_bytes, stream, true);
}

CPM_EXPORT int cpm_box_iou(const float* boxes, int N, const float* query, int K, float* out, void* stream) {
  CPM_REQUIRE(N >= 0 && K >= 0, "bad shape");
  if (N == 0 || K == 0) return CPM_OK;
  CPM_REQUIRE(boxes && query && out, "null pointer");
  int64_t total = (int64_t)N * K;
  int64_t b = (total + 255) / 256;
  hipLaunchKernelGGL(box_iou_kernel, dim3((int)(b > 4096 ? 4096 : b)), dim3(256), 0, (hipStream_t)stream, boxes, N,
                     query, K, out);
  return cpm::check_launch("box_iou");
}
