// Segmented top-k (k <= 2048) for the RPN proposal selection: per (image, level) the pre_nms_top_n best objectness
// scores of up to ~200k anchors (pet/rcnn/modeling/rpn/inference.py:79-84 `objectness.topk(pre_nms_top_n, dim=1,
// sorted=True)`), all rows of a level in ONE launch.
//
// One workgroup of 1024 lanes per segment.  Radix select on the order-preserving 32-bit image of the float: four
// histogram passes of 8 bits pin the k-th largest key exactly.  Objectness scores crowd into a few exponent bins, so
// a single LDS histogram would serialise 64 lanes on one address; each bin therefore has 32 lane-private copies
// (bin*32 + lane%32: distinct banks, at most two lanes per address).  A fifth pass appends
// every element above it (and the ties at it, lowest index first when there are more ties than places) to an LDS
// list, which a bitonic network then orders by (score descending, index ascending) -- the same composite key the
// NMS stage sorts by (nms.hip make_key), so equal scores keep a defined order end to end.  Traffic: 5 reads of
// the segment, 12 bytes written per selected element.
#include "common.h"
#include <algorithm>

namespace {

constexpr int TOPK_MAX = 2048;
constexpr int TOPK_THREADS = 1024;

__device__ __forceinline__ uint32_t order_key(float v) {
  uint32_t u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0;                            // -0 == +0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending in the float order
}

constexpr int COPIES = 32;

// A place in an LDS list for every lane that raises `flag`: one atomic per wavefront (the lanes of a wavefront that
// reach this call together), not one per lane -- 2 000 selected elements took 2 000 turns on one LDS word.
__device__ __forceinline__ int wave_slot(bool flag, int* counter) {
  const unsigned long long m = __ballot(flag);
  if (m == 0ull) return -1;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, leader, 64);
  return flag ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// f(value, index) over one row, every element once.  16-byte loads, two in flight per lane, when the row is aligned:
// a single workgroup per row lives on memory-level parallelism.
template <class F>
__device__ __forceinline__ void scan_row(const float* __restrict__ row, int n, F f) {
  const int tid = threadIdx.x;
  if ((((uintptr_t)row) & 15) == 0) {
    const int n4 = n >> 2;
    const float4* r4 = (const float4*)row;
    int i = tid;
    for (; i + 3 * TOPK_THREADS < n4; i += 4 * TOPK_THREADS) {
      const float4 a = r4[i], b = r4[i + TOPK_THREADS], c = r4[i + 2 * TOPK_THREADS], d = r4[i + 3 * TOPK_THREADS];
      f(a.x, 4 * i); f(a.y, 4 * i + 1); f(a.z, 4 * i + 2); f(a.w, 4 * i + 3);
      const int j = i + TOPK_THREADS;
      f(b.x, 4 * j); f(b.y, 4 * j + 1); f(b.z, 4 * j + 2); f(b.w, 4 * j + 3);
      const int l = j + TOPK_THREADS;
      f(c.x, 4 * l); f(c.y, 4 * l + 1); f(c.z, 4 * l + 2); f(c.w, 4 * l + 3);
      const int m = l + TOPK_THREADS;
      f(d.x, 4 * m); f(d.y, 4 * m + 1); f(d.z, 4 * m + 2); f(d.w, 4 * m + 3);
    }
    for (; i < n4; i += TOPK_THREADS) {
      const float4 a = r4[i];
      f(a.x, 4 * i); f(a.y, 4 * i + 1); f(a.z, 4 * i + 2); f(a.w, 4 * i + 3);
    }
    for (int j = (n4 << 2) + tid; j < n; j += TOPK_THREADS) f(row[j], j);
  } else {
    for (int i = tid; i < n; i += TOPK_THREADS) f(row[i], i);
  }
}

// Ascending bitonic sort of the first k (<= 2048) entries of an LDS list by a workgroup of 1024 threads.  Thread t holds
// elements t and t + 1024 in registers (places >= k count as the largest key): a compare-exchange whose partner is
// t ^ stride is a register pair (stride 1024), a wavefront shuffle (stride < 64: 51 of the 66 steps) or an exchange
// through the list with a barrier (strides 64 .. 512: 14 steps).  The plain network -- every step through LDS behind a
// barrier -- took ~20 us of a one-workgroup kernel for 2 000 keys.
__device__ __forceinline__ void bitonic_sort_2048(unsigned long long* list, int k) {
  const int tid = threadIdx.x;
  unsigned long long e[2] = {tid < k ? list[tid] : ~0ull, tid + 1024 < k ? list[tid + 1024] : ~0ull};
  __syncthreads();
  for (int size = 2; size <= 2048; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride == 1024) {                                       // (size 2048: one ascending run)
        const unsigned long long a = e[0], b = e[1];
        e[0] = a < b ? a : b;
        e[1] = a < b ? b : a;
        continue;
      }
      unsigned long long p[2];
      if (stride < 64) {
        p[0] = __shfl_xor(e[0], stride, 64);
        p[1] = __shfl_xor(e[1], stride, 64);
      } else {
        list[tid] = e[0];
        list[tid + 1024] = e[1];
        __syncthreads();
        p[0] = list[tid ^ stride];
        p[1] = list[(tid ^ stride) + 1024];
        __syncthreads();
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = tid + h * 1024;
        const bool up = (i & size) == 0, low = (i & stride) == 0;   // ascending run; the pair's lower place
        const unsigned long long mn = e[h] < p[h] ? e[h] : p[h], mx = e[h] < p[h] ? p[h] : e[h];
        e[h] = (low == up) ? mn : mx;
      }
    }
  }
  list[tid] = e[0];
  list[tid + 1024] = e[1];
  __syncthreads();
}

// From the 256 x COPIES lane-private counters: the bin b with  count(bins > b) < want <= count(bins >= b), and
// count(bins > b).  `tot` (256 ints) is scratch.  All threads call; results land in LDS scalars.
__device__ void find_bin(const int* hist, int* tot, int want, int* out_bin, int* out_above) {
  const int tid = threadIdx.x;
  if (tid < 256) {
    int s = 0;
    for (int c = 0; c < COPIES; ++c) s += hist[tid * COPIES + ((c + tid) & (COPIES - 1))];   // rotated: no bank conflict
    tot[tid] = s;
  }
  __syncthreads();
  if (tid < 64) {
    const int lane = tid;
    int s = tot[4 * lane] + tot[4 * lane + 1] + tot[4 * lane + 2] + tot[4 * lane + 3];
    int incl = s;                                          // suffix sums over lanes (higher lane = larger keys)
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_down(incl, d, 64);
      if (lane + d < 64) incl += t;
    }
    const int above_lane = incl - s;
    if (above_lane < want && want <= incl) {               // exactly one lane (1 <= want <= matching elements)
      int above = above_lane;
      for (int i = 3; i >= 0; --i) {
        const int c = tot[4 * lane + i];
        if (above + c >= want) {
          *out_bin = 4 * lane + i;
          *out_above = above;
          break;
        }
        above += c;
      }
    }
  }
}

constexpr int TOPK_LEVELS = 8;
// blockIdx.y picks the problem: the RPN selects on every FPN level, each a [rows][n_l] matrix of its own
struct TopkLevels {
  const float* scores[TOPK_LEVELS];
  float* out_scores[TOPK_LEVELS];
  int64_t* out_idx[TOPK_LEVELS];
  int n[TOPK_LEVELS], k[TOPK_LEVELS];
};

// Long rows in slices.  One workgroup per row lives on its own memory-level parallelism: the RPN's finest level
// (201 600 anchors per image) took 140 us of a 19 ms step with the rest of the chip idle.  With `cand` given, a row of
// at least 2 x TOPK_SLICE_MIN elements is cut into up to TOPK_SLICES slices (blockIdx.z): each slice's workgroup
// selects ITS k best the same way and leaves them unsorted in cand as (~key << 32 | index in the row); the row's
// k best are the k smallest of those composites (topk_merge_kernel).  Shorter rows are finished here as before.
constexpr int TOPK_SLICES = 8;
constexpr int TOPK_SLICE_MIN = 16384;

__host__ __device__ inline int topk_slices(int n) {
  const int s = n / TOPK_SLICE_MIN;
  return s < 2 ? 1 : (s > TOPK_SLICES ? TOPK_SLICES : s);
}
__host__ __device__ inline int topk_slice_len(int n, int slices) { return ((n + slices - 1) / slices + 3) & ~3; }

__global__ void __launch_bounds__(TOPK_THREADS) topk_rows_kernel(TopkLevels lv, unsigned long long* __restrict__ cand) {
  const float* __restrict__ scores = lv.scores[blockIdx.y];
  float* __restrict__ out_scores = lv.out_scores[blockIdx.y];
  int64_t* __restrict__ out_idx = lv.out_idx[blockIdx.y];
  const int n_row = lv.n[blockIdx.y], k_row = lv.k[blockIdx.y];
  const int slices = cand ? topk_slices(n_row) : 1;
  if ((int)blockIdx.z >= slices) return;
  const int slice_len = slices > 1 ? topk_slice_len(n_row, slices) : n_row;
  const int lo = (int)blockIdx.z * slice_len;              // (a multiple of 4: the slice is aligned like the row)
  const int n = min(n_row, lo + slice_len) - lo;
  const int k = min(k_row, n);
  __shared__ int hist[256 * COPIES];
  __shared__ int tot[256];
  __shared__ unsigned long long sel[TOPK_MAX];
  __shared__ int s_bin, s_above, s_count, s_ties;
  const float* row = scores + (int64_t)blockIdx.x * n_row + lo;
  const int tid = threadIdx.x;

  uint32_t prefix = 0;           // key bits fixed so far (in place)
  uint32_t prefix_mask = 0;
  int want = k;                  // rank still to locate among elements matching the prefix
  const int copy = tid & (COPIES - 1);
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256 * COPIES; i += TOPK_THREADS) hist[i] = 0;
    __syncthreads();
    scan_row(row, n, [&](float v, int) {
      const uint32_t key = order_key(v);
      if ((key & prefix_mask) == prefix) atomicAdd(&hist[((key >> shift) & 255u) * COPIES + copy], 1);
    });
    __syncthreads();
    find_bin(hist, tot, want, &s_bin, &s_above);
    __syncthreads();
    prefix |= (uint32_t)s_bin << shift;
    prefix_mask |= 255u << shift;
    want -= s_above;
    __syncthreads();
  }
  // prefix == key of the k-th largest element; `want` (>= 1) of the elements equal to it are still to be taken
  const uint32_t kth = prefix;
  if (tid == 0) { s_count = 0; s_ties = 0; }
  __syncthreads();
  scan_row(row, n, [&](float v, int i) {
    const uint32_t key = order_key(v);
    const int p = wave_slot(key > kth, &s_count);
    if (p >= 0) sel[p] = ((unsigned long long)(~key) << 32) | (uint32_t)i;
    // ties can be most of the row (saturated scores): one LDS atomic per wavefront, not per lane
    const unsigned long long m = __ballot(key == kth);
    if (key == kth && (m & ((1ull << (tid & 63)) - 1ull)) == 0) atomicAdd(&s_ties, __popcll(m));
  });
  __syncthreads();
  const int above = s_count;                    // == k - want
  if (s_ties == want) {                         // every tie has a place: order does not matter here, the sort fixes it
    scan_row(row, n, [&](float v, int i) {
      const uint32_t key = order_key(v);
      const int p = wave_slot(key == kth, &s_count);
      if (p >= 0) sel[p] = ((unsigned long long)(~key) << 32) | (uint32_t)i;
    });
  } else {
    // more ties than places: the lowest indices win.  Walk the row in index order, 1024 elements at a time, with a
    // ballot + prefix count over the workgroup.
    __shared__ int wave_cnt[TOPK_THREADS / 64];
    int taken = 0;                              // uniform across the workgroup
    for (int base = 0; base < n && taken < want; base += TOPK_THREADS) {
      const int i = base + tid;
      const bool tie = i < n && order_key(row[i]) == kth;
      const unsigned long long b = __ballot(tie);
      const int lane = tid & 63, wv = tid >> 6;
      if (lane == 0) wave_cnt[wv] = __popcll(b);
      __syncthreads();
      int before = 0, total = 0;
      for (int w = 0; w < TOPK_THREADS / 64; ++w) {
        const int c = wave_cnt[w];
        if (w < wv) before += c;
        total += c;
      }
      const int r = taken + before + __popcll(b & ((1ull << lane) - 1ull));
      if (tie && r < want) sel[above + r] = ((unsigned long long)(~kth) << 32) | (uint32_t)i;
      taken += total;
      __syncthreads();
    }
  }
  __syncthreads();
  if (slices > 1) {
    unsigned long long* out = cand + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * TOPK_SLICES + blockIdx.z) * TOPK_MAX;
    for (int j = tid; j < k; j += TOPK_THREADS) out[j] = sel[j] + (unsigned long long)lo;     // index in the row
    return;
  }
  // the k composite keys in order (ascending == score descending, index ascending)
  bitonic_sort_2048(sel, k);
  float* os = out_scores + (int64_t)blockIdx.x * k;
  int64_t* oi = out_idx + (int64_t)blockIdx.x * k;
  for (int j = tid; j < k; j += TOPK_THREADS) {
    const uint32_t idx = (uint32_t)(sel[j] & 0xffffffffull);
    os[j] = row[idx];
    oi[j] = (int64_t)idx;
  }
}

// The k smallest of a sliced row's candidate composites (they are distinct: the index is their low word), sorted.
// <= TOPK_SLICES x TOPK_MAX candidates, 16 per thread in registers; radix select on the complemented composite (so that
// find_bin's "largest" is the smallest), digit by digit from the top, stopping as soon as the chosen bin holds exactly
// what is still wanted -- without ties on the k-th score that is after the four score digits.
__global__ void __launch_bounds__(TOPK_THREADS) topk_merge_kernel(TopkLevels lv, const unsigned long long* __restrict__ cand) {
  const int n_row = lv.n[blockIdx.y], k = lv.k[blockIdx.y];
  const int slices = topk_slices(n_row);
  if (slices == 1) return;
  __shared__ int hist[256 * COPIES];
  __shared__ int tot[256];
  __shared__ unsigned long long sel[TOPK_MAX];
  __shared__ int s_bin, s_above, s_count;
  const int tid = threadIdx.x;
  const int slice_len = topk_slice_len(n_row, slices);
  const unsigned long long* base = cand + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * TOPK_SLICES * TOPK_MAX;
  constexpr int PER = TOPK_SLICES * TOPK_MAX / TOPK_THREADS;        // 16
  unsigned long long d[PER];                                          // ~composite, 0 = no candidate
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + i * TOPK_THREADS, z = e / TOPK_MAX, j = e - z * TOPK_MAX;
    const int len = min(n_row, (z + 1) * slice_len) - z * slice_len;
    d[i] = (z < slices && j < min(k, len)) ? ~base[(size_t)z * TOPK_MAX + j] : 0ull;
  }
  unsigned long long prefix = 0, mask = 0;
  int want = k;
  const int copy = tid & (COPIES - 1);
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256 * COPIES; i += TOPK_THREADS) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (d[i] != 0ull && (d[i] & mask) == prefix) atomicAdd(&hist[(int)((d[i] >> shift) & 255ull) * COPIES + copy], 1);
    __syncthreads();
    find_bin(hist, tot, want, &s_bin, &s_above);
    __syncthreads();
    prefix |= (unsigned long long)s_bin << shift;
    mask |= 255ull << shift;
    want -= s_above;
    const bool done = tot[s_bin] == want;                           // the whole bin is taken: nothing left to split
    __syncthreads();
    if (done) break;
  }
  if (tid == 0) s_count = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i)
  {
    const int p = wave_slot(d[i] != 0ull && (d[i] & mask) >= prefix, &s_count);
    if (p >= 0) sel[p] = ~d[i];
  }
  __syncthreads();
  bitonic_sort_2048(sel, k);
  const float* row = lv.scores[blockIdx.y] + (int64_t)blockIdx.x * n_row;
  float* os = lv.out_scores[blockIdx.y] + (int64_t)blockIdx.x * k;
  int64_t* oi = lv.out_idx[blockIdx.y] + (int64_t)blockIdx.x * k;
  for (int j = tid; j < k; j += TOPK_THREADS) {
    const uint32_t idx = (uint32_t)(sel[j] & 0xffffffffull);
    os[j] = row[idx];
    oi[j] = (int64_t)idx;
  }
}

}  // namespace

CPM_EXPORT int cpm_topk_rows(const float* scores, int rows, int n, int k, float* out_scores, int64_t* out_idx,
                             void* stream) {
  CPM_REQUIRE(rows >= 0 && n > 0, "bad shape");
  CPM_REQUIRE(k >= 1 && k <= n && k <= TOPK_MAX, "k must be in [1, min(n, 2048)]");
  CPM_REQUIRE((int64_t)rows * n < (1ll << 31), "too many elements");
  if (rows == 0) return CPM_OK;
  CPM_REQUIRE(scores && out_scores && out_idx, "null pointer");
  TopkLevels lv = {};
  lv.scores[0] = scores; lv.out_scores[0] = out_scores; lv.out_idx[0] = out_idx; lv.n[0] = n; lv.k[0] = k;
  hipLaunchKernelGGL(topk_rows_kernel, dim3(rows, 1), dim3(TOPK_THREADS), 0, (hipStream_t)stream, lv,
                     (unsigned long long*)nullptr);
  return cpm::check_launch("topk_rows");
}

CPM_EXPORT size_t cpm_topk_rows_multi_workspace_bytes(int levels, int rows) {
  if (levels < 1 || rows < 1) return 0;
  return (size_t)levels * rows * TOPK_SLICES * TOPK_MAX * sizeof(unsigned long long);
}

CPM_EXPORT int cpm_topk_rows_multi(const float* const* scores, const int* n, const int* k, int levels, int rows,
                                   float* const* out_scores, int64_t* const* out_idx, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  CPM_REQUIRE(levels >= 1 && levels <= TOPK_LEVELS && rows >= 0, "1 <= levels <= 8");
  CPM_REQUIRE(scores && n && k && out_scores && out_idx, "null pointer");
  if (rows == 0) return CPM_OK;
  TopkLevels lv = {};
  for (int l = 0; l < levels; ++l) {
    CPM_REQUIRE(n[l] > 0 && k[l] >= 1 && k[l] <= n[l] && k[l] <= TOPK_MAX, "k must be in [1, min(n, 2048)]");
    CPM_REQUIRE((int64_t)rows * n[l] < (1ll << 31), "too many elements");
    CPM_REQUIRE(scores[l] && out_scores[l] && out_idx[l], "null pointer");
    lv.scores[l] = scores[l]; lv.out_scores[l] = out_scores[l]; lv.out_idx[l] = out_idx[l];
    lv.n[l] = n[l]; lv.k[l] = k[l];
  }
  int max_slices = 1;
  for (int l = 0; l < levels; ++l) max_slices = std::max(max_slices, topk_slices(n[l]));
  const bool sliced = max_slices > 1 && workspace && (((uintptr_t)workspace & 7) == 0) &&
                      workspace_bytes >= cpm_topk_rows_multi_workspace_bytes(levels, rows);
  if (!sliced) {
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows, levels), dim3(TOPK_THREADS), 0, (hipStream_t)stream, lv,
                       (unsigned long long*)nullptr);
    return cpm::check_launch("topk_rows_multi");
  }
  hipLaunchKernelGGL(topk_rows_kernel, dim3(rows, levels, max_slices), dim3(TOPK_THREADS), 0, (hipStream_t)stream, lv,
                     (unsigned long long*)workspace);
  hipLaunchKernelGGL(topk_merge_kernel, dim3(rows, levels), dim3(TOPK_THREADS), 0, (hipStream_t)stream, lv,
                     (const unsigned long long*)workspace);
  return cpm::check_launch("topk_rows_multi (sliced)");
}
