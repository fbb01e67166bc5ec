// Device-resident RoI lists of the CPM training step: everything between the proposal NMS and the
// sampled RoI sets the heads run on stays on the device, as packed lists with a fixed capacity and a count per image.
//
// The reference builds these sets with per-image tensor ops (nonzero / boolean indexing / randperm / cat), each a
// launch plus a device->host round trip:
//   proposals   pet/rcnn/modeling/rpn/inference.py:101-196  (post-NMS top-n per level, top-k over the batch, add gts)
//   cls sample  pet/rcnn/modeling/grid_cascade_rcnn/loss.py:29-97 (match, label, 512 per image at <= 25 % positives)
//   positives   pet/rcnn/utils/misc.py:54-94 (keep_only_positive_boxes, <= MAX_SAMPLE_NUM_GRID per image)
//   next stage  pet/rcnn/modeling/grid_cascade_rcnn/inference.py:281-310 + loss.py:144-176 (filter, add gts, match)
//   RSM sample  pet/rcnn/modeling/grid_cascade_rcnn/grid_cascade_rcnn.py:231-245 (cls negatives + refined positives)
// Here each is ONE launch of one workgroup (the lists hold a few thousand rows at most: latency, not bandwidth,
// is what counts), and the host reads back nothing but the per-image counts it needs to size the next launch.
//
// Built with -ffp-contract=off like detect_glue.hip: IoU thresholds must see the reference's fp32 values.
#include "common.h"

namespace {

constexpr int THREADS = 1024;
constexpr int WAVES = THREADS / 64;
constexpr int MAX_IMAGES = 64;
constexpr int MAX_SEGS = 512;              // images x levels of the proposal stage
constexpr int MAX_ROWS = 4096;             // rows of one image a sampling workgroup ranks in LDS

__device__ __forceinline__ float iou_plus1(const float4 g, const float4 p) {
  // identical to detect_glue.hip (boxlist_ops.py:123-158, the gt is "box1")
  const float area1 = (g.z - g.x + 1.f) * (g.w - g.y + 1.f);
  const float area2 = (p.z - p.x + 1.f) * (p.w - p.y + 1.f);
  const float ltx = fmaxf(g.x, p.x), lty = fmaxf(g.y, p.y);
  const float rbx = fminf(g.z, p.z), rby = fminf(g.w, p.w);
  const float w = fmaxf(rbx - ltx + 1.f, 0.f), h = fmaxf(rby - lty + 1.f, 0.f);
  const float inter = w * h;
  return inter / (area1 + area2 - inter);
}

// the sampler's counter-based hash (sampler.hip): same keys => same samples as cpm_sample_pos_neg
__device__ __forceinline__ uint32_t sample_key(uint32_t seed, uint32_t i) {
  uint32_t x = i * 0x9E3779B9u + seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  x += seed * 0x27D4EB2Fu + 0x165667B1u;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  return x;
}

// Exclusive rank of a flag over the workgroup (thread order) plus the total; two barriers.
__device__ __forceinline__ int block_rank(bool flag, int* s_wave, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(flag);
  __syncthreads();                                   // s_wave may still be read from the previous call
  if (lane == 0) s_wave[wave] = __popcll(bal);
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    const int v = s_wave[w];
    before += w < wave ? v : 0;
    tot += v;
  }
  *total = tot;
  return before + __popcll(bal & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ uint32_t ordered_bits(float v) {
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // ascending unsigned order == ascending float order
}

// From a 256-bin histogram: the highest digit d with  count(bins > d) < want <= count(bins >= d), and count(bins > d).
// Wave 0 of the workgroup: four bins per lane, suffix sums over the lanes (a walk down the bins by one thread read 256
// LDS words one after the other, four times per selection).  Results in out[0] (digit), out[1] (above); the caller
// puts barriers around the call.
__device__ __forceinline__ void find_digit_desc(const int* hist, int want, int* out) {
  if (threadIdx.x >= 64) return;
  const int l = threadIdx.x;
  const int h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
  const int mine = h0 + h1 + h2 + h3;
  int incl = mine;                                     // this lane's bins and every higher lane's
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_down(incl, d, 64);
    if (l + d < 64) incl += t;
  }
  const int above_lane = incl - mine;
  // exactly one lane holds the crossing when 1 <= want <= total; with want beyond the total the walk ended at digit 0
  if (above_lane < want && (want <= incl || l == 0)) {
    int d = 4 * l + 3, above = above_lane;             // down the lane's bins to the first with above + count >= want
    if (above + h3 < want) {
      above += h3; --d;
      if (above + h2 < want) {
        above += h2; --d;
        if (above + h1 < want) { above += h1; --d; }     // (bin 4 l is the crossing, or digit 0 at the end of the walk)
      }
    }
    out[0] = d;
    out[1] = above;
  }
}

struct SegOff { int32_t off[MAX_SEGS + 1]; };

// ---- proposals: post-NMS top-n per (image, level), top-k over the whole batch, gts appended --------------------
// Candidate order = the order the reference concatenates in: image-major, then level, then NMS rank.  torch.topk
// keeps every score above the k-th largest and, of the scores equal to it, the lowest indices; so does this.
__global__ void __launch_bounds__(THREADS) proposals_finalize_kernel(
    const float4* __restrict__ seg_boxes, const float* __restrict__ seg_scores, const int64_t* __restrict__ keep,
    const int32_t* __restrict__ keep_count, SegOff segs, int n_img, int n_lvl, int post_top_n, int batch_top_k,
    const float4* __restrict__ gts, const int32_t* __restrict__ gt_off, int capacity, float4* __restrict__ out_boxes,
    float* __restrict__ out_obj, float* __restrict__ out_rois5, int32_t* __restrict__ out_img,
    int32_t* __restrict__ out_counts) {
  __shared__ int s_cbase[MAX_SEGS + 1];      // candidate prefix in (image, level) order
  __shared__ int s_hist[256];
  __shared__ int s_wave[WAVES];
  __shared__ int s_selcnt[MAX_IMAGES];
  __shared__ int s_selpre[MAX_IMAGES + 1];
  __shared__ uint32_t s_prefix;
  __shared__ int s_remaining;
  __shared__ int s_digit[2];
  constexpr int KEYCAP = 12288;              // candidate keys kept in LDS after the first pass (48 KB)
  __shared__ uint32_t s_keys[KEYCAP];
  const int n_seg = n_img * n_lvl;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int q = 0; q < n_seg; ++q) {
      const int s = (q % n_lvl) * n_img + q / n_lvl;             // segments are level-major, candidates image-major
      int c = keep_count[s];
      const int len = segs.off[s + 1] - segs.off[s];
      c = c < 0 ? 0 : (c > len ? len : c);
      if (post_top_n > 0 && c > post_top_n) c = post_top_n;
      s_cbase[q] = run;
      run += c;
    }
    s_cbase[n_seg] = run;
    s_prefix = 0u;
  }
  if (threadIdx.x < MAX_IMAGES) s_selcnt[threadIdx.x] = 0;
  __syncthreads();
  const int T = s_cbase[n_seg];
  auto locate = [&](int t, int& q, int& row) {
    int lo = 0, hi = n_seg;                                      // last q with cbase[q] <= t
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_cbase[mid] <= t) lo = mid; else hi = mid;
    }
    q = lo;
    const int s = (q % n_lvl) * n_img + q / n_lvl;
    row = segs.off[s] + (int)keep[segs.off[s] + (t - s_cbase[q])];
  };
  const bool take_all = T <= batch_top_k;
  uint32_t kth = 0u;
  int need_eq = 0;
  if (!take_all) {
    // radix select of the batch_top_k-th largest score: four 8-bit digits, most significant first
    if (threadIdx.x == 0) s_remaining = batch_top_k;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
      __syncthreads();
      const uint32_t himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
      const uint32_t prefix = s_prefix;
      // (a candidate's key costs a binary search and two dependent global loads: the first digit's pass keeps it in LDS)
      for (int t = threadIdx.x; t < T; t += THREADS) {
        uint32_t key;
        if (shift != 24 && t < KEYCAP) {
          key = s_keys[t];
        } else {
          int q, row;
          locate(t, q, row);
          key = ordered_bits(seg_scores[row]);
          if (t < KEYCAP) s_keys[t] = key;
        }
        if (((key ^ prefix) & himask) == 0u) atomicAdd(&s_hist[(key >> shift) & 255u], 1);
      }
      __syncthreads();
      const int remaining = s_remaining;
      find_digit_desc(s_hist, remaining, s_digit);
      __syncthreads();
      if (threadIdx.x == 0) {
        s_prefix = prefix | ((uint32_t)s_digit[0] << shift);
        s_remaining = remaining - s_digit[1];                    // still to take among keys == prefix (so far)
      }
      __syncthreads();
    }
    kth = s_prefix;
    need_eq = s_remaining;
  }
  // one index-order pass: select, rank, write
  int run_sel = 0, run_eq = 0;
  for (int start = 0; start < T; start += THREADS) {
    const int t = start + threadIdx.x;
    int q = 0, row = 0;
    float score = 0.f;
    bool gt_k = false, eq_k = false;
    if (t < T) {
      locate(t, q, row);
      score = seg_scores[row];
      const uint32_t key = ordered_bits(score);
      gt_k = take_all || key > kth;
      eq_k = !take_all && key == kth;
    }
    int tot_eq, tot_sel;
    const int eq_before = run_eq + block_rank(eq_k, s_wave, &tot_eq);
    const bool sel = gt_k || (eq_k && eq_before < need_eq);
    const int sel_before = run_sel + block_rank(sel, s_wave, &tot_sel);
    if (sel) {
      const int n = q / n_lvl;
      const int pos = sel_before + gt_off[n];
      if (pos < capacity) {
        const float4 b = seg_boxes[row];
        out_boxes[pos] = b;
        out_obj[pos] = score;
        out_img[pos] = n;
        if (out_rois5) {
          float* r5 = out_rois5 + (size_t)pos * 5;
          r5[0] = (float)n; r5[1] = b.x; r5[2] = b.y; r5[3] = b.z; r5[4] = b.w;
        }
      }
      atomicAdd(&s_selcnt[n], 1);
    }
    run_sel += tot_sel;
    run_eq += tot_eq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int n = 0; n < n_img; ++n) { s_selpre[n] = run; run += s_selcnt[n]; }
    s_selpre[n_img] = run;
    for (int n = 0; n < n_img; ++n) out_counts[n] = s_selcnt[n] + gt_off[n + 1] - gt_off[n];
    out_counts[n_img] = run + gt_off[n_img];
  }
  __syncthreads();
  const int G = gt_off[n_img];
  for (int g = threadIdx.x; g < G; g += THREADS) {
    int n = 0;
    while (n + 1 < n_img && gt_off[n + 1] <= g) ++n;
    const int pos = s_selpre[n + 1] + g;
    if (pos < capacity) {
      const float4 b = gts[g];
      out_boxes[pos] = b;
      out_obj[pos] = 1.f;
      out_img[pos] = n;
      if (out_rois5) {
        float* r5 = out_rois5 + (size_t)pos * 5;
        r5[0] = (float)n; r5[1] = b.x; r5[2] = b.y; r5[3] = b.z; r5[4] = b.w;
      }
    }
  }
  const int total = s_selpre[n_img] + G;
  for (int p = total + threadIdx.x; p < capacity; p += THREADS) {
    out_boxes[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    out_obj[p] = 0.f;
    out_img[p] = -1;
  }
}

// ---- match + label + fixed-size sample + compaction, optionally the positives of the sample --------------------
struct SampleArgs {
  const float4* boxes;            // packed input list, image-contiguous
  const float* obj;
  const int32_t* counts;          // [n_img] rows per image (device)
  const float4* gts;
  const int64_t* gt_labels;
  const int32_t* gt_off;          // [n_img + 1] device
  int n_img;
  float high, low;
  int batch, max_pos;
  uint32_t seed;
  int max_grid;                   // 0: no positives list
  uint32_t seed_grid;
  float grid_high;                // first grid stage's foreground IoU: p_gt = matched gt, or the image's first gt
  int cap_sample, cap_grid;
  // sample
  float4* s_boxes; float* s_obj; int64_t* s_labels; int32_t* s_img; float* s_rois5; int32_t* s_counts;  // [n_img + 1]
  // positives of the sample, at most max_grid per image
  float4* p_boxes; float4* p_gt; float* p_iou; int64_t* p_src; int32_t* p_img; float* p_rois5;
  int32_t* p_counts;              // [n_img + 1]
  int32_t* status;                // != 0: an image held more than MAX_ROWS rows (nothing was sampled)
};

// The `want`-th smallest (1-based) of the keys s_key[r], r < cnt, whose class byte s_cls[r] equals c: radix select,
// eight 8-bit digits from the top, a 256-bin LDS histogram per digit (the lists hold a few thousand rows: two or four
// per thread).  The keys are distinct (the row index is their low word), so `key <= result` marks exactly `want` members.
// Every thread of the workgroup calls; s_hist (256 ints) and s_sel (2 ints) are scratch.  Replaces a rank-by-counting
// loop over all rows per row (4 M LDS reads per image at 2 000 proposals: most of the kernel's time).
__device__ unsigned long long kth_smallest_key(const unsigned long long* s_key, const uint8_t* s_cls, int c, int cnt,
                                               int want, int* s_hist, int* s_sel) {
  unsigned long long prefix = 0;
  for (int shift = 56; shift >= 0; shift -= 8) {
    if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long himask = shift == 56 ? 0ull : ~0ull << (shift + 8);
    for (int r = threadIdx.x; r < cnt; r += THREADS) {
      const unsigned long long k = s_key[r];
      if (s_cls[r] == c && (k & himask) == prefix) atomicAdd(&s_hist[(int)((k >> shift) & 255)], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64) {                          // wave 0: four bins per lane, prefix over the lanes
      const int l = threadIdx.x;
      const int h0 = s_hist[4 * l], h1 = s_hist[4 * l + 1], h2 = s_hist[4 * l + 2], h3 = s_hist[4 * l + 3];
      const int mine = h0 + h1 + h2 + h3;
      int incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (l >= d) incl += t;
      }
      const int below = incl - mine;                 // members in lower bins
      if (below < want && want <= incl) {            // the crossing lies in this lane's bins
        int b = 4 * l, acc = below;
        if (acc + h0 < want) { acc += h0; ++b; if (acc + h1 < want) { acc += h1; ++b; if (acc + h2 < want) { acc += h2; ++b; } } }
        s_sel[0] = b;
        s_sel[1] = want - acc;                       // rank inside the bin
      }
    }
    __syncthreads();
    prefix |= (unsigned long long)s_sel[0] << shift;
    want = s_sel[1];
    __syncthreads();                                 // s_sel is rewritten by the next digit
  }
  return prefix;
}

__global__ void __launch_bounds__(THREADS) roi_sample_kernel(SampleArgs a) {
  __shared__ unsigned long long s_key[MAX_ROWS];   // (key << 32 | row) of the row's sampling draw
  __shared__ uint8_t s_cls[MAX_ROWS];              // 0 positive, 1 negative, 2 ignored
  __shared__ int s_label[MAX_ROWS];
  __shared__ uint8_t s_take[MAX_ROWS];
  __shared__ int s_wave[WAVES];
  __shared__ int s_n[2];
  __shared__ int s_hist[256];
  __shared__ int s_sel[2];
  int in_off = 0, out_s = 0, out_p = 0;
  for (int img = 0; img < a.n_img; ++img)
    if (a.counts[img] > MAX_ROWS) {                 // uniform: every thread reads the same counts
      if (threadIdx.x == 0) *a.status = 1;
      return;
    }
  if (threadIdx.x == 0) *a.status = 0;
  if (a.max_grid <= 0 && threadIdx.x <= a.n_img) a.p_counts[threadIdx.x] = 0;
  for (int img = 0; img < a.n_img; ++img) {
    const int cnt = a.counts[img];
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
    __syncthreads();
    const int g0 = a.gt_off[img], g1 = a.gt_off[img + 1];
    const uint32_t iseed = a.seed + (uint32_t)img * 0x632BE5ABu;
    // match (Matcher without low-quality matches, matcher.py:48-89) and label (loss.py:32-40)
    for (int r = threadIdx.x; r < cnt; r += THREADS) {
      const float4 p = a.boxes[in_off + r];
      float best = -1.f;
      int arg = 0;
      for (int g = g0; g < g1; ++g) {
        const float v = iou_plus1(a.gts[g], p);
        if (v > best) { best = v; arg = g - g0; }
      }
      int label;
      if (best < a.low) label = 0;
      else if (best < a.high) label = -1;
      else label = (int)a.gt_labels[g0 + arg];
      const int c = label >= 1 ? 0 : (label == 0 ? 1 : 2);
      s_label[r] = label;
      s_cls[r] = (uint8_t)c;
      s_key[r] = ((unsigned long long)sample_key(iseed, (uint32_t)r) << 32) | (uint32_t)r;
      if (c < 2) atomicAdd(&s_n[c], 1);
    }
    __syncthreads();
    const int n_pos = s_n[0], n_neg = s_n[1];
    const int q_pos = n_pos < a.max_pos ? n_pos : a.max_pos;
    const int room = a.batch - q_pos;
    const int q_neg = n_neg < room ? n_neg : room;
    // the sample of a class = its `quota` members with the smallest (key, row): BalancedPositiveNegativeSampler's
    // uniformly random subset (balanced_positive_negative_sampler.py:27-67), drawn as cpm_sample_pos_neg draws it
    // (a class with more members than places: the threshold key of its quota; uniform branches)
    unsigned long long thr_pos = ~0ull, thr_neg = ~0ull;
    if (q_pos > 0 && q_pos < n_pos) thr_pos = kth_smallest_key(s_key, s_cls, 0, cnt, q_pos, s_hist, s_sel);
    if (q_neg > 0 && q_neg < n_neg) thr_neg = kth_smallest_key(s_key, s_cls, 1, cnt, q_neg, s_hist, s_sel);
    for (int r = threadIdx.x; r < cnt; r += THREADS) {
      const int c = s_cls[r];
      uint8_t take = 0;
      if (c < 2) {
        const int quota = c == 0 ? q_pos : q_neg;
        take = quota > 0 && s_key[r] <= (c == 0 ? thr_pos : thr_neg);
      }
      s_take[r] = take;
    }
    __syncthreads();
    // positives of the sample: all, or the max_grid with the smallest second draw (misc.py:54-94, a random subset)
    const uint32_t gseed = a.seed_grid + (uint32_t)img * 0x632BE5ABu;
    const bool subset = a.max_grid > 0 && q_pos > a.max_grid;
    unsigned long long thr_grid = ~0ull;
    if (subset) {
      // second draw; the class byte of a sampled positive becomes 3 for the selection (nothing below reads the 0 again
      // except through `s_cls[r] == 0 || == 3`)
      for (int r = threadIdx.x; r < cnt; r += THREADS) {
        s_key[r] = ((unsigned long long)sample_key(gseed, (uint32_t)r) << 32) | (uint32_t)r;
        if (s_take[r] && s_cls[r] == 0) s_cls[r] = 3;
      }
      __syncthreads();
      thr_grid = kth_smallest_key(s_key, s_cls, 3, cnt, a.max_grid, s_hist, s_sel);
    }
    int run_s = 0, run_p = 0;
    for (int start = 0; start < cnt; start += THREADS) {
      const int r = start + threadIdx.x;
      const bool take = r < cnt && s_take[r];
      bool posi = a.max_grid > 0 && take && (s_cls[r] == 0 || s_cls[r] == 3);
      if (posi && subset) posi = s_key[r] <= thr_grid;
      int tot_s, tot_p;
      const int before_s = run_s + block_rank(take, s_wave, &tot_s);
      const int before_p = run_p + block_rank(posi, s_wave, &tot_p);
      if (take) {
        const int pos = out_s + before_s;
        if (pos < a.cap_sample) {
          const float4 b = a.boxes[in_off + r];
          a.s_boxes[pos] = b;
          a.s_obj[pos] = a.obj[in_off + r];
          a.s_labels[pos] = s_label[r];
          a.s_img[pos] = img;
          float* r5 = a.s_rois5 + (size_t)pos * 5;
          r5[0] = (float)img; r5[1] = b.x; r5[2] = b.y; r5[3] = b.z; r5[4] = b.w;
          if (posi) {
            const int pp = out_p + before_p;
            if (pp < a.cap_grid) {
              // the first grid stage's match (loss.py:144-162: gt = t.bbox[matched.clamp(min=0)], all RoIs kept)
              float best = -1.f;
              int arg = 0;
              for (int g = g0; g < g1; ++g) {
                const float v = iou_plus1(a.gts[g], b);
                if (v > best) { best = v; arg = g - g0; }
              }
              a.p_boxes[pp] = b;
              a.p_gt[pp] = a.gts[g0 + (best >= a.grid_high ? arg : 0)];
              a.p_iou[pp] = best;
              a.p_src[pp] = pos;
              a.p_img[pp] = img;
              float* q5 = a.p_rois5 + (size_t)pp * 5;
              q5[0] = (float)img; q5[1] = b.x; q5[2] = b.y; q5[3] = b.z; q5[4] = b.w;
            }
          }
        }
      }
      run_s += tot_s;
      run_p += tot_p;
    }
    if (threadIdx.x == 0) {
      a.s_counts[img] = run_s;
      if (a.max_grid > 0) a.p_counts[img] = run_p;
    }
    in_off += cnt;
    out_s += run_s;
    out_p += run_p;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.s_counts[a.n_img] = out_s;
    if (a.max_grid > 0) a.p_counts[a.n_img] = out_p;
  }
  // padding rows: a head may run on the whole capacity without knowing the count -- image -1 (RoIAlign pools zeros
  // and scatters nothing back for it) and cross_entropy's ignore_index as the label keep such rows out of the loss
  // and of every gradient
  for (int p = out_s + threadIdx.x; p < a.cap_sample; p += THREADS) {
    a.s_img[p] = -1;
    a.s_labels[p] = -100;
    a.s_boxes[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    a.s_obj[p] = 0.f;
    float* r5 = a.s_rois5 + (size_t)p * 5;
    r5[0] = -1.f; r5[1] = 0.f; r5[2] = 0.f; r5[3] = 0.f; r5[4] = 0.f;
  }
  if (a.max_grid > 0)
    for (int p = out_p + threadIdx.x; p < a.cap_grid; p += THREADS) a.p_img[p] = -1;
}

// ---- next cascade stage: decoded boxes that survived the filter and still match a gt, then the image's gts -----
struct AdvanceArgs {
  const float4* refined; const uint8_t* keep; const int64_t* matched; const float* iou; const int32_t* img;
  const int64_t* src;
  int R, n_img;
  int64_t gt_src_base;            // ride-along row of gt g = gt_src_base + g
  const float4* gts; const int32_t* gt_off;
  int capacity;
  float4* o_rois; float4* o_gt; float* o_iou; int64_t* o_src; int32_t* o_img; float* o_rois5; int32_t* o_counts;
};

__global__ void __launch_bounds__(THREADS) stage_advance_kernel(AdvanceArgs a) {
  __shared__ int s_wave[WAVES];
  __shared__ int s_cnt[MAX_IMAGES];
  __shared__ int s_pre[MAX_IMAGES + 1];
  if (threadIdx.x < MAX_IMAGES) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  int run = 0;
  for (int start = 0; start < a.R; start += THREADS) {
    const int i = start + threadIdx.x;
    const bool ok = i < a.R && a.keep[i] && a.matched[i] >= 0;
    int tot;
    const int before = run + block_rank(ok, s_wave, &tot);
    if (ok) {
      const int n = a.img[i];
      const int pos = before + a.gt_off[n];
      if (pos < a.capacity) {
        const float4 b = a.refined[i];
        a.o_rois[pos] = b;
        a.o_gt[pos] = a.gts[a.gt_off[n] + (int)a.matched[i]];
        a.o_iou[pos] = a.iou[i];
        a.o_src[pos] = a.src ? a.src[i] : (int64_t)i;
        a.o_img[pos] = n;
        float* r5 = a.o_rois5 + (size_t)pos * 5;
        r5[0] = (float)n; r5[1] = b.x; r5[2] = b.y; r5[3] = b.z; r5[4] = b.w;
      }
      atomicAdd(&s_cnt[n], 1);
    }
    run += tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int r = 0;
    for (int n = 0; n < a.n_img; ++n) { s_pre[n] = r; r += s_cnt[n]; }
    s_pre[a.n_img] = r;
    for (int n = 0; n < a.n_img; ++n) a.o_counts[n] = s_cnt[n] + a.gt_off[n + 1] - a.gt_off[n];
    a.o_counts[a.n_img] = r + a.gt_off[a.n_img];
  }
  __syncthreads();
  const int G = a.gt_off[a.n_img];
  for (int g = threadIdx.x; g < G; g += THREADS) {
    int n = 0;
    while (n + 1 < a.n_img && a.gt_off[n + 1] <= g) ++n;
    const int pos = s_pre[n + 1] + g;
    if (pos < a.capacity) {
      const float4 b = a.gts[g];
      a.o_rois[pos] = b;
      a.o_gt[pos] = b;                        // a gt matches itself with IoU exactly 1 (inter == area)
      a.o_iou[pos] = 1.f;
      a.o_src[pos] = a.gt_src_base + g;
      a.o_img[pos] = n;
      float* r5 = a.o_rois5 + (size_t)pos * 5;
      r5[0] = (float)n; r5[1] = b.x; r5[2] = b.y; r5[3] = b.z; r5[4] = b.w;
    }
  }
  for (int p = s_pre[a.n_img] + G + threadIdx.x; p < a.capacity; p += THREADS) a.o_img[p] = -1;
}

// ---- RSM candidates: the cls sample's negatives followed by the refined positives, per image -------------------
struct RescoreArgs {
  const float4* s_boxes; const float* s_obj; const int64_t* s_labels; const int32_t* s_counts;   // cls sample
  const float4* g_boxes; const int64_t* g_src; const int32_t* g_counts;                          // last stage
  const int64_t* p_src; int n_first;   // g_src < n_first: row p_src[g_src] of the cls sample; else an appended gt
  int n_img, capacity;
  float4* o_boxes; float* o_obj; int32_t* o_counts;
};

__global__ void __launch_bounds__(THREADS) rescore_gather_kernel(RescoreArgs a) {
  __shared__ int s_wave[WAVES];
  int in_s = 0, in_g = 0, out = 0;
  for (int img = 0; img < a.n_img; ++img) {
    const int ns = a.s_counts[img], ng = a.g_counts[img];
    int run = 0;
    for (int start = 0; start < ns; start += THREADS) {
      const int r = start + threadIdx.x;
      const bool neg = r < ns && a.s_labels[in_s + r] <= 0;
      int tot;
      const int before = run + block_rank(neg, s_wave, &tot);
      if (neg && out + before < a.capacity) {
        a.o_boxes[out + before] = a.s_boxes[in_s + r];
        a.o_obj[out + before] = a.s_obj[in_s + r];
      }
      run += tot;
    }
    for (int r = threadIdx.x; r < ng; r += THREADS)
      if (out + run + r < a.capacity) {
        a.o_boxes[out + run + r] = a.g_boxes[in_g + r];
        const int64_t src = a.g_src[in_g + r];
        a.o_obj[out + run + r] = src < a.n_first ? a.s_obj[a.p_src[src]] : 1.f;
      }
    if (threadIdx.x == 0) a.o_counts[img] = run + ng;
    in_s += ns;
    in_g += ng;
    out += run + ng;
  }
  if (threadIdx.x == 0) a.o_counts[a.n_img] = out;
}

}  // namespace

CPM_EXPORT int cpm_proposals_finalize(const float* seg_boxes, const float* seg_scores, const int64_t* keep,
                                      const int32_t* keep_count, const int32_t* h_seg_off, int n_images, int n_levels,
                                      int post_nms_top_n, int batch_top_k, const float* gts, const int32_t* gt_off,
                                      int capacity, float* out_boxes, float* out_obj, float* out_rois5,
                                      int32_t* out_img, int32_t* out_counts, void* stream) {
  CPM_REQUIRE(seg_boxes && seg_scores && keep && keep_count && h_seg_off && gts && gt_off, "null input");
  CPM_REQUIRE(out_boxes && out_obj && out_img && out_counts, "null output");
  CPM_REQUIRE(n_images >= 1 && n_images <= MAX_IMAGES && n_levels >= 1 && n_images * n_levels <= MAX_SEGS,
              "1 <= images <= 64, images x levels <= 512");
  CPM_REQUIRE(batch_top_k >= 1 && capacity >= 1, "batch_top_k, capacity >= 1");
  SegOff segs = {};
  for (int s = 0; s <= n_images * n_levels; ++s) {
    CPM_REQUIRE(h_seg_off[s] >= 0 && (s == 0 || h_seg_off[s] >= h_seg_off[s - 1]), "segment offsets must ascend");
    segs.off[s] = h_seg_off[s];
  }
  hipLaunchKernelGGL(proposals_finalize_kernel, dim3(1), dim3(THREADS), 0, (hipStream_t)stream,
                     (const float4*)seg_boxes, seg_scores, keep, keep_count, segs, n_images, n_levels, post_nms_top_n,
                     batch_top_k, (const float4*)gts, gt_off, capacity, (float4*)out_boxes, out_obj, out_rois5,
                     out_img, out_counts);
  return cpm::check_launch("proposals_finalize");
}

CPM_EXPORT int cpm_roi_sample_max_rows(void) { return MAX_ROWS; }

CPM_EXPORT int cpm_roi_sample(const float* boxes, const float* obj, const int32_t* counts, int n_images,
                              const float* gts, const int64_t* gt_labels, const int32_t* gt_off, float high, float low,
                              int batch_size_per_image, int max_pos, uint64_t seed, int max_grid, uint64_t seed_grid,
                              float grid_high, int cap_sample, float* s_boxes, float* s_obj, int64_t* s_labels,
                              int32_t* s_img, float* s_rois5, int32_t* s_counts, int cap_grid, float* p_boxes,
                              float* p_gt, float* p_iou, int64_t* p_src, int32_t* p_img, float* p_rois5,
                              int32_t* p_counts, int32_t* status, void* stream) {
  CPM_REQUIRE(boxes && obj && counts && gts && gt_labels && gt_off, "null input");
  CPM_REQUIRE(s_boxes && s_obj && s_labels && s_img && s_rois5 && s_counts && status, "null output");
  CPM_REQUIRE(n_images >= 1 && n_images <= MAX_IMAGES, "1 <= images <= 64");
  CPM_REQUIRE(batch_size_per_image >= 0 && max_pos >= 0 && max_pos <= batch_size_per_image, "bad sample sizes");
  CPM_REQUIRE(cap_sample >= n_images * batch_size_per_image, "sample capacity below images x batch size");
  CPM_REQUIRE(max_grid >= 0, "max_grid >= 0");
  CPM_REQUIRE(p_counts, "null positives counts");
  if (max_grid > 0) {
    CPM_REQUIRE(p_boxes && p_gt && p_iou && p_src && p_img && p_rois5, "null positives output");
    CPM_REQUIRE(cap_grid >= n_images * (max_grid < max_pos ? max_grid : max_pos), "positives capacity too small");
  }
  SampleArgs a;
  a.boxes = (const float4*)boxes; a.obj = obj; a.counts = counts; a.gts = (const float4*)gts;
  a.gt_labels = gt_labels; a.gt_off = gt_off; a.n_img = n_images; a.high = high; a.low = low;
  a.batch = batch_size_per_image; a.max_pos = max_pos; a.seed = (uint32_t)(seed ^ (seed >> 32));
  a.max_grid = max_grid; a.seed_grid = (uint32_t)(seed_grid ^ (seed_grid >> 32)); a.grid_high = grid_high;
  a.cap_sample = cap_sample; a.cap_grid = cap_grid;
  a.s_boxes = (float4*)s_boxes; a.s_obj = s_obj; a.s_labels = s_labels; a.s_img = s_img; a.s_rois5 = s_rois5;
  a.s_counts = s_counts;
  a.p_boxes = (float4*)p_boxes; a.p_gt = (float4*)p_gt; a.p_iou = p_iou; a.p_src = p_src; a.p_img = p_img; a.p_rois5 = p_rois5; a.p_counts = p_counts;
  a.status = status;
  hipLaunchKernelGGL(roi_sample_kernel, dim3(1), dim3(THREADS), 0, (hipStream_t)stream, a);
  return cpm::check_launch("roi_sample");
}

CPM_EXPORT int cpm_stage_advance(const float* refined, const uint8_t* keep, const int64_t* matched, const float* iou,
                                 const int32_t* img, const int64_t* src, int R, int n_images, int64_t gt_src_base,
                                 const float* gts, const int32_t* gt_off, int capacity, float* o_rois, float* o_gt,
                                 float* o_iou, int64_t* o_src, int32_t* o_img, float* o_rois5, int32_t* o_counts,
                                 void* stream) {
  CPM_REQUIRE(R >= 0 && n_images >= 1 && n_images <= MAX_IMAGES && capacity >= 1, "bad sizes");
  CPM_REQUIRE(R == 0 || (refined && keep && matched && iou && img), "null input");
  CPM_REQUIRE(gts && gt_off && o_rois && o_gt && o_iou && o_src && o_img && o_rois5 && o_counts, "null pointer");
  AdvanceArgs a;
  a.refined = (const float4*)refined; a.keep = keep; a.matched = matched; a.iou = iou; a.img = img; a.src = src;
  a.R = R; a.n_img = n_images; a.gt_src_base = gt_src_base; a.gts = (const float4*)gts; a.gt_off = gt_off;
  a.capacity = capacity; a.o_rois = (float4*)o_rois; a.o_gt = (float4*)o_gt; a.o_iou = o_iou; a.o_src = o_src;
  a.o_img = o_img; a.o_rois5 = o_rois5; a.o_counts = o_counts;
  hipLaunchKernelGGL(stage_advance_kernel, dim3(1), dim3(THREADS), 0, (hipStream_t)stream, a);
  return cpm::check_launch("stage_advance");
}

CPM_EXPORT int cpm_rescore_gather(const float* s_boxes, const float* s_obj, const int64_t* s_labels,
                                  const int32_t* s_counts, const float* g_boxes, const int64_t* g_src,
                                  const int32_t* g_counts, const int64_t* p_src, int n_first, int n_images,
                                  int capacity, float* o_boxes, float* o_obj, int32_t* o_counts, void* stream) {
  CPM_REQUIRE(s_boxes && s_obj && s_labels && s_counts && g_boxes && g_src && g_counts && p_src, "null input");
  CPM_REQUIRE(o_boxes && o_obj && o_counts && n_images >= 1 && n_images <= MAX_IMAGES && capacity >= 1, "bad output");
  RescoreArgs a;
  a.s_boxes = (const float4*)s_boxes; a.s_obj = s_obj; a.s_labels = s_labels; a.s_counts = s_counts;
  a.g_boxes = (const float4*)g_boxes; a.g_src = g_src; a.g_counts = g_counts; a.p_src = p_src;
  a.n_first = n_first; a.n_img = n_images;
  a.capacity = capacity; a.o_boxes = (float4*)o_boxes; a.o_obj = o_obj; a.o_counts = o_counts;
  hipLaunchKernelGGL(rescore_gather_kernel, dim3(1), dim3(THREADS), 0, (hipStream_t)stream, a);
  return cpm::check_launch("rescore_gather");
}
