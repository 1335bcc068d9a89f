// Photon map on the GPU: the reference's left-balanced kd-tree (kdtree.js) built level by level,
// the fixed-radius query (nearestNeighbours, kdtree.js:64-80), a k-nearest query with an
// LDS-resident top-k heap per lane, and the fused radiance gathers built on them.
//
// Build.  makeTree (kdtree.js:82-107) recursively (1) takes the bounding box of its subset,
// (2) picks the widest axis (ties -> lowest axis), (3) stably sorts the subset by that axis,
// (4) makes element findMedian(len) the node and recurses on the two sides.  The size of every
// subtree is a function of N alone (findMedian is), so all nodes of one level can be processed
// at once: the N photons sit in one array in which every subtree owns a contiguous segment and
// every finished node stays where its parent's sort left it.  One level =
//   bbox reduce per segment -> key = (segment start, sortable(coord[axis])) -> one stable radix
//   sort of the whole array -> the element at start + findMedian(len) of each segment becomes
//   nodes[pos]; the two sides become the segments of 2*pos and 2*pos+1.
// Finished nodes get key (own index, 0) and therefore never move.  The sort is stable, like
// the reference's Array.prototype.sort, so ties keep the order the parent's sort left them in.
//
// All accept/prune comparisons of the queries are evaluated in fp64 without FMA contraction on
// the exactly converted fp32 coordinates, so the returned index sets are exactly the
// reference's for the same (fp32-representable) photon positions.
#include <mutex>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "rt_pipeline.h"
#include "rt_internal.h"

// (compiled with -ffp-contract=off, see Makefile)

// ------------------------------------------------------------------ findMedian
__host__ __device__ static inline uint32_t find_median(uint32_t n) {     // kdtree.js:109-124
  if (n <= 1) return 0;
  uint32_t power_two = 1, two_bits = n;
  for (;;) {
    if (two_bits == 2) return n - power_two;
    if (two_bits == 3) return 2 * power_two - 1;
    power_two <<= 1; two_bits >>= 1;
  }
}
extern "C" uint32_t rt_find_median(uint32_t n) { return find_median(n); }

// stable radix sort of (64-bit key, 32-bit value) pairs on the context's stream; temp grows as needed
// (Tried: the library's radix sort configured for 6- and 8-bit digits instead of its own choice for this GPU -- the kd build
// unchanged at 3.3 ms, the frame's request sort slower.)
int rt_sort_pairs_u64(rt_ctx *ctx, DevMem &temp, const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n) {
  if (n == 0) return RT_OK;
  size_t bytes = 0;
  RT_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, ctx->stream));
  RT_HIP(ctx, temp.reserve(bytes ? bytes : 16));
  RT_HIP(ctx, rocprim::radix_sort_pairs(temp.p, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, ctx->stream));
  return RT_OK;
}

// The level sorts of the kd build.  The library sorts up to 2^20 items with a merge sort (block sort + ten merge passes at
// a million photons) unless told otherwise; its radix sort needs one pass per 8-bit digit of the 32 .. 48 key bits of a
// level, and every pass is a chain of workgroups that each wait for the one before (decoupled look-back): the passes of
// a million items are bound by that chain, not by bandwidth.  Measured, whole build of 10^6 photons, 512 threads x items
// per thread: 6: 2.50 ms, 8: 2.09, 9: 2.16, 10: 2.20, 12: 2.20, 16: 2.31, 24: 2.75; 256 x 12: 2.29, 256 x 24: 2.68,
// 1024 x 5: 2.39, 1024 x 12: 2.41; the library's own choice 2.26.  (8 items make about one workgroup per compute unit
// at a million photons; 12 and 16 items where those do -- 1.4 and 2 million photons -- gain 1 - 3 % there: left out.)
// (Tried: 6- and 7-bit digits: 3.3, 3.6 ms.)
#ifndef RT_KD_MERGE_LIMIT
#define RT_KD_MERGE_LIMIT 65536
#endif
using RtKdSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 32>, rocprim::kernel_config<512, 8>, 8, rocprim::block_radix_rank_algorithm::match>, RT_KD_MERGE_LIMIT>;
// temp == nullptr: only the size of the temporary storage is returned in `bytes`
static hipError_t kd_level_sort(void *temp, size_t &bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, int bits, hipStream_t stream) {
  return rocprim::radix_sort_pairs<RtKdSortConfig>(temp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, bits, stream);
}
// ------------------------------------------------------------------ build kernels
__device__ __forceinline__ uint32_t sortable(float x) {
  if (x == 0.0f) x = 0.0f;                         // -0 and +0 compare equal in the reference's comparator
  uint32_t u = __float_as_uint(x);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unsortable(uint32_t u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// per array position: heap position of its segment, or KD_DONE | heap position of the node it has become
// (start and length of a segment follow from the array's size and the heap position alone: the host tabulates them)
struct SegState { uint32_t *pos; const uint2 *tab; };
#define KD_DONE 0x80000000u
__global__ void __launch_bounds__(256) k_kd_iota(uint32_t n, uint32_t *idx) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = i;
}

// idx = identity, one segment (the root's) over the whole array, and every box of the global levels empty
__global__ void __launch_bounds__(256) k_kd_init(uint32_t n, uint32_t *idx, SegState sg, uint32_t n_box_words, uint32_t *bbmin, uint32_t *bbmax) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_box_words) { bbmin[i] = 0xFFFFFFFFu; bbmax[i] = 0u; }
  if (i >= n) return;
  idx[i] = i; sg.pos[i] = 1;
}
__global__ void __launch_bounds__(256) k_kd_pos4(uint32_t n, const float *pos3, float4 *pos4) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos4[i] = make_float4(pos3[(size_t)i * 3], pos3[(size_t)i * 3 + 1], pos3[(size_t)i * 3 + 2], 0.f);
}

// min/max of every active segment (kdtree.js:85-92) over the array positions a workgroup holds in registers:
// KD_BBOX_ITEMS rows of 256 consecutive positions, row r's segment (heap position; 0 = none) in sp[r] and photon in
// p[r].  Near the root the positions all lie in one segment: the workgroup reduces in registers, across lanes and through
// LDS and issues one atomic per axis and bound (thousands of same-address atomics would serialise).  Further down,
// waves that lie inside one segment reduce across lanes first; mixed waves fall back to per-lane atomics, which by
// then go to many different addresses.  All threads of the workgroup call this.
#ifndef KD_BBOX_ITEMS
#define KD_BBOX_ITEMS 4
#endif
__device__ __forceinline__ void kd_boxes_of_chunk(bool one_segment, uint32_t sp_first, const uint32_t (&sp)[KD_BBOX_ITEMS], const float4 (&p)[KD_BBOX_ITEMS],
                                                  uint32_t (*red)[6], uint32_t *bbmin, uint32_t *bbmax) {
  if (one_segment) {
    uint32_t lo[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, hi[3] = { 0, 0, 0 };
#pragma unroll
    for (int r = 0; r < KD_BBOX_ITEMS; r++) {
      if (!sp[r]) continue;
      const float c[3] = { p[r].x, p[r].y, p[r].z };
      for (int d = 0; d < 3; d++) { const uint32_t u = sortable(c[d]); lo[d] = min(lo[d], u); hi[d] = max(hi[d], u); }
    }
    for (int off = 32; off > 0; off >>= 1)
      for (int d = 0; d < 3; d++) {
        uint32_t a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
        lo[d] = min(lo[d], a); hi[d] = max(hi[d], b);
      }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int d = 0; d < 3; d++) { red[wv][d] = lo[d]; red[wv][3 + d] = hi[d]; }
    __syncthreads();
    if (threadIdx.x < 3) {
      const int d = threadIdx.x;
      atomicMin(&bbmin[(size_t)sp_first * 3 + d], min(min(red[0][d], red[1][d]), min(red[2][d], red[3][d])));
      atomicMax(&bbmax[(size_t)sp_first * 3 + d], max(max(red[0][3 + d], red[1][3 + d]), max(red[2][3 + d], red[3][3 + d])));
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < KD_BBOX_ITEMS; r++) {                    // (workgroup-uniform trip count: the shuffles below need whole waves)
    const uint32_t s = sp[r];
    uint32_t lo[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, hi[3] = { 0, 0, 0 };
    if (s) {
      const float c[3] = { p[r].x, p[r].y, p[r].z };
      for (int d = 0; d < 3; d++) { const uint32_t u = sortable(c[d]); lo[d] = u; hi[d] = u; }
    }
    const uint32_t s0 = __shfl(s, 0, 64);
    const bool uniform = __all(s == s0 || s == 0) && s0 != 0;
    if (uniform) {
      for (int off = 32; off > 0; off >>= 1)
        for (int d = 0; d < 3; d++) {
          uint32_t a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
          lo[d] = min(lo[d], a); hi[d] = max(hi[d], b);
        }
      if ((threadIdx.x & 63) == 0)
        for (int d = 0; d < 3; d++) { atomicMin(&bbmin[(size_t)s0 * 3 + d], lo[d]); atomicMax(&bbmax[(size_t)s0 * 3 + d], hi[d]); }
    } else if (s) {
      for (int d = 0; d < 3; d++) { atomicMin(&bbmin[(size_t)s * 3 + d], lo[d]); atomicMax(&bbmax[(size_t)s * 3 + d], hi[d]); }
    }
  }
}

// the root's box: every position is in segment 1 and the photons are in their input order
__global__ void __launch_bounds__(256) k_kd_bbox_root(uint32_t n, const float4 *pos4, uint32_t *bbmin, uint32_t *bbmax) {
  __shared__ uint32_t red[4][6];
  const uint32_t c0 = blockIdx.x * (256u * KD_BBOX_ITEMS);
  if (c0 >= n) return;
  uint32_t sp[KD_BBOX_ITEMS]; float4 p[KD_BBOX_ITEMS];
#pragma unroll
  for (int r = 0; r < KD_BBOX_ITEMS; r++) {
    const uint32_t i = c0 + r * 256u + threadIdx.x;
    sp[r] = i < n ? 1u : 0u;
    p[r] = i < n ? pos4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  kd_boxes_of_chunk(true, 1u, sp, p, red, bbmin, bbmax);
}

// widest axis in fp64, `>` strict so ties go to the lowest axis (kdtree.js:93-100)
__device__ __forceinline__ int widest_dim3(const uint32_t *lo, const uint32_t *hi) {
  double w[3];
  for (int d = 0; d < 3; d++) w[d] = (double)unsortable(hi[d]) - (double)unsortable(lo[d]);
  int best = 0; double mw = w[0];
  for (int d = 1; d < 3; d++) if (w[d] > mw) { best = d; mw = w[d]; }
  return best;
}

// Sort keys of one level (heap positions [first, 2 first), first = 2^level).  The array holds the level's segments in
// heap order with exactly one finished node between neighbours, so the high part of a key only has to number the
// segments: `level` bits instead of a full array position, which saves the radix sort two of its seven passes.  Low part:
// the coordinate along the segment's widest axis.  A finished node takes the number of the segment in front of it and the
// largest low part: the sort is stable and the node already stands behind that segment, so there it stays.
__global__ void __launch_bounds__(256) k_kd_keys(uint32_t n, const float4 *pos_cur, SegState sg, uint32_t first, int level,
                                                 const uint32_t *bbmin, const uint32_t *bbmax, uint8_t *segdim, uint64_t *keys) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t sp = sg.pos[i];
  if (sp & KD_DONE) {
    const uint32_t q = sp & ~KD_DONE;
    const int dq = 31 - __clz((int)q);                      // depth of node q; its right subtree starts at segment k of this level
    const uint32_t k = ((2u * q + 1u) << (level - dq - 1)) - first;
    keys[i] = ((uint64_t)(k - 1u) << 32) | 0xFFFFFFFFull;
    return;
  }
  const int d = widest_dim3(bbmin + (size_t)sp * 3, bbmax + (size_t)sp * 3);
  if (i == sg.tab[sp].x) segdim[sp] = (uint8_t)d;
  const float4 p = pos_cur[i];
  keys[i] = ((uint64_t)(sp - first) << 32) | sortable(d == 0 ? p.x : (d == 1 ? p.y : p.z));
}

// After a level's sort: the medians become nodes, the sides become the next level's segments (kdtree.js:104-118), the
// photons' positions are brought into the new order (pos_cur[i] = pos4[idx[i]]: the one scattered read of a level; the key
// kernel and the workgroups of the lower levels then read consecutive addresses), and -- boxes != 0 -- the new segments'
// boxes are taken from the positions while they are in registers.
__global__ void __launch_bounds__(256) k_kd_split(uint32_t n, const float4 *pos4, const uint32_t *idx, SegState sg, const uint8_t *segdim,
                                                  int32_t *perm, uint8_t *dim, float4 *pos_cur, int boxes, uint32_t *bbmin, uint32_t *bbmax) {
  __shared__ uint32_t red[4][6];
  __shared__ uint32_t head[3];
  const uint32_t c0 = blockIdx.x * (256u * KD_BBOX_ITEMS);
  if (c0 >= n) return;
  const uint32_t c1 = min(n, c0 + 256u * KD_BBOX_ITEMS);
  uint32_t sp_new[KD_BBOX_ITEMS]; float4 p[KD_BBOX_ITEMS];
#pragma unroll
  for (int r = 0; r < KD_BBOX_ITEMS; r++) {
    const uint32_t i = c0 + r * 256u + threadIdx.x;
    sp_new[r] = 0; p[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= n) continue;
    const uint32_t id = idx[i];
    p[r] = pos4[id];
    pos_cur[i] = p[r];
    const uint32_t sp = sg.pos[i];
    uint32_t s = 0, len = 0;
    if (!(sp & KD_DONE)) {
      const uint2 sl = sg.tab[sp];
      s = sl.x; len = sl.y;
      const uint32_t m = find_median(len);
      if (i == s + m) { perm[sp - 1] = (int32_t)id; dim[sp - 1] = segdim[sp]; sg.pos[i] = sp | KD_DONE; }
      else if (i < s + m) { sp_new[r] = 2 * sp; len = m; sg.pos[i] = sp_new[r]; }
      else { sp_new[r] = 2 * sp + 1; s = s + m + 1; len = len - m - 1; sg.pos[i] = sp_new[r]; }
    }
    if (r == 0 && threadIdx.x == 0) { head[0] = sp_new[0]; head[1] = s; head[2] = len; }
  }
  if (!boxes) return;                                        // (kernel argument)
  __syncthreads();
  // segments are contiguous runs of active positions, so the chunk lies in one segment iff the segment of its first
  // position reaches its end
  const uint32_t sp_first = head[0];
  const bool one_segment = sp_first != 0 && head[1] + head[2] >= c1;
  kd_boxes_of_chunk(one_segment, sp_first, sp_new, p, red, bbmin, bbmax);
}

// ---- the lower levels: one workgroup finishes a whole subtree of up to KD_LDS_S photons in LDS
// The same steps as above -- box, widest axis, stable sort along it, median -- level by level on the workgroup's own
// copy of the coordinates.  The "stable sort" is a sort of unique keys (coordinate, current position): a full bitonic
// network over the array while segments are long (group number in the key's top bits keeps every segment in place),
// a rank count inside each segment once every segment has at most 64 photons.  Per position the segment it lies in
// (heap position, start, length) is kept as in the global levels; the lengths of all segments of a level follow from the
// subtree's size alone, and so does the longest one: findMedian of the previous level's longest.
#define KD_LDS_S 2048
#define KD_LDS_T 1024
#define KD_LDS_BITS 11                                       // log2(KD_LDS_S): a sort key keeps its position in that many bits
struct KdSeg { uint32_t pos, start, len; };

// Bitonic sort of S unique 64-bit keys in LDS by T threads, ascending.  A compare-exchange step pairs the elements i and
// i | j.  For the steps with j < 64 E (E = S / T) a thread holds E consecutive elements in registers: pairs inside a
// thread need nothing, pairs inside a wave are exchanged by lane shuffles -- no LDS traffic and no barrier; only the steps
// across waves (10 of the 66 at S = 2048) go through LDS, one pair per thread and barrier.  Called by all T threads with
// the keys written and a barrier passed; returns with the sorted keys in LDS behind a barrier.
// lane-shift move (DPP): every lane of the rows in ROWS takes `v` of the lane CTRL names, or `old` where there is none
template <int CTRL, int ROWS, int BANKS = 0xF> __device__ __forceinline__ uint32_t kd_lane_move(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWS, BANKS, false);
}
// v of the lane whose number differs in the bit `dist` (wave-uniform, a power of two below 64): lane-shift moves where the
// partner is in the lane's row of 16 (quad permutes for 1 and 2, a shift either way for 4, a rotation for 8), else through
// the LDS crossbar
__device__ __forceinline__ uint32_t kd_lane_xor(uint32_t v, uint32_t dist) {
  switch (dist) {
    case 1: return kd_lane_move<0xB1, 0xF>(v, v);                                  // quad_perm [1,0,3,2]
    case 2: return kd_lane_move<0x4E, 0xF>(v, v);                                  // quad_perm [2,3,0,1]
    case 4: { const uint32_t t = kd_lane_move<0x104, 0xF, 0x5>(v, v); return kd_lane_move<0x114, 0xF, 0xA>(t, v); }   // row_shl:4 for lanes 0-3, 8-11; row_shr:4 for the others
    case 8: return kd_lane_move<0x128, 0xF>(v, v);                                 // row_ror:8
    default: return (uint32_t)__shfl_xor((int)v, (int)dist, 64);
  }
}
template <uint32_t S, uint32_t T> __device__ __forceinline__ void kd_lds_sort(unsigned long long *key, uint32_t tid) {
  constexpr uint32_t E = S / T, W = 64u * E;                 // W: elements of one wave
  unsigned long long e[E];
  auto load = [&]() {
#pragma unroll
    for (uint32_t q = 0; q < E; q++) e[q] = key[tid * E + q];
  };
  auto store = [&]() {
#pragma unroll
    for (uint32_t q = 0; q < E; q++) key[tid * E + q] = e[q];
    __syncthreads();
  };
  auto in_wave = [&](uint32_t k, uint32_t j_top) {           // the steps j_top, j_top / 2 .. 1 of merge size k on the registers
    for (uint32_t j = j_top; j >= E; j >>= 1) {
#pragma unroll
      for (uint32_t q = 0; q < E; q++) {
        const uint32_t i = tid * E + q;
        const unsigned long long mine = e[q];
        const uint32_t olo = kd_lane_xor((uint32_t)mine, j / E), ohi = kd_lane_xor((uint32_t)(mine >> 32), j / E);
        const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
        const bool want_min = ((i & j) == 0) == ((i & k) == 0);
        e[q] = ((other < mine) == want_min) ? other : mine;
      }
    }
#pragma unroll
    for (uint32_t j = E >> 1; j > 0; j >>= 1) {
      if (j > j_top) continue;
#pragma unroll
      for (uint32_t q = 0; q < E; q++) {
        if (q & j) continue;
        const uint32_t i = tid * E + q;
        const unsigned long long a = e[q], b = e[q | j];
        if ((a > b) == ((i & k) == 0)) { e[q] = b; e[q | j] = a; }
      }
    }
  };
  load();
  for (uint32_t k = 2; k <= W && k <= S; k <<= 1) in_wave(k, k >> 1);
  store();
  for (uint32_t k = 2 * W; k <= S; k <<= 1) {
    for (uint32_t j = k >> 1; j >= W; j >>= 1) {
#pragma unroll
      for (uint32_t r = 0; r < S / 2 / T; r++) {
        const uint32_t t = tid + r * T;
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
        const unsigned long long a = key[i], b = key[p];
        if ((a > b) == ((i & k) == 0)) { key[i] = b; key[p] = a; }
      }
      __syncthreads();
    }
    load();
    in_wave(k, W >> 1);
    store();
  }
}
__global__ void __launch_bounds__(KD_LDS_T) k_kd_lds(const KdSeg *segs, const float4 *pos_cur, const uint32_t *idx, uint32_t *bbmin_g, uint32_t *bbmax_g,
                                                     int32_t *perm, uint8_t *dim_out) {
  constexpr uint32_t S = KD_LDS_S, T = KD_LDS_T;
  __shared__ float crd[3][S];                              // coordinates by local id (the photon's place in the segment on entry)
  __shared__ uint16_t lid[2][S];                           // array position -> local id, double-buffered over the sorts
  __shared__ uint32_t spos[S];
  __shared__ uint16_t sstart[S], slen[S];
  __shared__ uint8_t sdim[S];
  __shared__ union { unsigned long long key[S]; uint32_t bb[(S / 2) * 6]; } u;   // boxes (by start / 2; segments of >= 2 photons) or sort keys
  const KdSeg sg = segs[blockIdx.x];
  const uint32_t L = sg.len, tid = threadIdx.x;
  if (L == 0 || L > S) return;                             // (never: the host sizes the segments)
  for (uint32_t i = tid; i < S; i += T) {
    if (i < L) {
      const float4 p = pos_cur[sg.start + i];
      crd[0][i] = p.x; crd[1][i] = p.y; crd[2][i] = p.z;
      spos[i] = sg.pos; sstart[i] = 0; slen[i] = (uint16_t)L;
    } else { spos[i] = 0; sstart[i] = (uint16_t)i; slen[i] = 0; crd[0][i] = crd[1][i] = crd[2][i] = 0.f; }
    lid[0][i] = (uint16_t)i; sdim[i] = 0;
  }
  int cur = 0;
  bool top = true;
  for (uint32_t k = tid; k < (S / 2) * 6; k += T) u.bb[k] = (k % 6 < 3) ? 0xFFFFFFFFu : 0u;
  for (uint32_t maxlen = L; maxlen >= 1; maxlen = find_median(maxlen)) {
    __syncthreads();
    // boxes of the segments with at least two photons: runs of lanes with one segment reduce towards their last lane
    // (a segmented scan over the wave by lane-shift moves: four steps inside the rows of 16 lanes, two across rows; a
    // lane takes what arrives iff it comes from its own segment), which adds the bounds to the segment's box -- one atomic
    // per run, axis and bound instead of one per photon on a few addresses
#pragma unroll
    for (uint32_t r = 0; r < S / T; r++) {
      const uint32_t i = tid + r * T, lane = tid & 63u;
      const bool active = spos[i] != 0 && slen[i] >= 2;
      const uint32_t s = active ? (uint32_t)sstart[i] : 0xFFFFFFFFu;
      uint32_t lo[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, hi[3] = { 0, 0, 0 };
      if (active) { const uint32_t l = lid[cur][i]; for (int d = 0; d < 3; d++) { lo[d] = hi[d] = sortable(crd[d][l]); } }
#define KD_SEG_STEP(CTRL, ROWS) {                                                                      \
        const bool take = kd_lane_move<CTRL, ROWS>(0xFFFFFFFEu, s) == s;                                \
        for (int d = 0; d < 3; d++) {                                                                   \
          const uint32_t a = kd_lane_move<CTRL, ROWS>(lo[d], lo[d]), b = kd_lane_move<CTRL, ROWS>(hi[d], hi[d]); \
          if (take) { lo[d] = min(lo[d], a); hi[d] = max(hi[d], b); }                                   \
        } }
      if (maxlen > 1) KD_SEG_STEP(0x111, 0xF)                 // row_shr:1 (a lane without a source keeps the first argument)
      if (maxlen > 2) KD_SEG_STEP(0x112, 0xF)
      if (maxlen > 4) KD_SEG_STEP(0x114, 0xF)
      if (maxlen > 8) KD_SEG_STEP(0x118, 0xF)
      KD_SEG_STEP(0x142, 0xA)                                 // row_bcast:15 into rows 1 and 3
      KD_SEG_STEP(0x143, 0xC)                                 // row_bcast:31 into rows 2 and 3
#undef KD_SEG_STEP
      const uint32_t s_after = __shfl_down(s, 1, 64);
      if (active && (lane == 63u || s_after != s))
        for (int d = 0; d < 3; d++) { atomicMin(&u.bb[(s >> 1) * 6 + d], lo[d]); atomicMax(&u.bb[(s >> 1) * 6 + 3 + d], hi[d]); }
    }
    __syncthreads();
    if (top && tid < 3) {                                   // the subtree's own box, for the caller (the root's is the map's)
      const uint32_t one = sortable(crd[tid][0]);
      bbmin_g[(size_t)sg.pos * 3 + tid] = L >= 2 ? u.bb[tid] : one; bbmax_g[(size_t)sg.pos * 3 + tid] = L >= 2 ? u.bb[3 + tid] : one;
    }
    top = false;
    // widest axis and sort key of every position
    unsigned long long mykey[S / T]; uint32_t myd[S / T];
#pragma unroll
    for (uint32_t r = 0; r < S / T; r++) {
      const uint32_t i = tid + r * T, sp = spos[i];
      const uint32_t d = (sp != 0 && slen[i] >= 2) ? (uint32_t)widest_dim3(&u.bb[(sstart[i] >> 1) * 6], &u.bb[(sstart[i] >> 1) * 6 + 3]) : 0u;
      const uint32_t c = sp ? sortable(crd[d][lid[cur][i]]) : 0u, grp = sp ? sstart[i] : i;
      myd[r] = d;
      mykey[r] = ((unsigned long long)grp << (32 + KD_LDS_BITS)) | ((unsigned long long)c << KD_LDS_BITS) | i;
    }
    __syncthreads();                                       // (the keys overwrite the boxes)
#pragma unroll
    for (uint32_t r = 0; r < S / T; r++) { const uint32_t i = tid + r * T; u.key[i] = mykey[r]; sdim[i] = (uint8_t)myd[r]; }
    __syncthreads();
    if (maxlen > 64) {
      kd_lds_sort<S, T>(u.key, tid);
#pragma unroll
      for (uint32_t r = 0; r < S / T; r++) { const uint32_t i = tid + r * T; lid[cur ^ 1][i] = lid[cur][(uint32_t)(u.key[i] & (S - 1u))]; }
    } else {
#pragma unroll
      for (uint32_t r = 0; r < S / T; r++) {
        const uint32_t i = tid + r * T;
        uint32_t np = i;
        if (spos[i] != 0 && slen[i] >= 2) {
          const uint32_t s = sstart[i], l = slen[i];
          const unsigned long long k = u.key[i];
          uint32_t rank = 0;
          for (uint32_t j = 0; j < l; j++) rank += u.key[s + j] < k ? 1u : 0u;
          np = s + rank;
        }
        lid[cur ^ 1][np] = lid[cur][i];
      }
    }
    __syncthreads();
    cur ^= 1;
    // the medians become nodes, the sides become the next level's segments
#pragma unroll
    for (uint32_t r = 0; r < S / T; r++) {
      const uint32_t i = tid + r * T, sp = spos[i];
      if (!sp) continue;
      const uint32_t s = sstart[i], l = slen[i], m = find_median(l);
      if (i == s + m) { perm[sp - 1] = (int32_t)idx[sg.start + lid[cur][i]]; dim_out[sp - 1] = sdim[i]; spos[i] = 0; sstart[i] = (uint16_t)i; }
      else if (i < s + m) { spos[i] = 2 * sp; slen[i] = (uint16_t)m; }
      else { spos[i] = 2 * sp + 1; sstart[i] = (uint16_t)(s + m + 1); slen[i] = (uint16_t)(l - m - 1); }
    }
    for (uint32_t k = tid; k < (S / 2) * 6; k += T) u.bb[k] = (k % 6 < 3) ? 0xFFFFFFFFu : 0u;      // (the keys are done with: empty boxes for the next level)
  }
}
static_assert(KD_LDS_S == (1u << KD_LDS_BITS) && KD_LDS_S % (2 * KD_LDS_T) == 0 && 32 + 2 * KD_LDS_BITS <= 64, "sort keys: group, coordinate, position");

static float unsortable_host(uint32_t u) { u ^= ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu); float f; memcpy(&f, &u, 4); return f; }

// Queues the build on the context's stream and returns; the caller waits for the stream.  d_pos: the positions as float4
// (w unused); want_box: min xyz, max xyz of all positions are copied to the context's pinned words (kd_take_box).
int rt_kd_build_device(rt_ctx *ctx, const float4 *d_pos, uint32_t n, int32_t *d_perm, uint8_t *d_dim, bool want_box) {
  if (n == 0) return rt_fail(ctx, RT_ERR_EMPTY_TREE, "tried to build a k-d tree with no data!");   // kdtree.js:6
  if (n >= 0x7FFFFFFFu) return rt_fail(ctx, RT_ERR_TOO_LARGE, "photon map too large");
  // the levels handled over the whole array: while the longest segment is too long for one workgroup
  std::vector<KdSeg> segs = { { 1u, 0u, n } };
  std::vector<uint32_t> &work = ctx->kd_work;                 // (read by a queued copy: it has to outlive this call)
  work.assign(2, 0u);                                        // first the table (start, length) by heap position of the levels split over the whole array
  int global_levels = 0;
  for (uint32_t maxlen = n; maxlen > KD_LDS_S; maxlen = find_median(maxlen)) {
    std::vector<KdSeg> next; next.reserve(segs.size() * 2);
    for (const KdSeg &g : segs) {
      work.push_back(g.start); work.push_back(g.len);        // (heap positions in order: this is entry g.pos)
      if (g.len == 0) { next.push_back({ 2 * g.pos, g.start, 0u }); next.push_back({ 2 * g.pos + 1, g.start, 0u }); continue; }
      const uint32_t m = find_median(g.len);
      next.push_back({ 2 * g.pos, g.start, m }); next.push_back({ 2 * g.pos + 1, g.start + m + 1, g.len - m - 1 });
    }
    segs.swap(next); global_levels++;
  }
  if (global_levels == 0) work.clear();
  const size_t n_tab = work.size() / 2;
  for (const KdSeg &g : segs) if (g.len) { work.push_back(g.pos); work.push_back(g.start); work.push_back(g.len); }
  const size_t n_work = (work.size() - 2 * n_tab) / 3;
  static_assert(sizeof(KdSeg) == 12, "three words per segment");
  const size_t n_bb = ((size_t)2 << global_levels) + 2;       // boxes by heap position, levels 0 .. global_levels
  // one grow-only workspace per context
  size_t tmp_bytes = 0;
  int seg_bits = global_levels;
  if (global_levels)
    RT_HIP(ctx, kd_level_sort(nullptr, tmp_bytes, nullptr, nullptr, nullptr, nullptr, n, 32 + seg_bits, ctx->stream));
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const bool glob = global_levels > 0;
  const size_t o_idxA = carve((size_t)n * 4), o_idxB = carve(glob ? (size_t)n * 4 : 0), o_keyA = carve(glob ? (size_t)n * 8 : 0), o_keyB = carve(glob ? (size_t)n * 8 : 0);
  const size_t o_pos = carve(glob ? (size_t)n * 4 : 0);
  const size_t o_bbmin = carve(n_bb * 12), o_bbmax = carve(n_bb * 12), o_segdim = carve(n_bb), o_segs = carve(work.size() * 4), o_tmp = carve(tmp_bytes);
  const size_t o_pcur = carve(glob ? (size_t)n * 16 : 0);
  RT_HIP(ctx, ctx->ws_kd.reserve(off));
  char *base = (char *)ctx->ws_kd.p;
  uint32_t *idx_cur = (uint32_t *)(base + o_idxA), *idx_nxt = (uint32_t *)(base + o_idxB);
  uint64_t *keyA = (uint64_t *)(base + o_keyA), *keyB = (uint64_t *)(base + o_keyB);
  SegState sg = { (uint32_t *)(base + o_pos), (const uint2 *)(base + o_segs) };
  uint32_t *bbmin = (uint32_t *)(base + o_bbmin), *bbmax = (uint32_t *)(base + o_bbmax);
  uint8_t *segdim = (uint8_t *)(base + o_segdim);
  KdSeg *d_segs = (KdSeg *)(base + o_segs + n_tab * 8);
  float4 *pos_cur = (float4 *)(base + o_pcur);               // the positions in the array's current order, from the first split on
  const uint32_t grid = div_up(n, 256), grid_chunks = div_up(n, 256u * KD_BBOX_ITEMS);
  if (glob) {
    const uint32_t n_box_words = (uint32_t)(n_bb * 3);
    RT_LAUNCH(ctx, "kd_init", k_kd_init, div_up(std::max(n, n_box_words), 256), 256, n, idx_cur, sg, n_box_words, bbmin, bbmax);
    RT_LAUNCH(ctx, "kd_bbox", k_kd_bbox_root, grid_chunks, 256, n, d_pos, bbmin, bbmax);
  } else {
    RT_LAUNCH(ctx, "kd_init", k_kd_iota, grid, 256, n, idx_cur);
  }
  RT_HIP(ctx, hipMemcpyAsync(base + o_segs, work.data(), work.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  uint32_t first = 1;
  for (int level = 0; level < global_levels; level++, first <<= 1) {      // heap positions [first, 2 first) form one level
    RT_LAUNCH(ctx, "kd_keys", k_kd_keys, grid, 256, n, level == 0 ? d_pos : (const float4 *)pos_cur, sg, first, level, (const uint32_t *)bbmin, (const uint32_t *)bbmax, segdim, keyA);
    {
      KernelTimer timer_(ctx, "kd_sort");
      RT_HIP(ctx, kd_level_sort(base + o_tmp, tmp_bytes, keyA, keyB, idx_cur, idx_nxt, n, 32 + level, ctx->stream));
    }
    std::swap(idx_cur, idx_nxt);
    // (the workgroups of the lower levels take their subtree's box themselves)
    RT_LAUNCH(ctx, "kd_split", k_kd_split, grid_chunks, 256, n, d_pos, (const uint32_t *)idx_cur, sg, (const uint8_t *)segdim, d_perm, d_dim, pos_cur,
              level + 1 < global_levels ? 1 : 0, bbmin, bbmax);
  }
  if (n_work)
    RT_LAUNCH(ctx, "kd_lds", k_kd_lds, (uint32_t)n_work, KD_LDS_T, (const KdSeg *)d_segs, glob ? (const float4 *)pos_cur : d_pos, (const uint32_t *)idx_cur, bbmin, bbmax, d_perm, d_dim);
  RT_HIP(ctx, hipGetLastError());
  if (want_box) {                                            // the root segment's box (heap position 1) is never cleared again
    RT_HIP(ctx, hipMemcpyAsync(ctx->pinned, bbmin + 3, 12, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(ctx->pinned + 3, bbmax + 3, 12, hipMemcpyDeviceToHost, ctx->stream));
  }
  return RT_OK;
}
// after the stream has been waited for: the box that a build with want_box left in the context's pinned words
static void kd_take_box(rt_ctx *ctx, float *bbox6) {
  for (int d = 0; d < 3; d++) { bbox6[d] = unsortable_host(ctx->pinned[d]); bbox6[3 + d] = unsortable_host(ctx->pinned[3 + d]); }
}

extern "C" int rt_kd_build(rt_ctx *ctx, const float *pos, uint32_t n, int32_t *perm, uint8_t *dim) {
  if (!ctx) return RT_ERR_INVALID;
  if (n == 0) return rt_fail(ctx, RT_ERR_EMPTY_TREE, "tried to build a k-d tree with no data!");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem tpos, tpos4, tperm, tdim; const void *dpos; void *dperm, *ddim;
  RT_TRY(rt_stage_in(ctx, pos, (size_t)n * 12, tpos, &dpos));
  RT_TRY(rt_stage_out_begin(ctx, perm, (size_t)n * 4, tperm, &dperm));
  RT_TRY(rt_stage_out_begin(ctx, dim, (size_t)n, tdim, &ddim));
  RT_HIP(ctx, tpos4.reserve((size_t)n * 16));
  RT_LAUNCH(ctx, "kd_pos4", k_kd_pos4, div_up(n, 256), 256, n, (const float *)dpos, tpos4.as<float4>());
  RT_TRY(rt_kd_build_device(ctx, tpos4.as<float4>(), n, (int32_t *)dperm, (uint8_t *)ddim, false));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RT_TRY(rt_stage_out_end(ctx, perm, (size_t)n * 4, tperm));
  RT_TRY(rt_stage_out_end(ctx, dim, (size_t)n, tdim));
  return RT_OK;
}

// ------------------------------------------------------------------ photon map object
// Where the photons come from: the boundary's 11-double records, a list of full fp32 records (11 floats) or a list of
// packed photons (9 floats: position, direction, power * colour).
enum { MAP_SRC_REC64 = 0, MAP_SRC_FULL32 = 1, MAP_SRC_PACKED = 2 };
template <int SRC> __device__ __forceinline__ void map_load(const void *src, size_t i, float pos[3], float pay[6]) {
  if (SRC == MAP_SRC_REC64) {
    const double *r = (const double *)src + i * 11;
    for (int d = 0; d < 3; d++) { pos[d] = (float)r[d]; pay[d] = (float)r[3 + d]; pay[3 + d] = (float)(r[7 + d] * r[6]); }
  } else if (SRC == MAP_SRC_FULL32) {
    const float *r = (const float *)src + i * 11;
    for (int d = 0; d < 3; d++) { pos[d] = r[d]; pay[d] = r[3 + d]; pay[3 + d] = r[7 + d] * r[6]; }
  } else {
    const float *r = (const float *)src + i * 9;
    for (int d = 0; d < 3; d++) { pos[d] = r[d]; pay[d] = r[3 + d]; pay[3 + d] = r[6 + d]; }
  }
}
template <int SRC> __global__ void __launch_bounds__(256) k_map_extract_pos(uint32_t n, const void *src, float4 *pos4) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float pos[3], pay[6];
  map_load<SRC>(src, i, pos, pay);
  pos4[i] = make_float4(pos[0], pos[1], pos[2], 0.f);
}
template <int SRC> __global__ void __launch_bounds__(256) k_map_permute(uint32_t n, const void *src, const int32_t *perm, const uint8_t *dim,
                                                                        float4 *node, float *pay_out, double *rec64) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  float pos[3], pay[6];
  map_load<SRC>(src, (size_t)perm[s], pos, pay);
  node[s] = make_float4(pos[0], pos[1], pos[2], __int_as_float((int)dim[s]));
  float2 *p = (float2 *)(pay_out + (size_t)s * 6);
  p[0] = make_float2(pay[0], pay[1]); p[1] = make_float2(pay[2], pay[3]); p[2] = make_float2(pay[4], pay[5]);
  if (SRC == MAP_SRC_REC64 && rec64) {
    const double *r = (const double *)src + (size_t)perm[s] * 11;
    for (int c = 0; c < 11; c++) rec64[(size_t)s * 11 + c] = r[c];
  }
}
// The same from fp64 records, a workgroup moving 256 records as 256 x 11 doubles: consecutive lanes read consecutive doubles
// of a record (88 contiguous bytes at a place the permutation picks) and write one contiguous block of the ordered copy;
// node and payload are then made from the workgroup's staged copy.  (One record per lane, as above, reads eleven doubles
// 88 bytes apart from its neighbours': 0.29 ms per million photons.)
__global__ void __launch_bounds__(256) k_map_permute_rec64(uint32_t n, const double *src, const int32_t *perm, const uint8_t *dim,
                                                           float4 *node, float *pay_out, double *rec64) {
  __shared__ double stage[256 * 11];
  __shared__ uint32_t sperm[256];
  const uint32_t s0 = blockIdx.x * 256u;
  if (s0 >= n) return;                                       // (workgroup-uniform)
  const uint32_t cnt = min(256u, n - s0);
  sperm[threadIdx.x] = threadIdx.x < cnt ? (uint32_t)perm[s0 + threadIdx.x] : 0u;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < cnt * 11u; k += 256u) {
    const uint32_t r = k / 11u, w = k - r * 11u;
    const double v = src[(size_t)sperm[r] * 11 + w];
    stage[k] = v;
    if (rec64) rec64[(size_t)s0 * 11 + k] = v;
  }
  __syncthreads();
  if (threadIdx.x >= cnt) return;
  const double *r = stage + threadIdx.x * 11u;
  const uint32_t s = s0 + threadIdx.x;
  node[s] = make_float4((float)r[0], (float)r[1], (float)r[2], __int_as_float((int)dim[s]));
  float2 *p = (float2 *)(pay_out + (size_t)s * 6);
  p[0] = make_float2((float)r[3], (float)r[4]); p[1] = make_float2((float)r[5], (float)(r[7] * r[6])); p[2] = make_float2((float)(r[8] * r[6]), (float)(r[9] * r[6]));
}

// src: device memory
int rt_map_build_device(rt_ctx *ctx, const void *src, int src_kind, uint32_t n, rt_photon_map **out) {
  *out = nullptr;
  if (n == 0) return rt_fail(ctx, RT_ERR_EMPTY_TREE, "tried to build a k-d tree with no data!");
  DevMem tpos;
  struct Back { rt_ctx *c; DevMem &d; ~Back() { c->pool_give(d); } } tpos_back{ ctx, tpos };      // (the build below synchronises before it returns)
  RT_HIP(ctx, ctx->pool_take(tpos, (size_t)n * 16));
  rt_photon_map *m = new rt_photon_map();
  m->ctx = ctx; m->n = n;
  hipError_t e = ctx->pool_take(m->node, (size_t)n * 16);
  if (e == hipSuccess) e = ctx->pool_take(m->pay, (size_t)n * 24);
  if (e == hipSuccess) e = ctx->pool_take(m->perm, (size_t)n * 4);
  if (e == hipSuccess) e = ctx->pool_take(m->dim, (size_t)n);
  // fp64 copy of the records for the parity (fp64) renderer; only from fp64 records, and skipped for very large maps
  const bool keep64 = src_kind == MAP_SRC_REC64 && n <= (1u << 24);
  if (e == hipSuccess && keep64) e = ctx->pool_take(m->rec64, (size_t)n * 88);
  if (e != hipSuccess) { delete m; return rt_fail(ctx, RT_ERR_HIP, "photon map allocation failed: %s", hipGetErrorString(e)); }
  const uint32_t grid = div_up(n, 256);
  if (src_kind == MAP_SRC_REC64) RT_LAUNCH(ctx, "map_extract_pos", k_map_extract_pos<MAP_SRC_REC64>, grid, 256, n, src, tpos.as<float4>());
  else if (src_kind == MAP_SRC_FULL32) RT_LAUNCH(ctx, "map_extract_pos", k_map_extract_pos<MAP_SRC_FULL32>, grid, 256, n, src, tpos.as<float4>());
  else RT_LAUNCH(ctx, "map_extract_pos", k_map_extract_pos<MAP_SRC_PACKED>, grid, 256, n, src, tpos.as<float4>());
  int rc = rt_kd_build_device(ctx, tpos.as<float4>(), n, m->perm.as<int32_t>(), m->dim.as<uint8_t>(), true);
  if (rc != RT_OK) { (void)hipStreamSynchronize(ctx->stream); delete m; return rc; }
  double *r64 = keep64 ? m->rec64.as<double>() : (double *)nullptr;
  if (src_kind == MAP_SRC_REC64) RT_LAUNCH(ctx, "map_permute", k_map_permute_rec64, grid, 256, n, (const double *)src, m->perm.as<int32_t>(), m->dim.as<uint8_t>(), m->node.as<float4>(), m->pay.as<float>(), r64);
  else if (src_kind == MAP_SRC_FULL32) RT_LAUNCH(ctx, "map_permute", k_map_permute<MAP_SRC_FULL32>, grid, 256, n, src, m->perm.as<int32_t>(), m->dim.as<uint8_t>(), m->node.as<float4>(), m->pay.as<float>(), r64);
  else RT_LAUNCH(ctx, "map_permute", k_map_permute<MAP_SRC_PACKED>, grid, 256, n, src, m->perm.as<int32_t>(), m->dim.as<uint8_t>(), m->node.as<float4>(), m->pay.as<float>(), r64);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) { delete m; return rt_fail(ctx, RT_ERR_HIP, "photon map build failed"); }
  kd_take_box(ctx, m->bbox);
  *out = m;
  return RT_OK;
}

extern "C" int rt_photon_map_build(rt_ctx *ctx, const double *rec, uint32_t n, rt_photon_map **out) {
  if (!ctx || !out) return RT_ERR_INVALID;
  *out = nullptr;
  if (n == 0) return rt_fail(ctx, RT_ERR_EMPTY_TREE, "tried to build a k-d tree with no data!");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem trec; const void *drec;
  RT_TRY(rt_stage_in(ctx, rec, (size_t)n * 88, trec, &drec));
  return rt_map_build_device(ctx, drec, MAP_SRC_REC64, n, out);
}
// packed photons: n x 9 floats (position, direction, power * colour), host or device memory
extern "C" int rt_photon_map_build_packed(rt_ctx *ctx, const float *photons, uint32_t n, rt_photon_map **out) {
  if (!ctx || !out) return RT_ERR_INVALID;
  *out = nullptr;
  if (n == 0) return rt_fail(ctx, RT_ERR_EMPTY_TREE, "tried to build a k-d tree with no data!");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem trec; const void *drec;
  RT_TRY(rt_stage_in(ctx, photons, (size_t)n * 36, trec, &drec));
  return rt_map_build_device(ctx, drec, MAP_SRC_PACKED, n, out);
}
extern "C" int rt_photon_map_build_lists(rt_photon_lists *lists, int which, rt_photon_map **out) {
  if (!lists || !out) return RT_ERR_INVALID;
  *out = nullptr;
  const void *rec = nullptr; uint32_t n = 0, words = 0; int precision = 0;
  int rc = rt_lists_view(lists, which, &rec, &n, &words, &precision);
  if (rc != RT_OK) return rc;
  rt_ctx *ctx = rt_lists_ctx(lists);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) return rt_fail(ctx, RT_ERR_EMPTY_TREE, "tried to build a k-d tree with no data!");
  const int kind = words == 9 ? MAP_SRC_PACKED : (words == 22 ? MAP_SRC_REC64 : MAP_SRC_FULL32);
  return rt_map_build_device(ctx, rec, kind, n, out);
}
// A map's blocks go back to its context's pool; everything queued that may still read them is waited for first (hipFree
// would have waited for the whole device).
extern "C" void rt_photon_map_free(rt_photon_map *m) {
  if (!m) return;
  rt_ctx *ctx = m->ctx;
  (void)hipSetDevice(ctx->device);
  if (rt_synchronize(ctx) == RT_OK) { ctx->pool_give(m->node); ctx->pool_give(m->pay); ctx->pool_give(m->perm); ctx->pool_give(m->dim); ctx->pool_give(m->rec64); }
  delete m;
}
extern "C" uint32_t rt_photon_map_size(const rt_photon_map *m) { return m ? m->n : 0; }
extern "C" int rt_photon_map_nodes(rt_photon_map *m, int32_t *perm, uint8_t *dim) {
  if (!m) return RT_ERR_INVALID;
  rt_ctx *ctx = m->ctx;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  if (perm) RT_HIP(ctx, hipMemcpy(perm, m->perm.p, (size_t)m->n * 4, hipMemcpyDefault));
  if (dim) RT_HIP(ctx, hipMemcpy(dim, m->dim.p, (size_t)m->n, hipMemcpyDefault));
  return RT_OK;
}

// ------------------------------------------------------------------ radius query
__global__ void __launch_bounds__(256) k_kd_radius(DevMap M, const float *q3, uint32_t nq, double r, uint32_t cap,
                                                   uint32_t *count, int32_t *idx) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const double x0 = q3[(size_t)q * 3], x1 = q3[(size_t)q * 3 + 1], x2 = q3[(size_t)q * 3 + 2];
  const double r2 = r * r;
  uint32_t c = 0;
  kd_radius_walk(M, x0, x1, x2, r, [&](uint32_t slot, float4 nd) {
    if (sq_dist(x0, x1, x2, (double)nd.x, (double)nd.y, (double)nd.z) < r2) {
      if (c < cap) idx[(size_t)q * cap + c] = M.perm[slot];
      c++;
    }
  });
  count[q] = c;
}

extern "C" int rt_kd_radius(rt_photon_map *m, const float *queries, uint32_t nq, double r, uint32_t cap, uint32_t *count, int32_t *idx) {
  if (!m) return RT_ERR_INVALID;
  rt_ctx *ctx = m->ctx;
  if (!queries || !count || (cap > 0 && !idx)) return rt_fail(ctx, RT_ERR_INVALID, "rt_kd_radius: queries, count and (for cap > 0) idx are required");
  if (nq == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem tq, tc, ti; const void *dq; void *dc, *di;
  RT_TRY(rt_stage_in(ctx, queries, (size_t)nq * 12, tq, &dq));
  RT_TRY(rt_stage_out_begin(ctx, count, (size_t)nq * 4, tc, &dc));
  RT_TRY(rt_stage_out_begin(ctx, idx, (size_t)nq * cap * 4, ti, &di));
  RT_LAUNCH(ctx, "kd_radius", k_kd_radius, div_up(nq, 256), 256, m->view(), (const float *)dq, nq, r, cap, (uint32_t *)dc, (int32_t *)di);
  RT_HIP(ctx, hipGetLastError());
  RT_TRY(rt_stage_out_end(ctx, count, (size_t)nq * 4, tc));
  RT_TRY(rt_stage_out_end(ctx, idx, (size_t)nq * cap * 4, ti));
  return RT_OK;
}

// ------------------------------------------------------------------ k-nearest query
// One lane per query; each lane owns a max-heap of k (squared distance, slot) entries in LDS,
// laid out [entry][lane] so that the 64 lanes of a wave touch 64 consecutive words.  Order is
// (d2, input index): the worst element is the largest d2 and, among equal d2, the largest index.
#define KNN_LANES 64
struct KnnHeap {
  double *d2; uint32_t *slot; const int32_t *perm; uint32_t k, size; int lane;
  __device__ __forceinline__ double &D(uint32_t i) { return d2[(size_t)i * KNN_LANES + lane]; }
  __device__ __forceinline__ uint32_t &S(uint32_t i) { return slot[(size_t)i * KNN_LANES + lane]; }
  __device__ __forceinline__ bool worse(double da, uint32_t sa, double db, uint32_t sb) const {
    return da > db || (da == db && perm[sa] > perm[sb]);
  }
  __device__ __forceinline__ void sift_down(uint32_t i, uint32_t n, double dv, uint32_t sv) {
    for (;;) {
      uint32_t c = 2 * i + 1;
      if (c >= n) break;
      double dc = D(c); uint32_t sc = S(c);
      if (c + 1 < n) { double dr = D(c + 1); uint32_t sr = S(c + 1); if (worse(dr, sr, dc, sc)) { c++; dc = dr; sc = sr; } }
      if (!worse(dc, sc, dv, sv)) break;
      D(i) = dc; S(i) = sc; i = c;
    }
    D(i) = dv; S(i) = sv;
  }
  __device__ __forceinline__ void offer(double dv, uint32_t sv) {
    if (size < k) {                                   // sift up
      uint32_t i = size++;
      while (i > 0) {
        uint32_t p = (i - 1) / 2;
        double dp = D(p); uint32_t sp = S(p);
        if (!worse(dv, sv, dp, sp)) break;
        D(i) = dp; S(i) = sp; i = p;
      }
      D(i) = dv; S(i) = sv;
    } else if (worse(D(0), S(0), dv, sv)) {
      sift_down(0, size, dv, sv);
    }
  }
  __device__ __forceinline__ bool full() const { return size == k; }
};

template <typename F> __device__ __forceinline__ void knn_search(const DevMap &M, double x0, double x1, double x2, KnnHeap &H, F &&dist2) {
  const uint32_t n = M.n;
  uint32_t pos = 1; int depth = 0; uint64_t pending = 0;
  for (;;) {
    if (pos <= n) {
      float4 nd = M.node[pos - 1];
      H.offer(dist2(nd), pos - 1);
      int dim = node_dim(nd);
      double diff = ((dim == 0) ? x0 : (dim == 1) ? x1 : x2) - (double)((dim == 0) ? nd.x : (dim == 1) ? nd.y : nd.z);
      if (2 * (uint64_t)pos <= n) pending |= (uint64_t)1 << depth;        // has at least one child
      pos = (diff < 0) ? 2 * pos : 2 * pos + 1;
      depth++;
      continue;
    }
    bool resumed = false;
    while (pending) {
      int k = 63 - __clzll((long long)pending);
      pending &= ~((uint64_t)1 << k);
      uint32_t anc = pos >> (depth - k);
      float4 nd = M.node[anc - 1];
      int dim = node_dim(nd);
      double diff = ((dim == 0) ? x0 : (dim == 1) ? x1 : x2) - (double)((dim == 0) ? nd.x : (dim == 1) ? nd.y : nd.z);
      // every point beyond the plane has d2 >= diff*diff (also after rounding), so `>` prunes safely, ties included
      if (!H.full() || !(diff * diff > H.D(0))) { pos = (diff < 0) ? 2 * anc + 1 : 2 * anc; depth = k + 1; resumed = true; break; }
      pos = anc; depth = k;
    }
    if (!resumed) break;
  }
}

__global__ void __launch_bounds__(KNN_LANES) k_kd_knn(DevMap M, const float *q3, uint32_t nq, uint32_t k, int32_t *idx, double *d2out) {
  extern __shared__ double knn_lds[];
  uint32_t q = blockIdx.x * KNN_LANES + threadIdx.x;
  KnnHeap H; H.d2 = knn_lds; H.slot = (uint32_t *)(knn_lds + (size_t)k * KNN_LANES); H.perm = M.perm; H.k = k; H.size = 0; H.lane = threadIdx.x;
  if (q >= nq) return;
  const double x0 = q3[(size_t)q * 3], x1 = q3[(size_t)q * 3 + 1], x2 = q3[(size_t)q * 3 + 2];
  knn_search(M, x0, x1, x2, H, [&](float4 nd) { return sq_dist(x0, x1, x2, (double)nd.x, (double)nd.y, (double)nd.z); });
  // heap sort: repeatedly move the worst to the tail -> ascending (d2, index)
  for (uint32_t j = k; j > H.size; j--) { idx[(size_t)q * k + j - 1] = -1; d2out[(size_t)q * k + j - 1] = __builtin_huge_val(); }
  for (uint32_t m = H.size; m > 0; m--) {
    double dw = H.D(0); uint32_t sw = H.S(0);
    idx[(size_t)q * k + m - 1] = M.perm[sw]; d2out[(size_t)q * k + m - 1] = dw;
    if (m > 1) { double dl = H.D(m - 1); uint32_t sl = H.S(m - 1); H.sift_down(0, m - 1, dl, sl); }
  }
}

static int knn_lds_bytes(uint32_t k) { return (int)((size_t)k * KNN_LANES * 12); }

extern "C" int rt_kd_knn(rt_photon_map *m, const float *queries, uint32_t nq, uint32_t k, int32_t *idx, double *d2) {
  if (!m) return RT_ERR_INVALID;
  rt_ctx *ctx = m->ctx;
  if (k == 0 || k > 192) return rt_fail(ctx, RT_ERR_INVALID, "k must be in 1..192 (LDS-resident heap), got %u", k);
  if (!queries || !idx || !d2) return rt_fail(ctx, RT_ERR_INVALID, "rt_kd_knn: queries, idx and d2 are required");
  if (nq == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem tq, ti, td; const void *dq; void *di, *dd;
  RT_TRY(rt_stage_in(ctx, queries, (size_t)nq * 12, tq, &dq));
  RT_TRY(rt_stage_out_begin(ctx, idx, (size_t)nq * k * 4, ti, &di));
  RT_TRY(rt_stage_out_begin(ctx, d2, (size_t)nq * k * 8, td, &dd));
  RT_HIP(ctx, hipFuncSetAttribute((const void *)k_kd_knn, hipFuncAttributeMaxDynamicSharedMemorySize, knn_lds_bytes(k)));
  {
    KernelTimer timer_(ctx, "kd_knn");
    hipLaunchKernelGGL(k_kd_knn, dim3(div_up(nq, KNN_LANES)), dim3(KNN_LANES), knn_lds_bytes(k), ctx->stream, m->view(),
                       (const float *)dq, nq, k, (int32_t *)di, (double *)dd);
  }
  RT_HIP(ctx, hipGetLastError());
  RT_TRY(rt_stage_out_end(ctx, idx, (size_t)nq * k * 4, ti));
  RT_TRY(rt_stage_out_end(ctx, d2, (size_t)nq * k * 8, td));
  return RT_OK;
}

// nearestNeighbourAndDistance(x) (kdtree.js:10-62): the reference never searches the far side of a splitting plane
// (`if (false)`, :34, :50), so the result is the closest node of ONE root-to-leaf path, settled from the leaf upwards
// with a strict `<` (:30, :46).  A path that leaves the tree through a missing upper child yields [undefined, undefined]
// (:13; `dist < undefined` is false all the way up): index -1, distance NaN here.  One lane per query.
__global__ void __launch_bounds__(256) k_kd_nearest_path(DevMap M, const float *q3, uint32_t nq, int32_t *idx, double *dist) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const double x0 = q3[(size_t)q * 3], x1 = q3[(size_t)q * 3 + 1], x2 = q3[(size_t)q * 3 + 2];
  const uint32_t n = M.n;
  uint32_t pos = 1;
  while (2 * (uint64_t)pos <= n) {                         // (:20: a node without children ends the descent)
    const float4 nd = M.node[pos - 1];
    const int dim = node_dim(nd);
    const bool below = ((dim == 0) ? x0 : (dim == 1) ? x1 : x2) < (double)((dim == 0) ? nd.x : (dim == 1) ? nd.y : nd.z);
    pos = below ? 2 * pos : 2 * pos + 1;
    if (pos > n) { idx[q] = -1; dist[q] = __builtin_nan(""); return; }
  }
  float4 nd = M.node[pos - 1];
  uint32_t best = pos;
  double bd = sqrt(sq_dist(x0, x1, x2, (double)nd.x, (double)nd.y, (double)nd.z));
  for (pos >>= 1; pos >= 1; pos >>= 1) {
    nd = M.node[pos - 1];
    const double d = sqrt(sq_dist(x0, x1, x2, (double)nd.x, (double)nd.y, (double)nd.z));
    if (d < bd) { bd = d; best = pos; }
  }
  idx[q] = M.perm[best - 1]; dist[q] = bd;
}

extern "C" int rt_kd_nearest_path(rt_photon_map *m, const float *queries, uint32_t nq, int32_t *idx, double *dist) {
  if (!m) return RT_ERR_INVALID;
  rt_ctx *ctx = m->ctx;
  if (!queries || !idx || !dist) return rt_fail(ctx, RT_ERR_INVALID, "rt_kd_nearest_path: queries, idx and dist are required");
  if (nq == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem tq, ti, td; const void *dq; void *di, *dd;
  RT_TRY(rt_stage_in(ctx, queries, (size_t)nq * 12, tq, &dq));
  RT_TRY(rt_stage_out_begin(ctx, idx, (size_t)nq * 4, ti, &di));
  RT_TRY(rt_stage_out_begin(ctx, dist, (size_t)nq * 8, td, &dd));
  {
    KernelTimer timer_(ctx, "kd_nearest_path");
    hipLaunchKernelGGL(k_kd_nearest_path, dim3(div_up(nq, 256)), dim3(256), 0, ctx->stream, m->view(), (const float *)dq, nq, (int32_t *)di, (double *)dd);
  }
  RT_HIP(ctx, hipGetLastError());
  RT_TRY(rt_stage_out_end(ctx, idx, (size_t)nq * 4, ti));
  RT_TRY(rt_stage_out_end(ctx, dist, (size_t)nq * 8, td));
  return RT_OK;
}

// ------------------------------------------------------------------ fused radiance gathers
// hits: pos3 normal3 albedo3 (fp32); rgb = sum * albedo/255 / (pi r^2)   (rayphoton.js:1779-1784, :1808-1815)
__global__ void __launch_bounds__(256) k_gather_radius(DevMap M, const float *hits, uint32_t nq, double r, float *rgb) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const float *h = hits + (size_t)q * 9;
  const double x0 = h[0], x1 = h[1], x2 = h[2];
  const float nx = h[3], ny = h[4], nz = h[5];
  const double r2 = r * r;
  float sr = 0, sg = 0, sb = 0;
  kd_radius_walk(M, x0, x1, x2, r, [&](uint32_t slot, float4 nd) {
    if (sq_dist(x0, x1, x2, (double)nd.x, (double)nd.y, (double)nd.z) < r2) {
      const float *pp = M.pay + (size_t)slot * 6;
      float w = fmaxf(0.0f, pp[0] * nx + pp[1] * ny + pp[2] * nz);
      sr += pp[3] * w; sg += pp[4] * w; sb += pp[5] * w;
    }
  });
  const float inv_area = (float)(1.0 / (3.141592653589793 * r2));
  rgb[(size_t)q * 3] = sr * (h[6] / 255.0f) * inv_area;
  rgb[(size_t)q * 3 + 1] = sg * (h[7] / 255.0f) * inv_area;
  rgb[(size_t)q * 3 + 2] = sb * (h[8] / 255.0f) * inv_area;
}

// `list` (optional): the queries k_gather_knn_wave left over (list[0 .. *list_count)), 64 to a workgroup; without it
// workgroup b serves queries [64 b, 64 b + 64)
__global__ void __launch_bounds__(KNN_LANES) k_gather_knn(DevMap M, const float *hits, uint32_t nq, uint32_t k, float *rgb, const uint32_t *list,
                                                          const uint32_t *list_count) {
  extern __shared__ double knn_lds[];
  uint32_t q = blockIdx.x * KNN_LANES + threadIdx.x;
  if (list) {
    if (q >= *list_count) return;
    q = list[q];
  }
  KnnHeap H; H.d2 = knn_lds; H.slot = (uint32_t *)(knn_lds + (size_t)k * KNN_LANES); H.perm = M.perm; H.k = k; H.size = 0; H.lane = threadIdx.x;
  if (q >= nq) return;
  const float *h = hits + (size_t)q * 9;
  const double x0 = h[0], x1 = h[1], x2 = h[2];
  const float nx = h[3], ny = h[4], nz = h[5];
  knn_search(M, x0, x1, x2, H, [&](float4 nd) { return sq_dist(x0, x1, x2, (double)nd.x, (double)nd.y, (double)nd.z); });
  float sr = 0, sg = 0, sb = 0;
  for (uint32_t j = 0; j < H.size; j++) {
    const float *pp = M.pay + (size_t)H.S(j) * 6;
    float w = fmaxf(0.0f, pp[0] * nx + pp[1] * ny + pp[2] * nz);
    sr += pp[3] * w; sg += pp[4] * w; sb += pp[5] * w;
  }
  const double r2 = H.size ? H.D(0) : 1.0;          // squared distance of the k-th nearest
  const float inv_area = (float)(1.0 / (3.141592653589793 * r2));
  rgb[(size_t)q * 3] = sr * (h[6] / 255.0f) * inv_area;
  rgb[(size_t)q * 3 + 1] = sg * (h[7] / 255.0f) * inv_area;
  rgb[(size_t)q * 3 + 2] = sb * (h[8] / 255.0f) * inv_area;
}


// ------------------------------------------------------------------ k-nearest gather, one wave per 64 queries
// Queries that arrive together are neighbours on a surface (the caller orders them by image tile), so their
// k-nearest sets overlap almost completely.  A wave therefore
//   0. descends the tree towards the centre of its queries to a subtree of k .. 4k photons and takes its search radii
//      from those photons: first from their density, then (if some query finds fewer than k photons inside its ball)
//      from the k-th smallest distance to them, which bounds the k-th neighbour distance from above;
//   1. takes the bounding box of its queries' balls and collects every photon inside with ONE cooperative walk of the
//      kd-tree (a queue of nodes in LDS, one node per lane per step);
//   2. checks that every query has at least k candidates inside its ball -- then all of its k nearest photons are
//      among the candidates, exactly;
//   3. lets every lane find the distance that separates its k nearest candidates from the rest by an
//      interpolation search on "how many candidates are closer than t" (the candidates are read from LDS as
//      wave-wide broadcasts; no per-lane storage at all, so occupancy is not limited by a top-k structure);
//   4. sums the radiance of the selected candidates.
// Distances are the same fp64 expression as everywhere else (sq_dist), ties at the k-th distance go to the lowest
// input index, so the selected set is exactly the one rt_kd_knn returns.  A wave whose candidates or node queue would
// overflow its LDS buffers hands its queries on as eight units of 8 queries, and those in turn as single queries, which
// k_gather_knn_single streams through the same buffers whatever their number of candidates).
#ifndef KNN2_CAP
#define KNN2_CAP 512
#endif
// first-try radius: this many times the radius that holds k photons at the density of the query's subtree
#ifndef KNN_FIRST_TRY
#define KNN_FIRST_TRY 1.3f
#endif
// a unit that does not fit is cut into this many (64 -> 8 -> single queries)
#define KNN_SPLIT 8u
#ifndef KNN2_FRONT
#define KNN2_FRONT 512
#endif
#ifdef RT_KNN_PROBE
__device__ unsigned long long knn_probe[9];          // cycles (lane 0 of every wave): walk, radius check, search, ties, sum, [5] bound; [6] waves, [7] passes, [8] candidates
#define KNN_PROBE(k) { if (lane == 0) { long long t1_ = clock64(); atomicAdd(&knn_probe[k], (unsigned long long)(t1_ - pt0)); pt0 = t1_; } }
#else
#define KNN_PROBE(k)
#endif

__device__ __forceinline__ double wave_min_d(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ double wave_max_d(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask, int lane) { return (uint32_t)__popcll(mask & (((unsigned long long)1 << lane) - 1)); }

// Work units: a unit is `count` consecutive queries starting at `first` (count = 64 or 8), served by one wave
// with the lanes beyond count shadowing its queries.  Without a unit list workgroup u serves [64 u, 64 u + 64).  A unit
// whose candidates do not fit the wave's LDS buffers is cut into eight (appended to `next` for the next launch: other
// waves serve them, in parallel) or, a single query, goes to the list of the per-lane heap kernel.  why[] counts the
// units given up by cause: candidates, frontier, second try short, search.
struct KnnQueues { uint2 *next; uint32_t *next_count; uint2 *single; uint32_t *single_count; uint32_t *heap_list; uint32_t *heap_count; uint32_t *why; };
// to_single: the unit is so far beyond the buffers that its eighths would not fit either -- its queries go straight to
// the single-query kernel's queue
__device__ __forceinline__ void knn_give_up(const KnnQueues &Q, uint32_t first, uint32_t count, uint32_t nq, int lane, int cause, bool to_single = false) {
  if (lane != 0) return;
  atomicAdd(&Q.why[cause], 1u);
  if (count <= 1) { Q.heap_list[atomicAdd(Q.heap_count, 1u)] = first; return; }
  const uint32_t part = to_single ? 1u : count / KNN_SPLIT, parts = count / part;
  uint2 *dst = part == 1 ? Q.single : Q.next;
  uint32_t *dst_count = part == 1 ? Q.single_count : Q.next_count;
  uint32_t live = 0;
  for (uint32_t j = 0; j < parts; j++) live += (first + j * part < nq) ? 1u : 0u;
  const uint32_t base = atomicAdd(dst_count, live);
  for (uint32_t j = 0, o = 0; j < parts; j++) if (first + j * part < nq) dst[base + o++] = make_uint2(first + j * part, part);
}

// number of nodes in the subtree rooted at heap position pos of a left-balanced tree of n nodes
__device__ __forceinline__ uint32_t kd_subtree_size(uint32_t pos, uint32_t n) {
  unsigned long long lo = pos, hi = pos; uint32_t c = 0;
  while (lo <= n) { c += (uint32_t)((hi < n ? hi : (unsigned long long)n) - lo + 1); lo = 2 * lo; hi = 2 * hi + 1; }
  return c;
}
__device__ __forceinline__ float wave_min_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_max_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }

// Descent towards the point c, t_depth levels down (the host picks the depth at which every subtree has k .. 4k photons),
// six levels per global round trip: the 63 nodes of the next six levels under the current node are fetched at once, one
// a lane, into `top` (64 nodes of LDS).  Returns the heap position reached, the split axes (two bits a level) and, in
// lane L, the split coordinate of level L.
__device__ __forceinline__ void knn_descend(const DevMap &M, float cx, float cy, float cz, uint32_t t_depth, float4 *top, int lane,
                                            uint32_t &t_pos, unsigned long long &path, float &plane_val) {
  const uint32_t n = M.n;
  int depth = 0;
  while (depth < (int)t_depth) {
    const int L = min(6, (int)t_depth - depth);
    if (lane >= 1 && lane < (1 << L)) {
      const int j = 31 - __clz(lane);
      const unsigned long long abs_pos = ((unsigned long long)t_pos << j) + (unsigned)(lane - (1 << j));
      if (abs_pos <= n) top[lane] = M.node[abs_pos - 1];
    }
    __syncthreads();
    uint32_t r = 1;
    for (int sdx = 0; sdx < L; sdx++, depth++) {
      const float4 nd = top[r];
      const int dim = node_dim(nd);
      // Which side?  By the split axis; where the point lies exactly on the plane (photons on an axis-aligned wall
      // share that coordinate, and the median falls inside the run), by what ordered the run: the stable sorts
      // of the levels above, i.e. the parent's axis, then the grandparent's, ...
      int go = -1;
      for (int j = depth; j >= 0 && go < 0; j--) {
        const int dd = j == depth ? dim : (int)((path >> (2 * j)) & 3u);
        const float cv = (dd == 0) ? cx : (dd == 1) ? cy : cz, nv = (dd == 0) ? nd.x : (dd == 1) ? nd.y : nd.z;
        if (cv < nv) go = 0; else if (cv > nv) go = 1;
      }
      if (go < 0) go = 0;
      path |= (unsigned long long)(dim & 3) << (2 * depth);
      if (lane == depth) plane_val = (dim == 0) ? nd.x : (dim == 1) ? nd.y : nd.z;
      r = 2 * r + (uint32_t)go;
    }
    t_pos = (t_pos << L) + (r - (1u << L));
    __syncthreads();
  }
}

// the photons of the subtree at t_pos into cand[] (level by level: contiguous heap positions); their number
__device__ __forceinline__ uint32_t knn_load_subtree(const DevMap &M, uint32_t t_pos, float4 *cand, int lane, uint32_t cap = KNN2_CAP) {
  const uint32_t n = M.n;
  uint32_t cnt = 0;
  unsigned long long lo = t_pos, hi = t_pos;
  while (lo <= n && cnt < cap) {
    const unsigned long long end = hi < n ? hi : (unsigned long long)n;
    for (unsigned long long p = lo + lane; p <= end && cnt + (uint32_t)(p - lo) < cap; p += 64) {
      float4 nd = M.node[p - 1]; nd.w = __uint_as_float((uint32_t)(p - 1));
      cand[cnt + (uint32_t)(p - lo)] = nd;
    }
    cnt += (uint32_t)(end - lo + 1);
    lo = 2 * lo; hi = 2 * hi + 1;
  }
  return cnt > cap ? cap : cnt;                                            // (never cut: the host's depth keeps subtrees within the buffer)
}

// CAP: candidates a wave can hold.  (Tried: units of 64 whose candidates do not fit KNN2_CAP again as a whole with a
// buffer of 2048 before they are cut up -- a wave's passes grow with its candidates and few such waves fit a CU: the
// eighths serve a query in 7 ns, that launch took 12.)
// A count that is the same in every lane, as a scalar: a loop up to it keeps its counter and its addresses out of the
// vector unit and can be unrolled (the compiler cannot see that a value computed from shuffles or LDS is uniform).
__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
#define RT_UNROLL_4 _Pragma("unroll 4")
#define RT_UNROLL_2 _Pragma("unroll 2")
template <int CAP>
__global__ void __launch_bounds__(64) k_gather_knn_wave(DevMap M, const float *hits, uint32_t nq, uint32_t k, float radius_scale, uint32_t t_depth, float *rgb,
                                                        const uint2 *units, const uint32_t *n_units, KnnQueues Q) {
  __shared__ float4 cand[CAP];                       // x y z, slot (as bits)
  __shared__ uint32_t front[KNN2_FRONT];             // the walk's queue of tree nodes (a ring)
  const int lane = threadIdx.x;
  uint32_t u_first = blockIdx.x * 64u, u_count = 64u;
  if (units) {
    if (blockIdx.x >= *n_units) return;              // (the grid is an upper bound)
    const uint2 u = units[blockIdx.x];
    u_first = u.x; u_count = u.y;
  }
  const uint32_t q = u_first + (uint32_t)lane % u_count;
  const bool valid = (uint32_t)lane < u_count && q < nq;
  const uint32_t qq = q < nq ? q : (u_first < nq ? u_first : nq - 1);     // surplus lanes shadow a query (no output)
  const float *h = hits + (size_t)qq * 9;
  const float qx = h[0], qy = h[1], qz = h[2];
  const double x0 = qx, x1 = qy, x2 = qz;
  const float nx = h[3], ny = h[4], nz = h[5];
  const uint32_t n = M.n;
#ifdef RT_KNN_PROBE
  long long pt0 = clock64();
#endif
  // ---- 0. where to look, from the tree itself.  Descend towards the centre of the queries' box to a subtree T of
  // k .. 4k photons.  First try: the local density of T's photons (their number over the two largest extents of their
  // box: photons lie on surfaces) gives the radius a query inside that box needs, 1.4 x for slack; a query outside it
  // first has to reach it.  If some query then has fewer than k candidates inside its ball, second try: the k-th
  // smallest distance from the query to T's photons, which bounds its k-th neighbour distance from above wherever
  // the query lies (fp32 with a margin: any upper bound will do).
  uint32_t m = 0, t_pos = 1;
  float plane_val = 0.f;                             // split coordinates along the descent: lane L holds level L's
  unsigned long long path = 0;                       // split axes along the descent, two bits a level
  knn_descend(M, 0.5f * (wave_min_f(qx) + wave_max_f(qx)), 0.5f * (wave_min_f(qy) + wave_max_f(qy)), 0.5f * (wave_min_f(qz) + wave_max_f(qz)),
              t_depth, (float4 *)&front[0], lane, t_pos, path, plane_val);
  auto load_subtree = [&]() { return knn_load_subtree(M, t_pos, cand, lane, (uint32_t)CAP < KNN2_CAP ? (uint32_t)CAP : KNN2_CAP); };
  const uint32_t mt = load_subtree();
  __syncthreads();
  float rq;                                          // radius of this query's search ball
  float rho;                                         // photons per unit area around the queries (from T)
  {
    float l0 = 3.4e38f, l1 = 3.4e38f, l2 = 3.4e38f, h0 = -3.4e38f, h1 = -3.4e38f, h2 = -3.4e38f;
    for (uint32_t j = lane; j < mt; j += 64) { const float4 c = cand[j]; l0 = fminf(l0, c.x); h0 = fmaxf(h0, c.x); l1 = fminf(l1, c.y); h1 = fmaxf(h1, c.y); l2 = fminf(l2, c.z); h2 = fmaxf(h2, c.z); }
    l0 = wave_min_f(l0); l1 = wave_min_f(l1); l2 = wave_min_f(l2); h0 = wave_max_f(h0); h1 = wave_max_f(h1); h2 = wave_max_f(h2);
    const float e0 = h0 - l0, e1 = h1 - l1, e2 = h2 - l2;
    const float emax = fmaxf(e0, fmaxf(e1, e2)), emin = fminf(e0, fminf(e1, e2)), emid = e0 + e1 + e2 - emax - emin;
    const float area = fmaxf(emax * fmaxf(emid, 1e-3f * emax), 1e-30f);
    rho = (float)mt / area;
    const float r_est = KNN_FIRST_TRY * sqrtf((float)k * area / (3.14159265f * (float)mt));
    const float dx = fmaxf(0.f, fmaxf(l0 - qx, qx - h0)), dy = fmaxf(0.f, fmaxf(l1 - qy, qy - h1)), dz = fmaxf(0.f, fmaxf(l2 - qz, qz - h2));
    rq = (r_est + sqrtf(dx * dx + dy * dy + dz * dz)) * radius_scale;
  }
  KNN_PROBE(5)
  bool overflow = false, ok = false, why_front = false;
  uint32_t c = 0, pc1 = 0, pc2 = 0, pc3 = 0;
  double b64 = 0.0, pt1 = 0.0, pt2 = 0.0, pt3 = 0.0;
  bool fast = false;                                 // the first pass settled `ok` in fp32: the fp32 search below may run
  float fbl = 0.f, fp1 = 0.f, fp2 = 0.f, fp3 = 0.f; uint32_t fc = 0, fc1 = 0, fc2 = 0, fc3 = 0;      // its threshold and pivots, and the counts below them
  for (int attempt = 0; attempt < 2 && !ok; attempt++) {
    overflow = false;                                // (a first try that found too many candidates was too wide: the second may fit)
    if (attempt == 1) {                              // the guaranteed radius: k-th smallest distance to T's photons
      __syncthreads();
      (void)load_subtree();
      __syncthreads();
      auto d2f = [&](float4 cd) { const float dx = qx - cd.x, dy = qy - cd.y, dz = qz - cd.z; return dx * dx + dy * dy + dz * dz; };
      float dmax = 0.f;
      RT_UNROLL_4
      for (uint32_t j = 0, n_ = wave_uniform(mt); j < n_; j++) dmax = fmaxf(dmax, d2f(cand[j]));
      float a = 0.f, b = dmax * 1.000002f + 1e-30f; float fa = 0.f, fb = (float)mt;      // #(d2 < a) = fa < k <= fb = #(d2 < b)
      const float kf = (float)k;
      for (int it = 0; it < 3 && __any(fb > kf * 1.06f + 1.f); it++) {
        // three pivots around the value where the count would reach a little over k if it grew linearly over the bracket
        const float f2 = fminf(0.9f, fmaxf(0.02f, (kf * 1.03f + 1.f - fa) / (fb - fa)));
        const float t2 = a + (b - a) * f2, t1 = a + (b - a) * f2 * 0.8f, t3 = a + (b - a) * fminf(0.95f, f2 * 1.3f);
        float c1 = 0.f, c2 = 0.f, c3 = 0.f;
        RT_UNROLL_4
        for (uint32_t j = 0, n_ = wave_uniform(mt); j < n_; j++) { const float d = d2f(cand[j]); c1 += d < t1 ? 1.f : 0.f; c2 += d < t2 ? 1.f : 0.f; c3 += d < t3 ? 1.f : 0.f; }
        if (c1 >= kf) { b = t1; fb = c1; } else if (c2 >= kf) { b = t2; fb = c2; a = t1; fa = c1; } else if (c3 >= kf) { b = t3; fb = c3; a = t2; fa = c2; } else { a = t3; fa = c3; }
      }
      rq = sqrtf(b * 1.00001f) * 1.000001f * radius_scale;                   // (margins for the fp32 distance arithmetic: about 1e-6 relative)
    }
    b64 = (double)rq * (double)rq;
    // pivots of the first counting pass: a first-try radius is 1.4 x the expected k-th distance; a guaranteed one is looser
    pt2 = attempt == 0 ? b64 / (double)(KNN_FIRST_TRY * KNN_FIRST_TRY) : 0.55 * b64;
    { const double pdt = pt2 / (double)k * (1.0 + 0.6 * sqrt((double)k)); pt1 = pt2 - pdt > 0.0 ? pt2 - pdt : 0.5 * pt2; pt3 = pt2 + pdt < b64 ? pt2 + pdt : 0.5 * (pt2 + b64); }
    // ---- 1. cooperative walk of the tree over the box of the queries' balls (fp32 bounds rounded outwards: the walk
    // only has to find a superset, the distances below decide)
    const float rw = rq * 1.000001f;
    const float lo0 = wave_min_f(__double2float_rd((double)qx - (double)rw)), lo1 = wave_min_f(__double2float_rd((double)qy - (double)rw)),
                lo2 = wave_min_f(__double2float_rd((double)qz - (double)rw));
    const float hi0 = wave_max_f(__double2float_ru((double)qx + (double)rw)), hi1 = wave_max_f(__double2float_ru((double)qy + (double)rw)),
                hi2 = wave_max_f(__double2float_ru((double)qz + (double)rw));
    // ... and, of that box, only what lies within reach of the ball around the queries (cuts the box's corners)
    const float gx = 0.5f * (wave_min_f(qx) + wave_max_f(qx)), gy = 0.5f * (wave_min_f(qy) + wave_max_f(qy)), gz = 0.5f * (wave_min_f(qz) + wave_max_f(qz));
    const float reach = (sqrtf(wave_max_f((qx - gx) * (qx - gx) + (qy - gy) * (qy - gy) + (qz - gz) * (qz - gz))) + wave_max_f(rw)) * 1.00001f;
    const float reach2 = reach * reach * 1.00001f;
    // How many photons will that box hold?  Its two largest extents times the density around the queries.  Far beyond
    // the buffer: no point in walking -- the unit is cut up at once, straight into single queries when its eighths
    // would not fit either.
    if (u_count > 1) {
      const float b0 = hi0 - lo0, b1 = hi1 - lo1, b2 = hi2 - lo2;
      const float bmax = fmaxf(b0, fmaxf(b1, b2)), bmin = fminf(b0, fminf(b1, b2));
      const float predicted = rho * bmax * (b0 + b1 + b2 - bmax - bmin);
      if (predicted > 3.0f * CAP) { knn_give_up(Q, u_first, u_count, nq, lane, 0, predicted > 6.0f * CAP); return; }
    }
    __syncthreads();                                 // (T has been consumed)
    m = 0;
    {
      // start at the deepest node of the descent whose subtree certainly holds every photon of the box: at every level
      // above it the box lay strictly on the side the descent took (photons ON a plane can sit on either side)
      uint32_t start_level = 0;
      for (; start_level < t_depth; start_level++) {
        const int dd = (int)((path >> (2 * start_level)) & 3u);
        const float v = __shfl(plane_val, (int)start_level, 64);
        const bool went_right = (t_pos >> (t_depth - 1 - start_level)) & 1u;
        const float blo = (dd == 0) ? lo0 : (dd == 1) ? lo1 : lo2, bhi = (dd == 0) ? hi0 : (dd == 1) ? hi1 : hi2;
        if (!(went_right ? blo > v : bhi < v)) break;
      }
      if (lane == 0) front[0] = t_pos >> (t_depth - start_level);
      __syncthreads();
      // the queue is a ring: up to 64 nodes leave at the head, their children join at the tail (the order of the visits
      // does not matter: the distances decide, and ties go by input index).  Once the ring is more than half full the
      // nodes leave at the TAIL instead -- the deepest ones, which soon end in leaves -- and only as many as leave room
      // for two children each, so a wide level of the tree does not overflow the ring.
      uint32_t head = 0, tail = 1;
      while (head != tail) {
        const uint32_t size = tail - head;
        const bool lifo = size > KNN2_FRONT / 2;
        const uint32_t take = lifo ? min(64u, max(1u, (KNN2_FRONT - size) / 2u)) : min(64u, size);
        const bool active = (uint32_t)lane < take;
        const uint32_t pos = active ? front[((lifo ? tail - take : head) + lane) % KNN2_FRONT] : 1u;      // 1 <= pos <= n by construction
        if (lifo) tail -= take; else head += take;
        const float4 nd = M.node[pos - 1];
        const float ddx = nd.x - gx, ddy = nd.y - gy, ddz = nd.z - gz;
        const bool inside = active && nd.x >= lo0 && nd.x <= hi0 && nd.y >= lo1 && nd.y <= hi1 && nd.z >= lo2 && nd.z <= hi2 &&
                            ddx * ddx + ddy * ddy + ddz * ddz <= reach2;
        const unsigned long long mi = __ballot(inside);
        const uint32_t ci = (uint32_t)__popcll(mi);
        if (m + ci > (uint32_t)CAP) { overflow = true; why_front = false; break; }
        if (inside) cand[m + lanes_below(mi, lane)] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(pos - 1));
        m += ci;
        const int dim = node_dim(nd);
        const float pd = (dim == 0) ? nd.x : (dim == 1) ? nd.y : nd.z;
        const float lod = (dim == 0) ? lo0 : (dim == 1) ? lo1 : lo2, hid = (dim == 0) ? hi0 : (dim == 1) ? hi1 : hi2;
        // kdtree.js:69-71 for a box: left only if the box ends before the plane, right only if it starts after it
        const bool gl = active && 2ull * pos <= n && !(lod > pd);
        const bool gr = active && 2ull * pos + 1 <= n && !(hid < pd);
        const unsigned long long ml = __ballot(gl), mr = __ballot(gr);
        const uint32_t cl = (uint32_t)__popcll(ml), cr = (uint32_t)__popcll(mr);
        if (tail - head + cl + cr > KNN2_FRONT) { overflow = true; why_front = true; break; }
        if (gl) front[(tail + lanes_below(ml, lane)) % KNN2_FRONT] = 2 * pos;
        if (gr) front[(tail + cl + lanes_below(mr, lane)) % KNN2_FRONT] = 2 * pos + 1;
        tail += cl + cr;
        __syncthreads();
      }
    }
    __syncthreads();
    if (overflow) continue;
    KNN_PROBE(0)
    // ---- 2. first counting pass: candidates inside the query's own ball and at three pivots.  In fp32: the squared
    // distance of two fp32 points evaluated in fp32 is within 3e-7 (relative) of the fp64 one, so a candidate whose fp32
    // distance is below (1 - 1e-6) r^2 IS inside the ball and one above (1 + 1e-6) r^2 is not.  If the candidates that
    // certainly are inside number k or more for every query, the k nearest of every query are among the candidates;
    // if even the possible ones fall short for some query, they are not; in between the fp64 pass decides.
    {
      const float bl = __double2float_rd(b64 * (1.0 - 1e-6)), bh = __double2float_ru(b64 * (1.0 + 1e-6));
      const float p1f = (float)pt1, p2f = (float)pt2, p3f = (float)pt3;
      uint32_t c_lo = 0, c_hi = 0, f1 = 0, f2 = 0, f3 = 0;
      RT_UNROLL_4
      for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) {
        const float4 cd = cand[j];
        const float dx = qx - cd.x, dy = qy - cd.y, dz = qz - cd.z;
        const float d = dx * dx + dy * dy + dz * dz;
        c_lo += d < bl ? 1u : 0u; c_hi += d <= bh ? 1u : 0u; f1 += d < p1f ? 1u : 0u; f2 += d < p2f ? 1u : 0u; f3 += d < p3f ? 1u : 0u;
      }
      fast = __all(c_lo >= k);
      fbl = bl; fc = c_lo; fp1 = p1f; fp2 = p2f; fp3 = p3f; fc1 = f1; fc2 = f2; fc3 = f3;
      if (fast) ok = true;
      else if (__any(c_hi < k)) ok = false;
      else {
        c = 0;
        RT_UNROLL_2
        for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) {
          const float4 cd = cand[j];
          c += sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z) < b64 ? 1u : 0u;
        }
        ok = __all(c >= k);                          // then every query's k nearest photons are among the candidates, exactly
      }
    }
    KNN_PROBE(1)
  }
  if (!ok) { knn_give_up(Q, u_first, u_count, nq, lane, overflow ? (why_front ? 1 : 0) : 2); return; }
  // ---- 3 (fp32). The same search as the fp64 one below on the fp32 distances, ending with the k-th smallest fp32
  // distance (lmaxf) and the next one above it (rminf).  If those two lie more than 1e-6 (relative) apart -- the fp32
  // error is 3e-7 -- the k candidates up to lmaxf ARE the k nearest in fp64 too, and step 4 (fp32) sums them.  A wave
  // with a lane where they do not (1.7 % of the waves at k = 64: a gap of 1e-6 where neighbours are 1 / k apart) or where
  // fp32 distances tie at the k-th place runs the exact search instead.
  float sr = 0, sg = 0, sb = 0; double dk2 = 0.0;
  float *pay_lds = (float *)&front[0];                        // 64 x 6 floats; the frontier buffers are free now
  static_assert(KNN2_FRONT * sizeof(uint32_t) >= 64 * 6 * sizeof(float), "payload staging needs 1536 bytes");
  bool exact = !fast;
  if (fast) {
    auto d2f = [&](float4 cd) { const float dx = qx - cd.x, dy = qy - cd.y, dz = qz - cd.z; return dx * dx + dy * dy + dz * dz; };
    const uint32_t kk = k;
    float af = 0.f, bf = fbl; uint32_t fa = 0, fb = fc;          // #(d < af) = fa < k <= fb = #(d < bf)
    float t_hit = -1.f;                                          // >= 0: a value with exactly k candidates below it is known; next: the tracking pass there
    if (fb == kk) t_hit = bf;
    else if (fc1 == kk) t_hit = fp1;
    else if (fc2 == kk) t_hit = fp2;
    else if (fc3 == kk) t_hit = fp3;
    else {
      if (fc3 < kk) { af = fp3; fa = fc3; } else if (fc2 < kk) { af = fp2; fa = fc2; } else if (fc1 < kk) { af = fp1; fa = fc1; }
      if (fc1 > kk) { bf = fp1; fb = fc1; } else if (fc2 > kk) { bf = fp2; fb = fc2; } else if (fc3 > kk) { bf = fp3; fb = fc3; }
    }
    bool donef = false, failf = false; float lmaxf = 0.f, rminf = 0.f;
    for (int pass = 0; pass < 64 && !__all(donef || failf); pass++) {
      const float span = (float)fb - (float)fa;                  // candidates in [af, bf), > 0
      const bool narrow = t_hit >= 0.f || span <= 6.f || pass >= 5;
      float t = (pass % 3 == 2) ? 0.5f * (af + bf) : af + (bf - af) * (((float)kk + 0.5f - (float)fa) / span);
      if (!(t > af && t < bf)) t = 0.5f * (af + bf);
      if (__any(!(donef || failf) && !narrow)) {                 // counting pass, pivots t1 < t < t3
        const float dt = (bf - af) / span * (1.0f + 0.6f * sqrtf(span));
        float t1 = t - dt, t3 = t + dt;
        if (!(t1 > af)) t1 = 0.5f * (af + t);
        if (!(t3 < bf)) t3 = 0.5f * (t + bf);
        uint32_t c1 = 0, c2 = 0, c3 = 0;
        RT_UNROLL_4
        for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) { const float d = d2f(cand[j]); c1 += d < t1 ? 1u : 0u; c2 += d < t ? 1u : 0u; c3 += d < t3 ? 1u : 0u; }
        if (donef || failf || t_hit >= 0.f || !(t1 > af && t1 < t && t < t3 && t3 < bf)) continue;
        if (c1 == kk) t_hit = t1; else if (c2 == kk) t_hit = t; else if (c3 == kk) t_hit = t3;
        else {
          if (c3 < kk) { af = t3; fa = c3; } else if (c2 < kk) { af = t; fa = c2; } else if (c1 < kk) { af = t1; fa = c1; }
          if (c1 > kk) { bf = t1; fb = c1; } else if (c2 > kk) { bf = t; fb = c2; } else if (c3 > kk) { bf = t3; fb = c3; }
        }
        continue;
      }
      // tracking pass: the count below t and the candidate values next to t on either side
      if (t_hit >= 0.f) t = t_hit; else if (!(t > af && t < bf)) t = bf;
      uint32_t cnt = 0, ties = 0, rties = 0; float lmax = -1.f, rmin = 3.4e38f;
      RT_UNROLL_4
      for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) {
        const float d = d2f(cand[j]);
        const bool lt = d < t, up = lt && d > lmax, eq = lt && d == lmax, dn = !lt && d < rmin, req = !lt && d == rmin;
        cnt += lt ? 1u : 0u;
        ties = up ? 1u : ties + (eq ? 1u : 0u); lmax = up ? d : lmax;
        rties = dn ? 1u : rties + (req ? 1u : 0u); rmin = dn ? d : rmin;
      }
      if (donef || failf) continue;
      t_hit = -1.f;
      if (cnt == kk) { donef = true; lmaxf = lmax; rminf = rmin; }
      else if (cnt < kk) {
        const float above = __uint_as_float(__float_as_uint(rmin) + 1u);    // next float after rmin (>= 0, finite while candidates remain)
        if (!(rmin < 3.0e38f)) failf = true;                       // (never: fb >= k candidates lie below bf)
        else if (cnt + rties == kk) t_hit = above;                 // d < above  <=>  d <= rmin: k below it
        else if (cnt + rties > kk) failf = true;                   // fp32 distances tie at the k-th place
        else { af = above; fa = cnt + rties; }
      }
      else if (cnt - ties == kk) t_hit = lmax;                     // #(d < lmax) = cnt - ties
      else if (cnt - ties < kk) failf = true;
      else { bf = lmax; fb = cnt - ties; }
    }
    exact = __any(!(donef && rminf > lmaxf * 1.000001f));
    KNN_PROBE(2)
    if (!exact) {
#ifdef RT_KNN_PROBE
      if (lane == 0) { atomicAdd(&knn_probe[6], 1ull); atomicAdd(&knn_probe[8], (unsigned long long)m); }
#endif
      // ---- 4 (fp32). radiance of the k candidates up to lmaxf; the k-th squared distance itself (the estimate's area) in fp64
      uint32_t j_kth = 0, n_kth = 0;
      for (uint32_t base = 0; base < m; base += 64) {
        __syncthreads();                                           // the previous chunk has been consumed
        if (base + lane < m) {
          const uint32_t slot = __float_as_uint(cand[base + lane].w);
          const float2 *pp = (const float2 *)(M.pay + (size_t)slot * 6);      // 24-byte records: 8-byte aligned
          const float2 p0 = pp[0], p1 = pp[1], p2 = pp[2];
          float *dst = pay_lds + lane * 6;
          dst[0] = p0.x; dst[1] = p0.y; dst[2] = p1.x; dst[3] = p1.y; dst[4] = p2.x; dst[5] = p2.y;
        }
        __syncthreads();
        const uint32_t nc = wave_uniform(min(64u, m - base));
        for (uint32_t jj = 0; jj < nc; jj++) {
          const float4 cd = cand[base + jj];
          const float d = d2f(cd);
          const float *pl = pay_lds + jj * 6;
          const float w = d <= lmaxf ? fmaxf(0.0f, pl[0] * nx + pl[1] * ny + pl[2] * nz) : 0.0f;
          sr += pl[3] * w; sg += pl[4] * w; sb += pl[5] * w;
          const bool kth = d == lmaxf;                            // the k-th nearest (or, with an fp32 tie below the next value, several)
          j_kth = kth ? base + jj : j_kth; n_kth += kth ? 1u : 0u;
        }
      }
      // the k-th squared distance in fp64: of the one candidate at lmaxf, or the largest of several
      if (__any(n_kth != 1u)) {
        RT_UNROLL_2
        for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) {
          const float4 cd = cand[j];
          if (d2f(cd) == lmaxf) dk2 = fmax(dk2, sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z));
        }
      } else {
        const float4 cd = cand[j_kth < m ? j_kth : 0u];
        dk2 = sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z);
      }
      KNN_PROBE(4)
    }
  }
  if (exact) {
  if (fast) {                                                    // (the fp32 pass stood in for the exact first counting pass: here it is)
    c = 0;
    RT_UNROLL_2
    for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) { const float4 cd = cand[j]; c += sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z) < b64 ? 1u : 0u; }
  }
  pc1 = 0; pc2 = 0; pc3 = 0;
  RT_UNROLL_2
  for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) {
    const float4 cd = cand[j];
    const double d2 = sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z);
    pc1 += d2 < pt1 ? 1u : 0u; pc2 += d2 < pt2 ? 1u : 0u; pc3 += d2 < pt3 ? 1u : 0u;
  }
  // ---- 3. per lane: t with exactly k candidates closer than t (or a tie group at the k-th distance)
  double a = 0.0, b = b64; uint32_t fa = 0, fb = c;             // #(d2 < a) = fa < k,  #(d2 < b) = fb >= k
  bool done = (fb == k); double tau = b; bool tie = false; double tie_v = 0.0; uint32_t need = 0;
  if (!done) {                                                   // bracket from the first pass's pivots (0 < pt1 < pt2 < pt3 < b)
    if (pc1 == k) { done = true; tau = pt1; }
    else if (pc2 == k) { done = true; tau = pt2; }
    else if (pc3 == k) { done = true; tau = pt3; }
    else {
      if (pc3 < k) { a = pt3; fa = pc3; } else if (pc2 < k) { a = pt2; fa = pc2; } else if (pc1 < k) { a = pt1; fa = pc1; }
      if (pc1 > k) { b = pt1; fb = pc1; } else if (pc2 > k) { b = pt2; fb = pc2; } else if (pc3 > k) { b = pt3; fb = pc3; }
    }
  }
  // The count of candidates closer than t grows almost linearly with t (photons lie on surfaces), so a few
  // interpolation steps bracket the k-th distance.  While some lane's bracket still holds many candidates a pass only
  // counts, at three pivots around the interpolated guess; once every bracket is narrow the passes also track the
  // candidate values next to the pivot, which is what settles the exact k-th value and ties.
  int pass = 0;
  for (; pass < 200 && !__all(done); pass++) {
    const double span = (double)fb - (double)fa;                  // candidates in [a, b), > 0
    const bool narrow = span <= 6.0 || pass >= 5;
    double t = (pass % 3 == 2) ? 0.5 * (a + b) : a + (b - a) * (((double)k + 0.5 - (double)fa) / span);
    if (!(t > a && t < b)) t = 0.5 * (a + b);
    if (__any(!done && !narrow)) {
      // ---- counting pass, pivots t1 < t < t3
      const double dt = (b - a) / span * (1.0 + 0.6 * sqrt(span));
      double t1 = t - dt, t3 = t + dt;
      if (!(t1 > a)) t1 = 0.5 * (a + t);
      if (!(t3 < b)) t3 = 0.5 * (t + b);
      uint32_t c1 = 0, c2 = 0, c3 = 0;
      RT_UNROLL_2
      for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) {
        const float4 cd = cand[j];
        const double d2 = sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z);
        c1 += d2 < t1 ? 1u : 0u; c2 += d2 < t ? 1u : 0u; c3 += d2 < t3 ? 1u : 0u;
      }
      if (done || !(t1 > a && t1 < t && t < t3 && t3 < b)) continue;       // (a degenerate bracket waits for the tracking passes)
      if (c1 == k) { done = true; tau = t1; }
      else if (c2 == k) { done = true; tau = t; }
      else if (c3 == k) { done = true; tau = t3; }
      else {
        if (c3 < k) { a = t3; fa = c3; } else if (c2 < k) { a = t; fa = c2; } else if (c1 < k) { a = t1; fa = c1; }
        if (c1 > k) { b = t1; fb = c1; } else if (c2 > k) { b = t; fb = c2; } else if (c3 > k) { b = t3; fb = c3; }
      }
      continue;
    }
    // ---- tracking pass
    if (!(t > a && t < b)) t = b;                                // a and b are neighbouring doubles: count below b, ties sit at a
    uint32_t cnt = 0, ties = 0, rties = 0; double lmax = -1.0, rmin = __builtin_huge_val();
    RT_UNROLL_2
    for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) {
      const float4 cd = cand[j];
      const double d2 = sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z);
      const bool lt = d2 < t, up = lt && d2 > lmax, eq = lt && d2 == lmax;
      const bool dn = !lt && d2 < rmin, req = !lt && d2 == rmin;
      cnt += lt ? 1u : 0u;
      ties = up ? 1u : ties + (eq ? 1u : 0u);
      lmax = up ? d2 : lmax;
      rties = dn ? 1u : rties + (req ? 1u : 0u);
      rmin = dn ? d2 : rmin;
    }
    if (done) continue;
    // lmax (x ties) is the largest candidate value below t, rmin (x rties) the smallest at or above it
    if (cnt == k) { done = true; tau = t; }
    else if (cnt < k) {
      const double above = __longlong_as_double(__double_as_longlong(rmin) + 1);      // next double after rmin (>= 0, finite)
      if (cnt + rties == k) { done = true; tau = above; }                               // d2 < above  <=>  d2 <= rmin
      else if (cnt + rties > k) { done = true; tie = true; tie_v = rmin; need = k - cnt; }
      else { a = above; fa = cnt + rties; }
    }
    else if (cnt - ties == k) { done = true; tau = lmax; }       // #(d2 < lmax) = cnt - ties
    else if (cnt - ties < k) { done = true; tie = true; tie_v = lmax; need = k - (cnt - ties); }
    else { b = lmax; fb = cnt - ties; }
  }
  if (!__all(done)) { knn_give_up(Q, u_first, u_count, nq, lane, 3); return; }
  KNN_PROBE(2)
#ifdef RT_KNN_PROBE
  if (lane == 0) { atomicAdd(&knn_probe[6], 1ull); atomicAdd(&knn_probe[7], (unsigned long long)pass); atomicAdd(&knn_probe[8], (unsigned long long)m); }
#endif
  // ties at the k-th distance: the `need` lowest input indices among the candidates at distance tie_v
  int32_t idx_thr = -1;
  if (__any(tie)) {
    uint32_t left = tie ? need : 0;
    for (int guard = 0; guard < CAP && __any(left > 0); guard++) {
      int32_t best = 0x7FFFFFFF;
      RT_UNROLL_2
      for (uint32_t j = 0, n_ = wave_uniform(m); j < n_; j++) {
        const float4 cd = cand[j];
        const double d2 = sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z);
        const int32_t pi = M.perm[__float_as_uint(cd.w)];
        if (d2 == tie_v && pi > idx_thr && pi < best) best = pi;
      }
      if (left > 0) { idx_thr = best; left--; }
    }
  }
  KNN_PROBE(3)
  // ---- 4. radiance of the selected candidates
  // The candidates' payloads (direction, power * colour) come in through LDS, 64 at a time: one scattered load per
  // lane and chunk instead of one dependent global load per candidate.
  for (uint32_t base = 0; base < m; base += 64) {
    __syncthreads();                                             // the previous chunk has been consumed
    if (base + lane < m) {
      const uint32_t slot = __float_as_uint(cand[base + lane].w);
      const float2 *pp = (const float2 *)(M.pay + (size_t)slot * 6);      // 24-byte records: 8-byte aligned
      const float2 p0 = pp[0], p1 = pp[1], p2 = pp[2];
      float *dst = pay_lds + lane * 6;
      dst[0] = p0.x; dst[1] = p0.y; dst[2] = p1.x; dst[3] = p1.y; dst[4] = p2.x; dst[5] = p2.y;
    }
    __syncthreads();
    const uint32_t nc = wave_uniform(min(64u, m - base));
    for (uint32_t jj = 0; jj < nc; jj++) {
      const float4 cd = cand[base + jj];
      const double d2 = sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z);
      bool sel;
      if (tie) sel = d2 < tie_v || (d2 == tie_v && M.perm[__float_as_uint(cd.w)] <= idx_thr);
      else sel = d2 < tau;
      const float *pl = pay_lds + jj * 6;
      const float w = fmaxf(0.0f, pl[0] * nx + pl[1] * ny + pl[2] * nz);
      if (sel) { sr += pl[3] * w; sg += pl[4] * w; sb += pl[5] * w; dk2 = fmax(dk2, d2); }
    }
  }
  KNN_PROBE(4)
  }                                                              // (exact)
  if (valid) {
    const double r2 = dk2;                                       // squared distance of the k-th nearest
    const float inv_area = (float)(1.0 / (3.141592653589793 * r2));
    rgb[(size_t)q * 3] = sr * (h[6] / 255.0f) * inv_area;
    rgb[(size_t)q * 3 + 1] = sg * (h[7] / 255.0f) * inv_area;
    rgb[(size_t)q * 3 + 2] = sb * (h[8] / 255.0f) * inv_area;
  }
}

// ---- one wave per query: the path for queries whose neighbourhood does not fit the buffers above in one piece
// The 64 lanes walk the tree for ONE query, nearest side first (a stack of nodes in LDS, up to 64 popped per step),
// keeping every photon within tau of the query in the candidate buffer.  tau starts as the k-th smallest distance to the
// photons of the query's subtree (an upper bound of the k-th neighbour distance) and, whenever the buffer fills, drops
// to the k-th smallest distance among the candidates, which are then cut down to those within it -- so any number of
// photons inside the first ball can be streamed through a 512-entry buffer.  Distances that decide are the fp64
// sq_dist; the fp32 tests of the walk only have to keep a superset.  Ties at the k-th distance go to the lowest
// input index, as everywhere.  A query whose stack would overflow goes to the heap kernel's list.
// (the candidate buffer of this kernel: CAP1 entries, 256 where the subtrees of the descent fit that -- 4 k <= 256 -- which
// lets six waves share a SIMD's LDS instead of four; the walk waits on one global round trip per step)
#ifdef RT_KNN_PROBE
__device__ unsigned long long knn1_probe[8];         // cycles: start, walk, refills, final; [4] waves, [5] refills, [6] walk steps, [7] candidates taken
#define KNN1_PROBE(k) { long long t1_ = clock64(); pacc[k] += (unsigned long long)(t1_ - pt0); pt0 = t1_; }
#define KNN1_COUNT(k, v) { pacc[k] += (unsigned long long)(v); }
#define KNN1_FLUSH() { if (lane == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&knn1_probe[i_], pacc[i_]); }
#else
#define KNN1_PROBE(k)
#define KNN1_COUNT(k, v)
#define KNN1_FLUSH()
#endif
// k-th smallest (k >= 1) of the wave's 64 x PER_LANE non-negative doubles (absent entries: +infinity), bit by bit
// from the top: the largest v with fewer than k entries below it
template <int PER_LANE> __device__ __forceinline__ double wave_kth_smallest(const double (&d)[PER_LANE], uint32_t k) {
  unsigned long long ans = 0;
  for (int bit = 62; bit >= 0; bit--) {
    const unsigned long long trial = ans | (1ull << bit);
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < PER_LANE; i++) c += (uint32_t)__popcll(__ballot((unsigned long long)__double_as_longlong(d[i]) < trial));
    if (c < k) ans = trial;
  }
  return __longlong_as_double((long long)ans);
}

#ifndef KNN1_CELLS
#define KNN1_CELLS 1                                  // 1: the walk's stack entries carry the query's distance to their cell (see the walk)
#endif
#ifndef KNN1_FRONT
#define KNN1_FRONT 512                                // entries of the single-query walk's stack
// (Tried: 1024 and 2048 entries, so that a step can always take 64 nodes -- 94 -> 71 walk steps per query on the emitted
// caustic map, but fewer waves fit a CU: 226 and 189 against 229 M gathers/s.  The walk visits some 2000 nodes for 400
// candidates when a child is pruned by its distance to the splitting plane alone; with the distance to the child's cell
// (KNN1_CELLS) 41 steps, and 241 M gathers/s with the offsets packed in one word per entry -- 235 with three halves in two.)
#endif
template <int CAP1>
__global__ void __launch_bounds__(64) k_gather_knn_single(DevMap M, const float *hits, uint32_t nq, uint32_t k, uint32_t t_depth, float *rgb,
                                                          const uint2 *units, const uint32_t *n_units, uint32_t *heap_list, uint32_t *heap_count) {
  constexpr int KNN1_PER_LANE = CAP1 / 64;
  __shared__ float4 cand[CAP1];                      // x y z, slot (as bits)
  __shared__ uint32_t stack[KNN1_FRONT];
#if KNN1_CELLS
  // per stack entry: how far the query lies outside the entry's cell along x, y, z, in 1023ths of the first search radius,
  // rounded down (10 bits each)
  __shared__ uint32_t soff[KNN1_FRONT];
#endif
  const int lane = threadIdx.x;
  if (blockIdx.x >= *n_units) return;                // (the grid is an upper bound)
  const uint32_t q = units[blockIdx.x].x;
  if (q >= nq) return;
  const float *h = hits + (size_t)q * 9;
  const float qx = h[0], qy = h[1], qz = h[2];
  const double x0 = qx, x1 = qy, x2 = qz;
  const uint32_t n = M.n;
#ifdef RT_KNN_PROBE
  long long pt0 = clock64(); unsigned long long pacc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
  uint32_t t_pos = 1; float plane_val = 0.f; unsigned long long path = 0;
  static_assert(KNN1_FRONT * sizeof(uint32_t) >= 64 * sizeof(float4), "descent staging needs 1 KB");
  knn_descend(M, qx, qy, qz, t_depth, (float4 *)&stack[0], lane, t_pos, path, plane_val);
  uint32_t m = knn_load_subtree(M, t_pos, cand, lane, CAP1);
  __syncthreads();
  double d[KNN1_PER_LANE];
  auto load_d = [&](uint32_t cnt) {
#pragma unroll
    for (int i = 0; i < KNN1_PER_LANE; i++) {
      const uint32_t j = (uint32_t)i * 64u + (uint32_t)lane;
      const float4 cd = cand[j < cnt ? j : 0];
      d[i] = j < cnt ? sq_dist(x0, x1, x2, (double)cd.x, (double)cd.y, (double)cd.z) : __builtin_huge_val();
    }
  };
  load_d(m);
  double tau = m >= k ? wave_kth_smallest(d, k) : __builtin_huge_val();      // (the host's depth gives every subtree k photons)
  float tauf = 0.f, reach = 0.f;
  auto set_tau = [&]() {                              // fp32 versions, rounded up, with room for the fp32 arithmetic of the walk
    tauf = tau < 3.0e38 ? __double2float_ru(tau) * 1.000002f + 1e-37f : 3.4e38f;
    reach = tau < 3.0e38 ? sqrtf(tauf) * 1.000001f : 3.4e38f;
  };
  set_tau();
  // start at the deepest node of the descent whose subtree certainly holds the whole ball
  uint32_t start_level = 0;
  for (; start_level < t_depth; start_level++) {
    const int dd = (int)((path >> (2 * start_level)) & 3u);
    const float v = __shfl(plane_val, (int)start_level, 64);
    const bool went_right = (t_pos >> (t_depth - 1 - start_level)) & 1u;
    const float qd = (dd == 0) ? qx : (dd == 1) ? qy : qz;
    if (!(went_right ? qd - reach > v : qd + reach < v)) break;
  }
  __syncthreads();                                   // (the subtree's photons have been consumed)
  m = 0;
#if KNN1_CELLS
  const float off_step = reach < 3.0e38f ? reach * (1.0f / 1023.0f) : 0.f, off_inv = off_step > 0.f ? 1.0f / off_step : 0.f;
#endif
  if (lane == 0) {
    stack[0] = t_pos >> (t_depth - start_level);
#if KNN1_CELLS
    soff[0] = 0u;                                    // (the ball lies inside that node's cell)
#endif
  }
  __syncthreads();
  uint32_t sp = 1; bool failed = false;
  KNN1_PROBE(0)
  while (sp > 0) {
    if (m + 64 > CAP1) {                              // no room for another step: lower tau to the k-th smallest so far
      KNN1_PROBE(1)
      load_d(m);
      tau = wave_kth_smallest(d, k);
      set_tau();
      // in place, 64 entries at a time: what a chunk keeps lands at or below the chunk's own positions, so the chunks
      // still to come are not overwritten
      uint32_t m2 = 0;
#pragma unroll
      for (int i = 0; i < KNN1_PER_LANE; i++) {
        const uint32_t j = (uint32_t)i * 64u + (uint32_t)lane;
        const float4 entry = cand[j < m ? j : 0];
        const bool kp = d[i] <= tau;                  // (absent entries are +infinity)
        const unsigned long long mk = __ballot(kp);
        __syncthreads();                              // (every lane holds its entry)
        if (kp) cand[m2 + lanes_below(mk, lane)] = entry;
        m2 += (uint32_t)__popcll(mk);
        __syncthreads();
      }
      m = m2;
      KNN1_PROBE(2) KNN1_COUNT(5, 1)
      if (m + 64 > CAP1) { failed = true; break; }                           // (hundreds of photons at one and the same distance)
    }
    // (the fuller the stack, the fewer nodes a step takes off its top: those are the deepest, which soon end in leaves,
    // and a step can add at most one entry per node it takes)
    const uint32_t take = min(min(64u, sp), max(1u, (KNN1_FRONT - sp) / 8u));
    const bool active = (uint32_t)lane < take;
    const uint32_t pos = active ? stack[sp - take + lane] : 1u;              // 1 <= pos <= n by construction
#if KNN1_CELLS
    // The cell of a node bounds its whole subtree: a node whose cell has fallen out of the ball (the ball shrinks when the
    // buffer is refilled) is dropped with its subtree, and a far child is pushed only if ITS cell reaches the ball -- the
    // distance to the splitting plane alone keeps whole slabs of the tree alive.  The offsets are lower bounds (rounded
    // down), tauf has room for the fp32 arithmetic.
    const uint32_t so = active ? soff[sp - take + lane] : 0u;
    const float o0 = (float)(so & 1023u) * off_step, o1 = (float)((so >> 10) & 1023u) * off_step, o2 = (float)(so >> 20) * off_step;
    const bool live = active && o0 * o0 + o1 * o1 + o2 * o2 <= tauf;
#else
    const bool live = active;
#endif
    sp -= take;
    const float4 nd = M.node[pos - 1];
    const float ddx = qx - nd.x, ddy = qy - nd.y, ddz = qz - nd.z;
    const bool inside = live && ddx * ddx + ddy * ddy + ddz * ddz <= tauf;
    const unsigned long long mi = __ballot(inside);
    if (inside) cand[m + lanes_below(mi, lane)] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(pos - 1));
    m += (uint32_t)__popcll(mi);
    KNN1_COUNT(6, 1) KNN1_COUNT(7, __popcll(mi))
    const int dim = node_dim(nd);
    const float delta = ((dim == 0) ? ddx : (dim == 1) ? ddy : ddz);        // query minus plane
#if KNN1_CELLS
    const float others = (dim == 0) ? o1 * o1 + o2 * o2 : (dim == 1) ? o0 * o0 + o2 * o2 : o0 * o0 + o1 * o1;
    const bool reachable = others + delta * delta <= tauf;                   // the far child's cell
    const uint32_t hd = (uint32_t)fminf(fabsf(delta) * off_inv, 1023.f);     // (truncates: a lower bound)
    const uint32_t so_far = dim == 0 ? (so & ~1023u) | hd : dim == 1 ? (so & ~(1023u << 10)) | (hd << 10) : (so & ~(1023u << 20)) | (hd << 20);
#else
    const bool reachable = delta * delta <= tauf;
#endif
    const bool gl = live && 2ull * pos <= n && (delta <= 0.f || reachable);
    const bool gr = live && 2ull * pos + 1 <= n && (delta >= 0.f || reachable);
    // far sides first, near sides on top of them
    const bool far_l = gl && delta > 0.f, far_r = gr && delta < 0.f, near_l = gl && !(delta > 0.f), near_r = gr && !(delta < 0.f);
    const unsigned long long b0 = __ballot(far_l), b1 = __ballot(far_r), b2 = __ballot(near_l), b3 = __ballot(near_r);
    const uint32_t c0 = (uint32_t)__popcll(b0), c1 = (uint32_t)__popcll(b1), c2 = (uint32_t)__popcll(b2), c3 = (uint32_t)__popcll(b3);
    if (sp + c0 + c1 + c2 + c3 > KNN1_FRONT) { failed = true; break; }
    __syncthreads();                                 // (every lane has taken its node)
#if KNN1_CELLS
    if (far_l) { const uint32_t at = sp + lanes_below(b0, lane); stack[at] = 2 * pos; soff[at] = so_far; }
    if (far_r) { const uint32_t at = sp + c0 + lanes_below(b1, lane); stack[at] = 2 * pos + 1; soff[at] = so_far; }
    if (near_l) { const uint32_t at = sp + c0 + c1 + lanes_below(b2, lane); stack[at] = 2 * pos; soff[at] = so; }
    if (near_r) { const uint32_t at = sp + c0 + c1 + c2 + lanes_below(b3, lane); stack[at] = 2 * pos + 1; soff[at] = so; }
#else
    if (far_l) stack[sp + lanes_below(b0, lane)] = 2 * pos;
    if (far_r) stack[sp + c0 + lanes_below(b1, lane)] = 2 * pos + 1;
    if (near_l) stack[sp + c0 + c1 + lanes_below(b2, lane)] = 2 * pos;
    if (near_r) stack[sp + c0 + c1 + c2 + lanes_below(b3, lane)] = 2 * pos + 1;
#endif
    sp += c0 + c1 + c2 + c3;
    __syncthreads();
  }
  if (failed || m < k) {                             // (m < k cannot happen on a map of k photons; the heap kernel copes with anything)
    if (lane == 0) heap_list[atomicAdd(heap_count, 1u)] = q;
    return;
  }
  // ---- the k nearest among the candidates
  KNN1_PROBE(1)
  __syncthreads();
  load_d(m);
  const double vk = wave_kth_smallest(d, k);        // the k-th neighbour's squared distance
  uint32_t below = 0, at = 0;
#pragma unroll
  for (int i = 0; i < KNN1_PER_LANE; i++) { below += (uint32_t)__popcll(__ballot(d[i] < vk)); at += (uint32_t)__popcll(__ballot(d[i] == vk)); }
  int32_t idx_thr = 0x7FFFFFFF;                      // ties at vk: those with input index <= idx_thr are taken
  if (below + at > k) {
    int32_t pi[KNN1_PER_LANE];
#pragma unroll
    for (int i = 0; i < KNN1_PER_LANE; i++) pi[i] = d[i] == vk ? M.perm[__float_as_uint(cand[(uint32_t)i * 64u + (uint32_t)lane].w)] : 0x7FFFFFFF;
    idx_thr = -1;
    for (uint32_t left = k - below; left > 0; left--) {
      int32_t best = 0x7FFFFFFF;
#pragma unroll
      for (int i = 0; i < KNN1_PER_LANE; i++) if (pi[i] > idx_thr && pi[i] < best) best = pi[i];
      for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
      idx_thr = best;
    }
  }
  const float nx = h[3], ny = h[4], nz = h[5];
  float sr = 0, sg = 0, sb = 0;
#pragma unroll
  for (int i = 0; i < KNN1_PER_LANE; i++) {
    const uint32_t j = (uint32_t)i * 64u + (uint32_t)lane;
    if (j < m && d[i] <= vk) {
      const uint32_t slot = __float_as_uint(cand[j].w);
      if (d[i] < vk || idx_thr == 0x7FFFFFFF || M.perm[slot] <= idx_thr) {
        const float2 *pp = (const float2 *)(M.pay + (size_t)slot * 6);
        const float2 p0 = pp[0], p1 = pp[1], p2 = pp[2];
        const float w = fmaxf(0.0f, p0.x * nx + p0.y * ny + p1.x * nz);
        sr += p1.y * w; sg += p2.x * w; sb += p2.y * w;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) { sr += __shfl_xor(sr, o, 64); sg += __shfl_xor(sg, o, 64); sb += __shfl_xor(sb, o, 64); }
  if (lane == 0) {
    const float inv_area = (float)(1.0 / (3.141592653589793 * vk));
    rgb[(size_t)q * 3] = sr * (h[6] / 255.0f) * inv_area;
    rgb[(size_t)q * 3 + 1] = sg * (h[7] / 255.0f) * inv_area;
    rgb[(size_t)q * 3 + 2] = sb * (h[8] / 255.0f) * inv_area;
  }
  KNN1_PROBE(3) KNN1_COUNT(4, 1) KNN1_FLUSH()
}

// ---- scratch of the k-nearest gather: counters, the unit queues of the two split levels (8- and 1-query
// units) and the list of queries left to the heap kernel.  rt_gather_radiance keeps it in a grow-only buffer of the map;
// rt_gather_radiance_static uses the array below, which exists once per GPU for the life of the process.
static size_t knn_scratch_bytes(uint32_t nq) {
  const size_t blocks = div_up(nq, KNN_LANES);
  return 256 + blocks * (8 + 64) * sizeof(uint2) + (size_t)nq * 4;
}
#define RT_STATIC_SCRATCH_BYTES ((256 + ((size_t)RT_GATHER_STATIC_MAX_QUERIES / KNN_LANES) * 72 * 8 + (size_t)RT_GATHER_STATIC_MAX_QUERIES * 4 + 255) / 256 * 256)
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];
// The array's name is only a host-side stand-in; the address of each GPU's copy is looked up once per device
// (rt_static_scratch_setup, called by rt_init with that device current) and host code uses that pointer from then on.
#define RT_MAX_DEVICES 64
static void *g_scratch_ptr[RT_MAX_DEVICES];
static std::mutex g_scratch_lock;                   // (two host threads may call rt_init for one device at the same time)

int rt_static_scratch_setup(rt_ctx *ctx) {
  if (ctx->device < 0 || ctx->device >= RT_MAX_DEVICES) return rt_fail(ctx, RT_ERR_NO_DEVICE, "device ordinal %d beyond %d", ctx->device, RT_MAX_DEVICES);
  std::lock_guard<std::mutex> guard(g_scratch_lock);
  if (g_scratch_ptr[ctx->device]) return RT_OK;
  void *ptr = nullptr;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
  if (!ptr) return rt_fail(ctx, RT_ERR_HIP, "hipGetSymbolAddress returned no address for the static scratch array");
  g_scratch_ptr[ctx->device] = ptr;
  return RT_OK;
}

// The k-nearest gather over device arrays, with its scratch in `ws` (knn_scratch_bytes(nq) bytes of device memory; the
// counters at its start are cleared here, everything else that is read has been written by an earlier kernel of the call).
static int gather_knn_run(rt_photon_map *m, const float *dh, uint32_t nq, uint32_t k, float *dout, void *ws) {
  rt_ctx *ctx = m->ctx;
  const RtKnobs &K = ctx->knobs;
  const uint32_t blocks = div_up(nq, KNN_LANES);
  const bool heap_all = m->n < k || m->n < 2 * KNN2_CAP || K.knn_heap;     // small maps: nothing to gain
  if (heap_all) {
    RT_HIP(ctx, hipFuncSetAttribute((const void *)k_gather_knn, hipFuncAttributeMaxDynamicSharedMemorySize, knn_lds_bytes(k)));
    KernelTimer timer_(ctx, "gather_knn");
    hipLaunchKernelGGL(k_gather_knn, dim3(blocks), dim3(KNN_LANES), knn_lds_bytes(k), ctx->stream, m->view(),
                       dh, nq, k, dout, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
  }
  // (test switch: the search balls of every unit scaled up, which drives units into the split levels and the heap kernel)
  const float scale = K.knn_radius_scale > 1 ? (float)K.knn_radius_scale : 1.0f;
  // depth at which the wave kernel takes its reference subtree: the first one where every subtree has at most 4k photons
  // (and what fits the candidate buffer); the smallest subtree of that depth then still has k
  auto subtree_size = [&](uint64_t pos) { uint64_t lo = pos, hi = pos, c = 0; while (lo <= m->n) { c += std::min<uint64_t>(hi, m->n) - lo + 1; lo *= 2; hi = 2 * hi + 1; } return c; };
  uint32_t t_depth = 0;
  while (t_depth < 30 && subtree_size((uint64_t)1 << t_depth) > std::min<uint64_t>(4ull * k, KNN2_CAP)) t_depth++;
  while (t_depth > 0 && subtree_size(((uint64_t)2 << t_depth) - 1) < k) t_depth--;
  // Level 0: one wave per 64 consecutive queries.  Level 1: the units whose candidates did not fit, cut into eight (one
  // row of an 8x8 tile each).  Level 2: the queries of the eighths that did not fit either, one wave each, streaming.
  // What is left then goes to the per-lane heap kernel.
  uint32_t *counters = (uint32_t *)ws;                                     // [0] eighths queued, [1] single queries queued, [3] queries for the heap kernel, [4..7] units given up by cause
  uint2 *queue[2] = { (uint2 *)((char *)ws + 256), nullptr };
  queue[1] = queue[0] + (size_t)blocks * 8;
  uint32_t *heap_list = (uint32_t *)(queue[1] + (size_t)blocks * 64);
  RT_HIP(ctx, hipMemsetAsync(counters, 0, 32, ctx->stream));
  uint32_t launched[3] = { blocks, 0, 0 };
  auto read_counters = [&]() -> int {
    RT_HIP(ctx, hipMemcpyAsync(ctx->pinned, counters, 32, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
  };
  KnnQueues Q; Q.single = queue[1]; Q.single_count = counters + 1; Q.heap_list = heap_list; Q.heap_count = counters + 3; Q.why = counters + 4;
  Q.next = queue[0]; Q.next_count = counters;
  RT_LAUNCH(ctx, "gather_knn", k_gather_knn_wave<KNN2_CAP>, blocks, 64, m->view(), dh, nq, k, scale, t_depth, dout, (const uint2 *)nullptr, (const uint32_t *)nullptr, Q);
  RT_TRY(read_counters());
  // One host wait, after the first launch.  The later launches' grids are upper bounds from what is known then (an
  // eighth leaves at most eight single queries, a single query at most itself for the heap kernel); their kernels read
  // the true counts on the device and surplus workgroups exit at once.
  launched[1] = std::min(ctx->pinned[0], blocks * KNN_SPLIT);
  if (launched[1]) {
    Q.next = queue[1]; Q.next_count = counters + 1;
    RT_LAUNCH(ctx, "gather_knn_split", k_gather_knn_wave<KNN2_CAP>, launched[1], 64, m->view(), dh, nq, k, scale, t_depth, dout, (const uint2 *)queue[0], (const uint32_t *)counters, Q);
  }
  launched[2] = std::min(ctx->pinned[1] + launched[1] * KNN_SPLIT, nq);
  if (launched[2]) {                                                       // single queries: one wave each, streaming
    // (Tried: a 128-entry buffer for k <= 64, so that the search ball shrinks after the first hundred candidates instead of
    // the first two hundred -- 113 against 229 M gathers/s on the emitted caustic map: the extra selections cost more than
    // the shorter walks save; a 512-entry one: 183 M/s; one extra selection once 96 or 128 candidates are in: 234 / 235
    // against 240 M/s.)
    if (4 * k <= 256) {
      RT_LAUNCH(ctx, "gather_knn_single", k_gather_knn_single<256>, launched[2], 64, m->view(), dh, nq, k, t_depth, dout,
                (const uint2 *)queue[1], (const uint32_t *)(counters + 1), heap_list, counters + 3);
    } else {
      RT_LAUNCH(ctx, "gather_knn_single", k_gather_knn_single<KNN2_CAP>, launched[2], 64, m->view(), dh, nq, k, t_depth, dout,
                (const uint2 *)queue[1], (const uint32_t *)(counters + 1), heap_list, counters + 3);
    }
  }
  const uint32_t left = std::min(ctx->pinned[3] + launched[2], nq);       // (upper bound)
  if (left > 0) {
    RT_HIP(ctx, hipFuncSetAttribute((const void *)k_gather_knn, hipFuncAttributeMaxDynamicSharedMemorySize, knn_lds_bytes(k)));
    KernelTimer timer_(ctx, "gather_knn_heap_fallback");
    hipLaunchKernelGGL(k_gather_knn, dim3(div_up(left, KNN_LANES)), dim3(KNN_LANES), knn_lds_bytes(k), ctx->stream, m->view(),
                       dh, nq, k, dout, (const uint32_t *)heap_list, (const uint32_t *)(counters + 3));
  }
  if (K.knn_verbose) RT_TRY(read_counters());
  RT_HIP(ctx, hipGetLastError());
#ifdef RT_KNN_PROBE
  if (K.knn_verbose) {
    unsigned long long hp[9], zero[9] = { 0 };
    RT_HIP(ctx, hipMemcpyFromSymbol(hp, HIP_SYMBOL(knn_probe), sizeof hp));
    RT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(knn_probe), zero, sizeof zero));
    const double w = hp[6] ? (double)hp[6] : 1.0;
    fprintf(stderr, "knn probe per wave: bound %.0f walk %.0f check %.0f search %.0f ties %.0f sum %.0f cycles; passes %.2f candidates %.1f (%llu waves)\n",
            hp[5] / w, hp[0] / w, hp[1] / w, hp[2] / w, hp[3] / w, hp[4] / w, hp[7] / w, hp[8] / w, hp[6]);
    unsigned long long h1[8], zero1[8] = { 0 };
    RT_HIP(ctx, hipMemcpyFromSymbol(h1, HIP_SYMBOL(knn1_probe), sizeof h1));
    RT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(knn1_probe), zero1, sizeof zero1));
    const double w1 = h1[4] ? (double)h1[4] : 1.0;
    fprintf(stderr, "single-query probe per wave: start %.0f walk %.0f refills %.0f final %.0f cycles; refills %.2f, walk steps %.1f, candidates taken %.1f (%llu waves)\n",
            h1[0] / w1, h1[1] / w1, h1[2] / w1, h1[3] / w1, h1[5] / w1, h1[6] / w1, h1[7] / w1, h1[4]);
  }
#endif
  if (K.knn_verbose)
    fprintf(stderr, "rt_gather_radiance: %u units of 64 queries, then %u of 8, %u of 1; given up for candidates %u, frontier %u, short %u, search %u; %u queries left to the heap kernel\n",
            launched[0], ctx->pinned[0], ctx->pinned[1], ctx->pinned[4], ctx->pinned[5], ctx->pinned[6], ctx->pinned[7], ctx->pinned[3]);
  return RT_OK;
}

// ---- putting the queries of a k-nearest gather in the order of their positions
// The wave kernel serves 64 consecutive queries with one walk over the box of their search balls: that pays while they
// are neighbours (a caller that orders hit points by 8x8 image tile gets there by itself) and falls apart when they are
// not -- row by row, 64 consecutive pixels are a strip whose box holds several times the photons of its balls, and
// nearly every unit ends up cut into eighths.  So rt_gather_radiance looks at the queries first: the mean diagonal of
// the box around 64 consecutive query points against the distance at which a query finds its k-th photon at the map's
// mean density; beyond 1.5 of those the queries are sorted by the Morton code of their positions (10 bits an axis over
// the map's bounding box), gathered in that order and the results put back where the caller wants them.
// (every `stride`-th group of 64 is looked at: a few thousand atomics on one address, not tens of thousands)
__global__ void __launch_bounds__(256) k_knn_spread(const float *hits, uint32_t nq, uint32_t stride, float *sum) {
  const uint32_t q = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64u * stride + (threadIdx.x & 63u);
  const bool have = q < nq;
  const float *h = hits + (size_t)(have ? q : 0) * 9;
  const float x = h[0], y = h[1], z = h[2], inf = 3.4e38f;
  const float l0 = wave_min_f(have ? x : inf), l1 = wave_min_f(have ? y : inf), l2 = wave_min_f(have ? z : inf);
  const float h0 = wave_max_f(have ? x : -inf), h1 = wave_max_f(have ? y : -inf), h2 = wave_max_f(have ? z : -inf);
  if ((threadIdx.x & 63) == 0 && have) {
    const float d = sqrtf((h0 - l0) * (h0 - l0) + (h1 - l1) * (h1 - l1) + (h2 - l2) * (h2 - l2));
    if (d == d && d < 1e30f) { atomicAdd(sum, d); atomicAdd(sum + 1, 1.0f); }
  }
}
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
  v &= 1023u; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ void __launch_bounds__(256) k_knn_keys(const float *hits, uint32_t nq, float lo0, float lo1, float lo2, float inv_cell, uint32_t *keys, uint32_t *vals) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const float *h = hits + (size_t)q * 9;
  auto cell = [&](float v, float lo) { const float c = (v - lo) * inv_cell; return (uint32_t)(c < 0.f ? 0.f : (c > 1023.f ? 1023.f : c)); };   // (NaN -> 0)
  keys[q] = spread10(cell(h[0], lo0)) | (spread10(cell(h[1], lo1)) << 1) | (spread10(cell(h[2], lo2)) << 2);
  vals[q] = q;
}
// dst[i] = src[order[i]] (GATHER) or dst[order[i]] = src[i] (!GATHER), rows of W floats
template <int W, bool GATHER> __global__ void __launch_bounds__(256) k_knn_reorder(const float *src, const uint32_t *order, uint32_t nq, float *dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const uint32_t o = order[i];
  if (o >= nq) return;                                       // (never true)
  const float *a = src + (size_t)(GATHER ? o : i) * W;
  float *b = dst + (size_t)(GATHER ? i : o) * W;
#pragma unroll
  for (int c = 0; c < W; c++) b[c] = a[c];
}
static int rt_sort_pairs_u32(rt_ctx *ctx, DevMem &temp, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, int bits) {
  if (n == 0) return RT_OK;
  size_t bytes = 0;
  RT_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, bits, ctx->stream));
  RT_HIP(ctx, temp.reserve(bytes ? bytes : 16));
  RT_HIP(ctx, rocprim::radix_sort_pairs(temp.p, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, bits, ctx->stream));
  return RT_OK;
}

// The box of the map's core: the nodes of the tree's top levels are the medians of ever smaller regions -- a sample of the
// photons spread like the photons themselves -- and the box drops the outermost 2 % of them on either side of every axis.
// (The full bounding box of an emitted map is set by a handful of photons that left the scene through a gap and landed far
// out on a plane.)  Worked out when a k-nearest gather first asks for it, not at every build.
static int map_core_box(rt_photon_map *m) {
  if (m->core_valid) return RT_OK;
  rt_ctx *ctx = m->ctx;
  const uint32_t ns = std::min<uint32_t>(m->n, 4095u);
  std::vector<float4> top(ns);
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RT_HIP(ctx, hipMemcpy(top.data(), m->node.p, (size_t)ns * 16, hipMemcpyDeviceToHost));
  std::vector<float> v(ns);
  for (int d = 0; d < 3; d++) {
    for (uint32_t i = 0; i < ns; i++) v[i] = d == 0 ? top[i].x : (d == 1 ? top[i].y : top[i].z);
    std::sort(v.begin(), v.end());
    const uint32_t cut = ns / 50;
    m->core[d] = v[cut]; m->core[3 + d] = v[ns - 1 - cut];
  }
  m->core_valid = true;
  return RT_OK;
}

// the k-nearest gather of rt_gather_radiance: in the caller's order, or sorted first (see above)
static int gather_knn_ordered(rt_photon_map *m, const float *dh, uint32_t nq, uint32_t k, float *dout) {
  rt_ctx *ctx = m->ctx;
  const RtKnobs &K = ctx->knobs;
  bool sort = K.knn_sort > 0;
  if (K.knn_sort < 0 && nq >= 4096 && m->n >= 2 * KNN2_CAP && m->n >= k && !K.knn_heap) {
    RT_HIP(ctx, m->ws_flags.reserve(knn_scratch_bytes(nq)));
    float *sum = (float *)m->ws_flags.p;
    RT_HIP(ctx, hipMemsetAsync(sum, 0, 8, ctx->stream));
    const uint32_t groups = div_up(nq, 64), stride = std::max(1u, groups / 2048u), sampled = div_up(groups, stride);
    RT_LAUNCH(ctx, "gather_knn_spread", k_knn_spread, div_up(sampled, 4), 256, dh, nq, stride, sum);
    RT_HIP(ctx, hipMemcpyAsync(ctx->pinned, sum, 8, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float total[2]; memcpy(total, ctx->pinned, 8);
    const double mean_diag = total[1] > 0 ? (double)total[0] / (double)total[1] : 0.0;
    // (the map's extent without its stragglers -- a photon that left through a gap in the scene lands far out on a plane:
    // the box of its core, see rt_photon_map::core)
    RT_TRY(map_core_box(m));
    const double e0 = m->core[3] - m->core[0], e1 = m->core[4] - m->core[1], e2 = m->core[5] - m->core[2];
    const double emax = std::max(e0, std::max(e1, e2));
    const double area = std::max(2.0 * (e0 * e1 + e1 * e2 + e2 * e0), 1e-6 * emax * emax);
    const double r_k = sqrt((double)k * area / (3.141592653589793 * (double)m->n));      // photons lie on surfaces: k of them in pi r^2 at the mean density
    sort = mean_diag > 1.5 * r_k;
    if (K.knn_verbose) fprintf(stderr, "rt_gather_radiance: 64 consecutive queries span %.4g on average, a k-th neighbour is about %.4g away: %s\n", mean_diag, r_k, sort ? "sorting the queries" : "taking them as they come");
  }
  if (!sort) {
    RT_HIP(ctx, m->ws_flags.reserve(knn_scratch_bytes(nq)));
    return gather_knn_run(m, dh, nq, k, dout, m->ws_flags.p);
  }
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_kin = carve((size_t)nq * 4), o_kout = carve((size_t)nq * 4), o_vin = carve((size_t)nq * 4), o_vout = carve((size_t)nq * 4),
               o_hits = carve((size_t)nq * 36), o_rgb = carve((size_t)nq * 12);
  RT_HIP(ctx, m->ws_knn_sort.reserve(off));
  char *base = (char *)m->ws_knn_sort.p;
  uint32_t *kin = (uint32_t *)(base + o_kin), *kout = (uint32_t *)(base + o_kout), *vin = (uint32_t *)(base + o_vin), *vout = (uint32_t *)(base + o_vout);
  float *hs = (float *)(base + o_hits), *rs = (float *)(base + o_rgb);
  RT_TRY(map_core_box(m));
  const double emax = std::max((double)m->core[3] - m->core[0], std::max((double)m->core[4] - m->core[1], (double)m->core[5] - m->core[2]));
  const float inv_cell = emax > 0 ? (float)(1024.0 / emax) : 1.0f;      // (queries outside the core box share its outermost cells)
  RT_LAUNCH(ctx, "gather_knn_keys", k_knn_keys, div_up(nq, 256), 256, dh, nq, m->core[0], m->core[1], m->core[2], inv_cell, kin, vin);
  { KernelTimer timer_(ctx, "gather_knn_sort"); RT_TRY(rt_sort_pairs_u32(ctx, m->ws_sort, kin, kout, vin, vout, nq, 30)); }
  RT_LAUNCH(ctx, "gather_knn_reorder", (k_knn_reorder<9, true>), div_up(nq, 256), 256, dh, (const uint32_t *)vout, nq, hs);
  RT_HIP(ctx, m->ws_flags.reserve(knn_scratch_bytes(nq)));
  RT_TRY(gather_knn_run(m, hs, nq, k, rs, m->ws_flags.p));
  RT_LAUNCH(ctx, "gather_knn_reorder", (k_knn_reorder<3, false>), div_up(nq, 256), 256, (const float *)rs, (const uint32_t *)vout, nq, dout);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

extern "C" int rt_gather_radiance(rt_photon_map *m, const float *hits, uint32_t nq, double r, uint32_t k, float *rgb_out) {
  if (!m) return RT_ERR_INVALID;
  rt_ctx *ctx = m->ctx;
  if (!rgb_out || !hits) return rt_fail(ctx, RT_ERR_INVALID, "rt_gather_radiance: hits and rgb_out are required");
  if (k > 192) return rt_fail(ctx, RT_ERR_INVALID, "k must be <= 192, got %u", k);
  if (k == 0 && !(r > 0)) return rt_fail(ctx, RT_ERR_INVALID, "rt_gather_radiance: need k > 0 or r > 0");
  if (nq == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem th, to; const void *dh; void *dout;
  RT_TRY(rt_stage_in(ctx, hits, (size_t)nq * 36, th, &dh));
  RT_TRY(rt_stage_out_begin(ctx, rgb_out, (size_t)nq * 12, to, &dout));
  if (k == 0) {
    // many queries on a large map: through the request queue of the frames' photon gathers (sorted by position, 64
    // served together, subtrees as work items) -- the same photons summed in another order; option defer_gathers
    // as for the frames (1: always, 0: never)
    const int dg = ctx->knobs.defer_gathers;
    const bool queued = nq < (1u << 26) && (dg < 0 ? (m->n >= 200000u && nq >= 16384u) : dg != 0);
    if (queued) RT_TRY(pl_gather_radius_queued(m, m->ws_queue, m->ws_sort, (const float *)dh, nq, r, (float *)dout));
    else RT_LAUNCH(ctx, "gather_radius", k_gather_radius, div_up(nq, 256), 256, m->view(), (const float *)dh, nq, r, (float *)dout);
    RT_HIP(ctx, hipGetLastError());
  } else {
    RT_TRY(gather_knn_ordered(m, (const float *)dh, nq, k, (float *)dout));
  }
  RT_TRY(rt_stage_out_end(ctx, rgb_out, (size_t)nq * 12, to));
  return RT_OK;
}

// rt_gather_radiance for callers that cannot allocate: nothing here calls hipMalloc / hipFree or any allocating
// routine.  Inputs and outputs must already be device memory (staging a host array would allocate); the scratch is
// this GPU's copy of the static array, shared by every call on the GPU -- calls that overlap on one GPU race.
extern "C" int rt_gather_radiance_static(rt_photon_map *m, const float *hits_dev, uint32_t nq, double r, uint32_t k, float *rgb_dev) {
  if (!m) return RT_ERR_INVALID;
  rt_ctx *ctx = m->ctx;
  if (!rgb_dev || !hits_dev) return rt_fail(ctx, RT_ERR_INVALID, "rt_gather_radiance_static: hits_dev and rgb_dev are required");
  if (k > 192) return rt_fail(ctx, RT_ERR_INVALID, "k must be <= 192, got %u", k);
  if (k == 0 && !(r > 0)) return rt_fail(ctx, RT_ERR_INVALID, "rt_gather_radiance_static: need k > 0 or r > 0");
  if (nq > RT_GATHER_STATIC_MAX_QUERIES)
    return rt_fail(ctx, RT_ERR_TOO_LARGE, "rt_gather_radiance_static: %u queries exceed the static scratch's limit of %u; split the call or use rt_gather_radiance",
                   nq, (unsigned)RT_GATHER_STATIC_MAX_QUERIES);
  if (nq == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  if (!rt_is_device_ptr(hits_dev) || !rt_is_device_ptr(rgb_dev))
    return rt_fail(ctx, RT_ERR_INVALID, "rt_gather_radiance_static: hits_dev and rgb_dev must be device memory (a host array would need a staging allocation)");
  for (const void *ptr : { (const void *)hits_dev, (const void *)rgb_dev }) {      // ... of THIS context's GPU: the kernels and the scratch are on it
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) { (void)hipGetLastError(); return rt_fail(ctx, RT_ERR_INVALID, "rt_gather_radiance_static: cannot identify a device array"); }
    if (attr.type == hipMemoryTypeDevice && attr.device != ctx->device)
      return rt_fail(ctx, RT_ERR_INVALID, "rt_gather_radiance_static: an array lives on device %d, the context runs on device %d", attr.device, ctx->device);
  }
  void *ws = g_scratch_ptr[ctx->device];
  if (!ws) return rt_fail(ctx, RT_ERR_HIP, "static scratch of device %d was not set up", ctx->device);
  if (knn_scratch_bytes(nq) > RT_STATIC_SCRATCH_BYTES) return rt_fail(ctx, RT_ERR_TOO_LARGE, "static scratch too small");      // (never: the limit above covers it)
  if (k == 0) {
    RT_LAUNCH(ctx, "gather_radius", k_gather_radius, div_up(nq, 256), 256, m->view(), hits_dev, nq, r, rgb_dev);
    RT_HIP(ctx, hipGetLastError());
  } else {
    RT_TRY(gather_knn_run(m, hits_dev, nq, k, rgb_dev, ws));
  }
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

// ------------------------------------------------------------------ adaptive-radius gather
// The search the reference sketches in a commented block (rayphoton.js:1766-1778, :1789-1801): start at r0 and, up to
// `max_attempts` times, triple the radius while fewer than `min_count` photons are inside; a result with fewer than
// `min_keep` photons is dropped.  The estimate is normalised by the area of the radius that was used (:1808, commented).
// One lane per query; every attempt is the reference's radius walk.
__global__ void __launch_bounds__(256) k_gather_adaptive(DevMap M, const float *hits, uint32_t nq, double r0, double grow, uint32_t min_count,
                                                         uint32_t min_keep, uint32_t max_attempts, float *rgb, float *r_used) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const float *h = hits + (size_t)q * 9;
  const double x0 = h[0], x1 = h[1], x2 = h[2];
  const float nx = h[3], ny = h[4], nz = h[5];
  double r = r0, r_last = r0;
  float sr = 0, sg = 0, sb = 0; uint32_t c = 0;
  for (uint32_t attempt = 0; attempt < max_attempts; attempt++) {
    const double r2 = r * r;
    sr = sg = sb = 0; c = 0; r_last = r;
    kd_radius_walk(M, x0, x1, x2, r, [&](uint32_t slot, float4 nd) {
      if (sq_dist(x0, x1, x2, (double)nd.x, (double)nd.y, (double)nd.z) < r2) {
        const float *pp = M.pay + (size_t)slot * 6;
        float w = fmaxf(0.0f, pp[0] * nx + pp[1] * ny + pp[2] * nz);
        sr += pp[3] * w; sg += pp[4] * w; sb += pp[5] * w; c++;
      }
    });
    if (c < min_count) r *= grow; else break;
  }
  if (c < min_keep) { sr = sg = sb = 0; }
  const float inv_area = (float)(1.0 / (3.141592653589793 * r_last * r_last));
  rgb[(size_t)q * 3] = sr * (h[6] / 255.0f) * inv_area;
  rgb[(size_t)q * 3 + 1] = sg * (h[7] / 255.0f) * inv_area;
  rgb[(size_t)q * 3 + 2] = sb * (h[8] / 255.0f) * inv_area;
  if (r_used) r_used[q] = (float)r_last;
}

extern "C" int rt_gather_radiance_adaptive(rt_photon_map *m, const float *hits, uint32_t nq, double r0, double grow, uint32_t min_count,
                                           uint32_t min_keep, uint32_t max_attempts, float *rgb_out, float *r_used_out) {
  if (!m) return RT_ERR_INVALID;
  rt_ctx *ctx = m->ctx;
  if (!(r0 > 0) || !(grow > 1) || max_attempts == 0 || max_attempts > 64) return rt_fail(ctx, RT_ERR_INVALID, "adaptive gather: need r0 > 0, grow > 1, 1 <= max_attempts <= 64");
  if (!hits || !rgb_out) return rt_fail(ctx, RT_ERR_INVALID, "adaptive gather: hits and rgb_out are required");
  if (nq == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem th, to, tr; const void *dh; void *dout, *dr;
  RT_TRY(rt_stage_in(ctx, hits, (size_t)nq * 36, th, &dh));
  RT_TRY(rt_stage_out_begin(ctx, rgb_out, (size_t)nq * 12, to, &dout));
  RT_TRY(rt_stage_out_begin(ctx, r_used_out, (size_t)nq * 4, tr, &dr));
  RT_LAUNCH(ctx, "gather_adaptive", k_gather_adaptive, div_up(nq, 256), 256, m->view(), (const float *)dh, nq, r0, grow, min_count, min_keep,
            max_attempts, (float *)dout, (float *)dr);
  RT_HIP(ctx, hipGetLastError());
  RT_TRY(rt_stage_out_end(ctx, rgb_out, (size_t)nq * 12, to));
  RT_TRY(rt_stage_out_end(ctx, r_used_out, (size_t)nq * 4, tr));
  return RT_OK;
}
