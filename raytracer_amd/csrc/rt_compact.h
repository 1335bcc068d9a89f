// Ordered stream compaction building blocks: per-workgroup counts -> exclusive scan of the counts by
// one workgroup -> scatter.  Order-preserving and free of contended atomics (a single returning
// atomic counter serialises at roughly ten nanoseconds per wave, which dominated the first
// wavefront passes of a frame).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// exclusive scan of n block counts by one workgroup of 1024; total goes to *total.
// item_count (may be null): device-side number of items; when given, n = ceil(*item_count / 256).
// Tiles of 4096 counts: a thread takes four consecutive counts (neighbouring threads read neighbouring addresses), the
// tile's thread sums are scanned with wave shuffles and 16 wave sums, and the running total carries into the next tile.
static __global__ void __launch_bounds__(1024) k_scan_blocks(uint32_t *block_count, uint32_t n, uint32_t *total,
                                                             const uint32_t *item_count = nullptr) {
  __shared__ uint32_t wave_sum[16];
  if (item_count) n = (*item_count + 255u) / 256u;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t tile = 0; tile < n; tile += 4096u) {           // (workgroup-uniform)
    const uint32_t i0 = tile + threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) v[k] = i0 + k < n ? block_count[i0 + k] : 0u;
    const uint32_t s = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = s;                                      // inclusive scan inside the wave
    for (int off = 1; off < 64; off <<= 1) { uint32_t o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) wave_sum[wv] = incl;
    __syncthreads();
    uint32_t base = carry, all = 0;
    for (int w = 0; w < 16; w++) { const uint32_t c = wave_sum[w]; if (w < wv) base += c; all += c; }
    uint32_t run = base + incl - s;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { if (i0 + k < n) block_count[i0 + k] = run; run += v[k]; }
    carry += all;
    __syncthreads();                                        // (the wave sums are read before the next tile overwrites them)
  }
  if (threadIdx.x == 0) *total = carry;
}

// Inside a 256-thread workgroup: publish this wave's keep-mask and the workgroup's keep-count.
__device__ __forceinline__ void block_publish_masks(bool keep, unsigned long long *wave_mask, uint32_t *block_count) {
  __shared__ uint32_t cnt[4];
  unsigned long long mask = __ballot(keep);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { wave_mask[(size_t)blockIdx.x * 4 + wv] = mask; cnt[wv] = (uint32_t)__popcll(mask); }
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

// Unordered variant, for a pass whose successor takes the items in any order: RT_SUBQ sub-queues, each with its own
// counter on its own cache line (counter s at subq_count[16 s]).  Workgroup b appends its kept items, in order, to
// sub-queue s = b / (cap / 256) with ONE returning atomic; a counter sees a few hundred atomics per frame instead of the
// thousands a single one would.  Sub-queue s is subq[s * cap .. s * cap + min(counter, cap)); cap is a multiple of
// 256 with RT_SUBQ * cap / 256 >= the number of workgroups.  Neighbouring workgroups share a sub-queue on purpose:
// paths of one image region tend to live equally long, and a consumer that follows a chunk of paths to their end
// (k_tail) wastes lanes when short and long paths are mixed.  No scan or scatter kernel follows.
// (256-thread workgroups; all threads must call.)
#define RT_SUBQ 64
__device__ __forceinline__ void block_append_subqueue(bool keep, uint32_t item, uint32_t *subq, uint32_t cap, uint32_t *subq_count, uint32_t s) {
  __shared__ uint32_t cnt[4];
  __shared__ uint32_t base_sh;
  const unsigned long long mask = __ballot(keep);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) cnt[wv] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (s >= RT_SUBQ) s = RT_SUBQ - 1;                       // (never taken)
  if (threadIdx.x == 0) {
    const uint32_t total = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    base_sh = total ? atomicAdd(&subq_count[s * 16], total) : 0u;
  }
  __syncthreads();
  if (keep) {
    uint32_t pos = base_sh + (uint32_t)__popcll(mask & (((unsigned long long)1 << lane) - 1));
    for (int w = 0; w < wv; w++) pos += cnt[w];
    if (pos < cap) subq[(size_t)s * cap + pos] = item;      // (always true when cap covers the workgroups that map to s)
  }
}

// The piece of the sub-queues that workgroup blockIdx.x takes, pieces of `chunk` entries being numbered through the
// sub-queues in order (lane s looks at sub-queue s, so n_subq <= 64).  false: there is none -- the grid of a consumer
// is an upper bound, n_items / chunk + n_subq.  The piece is subq[first .. first + n), part of sub-queue sq.
// Every lane of the wave must call; the result is the same in all waves of the workgroup.
// Two steps, so that a kernel can put other loads between the length's load and its first use.
__device__ __forceinline__ uint32_t subq_load_len(const uint32_t *counts, uint32_t n_subq) {
  const uint32_t lane = threadIdx.x & 63;
  return lane < n_subq ? counts[lane * 16] : 0u;
}
__device__ __forceinline__ bool subq_find_chunk(uint32_t len_raw, uint32_t cap, uint32_t chunk, uint32_t &sq, uint32_t &first, uint32_t &n) {
  const int lane = threadIdx.x & 63;
  const uint32_t len = min(len_raw, cap);
  const uint32_t pieces = (len + chunk - 1) / chunk;
  uint32_t incl = pieces;
  for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
  const unsigned long long mine = __ballot(blockIdx.x >= incl - pieces && blockIdx.x < incl);
  if (mine == 0) return false;
  const int s = __ffsll((long long)mine) - 1;
  const uint32_t j = blockIdx.x - __shfl(incl - pieces, s, 64), sq_len = __shfl(len, s, 64);
  sq = (uint32_t)s; first = (uint32_t)s * cap + j * chunk; n = min(chunk, sq_len - j * chunk);
  return true;
}

// next_queue[rank] = queue[i] for every kept i, rank = position among the kept items in queue order
static __global__ void __launch_bounds__(256) k_queue_scatter(const uint32_t *queue, const uint32_t *count_ptr, const unsigned long long *wave_mask,
                                                              const uint32_t *block_offset, uint32_t *next_queue) {
  const uint32_t count = *count_ptr;
  if (blockIdx.x * blockDim.x >= count) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long *m = wave_mask + (size_t)blockIdx.x * 4;
  const unsigned long long mine = m[wv];
  if (i >= count || !((mine >> lane) & 1ull)) return;
  uint32_t rank = block_offset[blockIdx.x] + (uint32_t)__popcll(mine & (((unsigned long long)1 << lane) - 1));
  for (int w = 0; w < wv; w++) rank += (uint32_t)__popcll(m[w]);
  if (rank < count) next_queue[rank] = queue[i];          // (always true: the kept items are a subset of the queue)
}
