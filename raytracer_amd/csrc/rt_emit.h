// Batched photon emission: one lane runs one emission attempt (createPhoton + the emitPhoton
// random walk, rayphoton.js:1326-1344, :1461-1557) on its own Philox stream
// (phase, attempt index), so results do not depend on batch size or on how attempts are
// split over GPUs.  Each attempt stores at most one photon; the dense per-attempt output is
// then compacted in attempt order (ballot + prefix count), which is what lets the host apply
// the reference's list caps exactly.
#pragma once
#include "rt_compact.h"
#include "rt_sample.h"

struct EmitCfg {
  double epsilon;
  int32_t max_depth;
  uint32_t num_diffuse, num_caustic;
  uint64_t seed;
};
// The emission's statistics.  Device code adds to `slots` (wave w: line w % 64, words 0..2 = segments, errors, overflows) --
// a single word that every wave of a launch adds to when it ends costs ~10 ns per wave, one after the other; the host sums the
// lines into the three words in front.  (Byte 64: the exact segment count of rt_emit_many*, see there.)
#define RT_EMIT_CTR_SLOTS 64
struct EmitCounters { unsigned long long segments, errors, overflows, pad_[29]; unsigned long long slots[RT_EMIT_CTR_SLOTS * 16]; };
static_assert(sizeof(EmitCounters) == 256 + RT_EMIT_CTR_SLOTS * 128, "the slots start behind the 256 bytes of single words");
RT_D void emit_tally(EmitCounters *ctr, unsigned long long segments, unsigned long long errors, unsigned long long overflows) {
  unsigned long long *line = &ctr->slots[((blockIdx.x * 4u + (threadIdx.x >> 6)) % RT_EMIT_CTR_SLOTS) * 16];
  if (segments) atomicAdd(line, segments);
  if (errors) atomicAdd(line + 1, errors);
  if (overflows) atomicAdd(line + 2, overflows);
}

#define RT_REC_STRIDE 11

// One segment of emitPhoton (rayphoton.js:1461-1557): photon `ph` (at recursion depth `depth`, inside the materials of
// `st`) is followed to its next hit.  Returns -1 when the photon goes on (ph, st, depth updated), else how it ended:
// 0 nothing stored (left the scene, dropped, or the reference would have thrown), 1 a diffuse photon stored, 2 a caustic
// one -- then its record (hit3, -dir3, power / list cap, rgba) is in r[0..10].
template <typename T, typename RNG>
RT_D int photon_step(const DevScene<T> &S, const ScanSrc<T> &G, const EmitCfg &C, RNG &rng, PhotonState<T> &ph, MatStack &st, int &depth,
                     bool track_only_caustic, T *r, unsigned long long &errors, unsigned long long &overflows) {
  const T eps = T(C.epsilon);
  T t;
  int idx = scan_scene<T, 1>(S, G, ph.o, ph.d, eps, T(0), t);
  if (idx < 0) return 0;                                                     // left the scene
  V3<T> hit = ph.o + t * ph.d;
  // everything about the shape in one batch of loads (from the workgroup's LDS copy where the scene is staged there)
  const HitRec<T> H = load_hit<T>(S, G, idx);
  V3<T> local = shape_colour_g<T>(H.geom, H.colour, H.proc(), hit);
  V3<T> n = shape_normal_g<T>(H.geom, H.type(), H.flags(), hit);
  const T cos1 = -dot(ph.d, n);
  const bool has_mat = H.mat_byte() < 8;
  const int mat = H.mat_byte() & 7;
  if (has_mat && H.transparent == T(1) && depth < C.max_depth) {             // :1484
    V3<T> ndir; bool ovf = false;
    if (!fresnel_bounce<T>(S, mat, ph.d, n, cos1, st, rng, ndir, ovf)) { errors++; return 0; }   // :1487-1490 throws
    if (ovf) overflows++;
    ph.o = hit; ph.d = normalize<T>(ndir, eps);
    if (ph.caustic < 0) ph.caustic = 1;
    depth++;
    return -1;
  }
  // opaque, or transparent at depth >= MAX_DEPTH where reflectance is undefined (never "< U")
  // (H.refl_k is 0 without a material and for the transparent ones, whose reflectance is undefined)
  if (rng.next_k() < H.refl_k) {                                             // :1523
    if (depth < C.max_depth) {
      if (rng.next_k() < H.spec_k) {                                         // :1525-1533 specular
        ph.o = hit; ph.d = normalize<T>(ph.d + (T(2) * cos1) * n, eps);
        if (ph.caustic < 0) ph.caustic = 1;
        depth++;
        return -1;
      } else if (!track_only_caustic) {                                      // :1536-1546 diffuse
        V3<T> ndir = vec_perturb<T>(n, eps, rng);
        T adj = ph.power * T(255 + 255 + 255 + 1) / (local.x + local.y + local.z + T(1));
        ph.r = ph.r / T(255) * local.x; ph.g = ph.g / T(255) * local.y; ph.b = ph.b / T(255) * local.z; ph.a = T(1);
        ph.power = adj;
        ph.o = hit; ph.d = normalize<T>(ndir, eps); ph.caustic = 0;
        depth++;
        return -1;
      }
    }
    return 0;                                                                // dropped
  }
  if (depth > 0) {                                                           // absorbed: store (:1549-1553)
    const bool is_c = ph.caustic == 1;
    T power = ph.power / (is_c ? T(C.num_caustic) : T(C.num_diffuse));
    V3<T> back = normalize<T>(T(-1) * ph.d, eps);
    r[0] = hit.x; r[1] = hit.y; r[2] = hit.z; r[3] = back.x; r[4] = back.y; r[5] = back.z;
    r[6] = power; r[7] = ph.r; r[8] = ph.g; r[9] = ph.b; r[10] = ph.a;
    return is_c ? 2 : 1;
  }
  return 0;
}
// the whole walk of one photon
template <typename T>
RT_D int photon_walk(const DevScene<T> &S, const ScanSrc<T> &G, const EmitCfg &C, RngStream &rng, PhotonState<T> &ph, MatStack &st, int depth,
                     bool track_only_caustic, T *r, unsigned long long &segments, unsigned long long &errors, unsigned long long &overflows) {
  for (;;) {
    segments++;
    const int k = photon_step<T>(S, G, C, rng, ph, st, depth, track_only_caustic, r, errors, overflows);
    if (k >= 0) return k;
  }
}

// Persistent waves, each with its own contiguous share of the batch: wave w runs attempts [w * per_wave, (w + 1) * per_wave).
// A lane runs one attempt at a time; a lane whose walk has ended takes the next attempt of its wave's share that nobody has
// started -- the lanes that need one are counted with a ballot and numbered with a prefix count, no atomic -- and joins
// the others at their next segment (walks differ in length: one to max_depth + 1 segments).  A share is about a thousand
// attempts, so the lanes that wait for the longest walks of its last iterations are a few per cent (shares could be
// claimed from a batch-wide cursor instead, but device-wide atomics on one address from five thousand waves cost more
// than those lanes: measured 3.2 against 2.5 ms).
// Which lane runs an attempt does not matter: its stream is keyed by the attempt number and its outputs go to the
// attempt's own slots.
// (Tried: a two-block window of the stream filled at one place per iteration instead of a conditional generator
// evaluation at every draw site -- 12 % fewer instructions, but 13 more registers, which spill at five workgroups per CU:
// 2.7 against 2.2 ms.)
#ifndef RT_EMIT_MIN_BLOCKS
#define RT_EMIT_MIN_BLOCKS 5      // workgroups per CU the register allocation aims for
#endif
template <typename T, bool LDS>
__global__ void __launch_bounds__(256, RT_EMIT_MIN_BLOCKS) k_emit(DevScene<T> S, EmitCfg C, int phase, unsigned long long a0, uint32_t count, uint32_t per_wave,
                                              uint8_t *kind, T *rec, uint8_t *nseg, EmitCounters *ctr) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds) : global_scan_src<T>(S);
  unsigned long long segments = 0, errors = 0, overflows = 0;
  const T eps = T(C.epsilon);
  const int lane = threadIdx.x & 63;
  const unsigned long long below = ((unsigned long long)1 << lane) - 1;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned long long first = (unsigned long long)wave * per_wave;
  uint32_t cur = (uint32_t)min(first, (unsigned long long)count);          // this wave's unstarted attempts: [cur, end) (wave-uniform)
  const uint32_t end = (uint32_t)min(first + per_wave, (unsigned long long)count);
  bool have = false;
  uint32_t i = 0, my_segments = 0;
  RngStream rng; PhotonState<T> ph; MatStack st; int depth = 0;
  rng.init(0, 0, 0, 0);
  for (;;) {
    const unsigned long long need = __ballot(!have);
    if (need != 0ull && cur != end) {
      const uint32_t r = (uint32_t)__popcll(need & below), left = end - cur;
      if (!have && r < left) {
        i = cur + r; have = true;
        const unsigned long long attempt = a0 + i;
        rng.init(C.seed, phase == 1 ? RT_DOMAIN_EMIT_PHASE1 : RT_DOMAIN_EMIT_PHASE2, (uint32_t)attempt, (uint32_t)(attempt >> 32));
        create_photon<T>(S, eps, rng, ph);
        st.init(); depth = 0; my_segments = 0;
      }
      cur += min((uint32_t)__popcll(need), left);
    }
    if (!__any(have)) break;
    if (have) {
      my_segments++;
      const int k = photon_step<T>(S, G, C, rng, ph, st, depth, phase == 2, rec + (size_t)i * RT_REC_STRIDE, errors, overflows);
      if (k >= 0) {                                          // this attempt is over: record it
        kind[i] = (uint8_t)k;
        nseg[i] = (uint8_t)(my_segments < 255 ? my_segments : 255);
        segments += my_segments;
        have = false;
      }
    }
  }
  // one atomic per wave for the statistics
  for (int off = 32; off > 0; off >>= 1) {
    segments += __shfl_down(segments, off, 64); errors += __shfl_down(errors, off, 64); overflows += __shfl_down(overflows, off, 64);
  }
  if ((threadIdx.x & 63) == 0) emit_tally(ctr, segments, errors, overflows);
}

// createPhoton(source) / emitPhoton(photon, depth, trackOnlyCaustic, materialStack) for explicit photons, one lane each
// (the host's Scene.createPhoton / emitPhoton; the bulk emission above never goes through here).  Photon i draws from
// stream (seed, domain, first_stream + i) starting at draw first_draw[i] (0 without the array).
//   create: io[i] receives createPhoton's photon (origin3 dir3 power rgba), source[i] >= 0 names the light
//   walk:   io[i] is followed to its end; kind[i] = 0 / 1 / 2 and, if stored, io[i] becomes the stored record
// draws[i] receives the stream position afterwards.
template <typename T>
__global__ void __launch_bounds__(64) k_photons_explicit(DevScene<T> S, EmitCfg C, uint32_t n, int domain, unsigned long long first_stream,
                                                         const uint32_t *first_draw, int create, const int32_t *source, const int32_t *depth,
                                                         const int8_t *caustic, const unsigned long long *stacks, int track_only_caustic,
                                                         double *io, uint8_t *kind, uint32_t *draws, EmitCounters *ctr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ScanSrc<T> G = global_scan_src<T>(S);
  const unsigned long long stream = first_stream + i;
  RngStream rng;
  rng.init(C.seed, (uint32_t)domain, (uint32_t)stream, (uint32_t)(stream >> 32), first_draw ? first_draw[i] : 0u);
  double *r = io + (size_t)i * RT_REC_STRIDE;
  PhotonState<T> ph;
  unsigned long long segments = 0, errors = 0, overflows = 0;
  if (create) {
    create_photon<T>(S, T(C.epsilon), rng, ph, source ? source[i] : -1);
    r[0] = (double)ph.o.x; r[1] = (double)ph.o.y; r[2] = (double)ph.o.z; r[3] = (double)ph.d.x; r[4] = (double)ph.d.y; r[5] = (double)ph.d.z;
    r[6] = (double)ph.power; r[7] = (double)ph.r; r[8] = (double)ph.g; r[9] = (double)ph.b; r[10] = (double)ph.a;
  } else {
    ph.o = mk<T>(T(r[0]), T(r[1]), T(r[2])); ph.d = mk<T>(T(r[3]), T(r[4]), T(r[5]));
    ph.power = T(r[6]); ph.r = T(r[7]); ph.g = T(r[8]); ph.b = T(r[9]); ph.a = T(r[10]);
    ph.caustic = caustic ? (int)caustic[i] : -1;
    MatStack st; st.init();
    if (stacks) st.bits = stacks[i];
    T rec[RT_REC_STRIDE];
    const int stored = photon_walk<T>(S, G, C, rng, ph, st, depth ? depth[i] : 0, track_only_caustic != 0, rec, segments, errors, overflows);
    kind[i] = (uint8_t)stored;
    if (stored) for (int c = 0; c < RT_REC_STRIDE; c++) r[c] = (double)rec[c];
    emit_tally(ctr, segments, errors, overflows);
  }
  if (draws) draws[i] = rng.draw;
}

// ---- ordered compaction: count per block -> scan of block counts -> scatter ----
RT_D bool kind_match(uint8_t k, int want) { return want == 0 ? (k != 0) : (k == want); }

static __global__ void __launch_bounds__(256) k_compact_count(const uint8_t *kind, uint32_t count, int want, uint32_t *block_count) {
  __shared__ uint32_t wave_cnt[4];
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool m = (i < count) && kind_match(kind[i], want);
  unsigned long long mask = __ballot(m);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

template <typename T>
__global__ void __launch_bounds__(256) k_compact_scatter(const uint8_t *kind, const T *rec, uint32_t count, int want,
                                                         const uint32_t *block_offset, unsigned long long a0, uint32_t out_base,
                                                         uint32_t out_cap, double *out_rec, long long *out_attempt, uint8_t *out_kind) {
  __shared__ uint32_t wave_cnt[4];
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool m = (i < count) && kind_match(kind[i], want);
  unsigned long long mask = __ballot(m);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (!m) return;
  uint32_t rank = block_offset[blockIdx.x] + (uint32_t)__popcll(mask & (((unsigned long long)1 << lane) - 1));
  for (int w = 0; w < wv; w++) rank += wave_cnt[w];
  uint32_t dst = out_base + rank;
  if (dst >= out_cap) return;
  const T *r = rec + (size_t)i * RT_REC_STRIDE;
  double *o = out_rec + (size_t)dst * RT_REC_STRIDE;
  for (int c = 0; c < RT_REC_STRIDE; c++) o[c] = (double)r[c];
  if (out_attempt) out_attempt[dst] = (long long)(a0 + i);
  if (out_kind) out_kind[dst] = kind[i];
}

// ------------------------------------------------------------------ device-resident photon lists
// A SHARD is what one rank's attempt range leaves: its stored photons, compacted in attempt order, as records of
// W + 1 32-bit words -- W words of payload and a tag word = (attempt - tag_base) | (caustic ? 1 << 31 : 0).
//   payload formats: full = the 11 record values in the working precision (W = 11 or 22 words);
//                    packed = position, direction, power * colour as 9 floats (W = 9: the 36-byte photon of the map).
// Shards of one batch share tag_base, so the shards of all ranks, concatenated in rank order, are the batch's stored
// photons in attempt order with directly comparable tags.
// (RT_SHARD_FULL / RT_SHARD_PACKED: include/rt355.h)
template <typename T> constexpr uint32_t shard_words(int format) { return format == RT_SHARD_PACKED ? 9u : (uint32_t)(11 * sizeof(T) / 4); }

// (A workgroup's records are made in LDS and written as one run of consecutive words: a lane writing its own record
// stores ten words 40 bytes apart from its neighbour's.)
template <typename T, int FORMAT>
__global__ void __launch_bounds__(256) k_compact_shard(const uint8_t *kind, const T *rec, uint32_t count, const uint32_t *block_offset,
                                                       uint32_t tag_offset, uint32_t cap, uint32_t *shard) {
  constexpr uint32_t W = shard_words<T>(FORMAT), RW = W + 1;
  __shared__ uint32_t wave_cnt[4];
  __shared__ uint32_t stage[256 * RW];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint8_t kd = i < count ? kind[i] : 0;
  const bool m = kd != 0;
  const unsigned long long mask = __ballot(m);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (m) {
    uint32_t local = (uint32_t)__popcll(mask & (((unsigned long long)1 << lane) - 1));
    for (int w = 0; w < wv; w++) local += wave_cnt[w];
    const T *r = rec + (size_t)i * RT_REC_STRIDE;
    uint32_t *o = stage + local * RW;
    if (FORMAT == RT_SHARD_PACKED) {
      float *f = (float *)o;
      f[0] = (float)r[0]; f[1] = (float)r[1]; f[2] = (float)r[2]; f[3] = (float)r[3]; f[4] = (float)r[4]; f[5] = (float)r[5];
      f[6] = (float)(r[7] * r[6]); f[7] = (float)(r[8] * r[6]); f[8] = (float)(r[9] * r[6]);
    } else {
      const uint32_t *src = (const uint32_t *)r;
#pragma unroll
      for (uint32_t c = 0; c < W; c++) o[c] = src[c];
    }
    o[W] = (tag_offset + i) | (kd == 2 ? 0x80000000u : 0u);
  }
  __syncthreads();
  const uint32_t base = block_offset[blockIdx.x], cnt = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
  const uint32_t words = (min(base + cnt, cap) - min(base, cap)) * RW;      // (records beyond the shard's capacity are dropped)
  uint32_t *out = shard + (size_t)base * RW;
  for (uint32_t k = threadIdx.x; k < words; k += 256u) out[k] = stage[k];
}

// State of the capped lists on the device; the host reads it back once per batch.
struct ListsState {
  uint32_t n_d, n_c;              // photons in the diffuse / caustic list
  uint32_t total_d, total_c;      // stored photons of either kind in the batch just applied
  uint32_t take_d, take_c;        // ... of which the lists took
  uint32_t last_tag;              // tag (without the kind bit) of the attempt that ended the phase inside this batch, or 0xFFFFFFFF
  uint32_t pad_;
};

// merged position i = shard (i / stride), record (i % stride); valid while the record index is below the shard's count
__device__ __forceinline__ uint32_t merged_tag(const uint32_t *shards, const uint32_t *counts, uint32_t n_shards, uint32_t stride, uint32_t W,
                                               uint32_t i, bool &valid) {
  const uint32_t sh = i / stride, j = i - sh * stride;
  valid = sh < n_shards && j < min(counts[sh], stride);
  return valid ? shards[(size_t)i * (W + 1) + W] : 0u;
}

// per-workgroup numbers of diffuse and caustic records of the merged batch
static __global__ void __launch_bounds__(256) k_lists_count(const uint32_t *shards, const uint32_t *counts, uint32_t n_shards, uint32_t stride,
                                                            uint32_t W, uint32_t *block_d, uint32_t *block_c) {
  __shared__ uint32_t cd[4], cc[4];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid; const uint32_t tag = merged_tag(shards, counts, n_shards, stride, W, i, valid);
  const unsigned long long md = __ballot(valid && !(tag >> 31)), mc = __ballot(valid && (tag >> 31));
  if ((threadIdx.x & 63) == 0) { cd[threadIdx.x >> 6] = (uint32_t)__popcll(md); cc[threadIdx.x >> 6] = (uint32_t)__popcll(mc); }
  __syncthreads();
  if (threadIdx.x == 0) { block_d[blockIdx.x] = cd[0] + cd[1] + cd[2] + cd[3]; block_c[blockIdx.x] = cc[0] + cc[1] + cc[2] + cc[3]; }
}

// The caps of emitManyPhotons (rayphoton.js:1346-1383) on one merged, attempt-ordered batch.
// Phase 1: the diffuse list takes records until it is full; the attempt that fills it is the last one the sequential
// loop runs, so the caustic list takes the caustic records up to that attempt (and up to its own cap).  Phase 2: only
// the caustic list takes records; the attempt that fills it ends the phase.  Pass A (kind 1 in phase 1, kind 2 in
// phase 2) finds that attempt; pass B (phase 1 only) takes the caustic records before it.
//   block_off: exclusive scans of the per-workgroup counts; totals[0] = diffuse, totals[1] = caustic records in the batch
// (The records a workgroup takes are consecutive in their list: they are collected in LDS and written as one run of
// consecutive words.)
#define RT_LIST_WORDS_MAX 22u
template <int PASS>
__global__ void __launch_bounds__(256) k_lists_take(const uint32_t *shards, const uint32_t *counts, uint32_t n_shards, uint32_t stride, uint32_t W,
                                                    const uint32_t *block_d, const uint32_t *block_c, const uint32_t *totals, int phase,
                                                    uint32_t cap_d, uint32_t cap_c, unsigned long long tag_base, ListsState *st,
                                                    uint32_t *list_d, long long *att_d, uint32_t *list_c, long long *att_c) {
  __shared__ uint32_t cd[4], cc[4];
  __shared__ uint32_t stage[256 * RT_LIST_WORDS_MAX];
  __shared__ uint32_t first_rank;                             // list position of the workgroup's first taken record
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid; const uint32_t tag = merged_tag(shards, counts, n_shards, stride, W, i, valid);
  const bool is_c = valid && (tag >> 31), is_d = valid && !(tag >> 31);
  const unsigned long long md = __ballot(is_d), mc = __ballot(is_c);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { cd[wv] = (uint32_t)__popcll(md); cc[wv] = (uint32_t)__popcll(mc); }
  if (threadIdx.x == 0) first_rank = 0xFFFFFFFFu;
  __syncthreads();
  const unsigned long long below = ((unsigned long long)1 << lane) - 1;
  uint32_t ld = (uint32_t)__popcll(md & below), lc = (uint32_t)__popcll(mc & below);      // rank among the workgroup's diffuse / caustic records
  for (int w = 0; w < wv; w++) { ld += cd[w]; lc += cc[w]; }
  const uint32_t rd = block_d[blockIdx.x] + ld, rc = block_c[blockIdx.x] + lc;           // ... among the batch's
  const uint32_t n_d = st->n_d, n_c = st->n_c;                     // list lengths before this batch
  const uint32_t room_d = cap_d - min(n_d, cap_d), room_c = cap_c - min(n_c, cap_c);
  const long long ph = (long long)phase << 62;
  // which list this pass fills, and whether this thread's record goes there (the records taken are the first of their
  // kind in the workgroup: local ranks 0 .. taken - 1)
  bool take = false; bool to_c = false; uint32_t local = 0, rank = 0;
  if (PASS == 0) {
    if (phase == 1) {
      take = is_d && rd < room_d; local = ld; rank = n_d + rd;
      if (take && rd + 1 == room_d) { st->last_tag = tag & 0x7FFFFFFFu; st->pad_ = rc; }      // rc: caustic records before the last attempt
    } else {
      take = is_c && rc < room_c; to_c = true; local = lc; rank = n_c + rc;
      if (take && rc + 1 == room_c) st->last_tag = tag & 0x7FFFFFFFu;
    }
  } else {                                                        // phase 1, caustic records up to the last attempt
    const uint32_t upto = st->last_tag == 0xFFFFFFFFu ? totals[1] : st->pad_;
    take = is_c && rc < min(upto, room_c); to_c = true; local = lc; rank = n_c + rc;
  }
  if (take) {
    const uint32_t *src = shards + (size_t)i * (W + 1);
    for (uint32_t c = 0; c < W; c++) stage[local * W + c] = src[c];
    (to_c ? att_c : att_d)[rank] = ph | (long long)(tag_base + (tag & 0x7FFFFFFFu));
    if (local == 0) first_rank = rank;
  }
  const uint32_t taken = __syncthreads_count(take);
  if (taken == 0) return;
  uint32_t *out = (PASS == 0 && phase == 1 ? list_d : list_c) + (size_t)first_rank * W;
  for (uint32_t k = threadIdx.x; k < taken * W; k += 256u) out[k] = stage[k];
}

// closes a batch: new list lengths, what was taken (one thread)
static __global__ void k_lists_finish(const uint32_t *totals, int phase, uint32_t cap_d, uint32_t cap_c, ListsState *st) {
  const uint32_t n_d = st->n_d, n_c = st->n_c;
  const uint32_t room_d = cap_d - min(n_d, cap_d), room_c = cap_c - min(n_c, cap_c);
  uint32_t take_d = 0, take_c = 0;
  if (phase == 1) {
    take_d = min(totals[0], room_d);
    const uint32_t upto = st->last_tag == 0xFFFFFFFFu ? totals[1] : st->pad_;
    take_c = min(upto, room_c);
  } else {
    take_c = min(totals[1], room_c);
  }
  st->total_d = totals[0]; st->total_c = totals[1]; st->take_d = take_d; st->take_c = take_c;
  st->n_d = n_d + take_d; st->n_c = n_c + take_c;
}

// sum of the per-attempt segment counts of attempts [0, upto) of a batch (emitPhoton calls the sequential loop makes)
static __global__ void __launch_bounds__(256) k_sum_segments(const uint8_t *nseg, uint32_t upto, unsigned long long *out) {
  unsigned long long s = 0;
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x, n16 = upto / 16u;
  const uint4 *words = (const uint4 *)nseg;                  // (the array starts a device allocation: 16-byte aligned)
  for (uint32_t i = gtid; i < n16; i += stride) {
    const uint4 v = words[i];
    s += __builtin_amdgcn_sad_u8(v.x, 0u, 0u) + __builtin_amdgcn_sad_u8(v.y, 0u, 0u) + __builtin_amdgcn_sad_u8(v.z, 0u, 0u) + __builtin_amdgcn_sad_u8(v.w, 0u, 0u);   // sums of four bytes
  }
  for (uint32_t i = n16 * 16u + gtid; i < upto; i += stride) s += nseg[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// list records -> the boundary's 11 doubles (full format) for the host's photon lists
template <typename T>
__global__ void __launch_bounds__(256) k_lists_export(const uint32_t *list, const long long *att, uint32_t n, double *rec_out, long long *att_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T *r = (const T *)(list + (size_t)i * (11 * sizeof(T) / 4));
  if (rec_out) for (int c = 0; c < 11; c++) rec_out[(size_t)i * 11 + c] = (double)r[c];
  if (att_out) { const long long a = att[i]; att_out[2 * (size_t)i] = (a >> 62) & 3; att_out[2 * (size_t)i + 1] = a & (((long long)1 << 62) - 1); }
}
