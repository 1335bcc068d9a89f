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
struct EmitCounters { unsigned long long segments, errors, overflows; };

#define RT_REC_STRIDE 11

template <typename T, bool LDS>
__global__ void __launch_bounds__(256) k_emit(DevScene<T> S, EmitCfg C, int phase, unsigned long long a0, uint32_t count,
                                              uint8_t *kind, T *rec, EmitCounters *ctr) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds) : global_scan_src<T>(S);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long segments = 0, errors = 0, overflows = 0;
  if (i < count) {
    const unsigned long long attempt = a0 + i;
    const T eps = T(C.epsilon);
    const bool track_only_caustic = (phase == 2);
    RngStream rng;
    rng.init(C.seed, phase == 1 ? RT_DOMAIN_EMIT_PHASE1 : RT_DOMAIN_EMIT_PHASE2, (uint32_t)attempt, (uint32_t)(attempt >> 32));
    PhotonState<T> ph; create_photon<T>(S, eps, rng, ph);
    MatStack st; st.init();
    int depth = 0, stored = 0;
    for (;;) {
      T t; segments++;
      int idx = scan_scene<T, 1>(S, G, ph.o, ph.d, eps, T(0), t);
      if (idx < 0) break;                                                      // left the scene
      V3<T> hit = ph.o + t * ph.d;
      V3<T> local = shape_colour<T>(S, idx, hit);
      V3<T> n = shape_normal<T>(S, idx, hit);
      const T cos1 = -dot(ph.d, n);
      const bool has_mat = S.meta[4 * idx + 1] < 8;
      const int mat = S.meta[4 * idx + 1] & 7;
      if (has_mat && S.mat[0][mat] == T(1) && depth < C.max_depth) {           // :1484
        V3<T> ndir; bool ovf = false;
        if (!fresnel_bounce<T>(S, mat, ph.d, n, cos1, st, rng, ndir, ovf)) { errors++; break; }   // :1487-1490 throws
        if (ovf) overflows++;
        ph.o = hit; ph.d = normalize<T>(ndir, eps);
        if (ph.caustic < 0) ph.caustic = 1;
        depth++;
        continue;
      }
      // opaque, or transparent at depth >= MAX_DEPTH where reflectance is undefined (never "< U")
      const bool opaque_mat = has_mat && S.mat[0][mat] != T(1);
      if (rng.next_k() < (opaque_mat ? S.refl_k[mat] : 0u)) {                  // :1523
        if (depth < C.max_depth) {
          if (rng.next_k() < S.spec_k[mat]) {                                  // :1525-1533 specular
            ph.o = hit; ph.d = normalize<T>(ph.d + (T(2) * cos1) * n, eps);
            if (ph.caustic < 0) ph.caustic = 1;
            depth++;
            continue;
          } else if (!track_only_caustic) {                                    // :1536-1546 diffuse
            V3<T> ndir = vec_perturb<T>(n, eps, rng);
            T adj = ph.power * T(255 + 255 + 255 + 1) / (local.x + local.y + local.z + T(1));
            ph.r = ph.r / T(255) * local.x; ph.g = ph.g / T(255) * local.y; ph.b = ph.b / T(255) * local.z; ph.a = T(1);
            ph.power = adj;
            ph.o = hit; ph.d = normalize<T>(ndir, eps); ph.caustic = 0;
            depth++;
            continue;
          }
        }
        break;                                                                 // dropped
      }
      if (depth > 0) {                                                         // absorbed: store (:1549-1553)
        const bool is_c = ph.caustic == 1;
        T power = ph.power / (is_c ? T(C.num_caustic) : T(C.num_diffuse));
        V3<T> back = normalize<T>(T(-1) * ph.d, eps);
        T *r = rec + (size_t)i * RT_REC_STRIDE;
        r[0] = hit.x; r[1] = hit.y; r[2] = hit.z; r[3] = back.x; r[4] = back.y; r[5] = back.z;
        r[6] = power; r[7] = ph.r; r[8] = ph.g; r[9] = ph.b; r[10] = ph.a;
        stored = is_c ? 2 : 1;
      }
      break;
    }
    kind[i] = (uint8_t)stored;
  }
  // one atomic per wave for the statistics
  for (int off = 32; off > 0; off >>= 1) {
    segments += __shfl_down(segments, off, 64); errors += __shfl_down(errors, off, 64); overflows += __shfl_down(overflows, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (segments) atomicAdd(&ctr->segments, segments);
    if (errors) atomicAdd(&ctr->errors, errors);
    if (overflows) atomicAdd(&ctr->overflows, overflows);
  }
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
