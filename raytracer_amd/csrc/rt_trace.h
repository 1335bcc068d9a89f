// Wavefront path tracer.  SoA path state in HBM; survivors of a pass are compacted, in order, into the
// next queue.
//
//   primary (raygen + closest hit + shade, state in registers)
//     -> [ intersect_closest -> shade ]*  while many paths are left
//     -> tail (every remaining path to its end in one launch) -> resolve
//
// Semantics are the reference's traceRay / directLight (rayphoton.js:1626-1880).  Every
// branch of traceRay makes at most one recursive call and adds its result unscaled, so a
// path is a loop that accumulates the local terms of its opaque hits and ends with the
// colour of the event that terminates it.
#pragma once
#include "rt_compact.h"
#include "rt_geom.h"
#include "rt_kdq.h"
#include "rt_sample.h"

template <typename T> struct PathSoA {
  T *ox, *oy, *oz, *dx, *dy, *dz;     // current segment
  T *max_dist, *importance;
  T *acc_r, *acc_g, *acc_b;           // sum of the local terms of the opaque hits so far
  T *hit_t; int32_t *hit_idx;         // written by intersect_closest
  uint32_t *draw;                     // RNG draws consumed by this path's stream
  int32_t *depth;                     // low 16 bits depth, bit 16: an opaque hit fixed alpha = 1
  uint64_t *stack;                    // MatStack bits
  uint32_t *key;                      // the path's RNG stream: canvas pixel Y * width + X
  T *out;                             // n x 4: finished paths' colour
};

// one path's state in registers
template <typename T> struct PathState {
  V3<T> o, d; T max_dist, importance; V3<T> acc;
  uint32_t draw; int32_t depth; uint64_t stack;
  uint32_t key;                       // canvas pixel Y * width + X: the first key word of the path's RNG stream (working it out from
                                      // the path number at every bounce costs three divisions by run-time constants)
  Philox4 blk; uint32_t blk_idx;      // not stored: the stream's current Philox block, if a pass already computed it
};
template <typename T> RT_D PathState<T> load_path(const PathSoA<T> &P, uint32_t p) {
  PathState<T> st;
  st.o = mk<T>(P.ox[p], P.oy[p], P.oz[p]); st.d = mk<T>(P.dx[p], P.dy[p], P.dz[p]);
  st.max_dist = P.max_dist[p]; st.importance = P.importance[p];
  st.acc = mk<T>(P.acc_r[p], P.acc_g[p], P.acc_b[p]);
  st.draw = P.draw[p]; st.depth = P.depth[p]; st.stack = P.stack[p]; st.key = P.key[p]; st.blk_idx = 0xFFFFFFFFu;
  return st;
}
template <typename T> RT_D void store_path(const PathSoA<T> &P, uint32_t p, const PathState<T> &st) {
  P.ox[p] = st.o.x; P.oy[p] = st.o.y; P.oz[p] = st.o.z; P.dx[p] = st.d.x; P.dy[p] = st.d.y; P.dz[p] = st.d.z;
  P.max_dist[p] = st.max_dist; P.importance[p] = st.importance;
  P.acc_r[p] = st.acc.x; P.acc_g[p] = st.acc.y; P.acc_b[p] = st.acc.z;
  P.draw[p] = st.draw; P.depth[p] = st.depth; P.stack[p] = st.stack; P.key[p] = st.key;
}

// The same for a workgroup's own copy in LDS (fp32 builds): 16 words per path, field f of slot s at L[f * n + s].
#define RT_LSTATE_WORDS 16
RT_D PathState<float> load_path_lds(const uint32_t *L, uint32_t n, uint32_t s) {
  PathState<float> st;
  auto f = [&](int k) { return __uint_as_float(L[k * n + s]); };
  st.o = mk<float>(f(0), f(1), f(2)); st.d = mk<float>(f(3), f(4), f(5));
  st.max_dist = f(6); st.importance = f(7); st.acc = mk<float>(f(8), f(9), f(10));
  st.draw = L[11 * n + s]; st.depth = (int32_t)L[12 * n + s];
  st.stack = (uint64_t)L[13 * n + s] | ((uint64_t)L[14 * n + s] << 32); st.key = L[15 * n + s]; st.blk_idx = 0xFFFFFFFFu;
  return st;
}
RT_D void store_path_lds(uint32_t *L, uint32_t n, uint32_t s, const PathState<float> &st) {
  auto f = [&](int k, float v) { L[k * n + s] = __float_as_uint(v); };
  f(0, st.o.x); f(1, st.o.y); f(2, st.o.z); f(3, st.d.x); f(4, st.d.y); f(5, st.d.z);
  f(6, st.max_dist); f(7, st.importance); f(8, st.acc.x); f(9, st.acc.y); f(10, st.acc.z);
  L[11 * n + s] = st.draw; L[12 * n + s] = (uint32_t)st.depth;
  L[13 * n + s] = (uint32_t)st.stack; L[14 * n + s] = (uint32_t)(st.stack >> 32); L[15 * n + s] = st.key;
}

struct TraceCfg {
  double epsilon, max_trace_dist, diffuse_radius, caustic_radius, standard_lamp_area, shadow_bias;
  int32_t max_depth, sub_sample, use_direct_light, use_photon_maps, max_bounces;
  float wave_gather_ratio;       // > 0: first hits gather their photon sums wave by wave while the box of a wave's balls is within this many ball areas
  uint64_t seed;
  int32_t x0, y0, w, h;               // tile origin (canvas pixels); columns and rows of traced samples
  int32_t block;                      // canvas pixels per traced sample along each axis (1 = every pixel)
  uint32_t n_paths;                   // w * h * sub_sample^2
  uint32_t magic_w, magic_ss, magic_s; // floor(2^32 / d) for d = w, sub_sample^2, sub_sample (see div_by)
  // with one sub-sample per pixel a finished path is the finished pixel: the shading kernels write the tile's outputs
  // themselves (any of the three may be null) and no resolve pass runs
  int32_t fuse_resolve; double *pre, *post; uint8_t *rgba8;
  // > 0: the first pass hands its survivors to k_tail through sub-queues of this capacity (block_append_subqueue)
  uint32_t subq_cap;
  uint32_t iter_base;                 // bounce passes (k_bounce) that ran between the first pass and k_tail
  // > 0: paths are numbered tile by tile, 32 x 8 samples per tile, tiles_x tiles per row of tiles (w % 32 == 0 and
  // h % 8 == 0); 0: row by row.  A workgroup of the first pass then covers a compact rectangle: a tighter culling
  // pyramid, and chunks of survivors that live about equally long.
  uint32_t tiles_x, magic_tx;
  // Row bands (multi-GPU tile split): band_shift > 0 means this call traces only the bands of 2^band_shift rows that
  // belong to one rank, band_first + j * band_period for j = 0, 1, ...; traced row r (and the outputs, which hold the
  // traced rows only) maps to tile row (r >> band_shift) * band_period + band_first + (r & (2^band_shift - 1)).
  uint32_t band_shift, band_period, band_first;
  // fp32 frames with photon maps: hits after the first (and first hits a wave could not gather together) do not walk
  // the maps where they are shaded -- one lane walking for thousands of steps while the rest of its wave waits -- but
  // leave a request (3 x float4: point + path, normal, albedo / 255) in gq; k_gather_deferred serves the requests with
  // every lane busy and adds the results to the path's entry of gacc (fixed point, steps of 2^-32: the order of the additions
  // does not matter), which the resolve pass adds to the path's colour.  gq == nullptr: gathers where they are shaded.
  // Slots 0 .. gq_first - 1 belong to the first pass: path p's first hit leaves its request in slot p (no counter: a word
  // that every wave of a frame adds to costs ~10 ns per wave); later hits take slots gq_first + (*gq_count)++.  gq_valid[slot]
  // != 0 marks a slot that holds a request of this frame (cleared per frame).
  float4 *gq; uint32_t *gq_count; uint32_t gq_cap; long long *gacc; uint32_t *gq_valid; uint32_t gq_first;
  unsigned long long *gsat;      // counts the partial sums that left the accumulators' range (gather_accumulate), or nullptr
  const uint32_t *gq_order;      // requests in the order k_gather_deferred serves them (sorted by position), or nullptr
  // fp32 first pass: per workgroup the list of primitives its rays can reach (cull_stride words each: 16 segment words,
  // then the sorted slots as 16-bit entries), made once per scene and tile layout by k_cull_lists, or nullptr: every
  // workgroup works its list out itself
  const uint32_t *cull_lists; uint32_t cull_stride;
};
// words of one workgroup's list of reachable primitives
static inline __host__ __device__ uint32_t cull_list_words(uint32_t n_shapes) { return 16u + (2u * n_shapes + 3u) / 4u; }
#define RT_GACC_SCALE 4294967296.0      // 2^32: values below 2^31 in steps of 2.3e-10
// row of the tile that traced row `row` stands for
RT_D uint32_t band_row(const TraceCfg &C, uint32_t row) {
  if (!C.band_shift) return row;
  return (row >> C.band_shift) * C.band_period + C.band_first + (row & ((1u << C.band_shift) - 1u));
}

#define RT_MAX_WAVE_BOUNCES 62
// device-side bookkeeping of one rt_trace_tile call; counts[b] = queue length of wavefront pass b
#define RT_CTR_SLOTS 64
#define RT_MAX_BOUNCE_PASSES 2
#define RT_MAX_TAIL_ROUNDS 3
struct TraceCounters {
  unsigned long long shadow_rays, gathers, stack_overflows, gather_saturations, tail_rays[RT_CTR_SLOTS];
  // per slot s a cache line: [16 s] ray-primitive tests counted by the scans that skip primitives, [16 s + 1] shadow rays,
  // [16 s + 2] gathers, [16 s + 3] material-stack overflows (flush_tally); the single words above are host-side sums
  unsigned long long prim_tests[RT_CTR_SLOTS * 16];
  uint32_t max_iter[RT_CTR_SLOTS];          // iterations the longest-lived workgroup of k_tail ran
  // lengths of the sub-queues the first pass ([0]) and the bounce passes after it ([1], [2]) leave (s at [16 s])
  // ([RT_MAX_BOUNCE_PASSES + 1 + r]: paths that launch r of k_tail handed on to launch r + 1)
  uint32_t subq_count[RT_MAX_BOUNCE_PASSES + 1 + RT_MAX_TAIL_ROUNDS][RT_SUBQ * 16];
  uint32_t counts[RT_MAX_WAVE_BOUNCES + 2];
#ifdef RT_TAIL_PROBE
  unsigned long long probe[3][6];           // [first / second / later iterations][phase] cycles seen by lane 0 of each workgroup; [..][5] = iterations
#endif
};

// per-lane tallies of a kernel; tests: ray-primitive tests (counted by the hierarchy scans only)
struct ShadeTally { unsigned long long shadow_rays = 0, gathers = 0, overflows = 0; uint32_t tests = 0; };

// adds a wave's tallies to the frame counters: one atomic per wave and counter (call with the wave converged)
RT_D void flush_tally(const ShadeTally &t, TraceCounters *ctr) {
  unsigned long long a = t.shadow_rays, b = t.gathers, c = t.overflows, e = t.tests;
  if (__ballot((a | b | c | e) != 0) == 0) return;
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); c += __shfl_down(c, off, 64); e += __shfl_down(e, off, 64); }
  if ((threadIdx.x & 63) == 0) {
    // one cache line per slot, the four tallies side by side in it: every wave of a frame ends here, and adds to ONE word
    // from tens of thousands of waves take ~10 ns each, one after the other (0.3 ms per tally on a 1080p frame)
    unsigned long long *line = &ctr->prim_tests[((blockIdx.x * 8 + (threadIdx.x >> 6)) % RT_CTR_SLOTS) * 16];
    if (e) atomicAdd(line, e);
    if (a) atomicAdd(line + 1, a);
    if (b) atomicAdd(line + 2, b);
    if (c) atomicAdd(line + 3, c);
  }
}


// q = x / d and r = x % d without the ~40-instruction integer division: one multiply-high with m = floor(2^32 / d)
// (never above the quotient, at most 2 below it for x < 2^32) and two corrections
RT_D uint32_t div_by(uint32_t x, uint32_t d, uint32_t m, uint32_t &r) {
  if (d == 1) { r = 0; return x; }
  uint32_t q = __umulhi(x, m);
  r = x - q * d;
  if (r >= d) { q++; r -= d; }
  if (r >= d) { q++; r -= d; }
  return q;
}
// column and row (in traced samples) of the pp-th pixel in path order
RT_D void pixel_colrow(const TraceCfg &C, uint32_t pp, uint32_t &col, uint32_t &row) {
  if (C.tiles_x) {
    uint32_t tx;
    const uint32_t ty = div_by(pp >> 8, C.tiles_x, C.magic_tx, tx), in = pp & 255u;
    col = tx * 32u + (in & 31u); row = ty * 8u + (in >> 5);
  } else {
    row = div_by(pp, (uint32_t)C.w, C.magic_w, col);
  }
}
// where the pp-th pixel in path order sits in the tile's (row-major) outputs
RT_D uint32_t pixel_out_index(const TraceCfg &C, uint32_t pp) {
  if (!C.tiles_x) return pp;
  uint32_t col, row;
  pixel_colrow(C, pp, col, row);
  return row * (uint32_t)C.w + col;
}
// pixel and sub-sample of path p
RT_D void path_pixel(const TraceCfg &C, uint32_t p, int &X, int &Y, uint32_t &sub) {
  uint32_t col, row;
  const uint32_t pix = div_by(p, (uint32_t)(C.sub_sample * C.sub_sample), C.magic_ss, sub);
  pixel_colrow(C, pix, col, row);
  X = C.x0 + (int)col * C.block; Y = C.y0 + (int)band_row(C, row) * C.block;
}

// traceOnCanvas ray set-up (rayphoton.js:1589-1609, superSampleScale 0) for path p: the initial path state
// with the camera ray as its segment.
template <typename T>
RT_D PathState<T> raygen_path(const DevScene<T> &S, const TraceCfg &C, uint32_t p) {
  int X, Y; uint32_t sub, sxu;
  path_pixel(C, p, X, Y, sub);
  const int sy = (int)div_by(sub, (uint32_t)C.sub_sample, C.magic_s, sxu), sx = (int)sxu;
  RngStream rng; rng.init(C.seed, RT_DOMAIN_RENDER, (uint32_t)(Y * (int)S.width + X), sub);
  T ssx = T(X), ssy = T(Y);
  if (C.block > 1) { ssx += T(C.block) * rng.next<T>(); ssy += T(C.block) * rng.next<T>(); }    // one point for the block, :1593-1596
  T u = ((ssx + (T(sx) + rng.next<T>()) / T(C.sub_sample)) * T(2) / T(S.width)) - T(1);
  T v = -(((ssy + (T(sy) + rng.next<T>()) / T(C.sub_sample)) * T(2) / T(S.height)) - T(1));
  u *= S.camera[12]; v *= S.camera[13];
  V3<T> co = ld3(S.camera), cu = ld3(S.camera + 3), cv = ld3(S.camera + 6), cw = ld3(S.camera + 9);
  V3<T> xyz = ((co + u * cu) + v * cv) + T(-1) * cw;                    // Camera.toXYZ, rayphoton.js:712
  PathState<T> st;
  st.o = co; st.d = normalize<T>(xyz - co, T(C.epsilon));
  st.max_dist = T(C.max_trace_dist); st.importance = T(1);
  st.acc = mk<T>(0, 0, 0);
  st.draw = rng.draw; st.depth = 0; st.key = (uint32_t)(Y * (int)S.width + X); st.blk = rng.block; st.blk_idx = rng.block_idx;
  MatStack ms; ms.init(); st.stack = ms.bits;
  return st;
}

// Closest hit of every queued path against the whole scene (rayphoton.js:1642-1650), for the segment the
// previous pass left in the path state.  This is the VALU-bound kernel: the primitive index is wave-uniform,
// so primitive operands are broadcast LDS (or scalar) loads and every lane spends its cycles on the distance
// arithmetic.  WHITTED: ray.js's scan order (equal distances go to the highest index) instead of rayphoton.js's.
template <typename T, bool LDS, bool WHITTED>
__global__ void __launch_bounds__(256) k_intersect_closest(DevScene<T> S, PathSoA<T> P, TraceCfg C, const uint32_t *queue, const uint32_t *count_ptr,
                                                           TraceCounters *ctr) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  // device-side queue length (the grid is only an upper bound)
  const uint32_t count = min(*count_ptr, C.n_paths);
  if (blockIdx.x * blockDim.x >= count) return;
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds) : global_scan_src<T>(S);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  ShadeTally tally;
  const uint32_t p = i < count ? queue[i] : 0xFFFFFFFFu;
  if (p < C.n_paths) {                                      // (false only for the lanes past the end of the queue)
    V3<T> o = mk<T>(P.ox[p], P.oy[p], P.oz[p]), d = mk<T>(P.dx[p], P.dy[p], P.dz[p]);
    T t;
    int idx = scan_scene<T, WHITTED ? 3 : 0>(S, G, o, d, T(C.epsilon), T(0), t, &tally.tests);
    P.hit_t[p] = t; P.hit_idx[p] = idx;
  }
  flush_tally(tally, ctr);
}

// first key word of path p's stream (what raygen_path leaves in the path state)
RT_D uint32_t path_key(uint32_t width, const TraceCfg &C, uint32_t p) {
  int X, Y; uint32_t sub;
  path_pixel(C, p, X, Y, sub);
  return (uint32_t)(Y * (int)width + X);
}
// stream of path p continued where its state left it: the key is (pixel, sub-sample)
template <typename T> RT_D RngStream path_rng(uint32_t width, const TraceCfg &C, uint32_t p, const PathState<T> &st) {
  uint32_t sub = 0;
  if (C.sub_sample != 1) (void)div_by(p, (uint32_t)(C.sub_sample * C.sub_sample), C.magic_ss, sub);
  RngStream rng; rng.init(C.seed, RT_DOMAIN_RENDER, st.key, sub, st.draw);
  rng.block = st.blk; rng.block_idx = st.blk_idx;
  return rng;
}

// directLight (rayphoton.js:1841-1880): per light one createPhoton() of a RANDOM light, scaled by
// THIS light's area; an unclamped cosine; floor at the end.
template <typename T>
RT_D V3<T> direct_light(const DevScene<T> &S, const ScanSrc<T> &G, const TraceCfg &C, V3<T> point, V3<T> normal, V3<T> local, RngStream &rng,
                        unsigned long long &shadow_rays, uint32_t *tests) {
  const T eps = T(C.epsilon);
  V3<T> col = mk<T>(0, 0, 0);
  for (uint32_t l = 0; l < S.n_lights; l++) {
    PhotonState<T> ph; create_photon<T>(S, eps, rng, ph);
    T power = ph.power * S.light_data[l][1];
    V3<T> dir = normalize<T>(ph.o - point, eps);
    T source_dist = rt_sqrt(sqlen(point - ph.o));
    T thit;
    shadow_rays++;
    if (!scan_scene<T, 2>(S, G, point, dir, eps, source_dist, thit, tests)) {
      T dn = dot(dir, normal);
      T k = T(C.standard_lamp_area);
      col.x += (ph.r / T(255)) * local.x * dn * power / k;
      col.y += (ph.g / T(255)) * local.y * dn * power / k;
      col.z += (ph.b / T(255)) * local.z * dn * power / k;
    }
  }
  return mk<T>(rt_floor(col.x), rt_floor(col.y), rt_floor(col.z));
}

// Radiance estimate from one photon map (rayphoton.js:1765-1815): sum over the photons within r
// of colour * power * max(0, dir . normal).  fp64 paths use the fp64 photon records when the map
// kept them; the fp32 path reads the 24-byte fp32 payload.
template <typename T>
RT_D V3<T> gather_map(const DevMap &M, V3<T> x, V3<T> normal, double r) {
  T sr = T(0), sg = T(0), sb = T(0);
  const double r2 = r * r;
  const bool use64 = (sizeof(T) == 8) && (M.rec64 != nullptr);
  kd_radius_walk(M, (double)x.x, (double)x.y, (double)x.z, r, [&](uint32_t slot, float4 nd) {
    if (use64) {
      const double *R = M.rec64 + (size_t)slot * 11;
      if (sq_dist((double)x.x, (double)x.y, (double)x.z, R[0], R[1], R[2]) < r2) {
        T w = dot(mk<T>(T(R[3]), T(R[4]), T(R[5])), normal);
        w = w > T(0) ? w : T(0);
        sr += T(R[7]) * T(R[6]) * w; sg += T(R[8]) * T(R[6]) * w; sb += T(R[9]) * T(R[6]) * w;
      }
    } else if (sq_dist((double)x.x, (double)x.y, (double)x.z, (double)nd.x, (double)nd.y, (double)nd.z) < r2) {
      const float *pp = M.pay + (size_t)slot * 6;
      T w = dot(mk<T>(T(pp[0]), T(pp[1]), T(pp[2])), normal);
      w = w > T(0) ? w : T(0);
      sr += T(pp[3]) * w; sg += T(pp[4]) * w; sb += T(pp[5]) * w;
    }
  });
  return mk<T>(sr, sg, sb);
}


// ---- the same radiance estimate for the 64 surface points of a wave at once (fp32 build, first hits of a frame).
// Primary rays of one wave hit neighbouring points, so their search balls overlap almost completely: instead of 64
// divergent tree walks, ONE cooperative walk over the bounding box of the balls collects every photon inside it into
// LDS (position and payload, one node per lane and step), and whenever the buffer fills every lane adds the candidates
// within its own ball.  Which photons a lane adds is decided as in gather_map -- the fp64 `d2 < r^2` of
// kdtree.js:75 -- through an fp32 distance with an error band (fp64 only for candidates inside the band), so the set
// is identical; the order of the additions is not, hence a tolerance against the one-lane gather, not identity.
#ifndef RT_WG_CAP
#define RT_WG_CAP 128                      // candidates per wave (a step adds up to 64; smaller buffers, more waves per SIMD: 256 was 12 % slower)
#endif
#define RT_WG_FRONT 256                    // node queue per wave (a ring)
// LDS written by some lanes of a wave is read by others: order the accesses (the waves of a workgroup run this code
// independently, so no workgroup barrier)
#define RT_WAVE_SYNC() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#ifdef RT_GATHER_PROBE
static __device__ unsigned long long g_gather_probe[8];
static __device__ unsigned long long g_gather_hist[40];   // request waves of k_gather_deferred by log2 of their duration; [32..]: sums of durations by the same buckets / 1024   // walks, steps, candidates, in-ball lane tests, needing lanes, consumes
#endif
#define RT_WG_STAGE 64                     // work items a wave collects before it appends them to the frame's list with one atomic
struct WaveGatherLds { float4 cand[RT_WG_CAP]; float pay[RT_WG_CAP * 6]; uint32_t front[RT_WG_FRONT]; uint2 stage[RT_WG_STAGE]; };
// workgroups of k_gather_items per CU: as many as the staging buffers of their four waves leave room for in 160 KB of LDS
#define RT_GATHER_ITEM_WGS ((int)((160 * 1024) / (4 * sizeof(WaveGatherLds))) > 8 ? 8 : (int)((160 * 1024) / (4 * sizeof(WaveGatherLds))))

RT_D float wg_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; }
RT_D float wg_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }

// Work items of the split gather: the subtree at heap position `y` of map (x >> 31) for the 64 requests of wave
// (x & 0x7FFFFFFF) of the sorted request queue.
// A wave collects the items of its walks in LDS (stage, stage_n of them) and appends them together, and the frame's list is
// cut into RT_ITEM_SHARDS lists with a length word (and, for the consumer, a cursor word) each, a cache line apart: a word
// that every wave of the frame adds to -- ~10 ns per add, one after the other -- was the bottleneck of both kernels.
// items: shard s at items + s * cap (cap items each); count: shard s's length at count[s * RT_ITEM_WORDS].
#define RT_ITEM_SHARDS 64u
#define RT_ITEM_WORDS 32u                  // words between two shards' counters: 128 bytes
// the header of a frame's request queue: word 0 counts the later requests, then a length word and a cursor word per shard
#define RT_GQ_HEADER_BYTES ((size_t)RT_ITEM_WORDS * (1u + 2u * RT_ITEM_SHARDS) * 4u)
struct GatherItems { uint2 *items; uint32_t *count; uint32_t cap; uint32_t tag; uint2 *stage; uint32_t stage_n; bool full; };
// Appends the wave's staged items to the frame's list (all lanes call).  Items the list has no room for stay staged, at the
// front, and `full` is set: nothing more is appended, the caller serves what is left itself.
RT_D void gather_items_flush(GatherItems &g) {
  if (g.stage_n == 0 || g.full) return;
  const uint32_t lane = threadIdx.x & 63u, n = g.stage_n;
  RT_WAVE_SYNC()
  const uint32_t shard = (blockIdx.x * 4u + (threadIdx.x >> 6)) % RT_ITEM_SHARDS;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(g.count + shard * RT_ITEM_WORDS, n);
  base = __shfl(base, 0, 64);
  const uint32_t room = base < g.cap ? min(g.cap - base, n) : 0u;
  const uint2 mine = lane < n ? g.stage[lane] : make_uint2(0u, 0u);          // (RT_WG_STAGE <= 64: one item a lane)
  if (lane < room) g.items[(size_t)shard * g.cap + base + lane] = mine;
  RT_WAVE_SYNC()
  if (lane >= room && lane < n) g.stage[lane - room] = mine;
  g.stage_n = n - room; g.full = room < n;
  RT_WAVE_SYNC()
}

// Every lane of the wave must call (need = false: the lane has no point).  false: the points are too far apart for one
// walk to pay (max_area_ratio > 0 only) -- nothing was computed and the caller gathers lane by lane.
// The walk covers the subtree at heap position `start`.  With `split`, subtrees at depth split_depth are not walked but
// appended to the item queue (tagged split->tag) for other waves to serve -- a gather next to a lamp has a million
// photons in its ball, and one wave takes 64 nodes per global round trip.
RT_D bool gather_map_wave(const DevMap &M, bool need, V3<float> x, V3<float> nrm, double r, float max_area_ratio, WaveGatherLds &L, V3<float> &sum,
                          uint32_t start = 1u, int split_depth = -1, GatherItems *split = nullptr) {
  const int lane = threadIdx.x & 63;
  sum = mk<float>(0.f, 0.f, 0.f);
  if (!__any(need)) return true;
  const uint32_t n = M.n;
  const float inf = 3.4e38f;
  // box of the balls, rounded outwards (the walk only has to find a superset)
  const float lo0 = wg_min(need ? __double2float_rd((double)x.x - r) : inf), lo1 = wg_min(need ? __double2float_rd((double)x.y - r) : inf),
              lo2 = wg_min(need ? __double2float_rd((double)x.z - r) : inf);
  const float hi0 = wg_max(need ? __double2float_ru((double)x.x + r) : -inf), hi1 = wg_max(need ? __double2float_ru((double)x.y + r) : -inf),
              hi2 = wg_max(need ? __double2float_ru((double)x.z + r) : -inf);
  {
    // photons lie on surfaces: the walk's work goes with the box's two largest extents, a lane's own with pi r^2
    const float b0 = hi0 - lo0, b1 = hi1 - lo1, b2 = hi2 - lo2;
    const float bmax = fmaxf(b0, fmaxf(b1, b2)), bmin = fminf(b0, fminf(b1, b2));
    const float area = bmax * (b0 + b1 + b2 - bmax - bmin);
    if (max_area_ratio > 0.f && area > max_area_ratio * 3.14159265f * (float)(r * r)) return false;
  }
  const double r2 = r * r;
  const float r2lo = (float)(r2 * (1.0 - 1e-6)), r2hi = (float)(r2 * (1.0 + 1e-6));      // fp32 d2 is within 3e-7 (relative) of the fp64 one
  // a second, rounder bound on where the wave's photons lie: within r of a point that is within `reach` of the box's
  // centre (the corners of the box hold no photon any lane wants); padded well beyond fp32 rounding
  const float cx = 0.5f * (lo0 + hi0), cy = 0.5f * (lo1 + hi1), cz = 0.5f * (lo2 + hi2);
  float reach = 0.f;
  if (need) { const float ex = x.x - cx, ey = x.y - cy, ez = x.z - cz; reach = sqrtf(ex * ex + ey * ey + ez * ez); }
  reach = wg_max(reach);
  const float far2 = ((float)r + reach) * ((float)r + reach) * 1.0001f + 1e-12f;
  float sr = 0.f, sg = 0.f, sb = 0.f;
  uint32_t m = 0;
  // (a lane without a point: bounds no distance is below)
  // fp32 decides for a distance below or above band_mid (r^2) by more than band_half; one within that band may fall on
  // either side of r^2 in fp64 and sends the batch through the fp64 test (a lane without a point: no distance is below
  // its bound, none in its band)
  const float band_mid = need ? 0.5f * (r2lo + r2hi) : -3.0e38f, band_half = need ? 0.5f * (r2hi - r2lo) : -1.f;
  const float lo_lane = need ? band_mid : -1.f;
  auto consume = [&]() {
    RT_WAVE_SYNC()
    // the batch in fp32; should any (lane, candidate) distance fall inside the band, the batch again with the fp64 test
    // (tried: candidates outside every lane's ball skip their payload reads and sums -- 6.04 against 5.91 ms per frame)
    float br = 0.f, bg = 0.f, bb = 0.f;
    float nearest = 3.4e38f;                                 // smallest |d^2 - band_mid| of the batch: one subtract and one minimum per candidate
#ifdef RT_GATHER_PROBE
    uint32_t inb = 0;
    for (uint32_t j = 0; j < m; j++) { const float4 cd = L.cand[j]; const float dx = x.x - cd.x, dy = x.y - cd.y, dz = x.z - cd.z; inb += (dx * dx + dy * dy + dz * dz < lo_lane) ? 1u : 0u; }
    atomicAdd(&g_gather_probe[3], (unsigned long long)inb);
    if (lane == 0) { atomicAdd(&g_gather_probe[2], (unsigned long long)m); atomicAdd(&g_gather_probe[5], 1ull); }
#endif
    // (the count is the same in every lane: as a scalar it keeps the loop's counter and addresses out of the vector unit;
    // four candidates per trip: their LDS reads go out together and the running minimum is touched once)
    const uint32_t mu = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    auto one = [&](uint32_t j) -> float {                 // candidate j into the sums; returns |d^2 - band_mid|
      const float4 cd = L.cand[j];
      const float dx = x.x - cd.x, dy = x.y - cd.y, dz = x.z - cd.z;
      const float d2f = dx * dx + dy * dy + dz * dz;
      const float *pl = L.pay + j * 6;
      const float w = fmaxf(pl[0] * nrm.x + pl[1] * nrm.y + pl[2] * nrm.z, 0.f);
      const float wi = d2f < lo_lane ? w : 0.f;
      br += pl[3] * wi; bg += pl[4] * wi; bb += pl[5] * wi;
      return fabsf(d2f - band_mid);
    };
    uint32_t j = 0;
    for (; j + 4u <= mu; j += 4u) {
      const float t0 = one(j), t1 = one(j + 1u), t2 = one(j + 2u), t3 = one(j + 3u);
      nearest = fminf(nearest, fminf(fminf(t0, t1), fminf(t2, t3)));
    }
    for (; j < mu; j++) nearest = fminf(nearest, one(j));
    if (__any(nearest <= band_half)) {                   // (wave-uniform; rare)
      br = 0.f; bg = 0.f; bb = 0.f;
      for (uint32_t j = 0; j < m; j++) {
        const float4 cd = L.cand[j];
        const bool in = need && sq_dist((double)x.x, (double)x.y, (double)x.z, (double)cd.x, (double)cd.y, (double)cd.z) < r2;
        const float *pl = L.pay + j * 6;
        const float w = fmaxf(pl[0] * nrm.x + pl[1] * nrm.y + pl[2] * nrm.z, 0.f);
        const float wi = in ? w : 0.f;
        br += pl[3] * wi; bg += pl[4] * wi; bb += pl[5] * wi;
      }
    }
    sr += br; sg += bg; sb += bb;
    m = 0;
    RT_WAVE_SYNC()
  };
  RT_WAVE_SYNC()                                       // (the buffers may still be read by a gather before this one)
  if (lane == 0) L.front[0] = start;
  RT_WAVE_SYNC()
  uint32_t head = 0, tail = 1;
  bool overflow = false;
#ifdef RT_GATHER_PROBE
  { const unsigned long long nb_ = __ballot(need); if (lane == 0) { atomicAdd(&g_gather_probe[0], 1ull); atomicAdd(&g_gather_probe[4], (unsigned long long)__popcll(nb_)); } }
#endif
  while (head != tail) {
#ifdef RT_GATHER_PROBE
    if (lane == 0) atomicAdd(&g_gather_probe[1], 1ull);
#endif
    if (m + 64 > RT_WG_CAP) consume();
    const uint32_t size = tail - head;
    // a wide level: take the deepest nodes, and only as many as leave room for their children; within 64 entries of a full
    // queue one node at a time (depth first: the queue then grows by at most one entry per level of the tree below it),
    // so the queue cannot overflow
    const bool lifo = size > RT_WG_FRONT / 2;
    const uint32_t take = lifo ? (size + 64u >= RT_WG_FRONT ? 1u : min(64u, max(1u, (RT_WG_FRONT - 64u - size) / 2u))) : min(64u, size);
    const bool active = (uint32_t)lane < take;
    const uint32_t pos = active ? L.front[((lifo ? tail - take : head) + lane) % RT_WG_FRONT] : 1u;      // 1 <= pos <= n by construction
    if (lifo) tail -= take; else head += take;
    const float4 nd = M.node[pos - 1];
    const float fx = nd.x - cx, fy = nd.y - cy, fz = nd.z - cz;
    const bool inside = active && nd.x >= lo0 && nd.x <= hi0 && nd.y >= lo1 && nd.y <= hi1 && nd.z >= lo2 && nd.z <= hi2 && fx * fx + fy * fy + fz * fz <= far2;
    const unsigned long long mi = __ballot(inside);
    if (inside) {
      const uint32_t at = m + (uint32_t)__popcll(mi & (((unsigned long long)1 << lane) - 1));
      L.cand[at] = nd;
      const float2 *pp = (const float2 *)(M.pay + (size_t)(pos - 1) * 6);      // 24-byte records: 8-byte aligned
      const float2 p0 = pp[0], p1 = pp[1], p2 = pp[2];
      float *dst = L.pay + at * 6;
      dst[0] = p0.x; dst[1] = p0.y; dst[2] = p1.x; dst[3] = p1.y; dst[4] = p2.x; dst[5] = p2.y;
    }
    m += (uint32_t)__popcll(mi);
    const int dim = node_dim(nd);
    const float pd = (dim == 0) ? nd.x : (dim == 1) ? nd.y : nd.z;
    const float lod = (dim == 0) ? lo0 : (dim == 1) ? lo1 : lo2, hid = (dim == 0) ? hi0 : (dim == 1) ? hi1 : hi2;
    // kdtree.js:70-73 for a box: left only if the box ends before the plane, right only if it starts after it
    bool gl = active && 2ull * pos <= n && !(lod > pd);
    bool gr = active && 2ull * pos + 1 <= n && !(hid < pd);
    const bool at_split = split && active && (31 - __clz((int)pos)) + 1 == split_depth;     // (the nodes of a step may lie on different levels)
    if (__any(at_split && (gl || gr))) {                                               // (wave-uniform)
      // these children are other waves' work: one item each, staged in LDS and appended to the frame's list a stageful at
      // a time.  Lower children, then upper children: at most 64 of either, so a (flushed) stage always has room -- only a
      // frame whose list is full keeps children for this wave to walk
      for (int side = 0; side < 2; side++) {
        const bool child = at_split && (side ? gr : gl);
        const unsigned long long mc_ = __ballot(child);
        if (mc_ == 0ull) continue;
        const uint32_t cnt = (uint32_t)__popcll(mc_);
        if (split->stage_n + cnt > RT_WG_STAGE) gather_items_flush(*split);
        const uint32_t at = split->stage_n + (uint32_t)__popcll(mc_ & (((unsigned long long)1 << lane) - 1));
        if (child && at < RT_WG_STAGE) { split->stage[at] = make_uint2(split->tag, 2 * pos + (uint32_t)side); if (side) gr = false; else gl = false; }
        split->stage_n = min(split->stage_n + cnt, (uint32_t)RT_WG_STAGE);
      }
    }
    const unsigned long long ml = __ballot(gl), mr = __ballot(gr);
    const uint32_t cl = (uint32_t)__popcll(ml), cr = (uint32_t)__popcll(mr);
    if (tail - head + cl + cr > RT_WG_FRONT) { overflow = true; break; }
    RT_WAVE_SYNC()
    if (gl) L.front[(tail + (uint32_t)__popcll(ml & (((unsigned long long)1 << lane) - 1))) % RT_WG_FRONT] = 2 * pos;
    if (gr) L.front[(tail + cl + (uint32_t)__popcll(mr & (((unsigned long long)1 << lane) - 1))) % RT_WG_FRONT] = 2 * pos + 1;
    tail += cl + cr;
    RT_WAVE_SYNC()
  }
#ifdef RT_GATHER_PROBE
  if (overflow && lane == 0) atomicAdd(&g_gather_probe[6], 1ull);
#endif
  if (overflow) return false;
  consume();
  sum = mk<float>(sr, sg, sb);
  return true;
}

template <typename T> RT_D void finish_path(T *out, V3<T> acc, V3<T> term, T term_a, bool alpha_fixed) {
  out[0] = acc.x + term.x; out[1] = acc.y + term.y; out[2] = acc.z + term.z;
  out[3] = alpha_fixed ? T(1) : term_a;
}


// One traceRay body (rayphoton.js:1651-1838) for path p, whose segment `st` hit primitive `idx` at distance t.
// A path either finishes (its colour goes to out[0..3]) or continues: then `st` holds the new segment and
// true is returned.  FULL = false compiles the direct-light and photon-gather terms out (for calls that
// switch both off); the remaining control flow and RNG draws are the same.
template <typename T, bool FULL>
RT_D bool shade_body(const DevScene<T> &S, const ScanSrc<T> &G, const TraceCfg &C, const DevMap &Md, const DevMap &Mc,
                     uint32_t p, PathState<T> &st, T t, int idx, int bounce, T *out, ShadeTally &tally) {
  bool alive = false;
  const T eps = T(C.epsilon);
  const int depth = st.depth & 0xFFFF; const bool alpha_fixed = st.depth & 0x10000;
  const V3<T> ZERO = mk<T>(0, 0, 0);
  if (idx < 0 || idx >= (int)S.n_shapes) {
    finish_path<T>(out, st.acc, mk<T>(16, 16, 16), T(1), alpha_fixed);                 // COL_VERY_DARK_GREY :1838
  } else if (const HitRec<T> H = load_hit<T>(S, G, idx); H.flags() & RT_FLAG_LIGHT) {     // :1654-1656
    int l = 0;
    for (uint32_t k = 0; k < S.n_lights; k++) if ((int)S.light_shape[k] == idx) l = (int)k;
    finish_path<T>(out, st.acc, mk<T>(S.light_data[l][3], S.light_data[l][4], S.light_data[l][5]), S.light_data[l][6], alpha_fixed);
  } else {
    const V3<T> d = st.d;
    const V3<T> hit = st.o + t * d;
    const V3<T> local = shape_colour_g<T>(H.geom, H.colour, H.proc(), hit);
    const V3<T> n = shape_normal_g<T>(H.geom, H.type(), H.flags(), hit);
    if (t < st.max_dist || depth < C.max_depth) {                                      // :1677 (note: OR)
      const T cos1 = -dot(d, n);
      const int mat = H.mat_byte() & 7;
      const bool has_mat = H.mat_byte() < 8;
      RngStream rng = path_rng<T>(S.width, C, p, st);
      if (has_mat && H.transparent == T(1)) {                                          // transparent :1679
        MatStack ms; ms.bits = st.stack;
        V3<T> ndir; bool ovf = false;
        if (!fresnel_bounce<T>(S, mat, d, n, cos1, ms, rng, ndir, ovf)) {
          finish_path<T>(out, st.acc, ZERO, T(0), alpha_fixed);                        // COL_BLACK :1699
        } else {
          if (ovf) tally.overflows++;
          st.o = hit; st.d = normalize<T>(ndir, eps);
          st.max_dist = st.max_dist - t; st.depth = st.depth + 1; st.stack = ms.bits; st.draw = rng.draw;
          alive = true;                                                                // importance unchanged, >= 0.01
        }
      } else if (cos1 < T(0)) {
        finish_path<T>(out, st.acc, ZERO, T(1), alpha_fixed);                          // inside an opaque object :1736-1747
      } else {
        const T refl = H.reflectance;                                                  // (NaN without a material)
        V3<T> direct = ZERO;
        if (FULL && refl < T(1) && C.use_direct_light) {
          const T bias = T(C.shadow_bias);
          direct = direct_light<T>(S, G, C, hit + bias * n, n, local, rng, tally.shadow_rays, &tally.tests);     // :1750-1752
        }
        bool spec = false;
        if (rng.next_k() < H.refl_k) spec = rng.next_k() < H.spec_k;                  // :1753, short-circuit
        V3<T> acc = st.acc;
        if (FULL) {
          V3<T> diff = ZERO, caus = ZERO;
          if (C.use_photon_maps) {
            int todo = (Md.n ? 1 : 0) | (Mc.n ? 2 : 0);                // maps still to be gathered for this hit
            if constexpr (sizeof(T) == 4) {
              if (todo && C.gq) {
                // one slot per lane that is here, reserved by the first of them
                const int lane = threadIdx.x & 63;
                uint32_t slot = p;                                                         // a path's first hit: its own slot
                if (bounce != 0 || p >= C.gq_first) {                                      // (wave-uniform in practice: a pass traces one bounce)
                  const unsigned long long later = __ballot(true);
                  const int leader2 = __ffsll((long long)later) - 1;
                  uint32_t base = 0;
                  if (lane == leader2) base = atomicAdd(C.gq_count, (uint32_t)__popcll(later));
                  base = __shfl(base, leader2, 64);
                  slot = C.gq_first + base + (uint32_t)__popcll(later & (((unsigned long long)1 << lane) - 1));
                }
                if (slot < C.gq_cap) {                                                     // (else: gathered here, below)
                  C.gq_valid[slot] = 1u;
                  float4 *q = C.gq + (size_t)slot * 3;
                  q[0] = make_float4(hit.x, hit.y, hit.z, __uint_as_float(p));
                  q[1] = make_float4(n.x, n.y, n.z, __uint_as_float((uint32_t)todo | ((uint32_t)bounce << 2)));
                  q[2] = make_float4(local.x / T(255), local.y / T(255), local.z / T(255), 0.f);
                  todo = 0;
                }
              }
            }
            if (todo & 1) diff = gather_map<T>(Md, hit, n, C.diffuse_radius);
            if (todo & 2) caus = gather_map<T>(Mc, hit, n, C.caustic_radius);
            tally.gathers += (Md.n ? 1 : 0) + (Mc.n ? 1 : 0);
          }
          const T PI_ = T(3.141592653589793);
          const T dArea = PI_ * T(C.diffuse_radius * C.diffuse_radius), cArea = PI_ * T(C.caustic_radius * C.caustic_radius);
          V3<T> lc = mk<T>(local.x / T(255), local.y / T(255), local.z / T(255));
          diff = mk<T>(diff.x * lc.x / dArea, diff.y * lc.y / dArea, diff.z * lc.z / dArea);      // :1808-1815
          caus = mk<T>(caus.x * lc.x / cArea, caus.y * lc.y / cArea, caus.z * lc.z / cArea);
          V3<T> term = ((T(1) - refl) * direct + diff) + caus;                                    // :1817-1822 minus the child
          acc = acc + term;
        } else if (!(refl == refl)) {
          acc = mk<T>(refl, refl, refl);                                                          // (1 - undefined) * direct is NaN
        }
        const T imp = st.importance * refl;
        if (spec && !(imp < T(0.01))) {                                                           // child would return at :1627
          st.o = hit; st.d = normalize<T>(d + (T(2) * cos1) * n, eps);
          st.max_dist = st.max_dist - t; st.importance = imp; st.depth = (st.depth + 1) | 0x10000;
          st.acc = acc; st.draw = rng.draw;
          alive = true;
        } else {
          finish_path<T>(out, acc, ZERO, T(1), true);
        }
      }
    } else {
      finish_path<T>(out, st.acc, ZERO, T(1), alpha_fixed);                            // :1828-1833
    }
  }
  if (alive && bounce + 1 >= C.max_bounces) {                                          // safety cap; the reference has none
    finish_path<T>(out, st.acc, ZERO, T(1), (st.depth & 0x10000) != 0);
    alive = false;
  }
  return alive;
}

// ------------------------------------------------------------------ the older tracer (ray.js:801-993)
// vecPerturb of ray.js:75-100: the frame is not normalised and the lower bound is cos(pi/2)^2. 2 draws.
template <typename T> RT_D V3<T> vec_perturb_a(V3<T> v, T eps, RngStream &rng) {
  V3<T> m = mk<T>(1, 0, 0);
  if (sqlen(cross(v, m)) < eps) m = mk<T>(0, 1, 0);
  V3<T> n = cross(v, m);
  m = cross(n, v);
  const T cl = rt_cos(T(3.141592653589793) / T(2));
  const T lower = cl * cl;
  T x = (T(1) - lower) * rng.next<T>() + lower;
  T cosT = rt_sqrt(x);
  T sinT = rt_sqrt(T(1) - cosT * cosT);
  T sinP, cosP; sincos_turns(rng.next<T>(), sinP, cosP);
  return (cosT * v + (sinT * cosP) * m) + (sinT * sinP) * n;
}

// ray.js's traceRay blends at every level: colour = clamp(round((1 - refl) * shapeCol + refl * child)).
// A path therefore leaves one record per level -- (1 - refl) * shapeCol (rgb) and refl -- in
// levels[(level * n_paths + p) * 4 ..], ends with a terminal colour in P.out and its level count in
// P.depth; k_resolve_a substitutes the terminal colour back up through the records.
// Returns true when the path goes on with a new segment.
template <typename T>
RT_D bool shade_body_a(const DevScene<T> &S, const ScanSrc<T> &G, const TraceCfg &C, T *levels, uint32_t p, PathState<T> &st, T t, int idx, T *out,
                       int32_t *depth_out, ShadeTally &tally) {
  const T eps = T(C.epsilon);
  const int level = st.depth;
  if (idx < 0 || idx >= (int)S.n_shapes) {                                       // ray.js:992 (the second test is never true)
    out[0] = T(16); out[1] = T(16); out[2] = T(16); out[3] = T(1); *depth_out = level;
    return false;
  }
  if (level < 0 || level > C.max_depth) {                                        // (never true; guards the record store)
    out[0] = T(0); out[1] = T(0); out[2] = T(0); out[3] = T(1); *depth_out = 0; tally.overflows++;
    return false;
  }
  const V3<T> d = st.d;
  const V3<T> hit = st.o + t * d;
  const HitRec<T> H = load_hit<T>(S, G, idx);
  const V3<T> sc = shape_colour_g<T>(H.geom, H.colour, H.proc(), hit);
  const V3<T> n = shape_normal_g<T>(H.geom, H.type(), H.flags(), hit);
  const T refl = H.surf_refl;
  T *rec = levels + ((size_t)level * C.n_paths + p) * 4;
  if (t < st.max_dist && level < C.max_depth) {                                  // ray.js:893, depth = max_depth - level
    const T cos1 = -dot(d, n);
    const int mat = H.mat_byte() & 7;
    const bool has_mat = H.mat_byte() < 8;
    RngStream rng = path_rng<T>(S.width, C, p, st);
    V3<T> ndir;
    if (!(has_mat && H.transparent == T(1))) {                                   // MAT_OPAQUE ray.js:895
      if (cos1 < T(0)) {                                                         // ray.js:897-907: undefined, or black far away
        const T v = (t < T(1e5)) ? Limits<T>::nan() : T(0);
        out[0] = v; out[1] = v; out[2] = v; out[3] = v; *depth_out = level;
        return false;
      }
      const V3<T> rdir = d + (T(2) * cos1) * n;
      ndir = rdir;
      if (rng.next_k() >= H.spec_gt_k) {                                    // Math.random() > specular, ray.js:910-923
        for (int attempt = 0; attempt < 100; attempt++) {
          ndir = vec_perturb_a<T>(rdir, eps, rng);
          if (dot(ndir, n) > eps) break;
          if (attempt == 99) ndir = rdir;
        }
      }
    } else {                                                                     // ray.js:925-977, no Fresnel term
      MatStack ms; ms.bits = st.stack;
      const T eta1 = ms.len() >= 1 ? pick8<T>(S.mat[1], ms.top()) : Limits<T>::nan();
      const T eta2 = (dot(d, n) < T(0)) ? H.refr_index : (ms.len() >= 2 ? pick8<T>(S.mat[1], ms.at(ms.len() - 2)) : Limits<T>::nan());
      const T ratio = eta1 / eta2;                                               // an undefined index flows on as NaN
      const T c2sq = T(1) - ratio * ratio * (T(1) - cos1 * cos1);
      if (c2sq < T(0)) ndir = d + (T(2) * cos1) * n;
      else {
        const T pm = (cos1 < T(0)) ? T(-1) : T(1);
        ndir = ratio * d + (ratio * cos1 - pm * rt_sqrt(c2sq)) * n;
        if (cos1 < T(0)) { if (ms.len() > 0) ms.pop(); }
        else if (!ms.push(mat)) tally.overflows++;
        st.stack = ms.bits;
      }
    }
    rec[0] = (T(1) - refl) * sc.x; rec[1] = (T(1) - refl) * sc.y; rec[2] = (T(1) - refl) * sc.z; rec[3] = refl;
    st.depth = level + 1;
    const T imp = st.importance * refl;
    if (imp < T(0.01)) {                                                         // the child returns COL_WHITE, ray.js:845-847
      out[0] = T(255); out[1] = T(255); out[2] = T(255); out[3] = T(1); *depth_out = level + 1;
      return false;
    }
    st.o = hit; st.d = normalize<T>(ndir, eps);
    st.max_dist = st.max_dist - t; st.importance = imp; st.draw = rng.draw;
    return true;
  }
  rec[0] = (T(1) - refl) * sc.x; rec[1] = (T(1) - refl) * sc.y; rec[2] = (T(1) - refl) * sc.z; rec[3] = refl;
  out[0] = sc.x; out[1] = sc.y; out[2] = sc.z; out[3] = T(1); *depth_out = level + 1;    // transmittedColour = shapeCol, ray.js:891
  return false;
}

// one shade step of either tracer for path p; finished paths leave their colour in P.out (and, for the older
// tracer, their level count in P.depth), continuing paths leave the new segment in `st`
RT_D double clamp01(double x, double lo, double hi) { return (x < lo) ? lo : ((x > hi) ? hi : x); }

// what k_resolve does for a pixel with one sub-sample (rayphoton.js:1610-1621): clamp, floor; pre = the unclamped colour
template <typename T> RT_D void resolve_store(const TraceCfg &C, uint32_t pp, const T *o) {
  const uint32_t pix = pixel_out_index(C, pp);
  uint32_t packed = 0;
  for (int c = 0; c < 4; c++) {
    const T last = o[c], hi = (c == 3) ? T(1) : T(255);
    const T v = rt_floor((last < T(0)) ? T(0) : ((last > hi) ? hi : last));     // exact in either precision
    if (C.pre) C.pre[(size_t)pix * 4 + c] = (double)last;
    if (C.post) C.post[(size_t)pix * 4 + c] = (double)v;
    const T b = (c == 3) ? v * T(255) : v;                                       // 0..255, or NaN
    packed |= ((b == b) ? (uint32_t)b : 0u) << (8 * c);
  }
  if (C.rgba8) ((uint32_t *)C.rgba8)[pix] = packed;          // (the tile's byte image is 4-byte aligned: one store per pixel)
}

// one shade step of either tracer for path p; finished paths leave their colour in P.out (and, for the older
// tracer, their level count in P.depth) or, when the resolve is fused, in the tile's outputs; continuing paths
// leave the new segment in `st`
template <typename T, bool WHITTED, bool FULL>
RT_D bool shade_step(const DevScene<T> &S, const ScanSrc<T> &G, const PathSoA<T> &P, const TraceCfg &C, const DevMap &Md, const DevMap &Mc,
                     T *levels, uint32_t p, PathState<T> &st, T t, int idx, int bounce, ShadeTally &tally) {
  if (WHITTED) return shade_body_a<T>(S, G, C, levels, p, st, t, idx, P.out + (size_t)p * 4, P.depth + p, tally);
  T o4[4];
  const bool alive = shade_body<T, FULL>(S, G, C, Md, Mc, p, st, t, idx, bounce, o4, tally);
  if (!alive) {
    if (C.fuse_resolve) resolve_store<T>(C, p, o4);
    else { T *out = P.out + (size_t)p * 4; out[0] = o4[0]; out[1] = o4[1]; out[2] = o4[2]; out[3] = o4[3]; }
  }
  return alive;
}

// Which primitives can the primary rays of this workgroup reach?  Its paths are consecutive, so their pixels span a
// few columns of one row (or whole rows); the rays through that rectangle lie inside a four-sided pyramid with its
// apex at the camera, and a primitive whose bounding sphere lies outside one of the four planes cannot be hit by any
// of them.  Wave 0 tests the primitives, 64 at a time, and compacts the survivors into vis[] / vseg[] (the slots are
// sorted by type, so one pass in slot order keeps the types together).  first64: the bounding sphere of slot `lane`,
// which the caller loads before it stages the scene so that the round trip is not paid here.
// Conservative (padded spheres, the rectangle widened by a pixel), so the scan over the list finds the same hits.
// wg: the workgroup of the first pass (wg_size consecutive paths) the list is for; called by one whole wave.
struct TilePyramid { V3<float> o, nl, nr, nb, nt; };      // apex and the four inward unit normals
RT_D TilePyramid tile_pyramid(const DevScene<float> &S, const TraceCfg &C, uint32_t wg, uint32_t wg_size) {
  const uint32_t p0 = wg * wg_size, p1 = min(p0 + wg_size, C.n_paths) - 1;
  uint32_t r_, c0, c1;
  const uint32_t ss = (uint32_t)(C.sub_sample * C.sub_sample);
  const uint32_t pix0 = div_by(p0, ss, C.magic_ss, r_), pix1 = div_by(p1, ss, C.magic_ss, r_);
  uint32_t row0, row1;
  pixel_colrow(C, pix0, c0, row0); pixel_colrow(C, pix1, c1, row1);
  if (C.tiles_x) {
    // pixels pix0 .. pix1 of one tile are whole rows of it unless they lie in one row; across tiles: no culling
    if ((pix0 >> 8) != (pix1 >> 8)) { c0 = 0; c1 = (uint32_t)C.w - 1; row0 = 0; row1 = (uint32_t)C.h - 1; }
    else if (row0 != row1) { c0 &= ~31u; c1 = c0 + 31u; }
  } else if (row0 != row1) { c0 = 0; c1 = (uint32_t)C.w - 1; }
  const float jit = (C.block > 1 ? (float)C.block : 0.0f) + 1.0f;        // block jitter + sub-sample jitter, in pixels
  const float xa = (float)(C.x0 + (int)c0 * C.block) - 1.0f, xb = (float)(C.x0 + (int)c1 * C.block) + jit + 1.0f;
  const float ya = (float)(C.y0 + (int)band_row(C, row0) * C.block) - 1.0f, yb = (float)(C.y0 + (int)band_row(C, row1) * C.block) + jit + 1.0f;
  const float ua = (xa * 2.0f / (float)S.width - 1.0f) * S.camera[12], ub = (xb * 2.0f / (float)S.width - 1.0f) * S.camera[12];
  const float vb = -(ya * 2.0f / (float)S.height - 1.0f) * S.camera[13], va = -(yb * 2.0f / (float)S.height - 1.0f) * S.camera[13];
  const V3<float> cu = ld3(S.camera + 3), cv = ld3(S.camera + 6), cw = ld3(S.camera + 9);
  auto dir = [&](float u, float v) { return (u * cu + v * cv) + (-1.0f) * cw; };
  const V3<float> d00 = dir(ua, va), d10 = dir(ub, va), d01 = dir(ua, vb), d11 = dir(ub, vb), dc = dir(0.5f * (ua + ub), 0.5f * (va + vb));
  auto plane = [&](V3<float> a, V3<float> b) {                 // unit normal of the plane through the apex, a and b, facing the pyramid's axis
    V3<float> nn = cross(a, b);
    const float l = rt_sqrt(sqlen(nn));
    const float sgn = dot(nn, dc) < 0.0f ? -1.0f : 1.0f;
    return (l > 0.0f ? sgn / l : 0.0f) * nn;                    // (degenerate: zero normal, never culls)
  };
  TilePyramid py;
  py.o = ld3(S.camera); py.nl = plane(d00, d01); py.nr = plane(d10, d11); py.nb = plane(d00, d10); py.nt = plane(d01, d11);
  return py;
}
RT_D void cull_primitives_wave(const DevScene<float> &S, const TraceCfg &C, uint32_t wg, uint32_t wg_size, uint16_t *vis, uint32_t *vseg, float4 first64) {
  {
    const int lane = threadIdx.x & 63;
    const TilePyramid py = tile_pyramid(S, C, wg, wg_size);
    const V3<float> o = py.o, nl = py.nl, nr = py.nr, nb = py.nb, nt = py.nt;
    uint32_t total = 0;
    uint32_t below[RT_NUM_TYPES] = {};                            // kept slots below the first slot of each type
    for (uint32_t base = 0; base < S.n_shapes; base += 64) {
      const uint32_t i = base + lane;
      bool keep = false;
      if (i < S.n_shapes) {
        const float4 bs = base == 0 ? first64 : S.sbound[i];
        const V3<float> c = mk<float>(bs.x, bs.y, bs.z) - o;
        const float R = bs.w;                                      // +inf (or NaN): never culled
        keep = !(dot(nl, c) < -R || dot(nr, c) < -R || dot(nb, c) < -R || dot(nt, c) < -R);
      }
      const unsigned long long m = __ballot(keep);
      if (keep) vis[total + (uint32_t)__popcll(m & (((unsigned long long)1 << lane) - 1))] = (uint16_t)i;
      total += (uint32_t)__popcll(m);
#pragma unroll
      for (int t = 0; t < RT_NUM_TYPES; t++) {
        const uint32_t e = S.seg[t];                               // lanes of this round with a slot below e
        const unsigned long long lt = e <= base ? 0ull : (e - base >= 64 ? ~0ull : (((unsigned long long)1 << (e - base)) - 1));
        below[t] += (uint32_t)__popcll(m & lt);
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < RT_NUM_TYPES; t++) vseg[t] = below[t];
    }
    if (lane == 0) vseg[RT_NUM_TYPES] = total;
  }
}
RT_D void cull_primitives(const DevScene<float> &S, const TraceCfg &C, uint16_t *vis, uint32_t *vseg, float4 first64) {
  if (threadIdx.x < 64) cull_primitives_wave(S, C, blockIdx.x, blockDim.x, vis, vseg, first64);      // wave 0 does it all; the others wait
  __syncthreads();
}
// The lists of a whole frame, one wave per workgroup of the first pass, into lists[wg * stride ...] (the layout of the
// LDS copy: 16 segment words, then the slots).  They depend on the scene, its camera and the tile's layout only, so the
// host keeps them from frame to frame.
static __global__ void __launch_bounds__(256) k_cull_lists(DevScene<float> S, TraceCfg C, uint32_t n_wg, uint32_t wg_size, uint32_t *lists, uint32_t stride) {
  const uint32_t wg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wg >= n_wg) return;
  const int lane = threadIdx.x & 63;
  uint32_t *rec = lists + (size_t)wg * stride;
  const float4 first64 = (uint32_t)lane < S.n_shapes ? S.sbound[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  cull_primitives_wave(S, C, wg, wg_size, (uint16_t *)(rec + 16), rec, first64);
}

// First pass of a frame, fused: camera ray, closest hit and shade of every path with the path state in
// registers; only paths that go on are written to the path arrays.  Survivors are recorded as one keep-mask
// per wave plus one count per workgroup; k_scan_blocks and k_queue_scatter then build the next queue in order.
// In the fp32 build every test's origin-dependent part is staged once per workgroup (all rays start at the camera).
template <typename T, bool LDS, bool WHITTED, bool FULL>
__global__ void __launch_bounds__(256) k_primary(DevScene<T> S, PathSoA<T> P, TraceCfg C, DevMap Md, DevMap Mc, T *levels,
                                                 uint32_t *queue, uint32_t *count_ptr, unsigned long long *wave_mask,
                                                 uint32_t *block_count, TraceCounters *ctr) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  constexpr bool OT = LDS && sizeof(T) == 4;
  const uint32_t count = C.n_paths;
  if (blockIdx.x == 0 && threadIdx.x == 0) *count_ptr = count;          // the queue length the later passes read
  const V3<T> cam_o = ld3(S.camera);
  // fp32: only the primitives this workgroup's rays can reach are scanned (the hierarchy, when present, does that job)
  bool culled = false; uint16_t *vis = nullptr; uint32_t *vseg = nullptr;
  float4 first64 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (OT) {
    culled = S.sbound && !S.bvh;
    if (culled) { vseg = scan_lds_vseg(S.n_shapes, rt_lds); vis = scan_lds_vis(S.n_shapes, rt_lds); }
    if (culled && !C.cull_lists && threadIdx.x < 64 && threadIdx.x < S.n_shapes) first64 = S.sbound[threadIdx.x];
  }
  const bool listed = OT && culled && C.cull_lists;           // this workgroup's list comes ready-made, staged with the scene
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds, OT ? &cam_o : nullptr, listed ? C.cull_lists + (size_t)blockIdx.x * C.cull_stride : nullptr,
                                                vseg, listed ? C.cull_stride : 0u, false)      // (one shade per lane: staging the hit records costs more than it saves)
                           : global_scan_src<T>(S);
  if constexpr (OT) {
    if (culled && !listed) cull_primitives(S, C, vis, vseg, first64);
  }
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  bool alive = false;
  ShadeTally tally;
  PathState<T> st;
  T t = T(0); int idx = -1;
  if (p < count) {
    if (!C.subq_cap) queue[p] = p;                                       // identity queue for the ordered compaction
    st = raygen_path<T>(S, C, p);
    if constexpr (OT) {
      if (culled) idx = scan_fast_visible<WHITTED ? 3 : 0>(G, vis, vseg, st.d, (float)C.epsilon, t);
      else idx = scan_scene<T, WHITTED ? 3 : 0, OT>(S, G, st.o, st.d, T(C.epsilon), T(0), t, &tally.tests);
    } else {
      idx = scan_scene<T, WHITTED ? 3 : 0, OT>(S, G, st.o, st.d, T(C.epsilon), T(0), t, &tally.tests);
    }
  }
  if (p < count) alive = shade_step<T, WHITTED, FULL>(S, G, P, C, Md, Mc, levels, p, st, t, idx, 0, tally);
  if (alive) store_path<T>(P, p, st);
  if constexpr (OT) {
    // tests executed by this workgroup's primary rays (workgroup-uniform list length): one atomic per workgroup
    if (culled && threadIdx.x == 0) atomicAdd(&ctr->prim_tests[(blockIdx.x % RT_CTR_SLOTS) * 16], (unsigned long long)vseg[RT_NUM_TYPES] * (min(blockIdx.x * blockDim.x + blockDim.x, count) - blockIdx.x * blockDim.x));
  }
  if (C.subq_cap) block_append_subqueue(alive, p, queue, C.subq_cap, ctr->subq_count[0], blockIdx.x / (C.subq_cap >> 8));
  else block_publish_masks(alive, wave_mask, block_count);
  flush_tally(tally, ctr);
}

// The same first pass with PERSISTENT workgroups (fp32, scene staged in LDS, ready-made lists of reachable primitives,
// survivors handed on through sub-queues): workgroup b traces the tiles -- blocks of 256 consecutive paths, what a
// workgroup of k_primary takes -- b * per_block .. (b + 1) * per_block - 1.  What a workgroup of k_primary pays per tile
// is paid here once per workgroup: the kernel's prologue, the staging of the scene (with the hit records, which one
// shade per lane could not pay for), the statistics' atomics, and the returning atomic that reserves the survivors'
// places in their sub-queue, on whose answer the whole workgroup waits -- the survivors of a workgroup's tiles are
// collected in LDS and appended when the sub-queue changes (or the buffer is full, or at the end).  The next tile's list
// is fetched while the current tile is traced.  Same rays, same hits, same shading; the order of the paths inside a
// sub-queue differs, which its consumer does not care about.
// Measured on the 1080p bench frame: alone on the chip 55 against 51 us (a workgroup's tiles run one after the other),
// with four frames in flight 79.4 against 81.3 us per frame -- so it is the first pass from three frames in flight on,
// like the tail kernel with persistent lanes.  (Tried on the latter: raised wave priority for waves that carry deep
// paths -- no difference; 8, 16, 32 queued paths per lane instead of 4 -- 89, 123, 183 us per frame.)
#define RT_PT_MAX_TILES 8
// Waves per SIMD the register allocation aims for.  The first pass at 8 (at most 64 registers instead of the 82 it takes
// when left alone: five waves): 0.0805 against 0.0822 ms per frame with four frames in flight, three runs each; the
// persistent tail left alone (71 registers, seven waves): capped at 8 it is 4 % slower.
#ifndef RT_PT_MIN_WAVES
#define RT_PT_MIN_WAVES 8
#endif
#ifndef RT_TP_MIN_WAVES
#define RT_TP_MIN_WAVES 1
#endif
template <bool WHITTED, bool FULL>
__global__ void __launch_bounds__(256, RT_PT_MIN_WAVES) k_primary_tiles(DevScene<float> S, PathSoA<float> P, TraceCfg C, DevMap Md, DevMap Mc, float *levels,
                                                       uint32_t *queue, uint32_t *count_ptr, TraceCounters *ctr, uint32_t n_tiles, uint32_t per_block) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  __shared__ uint32_t surv[RT_PT_MAX_TILES * 256];
  __shared__ uint32_t cnt[4];
  __shared__ uint32_t base_sh;
  const uint32_t count = C.n_paths, tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) *count_ptr = count;      // the queue length the later passes read
  const uint32_t t0 = blockIdx.x * per_block, t1 = min(n_tiles, t0 + per_block);
  if (t0 >= t1) return;                                     // (workgroup-uniform)
  const V3<float> cam_o = ld3(S.camera);
  uint32_t *vseg = scan_lds_vseg(S.n_shapes, rt_lds);
  const uint16_t *vis = scan_lds_vis(S.n_shapes, rt_lds);
  const ScanSrc<float> G = stage_scene_lds<float>(S, rt_lds, &cam_o, C.cull_lists + (size_t)t0 * C.cull_stride, vseg, C.cull_stride, true);
  const int lane = tid & 63, wv = tid >> 6;
  const uint32_t tiles_per_subq = C.subq_cap >> 8;
  ShadeTally tally;
  unsigned long long tests = 0;                             // (thread 0) ray-primitive tests of this workgroup's primary rays
  uint32_t n_surv = 0, cur_s = min(t0 / tiles_per_subq, (uint32_t)RT_SUBQ - 1u);      // (workgroup-uniform)
  auto flush = [&]() {                                      // all threads; appends surv[0 .. n_surv) to sub-queue cur_s
    if (n_surv == 0) return;
    if (tid == 0) base_sh = atomicAdd(&ctr->subq_count[0][cur_s * 16], n_surv);
    __syncthreads();
    const uint32_t base = base_sh;
    for (uint32_t k = tid; k < n_surv; k += 256) if (base + k < C.subq_cap) queue[(size_t)cur_s * C.subq_cap + base + k] = surv[k];
    __syncthreads();
    n_surv = 0;
  };
  for (uint32_t t = t0; t < t1; t++) {
    uint32_t next_word = 0;                                 // the next tile's list, in flight while this tile is traced
    if (t + 1 < t1 && tid < C.cull_stride) next_word = C.cull_lists[(size_t)(t + 1) * C.cull_stride + tid];
    const uint32_t s_t = min(t / tiles_per_subq, (uint32_t)RT_SUBQ - 1u);
    if (s_t != cur_s || n_surv + 256 > RT_PT_MAX_TILES * 256) { flush(); cur_s = s_t; }
    const uint32_t p = t * 256 + tid;
    bool alive = false;
    if (p < count) {
      PathState<float> st = raygen_path<float>(S, C, p);
      float th = 0.f;
      const int idx = scan_fast_visible<WHITTED ? 3 : 0>(G, vis, vseg, st.d, (float)C.epsilon, th);
      alive = shade_step<float, WHITTED, FULL>(S, G, P, C, Md, Mc, levels, p, st, th, idx, 0, tally);
      if (alive) store_path<float>(P, p, st);
    }
    if (tid == 0) tests += (unsigned long long)vseg[RT_NUM_TYPES] * (min(t * 256 + 256, count) - t * 256);
    const unsigned long long mask = __ballot(alive);
    if (lane == 0) cnt[wv] = (uint32_t)__popcll(mask);
    __syncthreads();                                        // (every thread is also done with this tile's list)
    uint32_t pos = n_surv + (uint32_t)__popcll(mask & (((unsigned long long)1 << lane) - 1));
    for (int w = 0; w < wv; w++) pos += cnt[w];
    if (alive) surv[pos] = p;
    n_surv += cnt[0] + cnt[1] + cnt[2] + cnt[3];
    if (t + 1 < t1 && tid < C.cull_stride) vseg[tid] = next_word;
    __syncthreads();
  }
  flush();
  if (tid == 0 && tests) atomicAdd(&ctr->prim_tests[(blockIdx.x % RT_CTR_SLOTS) * 16], tests);
  flush_tally(tally, ctr);
}

// ---- the first pass of scenes that are scanned through the hierarchy (too many primitives for LDS) ----
// All rays of a first-pass workgroup start at the camera and pass through one small rectangle of pixels, so the few
// primitives they can reach are known per workgroup and per camera: k_cull_lists_big tests every primitive's bounding
// sphere against every workgroup's pyramid once (the host keeps the lists from frame to frame, like the small scenes'),
// and k_primary_gathered copies its workgroup's primitives -- records, the camera's origin terms, flags -- into LDS and
// scans them there, every lane the same primitive at the same time: no walk, no divergence, no dependent global loads.
// lists: per workgroup `stride` words: 9 type bounds (word 8 = the length), 7 unused, then up to cap sorted slots.  A list
// longer than cap keeps its length (and no slots): that workgroup walks the hierarchy per ray as before.
static __global__ void __launch_bounds__(256) k_cull_lists_big(DevScene<float> S, TraceCfg C, uint32_t n_wg, uint32_t wg_size, uint32_t *lists, uint32_t stride, uint32_t cap) {
  const uint32_t wg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wg >= n_wg) return;
  const int lane = threadIdx.x & 63;
  uint32_t *rec = lists + (size_t)wg * stride;
  const TilePyramid py = tile_pyramid(S, C, wg, wg_size);
  uint32_t total = 0;
  uint32_t below[RT_NUM_TYPES] = {};                          // kept slots below the first slot of each type
  for (uint32_t base = 0; base < S.n_shapes; base += 64) {
    const uint32_t i = base + lane;
    bool keep = false;
    if (i < S.n_shapes) {
      const float4 bs = S.sbound[i];
      const V3<float> c = mk<float>(bs.x, bs.y, bs.z) - py.o;
      const float R = bs.w;                                    // +inf (or NaN): never culled
      keep = !(dot(py.nl, c) < -R || dot(py.nr, c) < -R || dot(py.nb, c) < -R || dot(py.nt, c) < -R);
    }
    const unsigned long long m = __ballot(keep);
    if (m == 0ull) continue;
    const uint32_t pos = total + (uint32_t)__popcll(m & (((unsigned long long)1 << lane) - 1));
    if (keep && pos < cap) rec[16 + pos] = i;
    total += (uint32_t)__popcll(m);
#pragma unroll
    for (int t = 0; t < RT_NUM_TYPES; t++) {
      const uint32_t e = S.seg[t];
      const unsigned long long lt = e <= base ? 0ull : (e - base >= 64 ? ~0ull : (((unsigned long long)1 << (e - base)) - 1));
      below[t] += (uint32_t)__popcll(m & lt);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < RT_NUM_TYPES; t++) rec[t] = below[t];
    rec[RT_NUM_TYPES] = total;
  }
}

// LDS of k_primary_gathered for lists of up to cap primitives: records, origin terms, original indices, flags, 16 bounds
static inline __host__ __device__ size_t gathered_lds_bytes(uint32_t cap) { return (size_t)cap * (RT_GEOM_STRIDE * 4 + RT_OT_STRIDE * 4 + 8) + 64; }

// closest hit among the gathered primitives k = vseg[t] .. vseg[t + 1] of type t; equal distances go to the lowest
// original index (MODE 3: the highest), which is the hierarchy walk's rule, so either route gives the same hit
template <int MODE> RT_D int scan_gathered(const float *lg, const float *lot, const uint32_t *lorig, const uint32_t *lflags, const uint32_t *vseg, V3<float> d,
                                           float eps, float &t_out) {
  float best = __builtin_huge_valf(); int best_o = -1;
#define RT_GATHERED_LOOP(TYPE)                                                                          \
  RT_SCAN_UNROLL_PRAGMA                                                                                 \
  for (uint32_t k = vseg[TYPE]; k < vseg[TYPE + 1]; k++) {                                              \
    float f[RT_GEOM_STRIDE], ot[RT_OT_STRIDE];                                                          \
    load_record<TYPE>(lg + (size_t)k * RT_GEOM_STRIDE, f);                                              \
    load_origin_terms<TYPE>(lot + (size_t)k * RT_OT_STRIDE, ot);                                        \
    const int flags = (TYPE == RT_CYLINDER) ? (int)lflags[k] : 0;                                       \
    const float t = isect_fast<TYPE>(f, ot, d, eps, flags);                                             \
    if (t > eps && t <= best) {                                                                         \
      const int o_ = (int)lorig[k];                                                                     \
      if (t < best || (best_o >= 0 && (MODE == 3 ? o_ > best_o : o_ < best_o))) { best = t; best_o = o_; } \
    }                                                                                                   \
  }
  RT_GATHERED_LOOP(RT_PLANE) RT_GATHERED_LOOP(RT_SPHERE) RT_GATHERED_LOOP(RT_HEMISPHERE) RT_GATHERED_LOOP(RT_CYLINDER)
  RT_GATHERED_LOOP(RT_TRIANGLE) RT_GATHERED_LOOP(RT_SQUARE) RT_GATHERED_LOOP(RT_DISC) RT_GATHERED_LOOP(RT_ANNULUS)
#undef RT_GATHERED_LOOP
  t_out = best;
  return best_o;
}

template <bool WHITTED, bool FULL>
__global__ void __launch_bounds__(256) k_primary_gathered(DevScene<float> S, PathSoA<float> P, TraceCfg C, DevMap Md, DevMap Mc, float *levels,
                                                          uint32_t *queue, uint32_t *count_ptr, unsigned long long *wave_mask, uint32_t *block_count,
                                                          TraceCounters *ctr, const uint32_t *lists, uint32_t stride, uint32_t cap) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  float *lg = (float *)rt_lds, *lot = lg + (size_t)cap * RT_GEOM_STRIDE;
  uint32_t *lorig = (uint32_t *)(lot + (size_t)cap * RT_OT_STRIDE), *lflags = lorig + cap, *vseg = lflags + cap;
  const uint32_t count = C.n_paths, tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) *count_ptr = count;        // the queue length the later passes read
  const uint32_t *rec = lists + (size_t)blockIdx.x * stride;
  const uint32_t n = rec[RT_NUM_TYPES];                       // (workgroup-uniform)
  const bool gathered = n <= cap && S.sot_cam != nullptr;
  if (gathered) {
    if (tid <= RT_NUM_TYPES) vseg[tid] = rec[tid];
    // every thread issues the loads of its primitive before it stores any of them: one global round trip for the slot, one for the rest
    for (uint32_t k = tid; k < n; k += 256) {
      const uint32_t s = min(rec[16 + k], S.n_shapes - 1u);
      const uint4 *src = (const uint4 *)(S.sgeom + (size_t)s * RT_GEOM_STRIDE);
      const uint4 a = src[0], b = src[1], c = src[2];
      const float *so = S.sot_cam + (size_t)s * RT_OT_STRIDE;
      const float o0 = so[0], o1 = so[1], o2 = so[2], o3 = so[3], o4 = so[4];
      const uint32_t og = S.sorig[s], fl = S.sflags32[s];
      uint4 *dst = (uint4 *)(lg + (size_t)k * RT_GEOM_STRIDE);
      dst[0] = a; dst[1] = b; dst[2] = c;
      float *dt = lot + (size_t)k * RT_OT_STRIDE;
      dt[0] = o0; dt[1] = o1; dt[2] = o2; dt[3] = o3; dt[4] = o4;
      lorig[k] = og; lflags[k] = fl;
    }
    __syncthreads();
  }
  const ScanSrc<float> G = global_scan_src<float>(S);         // shading reads hit records (and walks the hierarchy for shadow rays) in global memory
  const uint32_t p = blockIdx.x * blockDim.x + tid;
  bool alive = false;
  ShadeTally tally;
  if (p < count) {
    if (!C.subq_cap) queue[p] = p;                            // identity queue for the ordered compaction
    PathState<float> st = raygen_path<float>(S, C, p);
    float t = 0.f; int idx;
    if (gathered) idx = scan_gathered<WHITTED ? 3 : 0>(lg, lot, lorig, lflags, vseg, st.d, (float)C.epsilon, t);
    else idx = scan_scene<float, WHITTED ? 3 : 0>(S, G, st.o, st.d, (float)C.epsilon, 0.0f, t, &tally.tests);
    alive = shade_step<float, WHITTED, FULL>(S, G, P, C, Md, Mc, levels, p, st, t, idx, 0, tally);
    if (alive) store_path<float>(P, p, st);
  }
  if (gathered && tid == 0) atomicAdd(&ctr->prim_tests[(blockIdx.x % RT_CTR_SLOTS) * 16], (unsigned long long)n * (min(blockIdx.x * blockDim.x + blockDim.x, count) - blockIdx.x * blockDim.x));
  if (C.subq_cap) block_append_subqueue(alive, p, queue, C.subq_cap, ctr->subq_count[0], blockIdx.x / (C.subq_cap >> 8));
  else block_publish_masks(alive, wave_mask, block_count);
  flush_tally(tally, ctr);
}

// One more bounce of every path in the sub-queues `qin` (lengths cin[16 s]), in the manner of the first pass: one
// lane per path, no loop, state from and to the SoA arrays, hence few registers and many waves in flight.  The
// survivors go to the same sub-queue of `qout` (lengths cout[16 s]).  Used for the first bounces after the first
// pass, while most paths are still alive; k_tail takes the rest.
template <typename T, bool LDS, bool WHITTED, bool FULL>
__global__ void __launch_bounds__(256) k_bounce(DevScene<T> S, PathSoA<T> P, TraceCfg C, DevMap Md, DevMap Mc, T *levels,
                                                const uint32_t *qin, const uint32_t *cin, uint32_t *qout, uint32_t *cout,
                                                TraceCounters *ctr, int bounce) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  uint32_t sq, first, n;
  const uint32_t len_raw = subq_load_len(cin, RT_SUBQ);     // (in flight while the scene is staged)
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds) : global_scan_src<T>(S);
  if (!subq_find_chunk(len_raw, C.subq_cap, 256u, sq, first, n)) return;      // (workgroup-uniform: the grid is an upper bound)
  bool alive = false;
  uint32_t p = 0;
  ShadeTally tally;
  if (threadIdx.x < n) {
    p = qin[first + threadIdx.x];
    if (p >= C.n_paths) tally.overflows++;                  // (never taken; guards the indexed accesses)
    else {
      PathState<T> st = load_path<T>(P, p);
      T t;
      const int idx = scan_scene<T, WHITTED ? 3 : 0>(S, G, st.o, st.d, T(C.epsilon), T(0), t, &tally.tests);
      alive = shade_step<T, WHITTED, FULL>(S, G, P, C, Md, Mc, levels, p, st, t, idx, bounce, tally);
      if (alive) store_path<T>(P, p, st);
    }
  }
  if (threadIdx.x == 0) {
    const uint32_t slot = blockIdx.x % RT_CTR_SLOTS;
    atomicAdd(&ctr->tail_rays[slot], (unsigned long long)n);
    atomicMax(&ctr->max_iter[slot], (uint32_t)bounce);
  }
  block_append_subqueue(alive, p, qout, C.subq_cap, cout, sq);
  flush_tally(tally, ctr);
}

// Wavefront shade pass: one traceRay body per queued path, for the hit k_intersect_closest left.
template <typename T, bool LDS, bool WHITTED, bool FULL>
__global__ void __launch_bounds__(256) k_shade(DevScene<T> S, PathSoA<T> P, TraceCfg C, DevMap Md, DevMap Mc, T *levels,
                                               const uint32_t *queue, const uint32_t *count_ptr, unsigned long long *wave_mask,
                                               uint32_t *block_count, TraceCounters *ctr, int bounce) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  const uint32_t count = min(*count_ptr, C.n_paths);
  if (blockIdx.x * blockDim.x >= count) return;
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds) : global_scan_src<T>(S);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool alive = false;
  ShadeTally tally;
  if (i < count) {
    const uint32_t p = queue[i];
    if (p >= C.n_paths) tally.overflows++;                  // (never taken; guards the indexed accesses)
    else {
      PathState<T> st = load_path<T>(P, p);
      alive = shade_step<T, WHITTED, FULL>(S, G, P, C, Md, Mc, levels, p, st, P.hit_t[p], P.hit_idx[p], bounce, tally);
      if (alive) store_path<T>(P, p, st);
    }
  }
  block_publish_masks(alive, wave_mask, block_count);
  flush_tally(tally, ctr);
}

// One launch that follows paths to their end.  A workgroup of RT_TAIL_THREADS lanes takes that many paths;
// after every bounce the survivors are compacted to the front of an LDS queue, and waves that are left without a
// path retire.  Only a handle moves in that queue: the path index, with the state going through the SoA arrays
// (fp64), or -- fp32 -- the slot of the workgroup's own LDS copy of the state, which a path keeps from its first
// iteration here to its end: no global round trip per bounce, and the barriers wait for no global store.
// The loop is bounded by max_bounces.
//   PRIMARY = false: the tail of a wavefront frame -- the paths, already under way, are those of n_subq sub-queues
//                    of capacity cap in `queue` with their lengths at counts[16 s] (n_subq = 1: one ordered queue).
//                    The grid is an upper bound; a workgroup finds its chunk from the lengths.
//   PRIMARY = true:  the whole frame -- the workgroup's paths are blockIdx.x * RT_TAIL_THREADS + lane and the
//                    first iteration is the fused primary pass (camera ray in registers, staged origin terms).
//                    Workgroups in their latency-bound late bounces share the CUs with workgroups in their
//                    arithmetic-bound first pass, and no queue is built at all.
// Hand-off (handoff > 0, at most 64): a workgroup that is down to `handoff` paths or fewer -- one thinly filled wave that
// would hold the workgroup's LDS for the rest of the longest path -- writes their states to the path arrays, appends
// them to the sub-queues q_out / c_out (capacity cap) and ends; the host follows with another, much smaller launch
// over those sub-queues, where the paths fill whole waves again.  Such a launch gets per_path = 1: its paths are at
// different bounces, which each lane takes from its path's depth (depth and bounce number advance together).
// Segments traced here are counted in ctr->tail_rays[], the longest path in ctr->max_iter[] (spread over
// RT_CTR_SLOTS addresses: same-address atomics from thousands of workgroups serialise).
#ifndef RT_TAIL_THREADS
#define RT_TAIL_THREADS 256
#endif
#ifndef RT_TAIL_MIN_WAVES
#define RT_TAIL_MIN_WAVES 1
#endif
enum { RT_TAIL_QUEUE = 0, RT_TAIL_PRIMARY = 1 };
template <typename T, bool LDS, bool WHITTED, bool FULL, int SRC>
__global__ void __launch_bounds__(RT_TAIL_THREADS, RT_TAIL_MIN_WAVES) k_tail(DevScene<T> S, PathSoA<T> P, TraceCfg C, DevMap Md, DevMap Mc, T *levels,
                                                          const uint32_t *queue, const uint32_t *counts, uint32_t n_subq, uint32_t cap,
                                                          TraceCounters *ctr, int bounce0, uint32_t handoff, uint32_t *q_out, uint32_t *c_out, int per_path) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  __shared__ uint32_t live[2][RT_TAIL_THREADS];
  __shared__ uint32_t wave_cnt[RT_TAIL_THREADS / 64];
  constexpr bool LST = sizeof(T) == 4;                      // path states in LDS
  constexpr uint32_t N = RT_TAIL_THREADS;
  __shared__ uint32_t lstate[LST ? (RT_LSTATE_WORDS + 1) * N : 1];      // [RT_LSTATE_WORDS][N] state words, then [N] path indices
  constexpr bool PRIMARY = SRC == RT_TAIL_PRIMARY;
  constexpr bool OT = PRIMARY && LDS && sizeof(T) == 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t first_path = blockIdx.x * RT_TAIL_THREADS;      // PRIMARY: first path; else: first entry of this workgroup's chunk in `queue`
  uint32_t n_live;
  uint32_t len_raw = 0;
  if (SRC == RT_TAIL_QUEUE) len_raw = subq_load_len(counts, n_subq);        // (in flight while the scene is staged)
  else if (first_path >= C.n_paths) return;
  const V3<T> cam_o = ld3(S.camera);
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds, OT ? &cam_o : nullptr) : global_scan_src<T>(S);
  if (SRC == RT_TAIL_QUEUE) {
    uint32_t sq;
    if (!subq_find_chunk(len_raw, cap, N, sq, first_path, n_live)) return;   // (workgroup-uniform: the grid is an upper bound)
  } else {
    n_live = min(N, C.n_paths - first_path);
  }
  if (SRC == RT_TAIL_QUEUE) {
    if (threadIdx.x < n_live) {
      const uint32_t q = queue[first_path + threadIdx.x];
      if (LST) { lstate[RT_LSTATE_WORDS * N + threadIdx.x] = q; live[0][threadIdx.x] = threadIdx.x; }
      else live[0][threadIdx.x] = q;
    }
    __syncthreads();
  }
  if (SRC != RT_TAIL_PRIMARY && (uint32_t)(wv * 64) >= n_live) return;     // waves without a path (no barrier waits for an ended wave)
  ShadeTally tally;
  unsigned long long rays = 0;                            // segments traced by this workgroup (n_live is workgroup-uniform)
  int cur = 0;
  uint32_t iters = 0;
#ifdef RT_TAIL_PROBE
  unsigned long long pr[3][6] = {}; long long pt0 = clock64();
#define RT_PROBE(k) { long long t1_ = clock64(); pr[iters > 3 ? 2 : (iters > 1 ? 1 : 0)][k] += (unsigned long long)(t1_ - pt0); pt0 = t1_; }
#else
#define RT_PROBE(k)
#endif
  int last_b = -1;                                         // per_path: bounce number of the last segment this lane traced
  bool handed = false;
  for (int b = bounce0; n_live > 0 && b < C.max_bounces; b++) {
    if (SRC == RT_TAIL_QUEUE && handoff && iters > 0 && n_live <= handoff && b + 1 < C.max_bounces) {
      // (only wave 0 is left: handoff <= 64)
      uint32_t p = 0;
      const bool mine = threadIdx.x < n_live;
      if (mine) {
        if constexpr (LST) {
          const uint32_t hd = min(live[cur][threadIdx.x], N - 1);
          p = lstate[RT_LSTATE_WORDS * N + hd];
          if (p < C.n_paths) { const PathState<T> st = load_path_lds(lstate, N, hd); store_path<T>(P, p, st); }
        } else p = live[cur][threadIdx.x];
      }
      const uint32_t sq = blockIdx.x % RT_SUBQ;
      uint32_t base = 0;
      if (threadIdx.x == 0) base = atomicAdd(&c_out[sq * 16], n_live);
      base = __shfl(base, 0, 64);
      if (mine && base + threadIdx.x < cap) q_out[(size_t)sq * cap + base + threadIdx.x] = p;
      handed = true;
      break;
    }
    rays += n_live; iters++;
    RT_PROBE(4)
    // waves 0 .. nw-1 are still running this iteration; the others have retired and their wave_cnt slots are stale
    const int nw = (int)((n_live + 63u) / 64u);
    const bool first = PRIMARY && b == bounce0;             // workgroup-uniform
    const bool fresh = b == bounce0;                        // this iteration's states are where the previous kernel left them
    bool alive = false; uint32_t p = 0, hd = 0;             // hd: the path's handle in the live queue
    if (threadIdx.x < n_live) {
      if (LST) {
        hd = first ? threadIdx.x : min(live[cur][threadIdx.x], N - 1);
        p = PRIMARY ? first_path + hd : lstate[RT_LSTATE_WORDS * N + hd];
      } else {
        hd = p = first ? first_path + threadIdx.x : live[cur][threadIdx.x];
      }
      if (p < C.n_paths) {                                  // (always true; guards the indexed accesses below)
        PathState<T> st;
        T t; int idx;
        if (first) {
          st = raygen_path<T>(S, C, p);
          idx = scan_scene<T, WHITTED ? 3 : 0, OT>(S, G, st.o, st.d, T(C.epsilon), T(0), t, &tally.tests);
        } else {
          if constexpr (LST) { if (fresh) st = load_path<T>(P, p); else st = load_path_lds(lstate, N, hd); }
          else st = load_path<T>(P, p);
#ifdef RT_TAIL_PROBE
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
          RT_PROBE(0)
          idx = scan_scene<T, WHITTED ? 3 : 0>(S, G, st.o, st.d, T(C.epsilon), T(0), t, &tally.tests);
        }
        RT_PROBE(1)
        const int bl = per_path ? (int)(st.depth & 0xFFFF) : b;
        last_b = bl;
        alive = shade_step<T, WHITTED, FULL>(S, G, P, C, Md, Mc, levels, p, st, t, idx, bl, tally);
        if (alive) {
          if constexpr (LST) store_path_lds(lstate, N, hd, st); else store_path<T>(P, p, st);
        }
        RT_PROBE(2)
      } else {
        tally.overflows++;
      }
    }
    // ordered compaction of the survivors into the other LDS queue
    unsigned long long mask = __ballot(alive);
    if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < nw; w++) { uint32_t c = wave_cnt[w]; if (w < wv) base += c; total += c; }
    const uint32_t slot = base + (uint32_t)__popcll(mask & (((unsigned long long)1 << lane) - 1));
    if (alive && slot < RT_TAIL_THREADS) live[cur ^ 1][slot] = hd;
    __syncthreads();
    cur ^= 1; n_live = min(total, (uint32_t)RT_TAIL_THREADS);
    RT_PROBE(3)
    if ((uint32_t)(wv * 64) >= n_live) break;              // this wave has no path left (retired waves do not hold up barriers)
  }
  (void)handed;
  if (per_path) {                                           // iterations the first launch would have run for this wave's longest path
    for (int off = 32; off > 0; off >>= 1) last_b = max(last_b, __shfl_xor(last_b, off, 64));
    if (lane == 0 && last_b >= 0) atomicMax(&ctr->max_iter[(blockIdx.x + wv) % RT_CTR_SLOTS], (uint32_t)(last_b - bounce0 + 1) + C.iter_base);
  }
  if (threadIdx.x == 0) {                                   // wave 0 is the last to retire: it has seen every iteration
    const uint32_t slot = blockIdx.x % RT_CTR_SLOTS;
    if (rays) atomicAdd(&ctr->tail_rays[slot], rays);
    if (!per_path) atomicMax(&ctr->max_iter[slot], iters ? iters + C.iter_base : 0u);
#ifdef RT_TAIL_PROBE
    pr[0][5] = iters > 0; pr[1][5] = iters > 1 ? (iters > 3 ? 2 : iters - 1) : 0; pr[2][5] = iters > 3 ? iters - 3 : 0;
    for (int a = 0; a < 3; a++) for (int k = 0; k < 6; k++) if (pr[a][k]) atomicAdd(&ctr->probe[a][k], pr[a][k]);
#endif
  }
  flush_tally(tally, ctr);
}

// The tail with persistent lanes (option tail_persistent = paths per lane): a workgroup takes a chunk of per_wg queued
// paths; every lane follows one path with the state in registers and, when that path ends, takes the chunk's next path
// nobody has started (a counter in LDS) -- no compaction, no barrier and no LDS traffic for path states inside the loop,
// and lanes stay busy until the chunk runs dry.  Paths are at different bounces: each lane takes its bounce number from
// its path's depth (depth and bounce number advance together).
// Hand-off (handoff > 0): once the chunk is dry, a wave that is down to `handoff` paths or fewer writes their states to
// the path arrays, appends them to sub-queue (its number % RT_SUBQ) of q_out / c_out (capacity cap_out each) and ends; the
// host follows with a second, much smaller launch of this kernel over those sub-queues (handoff = 0), where the paths fill
// whole waves again -- instead of a few lanes per wave following the longest paths while the others idle.
// (Off by default.  Measured on the 1080p bench frame with four frames in flight: 0.0872 / 0.0860 / 0.0863 / 0.0859 ms per
// frame with handoff = 8 / 16 / 24 / 32 against 0.0821 ms without -- the other frames' kernels already fill the lanes that
// a frame's last paths leave idle, and the second launch has to stage the scene again and wait for the first to end.)
template <typename T, bool LDS, bool WHITTED, bool FULL>
__global__ void __launch_bounds__(256, RT_TP_MIN_WAVES) k_tail_persistent(DevScene<T> S, PathSoA<T> P, TraceCfg C, DevMap Md, DevMap Mc, T *levels, const uint32_t *queue,
                                                         const uint32_t *counts, uint32_t n_subq, uint32_t cap, TraceCounters *ctr, int bounce0, uint32_t per_wg,
                                                         uint32_t handoff, uint32_t *q_out, uint32_t *c_out, uint32_t cap_out) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  __shared__ uint32_t next_path;
  const uint32_t len_raw = subq_load_len(counts, n_subq);   // (in flight while the scene is staged)
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds) : global_scan_src<T>(S);
  uint32_t sq, first, n;
  if (!subq_find_chunk(len_raw, cap, per_wg, sq, first, n)) return;          // (workgroup-uniform: the grid is an upper bound)
  if (threadIdx.x == 0) next_path = blockDim.x;
  __syncthreads();
  ShadeTally tally;
  unsigned long long rays = 0;
  int last_b = -1;
  uint32_t j = threadIdx.x;
  bool have = j < n;
  uint32_t p = 0;
  PathState<T> st;
  // (taking a path costs the wave two dependent global round trips; loading a lane's next path ahead of time was
  // measured slower: 17 more registers and the loads of paths that the chunk's last lanes never reach)
  auto take = [&]() {                                       // lane holds queue entry j: its path, or the chunk's next valid one
    for (;;) {
      p = queue[first + j];
      if (p < C.n_paths) { st = load_path<T>(P, p); return; }
      if (p != 0xFFFFFFFFu) tally.overflows++;              // (never taken; 0xFFFFFFFF: a slot that a hand-off reserved and gave up)
      j = atomicAdd(&next_path, 1u);
      if (j >= n) { have = false; p = 0; return; }
    }
  };
  if (have) take();
  while (have) {
    T t;
    const int b = (int)(st.depth & 0xFFFF);
    const int idx = scan_scene<T, WHITTED ? 3 : 0>(S, G, st.o, st.d, T(C.epsilon), T(0), t, &tally.tests);
    rays++; last_b = max(last_b, b);
    const bool alive = shade_step<T, WHITTED, FULL>(S, G, P, C, Md, Mc, levels, p, st, t, idx, b, tally);
    if (!alive) {
      j = atomicAdd(&next_path, 1u);
      have = j < n;
      if (have) take();
    }
    if (handoff) {
      // the chunk has run dry and few lanes of this wave still hold a path: those paths go to the follow-up launch
      const unsigned long long live = __ballot(have);
      const uint32_t n_live = (uint32_t)__popcll(live);
      if (n_live != 0u && n_live <= handoff && *(volatile uint32_t *)&next_path >= n) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)live) - 1;
        const uint32_t sq_out = (blockIdx.x * 4u + (threadIdx.x >> 6)) % RT_SUBQ;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&c_out[sq_out * 16], n_live);
        base = __shfl(base, leader, 64);
        const bool fits = base + n_live <= cap_out;           // (else: the wave keeps its paths and marks the slots it reserved)
        if (have) {
          const uint32_t pos = base + (uint32_t)__popcll(live & (((unsigned long long)1 << lane) - 1));
          if (fits) { store_path<T>(P, p, st); q_out[(size_t)sq_out * cap_out + pos] = p; have = false; }
          else if (pos < cap_out) q_out[(size_t)sq_out * cap_out + pos] = 0xFFFFFFFFu;
        }
        if (!fits) handoff = 0u;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) { rays += __shfl_xor(rays, off, 64); last_b = max(last_b, __shfl_xor(last_b, off, 64)); }
  if ((threadIdx.x & 63) == 0) {
    const uint32_t slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) % RT_CTR_SLOTS;
    if (rays) atomicAdd(&ctr->tail_rays[slot], rays);
    if (last_b >= 0) atomicMax(&ctr->max_iter[slot], (uint32_t)(last_b - bounce0 + 1) + C.iter_base);
  }
  flush_tally(tally, ctr);
}

// (diffuse sum / diffuse area + caustic sum / caustic area) x albedo (rayphoton.js:1808-1815) into the path's
// fixed-point accumulators; partial sums of one request may arrive from several waves, in any order
// A path's accumulator holds |x| < 2^31 in steps of 2^-32 (radiance is in 0..255 units, so the range is 8e6 times the
// brightest displayable value).  A partial sum beyond +-2^30 is clamped to that and counted in the frame's gather_saturations,
// a NaN is left out and counted (C.gsat) -- the gather-in-place path would return inf / NaN there, which the resolve clamps or
// zeroes as well; the count tells a caller that the two routes may differ.
// a partial sum in the accumulators' fixed-point units (clamped and counted as described above)
RT_D void gather_quantise(const TraceCfg &C, float4 albedo, V3<float> sd, V3<float> sc, long long k[3]) {
  const float PI_ = 3.141592653589793f;
  const float dArea = PI_ * (float)(C.diffuse_radius * C.diffuse_radius), cArea = PI_ * (float)(C.caustic_radius * C.caustic_radius);
  const float lc[3] = { albedo.x, albedo.y, albedo.z }, d3[3] = { sd.x, sd.y, sd.z }, c3[3] = { sc.x, sc.y, sc.z };
  for (int c = 0; c < 3; c++) {
    const double f = ((double)(d3[c] * lc[c] / dArea) + (double)(c3[c] * lc[c] / cArea)) * RT_GACC_SCALE;
    const double lim = 1073741824.0 * RT_GACC_SCALE;            // 2^30 units: two such partial sums still fit
    const bool ok = f == f && f > -lim && f < lim;
    k[c] = ok ? __double2ll_rn(f) : (f == f ? __double2ll_rn(f < 0 ? -lim : lim) : 0ll);
    if (!ok && C.gsat) atomicAdd(C.gsat, 1ull);
  }
}
RT_D void gather_add(const TraceCfg &C, uint32_t p, const long long k[3]) {
  for (int c = 0; c < 3; c++) if (k[c]) atomicAdd((unsigned long long *)&C.gacc[(size_t)p * 3 + c], (unsigned long long)k[c]);
}
RT_D void gather_accumulate(const TraceCfg &C, uint32_t p, float4 albedo, V3<float> sd, V3<float> sc) {
  long long k[3];
  gather_quantise(C, albedo, sd, sc, k);
  gather_add(C, p, k);
}

// Serves the gather requests of a frame (TraceCfg::gq): one lane per request, a wave's 64 requests together where
// they lie close enough (gather_map_wave), lane by lane otherwise; (diffuse sum / diffuse area + caustic sum / caustic
// area) x albedo goes to the path's fixed-point accumulators.  The grid is an upper bound (gq_cap requests).
static __global__ void __launch_bounds__(256) k_gather_deferred(TraceCfg C, DevMap Md, DevMap Mc, GatherItems items, int split_d, int split_c) {
  __shared__ WaveGatherLds wg_lds[4];
  const uint32_t total = min(C.gq_first + *C.gq_count, C.gq_cap);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ((i & ~63u) >= total) return;                           // (wave-uniform)
#ifdef RT_GATHER_PROBE
  const long long probe_t0 = clock64();
#endif
  bool need = i < total;
  uint32_t r = need ? (C.gq_order ? C.gq_order[i] : i) : 0u;
  if (r >= C.gq_cap) { r = 0; need = false; }                // (never true)
  if (need && !C.gq_valid[r]) need = false;                  // a first-pass slot whose path left no request (sorted: these come last)
  if (!__any(need)) return;
  const float4 *q = C.gq + (size_t)r * 3;
  const float4 q0 = q[0], q1 = q[1], q2 = q[2];
  const V3<float> hit = mk<float>(q0.x, q0.y, q0.z), nrm = mk<float>(q1.x, q1.y, q1.z);
  const uint32_t p = __float_as_uint(q0.w), todo = need ? (__float_as_uint(q1.w) & 3u) : 0u;
  V3<float> sd = mk<float>(0.f, 0.f, 0.f), sc = mk<float>(0.f, 0.f, 0.f);
  WaveGatherLds &L = wg_lds[threadIdx.x >> 6];
  const bool nd = (todo & 1u) && Md.n, nc = (todo & 2u) && Mc.n;
  // With a sorted queue the wave's requests are neighbours: one walk for all of them, whatever their spread, and the
  // subtrees below the split depth as items for k_gather_items.  Unsorted: together only where that pays.
  const bool sorted = C.gq_order != nullptr;
  const float ratio = sorted ? 0.f : C.wave_gather_ratio;
  const bool coop = sorted || C.wave_gather_ratio > 0.f;
  GatherItems it = items;
  it.stage = L.stage; it.stage_n = 0; it.full = false;
  // (Tried: halves, quarters or eighths of the wave walk on their own when their boxes are much smaller than the whole
  // wave's -- 11 % fewer candidates, the same frame time; the first 511 nodes of both trees in LDS -- the same time.)
  it.tag = i >> 6;
  if (__any(nd) && !(coop && gather_map_wave(Md, nd, hit, nrm, C.diffuse_radius, ratio, L, sd, 1u, split_d, (sorted && split_d > 0) ? &it : nullptr)) && nd)
    sd = gather_map<float>(Md, hit, nrm, C.diffuse_radius);
  it.tag = (i >> 6) | 0x80000000u;
  if (__any(nc) && !(coop && gather_map_wave(Mc, nc, hit, nrm, C.caustic_radius, ratio, L, sc, 1u, split_c, (sorted && split_c > 0) ? &it : nullptr)) && nc)
    sc = gather_map<float>(Mc, hit, nrm, C.caustic_radius);
  // the wave's items go to the frame's list together; any the list has no room for are served here
  gather_items_flush(it);
  for (uint32_t k = 0; k < it.stage_n; k++) {               // (wave-uniform; normally no iteration)
    const uint2 item = it.stage[k];
    const bool caustic = item.x >> 31;
    V3<float> part;
    (void)gather_map_wave(caustic ? Mc : Md, caustic ? nc : nd, hit, nrm, caustic ? C.caustic_radius : C.diffuse_radius, 0.f, L, part, item.y);
    if (caustic ? nc : nd) { if (caustic) sc = sc + part; else sd = sd + part; }
  }
#ifdef RT_GATHER_PROBE
  if ((threadIdx.x & 63) == 0) { const unsigned long long dt_ = (unsigned long long)(clock64() - probe_t0); atomicMax(&g_gather_probe[7], dt_); atomicAdd(&g_gather_hist[63 - __clzll((long long)(dt_ | 1ull))], 1ull); }
#endif
  if (!need || p >= C.n_paths) return;
  gather_accumulate(C, p, q2, sd, sc);
}

// The items k_gather_deferred has split off: persistent waves, each taking the next item while there are any (the
// cursor and the item count live in device memory; the count is final when this kernel starts).
// (Tried: a wave takes 4, 8 or 16 consecutive items at a time and keeps the requests and their sums in registers while
// the items name the same requests -- 6.95, 8.5, 11.4 against 5.99 ms per frame: consecutive items cost alike, so runs of
// them undo the balance that taking items one by one gives.)
static __global__ void __launch_bounds__(256) k_gather_items(TraceCfg C, DevMap Md, DevMap Mc, GatherItems items, uint32_t *cursor) {
  __shared__ WaveGatherLds wg_lds[4];
  const int lane = threadIdx.x & 63;
  WaveGatherLds &L = wg_lds[threadIdx.x >> 6];
  const uint32_t total = min(C.gq_first + *C.gq_count, C.gq_cap);
  // every wave starts at a shard of its own and goes round all of them (cursor[s * RT_ITEM_WORDS]: shard s's next item)
  const uint32_t first_shard = (blockIdx.x * 4u + (threadIdx.x >> 6)) % RT_ITEM_SHARDS;
  for (uint32_t step = 0; step < RT_ITEM_SHARDS; step++) {
    const uint32_t shard = (first_shard + step) % RT_ITEM_SHARDS;
    const uint32_t n_items = min(items.count[shard * RT_ITEM_WORDS], items.cap);
    const uint2 *list = items.items + (size_t)shard * items.cap;
    for (;;) {
      uint32_t k = 0;
      if (lane == 0) k = atomicAdd(cursor + shard * RT_ITEM_WORDS, 1u);
      k = __shfl(k, 0, 64);
      if (k >= n_items) break;
      const uint2 item = list[k];
      const bool caustic = item.x >> 31;
      const uint32_t i = (item.x & 0x7FFFFFFFu) * 64u + (uint32_t)lane;
      bool need = i < total;
      uint32_t r = need ? (C.gq_order ? C.gq_order[i] : i) : 0u;
      if (r >= C.gq_cap) { r = 0; need = false; }            // (never true)
      if (need && !C.gq_valid[r]) need = false;
      const float4 *q = C.gq + (size_t)r * 3;
      const float4 q0 = q[0], q1 = q[1], q2 = q[2];
      const V3<float> hit = mk<float>(q0.x, q0.y, q0.z), nrm = mk<float>(q1.x, q1.y, q1.z);
      const uint32_t p = __float_as_uint(q0.w), todo = need ? (__float_as_uint(q1.w) & 3u) : 0u;
      const DevMap &M = caustic ? Mc : Md;
      const bool want = (todo & (caustic ? 2u : 1u)) && M.n && item.y >= 1u && item.y <= M.n;
      V3<float> sum = mk<float>(0.f, 0.f, 0.f), zero = mk<float>(0.f, 0.f, 0.f);
      (void)gather_map_wave(M, want, hit, nrm, caustic ? C.caustic_radius : C.diffuse_radius, 0.f, L, sum, item.y >= 1u && item.y <= M.n ? item.y : 1u);
      if (want && p < C.n_paths) gather_accumulate(C, p, q2, caustic ? zero : sum, caustic ? sum : zero);
    }
  }
}

// Sort keys of the requests: 30-bit Hilbert index of the point in cells of `cell` (10 bits an axis around the origin,
// clamped; the sort only has to bring neighbours together), then the path and the bounce, which make the order -- and
// with it the groups of 64 and the order of every sum -- the same whatever order the requests were queued in; all ones
// for the unused slots of the queue.  Hilbert rather than Morton order: consecutive cells of the curve are always
// neighbours in space, so the 64 requests of a wave do not straddle one of the Z curve's jumps and the box around their
// balls (what the wave's walk collects candidates from) stays close to one ball's.
// cells of the keys per (largest) gather radius
#define RT_GATHER_CELLS_PER_RADIUS 8.0
RT_D uint32_t spread10(uint32_t v) {
  v &= 1023u; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
  return v;
}
// (J. Skilling, "Programming the Hilbert curve", 2004: axes to transposed index, then the bits interleaved)
RT_D uint32_t hilbert30(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t X[3] = { x & 1023u, y & 1023u, z & 1023u };
  for (uint32_t Q = 512u; Q > 1u; Q >>= 1) {
    const uint32_t P = Q - 1u;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (X[i] & Q) X[0] ^= P;
      else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
    }
  }
  X[1] ^= X[0]; X[2] ^= X[1];
  uint32_t t = 0;
  for (uint32_t Q = 512u; Q > 1u; Q >>= 1) if (X[2] & Q) t ^= Q - 1u;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  return (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
}
static __global__ void __launch_bounds__(256) k_gather_keys(TraceCfg C, float inv_cell, unsigned long long *keys, uint32_t *vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C.gq_cap) return;
  const uint32_t total = min(C.gq_first + *C.gq_count, C.gq_cap);
  unsigned long long key = ~0ull;
  if (i < total && C.gq_valid[i]) {
    const float4 q0 = C.gq[(size_t)i * 3], q1 = C.gq[(size_t)i * 3 + 1];
    auto cellof = [&](float v) { const float c = v * inv_cell + 512.0f; return (uint32_t)(c < 0.f ? 0.f : (c > 1023.f ? 1023.f : c)); };   // (NaN -> 0)
    const uint32_t h = hilbert30(cellof(q0.x), cellof(q0.y), cellof(q0.z));
    key = ((unsigned long long)h << 34) | ((unsigned long long)(__float_as_uint(q0.w) & 0x0FFFFFFFu) << 6) | ((__float_as_uint(q1.w) >> 2) & 63u);
  }
  keys[i] = key; vals[i] = i;
}

// Pixel resolve (rayphoton.js:1610-1623): clamp each sub-sample (`colour()`), sum, floor of the
// mean.  pre = the last sub-sample's unclamped colour; rgba8 = post as bytes.
template <typename T>
__global__ void __launch_bounds__(256) k_resolve(PathSoA<T> P, TraceCfg C, int n_pixels, int sub_sample, double *pre, double *post, uint8_t *rgba8) {
  const int pp = blockIdx.x * blockDim.x + threadIdx.x;    // pixel in path order; `pix` below: in output order
  if (pp >= n_pixels) return;
  const int pix = (int)pixel_out_index(C, (uint32_t)pp);
  if (pix >= n_pixels) return;                             // (never true)
  const int SS = sub_sample * sub_sample;
  double tot[4] = { 0, 0, 0, 0 }, last[4] = { 0, 0, 0, 0 };
  for (int s = 0; s < SS; s++) {
    const T *o = P.out + ((size_t)pp * SS + s) * 4;
    for (int c = 0; c < 4; c++) {
      last[c] = (double)o[c];
      if (C.gacc && c < 3) last[c] = (double)(T)(last[c] + (double)C.gacc[((size_t)pp * SS + s) * 3 + c] / RT_GACC_SCALE);    // the deferred photon gathers
      tot[c] += clamp01(last[c], 0, c == 3 ? 1 : 255);
    }
  }
  for (int c = 0; c < 4; c++) {
    double v = floor(tot[c] / SS);
    if (pre) pre[(size_t)pix * 4 + c] = last[c];
    if (post) post[(size_t)pix * 4 + c] = v;
    if (rgba8) {
      double b = (c == 3) ? v * 255.0 : v;
      rgba8[(size_t)pix * 4 + c] = (b == b) ? (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b)) : 0;
    }
  }
}

// Pixel resolve of the older tracer (ray.js:836-841 and the blend of :979-986): substitute the terminal
// colour back up through the path's level records, then average the sub-samples (colour() does not clamp
// there).  A sub-sample whose trace the reference leaves undefined (ray.js:905) makes the pixel NaN.
template <typename T>
__global__ void __launch_bounds__(256) k_resolve_a(PathSoA<T> P, TraceCfg C, const T *levels, uint32_t n_paths, int max_depth, int n_pixels,
                                                   int sub_sample, double *pre, double *post, uint8_t *rgba8) {
  const int pp = blockIdx.x * blockDim.x + threadIdx.x;    // pixel in path order; `pix` below: in output order
  if (pp >= n_pixels) return;
  const int pix = (int)pixel_out_index(C, (uint32_t)pp);
  if (pix >= n_pixels) return;                             // (never true)
  const int SS = sub_sample * sub_sample;
  double tot[4] = { 0, 0, 0, 0 }, last[4] = { 0, 0, 0, 0 };
  bool bad = false;
  for (int s = 0; s < SS; s++) {
    const size_t p = (size_t)pp * SS + s;
    if (p >= n_paths) break;                                // (never true)
    const T *o = P.out + p * 4;
    T c[3] = { o[0], o[1], o[2] }; T a = o[3];
    int nlev = P.depth[p];
    nlev = nlev < 0 ? 0 : (nlev > max_depth + 1 ? max_depth + 1 : nlev);
    for (int j = nlev - 1; j >= 0; j--) {
      const T *rec = levels + ((size_t)j * n_paths + p) * 4;
      for (int k = 0; k < 3; k++) {
        T v = rt_floor((rec[k] + rec[3] * c[k]) + T(0.5));                       // Math.round
        c[k] = v < T(0) ? T(0) : (v > T(255) ? T(255) : v);
      }
    }
    if (nlev > 0) a = T(1);
    if (!(c[0] == c[0]) || !(c[1] == c[1]) || !(c[2] == c[2])) bad = true;
    last[0] = (double)c[0]; last[1] = (double)c[1]; last[2] = (double)c[2]; last[3] = (double)a;
    for (int k = 0; k < 4; k++) tot[k] += last[k];
  }
  const double qnan = __builtin_nan("");
  for (int c = 0; c < 4; c++) {
    double v = bad ? qnan : floor(tot[c] / SS);
    if (pre) pre[(size_t)pix * 4 + c] = bad ? qnan : last[c];
    if (post) post[(size_t)pix * 4 + c] = v;
    if (rgba8) {
      double b = (c == 3) ? v * 255.0 : v;
      rgba8[(size_t)pix * 4 + c] = (b == b) ? (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b)) : 0;
    }
  }
}
