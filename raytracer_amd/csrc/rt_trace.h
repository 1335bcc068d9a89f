// Wavefront path tracer: SoA path state in HBM, one kernel per stage, survivors compacted
// into the next queue with a wave ballot + prefix count and one atomic per wave.
//
//   raygen -> [ intersect_closest -> shade ]* -> resolve
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
  T *out;                             // n x 4: finished paths' colour
};

struct TraceCfg {
  double epsilon, max_trace_dist, diffuse_radius, caustic_radius, standard_lamp_area, shadow_bias;
  int32_t max_depth, sub_sample, use_direct_light, use_photon_maps, max_bounces;
  uint64_t seed;
  int32_t x0, y0, w, h;
  uint32_t n_paths;                   // w * h * sub_sample^2
};

#define RT_MAX_WAVE_BOUNCES 62
// device-side bookkeeping of one rt_trace_tile call; counts[b] = queue length of wavefront pass b
struct TraceCounters { unsigned long long shadow_rays, gathers, stack_overflows, tail_rays; uint32_t counts[RT_MAX_WAVE_BOUNCES + 2]; };

// traceOnCanvas ray set-up (rayphoton.js:1589-1609, superSampleScale 0) for path p: writes the initial
// path state and returns the camera ray.
template <typename T>
RT_D void raygen_path(const DevScene<T> &S, const PathSoA<T> &P, const TraceCfg &C, uint32_t p, V3<T> &o, V3<T> &d) {
  const int SS = C.sub_sample * C.sub_sample;
  uint32_t pix = p / SS, sub = p % SS;
  int X = C.x0 + (int)(pix % C.w), Y = C.y0 + (int)(pix / C.w);
  int sx = sub % C.sub_sample, sy = sub / C.sub_sample;
  RngStream rng; rng.init(C.seed, RT_DOMAIN_RENDER, (uint32_t)(Y * (int)S.width + X), sub);
  T u = ((T(X) + (T(sx) + rng.next<T>()) / T(C.sub_sample)) * T(2) / T(S.width)) - T(1);
  T v = -(((T(Y) + (T(sy) + rng.next<T>()) / T(C.sub_sample)) * T(2) / T(S.height)) - T(1));
  u *= S.camera[12]; v *= S.camera[13];
  V3<T> co = ld3(S.camera), cu = ld3(S.camera + 3), cv = ld3(S.camera + 6), cw = ld3(S.camera + 9);
  V3<T> xyz = ((co + u * cu) + v * cv) + T(-1) * cw;                    // Camera.toXYZ, rayphoton.js:712
  o = co; d = normalize<T>(xyz - co, T(C.epsilon));
  P.ox[p] = o.x; P.oy[p] = o.y; P.oz[p] = o.z; P.dx[p] = d.x; P.dy[p] = d.y; P.dz[p] = d.z;
  P.max_dist[p] = T(C.max_trace_dist); P.importance[p] = T(1);
  P.acc_r[p] = T(0); P.acc_g[p] = T(0); P.acc_b[p] = T(0);
  P.draw[p] = rng.draw; P.depth[p] = 0;
  MatStack st; st.init(); P.stack[p] = st.bits;
}

// Closest hit of every queued path against the whole scene (rayphoton.js:1642-1650).
// This is the VALU-bound kernel: the primitive index is wave-uniform, so primitive operands
// are scalar loads and every lane spends its cycles on the distance arithmetic.
// PRIMARY: the queue is the identity and the ray is generated in place (no ray loads); otherwise
// the segment left by the previous shade pass is read from the path state.
// WHITTED: ray.js's scan order (equal distances go to the highest index) instead of rayphoton.js's.
template <typename T, bool PRIMARY, bool LDS, bool WHITTED>
__global__ void __launch_bounds__(256) k_intersect_closest(DevScene<T> S, PathSoA<T> P, TraceCfg C, uint32_t *queue, uint32_t *count_ptr) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  // device-side queue length (the grid is only an upper bound); the primary pass knows it and publishes it
  const uint32_t count = PRIMARY ? C.n_paths : min(*count_ptr, C.n_paths);
  if (PRIMARY && blockIdx.x == 0 && threadIdx.x == 0) *count_ptr = count;
  if (blockIdx.x * blockDim.x >= count) return;
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds) : global_scan_src<T>(S);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint32_t p; V3<T> o, d;
  if (PRIMARY) { p = i; queue[i] = i; raygen_path<T>(S, P, C, p, o, d); }     // also seeds the identity queue the shade pass reads
  else {
    p = queue[i];
    if (p >= C.n_paths) return;                             // (never true; guards the indexed accesses below)
    o = mk<T>(P.ox[p], P.oy[p], P.oz[p]); d = mk<T>(P.dx[p], P.dy[p], P.dz[p]);
  }
  T t;
  int idx = scan_scene<T, WHITTED ? 3 : 0>(S, G, o, d, T(C.epsilon), T(0), t);
  P.hit_t[p] = t; P.hit_idx[p] = idx;
}

// directLight (rayphoton.js:1841-1880): per light one createPhoton() of a RANDOM light, scaled by
// THIS light's area; an unclamped cosine; floor at the end.
template <typename T>
RT_D V3<T> direct_light(const DevScene<T> &S, const ScanSrc<T> &G, const TraceCfg &C, V3<T> point, V3<T> normal, V3<T> local, RngStream &rng,
                        unsigned long long &shadow_rays) {
  const T eps = T(C.epsilon);
  V3<T> col = mk<T>(0, 0, 0);
  for (uint32_t l = 0; l < S.n_lights; l++) {
    PhotonState<T> ph; create_photon<T>(S, eps, rng, ph);
    T power = ph.power * S.light_data[l][1];
    V3<T> dir = normalize<T>(ph.o - point, eps);
    T source_dist = rt_sqrt(sqlen(point - ph.o));
    T thit;
    shadow_rays++;
    if (!scan_scene<T, 2>(S, G, point, dir, eps, source_dist, thit)) {
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

template <typename T> RT_D void finish_path(const PathSoA<T> &P, uint32_t p, V3<T> acc, V3<T> term, T term_a, bool alpha_fixed) {
  T *o = P.out + (size_t)p * 4;
  o[0] = acc.x + term.x; o[1] = acc.y + term.y; o[2] = acc.z + term.z;
  o[3] = alpha_fixed ? T(1) : term_a;
}

// One traceRay body per queued path (rayphoton.js:1651-1838).  A path either finishes (its
// colour goes to P.out) or leaves a new segment in the path state and enters `next_queue`.
struct ShadeTally { unsigned long long shadow_rays = 0, gathers = 0, overflows = 0; };

template <typename T>
RT_D bool shade_body(const DevScene<T> &S, const ScanSrc<T> &G, const PathSoA<T> &P, const TraceCfg &C, const DevMap &Md, const DevMap &Mc,
                     uint32_t p, T t, int idx, int bounce, ShadeTally &tally) {
  bool alive = false;
  unsigned long long &shadow_rays = tally.shadow_rays, &gathers = tally.gathers, &overflows = tally.overflows;
  {
    const T eps = T(C.epsilon);
    V3<T> acc = mk<T>(P.acc_r[p], P.acc_g[p], P.acc_b[p]);
    const int dword = P.depth[p];
    const int depth = dword & 0xFFFF; const bool alpha_fixed = dword & 0x10000;
    const V3<T> ZERO = mk<T>(0, 0, 0);
    if (idx < 0) {
      finish_path<T>(P, p, acc, mk<T>(16, 16, 16), T(1), alpha_fixed);                 // COL_VERY_DARK_GREY :1838
    } else if (S.meta[4 * idx + 2] & RT_FLAG_LIGHT) {                                  // :1654-1656
      int l = 0;
      for (uint32_t k = 0; k < S.n_lights; k++) if ((int)S.light_shape[k] == idx) l = (int)k;
      finish_path<T>(P, p, acc, mk<T>(S.light_data[l][3], S.light_data[l][4], S.light_data[l][5]), S.light_data[l][6], alpha_fixed);
    } else {
      V3<T> o = mk<T>(P.ox[p], P.oy[p], P.oz[p]), d = mk<T>(P.dx[p], P.dy[p], P.dz[p]);
      const T max_dist = P.max_dist[p];
      V3<T> hit = o + t * d;
      V3<T> local = shape_colour<T>(S, idx, hit);
      V3<T> n = shape_normal<T>(S, idx, hit);
      if (t < max_dist || depth < C.max_depth) {                                       // :1677 (note: OR)
        const T cos1 = -dot(d, n);
        const int mat = S.meta[4 * idx + 1] & 7;
        const bool has_mat = S.meta[4 * idx + 1] < 8;
        RngStream rng; rng.init(C.seed, RT_DOMAIN_RENDER, 0, 0, P.draw[p]);
        {   // the stream key is (pixel, sub-sample); recover it from the path index
          const int SS = C.sub_sample * C.sub_sample;
          uint32_t pix = p / SS;
          int X = C.x0 + (int)(pix % C.w), Y = C.y0 + (int)(pix / C.w);
          rng.a = (uint32_t)(Y * (int)S.width + X); rng.b = p % SS;
        }
        if (has_mat && S.mat[0][mat] == T(1)) {                                        // transparent :1679
          MatStack st; st.bits = P.stack[p];
          V3<T> ndir; bool ovf = false;
          if (!fresnel_bounce<T>(S, mat, d, n, cos1, st, rng, ndir, ovf)) {
            finish_path<T>(P, p, acc, ZERO, T(0), alpha_fixed);                        // COL_BLACK :1699
          } else {
            if (ovf) overflows++;
            ndir = normalize<T>(ndir, eps);
            P.ox[p] = hit.x; P.oy[p] = hit.y; P.oz[p] = hit.z; P.dx[p] = ndir.x; P.dy[p] = ndir.y; P.dz[p] = ndir.z;
            P.max_dist[p] = max_dist - t; P.depth[p] = dword + 1; P.stack[p] = st.bits; P.draw[p] = rng.draw;
            alive = true;                                                              // importance unchanged, >= 0.01
          }
        } else if (cos1 < T(0)) {
          finish_path<T>(P, p, acc, ZERO, T(1), alpha_fixed);                          // inside an opaque object :1736-1747
        } else {
          const T refl = has_mat ? S.mat[2][mat] : Limits<T>::nan();
          V3<T> direct = ZERO;
          if (refl < T(1) && C.use_direct_light) {
            const T bias = T(C.shadow_bias);
            direct = direct_light<T>(S, G, C, hit + bias * n, n, local, rng, shadow_rays);          // :1750-1752
          }
          bool spec = false;
          if (rng.next_k() < (has_mat ? S.refl_k[mat] : 0u)) spec = rng.next_k() < S.spec_k[mat];   // :1753, short-circuit
          V3<T> diff = ZERO, caus = ZERO;
          if (C.use_photon_maps) {
            if (Md.n) { diff = gather_map<T>(Md, hit, n, C.diffuse_radius); gathers++; }
            if (Mc.n) { caus = gather_map<T>(Mc, hit, n, C.caustic_radius); gathers++; }
          }
          const T PI_ = T(3.141592653589793);
          const T dArea = PI_ * T(C.diffuse_radius * C.diffuse_radius), cArea = PI_ * T(C.caustic_radius * C.caustic_radius);
          V3<T> lc = mk<T>(local.x / T(255), local.y / T(255), local.z / T(255));
          diff = mk<T>(diff.x * lc.x / dArea, diff.y * lc.y / dArea, diff.z * lc.z / dArea);      // :1808-1815
          caus = mk<T>(caus.x * lc.x / cArea, caus.y * lc.y / cArea, caus.z * lc.z / cArea);
          V3<T> term = ((T(1) - refl) * direct + diff) + caus;                                    // :1817-1822 minus the child
          acc = acc + term;
          const T imp = P.importance[p] * refl;
          if (spec && !(imp < T(0.01))) {                                                         // child would return at :1627
            V3<T> ndir = normalize<T>(d + (T(2) * cos1) * n, eps);
            P.ox[p] = hit.x; P.oy[p] = hit.y; P.oz[p] = hit.z; P.dx[p] = ndir.x; P.dy[p] = ndir.y; P.dz[p] = ndir.z;
            P.max_dist[p] = max_dist - t; P.importance[p] = imp; P.depth[p] = (dword + 1) | 0x10000;
            P.acc_r[p] = acc.x; P.acc_g[p] = acc.y; P.acc_b[p] = acc.z; P.draw[p] = rng.draw;
            alive = true;
          } else {
            finish_path<T>(P, p, acc, ZERO, T(1), true);
          }
        }
      } else {
        finish_path<T>(P, p, acc, ZERO, T(1), alpha_fixed);                            // :1828-1833
      }
    }
    if (alive && bounce + 1 >= C.max_bounces) {                                        // safety cap; the reference has none
      finish_path<T>(P, p, mk<T>(P.acc_r[p], P.acc_g[p], P.acc_b[p]), ZERO, T(1), (P.depth[p] & 0x10000) != 0);
      alive = false;
    }
  }
  return alive;
}


// ------------------------------------------------------------------ the older tracer (ray.js:801-993)
// stream of path p continued at draw `draw`: the key is (pixel, sub-sample)
RT_D RngStream path_rng(uint32_t width, const TraceCfg &C, uint32_t p, uint32_t draw) {
  const int SS = C.sub_sample * C.sub_sample;
  const uint32_t pix = p / SS;
  const int X = C.x0 + (int)(pix % C.w), Y = C.y0 + (int)(pix / C.w);
  RngStream rng; rng.init(C.seed, RT_DOMAIN_RENDER, (uint32_t)(Y * (int)width + X), p % SS, draw);
  return rng;
}

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
  T phi = two_pi<T>() * rng.next<T>();
  return (cosT * v + (sinT * rt_cos(phi)) * m) + (sinT * rt_sin(phi)) * n;
}

// ray.js's traceRay blends at every level: colour = clamp(round((1 - refl) * shapeCol + refl * child)).
// A path therefore leaves one record per level -- (1 - refl) * shapeCol (rgb) and refl -- in
// levels[(level * n_paths + p) * 4 ..], ends with a terminal colour in P.out and its level count in
// P.depth; k_resolve_a substitutes the terminal colour back up through the records.
// Returns true when the path goes on with a new segment.
template <typename T>
RT_D bool shade_body_a(const DevScene<T> &S, const PathSoA<T> &P, const TraceCfg &C, T *levels, uint32_t p, T t, int idx, ShadeTally &tally) {
  const T eps = T(C.epsilon);
  const int level = P.depth[p];
  T *out = P.out + (size_t)p * 4;
  if (idx < 0 || idx >= (int)S.n_shapes) {                                       // ray.js:992 (the second test is never true)
    out[0] = T(16); out[1] = T(16); out[2] = T(16); out[3] = T(1);
    return false;
  }
  if (level < 0 || level > C.max_depth) {                                        // (never true; guards the record store)
    out[0] = T(0); out[1] = T(0); out[2] = T(0); out[3] = T(1); tally.overflows++;
    return false;
  }
  const V3<T> o = mk<T>(P.ox[p], P.oy[p], P.oz[p]), d = mk<T>(P.dx[p], P.dy[p], P.dz[p]);
  const T max_dist = P.max_dist[p];
  const V3<T> hit = o + t * d;
  const V3<T> sc = shape_colour<T>(S, idx, hit);
  const V3<T> n = shape_normal<T>(S, idx, hit);
  const T refl = S.surface ? S.surface[2 * (size_t)idx] : Limits<T>::nan();
  T *rec = levels + ((size_t)level * C.n_paths + p) * 4;
  if (t < max_dist && level < C.max_depth) {                                     // ray.js:893, depth = max_depth - level
    const T cos1 = -dot(d, n);
    const int mat = S.meta[4 * idx + 1] & 7;
    const bool has_mat = S.meta[4 * idx + 1] < 8;
    RngStream rng = path_rng(S.width, C, p, P.draw[p]);
    V3<T> ndir;
    if (!(has_mat && S.mat[0][mat] == T(1))) {                                   // MAT_OPAQUE ray.js:895
      if (cos1 < T(0)) {                                                         // ray.js:897-907: undefined, or black far away
        const T v = (t < T(1e5)) ? Limits<T>::nan() : T(0);
        out[0] = v; out[1] = v; out[2] = v; out[3] = v;
        return false;
      }
      const V3<T> rdir = d + (T(2) * cos1) * n;
      ndir = rdir;
      if (rng.next_k() >= S.spec_gt_k[idx]) {                                    // Math.random() > specular, ray.js:910-923
        for (int attempt = 0; attempt < 100; attempt++) {
          ndir = vec_perturb_a<T>(rdir, eps, rng);
          if (dot(ndir, n) > eps) break;
          if (attempt == 99) ndir = rdir;
        }
      }
    } else {                                                                     // ray.js:925-977, no Fresnel term
      MatStack st; st.bits = P.stack[p];
      const T eta1 = st.len() >= 1 ? S.mat[1][st.top()] : Limits<T>::nan();
      const T eta2 = (dot(d, n) < T(0)) ? S.mat[1][mat] : (st.len() >= 2 ? S.mat[1][st.at(st.len() - 2)] : Limits<T>::nan());
      const T ratio = eta1 / eta2;                                               // an undefined index flows on as NaN
      const T c2sq = T(1) - ratio * ratio * (T(1) - cos1 * cos1);
      if (c2sq < T(0)) ndir = d + (T(2) * cos1) * n;
      else {
        const T pm = (cos1 < T(0)) ? T(-1) : T(1);
        ndir = ratio * d + (ratio * cos1 - pm * rt_sqrt(c2sq)) * n;
        if (cos1 < T(0)) { if (st.len() > 0) st.pop(); }
        else if (!st.push(mat)) tally.overflows++;
        P.stack[p] = st.bits;
      }
    }
    rec[0] = (T(1) - refl) * sc.x; rec[1] = (T(1) - refl) * sc.y; rec[2] = (T(1) - refl) * sc.z; rec[3] = refl;
    P.depth[p] = level + 1;
    const T imp = P.importance[p] * refl;
    if (imp < T(0.01)) {                                                         // the child returns COL_WHITE, ray.js:845-847
      out[0] = T(255); out[1] = T(255); out[2] = T(255); out[3] = T(1);
      return false;
    }
    ndir = normalize<T>(ndir, eps);
    P.ox[p] = hit.x; P.oy[p] = hit.y; P.oz[p] = hit.z; P.dx[p] = ndir.x; P.dy[p] = ndir.y; P.dz[p] = ndir.z;
    P.max_dist[p] = max_dist - t; P.importance[p] = imp; P.draw[p] = rng.draw;
    return true;
  }
  rec[0] = (T(1) - refl) * sc.x; rec[1] = (T(1) - refl) * sc.y; rec[2] = (T(1) - refl) * sc.z; rec[3] = refl;
  P.depth[p] = level + 1;
  out[0] = sc.x; out[1] = sc.y; out[2] = sc.z; out[3] = T(1);                    // transmittedColour = shapeCol, ray.js:891
  return false;
}

RT_D void flush_tally(const ShadeTally &t, TraceCounters *ctr) {
  if (t.shadow_rays) atomicAdd(&ctr->shadow_rays, t.shadow_rays);
  if (t.gathers) atomicAdd(&ctr->gathers, t.gathers);
  if (t.overflows) atomicAdd(&ctr->stack_overflows, t.overflows);
}

// Wavefront shade pass: one traceRay body per queued path.  Survivors are recorded as one keep-mask per
// wave plus one count per workgroup; k_scan_blocks and k_queue_scatter then build the next queue in order.
template <typename T, bool LDS, bool WHITTED>
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
    else if (WHITTED) alive = shade_body_a<T>(S, P, C, levels, p, P.hit_t[p], P.hit_idx[p], tally);
    else alive = shade_body<T>(S, G, P, C, Md, Mc, p, P.hit_t[p], P.hit_idx[p], bounce, tally);
    //                                 // (never taken; guards the indexed accesses)
  }
  block_publish_masks(alive, wave_mask, block_count);
  flush_tally(tally, ctr);
}

// Tail of a frame: once few paths are left, a wavefront pass per bounce is all launch and
// synchronisation overhead, so one launch follows every remaining path to its end.  A workgroup of
// RT_TAIL_THREADS lanes takes that many queued paths; after every bounce the survivors' path indices
// are compacted to the front of an LDS queue (the path state itself lives in the SoA arrays, so only
// the index moves), and waves that are left without a path retire.  The loop is bounded by max_bounces.
// Segments traced here are counted in ctr->tail_rays.
#define RT_TAIL_THREADS 512
template <typename T, bool LDS, bool WHITTED>
__global__ void __launch_bounds__(RT_TAIL_THREADS) k_tail(DevScene<T> S, PathSoA<T> P, TraceCfg C, DevMap Md, DevMap Mc, T *levels,
                                                          const uint32_t *queue, const uint32_t *count_ptr, TraceCounters *ctr, int bounce0) {
  extern __shared__ __align__(16) unsigned char rt_lds[];
  __shared__ uint32_t live[2][RT_TAIL_THREADS];
  __shared__ uint32_t wave_cnt[RT_TAIL_THREADS / 64];
  const uint32_t count = min(*count_ptr, C.n_paths);
  if (blockIdx.x * RT_TAIL_THREADS >= count) return;
  const ScanSrc<T> G = LDS ? stage_scene_lds<T>(S, rt_lds) : global_scan_src<T>(S);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t n_live = min((uint32_t)RT_TAIL_THREADS, count - blockIdx.x * RT_TAIL_THREADS);
  if (threadIdx.x < n_live) live[0][threadIdx.x] = queue[blockIdx.x * RT_TAIL_THREADS + threadIdx.x];
  __syncthreads();
  ShadeTally tally;
  unsigned long long rays = 0;
  int cur = 0;
  for (int b = bounce0; n_live > 0 && b < C.max_bounces; b++) {
    // waves 0 .. nw-1 are still running this iteration; the others have retired and their wave_cnt slots are stale
    const int nw = (int)((n_live + 63u) / 64u);
    bool alive = false; uint32_t p = 0;
    if (threadIdx.x < n_live) {
      p = live[cur][threadIdx.x];
      if (p < C.n_paths) {                                  // (always true; guards the indexed accesses below)
        V3<T> o = mk<T>(P.ox[p], P.oy[p], P.oz[p]), d = mk<T>(P.dx[p], P.dy[p], P.dz[p]);
        T t;
        int idx = scan_scene<T, WHITTED ? 3 : 0>(S, G, o, d, T(C.epsilon), T(0), t);
        rays++;
        if (WHITTED) alive = shade_body_a<T>(S, P, C, levels, p, t, idx, tally);
        else alive = shade_body<T>(S, G, P, C, Md, Mc, p, t, idx, b, tally);
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
    if (alive && slot < RT_TAIL_THREADS) live[cur ^ 1][slot] = p;
    __syncthreads();
    cur ^= 1; n_live = min(total, (uint32_t)RT_TAIL_THREADS);
    if ((uint32_t)(wv * 64) >= n_live) break;              // this wave has no path left (retired waves do not hold up barriers)
  }
  for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
  if (lane == 0 && rays) atomicAdd(&ctr->tail_rays, rays);
  flush_tally(tally, ctr);
}

RT_D double clamp01(double x, double lo, double hi) { return (x < lo) ? lo : ((x > hi) ? hi : x); }

// Pixel resolve (rayphoton.js:1610-1623): clamp each sub-sample (`colour()`), sum, floor of the
// mean.  pre = the last sub-sample's unclamped colour; rgba8 = post as bytes.
template <typename T>
__global__ void __launch_bounds__(256) k_resolve(PathSoA<T> P, int n_pixels, int sub_sample, double *pre, double *post, uint8_t *rgba8) {
  int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= n_pixels) return;
  const int SS = sub_sample * sub_sample;
  double tot[4] = { 0, 0, 0, 0 }, last[4] = { 0, 0, 0, 0 };
  for (int s = 0; s < SS; s++) {
    const T *o = P.out + ((size_t)pix * SS + s) * 4;
    for (int c = 0; c < 4; c++) { last[c] = (double)o[c]; tot[c] += clamp01(last[c], 0, c == 3 ? 1 : 255); }
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
__global__ void __launch_bounds__(256) k_resolve_a(PathSoA<T> P, const T *levels, uint32_t n_paths, int max_depth, int n_pixels,
                                                   int sub_sample, double *pre, double *post, uint8_t *rgba8) {
  int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= n_pixels) return;
  const int SS = sub_sample * sub_sample;
  double tot[4] = { 0, 0, 0, 0 }, last[4] = { 0, 0, 0, 0 };
  bool bad = false;
  for (int s = 0; s < SS; s++) {
    const size_t p = (size_t)pix * SS + s;
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
