// Geometry kernel: ray / primitive distance for the eight primitive types, surface normals,
// albedo lookup and the three scene scans (closest, closest-without-lights, any-hit).
// Templated on the working precision; the fp64 instantiation follows the reference's
// operation order (Ray.intersectDist, rayphoton.js:490-644) so that it rounds identically.
#pragma once
#include "rt_math.h"
#include "rt_trig.h"
#include "../../include/rt355.h"

#define RT_MAX_LIGHTS 16
#define RT_GEOM_STRIDE 12

// Node of the hierarchy, nodes in depth-first order: the first child of an interior node is the next node, and `skip`
// is the node to go on with when this node's box is missed or its subtree is done (n_bvh_nodes = end).  No stack.
struct BvhNode {
  float lo[3]; uint32_t skip;
  float hi[3]; uint32_t leaf;     // 0 for interior nodes, else first entry in bvh_prims | count << 24
};

// Everything traceRay needs to know about the shape a ray hit, gathered at upload so that shading fetches it in ONE
// batch of loads: following meta byte -> material id -> material tables costs a chain of dependent global round
// trips per bounce, which is what a path's latency is made of once few waves are left to hide it.
template <typename T> struct alignas(16) HitRec {
  T geom[RT_GEOM_STRIDE];     // the shape's geometry slots (normals, procedural colours)
  T colour[4];
  T transparent, refr_index, reflectance, specular;      // the shape's material; NaN = undefined (all four without a material)
  uint32_t refl_k, spec_k;    // integer thresholds of `U < reflectance`, `U < specular` (0 without a material)
  uint32_t meta;              // type | material byte << 8 | flags << 16 | proc << 24
  uint32_t spec_gt_k;         // ray.js: smallest 24-bit draw k with k * 2^-24 > specular
  T surf_refl, pad_[3];       // ray.js shape model: per-shape reflectance (NaN without one)
  RT_D int type() const { return (int)(meta & 255u); }
  RT_D int mat_byte() const { return (int)((meta >> 8) & 255u); }
  RT_D int flags() const { return (int)((meta >> 16) & 255u); }
  RT_D int proc() const { return (int)(meta >> 24); }
};

// Device view of an uploaded scene.  Shapes are kept twice: in the host's order (for shading:
// normals, colours, materials by original index) and sorted by primitive type (for the scan
// loops: one branch-free inner loop per type, wave-uniform primitive loads).
template <typename T> struct DevScene {
  uint32_t n_shapes, n_lights;
  const uint8_t *meta;        // n x 4, original order
  const T *geom;              // n x 12, original order
  const T *colour;            // n x 4, original order
  const T *sgeom;             // n x 12, type-sorted
  const uint32_t *sorig;      // sorted slot -> original index
  const uint32_t *sflags32;   // sorted slot -> flags
  uint32_t seg[RT_NUM_TYPES + 1];   // type t occupies sorted slots [seg[t], seg[t+1])
  uint32_t light_shape[RT_MAX_LIGHTS];
  T light_data[RT_MAX_LIGHTS][8];   // wattage area prob r g b a reserved
  T camera[16];
  uint32_t width, height;
  T mat[4][8];                // transparent, refrIndex, reflectance, specular (NaN = undefined)
  uint32_t refl_k[8], spec_k[8];    // integer thresholds of `U < reflectance`, `U < specular`
  const HitRec<T> *hit;       // n, original order
  const T *surface;           // n x 2 per-shape reflectance, specular (ray.js shape model) or nullptr
  const uint32_t *spec_gt_k;  // per shape: smallest 24-bit draw k with k * 2^-24 > specular
  // bounding-volume hierarchy over the bounded primitives (scenes too large for the all-primitives scan), or nullptr
  const BvhNode *bvh;
  const uint32_t *bvh_prims;  // leaf entries: sorted slot | type << 28
  const uint32_t *unbounded;  // entries of the same form for planes and infinite cylinders: always tested
  uint32_t n_bvh_nodes, n_unbounded;
  uint32_t bvh_octant_stride;    // nodes between the per-octant copies of the hierarchy (0: one copy)
  const float4 *sbound;       // per sorted slot: bounding sphere (centre, radius; +inf = unbounded), fp32 build only, or nullptr
  const float *sot_cam;       // per sorted slot: the origin terms (RT_OT_STRIDE floats) for rays that start at the camera, fp32 build only, or nullptr
  // fp32 build: GROUPS of primitives that lie close together.  A scan first finds the groups whose bounding sphere some
  // ray of its wave reaches and skips the sorted slots of the others (scan_fast).  gsphere[g] = centre, padded radius
  // of group g = 1 .. n_groups; gmask[g] = its sorted slots as a bit mask (slots 0 .. 63; later slots are always
  // tested); gmask[0] = the slots outside every group.
  uint32_t n_groups;          // <= RT_MAX_GROUPS
  const float4 *gsphere;
  const uint2 *gmask;
};
#define RT_MAX_GROUPS 31

// A miss is reported as -1 (every accepted distance is > epsilon > 0).
template <typename T> RT_D T miss() { return T(-1); }

// qRoots (rayphoton.js:1883-1898).  a == 0 makes the reference return a scalar, which its
// callers index as an array and get `undefined`: a miss.
template <typename T> RT_D bool q_roots(T a, T halfB, T c, T &t0, T &t1) {
  if (a == T(0)) return false;
  T disc = halfB * halfB - a * c;
  if (disc < T(0)) return false;
  disc = rt_sqrt(disc);
  t0 = (-halfB - disc) / a; t1 = (-halfB + disc) / a;
  return true;
}

// a == 1 (sphere, hemisphere): the divisions by a are exact and dropped
template <typename T> RT_D bool q_roots_unit(T halfB, T c, T &t0, T &t1) {
  T disc = halfB * halfB - c;
  if (disc < T(0)) return false;
  disc = rt_sqrt(disc);
  t0 = -halfB - disc; t1 = -halfB + disc;
  return true;
}

template <typename T> RT_D T isect_plane(V3<T> o, V3<T> d, const T *g, T eps) {           // :492-501
  V3<T> a = ld3(g), n = ld3(g + 3);
  T dn = dot(d, n);
  if (rt_abs(dn) < eps) return miss<T>();
  T t = dot(a - o, n) / dn;
  return t > eps ? t : miss<T>();
}
template <typename T> RT_D T isect_sphere(V3<T> o, V3<T> d, const T *g, T eps) {          // :502-519
  V3<T> oc = o - ld3(g); T r = g[6], t0, t1;
  T halfB = dot(d, oc);
  T c = sqlen(oc) - r * r;
  if (!q_roots_unit<T>(halfB, c, t0, t1)) return miss<T>();
  if (t0 > eps) return t0;
  if (t1 > eps) return t1;
  return miss<T>();
}
template <typename T> RT_D T isect_hemisphere(V3<T> o, V3<T> d, const T *g, int flags, T eps) {   // :520-539
  V3<T> c = ld3(g), n = ld3(g + 3); T r = g[6], tmin = g[7], tmax = g[8], t0, t1;
  V3<T> oc = o - c;
  T halfB = dot(d, oc);
  T cc = sqlen(oc) - r * r;
  if (!q_roots_unit<T>(halfB, cc, t0, t1)) return miss<T>();
  bool has_max = flags & RT_FLAG_TRUNC_MAX;
  if (t0 > eps) {
    T proj = dot((o + t0 * d) - c, n);
    if (proj > tmin && (!has_max || proj < tmax)) return t0;
  }
  if (t1 > eps) {
    T proj = dot((o + t1 * d) - c, n);
    if (proj > tmin && (!has_max || proj < tmax)) return t1;
  }
  return miss<T>();
}
template <typename T> RT_D T isect_cylinder(V3<T> o, V3<T> d, const T *g, int flags, T eps) {     // :540-567
  V3<T> c = ld3(g), axis = ld3(g + 3); T r = g[6], h = g[7], t0, t1;
  V3<T> v = o - c;
  T vd = dot(v, d), va = dot(v, axis), da = dot(d, axis);
  T a = T(1) - da * da;
  T halfB = vd - va * da;
  T cc = sqlen(v) - va * va - r * r;
  if (!q_roots<T>(a, halfB, cc, t0, t1)) return miss<T>();
  if (flags & RT_FLAG_FIRST_ROOT) return (t0 <= T(0)) ? miss<T>() : t0;       // ray.js:458-461
  bool finite = flags & RT_FLAG_HALF_HEIGHT;
  if (t0 > eps) {
    if (!finite) return t0;
    if (rt_abs(dot((o + t0 * d) - c, axis)) < h - eps) return t0;
  }
  if (t1 > eps) {
    if (!finite) return t1;
    if (rt_abs(dot((o + t1 * d) - c, axis)) < h - eps) return t1;
  }
  return miss<T>();
}
// Moeller-Trumbore for triangle (:568-588) and parallelogram "square" (:589-609)
template <typename T, bool SQUARE> RT_D T isect_tri(V3<T> o, V3<T> d, const T *g, T eps) {
  V3<T> A = ld3(g), AB = ld3(g + 3), AC = ld3(g + 6);
  V3<T> h = cross(d, AC);
  T a = dot(AB, h);
  if (a > -eps && a < eps) return miss<T>();
  T f = T(1) / a;
  V3<T> s = o - A;
  T u = f * dot(s, h);
  if (u < T(0) || u > T(1)) return miss<T>();
  V3<T> q = cross(s, AB);
  T v = f * dot(d, q);
  if (SQUARE) { if (v < T(0) || v > T(1)) return miss<T>(); }
  else { if (v < T(0) || u + v > T(1)) return miss<T>(); }
  T t = f * dot(AC, q);
  return t > eps ? t : miss<T>();
}
template <typename T, bool ANNULUS> RT_D T isect_disc(V3<T> o, V3<T> d, const T *g, T eps) {      // :610-638
  V3<T> c = ld3(g), n = ld3(g + 3);
  T dn = dot(d, n);
  if (rt_abs(dn) < eps) return miss<T>();
  T t = dot(c - o, n) / dn;
  if (!(t > eps)) return miss<T>();
  T rSq = sqlen(c - (o + t * d));
  if (ANNULUS) return (rSq + eps < g[6] * g[6] && rSq - eps > g[7] * g[7]) ? t : miss<T>();
  return (rSq + eps < g[6] * g[6]) ? t : miss<T>();
}

template <typename T> RT_D T isect_any_type(V3<T> o, V3<T> d, int type, int flags, const T *g, T eps) {
  switch (type) {
    case RT_PLANE: return isect_plane<T>(o, d, g, eps);
    case RT_SPHERE: return isect_sphere<T>(o, d, g, eps);
    case RT_HEMISPHERE: return isect_hemisphere<T>(o, d, g, flags, eps);
    case RT_CYLINDER: return isect_cylinder<T>(o, d, g, flags, eps);
    case RT_TRIANGLE: return isect_tri<T, false>(o, d, g, eps);
    case RT_SQUARE: return isect_tri<T, true>(o, d, g, eps);
    case RT_DISC: return isect_disc<T, false>(o, d, g, eps);
    case RT_ANNULUS: return isect_disc<T, true>(o, d, g, eps);
  }
  return miss<T>();
}

// ------------------------------------------------------------------ fp32 scan records
// The fp32 build does not follow the reference's operation order (it is held to a tolerance, not to bit
// parity), so its scans read per-primitive records that the host precomputes in fp64 (rt_fast_record in
// rt_api.hip) and evaluate a shorter, branch-free form of every test:
//   plane     n[0-2]  k = a.n [3]
//   sphere    c[0-2]  r^2 [3]
//   hemisphere c[0-2] r^2 [3]  n[4-6]  truncMin [7]  truncMax or +inf [8]
//   cylinder  c[0-2]  r^2 [3]  axis[4-6]  halfHeight or +inf [7]
//   triangle / square   N = AC x AB [0-2]  A.N [3]  U*[4-6]  A.U* [7]  V*[8-10]  A.V* [11]
//             (U*, V*: the in-plane dual basis of AB, AC, so that u = (P - A).U*, v = (P - A).V*)
//   disc / annulus      n[0-2]  c.n [3]  c[4-6]  outer^2 [7]  inner^2 [8]
// Everything that depends on the ray origin only ("origin terms", RT_OT_STRIDE floats per primitive) is
// split off: secondary rays compute it per test, the primary pass -- where every ray starts at the camera --
// computes it once per workgroup into LDS.
#define RT_OT_STRIDE 5

RT_D float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
RT_D float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }       // NaN for x < 0: the caller's compares reject it
RT_D float dot3(const float *f, V3<float> v) { return f[0] * v.x + f[1] * v.y + f[2] * v.z; }

template <int TYPE> RT_D void origin_terms(const float *f, V3<float> o, float *ot) {
  if (TYPE == RT_PLANE) {
    ot[0] = f[3] - dot3(f, o);
  } else if (TYPE == RT_SPHERE || TYPE == RT_HEMISPHERE) {
    V3<float> oc = o - ld3(f);
    ot[0] = oc.x; ot[1] = oc.y; ot[2] = oc.z; ot[3] = sqlen(oc) - f[3];
    if (TYPE == RT_HEMISPHERE) ot[4] = dot3(f + 4, oc);
  } else if (TYPE == RT_CYLINDER) {
    V3<float> v = o - ld3(f);
    float va = dot3(f + 4, v);
    ot[0] = v.x; ot[1] = v.y; ot[2] = v.z; ot[3] = va; ot[4] = sqlen(v) - va * va - f[3];
  } else if (TYPE == RT_TRIANGLE || TYPE == RT_SQUARE) {
    ot[0] = f[3] - dot3(f, o); ot[1] = dot3(f + 4, o) - f[7]; ot[2] = dot3(f + 8, o) - f[11];
  } else {                                                  // disc, annulus
    V3<float> oc = o - ld3(f + 4);
    ot[0] = f[3] - dot3(f, o); ot[1] = oc.x; ot[2] = oc.y; ot[3] = oc.z;
  }
}
RT_D void origin_terms_any(int type, const float *f, V3<float> o, float *ot) {
  switch (type) {
    case RT_PLANE: origin_terms<RT_PLANE>(f, o, ot); break;
    case RT_SPHERE: origin_terms<RT_SPHERE>(f, o, ot); break;
    case RT_HEMISPHERE: origin_terms<RT_HEMISPHERE>(f, o, ot); break;
    case RT_CYLINDER: origin_terms<RT_CYLINDER>(f, o, ot); break;
    case RT_TRIANGLE: case RT_SQUARE: origin_terms<RT_SQUARE>(f, o, ot); break;
    default: origin_terms<RT_DISC>(f, o, ot); break;
  }
}

// distance along direction d, or a value that fails `t > eps` (-1 or NaN) on a miss
template <int TYPE> RT_D float isect_fast(const float *f, const float *ot, V3<float> d, float eps, int flags) {
  const float MISS = -1.0f;
  if (TYPE == RT_PLANE) {
    float dn = dot3(f, d);
    float t = ot[0] * rcp_fast(dn);
    return fabsf(dn) < eps ? MISS : t;
  } else if (TYPE == RT_SPHERE) {
    float hb = d.x * ot[0] + d.y * ot[1] + d.z * ot[2];
    float sq = sqrt_fast(hb * hb - ot[3]);
    float t0 = -hb - sq, t1 = -hb + sq;
    return t0 > eps ? t0 : t1;
  } else if (TYPE == RT_HEMISPHERE) {
    float hb = d.x * ot[0] + d.y * ot[1] + d.z * ot[2];
    float sq = sqrt_fast(hb * hb - ot[3]);
    float t0 = -hb - sq, t1 = -hb + sq;
    float dn = dot3(f + 4, d);
    float p0 = ot[4] + t0 * dn, p1 = ot[4] + t1 * dn;
    const bool ok0 = (t0 > eps) & (p0 > f[7]) & (p0 < f[8]);
    const bool ok1 = (t1 > eps) & (p1 > f[7]) & (p1 < f[8]);
    return ok0 ? t0 : (ok1 ? t1 : MISS);
  } else if (TYPE == RT_CYLINDER) {
    float da = dot3(f + 4, d);
    // The quadratic through the part of d across the axis, dp = d - (d.axis) axis: a = |dp|^2 and halfB = v.dp are the
    // reference's 1 - (d.axis)^2 and v.d - (v.axis)(d.axis) without their cancellation (for a ray 0.1 degrees off the
    // axis 1 - da^2 is 3e-6 with 1.2e-7 of rounding: distances 4 % off; |dp|^2 has 2e-4 there).
    float px = d.x - da * f[4], py = d.y - da * f[5], pz = d.z - da * f[6];
    float a = px * px + py * py + pz * pz;
    float hb = px * ot[0] + py * ot[1] + pz * ot[2];
    float sq = sqrt_fast(hb * hb - a * ot[4]);
    float ia = rcp_fast(a);
    float t0 = (-hb - sq) * ia, t1 = (-hb + sq) * ia;
    // a == 0 is the reference's "ray parallel to the axis: miss" (qRoots, rayphoton.js:1884-1889); dp carries about 1e-7
    // of rounding per component, so a ray within 1e-5 rad of the axis (a < 1e-10) counts as parallel
    const bool parallel = a < 1e-10f;
    if (flags & RT_FLAG_FIRST_ROOT) return (parallel | !(t0 > 0.0f)) ? MISS : t0;      // wave-uniform branch
    float hlim = f[7] - eps;
    const bool ok0 = (t0 > eps) & (fabsf(ot[3] + t0 * da) < hlim);
    const bool ok1 = (t1 > eps) & (fabsf(ot[3] + t1 * da) < hlim);
    float t = ok0 ? t0 : (ok1 ? t1 : MISS);
    return parallel ? MISS : t;
  } else if (TYPE == RT_TRIANGLE || TYPE == RT_SQUARE) {
    float dN = dot3(f, d);
    float t = ot[0] * rcp_fast(dN);
    float u = ot[1] + t * dot3(f + 4, d), v = ot[2] + t * dot3(f + 8, d);
    const bool ok = !(fabsf(dN) < eps) & (u >= 0.0f) & (u <= 1.0f) & (v >= 0.0f) & (TYPE == RT_SQUARE ? v <= 1.0f : u + v <= 1.0f);
    return ok ? t : MISS;
  } else {                                                  // disc, annulus
    float dn = dot3(f, d);
    float t = ot[0] * rcp_fast(dn);
    float hx = ot[1] + t * d.x, hy = ot[2] + t * d.y, hz = ot[3] + t * d.z;
    float rSq = hx * hx + hy * hy + hz * hz;
    bool ok = !(fabsf(dn) < eps) & (rSq + eps < f[7]);
    if (TYPE == RT_ANNULUS) ok = ok & (rSq - eps > f[8]);
    return ok ? t : MISS;
  }
}

// Where a scan reads the type-sorted primitives from: the scene's global arrays (the primitive index
// is wave-uniform, so these become scalar loads) or a copy the workgroup staged in LDS.  `ot`: staged
// origin terms of the fp32 records for the one origin all rays of the workgroup share, or nullptr.
// A primitive's record and staged origin terms in registers, fetched whole before anything is computed: left to
// itself the compiler reads the fields where they are first used, inside the tests' conditions, and a wave then
// pays an LDS round trip per field group instead of one per primitive.
template <int TYPE> RT_D void load_record(const float *f, float *rec) {
  constexpr int NV = (TYPE == RT_PLANE || TYPE == RT_SPHERE) ? 1 : (TYPE == RT_CYLINDER ? 2 : 3);     // 16-byte pieces in use
  const float4 *f4 = (const float4 *)f;
#pragma unroll
  for (int k = 0; k < NV; k++) { const float4 v = f4[k]; rec[4 * k] = v.x; rec[4 * k + 1] = v.y; rec[4 * k + 2] = v.z; rec[4 * k + 3] = v.w; }
}
template <int TYPE> RT_D void load_origin_terms(const float *ot, float *out) {
  constexpr int N = (TYPE == RT_PLANE) ? 1 : (TYPE == RT_SPHERE || TYPE == RT_DISC || TYPE == RT_ANNULUS) ? 4 : (TYPE == RT_TRIANGLE || TYPE == RT_SQUARE) ? 3 : 5;
#pragma unroll
  for (int k = 0; k < N; k++) out[k] = ot[k];
}

template <typename T> struct ScanSrc { const T *geom; const uint32_t *orig; const uint32_t *flags; const T *ot; const float4 *gsphere; const uint2 *gmask;
                                       const HitRec<T> *hit; };      // hit: the hit records staged in LDS (fp32, small scenes), or nullptr: read S.hit

// LDS bytes needed to stage n primitives
// (fp32: + origin terms, + the list of primitives a workgroup's primary rays can reach: 9 segment bounds, n slots,
// n flag bytes)
// (fp32: + the scan groups' spheres and slot masks)
constexpr size_t scan_lds_groups_offset(uint32_t n) { return (((size_t)n * (RT_GEOM_STRIDE * 4 + 8 + RT_OT_STRIDE * 4 + 3) + 64) + 15) & ~(size_t)15; }
// (fp32, scenes of up to RT_LDS_HIT_MAX primitives: + the hit records -- shading reads a record per hit, and a round trip
// to global memory per bounce is a large part of a path's latency)
#define RT_LDS_HIT_MAX 128u
constexpr size_t scan_lds_hits_offset(uint32_t n) { return scan_lds_groups_offset(n) + (RT_MAX_GROUPS + 1) * 24; }
template <typename T> constexpr size_t scan_lds_bytes(uint32_t n) {
  return sizeof(T) == 4 ? scan_lds_hits_offset(n) + (n <= RT_LDS_HIT_MAX ? (size_t)n * sizeof(HitRec<float>) : 0)
                        : (size_t)n * (RT_GEOM_STRIDE * sizeof(T) + 8);
}
// where that list lives in a workgroup's staging area
RT_D uint32_t *scan_lds_vseg(uint32_t n, void *lds) { return (uint32_t *)((char *)lds + (size_t)n * (RT_GEOM_STRIDE * 4 + 8 + RT_OT_STRIDE * 4)); }
RT_D uint16_t *scan_lds_vis(uint32_t n, void *lds) { return (uint16_t *)(scan_lds_vseg(n, lds) + 16); }

// Cooperative, coalesced copy of the type-sorted scene into LDS (all threads of the block must call it).
// With `origin` (fp32 only) the origin terms of every primitive are staged behind it.
// Every thread issues all its loads of a round (16-byte pieces of the records, four per thread, then the index and
// flag words) before it stores any of them: a small scene costs the workgroup ONE global round trip, not one per
// element.  The origin terms are copied too when `origin` is the camera and the scene holds the camera's terms
// (S.sot_cam, computed at upload by this same code), else computed from the LDS copy.  `extra`: n_extra more words the
// caller wants copied to `extra_dst` (LDS) in the same round trip.
template <typename T> RT_D ScanSrc<T> stage_scene_lds(const DevScene<T> &S, void *lds, const V3<T> *origin = nullptr, const uint32_t *extra = nullptr,
                                                      uint32_t *extra_dst = nullptr, uint32_t n_extra = 0, bool with_hits = true) {
  T *lg = (T *)lds;
  uint32_t *lo = (uint32_t *)(lg + (size_t)S.n_shapes * RT_GEOM_STRIDE), *lf = lo + S.n_shapes;
  const uint32_t n = S.n_shapes, tid = threadIdx.x, nt = blockDim.x;
  const uint32_t n16 = n * (uint32_t)(RT_GEOM_STRIDE * sizeof(T) / 16);           // the records as 16-byte pieces
  const uint4 *src = (const uint4 *)S.sgeom;
  uint4 *dst = (uint4 *)lg;
  const uint32_t o0 = S.sorig[tid < n ? tid : 0], f0 = S.sflags32[tid < n ? tid : 0];     // (first round of the words: in flight with the pieces)
  const uint32_t x0 = (extra && tid < n_extra) ? extra[tid] : 0u;
  uint32_t n_mask_words = 0, n_sphere_words = 0, m0 = 0, g0 = 0;
  if constexpr (sizeof(T) == 4) {
    n_mask_words = 2u * (S.n_groups + 1u); n_sphere_words = 4u * (S.n_groups + 1u);      // (at most 64 and 128: one word per thread)
    m0 = ((const uint32_t *)S.gmask)[tid < n_mask_words ? tid : 0]; g0 = ((const uint32_t *)S.gsphere)[tid < n_sphere_words ? tid : 0];
  }
  bool ot_copied = false;
  float t0 = 0.f;
  if constexpr (sizeof(T) == 4) {
    ot_copied = origin && S.sot_cam;
    if (ot_copied) t0 = S.sot_cam[tid < n * RT_OT_STRIDE ? tid : 0];
  }
  // (fp32, small scenes: the hit records ride along as further 16-byte pieces)
  uint32_t nh16 = 0; const uint4 *hsrc = src; uint4 *hdst = dst;
  if constexpr (sizeof(T) == 4) {
    if (with_hits && n <= RT_LDS_HIT_MAX) { nh16 = n * (uint32_t)(sizeof(HitRec<T>) / 16); hsrc = (const uint4 *)S.hit; hdst = (uint4 *)((char *)lds + scan_lds_hits_offset(n)); }
  }
  const uint32_t nall = n16 + nh16;
  for (uint32_t base = 0; base < nall; base += 4 * nt) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t i = base + (uint32_t)k * nt + tid; v[k] = i < n16 ? src[i] : hsrc[i < nall ? i - n16 : 0]; }
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t i = base + (uint32_t)k * nt + tid; if (i < n16) dst[i] = v[k]; else if (i < nall) hdst[i - n16] = v[k]; }
  }
  if (tid < n) { lo[tid] = o0; lf[tid] = f0; }
  for (uint32_t i = tid + nt; i < n; i += nt) { lo[i] = S.sorig[i]; lf[i] = S.sflags32[i]; }
  if (extra) {
    if (tid < n_extra) extra_dst[tid] = x0;
    for (uint32_t i = tid + nt; i < n_extra; i += nt) extra_dst[i] = extra[i];
  }
  ScanSrc<T> G; G.geom = lg; G.orig = lo; G.flags = lf; G.ot = nullptr; G.gsphere = nullptr; G.gmask = nullptr;
  G.hit = nh16 ? (const HitRec<T> *)hdst : nullptr;
  if constexpr (sizeof(T) == 4) {
    uint32_t *lsph = (uint32_t *)((char *)lds + scan_lds_groups_offset(n)), *lm = lsph + 4u * (RT_MAX_GROUPS + 1);
    if (tid < n_sphere_words) lsph[tid] = g0;
    for (uint32_t i = tid + nt; i < n_sphere_words; i += nt) lsph[i] = ((const uint32_t *)S.gsphere)[i];
    if (tid < n_mask_words) lm[tid] = m0;
    for (uint32_t i = tid + nt; i < n_mask_words; i += nt) lm[i] = ((const uint32_t *)S.gmask)[i];
    G.gsphere = (const float4 *)lsph; G.gmask = (const uint2 *)lm;
  }
  if constexpr (sizeof(T) == 4) {
    if (ot_copied) {
      float *lt = (float *)(lf + n);
      if (tid < n * RT_OT_STRIDE) lt[tid] = t0;
      for (uint32_t i = tid + nt; i < n * RT_OT_STRIDE; i += nt) lt[i] = S.sot_cam[i];
      G.ot = lt;
    } else if (origin) {
      __syncthreads();
      float *lt = (float *)(lf + n);
      for (uint32_t i = tid; i < n; i += nt) {
        int type = 0;
        while (type < RT_NUM_TYPES - 1 && i >= S.seg[type + 1]) type++;
        origin_terms_any(type, lg + (size_t)i * RT_GEOM_STRIDE, *origin, lt + (size_t)i * RT_OT_STRIDE);
      }
      G.ot = lt;
    }
  }
  __syncthreads();
  return G;
}
template <typename T> RT_D ScanSrc<T> global_scan_src(const DevScene<T> &S) {
  ScanSrc<T> G; G.geom = S.sgeom; G.orig = S.sorig; G.flags = S.sflags32; G.ot = nullptr; G.gsphere = S.gsphere; G.gmask = S.gmask; G.hit = nullptr;
  return G;
}

// bit mask of the sorted slots below s (s = 0 .. 64; a shift by 64 is undefined, and a type with no primitives behind the
// last slot of a 64-primitive scene has s == 64 at both ends of its range)
RT_D unsigned long long low_slots(uint32_t s) { return s >= 64u ? ~0ull : ((1ull << s) - 1ull); }

#ifndef RT_SCAN_UNROLL
#define RT_SCAN_UNROLL 2
#endif
#define RT_PRAGMA_STR_(x) _Pragma(#x)
#define RT_PRAGMA_(x) RT_PRAGMA_STR_(x)
#define RT_SCAN_UNROLL_PRAGMA RT_PRAGMA_(unroll RT_SCAN_UNROLL)
// fp32 scan over the precomputed records; same MODEs as scan_scene below.  Equal distances keep the first
// primitive in type-sorted order.
template <int MODE, bool OT> RT_D int scan_fast(const DevScene<float> &S, const ScanSrc<float> &G, V3<float> o, V3<float> d, float eps, float limit, float &t_out,
                                                uint32_t *tests = nullptr) {
  float best = __builtin_huge_valf(); int best_s = -1;
  const float lim = (MODE == 2) ? limit - eps : 0.0f;
  bool occluded = false;
  // Sorted slots (0 .. 63) whose primitives some ray of this wave can reach: those outside every group, and those of
  // the groups whose sphere a ray reaches.  A ray reaches a sphere if its line passes within the radius and the centre
  // lies ahead of the origin or the origin inside; the distance from the line is taken of q = oc - (oc.d) d itself,
  // whose rounding error grows with |oc|, not |oc|^2, and the radii are padded, so a wave never skips a primitive that
  // one of its rays would hit (d is a unit vector).  The next group's sphere is fetched while this one is tested.
  unsigned long long act = ~0ull;
  if (S.n_groups) {
    const uint2 m0 = G.gmask[0];
    act = (unsigned long long)m0.x | ((unsigned long long)m0.y << 32);
    float4 bs = G.gsphere[1]; uint2 gm = G.gmask[1];
    for (uint32_t g = 1; g <= S.n_groups; g++) {
      const uint32_t gn = g < S.n_groups ? g + 1 : g;
      const float4 bs_next = G.gsphere[gn]; const uint2 gm_next = G.gmask[gn];
      const V3<float> oc = o - mk<float>(bs.x, bs.y, bs.z);
      const float b = dot(oc, d), r2 = bs.w * bs.w;
      const V3<float> q = oc - b * d;
      const bool may = sqlen(q) <= r2 && (b < 0.0f || sqlen(oc) <= r2);
      if (__ballot(may) != 0ull) act |= (unsigned long long)gm.x | ((unsigned long long)gm.y << 32);
      bs = bs_next; gm = gm_next;
    }
    // (statistics: the primitives this closest-hit scan goes on to test)
    if (MODE != 2 && tests) *tests += (uint32_t)__popcll(S.n_shapes >= 64u ? act : act & ((1ull << S.n_shapes) - 1ull));
  }
#define RT_FAST_BODY(TYPE, SLOT)                                                                        \
  {                                                                                                     \
    const uint32_t s = (SLOT);                                                                          \
    float f[RT_GEOM_STRIDE], ot[RT_OT_STRIDE];                                                          \
    load_record<TYPE>(G.geom + (size_t)s * RT_GEOM_STRIDE, f);                                          \
    const int flags = (MODE == 1 || MODE == 2 || TYPE == RT_CYLINDER) ? (int)G.flags[s] : 0;            \
    if (OT) load_origin_terms<TYPE>(G.ot + (size_t)s * RT_OT_STRIDE, ot);                               \
    const bool counts = !((MODE == 1 || MODE == 2) && (flags & RT_FLAG_LIGHT));                         \
    if (!OT) origin_terms<TYPE>(f, o, ot);                                                              \
    const float t = isect_fast<TYPE>(f, ot, d, eps, flags);                                             \
    if (MODE == 2) occluded = occluded || (counts && t > eps && t < lim);                               \
    else { const bool upd = counts && t > eps && t < best; best = upd ? t : best; best_s = upd ? (int)s : best_s; } \
  }
  if (S.n_shapes <= 64u) {
    // the slots to test as bits, taken two at a time (two tests in flight hide each other's LDS round trips)
#define RT_FAST_LOOP(TYPE)                                                                              \
    {                                                                                                   \
      const uint32_t s0 = S.seg[TYPE], s1 = S.seg[TYPE + 1];                                            \
      unsigned long long m = act & low_slots(s1) & ~low_slots(s0);                                      \
      while (m & (m - 1ull)) {                                                                          \
        const uint32_t sa = (uint32_t)__builtin_ctzll(m); m &= m - 1ull;                                \
        const uint32_t sb = (uint32_t)__builtin_ctzll(m); m &= m - 1ull;                                \
        RT_FAST_BODY(TYPE, sa) RT_FAST_BODY(TYPE, sb)                                                   \
      }                                                                                                 \
      if (m) RT_FAST_BODY(TYPE, (uint32_t)__builtin_ctzll(m))                                           \
    }
    RT_FAST_LOOP(RT_PLANE) RT_FAST_LOOP(RT_SPHERE) RT_FAST_LOOP(RT_HEMISPHERE) RT_FAST_LOOP(RT_CYLINDER)
    RT_FAST_LOOP(RT_TRIANGLE) RT_FAST_LOOP(RT_SQUARE) RT_FAST_LOOP(RT_DISC) RT_FAST_LOOP(RT_ANNULUS)
#undef RT_FAST_LOOP
  } else {
#define RT_FAST_LOOP(TYPE)                                                                              \
    RT_SCAN_UNROLL_PRAGMA                                                                               \
    for (uint32_t s_ = S.seg[TYPE]; s_ < S.seg[TYPE + 1]; s_++) RT_FAST_BODY(TYPE, s_)
    RT_FAST_LOOP(RT_PLANE) RT_FAST_LOOP(RT_SPHERE) RT_FAST_LOOP(RT_HEMISPHERE) RT_FAST_LOOP(RT_CYLINDER)
    RT_FAST_LOOP(RT_TRIANGLE) RT_FAST_LOOP(RT_SQUARE) RT_FAST_LOOP(RT_DISC) RT_FAST_LOOP(RT_ANNULUS)
#undef RT_FAST_LOOP
  }
#undef RT_FAST_BODY
  if (MODE == 2) { t_out = 0.0f; return occluded ? 1 : 0; }
  t_out = best;
  return best_s >= 0 ? (int)G.orig[best_s] : -1;
}

// The same closest-hit scan over a subset: vis[vseg[t] .. vseg[t+1]) are the sorted slots of type t that the rays of
// this workgroup can reach (see cull_primitives in rt_trace.h).  Origin terms are the staged ones.
template <int MODE> RT_D int scan_fast_visible(const ScanSrc<float> &G, const uint16_t *vis, const uint32_t *vseg, V3<float> d, float eps, float &t_out) {
  float best = __builtin_huge_valf(); int best_s = -1;
#define RT_VIS_LOOP(TYPE)                                                                               \
  RT_SCAN_UNROLL_PRAGMA                                                                                 \
  for (uint32_t k = vseg[TYPE]; k < vseg[TYPE + 1]; k++) {                                              \
    const uint32_t s = vis[k];                                                                          \
    float f[RT_GEOM_STRIDE], ot[RT_OT_STRIDE];                                                          \
    load_record<TYPE>(G.geom + (size_t)s * RT_GEOM_STRIDE, f);                                          \
    load_origin_terms<TYPE>(G.ot + (size_t)s * RT_OT_STRIDE, ot);                                       \
    const int flags = (TYPE == RT_CYLINDER) ? (int)G.flags[s] : 0;                                      \
    const float t = isect_fast<TYPE>(f, ot, d, eps, flags);                                             \
    const bool upd = t > eps && t < best; best = upd ? t : best; best_s = upd ? (int)s : best_s;        \
  }
  RT_VIS_LOOP(RT_PLANE) RT_VIS_LOOP(RT_SPHERE) RT_VIS_LOOP(RT_HEMISPHERE) RT_VIS_LOOP(RT_CYLINDER)
  RT_VIS_LOOP(RT_TRIANGLE) RT_VIS_LOOP(RT_SQUARE) RT_VIS_LOOP(RT_DISC) RT_VIS_LOOP(RT_ANNULUS)
#undef RT_VIS_LOOP
  t_out = best;
  return best_s >= 0 ? (int)G.orig[best_s] : -1;
}

RT_D float isect_fast_any(int type, const float *f, V3<float> o, V3<float> d, float eps, int flags) {
  float ot[RT_OT_STRIDE];
  switch (type) {
    case RT_PLANE: origin_terms<RT_PLANE>(f, o, ot); return isect_fast<RT_PLANE>(f, ot, d, eps, flags);
    case RT_SPHERE: origin_terms<RT_SPHERE>(f, o, ot); return isect_fast<RT_SPHERE>(f, ot, d, eps, flags);
    case RT_HEMISPHERE: origin_terms<RT_HEMISPHERE>(f, o, ot); return isect_fast<RT_HEMISPHERE>(f, ot, d, eps, flags);
    case RT_CYLINDER: origin_terms<RT_CYLINDER>(f, o, ot); return isect_fast<RT_CYLINDER>(f, ot, d, eps, flags);
    case RT_TRIANGLE: origin_terms<RT_TRIANGLE>(f, o, ot); return isect_fast<RT_TRIANGLE>(f, ot, d, eps, flags);
    case RT_SQUARE: origin_terms<RT_SQUARE>(f, o, ot); return isect_fast<RT_SQUARE>(f, ot, d, eps, flags);
    case RT_DISC: origin_terms<RT_DISC>(f, o, ot); return isect_fast<RT_DISC>(f, ot, d, eps, flags);
    default: origin_terms<RT_ANNULUS>(f, o, ot); return isect_fast<RT_ANNULUS>(f, ot, d, eps, flags);
  }
}

// The same scans through the hierarchy.  Every primitive whose box the ray can reach before the best hit so far is
// tested with the same distance function as in the all-primitives scan, and equal distances are settled on the
// original index, so the result is the all-primitives scan's (boxes are padded; a NaN slab interval counts as hit).
// (Tried for fp32: every lane first walks on to its next leaf, then the wave tests the leaves type by type with the
// type-specialised tests -- 2.15 against 2.16 ms on the 10 000-primitive 4K frame: the walk waits for its dependent node
// loads, not for divergent tests.  Also tried: the wavefront passes' closest-hit scan as persistent waves whose lanes take the
// next queued ray as soon as their walk ends (the walk as a resumable state, eight nodes at a time) -- 2.73 against 2.17 ms: the
// one-ray-per-lane kernel needs 20 registers and keeps every SIMD full of waves, which a walk that waits needs more than busy lanes.)
template <typename T, int MODE> RT_D int scan_bvh(const DevScene<T> &S, V3<T> o, V3<T> d, T eps, T limit, T &t_out, uint32_t *tests) {
  T best = Limits<T>::inf(); int best_idx = -1;
  const T lim = (MODE == 2) ? limit - eps : T(0);
  uint32_t n_tests = 0;
  auto test = [&](uint32_t entry) -> bool {                 // true: an occluder was found (MODE 2)
    const uint32_t s = entry & 0x0FFFFFFFu; const int type = (int)(entry >> 28);
    if (s >= S.n_shapes) return false;                      // (never true; guards the indexed accesses)
    const int flags = (int)S.sflags32[s];
    if ((MODE == 1 || MODE == 2) && (flags & RT_FLAG_LIGHT)) return false;
    const T *g = S.sgeom + (size_t)s * RT_GEOM_STRIDE;
    T t;
    if constexpr (sizeof(T) == 4) t = isect_fast_any(type, g, o, d, eps, flags);
    else t = isect_any_type<T>(o, d, type, flags, g, eps);
    n_tests++;
    if (MODE == 2) return t > eps && t < lim;
    const int orig = (int)S.sorig[s];
    if (t > eps && (t < best || (t == best && best_idx >= 0 && (MODE == 3 ? orig > best_idx : orig < best_idx)))) { best = t; best_idx = orig; }
    return false;
  };
  bool occluded = false;
  for (uint32_t u = 0; u < S.n_unbounded && !occluded; u++) occluded = test(S.unbounded[u]);
  const T ix = T(1) / d.x, iy = T(1) / d.y, iz = T(1) / d.z;
  // the copy of the node array threaded near child first for this ray's direction octant
  const uint32_t octant = (d.x < T(0) ? 1u : 0u) | (d.y < T(0) ? 2u : 0u) | (d.z < T(0) ? 4u : 0u);
  const BvhNode *nodes = S.bvh + (size_t)octant * S.bvh_octant_stride;
  uint32_t i = 0, guard = 0;
  while (i < S.n_bvh_nodes && !occluded && guard++ < 2u * S.n_bvh_nodes) {
    const BvhNode nd = nodes[i];
    // slabs; fmin/fmax return the other operand for a NaN (0 * inf), which keeps the interval conservative
    T t1 = (T(nd.lo[0]) - o.x) * ix, t2 = (T(nd.hi[0]) - o.x) * ix;
    T tn = rt_fmin(t1, t2), tf = rt_fmax(t1, t2);
    t1 = (T(nd.lo[1]) - o.y) * iy; t2 = (T(nd.hi[1]) - o.y) * iy;
    tn = rt_fmax(tn, rt_fmin(t1, t2)); tf = rt_fmin(tf, rt_fmax(t1, t2));
    t1 = (T(nd.lo[2]) - o.z) * iz; t2 = (T(nd.hi[2]) - o.z) * iz;
    tn = rt_fmax(tn, rt_fmin(t1, t2)); tf = rt_fmin(tf, rt_fmax(t1, t2));
    const T reach = (MODE == 2) ? lim : best;
    const bool hit = !(tf < T(0)) && !(tn > tf) && !(tn > reach);
    if (!hit) { i = nd.skip; continue; }
    if (nd.leaf) {
      const uint32_t first = nd.leaf & 0x00FFFFFFu, count = nd.leaf >> 24;
      for (uint32_t k = 0; k < count && !occluded; k++) occluded = test(S.bvh_prims[first + k]);
      i = nd.skip;
    } else {
      i = i + 1;
    }
  }
  if (tests) *tests += n_tests;
  if (MODE == 2) { t_out = T(0); return occluded ? 1 : 0; }
  t_out = best;
  return best_idx;
}

// Scan all primitives in type-sorted order.  MODE 0: closest incl. lights (rayphoton.js:1642-1650),
// MODE 1: closest excluding lights (:1467-1475), MODE 2: any occluder closer than limit - eps
// (:1853-1857).  The reference scans in index order and keeps the first of equal distances
// (strict `<`), so ties go to the lowest original index here too.  MODE 3: closest incl. lights with ties going
// to the HIGHEST index, for ray.js, which scans in descending index order (ray.js:863).
// OT (fp32 only): G.ot holds the origin terms of `o`, staged by stage_scene_lds.
template <typename T, int MODE, bool OT = false> RT_D int scan_scene(const DevScene<T> &S, const ScanSrc<T> &G, V3<T> o, V3<T> d, T eps, T limit, T &t_out,
                                                                     uint32_t *tests = nullptr) {
  if (S.bvh) return scan_bvh<T, MODE>(S, o, d, eps, limit, t_out, tests);
  if constexpr (sizeof(T) == 4) return scan_fast<MODE, OT>(S, G, o, d, eps, limit, t_out, tests);
  T best = Limits<T>::inf(); int best_idx = -1;
  const T lim = (MODE == 2) ? limit - eps : T(0);
#define RT_SCAN_LOOP(TYPE, EXPR)                                                        \
  _Pragma("unroll 2")                                                                   \
  for (uint32_t s = S.seg[TYPE]; s < S.seg[TYPE + 1]; s++) {                            \
    const T *g = G.geom + (size_t)s * RT_GEOM_STRIDE;                                   \
    const int flags = (int)G.flags[s]; (void)flags;                                     \
    T t = EXPR;                                                                         \
    if ((MODE == 1 || MODE == 2) && (flags & RT_FLAG_LIGHT)) continue;                  \
    if (MODE == 2) { if (t > eps && t < lim) { t_out = t; return 1; } }                 \
    else {                                                                              \
      int orig = (int)G.orig[s];                                                        \
      if (t > eps && (t < best || (t == best && (MODE == 3 ? (best_idx >= 0 && orig > best_idx) : orig < best_idx)))) { best = t; best_idx = orig; } \
    }                                                                                   \
  }
  RT_SCAN_LOOP(RT_PLANE, isect_plane<T>(o, d, g, eps))
  RT_SCAN_LOOP(RT_SPHERE, isect_sphere<T>(o, d, g, eps))
  RT_SCAN_LOOP(RT_HEMISPHERE, isect_hemisphere<T>(o, d, g, flags, eps))
  RT_SCAN_LOOP(RT_CYLINDER, isect_cylinder<T>(o, d, g, flags, eps))
  RT_SCAN_LOOP(RT_TRIANGLE, (isect_tri<T, false>(o, d, g, eps)))
  RT_SCAN_LOOP(RT_SQUARE, (isect_tri<T, true>(o, d, g, eps)))
  RT_SCAN_LOOP(RT_DISC, (isect_disc<T, false>(o, d, g, eps)))
  RT_SCAN_LOOP(RT_ANNULUS, (isect_disc<T, true>(o, d, g, eps)))
#undef RT_SCAN_LOOP
  if (MODE == 2) return 0;
  t_out = best;
  return best_idx;
}

// shape.normal(p) (rayphoton.js:368-371, :385, :403, :417, :433, :449, :464, :480); not renormalised
template <typename T> RT_D V3<T> shape_normal_g(const T *g, int type, int flags, V3<T> p) {
  const T sign = (flags & RT_FLAG_CONVEX) ? T(1) : T(-1);
  switch (type) {
    case RT_TRIANGLE: case RT_SQUARE: return ld3(g + 9);
    case RT_SPHERE: return (T(1) / g[6]) * (p - ld3(g));
    case RT_HEMISPHERE: return (sign / g[6]) * (p - ld3(g));
    case RT_CYLINDER: {
      V3<T> v = p - ld3(g), axis = ld3(g + 3);
      return (sign / g[6]) * (v - dot(v, axis) * axis);
    }
    default: return ld3(g + 3);   // plane, disc, annulus
  }
}

template <typename T> RT_D V3<T> shape_normal(const DevScene<T> &S, int i, V3<T> p) {
  return shape_normal_g<T>(S.geom + (size_t)i * RT_GEOM_STRIDE, S.meta[4 * i], S.meta[4 * i + 2], p);
}

// The hit record of shape i in registers: all of it is requested before any of it is used.
template <typename T> RT_D HitRec<T> load_hit(const DevScene<T> &S, const ScanSrc<T> &G, int i) {
  HitRec<T> r;
  const uint4 *src = (const uint4 *)((G.hit ? G.hit : S.hit) + i);
  uint4 v[sizeof(HitRec<T>) / 16];
#pragma unroll
  for (int k = 0; k < (int)(sizeof(HitRec<T>) / 16); k++) v[k] = src[k];
  __builtin_memcpy(&r, v, sizeof r);
  return r;
}
// entry x (0..7, per lane) of a table of eight wave-uniform values: selects instead of an indexed load
template <typename T> RT_D T pick8(const T *tab, int x) {
  T v = tab[0];
#pragma unroll
  for (int k = 1; k < 8; k++) v = (x == k) ? tab[k] : v;
  return v;
}

// JavaScript ToInt32 of a sum of two floors, as the checker patterns' `& 1` sees it
template <typename T> RT_D int js_parity(T x) { return (int)((long long)x & 1LL); }

// shape.colour(p): the base colour (rayphoton.js:350) or one of the presets' closures
template <typename T> RT_D V3<T> shape_colour_g(const T *g, const T *c, int proc, V3<T> p) {
  if (proc == RT_PROC_NONE) return mk<T>(c[0], c[1], c[2]);
  const V3<T> WHITE = mk<T>(255, 255, 255), BLACK = mk<T>(0, 0, 0);
  switch (proc) {
    case RT_PROC_STRIPE_CHECKER: {                                        // rayphoton.js:746-755
      T f = rt_sin(p.x) + p.y;
      if (f > T(0) && f < T(1)) return mk<T>(255, 32, 144);
      int idx = 1 + js_parity<T>(rt_floor(p.x / T(4)) + rt_floor(p.y / T(4)));
      return idx == 1 ? mk<T>(128, 128, 128) : BLACK;
    }
    case RT_PROC_EIGHT_BALL: {                                            // rayphoton.js:756-768
      V3<T> pc = p - ld3(g); T r = g[6];
      T c1 = dot(pc, normalize<T>(mk<T>(-1, -1, 0), T(1e-6))) / r;
      if (c1 < T(0.905)) return BLACK;
      c1 = dot(pc, normalize<T>(mk<T>(-1, -1, T(0.15)), T(1e-6))) / r;
      T c2 = dot(pc, normalize<T>(mk<T>(-1, -1, T(-0.12)), T(1e-6))) / r;
      return ((c1 > T(0.9920) && c1 < T(0.999)) || (c2 > T(0.9900) && c2 < T(0.9980))) ? BLACK : WHITE;
    }
    case RT_PROC_TILE_GRID: {                                             // rayphoton.js:784-792
      T x = rt_abs(rt_fmod(p.x + T(100.7), T(2.3)) - T(1.15)), y = rt_abs(rt_fmod(p.y + T(102.7), T(2.3)) - T(1.15));
      return (x < T(0.03) || y < T(0.03) || x + y < T(0.6)) ? BLACK : WHITE;
    }
    case RT_PROC_ROTATED_CHECKER: {                                       // rayphoton.js:871-874
      int idx = js_parity<T>(rt_floor((T(0.6) * p.x + T(0.8) * p.y + T(0.7)) / T(3.2)) +
                             rt_floor((T(0.8) * p.x - T(0.6) * p.y + T(0.2)) / T(3.2)));
      return idx ? BLACK : WHITE;
    }
    case RT_PROC_STRIPE_CHECKER3: {                                       // ray.js:650-659
      T f = rt_sin(p.x) + p.y;
      if (f > T(0) && f < T(1)) return mk<T>(255, 32, 144);
      int idx = 1 + js_parity<T>(rt_floor(p.x * T(2))) + js_parity<T>(rt_floor(p.y * T(2)));
      return idx == 1 ? mk<T>(128, 128, 128) : idx == 2 ? BLACK : mk<T>(8, 8, 64);
    }
  }
  return mk<T>(c[0], c[1], c[2]);
}
template <typename T> RT_D V3<T> shape_colour(const DevScene<T> &S, int i, V3<T> p) {
  return shape_colour_g<T>(S.geom + (size_t)i * RT_GEOM_STRIDE, S.colour + (size_t)i * 4, S.meta[4 * i + 3], p);
}
