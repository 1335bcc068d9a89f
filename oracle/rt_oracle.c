/*
 * TEST INFRASTRUCTURE -- CPU oracle (see rt_oracle.h).  Plain C, IEEE fp64, compiled with
 * -ffp-contract=off so that every operation rounds exactly where the reference's JavaScript
 * does.  Each function cites the reference lines it restates.
 *
 * Parity pin: tests/test_oracle_golden.py checks every exported function against fixtures
 * produced by running the reference itself (oracle/harness/gen_golden.js).
 */
#include "rt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ora_fdlibm.h"
#define ORA_SIN fd_sin   /* Math.sin / Math.cos as V8 computes them (fdlibm), not libm */
#define ORA_COS fd_cos

#define PI 3.141592653589793

/* ------------------------------------------------------------------ RNG */
/* Philox4x32-10 (Salmon et al. SC'11, Random123 constants); same stream convention as
 * oracle/harness/philox.js: key=(seed_lo,seed_hi), counter=(draw>>2,a,b,domain),
 * U=(word[draw&3]>>8)*2^-24. */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct rng_t {
  uint32_t key[2], a, b, domain, draw, block_idx, block[4];
  uint64_t total;
} rng_t;

static void rng_set(rng_t *g, uint64_t seed, uint32_t domain, uint32_t a, uint32_t b) {
  g->key[0] = (uint32_t)seed; g->key[1] = (uint32_t)(seed >> 32);
  g->a = a; g->b = b; g->domain = domain; g->draw = 0; g->block_idx = 0xFFFFFFFFu;
}
static double rng_next(rng_t *g) {
  uint32_t bi = g->draw >> 2;
  if (bi != g->block_idx) {
    uint32_t c[4] = { bi, g->a, g->b, g->domain };
    ora_philox4x32_10(c, g->key, g->block);
    g->block_idx = bi;
  }
  uint32_t w = g->block[g->draw & 3];
  g->draw++; g->total++;
  return (double)(w >> 8) * (1.0 / 16777216.0);
}
double ora_uniform(uint64_t seed, uint32_t domain, uint32_t a, uint32_t b, uint32_t draw) {
  rng_t g; rng_set(&g, seed, domain, a, b); g.total = 0; g.draw = draw; return rng_next(&g);
}

void ora_default_params(ora_params *p) {
  p->epsilon = 1e-6; p->max_trace_dist = 20; p->max_depth = 5; p->sub_sample = 1;
  p->num_photons_diffuse = 20000; p->num_photons_caustic = 50000;
  p->diffuse_radius = 0.40; p->caustic_radius = 0.15; p->standard_lamp_area = 100;
  p->seed = 12345; p->use_direct_light = 1; p->use_photon_maps = 1; p->super_sample_block = 0; p->pad_ = 0;
}

/* ------------------------------------------------------------------ vectors (rayphoton.js:104-112) */
typedef struct { double x, y, z; } v3;
static v3 V(double x, double y, double z) { v3 r = { x, y, z }; return r; }
static v3 ld(const double *p) { return V(p[0], p[1], p[2]); }
static v3 plus(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 minus(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 scal(double k, v3 a) { return V(k * a.x, k * a.y, k * a.z); }
static double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 cross(v3 v, v3 w) { return V(v.y * w.z - v.z * w.y, v.z * w.x - v.x * w.z, v.x * w.y - v.y * w.x); }
static double sqlen(v3 a) { return dot(a, a); }
static int is_zero(v3 a, double eps) { return sqlen(a) < eps; }
/* rayphoton.js:110 -- near-zero vectors become [0,0,1] */
static v3 normalize(v3 a, double eps) { return is_zero(a, eps) ? V(0, 0, 1) : scal(1 / sqrt(sqlen(a)), a); }

void ora_normalize(const double v[3], double out[3]) {
  v3 r = normalize(ld(v), 1e-6); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* rayphoton.js:113-122 */
static void orthonormal(v3 v, double eps, v3 *m_out, v3 *n_out) {
  v3 v2 = normalize(v, eps);
  v3 m = V(1, 0, 0);
  if (is_zero(cross(v2, m), eps)) m = V(0, 1, 0);
  v3 n = normalize(cross(v2, m), eps);
  *m_out = cross(n, v2); *n_out = n;
}
/* rayphoton.js:123-137 with maxDeviation undefined (the only live use) */
static v3 vec_perturb(v3 v, double eps, rng_t *g) {
  v3 m, n; orthonormal(v, eps, &m, &n);
  double x = (1 - 0.0) * rng_next(g) + 0.0;
  double cosT = sqrt(x);
  double sinT = sqrt(1 - cosT * cosT);
  double phi = 2 * PI * rng_next(g);
  return plus(plus(scal(cosT, v), scal(sinT * ORA_COS(phi), m)), scal(sinT * ORA_SIN(phi), n));
}
/* rayphoton.js:1921-1926 */
static v3 disc_sample(v3 centre, double radius, v3 normal_dir, double eps, rng_t *g) {
  double r = sqrt(rng_next(g)) * radius;
  double theta = 2 * PI * rng_next(g);
  v3 m, n; orthonormal(normal_dir, eps, &m, &n);
  return plus(plus(centre, scal(r * ORA_COS(theta), m)), scal(r * ORA_SIN(theta), n));
}

/* ------------------------------------------------------------------ geometry kernel */
/* rayphoton.js:1883-1898; returns 0 for "undefined" (incl. the a==0 scalar-return quirk, which
 * makes callers' t[0] undefined and therefore a miss) */
static int q_roots(double a, double halfB, double c, double t[2]) {
  if (a == 0) return 0;
  double disc = halfB * halfB - a * c;
  if (disc < 0) return 0;
  disc = sqrt(disc);
  t[0] = (-halfB - disc) / a; t[1] = (-halfB + disc) / a;
  return 1;
}

/* rayphoton.js:490-644.  NaN return = undefined. */
double ora_intersect(const double o_[3], const double d_[3], int type, int flags, const double g[12], double EPS) {
  v3 o = ld(o_), d = ld(d_);
  switch (type) {
    case ORA_PLANE: {                                          /* :492-501 */
      v3 a = ld(g), n = ld(g + 3);
      if (fabs(dot(d, n)) < EPS) return NAN;
      double t = dot(minus(a, o), n) / dot(d, n);
      return t > EPS ? t : NAN;
    }
    case ORA_SPHERE: {                                         /* :502-519 */
      v3 c = ld(g); double r = g[6], t[2];
      double halfB = dot(d, minus(o, c));
      double cc = sqlen(minus(o, c)) - r * r;
      if (!q_roots(1, halfB, cc, t)) return NAN;
      if (t[0] > EPS) return t[0];
      if (t[1] > EPS) return t[1];
      return NAN;
    }
    case ORA_HEMISPHERE: {                                     /* :520-539 */
      v3 c = ld(g), n = ld(g + 3); double r = g[6], tmin = g[7], tmax = g[8], t[2];
      double halfB = dot(d, minus(o, c));
      double cc = sqlen(minus(o, c)) - r * r;
      if (!q_roots(1, halfB, cc, t)) return NAN;
      for (int i = 0; i < 2; i++) {
        if (t[i] > EPS) {
          v3 pos = plus(o, scal(t[i], d));
          double proj = dot(minus(pos, c), n);
          if (proj > tmin && (!(flags & ORA_FLAG_TRUNC_MAX) || proj < tmax)) return t[i];
        }
      }
      return NAN;
    }
    case ORA_CYLINDER: {                                       /* :540-567 */
      v3 c = ld(g), axis = ld(g + 3); double r = g[6], h = g[7], t[2];
      v3 v = minus(o, c);
      double vd = dot(v, d), va = dot(v, axis), da = dot(d, axis);
      double a = 1 - da * da;
      double halfB = vd - va * da;
      double cc = sqlen(v) - va * va - r * r;
      if (!q_roots(a, halfB, cc, t)) return NAN;
      if (flags & ORA_FLAG_FIRST_ROOT) return (t[0] <= 0) ? NAN : t[0];   /* ray.js:443-461: infinite, first root only */
      for (int i = 0; i < 2; i++) {
        if (t[i] > EPS) {
          if (!(flags & ORA_FLAG_HALF_HEIGHT)) return t[i];
          v3 pos = plus(o, scal(t[i], d));
          if (fabs(dot(minus(pos, c), axis)) < h - EPS) return t[i];
        }
      }
      return NAN;
    }
    case ORA_TRIANGLE: case ORA_SQUARE: {                      /* :568-609 Moller-Trumbore */
      v3 A = ld(g), AB = ld(g + 3), AC = ld(g + 6);
      v3 h = cross(d, AC);
      double a = dot(AB, h);
      if (a > -EPS && a < EPS) return NAN;
      double f = 1 / a;
      v3 s = minus(o, A);
      double u = f * dot(s, h);
      if (u < 0 || u > 1) return NAN;
      v3 q = cross(s, AB);
      double v = f * dot(d, q);
      if (type == ORA_TRIANGLE) { if (v < 0 || u + v > 1) return NAN; }
      else { if (v < 0 || v > 1) return NAN; }
      double t = f * dot(AC, q);
      return (t > EPS) ? t : NAN;
    }
    case ORA_DISC: {                                           /* :610-623 */
      v3 c = ld(g), n = ld(g + 3); double r = g[6];
      if (fabs(dot(d, n)) < EPS) return NAN;
      double t = dot(minus(c, o), n) / dot(d, n);
      if (t > EPS) {
        v3 pos = plus(o, scal(t, d));
        return (sqlen(minus(c, pos)) + EPS < r * r) ? t : NAN;
      }
      return NAN;
    }
    case ORA_ANNULUS: {                                        /* :624-638 */
      v3 c = ld(g), n = ld(g + 3); double ro = g[6], ri = g[7];
      if (fabs(dot(d, n)) < EPS) return NAN;
      double t = dot(minus(c, o), n) / dot(d, n);
      if (t > EPS) {
        v3 pos = plus(o, scal(t, d));
        double rSq = sqlen(minus(c, pos));
        return (rSq + EPS < ro * ro && rSq - EPS > ri * ri) ? t : NAN;
      }
      return NAN;
    }
  }
  return NAN;
}

void ora_intersect_batch(int n, const double *o, const double *d, const uint8_t *meta, const double *geom,
                         double eps, int normalize_dir, double *t_out) {
  for (int i = 0; i < n; i++) {
    double dn[3] = { d[3 * i], d[3 * i + 1], d[3 * i + 2] };
    if (normalize_dir) ora_normalize(dn, dn);                  /* Ray ctor, rayphoton.js:484-487 */
    t_out[i] = ora_intersect(o + 3 * i, dn, meta[4 * i], meta[4 * i + 2], geom + 12 * i, eps);
  }
}

static int shape_type(const ora_scene *s, int i) { return s->shape_meta[4 * i]; }
static int shape_mat(const ora_scene *s, int i) { return s->shape_meta[4 * i + 1]; }
static int shape_flags(const ora_scene *s, int i) { return s->shape_meta[4 * i + 2]; }
static int shape_proc(const ora_scene *s, int i) { return s->shape_meta[4 * i + 3]; }
static const double *shape_g(const ora_scene *s, int i) { return s->shape_geom + 12 * i; }

static uint64_t g_tests;   /* ray-primitive test counter (statistics only) */

/* rayphoton.js:1642-1650 (skip_lights=0) and :1467-1475 (skip_lights=1) */
static int closest_hit(const ora_scene *s, double eps, v3 o, v3 d, int skip_lights, double *t_out) {
  double min = INFINITY; int minS = -1;
  double oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
  for (uint32_t i = 0; i < s->n_shapes; i++) {
    double t = ora_intersect(oo, dd, shape_type(s, i), shape_flags(s, i), shape_g(s, i), eps);
    g_tests++;
    if (t > eps && t < min && !(skip_lights && (shape_flags(s, i) & ORA_FLAG_LIGHT))) { min = t; minS = (int)i; }
  }
  *t_out = min; return minS;
}
/* rayphoton.js:1853-1857 */
static int any_hit(const ora_scene *s, double eps, v3 o, v3 d, double source_dist) {
  double oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
  for (uint32_t i = 0; i < s->n_shapes; i++) {
    double t = ora_intersect(oo, dd, shape_type(s, i), shape_flags(s, i), shape_g(s, i), eps);
    g_tests++;
    if (t > eps && t < source_dist - eps && !(shape_flags(s, i) & ORA_FLAG_LIGHT)) return 1;
  }
  return 0;
}

void ora_scan_batch(const ora_scene *s, double eps, int mode, int n, const double *o, const double *d,
                    const double *source_dist, double *t_out, int32_t *idx_out) {
  for (int i = 0; i < n; i++) {
    v3 oo = ld(o + 3 * i), dd = ld(d + 3 * i);
    if (mode == 2) { idx_out[i] = any_hit(s, eps, oo, dd, source_dist[i]); if (t_out) t_out[i] = 0; }
    else { double t; idx_out[i] = closest_hit(s, eps, oo, dd, mode == 1, &t); t_out[i] = t; }
  }
}

/* shape.normal(p): rayphoton.js:368-371, :385, :403, :417, :433, :449, :464, :480 */
static v3 shape_normal(const ora_scene *s, int i, v3 p) {
  const double *g = shape_g(s, i);
  int convex = shape_flags(s, i) & ORA_FLAG_CONVEX;
  switch (shape_type(s, i)) {
    case ORA_PLANE: case ORA_DISC: case ORA_ANNULUS: return ld(g + 3);
    case ORA_TRIANGLE: case ORA_SQUARE: return ld(g + 9);
    case ORA_SPHERE: return scal(1 / g[6], minus(p, ld(g)));
    case ORA_HEMISPHERE: return scal((convex ? 1 : -1) / g[6], minus(p, ld(g)));
    case ORA_CYLINDER: {
      v3 v = minus(p, ld(g)), axis = ld(g + 3);
      return scal((convex ? 1 : -1) / g[6], minus(v, scal(dot(v, axis), axis)));
    }
  }
  return V(0, 0, 1);
}

/* JavaScript ToInt32 for the `& 1` in the checker patterns */
static int32_t to_int32(double x) {
  if (!isfinite(x)) return 0;
  double m = fmod(trunc(x), 4294967296.0);
  if (m < 0) m += 4294967296.0;
  return (int32_t)(uint32_t)m;
}

/* shape.colour(p): base colour (rayphoton.js:350) or one of the preset closures */
static void shape_colour(const ora_scene *s, int i, v3 p, double out[3]) {
  static const double PINK[3] = { 255, 32, 144 }, GREY[3] = { 128, 128, 128 }, BLACK[3] = { 0, 0, 0 },
                      BLUE[3] = { 8, 8, 64 }, WHITE[3] = { 255, 255, 255 };
  const double *c = s->shape_colour + 4 * i;
  const double *g = shape_g(s, i);
  switch (shape_proc(s, i)) {
    case 1: {                                                  /* rayphoton.js:746-755 */
      double f = ORA_SIN(p.x) + p.y;
      if (f > 0 && f < 1) c = PINK;
      else { int idx = 1 + ((to_int32(floor(p.x / 4) + floor(p.y / 4))) & 1); c = (idx == 1) ? GREY : BLACK; }
      (void)BLUE;
      break;
    }
    case 2: {                                                  /* rayphoton.js:756-768 ("8-ball") */
      v3 pc = minus(p, ld(g)); double r = g[6];
      double c1 = dot(pc, normalize(V(-1, -1, 0), 1e-6)) / r;
      if (c1 < 0.905) { c = BLACK; break; }
      c1 = dot(pc, normalize(V(-1, -1, 0.15), 1e-6)) / r;
      double c2 = dot(pc, normalize(V(-1, -1, -0.12), 1e-6)) / r;
      c = ((c1 > 0.9920 && c1 < 0.999) || (c2 > 0.9900 && c2 < 0.9980)) ? BLACK : WHITE;
      break;
    }
    case 3: {                                                  /* rayphoton.js:784-792, :820-829 */
      double x = fabs(fmod(p.x + 100.7, 2.3) - 1.15), y = fabs(fmod(p.y + 102.7, 2.3) - 1.15);
      c = (x < 0.03 || y < 0.03 || x + y < 0.6) ? BLACK : WHITE;
      break;
    }
    case 4: {                                                  /* rayphoton.js:871-874, :889-892 */
      int idx = to_int32(floor((0.6 * p.x + 0.8 * p.y + 0.7) / 3.2) + floor((0.8 * p.x - 0.6 * p.y + 0.2) / 3.2)) & 1;
      c = idx ? BLACK : WHITE;
      break;
    }
    case 5: {                                                  /* ray.js:650-659 */
      double f = ORA_SIN(p.x) + p.y;
      if (f > 0 && f < 1) c = PINK;
      else {
        int idx = 1 + (to_int32(floor(p.x * 2)) & 1) + (to_int32(floor(p.y * 2)) & 1);
        c = (idx == 1) ? GREY : (idx == 2) ? BLACK : BLUE;
      }
      break;
    }
    default: break;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
}

/* ------------------------------------------------------------------ photons */
typedef struct photon_t { v3 o, d; double power; double col[4]; int caustic; /* -1 undefined, 0, 1 */ } photon_t;

static int light_of_shape(const ora_scene *s, int shape) {
  for (uint32_t l = 0; l < s->n_lights; l++) if ((int)s->light_shape[l] == shape) return (int)l;
  return -1;
}

/* rayphoton.js:1326-1344.  Always 5 draws.  Returns the chosen light (clamped if the cumulative
 * walk runs off the end, where the reference would throw a TypeError). */
static int create_photon(const ora_scene *s, double eps, rng_t *g, photon_t *ph) {
  double r = rng_next(g);
  uint32_t l = 0;
  while (l < s->n_lights && r > s->light_data[8 * l + 2]) { r -= s->light_data[8 * l + 2]; l++; }
  if (l >= s->n_lights) l = s->n_lights - 1;
  const double *L = s->light_data + 8 * l;
  const double *lg = shape_g(s, (int)s->light_shape[l]);
  v3 centre = ld(lg), nrm = ld(lg + 3); double radius = lg[6];
  v3 origin = disc_sample(centre, radius, nrm, eps, g);
  v3 dir = vec_perturb(nrm, eps, g);
  ph->o = origin; ph->d = normalize(dir, eps);                 /* Photon extends Ray: ctor normalises */
  ph->power = L[0];
  ph->col[0] = L[3]; ph->col[1] = L[4]; ph->col[2] = L[5]; ph->col[3] = L[6];
  ph->caustic = -1;
  return (int)l;
}

int ora_create_photon_keyed(const ora_scene *s, uint64_t seed, uint32_t domain, uint32_t a, uint32_t b, double out8[8]) {
  rng_t g; rng_set(&g, seed, domain, a, b); g.total = 0;
  photon_t ph; int l = create_photon(s, 1e-6, &g, &ph);
  out8[0] = ph.o.x; out8[1] = ph.o.y; out8[2] = ph.o.z; out8[3] = ph.d.x; out8[4] = ph.d.y; out8[5] = ph.d.z;
  out8[6] = ph.power; out8[7] = g.draw;
  return l;
}
void ora_vec_perturb_keyed(uint64_t seed, uint32_t domain, uint32_t a, uint32_t b, uint32_t first_draw,
                           const double v[3], double out[3]) {
  rng_t g; rng_set(&g, seed, domain, a, b); g.draw = first_draw;
  v3 r = vec_perturb(ld(v), 1e-6, &g); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ora_disc_sample_keyed(uint64_t seed, uint32_t domain, uint32_t a, uint32_t b, uint32_t first_draw,
                           const double c[3], double radius, const double n[3], double out[3]) {
  rng_t g; rng_set(&g, seed, domain, a, b); g.draw = first_draw;
  v3 r = disc_sample(ld(c), radius, ld(n), 1e-6, &g); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

#define STACK_MAX 64
typedef struct { uint8_t m[STACK_MAX]; int len; } mstack;

static double mat_refr(const ora_scene *s, int mat) { return (mat >= 0 && mat < 8) ? s->materials[1][mat] : NAN; }
static int mat_transparent(const ora_scene *s, int mat) { return (mat >= 0 && mat < 8) ? (s->materials[0][mat] == 1.0) : 0; }
static double mat_reflectance(const ora_scene *s, int mat) { return (mat >= 0 && mat < 8) ? s->materials[2][mat] : NAN; }
static double mat_specular(const ora_scene *s, int mat) { return (mat >= 0 && mat < 8) ? s->materials[3][mat] : NAN; }

/* result of one emission attempt */
typedef struct { int stored; photon_t rec; int error; uint64_t segments; } emit_result;

/* rayphoton.js:1461-1557, recursion unrolled into a loop (every branch makes at most one
 * recursive call, in tail position).  The material stack is one mutable array per attempt:
 * pop mutates in place (:1514), push copies (:1516) -- indistinguishable for a linear path. */
static void emit_photon(const ora_scene *s, const ora_params *P, photon_t ph, int track_only_caustic,
                        rng_t *g, emit_result *res) {
  const double EPS = P->epsilon;
  mstack st; st.len = 1; st.m[0] = 0;                         /* [MAT_AIR] */
  int depth = 0;
  res->stored = 0; res->error = 0; res->segments = 0;
  for (;;) {
    double t; res->segments++;
    int minS = closest_hit(s, EPS, ph.o, ph.d, 1, &t);
    if (minS < 0) return;                                      /* lost */
    v3 hit = plus(ph.o, scal(t, ph.d));
    double local[3]; shape_colour(s, minS, hit, local);
    v3 n = shape_normal(s, minS, hit);
    double cos1 = -dot(ph.d, n);
    int mat = shape_mat(s, minS);
    if (mat_transparent(s, mat) && depth < P->max_depth) {     /* :1484 */
      double eta1 = mat_refr(s, st.m[st.len - 1]);
      double eta2 = (dot(ph.d, n) < 0) ? mat_refr(s, mat) : (st.len >= 2 ? mat_refr(s, st.m[st.len - 2]) : NAN);
      if (isnan(eta2)) { res->error = 1; return; }             /* :1487-1490 throws */
      double ratio = eta1 / eta2;
      double c2sq = 1 - ratio * ratio * (1 - cos1 * cos1);
      int caustic = (ph.caustic < 0) ? 1 : ph.caustic;
      v3 ndir;
      if (c2sq < 0) {                                          /* total internal reflection :1493-1496 */
        ndir = plus(ph.d, scal(2 * cos1, n));
      } else {
        double cos2 = sqrt(c2sq);
        double a1 = (eta1 * cos1 - eta2 * cos2) / (eta1 * cos1 + eta2 * cos2);
        double a2 = (eta1 * cos2 - eta2 * cos1) / (eta1 * cos2 + eta2 * cos1);
        double rS = a1 * a1, rP = a2 * a2;
        if (cos1 < 0) { rS = 1 / rS; rP = 1 / rP; }
        if (rng_next(g) < 0.5 * (rS + rP)) {
          ndir = plus(ph.d, scal(2 * cos1, n));
        } else {
          double pm = (cos1 < 0) ? -1 : 1;
          ndir = plus(scal(ratio, ph.d), scal(ratio * cos1 - pm * cos2, n));
          if (cos1 < 0) { st.len--; } else if (st.len < STACK_MAX) { st.m[st.len++] = (uint8_t)mat; }
        }
      }
      ph.o = hit; ph.d = normalize(ndir, EPS); ph.caustic = caustic; depth++;
      continue;
    }
    /* opaque (or transparent at depth >= MAX_DEPTH, where reflectance is undefined) :1522 */
    if (rng_next(g) < mat_reflectance(s, mat)) {
      if (depth < P->max_depth) {
        if (rng_next(g) < mat_specular(s, mat)) {              /* :1525-1533 */
          v3 ndir = plus(ph.d, scal(2 * cos1, n));
          int caustic = (ph.caustic < 0) ? 1 : ph.caustic;
          ph.o = hit; ph.d = normalize(ndir, EPS); ph.caustic = caustic; depth++;
          continue;
        } else if (!track_only_caustic) {                      /* :1536-1546 */
          v3 ndir = vec_perturb(n, EPS, g);
          double tc[3] = { ph.col[0] / 255 * local[0], ph.col[1] / 255 * local[1], ph.col[2] / 255 * local[2] };
          double adj = ph.power * (255 + 255 + 255 + 1) / (local[0] + local[1] + local[2] + 1);
          ph.o = hit; ph.d = normalize(ndir, EPS); ph.power = adj;
          ph.col[0] = tc[0]; ph.col[1] = tc[1]; ph.col[2] = tc[2]; ph.col[3] = 1;
          ph.caustic = 0; depth++;
          continue;
        }
      }
      return;                                                  /* dropped */
    }
    if (depth > 0) {                                           /* absorb; store :1549-1553 */
      int is_c = (ph.caustic == 1);
      ph.power /= is_c ? (double)P->num_photons_caustic : (double)P->num_photons_diffuse;
      res->stored = 1;
      res->rec = ph;
      res->rec.o = hit;
      res->rec.d = normalize(scal(-1, ph.d), EPS);
      res->rec.caustic = is_c;
    }
    return;
  }
}

static void write_rec(double *dst, const photon_t *p) {
  dst[0] = p->o.x; dst[1] = p->o.y; dst[2] = p->o.z; dst[3] = p->d.x; dst[4] = p->d.y; dst[5] = p->d.z;
  dst[6] = p->power; dst[7] = p->col[0]; dst[8] = p->col[1]; dst[9] = p->col[2]; dst[10] = p->col[3];
}

/* rayphoton.js:1354-1383 with one RNG stream per attempt (domain = phase, a/b = attempt) */
int ora_emit_many(const ora_scene *s, const ora_params *P,
                  double *diffuse, int64_t *dtag, uint32_t *n_diffuse,
                  double *caustic, int64_t *ctag, uint32_t *n_caustic,
                  uint64_t *attempts1, uint64_t *attempts2, uint64_t *draws, uint64_t *segments) {
  uint32_t nd = 0, nc = 0; int capped = 0;
  const uint32_t ND = P->num_photons_diffuse, NC = P->num_photons_caustic;
  rng_t g; g.total = 0; uint64_t segs = 0;
  *attempts1 = *attempts2 = 0;
  if (s->n_lights == 0) { *n_diffuse = *n_caustic = 0; return 0; }
  for (int phase = 1; phase <= 2; phase++) {
    uint64_t attempt = 0;
    uint64_t cap = 100ull * (phase == 1 ? ND : NC);
    while (phase == 1 ? (nd < ND) : (nc < NC)) {
      rng_set(&g, P->seed, phase == 1 ? ORA_DOMAIN_EMIT1 : ORA_DOMAIN_EMIT2, (uint32_t)attempt, (uint32_t)(attempt >> 32));
      photon_t ph; create_photon(s, P->epsilon, &g, &ph);
      emit_result r; emit_photon(s, P, ph, phase == 2, &g, &r);
      segs += r.segments;
      if (r.stored) {                                          /* storePhoton :1346-1352 */
        if (r.rec.caustic) {
          if (nc < NC) { write_rec(caustic + ORA_PHOTON_STRIDE * nc, &r.rec); ctag[3 * nc] = 1; ctag[3 * nc + 1] = phase; ctag[3 * nc + 2] = (int64_t)attempt; nc++; }
        } else {
          if (nd < ND) { write_rec(diffuse + ORA_PHOTON_STRIDE * nd, &r.rec); dtag[3 * nd] = 0; dtag[3 * nd + 1] = phase; dtag[3 * nd + 2] = (int64_t)attempt; nd++; }
        }
      }
      attempt++;
      if (attempt > cap) { capped = 1; break; }
    }
    if (phase == 1) *attempts1 = attempt; else *attempts2 = attempt;
    if (capped) break;                                         /* the reference returns from emitManyPhotons */
  }
  *n_diffuse = nd; *n_caustic = nc; *draws = g.total; *segments = segs;
  return capped;
}

void ora_emit_attempts(const ora_scene *s, const ora_params *P, int phase, uint64_t a0, uint64_t a1,
                       int32_t *flag, double *rec) {
  rng_t g; g.total = 0;
  for (uint64_t a = a0; a < a1; a++) {
    rng_set(&g, P->seed, phase == 1 ? ORA_DOMAIN_EMIT1 : ORA_DOMAIN_EMIT2, (uint32_t)a, (uint32_t)(a >> 32));
    photon_t ph; create_photon(s, P->epsilon, &g, &ph);
    emit_result r; emit_photon(s, P, ph, phase == 2, &g, &r);
    flag[a - a0] = r.stored ? (r.rec.caustic ? 2 : 1) : 0;
    if (r.stored) write_rec(rec + ORA_PHOTON_STRIDE * (a - a0), &r.rec);
    else memset(rec + ORA_PHOTON_STRIDE * (a - a0), 0, sizeof(double) * ORA_PHOTON_STRIDE);
  }
}

/* ------------------------------------------------------------------ kd-tree (kdtree.js) */
uint32_t ora_find_median(uint32_t n) {                         /* kdtree.js:109-124 */
  if (n <= 1) return 0;
  uint32_t powerTwo = 1, twoBits = n;
  for (;;) {
    if (twoBits == 2) return n - powerTwo;
    if (twoBits == 3) return 2 * powerTwo - 1;
    powerTwo <<= 1; twoBits >>= 1;
  }
}

/* stable merge sort of idx[0..n) by pos[idx*3+dim]; equal keys keep their order, as the
 * stable Array.prototype.sort with comparator x-y does (kdtree.js:102) */
static void merge_sort(int32_t *idx, int32_t *tmp, uint32_t n, const double *pos, int dim) {
  if (n < 2) return;
  uint32_t h = n / 2;
  merge_sort(idx, tmp, h, pos, dim); merge_sort(idx + h, tmp, n - h, pos, dim);
  uint32_t i = 0, j = h, k = 0;
  while (i < h && j < n) {
    if (pos[3 * (size_t)idx[j] + dim] - pos[3 * (size_t)idx[i] + dim] < 0) tmp[k++] = idx[j++];
    else tmp[k++] = idx[i++];
  }
  while (i < h) tmp[k++] = idx[i++];
  while (j < n) tmp[k++] = idx[j++];
  memcpy(idx, tmp, sizeof(int32_t) * n);
}

static void make_tree(int32_t *data, int32_t *tmp, uint32_t len, uint32_t posn, const double *pos,
                      int32_t *perm, uint8_t *dimv) {         /* kdtree.js:82-107 */
  double mx[3] = { -INFINITY, -INFINITY, -INFINITY }, mn[3] = { INFINITY, INFINITY, INFINITY };
  for (int d = 0; d < 3; d++)
    for (uint32_t i = 0; i < len; i++) {
      double v = pos[3 * (size_t)data[i] + d];
      if (v > mx[d]) mx[d] = v;
      if (v < mn[d]) mn[d] = v;
    }
  int widest = 0; double maxw = mx[0] - mn[0];
  for (int d = 1; d < 3; d++) if (mx[d] - mn[d] > maxw) { widest = d; maxw = mx[d] - mn[d]; }
  merge_sort(data, tmp, len, pos, widest);
  uint32_t m = ora_find_median(len);
  perm[posn - 1] = data[m]; dimv[posn - 1] = (uint8_t)widest;
  if (len > 1) make_tree(data, tmp, m, 2 * posn, pos, perm, dimv);
  if (m < len - 1) make_tree(data + m + 1, tmp, len - m - 1, 2 * posn + 1, pos, perm, dimv);
}

int ora_kd_build(const double *pos, uint32_t n, int32_t *perm, uint8_t *dim) {
  if (n == 0) return 1;                                        /* kdtree.js:6 throws */
  int32_t *data = (int32_t *)malloc(sizeof(int32_t) * n), *tmp = (int32_t *)malloc(sizeof(int32_t) * n);
  for (uint32_t i = 0; i < n; i++) data[i] = (int32_t)i;
  make_tree(data, tmp, n, 1, pos, perm, dim);
  free(data); free(tmp);
  return 0;
}

typedef struct { const double *pos; uint32_t n; const int32_t *perm; const uint8_t *dim;
                 const double *x; double r; int32_t *out; uint32_t cap, count; } rq_t;

static void radius_rec(rq_t *q, uint32_t posn) {               /* kdtree.js:64-80 */
  if (posn >= q->n + 1) return;
  const double *p = q->pos + 3 * (size_t)q->perm[posn - 1];
  int d = q->dim[posn - 1];
  if (q->x[d] + q->r < p[d]) { radius_rec(q, 2 * posn); return; }
  if (q->x[d] - q->r > p[d]) { radius_rec(q, 2 * posn + 1); return; }
  v3 df = minus(ld(q->x), ld(p));
  if (sqlen(df) < q->r * q->r) { if (q->count < q->cap) q->out[q->count] = q->perm[posn - 1]; q->count++; }
  radius_rec(q, 2 * posn); radius_rec(q, 2 * posn + 1);
}
uint32_t ora_kd_radius(const double *pos, uint32_t n, const int32_t *perm, const uint8_t *dim,
                       const double x[3], double r, int32_t *out, uint32_t cap) {
  rq_t q = { pos, n, perm, dim, x, r, out, cap, 0 };
  radius_rec(&q, 1);
  return q.count;
}

/* nearestNeighbourAndDistance(x, pos) (kdtree.js:10-62): the closest of the nodes on ONE root-to-leaf path (the far side
 * of a splitting plane is never searched: `if (false)`, :34, :50).  A node past the end gives [undefined, undefined]
 * (:13), and `dist < undefined` is false (:30, :46), so a path that leaves the tree through a missing upper child
 * returns undefined all the way up.  Returns 0 and (*idx, *dist), or 1 for [undefined, undefined]. */
static int nearest_rec(const double *pos, uint32_t n, const int32_t *perm, const uint8_t *dim, const double x[3], uint32_t posn,
                       int32_t *idx, double *dist) {
  if (posn > n) return 1;                                              /* pos >= nodes.length */
  const double *pt = pos + 3 * (size_t)perm[posn - 1];
  int dm = dim[posn - 1];
  double d = sqrt(sqlen(minus(ld(x), ld(pt))));                        /* vecLength(vecMinus(x, point)) */
  if (2 * (uint64_t)posn > n) { *idx = perm[posn - 1]; *dist = d; return 0; }     /* pos * 2 >= nodes.length: leaf */
  int32_t bi; double bd;
  int undef = nearest_rec(pos, n, perm, dim, x, (x[dm] < pt[dm]) ? 2 * posn : 2 * posn + 1, &bi, &bd);
  if (undef) return 1;                                                 /* dist < undefined is false */
  if (d < bd) { bi = perm[posn - 1]; bd = d; }
  *idx = bi; *dist = bd;
  return 0;
}
int ora_kd_nearest(const double *pos, uint32_t n, const int32_t *perm, const uint8_t *dim, const double x[3], int32_t *idx, double *dist) {
  int undef = nearest_rec(pos, n, perm, dim, x, 1, idx, dist);
  if (undef) { *idx = -1; *dist = NAN; }
  return undef;
}

/* k nearest by brute force, ordered by (d2, index).  The reference has no k-NN query (its only
 * NN method, kdtree.js:10-62 (ora_kd_nearest above), is approximate and never called): this ordering is the definition
 * the device k-NN is held to.  d2 = dx*dx + dy*dy + dz*dz as vecSqLength(vecMinus(x,p)). */
void ora_knn_brute(const double *pos, uint32_t n, const double x[3], uint32_t k, int32_t *idx, double *d2) {
  uint32_t m = 0;
  for (uint32_t j = 0; j < n; j++) {
    v3 df = minus(ld(x), ld(pos + 3 * (size_t)j));
    double dd = sqlen(df);
    if (m == k && !(dd < d2[k - 1])) continue;                 /* ties keep the lower index */
    uint32_t i = (m < k) ? m : k - 1;
    while (i > 0 && d2[i - 1] > dd) { d2[i] = d2[i - 1]; idx[i] = idx[i - 1]; i--; }
    d2[i] = dd; idx[i] = (int32_t)j;
    if (m < k) m++;
  }
  for (uint32_t i = m; i < k; i++) { idx[i] = -1; d2[i] = INFINITY; }
}

/* ------------------------------------------------------------------ renderer */
struct ora_maps {
  uint32_t n[2]; double *rec[2]; double *pos[2]; int32_t *perm[2]; uint8_t *dim[2];
};
ora_maps *ora_maps_create(const double *diffuse, uint32_t nd, const double *caustic, uint32_t nc) {
  ora_maps *m = (ora_maps *)calloc(1, sizeof(ora_maps));
  const double *src[2] = { diffuse, caustic }; uint32_t cnt[2] = { nd, nc };
  for (int k = 0; k < 2; k++) {
    m->n[k] = cnt[k];
    if (!cnt[k]) continue;
    m->rec[k] = (double *)malloc(sizeof(double) * ORA_PHOTON_STRIDE * cnt[k]);
    memcpy(m->rec[k], src[k], sizeof(double) * ORA_PHOTON_STRIDE * cnt[k]);
    m->pos[k] = (double *)malloc(sizeof(double) * 3 * cnt[k]);
    for (uint32_t i = 0; i < cnt[k]; i++) memcpy(m->pos[k] + 3 * i, src[k] + ORA_PHOTON_STRIDE * i, sizeof(double) * 3);
    m->perm[k] = (int32_t *)malloc(sizeof(int32_t) * cnt[k]);
    m->dim[k] = (uint8_t *)malloc(cnt[k]);
    ora_kd_build(m->pos[k], cnt[k], m->perm[k], m->dim[k]);
  }
  return m;
}
void ora_maps_destroy(ora_maps *m) {
  if (!m) return;
  for (int k = 0; k < 2; k++) { free(m->rec[k]); free(m->pos[k]); free(m->perm[k]); free(m->dim[k]); }
  free(m);
}

typedef struct { double r, g, b, a; } col_t;
typedef struct { const ora_scene *s; const ora_params *P; const ora_maps *m; rng_t *g; uint64_t rays; } rctx;

/* rayphoton.js:1841-1880 */
static void direct_light(rctx *c, v3 point, v3 normal, const double local[3], double out[3]) {
  const ora_scene *s = c->s; const double EPS = c->P->epsilon;
  double col[3] = { 0, 0, 0 };
  for (uint32_t l = 0; l < s->n_lights; l++) {
    photon_t ph; create_photon(s, EPS, c->g, &ph);             /* random light, NOT light l (:1847) */
    double power = ph.power * s->light_data[8 * l + 1];        /* ... scaled by light l's area (:1849) */
    v3 dir = normalize(minus(ph.o, point), EPS);
    double source_dist = sqrt(sqlen(minus(point, ph.o)));
    if (!any_hit(s, EPS, point, dir, source_dist)) {
      double dn = dot(dir, normal);
      for (int k = 0; k < 3; k++)
        col[k] += (ph.col[k] / 255) * local[k] * dn * power / c->P->standard_lamp_area;   /* :1871 */
    }
  }
  for (int k = 0; k < 3; k++) out[k] = floor(col[k] / 1);      /* :1876-1878 */
}

/* directLight at one point on a keyed stream (for tests of the product's rt_direct_light); returns the stream position */
uint32_t ora_direct_light_keyed(const ora_scene *s, const ora_params *P, uint32_t domain, uint32_t a, uint32_t b, uint32_t first_draw,
                                const double point[3], const double normal[3], const double local[3], double out[3]) {
  rng_t g; rng_set(&g, P->seed, domain, a, b); g.draw = first_draw;
  rctx c = { s, P, NULL, &g, 0 };
  direct_light(&c, ld(point), ld(normal), local, out);
  return g.draw;
}

typedef struct { const double *rec; v3 normal; double sum[3]; } gather_t;
static void gather_rec(const ora_maps *m, int k, uint32_t posn, const double x[3], double r, gather_t *G) {
  if (posn >= m->n[k] + 1) return;                             /* kdtree.js:64-80, visiting instead of listing */
  int32_t pi = m->perm[k][posn - 1];
  const double *p = m->pos[k] + 3 * (size_t)pi;
  int d = m->dim[k][posn - 1];
  if (x[d] + r < p[d]) { gather_rec(m, k, 2 * posn, x, r, G); return; }
  if (x[d] - r > p[d]) { gather_rec(m, k, 2 * posn + 1, x, r, G); return; }
  if (sqlen(minus(ld(x), ld(p))) < r * r) {
    const double *R = m->rec[k] + ORA_PHOTON_STRIDE * (size_t)pi;
    double w = fmax(0, dot(ld(R + 3), G->normal));             /* rayphoton.js:1781, :1804 */
    for (int ch = 0; ch < 3; ch++) G->sum[ch] += R[7 + ch] * R[6] * w;
  }
  gather_rec(m, k, 2 * posn, x, r, G); gather_rec(m, k, 2 * posn + 1, x, r, G);
}

/* rayphoton.js:1626-1839 */
static col_t trace_ray(rctx *c, v3 o, v3 d, double max_dist, int depth, double importance, mstack *st) {
  const ora_scene *s = c->s; const ora_params *P = c->P; const double EPS = P->epsilon;
  col_t BLACK = { 0, 0, 0, 0 };
  if (importance < 0.01) return BLACK;
  c->rays++;
  double t; int minS = closest_hit(s, EPS, o, d, 0, &t);
  if (minS < 0) { col_t dark = { 16, 16, 16, 1 }; return dark; }
  if (shape_flags(s, minS) & ORA_FLAG_LIGHT) {
    const double *L = s->light_data + 8 * light_of_shape(s, minS);
    col_t lc = { L[3], L[4], L[5], L[6] }; return lc;
  }
  v3 hit = plus(o, scal(t, d));
  double local[3]; shape_colour(s, minS, hit, local);
  v3 n = shape_normal(s, minS, hit);
  col_t out = { 0, 0, 0, 1 };
  if (t < max_dist || depth < P->max_depth) {
    double cos1 = -dot(d, n);
    int mat = shape_mat(s, minS);
    if (mat_transparent(s, mat)) {
      double eta1 = mat_refr(s, st->m[st->len - 1]);
      double eta2 = (dot(d, n) < 0) ? mat_refr(s, mat) : (st->len >= 2 ? mat_refr(s, st->m[st->len - 2]) : NAN);
      if (isnan(eta2)) return BLACK;                           /* :1682-1700 with warnings=false */
      double ratio = eta1 / eta2;
      double c2sq = 1 - ratio * ratio * (1 - cos1 * cos1);
      v3 ndir;
      if (c2sq < 0) {
        ndir = plus(d, scal(2 * cos1, n));
      } else {
        double cos2 = sqrt(c2sq);
        double a1 = (eta1 * cos1 - eta2 * cos2) / (eta1 * cos1 + eta2 * cos2);
        double a2 = (eta1 * cos2 - eta2 * cos1) / (eta1 * cos2 + eta2 * cos1);
        double rS = a1 * a1, rP = a2 * a2;
        if (cos1 < 0) { rS = 1 / rS; rP = 1 / rP; }
        if (rng_next(c->g) < 0.5 * (rS + rP)) {
          ndir = plus(d, scal(2 * cos1, n));
        } else {
          double pm = (cos1 < 0) ? -1 : 1;
          ndir = plus(scal(ratio, d), scal(ratio * cos1 - pm * cos2, n));
          if (cos1 < 0) { st->len--; } else if (st->len < STACK_MAX) { st->m[st->len++] = (uint8_t)mat; }
        }
      }
      return trace_ray(c, hit, normalize(ndir, EPS), max_dist - t, depth + 1, importance, st);   /* :1734 */
    }
    if (cos1 < 0) return out;                                  /* :1736-1747 -> {0,0,0,1} */
    double refl = mat_reflectance(s, mat), spec = mat_specular(s, mat);
    double direct[3] = { 0, 0, 0 };
    if (refl < 1 && P->use_direct_light) direct_light(c, plus(hit, scal(EPS, n)), n, local, direct);
    col_t trans = { 0, 0, 0, 1 };
    if (rng_next(c->g) < refl && rng_next(c->g) < spec) {      /* :1753 (short-circuit) */
      v3 ndir = plus(d, scal(2 * cos1, n));
      trans = trace_ray(c, hit, normalize(ndir, EPS), max_dist - t, depth + 1, importance * refl, st);
    }
    double diff[3] = { 0, 0, 0 }, caus[3] = { 0, 0, 0 };
    if (P->use_photon_maps && c->m) {
      double x[3] = { hit.x, hit.y, hit.z };
      gather_t G; G.normal = n;
      if (c->m->n[0]) { G.sum[0] = G.sum[1] = G.sum[2] = 0; gather_rec(c->m, 0, 1, x, P->diffuse_radius, &G); memcpy(diff, G.sum, sizeof diff); }
      if (c->m->n[1]) { G.sum[0] = G.sum[1] = G.sum[2] = 0; gather_rec(c->m, 1, 1, x, P->caustic_radius, &G); memcpy(caus, G.sum, sizeof caus); }
    }
    const double dArea = PI * (P->diffuse_radius * P->diffuse_radius), cArea = PI * (P->caustic_radius * P->caustic_radius);
    double tr[3] = { trans.r, trans.g, trans.b }, res[3];
    for (int k = 0; k < 3; k++) {                              /* :1808-1822 */
      diff[k] *= local[k] / 255; diff[k] /= dArea;
      caus[k] *= local[k] / 255; caus[k] /= cArea;
      res[k] = (1 - refl) * direct[k];
      res[k] += tr[k]; res[k] += diff[k]; res[k] += caus[k];
    }
    out.r = res[0]; out.g = res[1]; out.b = res[2];
    return out;
  }
  return out;                                                  /* :1828-1833: clamp(0+0) */
}

static double clampd(double x, double lo, double hi) { return (x < lo) ? lo : ((x > hi) ? hi : x); }

void ora_render(const ora_scene *s, const ora_params *P, const ora_maps *m,
                int x0, int y0, int w, int h, double *pre, double *post, uint64_t *n_rays, uint64_t *n_tests) {
  rng_t g; g.total = 0;
  rctx c = { s, P, m, &g, 0 };
  g_tests = 0;
  const double *cam = s->camera;
  v3 co = ld(cam), cu = ld(cam + 3), cv = ld(cam + 6), cw = ld(cam + 9);
  const int S = P->sub_sample;
  const int B = P->super_sample_block > 1 ? P->super_sample_block : 1;      /* block side: SUPER_SAMPLE_BASE ** scale */
  const int wb = (w + B - 1) / B, hb = (h + B - 1) / B;
  w = wb;                                                                   /* (outputs are indexed by block) */
  for (int yy = 0; yy < hb; yy++)
    for (int xx = 0; xx < wb; xx++) {
      int X = x0 + xx * B, Y = y0 + yy * B;
      uint32_t pixel = (uint32_t)(Y * (int)s->width + X);
      double tot[4] = { 0, 0, 0, 0 }; col_t last = { 0, 0, 0, 0 };
      for (int sy = 0; sy < S; sy++)
        for (int sx = 0; sx < S; sx++) {                       /* rayphoton.js:1589-1616 */
          rng_set(&g, P->seed, ORA_DOMAIN_RENDER, pixel, (uint32_t)(sy * S + sx));
          double ssx = X, ssy = Y;
          if (B > 1) { ssx += B * rng_next(&g); ssy += B * rng_next(&g); }    /* :1593-1596 */
          double u = ((ssx + (sx + rng_next(&g)) / S) * 2 / s->width) - 1;
          double v = -(((ssy + (sy + rng_next(&g)) / S) * 2 / s->height) - 1);
          u *= cam[12]; v *= cam[13];
          v3 xyz = plus(plus(plus(co, scal(u, cu)), scal(v, cv)), scal(-1, cw));   /* Camera.toXYZ :712 */
          mstack st; st.len = 1; st.m[0] = 0;
          last = trace_ray(&c, co, normalize(minus(xyz, co), P->epsilon), P->max_trace_dist, 0, 1, &st);
          tot[0] += clampd(last.r, 0, 255); tot[1] += clampd(last.g, 0, 255);
          tot[2] += clampd(last.b, 0, 255); tot[3] += clampd(last.a, 0, 1);
        }
      size_t o4 = 4 * ((size_t)yy * w + xx);
      if (pre) { pre[o4] = last.r; pre[o4 + 1] = last.g; pre[o4 + 2] = last.b; pre[o4 + 3] = last.a; }
      if (post) for (int k = 0; k < 4; k++) post[o4 + k] = floor(tot[k] / (S * S));   /* :1618-1621 */
    }
  if (n_rays) *n_rays = c.rays;
  if (n_tests) *n_tests = g_tests;
}

/* ------------------------------------------------------------------ mode A: ray.js */
/* vecPerturb of ray.js:75-100: basis NOT normalised; lower bound cos(pi/2)^2 */
static v3 vec_perturb_a(v3 v, double eps, rng_t *g) {
  v3 m = V(1, 0, 0);
  if (is_zero(cross(v, m), eps)) m = V(0, 1, 0);
  v3 n = cross(v, m);
  m = cross(n, v);
  double cl = ORA_COS(PI / 2);
  double lower = cl * cl;
  double x = (1 - lower) * rng_next(g) + lower;
  double cosT = sqrt(x);
  double sinT = sqrt(1 - cosT * cosT);
  double phi = 2 * PI * rng_next(g);
  return plus(plus(scal(cosT, v), scal(sinT * ORA_COS(phi), m)), scal(sinT * ORA_SIN(phi), n));
}

typedef struct { int ok; double r, g, b, a; } cola_t;      /* ok = 0: undefined */

/* ray.js:844-993.  Scan is in DESCENDING index order, so equal distances go to the highest index. */
static cola_t trace_ray_a(rctx *c, v3 o, v3 d, double max_dist, int depth, double importance, mstack *st) {
  const ora_scene *s = c->s; const double EPS = c->P->epsilon;
  cola_t out = { 1, 0, 0, 0, 1 };
  if (importance < 0.01) { out.r = out.g = out.b = 255; return out; }          /* COL_WHITE :845-847 */
  c->rays++;
  double min = INFINITY; int minS = -1;
  double oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
  for (int i = (int)s->n_shapes; i--;) {
    double t = ora_intersect(oo, dd, shape_type(s, i), shape_flags(s, i), shape_g(s, i), EPS);
    if (t > EPS && t < min) { min = t; minS = i; }
  }
  if (minS < 0) { out.r = out.g = out.b = 16; return out; }                     /* :992 */
  v3 hit = plus(o, scal(min, d));
  double sc[3]; shape_colour(s, minS, hit, sc);
  v3 n = shape_normal(s, minS, hit);
  const double refl = s->shape_surface ? s->shape_surface[2 * minS] : NAN;
  const double spec = s->shape_surface ? s->shape_surface[2 * minS + 1] : NAN;
  double T[3] = { sc[0], sc[1], sc[2] };                                        /* transmittedColour = shapeCol :891 */
  if (min < max_dist && depth > 0) {                                            /* :893 */
    double cos1 = -dot(d, n);
    int mat = shape_mat(s, minS);
    cola_t child;
    if (!mat_transparent(s, mat)) {                                             /* MAT_OPAQUE :895 */
      if (cos1 < 0) { if (min < 1e5) { out.ok = 0; return out; } out.r = out.g = out.b = 0; out.a = 0; return out; }   /* :905 */
      v3 rdir = plus(d, scal(2 * cos1, n));
      v3 pdir = rdir;
      if (rng_next(c->g) > spec) {                                              /* :910-923 */
        for (int attempt = 0; attempt < 100; attempt++) {
          pdir = vec_perturb_a(rdir, EPS, c->g);
          if (dot(pdir, n) > EPS) break;
          if (attempt == 99) pdir = rdir;
        }
      }
      child = trace_ray_a(c, hit, normalize(pdir, EPS), max_dist - min, depth - 1, importance * refl, st);
    } else {
      double eta1 = mat_refr(s, st->len >= 1 ? st->m[st->len - 1] : -1);
      double eta2 = (dot(d, n) < 0) ? mat_refr(s, mat) : (st->len >= 2 ? mat_refr(s, st->m[st->len - 2]) : NAN);
      double ratio = eta1 / eta2;                                               /* eta2 undefined: NaN flows on (:929-931) */
      double c2sq = 1 - ratio * ratio * (1 - cos1 * cos1);
      if (c2sq < 0) {
        v3 rdir = plus(d, scal(2 * cos1, n));
        child = trace_ray_a(c, hit, normalize(rdir, EPS), max_dist - min, depth - 1, importance * refl, st);
      } else {
        double pm = (cos1 < 0) ? -1 : 1;
        v3 fdir = plus(scal(ratio, d), scal(ratio * cos1 - pm * sqrt(c2sq), n));
        if (cos1 < 0) { if (st->len > 0) st->len--; } else if (st->len < STACK_MAX) { st->m[st->len++] = (uint8_t)mat; }
        child = trace_ray_a(c, hit, normalize(fdir, EPS), max_dist - min, depth - 1, (1 - 0.0) * importance * refl, st);
      }
    }
    if (!child.ok) { out.ok = 0; return out; }                                  /* the caller dereferences undefined and throws */
    T[0] = child.r; T[1] = child.g; T[2] = child.b;
  }
  double res[3];
  for (int k = 0; k < 3; k++) {                                                 /* :979-986, Math.round = floor(x + 0.5) */
    double v = floor(((1 - refl) * sc[k] + refl * T[k]) + 0.5);
    res[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
  }
  out.r = res[0]; out.g = res[1]; out.b = res[2];
  return out;
}

void ora_render_whitted(const ora_scene *s, const ora_params *P, int x0, int y0, int w, int h, double *rgba,
                        double *post, uint64_t *n_rays) {
  rng_t g; g.total = 0;
  rctx c = { s, P, NULL, &g, 0 };
  const double *cam = s->camera;
  v3 co = ld(cam), cu = ld(cam + 3), cv = ld(cam + 6), cw = ld(cam + 9);
  const int S = P->sub_sample;
  for (int yy = 0; yy < h; yy++)
    for (int xx = 0; xx < w; xx++) {
      int X = x0 + xx, Y = y0 + yy;
      uint32_t pixel = (uint32_t)(Y * (int)s->width + X);
      double tot[4] = { 0, 0, 0, 0 }; cola_t last = { 1, 0, 0, 0, 0 }; int bad = 0;
      for (int sy = 0; sy < S; sy++)
        for (int sx = 0; sx < S; sx++) {                                       /* ray.js:814-835 */
          rng_set(&g, P->seed, ORA_DOMAIN_RENDER, pixel, (uint32_t)(sy * S + sx));
          double u = ((X + (sx + rng_next(&g)) / S) * 2 / s->width) - 1;
          double v = -(((Y + (sy + rng_next(&g)) / S) * 2 / s->height) - 1);
          u *= cam[12]; v *= cam[13];
          v3 xyz = plus(plus(plus(co, scal(u, cu)), scal(v, cv)), scal(-1, cw));
          mstack st; st.len = 1; st.m[0] = 0;
          last = trace_ray_a(&c, co, normalize(minus(xyz, co), P->epsilon), P->max_trace_dist, P->max_depth, 1, &st);
          if (!last.ok) bad = 1;
          tot[0] += last.r; tot[1] += last.g; tot[2] += last.b; tot[3] += last.a;   /* colour() does not clamp in ray.js:102 */
        }
      size_t o4 = 4 * ((size_t)yy * w + xx);
      if (rgba) { rgba[o4] = bad ? NAN : last.r; rgba[o4 + 1] = bad ? NAN : last.g; rgba[o4 + 2] = bad ? NAN : last.b; rgba[o4 + 3] = bad ? NAN : last.a; }
      if (post) for (int k = 0; k < 4; k++) post[o4 + k] = bad ? NAN : floor(tot[k] / (S * S));   /* :837-840 */
    }
  if (n_rays) *n_rays = c.rays;
}
