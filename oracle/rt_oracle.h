/*
 * TEST INFRASTRUCTURE -- CPU oracle for the ray-tracing / photon-mapping hot path.
 *
 * Plain-C, IEEE fp64, no FMA contraction restatement of the reference algorithm
 * (rayphoton.js / kdtree.js).  It exists only as the checker for tests/, for
 * __graft_entry__.smoke() and for bench.py's cpu_baseline leg; nothing in the product
 * path (raytracer_amd/, include/) may link, load or call it.
 *
 * Pinned against fixtures generated from the reference itself (tests/golden/, made by
 * oracle/harness/gen_golden.js); see tests/test_oracle_golden.py.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_PLANE = 0, ORA_SPHERE, ORA_HEMISPHERE, ORA_CYLINDER, ORA_TRIANGLE, ORA_SQUARE, ORA_DISC, ORA_ANNULUS };
enum { ORA_FLAG_CONVEX = 1, ORA_FLAG_LIGHT = 2, ORA_FLAG_TRUNC_MAX = 4, ORA_FLAG_HALF_HEIGHT = 8, ORA_FLAG_FIRST_ROOT = 16 };
enum { ORA_DOMAIN_RENDER = 0, ORA_DOMAIN_EMIT1 = 1, ORA_DOMAIN_EMIT2 = 2, ORA_DOMAIN_AUX = 3 };

/* Scene description: the same flat arrays the product C ABI takes (include/rt355.h). */
typedef struct ora_scene {
  uint32_t n_shapes;
  const uint8_t *shape_meta;   /* n x 4: type, material (255 = none), flags, proc colour id */
  const double *shape_geom;    /* n x 12 */
  const double *shape_colour;  /* n x 4 */
  uint32_t n_lights;
  const uint32_t *light_shape; /* n_lights */
  const double *light_data;    /* n_lights x 8: wattage area prob r g b a reserved */
  double camera[16];           /* origin3 u3 v3 w3 fovScaleWidth fovScaleHeight */
  uint32_t width, height;
  double materials[4][8];      /* transparent, refrIndex, reflectance, specular; NaN = undefined */
  const double *shape_surface; /* n x 2 per-shape reflectance, specular (ray.js shapes), or NULL */
} ora_scene;

typedef struct ora_params {
  double epsilon;              /* rayphoton.js:6   1e-6 */
  double max_trace_dist;       /* rayphoton.js:8   20   */
  int32_t max_depth;           /* rayphoton.js:9   5    */
  int32_t sub_sample;          /* rayphoton.js:11       */
  uint32_t num_photons_diffuse;/* rayphoton.js:16       */
  uint32_t num_photons_caustic;/* rayphoton.js:17       */
  double diffuse_radius;       /* rayphoton.js:19  0.40 */
  double caustic_radius;       /* rayphoton.js:18  0.15 */
  double standard_lamp_area;   /* rayphoton.js:23  100  */
  uint64_t seed;
  int32_t use_direct_light;    /* 0 disables directLight (config "primary + reflect/refract only") */
  int32_t use_photon_maps;     /* 0 disables both radius gathers */
  int32_t super_sample_block;  /* SUPER_SAMPLE_BASE ** superSampleScale (rayphoton.js:1589-1596): one jittered sample per
                                  block of this many pixels squared; 0 or 1 = every pixel */
  int32_t pad_;
} ora_params;

/* photon record: origin3 dir3 power r g b a  (11 doubles), as rayphoton.js:647-654 */
#define ORA_PHOTON_STRIDE 11

void ora_default_params(ora_params *p);

/* RNG */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double ora_uniform(uint64_t seed, uint32_t domain, uint32_t a, uint32_t b, uint32_t draw);

/* geometry kernel */
void ora_normalize(const double v[3], double out[3]);
double ora_intersect(const double o[3], const double d[3], int type, int flags, const double g[12], double eps);
void ora_intersect_batch(int n, const double *o, const double *d, const uint8_t *meta, const double *geom,
                         double eps, int normalize_dir, double *t_out);
/* mode: 0 closest incl. lights, 1 closest excluding lights, 2 any-hit (needs source_dist) */
void ora_scan_batch(const ora_scene *s, double eps, int mode, int n, const double *o, const double *d,
                    const double *source_dist, double *t_out, int32_t *idx_out);

/* samplers (stream = (domain,a,b), consumed from draw 0) */
void ora_vec_perturb_keyed(uint64_t seed, uint32_t domain, uint32_t a, uint32_t b, uint32_t first_draw,
                           const double v[3], double out[3]);
void ora_disc_sample_keyed(uint64_t seed, uint32_t domain, uint32_t a, uint32_t b, uint32_t first_draw,
                           const double c[3], double radius, const double n[3], double out[3]);
int ora_create_photon_keyed(const ora_scene *s, uint64_t seed, uint32_t domain, uint32_t a, uint32_t b,
                            double out8[8]);
/* directLight (rayphoton.js:1841-1880) at one point on stream (P->seed, domain, a, b) from first_draw; returns the
 * stream position afterwards */
uint32_t ora_direct_light_keyed(const ora_scene *s, const ora_params *P, uint32_t domain, uint32_t a, uint32_t b, uint32_t first_draw,
                                const double point[3], const double normal[3], const double local[3], double out[3]);

/* photon emission: keyed restatement of emitManyPhotons.  Returns 0, or 1 when an attempt cap
 * was hit.  Lists are in store order.  tags: isCaustic, phase, attempt (as int64 triplets). */
int ora_emit_many(const ora_scene *s, const ora_params *p,
                  double *diffuse, int64_t *diffuse_tag, uint32_t *n_diffuse,
                  double *caustic, int64_t *caustic_tag, uint32_t *n_caustic,
                  uint64_t *attempts_phase1, uint64_t *attempts_phase2, uint64_t *draws, uint64_t *segments);
/* one batch of attempts of one phase, for sharding tests: out row per attempt:
 * flag (0 none, 1 diffuse, 2 caustic) + record */
void ora_emit_attempts(const ora_scene *s, const ora_params *p, int phase, uint64_t a0, uint64_t a1,
                       int32_t *flag, double *rec);

/* kd-tree */
uint32_t ora_find_median(uint32_t n);
int ora_kd_build(const double *pos, uint32_t n, int32_t *perm, uint8_t *dim);
/* radius query in reference pre-order; returns number of hits (may exceed cap; only cap written) */
uint32_t ora_kd_radius(const double *pos, uint32_t n, const int32_t *perm, const uint8_t *dim,
                       const double x[3], double r, int32_t *out, uint32_t cap);
/* nearestNeighbourAndDistance(x) (kdtree.js:10-62); returns 1 and (-1, NaN) for [undefined, undefined] */
int ora_kd_nearest(const double *pos, uint32_t n, const int32_t *perm, const uint8_t *dim, const double x[3], int32_t *idx, double *dist);
void ora_knn_brute(const double *pos, uint32_t n, const double x[3], uint32_t k, int32_t *idx, double *d2);

/* renderer (mode B).  Photon maps are given as record lists; trees are built inside. */
typedef struct ora_maps ora_maps;
ora_maps *ora_maps_create(const double *diffuse, uint32_t nd, const double *caustic, uint32_t nc);
void ora_maps_destroy(ora_maps *m);
/* renders pixels [x0,x0+w) x [y0,y0+h) -- with super_sample_block b > 1 one sample per b x b block, outputs sized
 * ceil(h/b) x ceil(w/b); pre = raw traceRay sums (rgba per sub-sample averaged is NOT
 * applied: with sub_sample==1 pre is the single traceRay result), post = what putPixel receives. */
void ora_render(const ora_scene *s, const ora_params *p, const ora_maps *m,
                int x0, int y0, int w, int h, double *pre_rgba, double *post_rgba,
                uint64_t *n_rays, uint64_t *n_tests);

/* the older tracer, ray.js:801-993 ("mode A"): raw traceRay colour per pixel (sub_sample 1: one path per
 * pixel); rgba = NaN where the reference's traceRay returns undefined (ray.js:905) */
void ora_render_whitted(const ora_scene *s, const ora_params *p, int x0, int y0, int w, int h, double *rgba,
                        double *post_rgba, uint64_t *n_rays);

#ifdef __cplusplus
}
#endif
#endif
