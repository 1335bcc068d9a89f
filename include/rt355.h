/*
 * rt355 -- C ABI of the MI355X ray-tracing / photon-mapping core (librt355.so).
 *
 * The reference (TheophileMot/raytracer) is one JavaScript realm with no FFI of its own, so
 * each entry point below names the reference function(s) it stands in for; the Node host
 * (raytracer_amd/host) calls them through the N-API shim and keeps the reference's
 * JavaScript-visible surface (Scene, Ray, KdTree ...) on top.
 *
 * Conventions
 *   - every function returns an int status (RT_OK = 0); rt_last_error() gives the message.
 *     "No hit" is a value (index -1, t = +inf), never an error -- as `undefined` is in the
 *     reference.
 *   - array arguments may be host pointers or device pointers of the context's GPU; the
 *     library detects which.  Host arrays are borrowed for the duration of the call only.
 *   - the kernels of one context run on the HIP stream rt_stream() returns, with one exception: after
 *     rt_set_frames_in_flight(ctx, n > 1), frames that rt_trace_tile only queues rotate through n streams of the
 *     context.  rt_synchronize() waits for all of them on the host; rt_join() orders rt_stream() behind all of them
 *     on the device.  A consumer that waits on rt_stream() alone must call rt_join() first.
 *   - nothing here allocates through, or exposes, any framework type: plain pointers/sizes.
 */
#ifndef RT355_H
#define RT355_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

enum { RT_OK = 0, RT_ERR_INVALID = -1, RT_ERR_HIP = -2, RT_ERR_NO_DEVICE = -3, RT_ERR_EMPTY_TREE = -4,
       RT_ERR_ATTEMPT_CAP = -5, RT_ERR_MATERIAL_STACK = -6, RT_ERR_UNKNOWN_SHAPE = -7, RT_ERR_TOO_LARGE = -8 };

/* primitive types: the `type` strings of rayphoton.js:353-481 */
enum { RT_PLANE = 0, RT_SPHERE, RT_HEMISPHERE, RT_CYLINDER, RT_TRIANGLE, RT_SQUARE, RT_DISC, RT_ANNULUS, RT_NUM_TYPES };
enum { RT_FLAG_CONVEX = 1, RT_FLAG_LIGHT = 2, RT_FLAG_TRUNC_MAX = 4, RT_FLAG_HALF_HEIGHT = 8,
       RT_FLAG_FIRST_ROOT = 16 /* ray.js cylinder: infinite, first root only (ray.js:443-461) */ };
/* procedural albedo closures of the presets (rayphoton.js:746-755, :756-768, :784-792, :871-874) */
enum { RT_PROC_NONE = 0, RT_PROC_STRIPE_CHECKER = 1, RT_PROC_EIGHT_BALL = 2, RT_PROC_TILE_GRID = 3, RT_PROC_ROTATED_CHECKER = 4,
       RT_PROC_STRIPE_CHECKER3 = 5 /* ray.js:650-659 */ };
enum { RT_PRECISION_F32 = 0, RT_PRECISION_F64 = 1 };
enum { RT_MODE_PHOTON = 0 /* rayphoton.js traceRay */, RT_MODE_WHITTED = 1 /* ray.js traceRay */ };
enum { RT_SCAN_CLOSEST = 0, RT_SCAN_CLOSEST_NOLIGHT = 1, RT_SCAN_ANY = 2 };
/* RNG stream domains: key=(seed), counter=(draw>>2, a, b, domain); U=(word[draw&3]>>8)*2^-24 */
enum { RT_DOMAIN_RENDER = 0, RT_DOMAIN_EMIT_PHASE1 = 1, RT_DOMAIN_EMIT_PHASE2 = 2, RT_DOMAIN_AUX = 3 };

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;
typedef struct rt_photon_map rt_photon_map;
typedef struct rt_photon_lists rt_photon_lists;

/* Flat scene: what `new Scene(ctx)` + `loadPreset` leave in scene.shapes / scene.lights /
 * scene.camera (rayphoton.js:715-727, :1313-1324), flattened by the host. */
typedef struct rt_scene_desc {
  uint32_t n_shapes;
  const uint8_t *shape_meta;    /* n_shapes x 4: type, material (255 = none), flags, procedural colour id */
  const double *shape_geom;     /* n_shapes x 12, slot assignment documented in raytracer_amd/scene.py */
  const double *shape_colour;   /* n_shapes x 4: baseColour r g b (0..255) a (0..1; NaN = absent) */
  uint32_t n_lights;
  const uint32_t *light_shape;  /* n_lights: index of the light's disc in the shape list (rayphoton.js:322) */
  const double *light_data;     /* n_lights x 8: wattage, area, prob, lightColour r g b a, reserved */
  double camera[16];            /* origin3 ONBu3 ONBv3 ONBw3 fovScaleWidth fovScaleHeight, 2 reserved */
  uint32_t width, height;       /* canvas size */
  double materials[4][8];       /* rows: transparent(0/1), refrIndex, reflectance, specular; NaN = undefined */
  const double *shape_surface;  /* n_shapes x 2: per-shape reflectance, specular -- the ray.js shape model
                                   (ray.js:244-385), read by RT_MODE_WHITTED only; may be NULL otherwise */
} rt_scene_desc;

/* The reference's top-of-file constants (rayphoton.js:6-23) plus run controls. */
typedef struct rt_params {
  double epsilon;               /* EPSILON 1e-6 */
  double max_trace_dist;        /* MAX_TRACE_DIST 20 (ray.js: 100) */
  int32_t max_depth;            /* MAX_DEPTH 5 (ray.js: 20) */
  int32_t sub_sample;           /* SUB_SAMPLE: sub_sample^2 paths per pixel */
  uint32_t num_photons_diffuse; /* NUM_PHOTONS_DIFFUSE */
  uint32_t num_photons_caustic; /* NUM_PHOTONS_CAUSTIC */
  double diffuse_radius;        /* DIFFUSE_RADIUS 0.40 */
  double caustic_radius;        /* CAUSTIC_RADIUS 0.15 */
  double standard_lamp_area;    /* STANDARD_LAMP_AREA 100 */
  uint64_t seed;                /* Philox key */
  int32_t use_direct_light;     /* 0: skip directLight (rayphoton.js:1750-1752) */
  int32_t use_photon_maps;      /* 0: skip both radius gathers (rayphoton.js:1765, :1788) */
  int32_t precision;            /* RT_PRECISION_F32 (throughput) or RT_PRECISION_F64 (parity) */
  int32_t mode;                 /* RT_MODE_PHOTON or RT_MODE_WHITTED */
  double shadow_bias;           /* offset along the normal for shadow rays; 0 = the working epsilon (rayphoton.js:1751) */
  double epsilon_f32;           /* the epsilon the fp32 kernels use instead of `epsilon` (0 = 1e-4): 1e-6 is below
                                   fp32 resolution at scene scale, so on-surface origins would re-hit their surface */
  int32_t max_bounces;          /* hard cap on path segments (0 = 4096); the reference has none: a path caught between
                                   glass surfaces goes on until max_trace_dist is used up (hundreds of segments) */
  int32_t super_sample_block;   /* SUPER_SAMPLE_BASE ** superSampleScale (rayphoton.js:1574-1596, :1911): rt_trace_tile traces one
                                   jittered sample per block of this many pixels squared and its outputs hold one value per
                                   block, ceil(h / block) rows of ceil(w / block); 0 or 1 = every pixel */
  /* Multi-GPU tile split (no counterpart in the reference, whose traceTile calls are serial): with band_count > 1
   * rt_trace_tile traces only the rows of the tile that lie in the bands band_index, band_index + band_count, ... of
   * band_rows rows each (a power of two, >= 8) -- rank band_index's interleaved share -- and its outputs hold those
   * rt_band_rows(h, ...) rows only, top to bottom.  Pixels come out exactly as in the undivided tile (RNG streams are
   * keyed by canvas pixel). */
  int32_t band_rows, band_count, band_index;
  int32_t reserved_;
} rt_params;

typedef struct rt_trace_stats {
  uint64_t rays;                /* closest-hit segments traced (one per reference traceRay call that scans) */
  uint64_t shadow_rays;         /* any-hit segments (directLight) */
  uint64_t gathers;             /* photon-map queries (one per map per opaque hit) */
  uint64_t bounces;             /* segments of the longest path */
  uint64_t stack_overflows;     /* paths whose material stack exceeded the fixed capacity */
  uint64_t tail_rays;           /* of `rays`: segments traced after the first pass by the kernels that follow paths to their end */
  uint64_t prim_tests;          /* ray-primitive distance tests executed for `rays` (+ shadow rays when a hierarchy is used) */
  uint64_t prims_per_ray_brute_force;   /* primitives in the scene: rays * this = the tests of the reference's loops */
  uint64_t gather_saturations;  /* queued photon gathers (fp32 frames, see rt_trace_tile): partial sums that left the range of a path's
                                   fixed-point accumulator (|x| < 2^31 in steps of 2^-32 per colour channel) and were clamped to it,
                                   or were NaN and left out.  0 on any scene whose radiance stays below 2e9 */
} rt_trace_stats;

typedef struct rt_emit_stats {
  uint64_t attempts_phase1, attempts_phase2;
  uint64_t segments;            /* emitPhoton calls: of exactly the attempts the sequential loop runs (rt_emit_many*), of the whole
                                   range (rt_emit_photons, rt_emit_shard) */
  uint32_t n_diffuse, n_caustic;
  int32_t capped;               /* 1 when an attempt cap was hit (rayphoton.js:1365-1368, :1377-1380) */
  int32_t errors;               /* attempts that would throw "photon exited all materials" (rayphoton.js:1489) */
} rt_emit_stats;

/* ---- context ---- */
void rt_default_params(rt_params *p);
/* device_ids: HIP ordinals; n == 0 or NULL means device 0.  One context drives one GPU
 * (multi-GPU = one process and one context per GPU). */
int rt_init(const int *device_ids, int n, rt_ctx **out);
/* Scenes, photon maps and photon lists of a context are freed before it (freeing a map hands its device blocks to the
 * context's pool). */
void rt_shutdown(rt_ctx *ctx);
const char *rt_last_error(rt_ctx *ctx);     /* ctx may be NULL: last rt_init failure */
void *rt_stream(rt_ctx *ctx);               /* the hipStream_t work is launched on (see the conventions above) */
int rt_synchronize(rt_ctx *ctx);            /* host wait for everything the context has queued, on all its streams */
/* Device-side join: work put on rt_stream() after this call runs after every frame queued so far, whichever of the
 * context's streams it went to.  No host wait.  Needed only with rt_set_frames_in_flight(ctx, n > 1) by callers that
 * order their own work behind rt_stream() (an event, hipStreamWaitEvent, a framework's external-stream wrapper)
 * instead of calling rt_synchronize().  No counterpart in the reference. */
int rt_join(rt_ctx *ctx);
/* Experiment / test switches, by name: tail_threshold, whole_frame, bounce_passes, tail_handoff, tail_rounds,
 * tail_persistent, tail_persistent_handoff, no_lds, no_cull, cull_lists, primary_tiles, raster_order, bvh (-1 by scene size / 0 / 1) and scan_groups (0 / 1)
 * (both read by rt_scene_upload), bvh_one_order, defer_gathers, gather_sort, gather_split, wave_gather, knn_heap, knn_sort (-1: order the k-nearest queries
 * by position when they arrive scattered / 0 never / 1 always),
 * knn_radius_scale, knn_verbose, probe.  rt_init reads
 * RT355_<NAME> from the environment once; this call changes one switch of a live context.  The defaults are the
 * measured best; results do not depend on any of them (fp32 images: bit for bit).  No counterpart in the reference. */
int rt_set_option(rt_ctx *ctx, const char *name, double value);
/* n = 2..8: up to n consecutive frames that rt_trace_tile only queues (see there) may overlap -- the long, thinly
 * occupied end of one frame's path tracing runs beside the start of the next; frames in flight together must not
 * share output buffers.  n = 1 (default): queued frames run one after the other.  No counterpart in the reference (its frames are
 * serial by construction). */
int rt_set_frames_in_flight(rt_ctx *ctx, int n);
/* Device memory of the context's GPU for hosts without a HIP binding of their own (the Node addon: N-API has no device
 * arrays): buffers for the outputs of queued frames, for shards and for gather inputs.  rt_device_copy copies between
 * any two of host and device memory and returns when the bytes are there; it first waits for everything the context has
 * queued (so a queued frame's output can be read straight after queueing it), and rt_device_free does too.  A caller that
 * has hipMalloc (or torch) uses that instead: the library accepts any device pointer.  No counterpart in the reference. */
int rt_device_alloc(rt_ctx *ctx, uint64_t bytes, void **out);
int rt_device_free(rt_ctx *ctx, void *ptr);
int rt_device_copy(rt_ctx *ctx, void *dst, const void *src, uint64_t bytes);
/* Per-kernel HIP-event timing (off by default).  rt_profile_get sums the launches of one
 * kernel name since the last reset; it synchronises the stream.  on = n > 1: only every n-th launch of a kernel
 * name is timed (a sample: a launch that carries events does not overlap its neighbours as freely). */
int rt_profile_enable(rt_ctx *ctx, int on);
/* restrict event timing to the launches of the named kernels, comma-separated, e.g. "primary,tail"
 * (NULL or "" = all kernels again) */
int rt_profile_select(rt_ctx *ctx, const char *kernels);
int rt_profile_reset(rt_ctx *ctx);
int rt_profile_get(rt_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches);

/* ---- scene: `new Scene(ctx)` + `loadPreset` + `initLights` (rayphoton.js:715-727, :1313-1324) ---- */
int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *desc, rt_scene **out);
void rt_scene_free(rt_scene *scene);

/* ---- geometry kernel: `new Ray(o,d).intersectDist(shape)` (rayphoton.js:483-645) ----
 * pairwise: ray i against shape record i (n shapes given inline). t = NaN for `undefined`. */
int rt_intersect_pairs(rt_ctx *ctx, const rt_params *p, uint32_t n, const double *ray_o, const double *ray_d,
                       int normalize_dir, const uint8_t *shape_meta, const double *shape_geom, double *t_out);
/* scene scans: the closest-hit loops (rayphoton.js:1642-1650, :1467-1475) and the any-hit
 * shadow loop (:1853-1857).  idx = -1 / t = +inf on a miss; mode RT_SCAN_ANY writes 0/1 to idx.
 * ray_d: unit vectors, as the reference's Ray constructor leaves them (its sphere, hemisphere and cylinder tests
 * assume a unit direction, and so does the fp32 scans' skipping of primitive groups out of a wave's reach). */
int rt_intersect(rt_scene *scene, const rt_params *p, int mode, uint32_t n, const double *ray_o,
                 const double *ray_d, const double *source_dist, double *t_out, int32_t *idx_out);

/* ---- photon emission: `emitManyPhotons` / `createPhoton` / `emitPhoton` / `storePhoton`
 *      (rayphoton.js:1326-1383, :1461-1557) ----
 * One RNG stream per attempt: (RT_DOMAIN_EMIT_PHASE<phase>, attempt).  Records are 11 doubles:
 * origin3 dir3 power colour_rgba (rayphoton.js:647-654). */
/* attempts [begin,end) of one phase, compacted in attempt order.  kind: 1 diffuse, 2 caustic.
 * cap = capacity of the output arrays in records (end-begin always suffices). */
int rt_emit_photons(rt_scene *scene, const rt_params *p, int phase, uint64_t attempt_begin, uint64_t attempt_end,
                    uint32_t cap, double *rec_out, int64_t *attempt_out, uint8_t *kind_out, uint32_t *n_out,
                    rt_emit_stats *stats);
/* the whole two-phase loop with the reference's caps; lists come back in store order.
 * diffuse/caustic hold num_photons_* x 11 doubles; *_attempt hold (phase, attempt) pairs. */
int rt_emit_many(rt_scene *scene, const rt_params *p, double *diffuse, int64_t *diffuse_attempt,
                 double *caustic, int64_t *caustic_attempt, rt_emit_stats *stats);

/* ---- device-resident emission: the same loop with the photon lists kept in GPU memory ----
 * photonListDiffuse / photonListCaustic with their caps (storePhoton, rayphoton.js:1346-1352) live in an rt_photon_lists
 * object on the GPU.  A SHARD is what an attempt range leaves: its stored photons compacted in attempt order as records of
 * W + 1 32-bit words, W words of payload and a tag = (attempt - tag_base) | (caustic ? 1 << 31 : 0).
 *   RT_SHARD_FULL   payload = the 11 record values in the working precision (W = 11 for fp32, 22 for fp64)
 *   RT_SHARD_PACKED payload = position, direction, power * colour as 9 floats (W = 9): the 36-byte photon of the map
 * Multi-GPU emission (one process per GPU): every rank runs rt_emit_shard on its sub-range of a batch with the batch's
 * first attempt as tag_base, the shards and their counts are all-gathered in rank order (any transport; the buffers are
 * plain device memory), and every rank calls rt_lists_apply on the identical merged batch: identical lists, hence
 * identical photon maps, on every rank.  No counterpart in the reference (its loop is sequential); results are the
 * sequential loop's for any batch size and rank count because every attempt has its own RNG stream. */
enum { RT_SHARD_FULL = 0, RT_SHARD_PACKED = 1 };
typedef struct rt_lists_info {
  uint32_t n_diffuse, n_caustic;          /* list lengths after the batch */
  uint32_t batch_diffuse, batch_caustic;  /* stored photons of either kind in the batch */
  uint32_t taken_diffuse, taken_caustic;  /* ... of which the lists took */
  int32_t phase_ended;                    /* 1: an attempt of this batch filled the phase's list (rayphoton.js:1361, :1373) */
  uint64_t last_attempt;                  /* that attempt: the phase's attempt counter is last_attempt + 1 */
} rt_lists_info;
/* lists with the caps num_photons_diffuse / num_photons_caustic of p, records in `format` and p->precision */
int rt_lists_create(rt_ctx *ctx, const rt_params *p, int format, rt_photon_lists **out);
void rt_lists_free(rt_photon_lists *lists);
uint32_t rt_lists_record_words(const rt_photon_lists *lists);     /* W */
/* attempts [begin, end) of one phase -> shard_out (capacity cap records of W + 1 words; end - begin always suffices),
 * number of stored photons -> *n_out (4 bytes; like shard_out, device or host memory). */
int rt_emit_shard(rt_scene *scene, const rt_params *p, int phase, uint64_t attempt_begin, uint64_t attempt_end, int format,
                  uint64_t tag_base, uint32_t cap, uint32_t *shard_out, uint32_t *n_out, rt_emit_stats *stats);
/* one batch of n_shards shards, shard s at shards + s * stride * (W + 1) words with counts[s] records (device or host
 * memory), in attempt order: appended to the lists under the reference's caps.  Returns after the (tiny) state read-back. */
int rt_lists_apply(rt_photon_lists *lists, int phase, uint64_t tag_base, const uint32_t *shards, const uint32_t *counts,
                   uint32_t n_shards, uint32_t stride, rt_lists_info *info);
int rt_lists_counts(rt_photon_lists *lists, uint32_t *n_diffuse, uint32_t *n_caustic);
/* which: 0 diffuse, 1 caustic.  rec_out: n x 11 doubles (RT_SHARD_FULL lists only), attempt_out: n (phase, attempt) pairs;
 * either may be NULL; device or host memory */
int rt_lists_export(rt_photon_lists *lists, int which, double *rec_out, int64_t *attempt_out);
/* the list's records where they are: device pointer, length, W, precision */
int rt_lists_view(rt_photon_lists *lists, int which, const void **records_dev, uint32_t *n, uint32_t *words, int *precision);
/* rt_emit_many that leaves the lists on the GPU (*lists_out is set whenever lists exist, also with RT_ERR_ATTEMPT_CAP) */
int rt_emit_many_lists(rt_scene *scene, const rt_params *p, int format, rt_photon_lists **lists_out, rt_emit_stats *stats);

/* ---- photon map: `new KdTree(photons)` (kdtree.js:1-8, :82-124) ---- */
/* rec: n x 11 doubles as above.  Positions are rounded to fp32 for the device record
 * (pos 3xf32, dir 3xf32, power*colour 3xf32); the tree is the reference's left-balanced
 * tree over those rounded positions.  n == 0 -> RT_ERR_EMPTY_TREE (kdtree.js:6). */
int rt_photon_map_build(rt_ctx *ctx, const double *rec, uint32_t n, rt_photon_map **out);
/* the same from packed photons (n x 9 floats: position, direction, power * colour; device or host memory) and from a
 * device-resident list (which: 0 diffuse, 1 caustic) without the records leaving the GPU */
int rt_photon_map_build_packed(rt_ctx *ctx, const float *photons, uint32_t n, rt_photon_map **out);
int rt_photon_map_build_lists(rt_photon_lists *lists, int which, rt_photon_map **out);
void rt_photon_map_free(rt_photon_map *map);
uint32_t rt_photon_map_size(const rt_photon_map *map);
/* nodes[pos] for pos = 1..n: perm[pos-1] = index of nodes[pos].photon in the input list,
 * dim[pos-1] = nodes[pos].dim (kdtree.js:104) */
int rt_photon_map_nodes(rt_photon_map *map, int32_t *perm, uint8_t *dim);
/* tree build alone, on fp32 positions (n x 3) */
int rt_kd_build(rt_ctx *ctx, const float *pos, uint32_t n, int32_t *perm, uint8_t *dim);
uint32_t rt_find_median(uint32_t n);        /* kdtree.js:109-124 */

/* `nearestNeighbours(x, r)` (kdtree.js:64-80): count[q] hits, the first min(count, cap) input
 * indices in the reference's pre-order at idx[q*cap ...]. queries: nq x 3 fp32. */
int rt_kd_radius(rt_photon_map *map, const float *queries, uint32_t nq, double r, uint32_t cap,
                 uint32_t *count, int32_t *idx);
/* `nearestNeighbourAndDistance(x)` (kdtree.js:10-62) with the reference's own semantics: the closest node of the one
 * root-to-leaf path the query descends (the far side of a plane is never searched), idx[q] = input index of that photon,
 * dist[q] = its distance; `[undefined, undefined]` (a path that leaves through a missing upper child, :13) is -1 / NaN.
 * The exact nearest photon is rt_kd_knn with k = 1. */
int rt_kd_nearest_path(rt_photon_map *map, const float *queries, uint32_t nq, int32_t *idx, double *dist);
/* k nearest photons ordered by (squared distance, input index); slots beyond n hold -1 / +inf.
 * The reference has no k-NN query; the order is defined by brute force (oracle/rt_oracle.c). */
int rt_kd_knn(rt_photon_map *map, const float *queries, uint32_t nq, uint32_t k, int32_t *idx, double *d2);
/* fused radiance estimate at nq surface points (rayphoton.js:1779-1784, :1808-1815):
 *   rgb = sum_p colour_p * power_p * max(0, dir_p . normal) * albedo/255 / (pi r^2)
 * over the photons within r (k == 0) or over the k nearest with r^2 = the k-th squared
 * distance (k > 0).  hits: nq x 9 fp32 = pos3 normal3 albedo3; rgb_out: nq x 3 fp32.
 * k == 0 with 16 384 queries or more on a map of 200 000 photons or more: the queries are sorted by position and served
 * 64 at a time (the photons of a ball are the same -- the fp64 d2 < r^2 decides -- their sum is taken in another order
 * than the tree walk's; the same from run to run).  Option defer_gathers = 0: every query walks on its own. */
int rt_gather_radiance(rt_photon_map *map, const float *hits, uint32_t nq, double r, uint32_t k, float *rgb_out);

/* rt_gather_radiance for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc, code between the captures of HIP graphs -- the call itself cannot be captured: it reads a few counters
 * back to the host between its launches and ends with a stream synchronisation).  NO COUNTERPART IN THE REFERENCE.  Same results as rt_gather_radiance, but the
 * call allocates nothing: `hits_dev` and `rgb_dev` must already be device memory of the context's GPU (a host array, or an
 * array on another GPU, is rejected with RT_ERR_INVALID), and the k-nearest gather's scratch (fallback flags and work lists) lives in a static
 * device array of about 3.8 MB that each GPU the library runs on holds until the process ends.  nq may be at most
 * RT_GATHER_STATIC_MAX_QUERIES (RT_ERR_TOO_LARGE beyond: split the queries over several calls).  All calls on one GPU
 * share that array: calls that overlap in time on one GPU (two contexts, two host threads) race.  Returns when the
 * results are complete.  With k == 0 every query walks the tree on its own lane: the bits of rt_gather_radiance's own
 * lane-by-lane walk; the queue rt_gather_radiance takes for many queries on a large map (the same photons summed in
 * another order, equal to fp32 rounding) needs buffers that grow with nq. */
#define RT_GATHER_STATIC_MAX_QUERIES 262144u
int rt_gather_radiance_static(rt_photon_map *map, const float *hits_dev, uint32_t nq, double r, uint32_t k, float *rgb_dev);

/* The adaptive-radius search sketched in the reference's commented block (rayphoton.js:1766-1778, :1789-1801): start at
 * r0 and, up to max_attempts times (10 there), multiply the radius by `grow` (3) while fewer than min_count (10) photons
 * are inside; fewer than min_keep (3) photons in the end give no contribution.  rgb as rt_gather_radiance with the area of
 * the radius that was used; r_used_out (nq floats, may be NULL) returns that radius. */
int rt_gather_radiance_adaptive(rt_photon_map *map, const float *hits, uint32_t nq, double r0, double grow, uint32_t min_count,
                                uint32_t min_keep, uint32_t max_attempts, float *rgb_out, float *r_used_out);

/* ---- renderer: `traceTile` / `traceOnCanvas` / `traceRay` / `directLight`
 *      (rayphoton.js:1573-1880; ray.js:801-993 for RT_MODE_WHITTED) ----
 * Renders pixels [x0,x0+w) x [y0,y0+h) of the scene's canvas with superSampleScale 0.
 * RNG stream per path: (RT_DOMAIN_RENDER, pixel index, sub-sample index).
 *   pre_rgba  : w*h*4 doubles, the traceRay colour of the pixel's last sub-sample before
 *               clamping (with sub_sample == 1: the pixel's radiance), alpha NaN-able
 *   post_rgba : w*h*4 doubles, what putPixel receives (rayphoton.js:1618-1623)
 *   rgba8     : w*h*4 bytes, post_rgba as 8-bit RGBA (alpha*255)
 * any of the three may be NULL. diffuse/caustic may be NULL (no map).
 * With `stats` the call returns when the frame is finished.  With stats == NULL and every output in device
 * memory it only queues the frame -- on rt_stream(), or, after rt_set_frames_in_flight(ctx, n > 1), on one of n streams
 * of the context in turn: the outputs are complete after rt_synchronize() (or any later call that returns host data),
 * or, for work the caller puts on rt_stream(), after rt_join(); consecutive frames run back to back without a host
 * round trip in between. */
uint32_t rt_band_rows(uint32_t height, int band_rows, int band_count, int band_index);
int rt_trace_tile(rt_scene *scene, const rt_params *p, rt_photon_map *diffuse, rt_photon_map *caustic,
                  int x0, int y0, int w, int h, double *pre_rgba, double *post_rgba, uint8_t *rgba8,
                  rt_trace_stats *stats);

/* ---- the scene's single-item methods, batched: explicit rays, points and photons ----
 * The reference calls these one item at a time from script; here each takes n items, one GPU lane an item (no
 * queues, no sorting: they are for callers that drive the tracer themselves, not for frames).  Item i draws from the
 * keyed stream (p->seed, domain, first_stream + i) -- for rays and points the domain is RT_DOMAIN_RENDER, so
 * first_stream = y * width + x continues the stream of pixel (x, y) of a frame with the same seed -- starting at draw
 * first_draw[i] (0 when first_draw is NULL).
 *
 * rt_trace_rays: `Scene.traceRay(ray, maxDist, depth, importance, materialStack)` (rayphoton.js:1626-1838).
 *   ray_o, ray_d : n*3 doubles; directions as the reference's Ray holds them, i.e. already of unit length
 *   max_dist, depth, importance : per ray, or NULL for p->max_trace_dist, 0, 1 (what traceOnCanvas passes, :1606)
 *   stacks       : n*16 bytes, host memory: byte 0 the length (1..15), then the material ids from the bottom of the
 *                  stack; NULL for [MAT_AIR]
 *   rgba_out     : n*4 doubles, the colour traceRay returns (alpha NaN-able)
 * rt_direct_light: `Scene.directLight(point, normal, localColour)` (rayphoton.js:1841-1880); points, normals and
 *   local_rgb are n*3 doubles, rgb_out n*3 doubles, draws_out (may be NULL) the stream position afterwards.
 * rt_create_photons: `Scene.createPhoton(source)` (rayphoton.js:1326-1344) on the streams of emission phase `phase`
 *   (1 or 2: the stream of attempt first_stream + i of that phase); source[i] >= 0 names a light, < 0 (or source ==
 *   NULL) lets the draw choose.  photons_out: n*11 doubles, origin3 dir3 power rgba.  Always 5 draws.
 * rt_walk_photons: `Scene.emitPhoton(photon, depth, trackOnlyCaustic, materialStack)` (rayphoton.js:1461-1557) for
 *   the n photons in `photons` (n*11 doubles as above; caustic[i] = -1 / 0 / 1 is the photon's isCaustic: undefined,
 *   false, true; NULL = all undefined).  kind_out[i] = 0: nothing stored, 1: a diffuse photon, 2: a caustic photon;
 *   where something is stored, photons[i] is replaced by the stored record (what storePhoton receives, power already
 *   divided by the list cap).  A createPhoton followed by emitPhoton on the same stream, first_draw = 5, is exactly
 *   attempt first_stream + i of rt_emit_photons. */
int rt_trace_rays(rt_scene *scene, const rt_params *p, rt_photon_map *diffuse, rt_photon_map *caustic, uint32_t n,
                  const double *ray_o, const double *ray_d, const double *max_dist, const int32_t *depth,
                  const double *importance, const uint8_t *stacks, uint32_t first_stream, const uint32_t *first_draw,
                  double *rgba_out, rt_trace_stats *stats);
int rt_direct_light(rt_scene *scene, const rt_params *p, uint32_t n, const double *points, const double *normals,
                    const double *local_rgb, uint32_t first_stream, const uint32_t *first_draw, double *rgb_out,
                    uint32_t *draws_out);
int rt_create_photons(rt_scene *scene, const rt_params *p, uint32_t n, int phase, uint64_t first_stream,
                      const int32_t *source, double *photons_out, uint32_t *draws_out);
int rt_walk_photons(rt_scene *scene, const rt_params *p, uint32_t n, int phase, uint64_t first_stream,
                    const uint32_t *first_draw, const int32_t *depth, const int8_t *caustic, const uint8_t *stacks,
                    int track_only_caustic, double *photons, uint8_t *kind_out);

#ifdef __cplusplus
}
#endif
#endif /* RT355_H */
