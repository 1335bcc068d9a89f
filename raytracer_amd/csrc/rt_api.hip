// C ABI of librt355.so (include/rt355.h): context, scene upload, and the entry points that
// dispatch to the fp32 / fp64 pipelines.  No CPU fallback exists: every entry point runs HIP
// kernels or fails.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <algorithm>
#include <limits>
#include <vector>
#include "rt_emit.h"
#include "rt_pipeline.h"

static std::string g_init_error;

int rt_fail(rt_ctx *ctx, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (ctx) ctx->err = buf; else g_init_error = buf;
  return code;
}

bool rt_is_device_ptr(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }     // plain host memory is "invalid value": clear it
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int rt_stage_in(rt_ctx *ctx, const void *src, size_t bytes, DevMem &tmp, const void **out) {
  if (!src && bytes) return rt_fail(ctx, RT_ERR_INVALID, "null input array");
  if (rt_is_device_ptr(src)) { *out = src; return RT_OK; }
  RT_HIP(ctx, tmp.alloc(bytes));
  if (bytes) RT_HIP(ctx, hipMemcpyAsync(tmp.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  *out = tmp.p;
  return RT_OK;
}
int rt_stage_out_begin(rt_ctx *ctx, void *dst, size_t bytes, DevMem &tmp, void **out) {
  if (!dst) { *out = nullptr; return RT_OK; }
  if (rt_is_device_ptr(dst)) { *out = dst; return RT_OK; }
  RT_HIP(ctx, tmp.alloc(bytes));
  *out = tmp.p;
  return RT_OK;
}
int rt_stage_out_end(rt_ctx *ctx, void *dst, size_t bytes, DevMem &tmp) {
  if (dst && tmp.p && bytes) RT_HIP(ctx, hipMemcpyAsync(dst, tmp.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

// (the entry points below take C linkage from their declarations in rt355.h)

void rt_default_params(rt_params *p) {
  memset(p, 0, sizeof *p);
  p->epsilon = 1e-6; p->max_trace_dist = 20; p->max_depth = 5; p->sub_sample = 1;
  p->num_photons_diffuse = 20000; p->num_photons_caustic = 50000;
  p->diffuse_radius = 0.40; p->caustic_radius = 0.15; p->standard_lamp_area = 100;
  p->seed = 12345; p->use_direct_light = 1; p->use_photon_maps = 1;
  p->precision = RT_PRECISION_F32; p->mode = RT_MODE_PHOTON; p->shadow_bias = 0; p->epsilon_f32 = 1e-4; p->max_bounces = 4096; p->super_sample_block = 0;
  p->band_rows = 0; p->band_count = 0; p->band_index = 0;
}

int rt_init(const int *device_ids, int n, rt_ctx **out) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    (void)hipGetLastError();
    return rt_fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device visible: librt355 has no CPU path");
  }
  int dev = (device_ids && n > 0) ? device_ids[0] : 0;
  if (dev < 0 || dev >= count) return rt_fail(nullptr, RT_ERR_NO_DEVICE, "device %d out of range (%d visible)", dev, count);
  rt_ctx *ctx = new rt_ctx();
  ctx->device = dev;
  bool streams_ok = hipSetDevice(dev) == hipSuccess;
  for (int i = 0; i < RT_MAX_IN_FLIGHT && streams_ok; i++) streams_ok = hipStreamCreateWithFlags(&ctx->streams[i], hipStreamNonBlocking) == hipSuccess;
  if (!streams_ok ||
      hipHostMalloc((void **)&ctx->pinned, 256, hipHostMallocDefault) != hipSuccess) {
    rt_fail(nullptr, RT_ERR_HIP, "context creation failed: %s", hipGetErrorString(hipGetLastError()));
    delete ctx;
    return RT_ERR_HIP;
  }
  ctx->stream = ctx->streams[0];
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ctx->n_cu = cus; else (void)hipGetLastError(); }
  for (int i = 1; i < RT_MAX_IN_FLIGHT; i++)
    if (hipEventCreateWithFlags(&ctx->frame_done[i], hipEventDisableTiming) != hipSuccess) {
      rt_fail(nullptr, RT_ERR_HIP, "context creation failed: %s", hipGetErrorString(hipGetLastError()));
      rt_shutdown(ctx);
      return RT_ERR_HIP;
    }
  // the experiment switches: the environment is read here, once per context
  static const char *names[] = { "tail_threshold", "whole_frame", "bounce_passes", "tail_handoff", "tail_rounds", "tail_persistent", "tail_persistent_handoff", "no_lds", "no_cull", "cull_lists", "primary_tiles", "scan_groups", "raster_order", "bvh", "bvh_one_order", "wave_gather", "defer_gathers", "gather_sort", "gather_split", "knn_heap", "knn_sort",
                                 "knn_verbose", "probe", "knn_radius_scale" };
  for (const char *nm : names) {
    std::string env = "RT355_";
    for (const char *c = nm; *c; c++) env += (char)toupper((unsigned char)*c);
    if (const char *e = getenv(env.c_str())) (void)rt_set_option(ctx, nm, atof(e));
  }
  int rc = rt_static_scratch_setup(ctx);
  if (rc != RT_OK) { g_init_error = ctx->err; rt_shutdown(ctx); return rc; }
  *out = ctx;
  return RT_OK;
}

// rows of a tile of `height` rows that fall into the bands index, index + count, index + 2 count, ... of band_rows rows
uint32_t rt_band_rows(uint32_t height, int band_rows, int band_count, int band_index) {
  if (band_count <= 1) return height;
  if (band_rows <= 0 || band_index < 0 || band_index >= band_count) return 0;
  const uint32_t period = (uint32_t)band_rows * (uint32_t)band_count, first = (uint32_t)band_rows * (uint32_t)band_index;
  const uint32_t full = height / period, rest = height % period;
  uint32_t rows = full * (uint32_t)band_rows;
  if (rest > first) rows += std::min<uint32_t>(rest - first, (uint32_t)band_rows);
  return rows;
}

int rt_set_option(rt_ctx *ctx, const char *name, double value) {
  if (!ctx || !name) return RT_ERR_INVALID;
  RtKnobs &K = ctx->knobs;
  const std::string n = name;
  if (n == "tail_threshold") K.tail_threshold = (long)value;
  else if (n == "whole_frame") K.whole_frame = value != 0;
  else if (n == "bounce_passes") K.bounce_passes = (int)value;
  else if (n == "tail_handoff") K.tail_handoff = (int)value;
  else if (n == "tail_rounds") K.tail_rounds = (int)value;
  else if (n == "tail_persistent") K.tail_persistent = (int)value;
  else if (n == "tail_persistent_handoff") K.tail_persistent_handoff = (int)value;
  else if (n == "no_lds") K.no_lds = value != 0;
  else if (n == "no_cull") K.no_cull = value != 0;
  else if (n == "cull_lists") K.cull_lists = value != 0;
  else if (n == "primary_tiles") K.primary_tiles = (int)value;
  else if (n == "scan_groups") K.scan_groups = value != 0;
  else if (n == "raster_order") K.raster_order = value != 0;
  else if (n == "bvh") K.bvh = (int)value;
  else if (n == "bvh_one_order") K.bvh_one_order = value != 0;
  else if (n == "wave_gather") K.wave_gather = value;
  else if (n == "defer_gathers") K.defer_gathers = (int)value;
  else if (n == "gather_sort") K.gather_sort = value != 0;
  else if (n == "gather_split") K.gather_split = (int)value;
  else if (n == "knn_heap") K.knn_heap = value != 0;
  else if (n == "knn_sort") K.knn_sort = (int)value;
  else if (n == "knn_verbose") K.knn_verbose = value != 0;
  else if (n == "probe") K.probe = value != 0;
  else if (n == "knn_radius_scale") K.knn_radius_scale = value;
  else return rt_fail(ctx, RT_ERR_INVALID, "unknown option '%s'", name);
  return RT_OK;
}

void rt_shutdown(rt_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (int i = 0; i < RT_MAX_IN_FLIGHT; i++) if (ctx->streams[i]) (void)hipStreamSynchronize(ctx->streams[i]);
  for (auto &kv : ctx->events) for (auto &ev : kv.second) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  for (auto &b : ctx->map_pool) (void)hipFree(b.second);
  ctx->map_pool.clear(); ctx->map_pool_bytes = 0;
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  for (int i = 0; i < RT_MAX_IN_FLIGHT; i++) if (ctx->frame_done[i]) (void)hipEventDestroy(ctx->frame_done[i]);
  for (int i = 0; i < RT_MAX_IN_FLIGHT; i++) if (ctx->streams[i]) (void)hipStreamDestroy(ctx->streams[i]);
  delete ctx;
}

const char *rt_last_error(rt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }
void *rt_stream(rt_ctx *ctx) { return ctx ? (void *)ctx->streams[0] : nullptr; }
static int rt_sync_other_streams(rt_ctx *ctx) {
  if (!ctx->alt_used) return RT_OK;
  for (int i = 1; i < RT_MAX_IN_FLIGHT; i++) { RT_HIP(ctx, hipStreamSynchronize(ctx->streams[i])); ctx->frame_pending[i] = false; }
  ctx->alt_used = false;
  return RT_OK;
}
// Device-side join: everything queued on rt_stream() from now on runs after every frame queued so far, on whichever
// of the context's streams it was put.  No host wait.
int rt_join(rt_ctx *ctx) {
  if (!ctx) return RT_ERR_INVALID;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  for (int i = 1; i < RT_MAX_IN_FLIGHT; i++)
    if (ctx->frame_pending[i]) { RT_HIP(ctx, hipStreamWaitEvent(ctx->streams[0], ctx->frame_done[i], 0)); ctx->frame_pending[i] = false; }
  return RT_OK;
}
int rt_synchronize(rt_ctx *ctx) {
  if (!ctx) return RT_ERR_INVALID;
  RT_HIP(ctx, hipStreamSynchronize(ctx->streams[0]));
  RT_TRY(rt_sync_other_streams(ctx));
  return RT_OK;
}

// Plain device memory of the context's GPU for hosts that have no HIP binding of their own (the Node addon): outputs of
// queued frames, shards, gather inputs.  The copy waits for everything the context has queued first.
int rt_device_alloc(rt_ctx *ctx, uint64_t bytes, void **out) {
  if (!ctx || !out) return RT_ERR_INVALID;
  *out = nullptr;
  if (bytes == 0) return rt_fail(ctx, RT_ERR_INVALID, "rt_device_alloc: zero bytes");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(out, (size_t)bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return rt_fail(ctx, RT_ERR_HIP, "rt_device_alloc(%llu bytes): %s", (unsigned long long)bytes, hipGetErrorString(e)); }
  return RT_OK;
}
int rt_device_free(rt_ctx *ctx, void *ptr) {
  if (!ctx) return RT_ERR_INVALID;
  if (!ptr) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_TRY(rt_synchronize(ctx));                       // queued frames may still write to it
  RT_HIP(ctx, hipFree(ptr));
  return RT_OK;
}
int rt_device_copy(rt_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
  if (!ctx || ((!dst || !src) && bytes)) return RT_ERR_INVALID;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_TRY(rt_synchronize(ctx));
  if (bytes) RT_HIP(ctx, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDefault));
  return RT_OK;
}

int rt_set_frames_in_flight(rt_ctx *ctx, int n) {
  if (!ctx) return RT_ERR_INVALID;
  if (n < 1 || n > RT_MAX_IN_FLIGHT) return rt_fail(ctx, RT_ERR_INVALID, "frames in flight must be in 1..%d", RT_MAX_IN_FLIGHT);
  RT_TRY(rt_synchronize(ctx));
  ctx->frames_in_flight = n;
  return RT_OK;
}

int rt_profile_enable(rt_ctx *ctx, int on) {
  if (!ctx) return RT_ERR_INVALID;
  ctx->profiling = on != 0; ctx->profile_every = on > 1 ? on : 1; ctx->profile_seen.clear();
  return RT_OK;
}
int rt_profile_select(rt_ctx *ctx, const char *kernels) {
  if (!ctx) return RT_ERR_INVALID;
  ctx->profile_only = (kernels && *kernels) ? "," + std::string(kernels) + "," : "";
  return RT_OK;
}
int rt_profile_reset(rt_ctx *ctx) {
  if (!ctx) return RT_ERR_INVALID;
  RT_TRY(rt_synchronize(ctx));
  for (auto &kv : ctx->events) for (auto &ev : kv.second) { ctx->event_pool.push_back(ev.first); ctx->event_pool.push_back(ev.second); }
  ctx->events.clear();
  return RT_OK;
}
int rt_profile_get(rt_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches) {
  if (!ctx || !kernel) return RT_ERR_INVALID;
  RT_TRY(rt_synchronize(ctx));
  double tot = 0; uint64_t n = 0;
  auto it = ctx->events.find(kernel);
  if (it != ctx->events.end())
    for (auto &ev : it->second) { float ms = 0; if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { tot += ms; n++; } }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return RT_OK;
}

// ------------------------------------------------------------------ scene
// fp32 scan record of one primitive (layout: rt_geom.h), computed in fp64 from the boundary's geometry slots
static void rt_fast_record(int type, int flags, const double *g, float *f) {
  const double INF = __builtin_huge_val();
  auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  auto cross = [](const double *a, const double *b, double *r) {
    r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0];
  };
  double r[12] = { 0 };
  switch (type) {
    case RT_PLANE: r[0] = g[3]; r[1] = g[4]; r[2] = g[5]; r[3] = dot(g, g + 3); break;
    case RT_SPHERE: r[0] = g[0]; r[1] = g[1]; r[2] = g[2]; r[3] = g[6] * g[6]; break;
    case RT_HEMISPHERE:
      r[0] = g[0]; r[1] = g[1]; r[2] = g[2]; r[3] = g[6] * g[6]; r[4] = g[3]; r[5] = g[4]; r[6] = g[5];
      r[7] = g[7]; r[8] = (flags & RT_FLAG_TRUNC_MAX) ? g[8] : INF; break;
    case RT_CYLINDER:
      r[0] = g[0]; r[1] = g[1]; r[2] = g[2]; r[3] = g[6] * g[6]; r[4] = g[3]; r[5] = g[4]; r[6] = g[5];
      r[7] = (flags & RT_FLAG_HALF_HEIGHT) ? g[7] : INF; break;
    case RT_TRIANGLE: case RT_SQUARE: {
      const double *A = g, *AB = g + 3, *AC = g + 6;
      double N[3], n2[3], u[3], v[3];
      cross(AC, AB, N); cross(AB, AC, n2);
      cross(AC, n2, u); cross(n2, AB, v);
      const double su = dot(AB, u), sv = dot(AC, v);
      for (int k = 0; k < 3; k++) { r[k] = N[k]; r[4 + k] = u[k] / su; r[8 + k] = v[k] / sv; }
      r[3] = dot(A, N); r[7] = dot(A, r + 4); r[11] = dot(A, r + 8); break;
    }
    default:                                                  // disc, annulus
      r[0] = g[3]; r[1] = g[4]; r[2] = g[5]; r[3] = dot(g, g + 3); r[4] = g[0]; r[5] = g[1]; r[6] = g[2];
      r[7] = g[6] * g[6]; r[8] = (type == RT_ANNULUS) ? g[7] * g[7] : 0.0; break;
  }
  for (int k = 0; k < 12; k++) f[k] = (float)r[k];
}

// ------------------------------------------------------------------ hierarchy for large scenes
// Axis-aligned box of a primitive from the boundary's geometry slots; false for unbounded primitives (planes,
// infinite cylinders), which stay outside the hierarchy.
static bool rt_prim_box(int type, int flags, const double *g, double lo[3], double hi[3]) {
  auto set = [&](const double c[3], const double e[3]) { for (int k = 0; k < 3; k++) { lo[k] = c[k] - e[k]; hi[k] = c[k] + e[k]; } };
  switch (type) {
    case RT_SPHERE: case RT_HEMISPHERE: { double e[3] = { g[6], g[6], g[6] }; set(g, e); return true; }
    case RT_CYLINDER: {
      if (!(flags & RT_FLAG_HALF_HEIGHT) || (flags & RT_FLAG_FIRST_ROOT)) return false;
      const double n = sqrt(g[3] * g[3] + g[4] * g[4] + g[5] * g[5]);
      double e[3];
      for (int k = 0; k < 3; k++) { double a = n > 0 ? g[3 + k] / n : 0; double s2 = 1 - a * a; e[k] = g[7] * fabs(a) + g[6] * sqrt(s2 > 0 ? s2 : 0); }
      set(g, e); return true;
    }
    case RT_TRIANGLE: case RT_SQUARE: {
      for (int k = 0; k < 3; k++) {
        double p0 = g[k], p1 = g[k] + g[3 + k], p2 = g[k] + g[6 + k], p3 = (type == RT_SQUARE) ? g[k] + g[3 + k] + g[6 + k] : p0;
        lo[k] = fmin(fmin(p0, p1), fmin(p2, p3)); hi[k] = fmax(fmax(p0, p1), fmax(p2, p3));
      }
      return true;
    }
    case RT_DISC: case RT_ANNULUS: {
      const double n = sqrt(g[3] * g[3] + g[4] * g[4] + g[5] * g[5]);
      double e[3];
      for (int k = 0; k < 3; k++) { double a = n > 0 ? g[3 + k] / n : 0; double s2 = 1 - a * a; e[k] = g[6] * sqrt(s2 > 0 ? s2 : 0); }
      set(g, e); return true;
    }
    default: return false;                                  // plane
  }
}

// ------------------------------------------------------------------ scan groups (fp32 all-primitives scans)
// Primitives that lie close together -- the faces of a box, a lamp's shells, rim and disc, a tube's walls and ends --
// are put into a group with one bounding sphere; a scan skips the groups that no ray of its wave reaches.  gid[i]:
// group of shape i (0: none -- unbounded, alone and cheap, or too spread out), spheres: 4 floats per group 0 .. n_groups.
static double rt_prim_extent(int type, const double *g, const double X[3]) {      // farthest point of a bounded primitive from X
  auto dist = [&](const double *p) { return sqrt((p[0] - X[0]) * (p[0] - X[0]) + (p[1] - X[1]) * (p[1] - X[1]) + (p[2] - X[2]) * (p[2] - X[2])); };
  switch (type) {
    case RT_SPHERE: case RT_HEMISPHERE: return dist(g) + fabs(g[6]);
    case RT_DISC: case RT_ANNULUS: case RT_CYLINDER: {      // rim point(s) on the far side of the axis
      const double nl = sqrt(g[3] * g[3] + g[4] * g[4] + g[5] * g[5]);
      if (!(nl > 0)) return dist(g) + fabs(g[6]) + (type == RT_CYLINDER ? fabs(g[7]) : 0);
      const double v[3] = { g[0] - X[0], g[1] - X[1], g[2] - X[2] };
      const double h = (v[0] * g[3] + v[1] * g[4] + v[2] * g[5]) / nl;
      const double rho2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] - h * h;
      const double along = fabs(h) + (type == RT_CYLINDER ? fabs(g[7]) : 0), across = sqrt(rho2 > 0 ? rho2 : 0) + fabs(g[6]);
      return sqrt(along * along + across * across);
    }
    default: {                                               // triangle, square: the corners
      double m = 0;
      for (int c = 0; c < 4; c++) {
        if (c == 3 && type != RT_SQUARE) break;
        double p[3];
        for (int k = 0; k < 3; k++) p[k] = g[k] + ((c & 1) ? g[3 + k] : 0) + ((c & 2) ? g[6 + k] : 0);
        m = fmax(m, dist(p));
      }
      return m;
    }
  }
}
static uint32_t rt_scan_groups(const rt_scene_desc *d, std::vector<uint8_t> &gid, std::vector<float> &spheres) {
  static const int cost[RT_NUM_TYPES] = { 8, 20, 30, 45, 37, 37, 22, 24 };     // instructions of one test, roughly
  const uint32_t n = d->n_shapes;
  gid.assign(n, 0); spheres.assign(4, 0.f);
  struct Cluster { std::vector<uint32_t> members; double lo[3], hi[3], c[3], r; int cost; };
  std::vector<Cluster> cl;
  for (uint32_t i = 0; i < n; i++) {
    const int type = d->shape_meta[4 * i], flags = d->shape_meta[4 * i + 2];
    Cluster k; k.members.push_back(i); k.cost = cost[type];
    if (!rt_prim_box(type, flags, d->shape_geom + (size_t)i * 12, k.lo, k.hi)) continue;
    bool finite = true;
    for (int a = 0; a < 3; a++) finite = finite && (k.lo[a] - k.lo[a]) == 0 && (k.hi[a] - k.hi[a]) == 0;
    if (!finite) continue;
    cl.push_back(k);
  }
  if (cl.empty() || cl.size() > 96) return 0;                // (larger scenes: the pairwise merging below is quartic)
  auto bound = [&](Cluster &k) {                             // sphere around the centre of the members' boxes
    for (int a = 0; a < 3; a++) k.c[a] = 0.5 * (k.lo[a] + k.hi[a]);
    k.r = 0;
    for (uint32_t i : k.members) k.r = fmax(k.r, rt_prim_extent(d->shape_meta[4 * i], d->shape_geom + (size_t)i * 12, k.c));
  };
  for (Cluster &k : cl) bound(k);
  // merge the pair whose common sphere is the smallest multiple of the larger of its two spheres, while that is <= 1.2
  for (;;) {
    double best = 1.2; size_t ba = 0, bb = 0; Cluster bm;
    for (size_t a = 0; a < cl.size(); a++)
      for (size_t b = a + 1; b < cl.size(); b++) {
        Cluster m; m.members = cl[a].members; m.members.insert(m.members.end(), cl[b].members.begin(), cl[b].members.end());
        for (int k = 0; k < 3; k++) { m.lo[k] = fmin(cl[a].lo[k], cl[b].lo[k]); m.hi[k] = fmax(cl[a].hi[k], cl[b].hi[k]); }
        m.cost = cl[a].cost + cl[b].cost;
        bound(m);
        const double ratio = m.r / fmax(fmax(cl[a].r, cl[b].r), 1e-300);
        if (ratio <= best) { best = ratio; ba = a; bb = b; bm = m; }
      }
    if (ba == bb) break;
    cl[ba] = bm; cl.erase(cl.begin() + (long)bb);
  }
  // a group's sphere test costs about 16 instructions per scan: only clusters worth several tests become groups
  std::vector<size_t> keep;
  for (size_t k = 0; k < cl.size(); k++) if (cl[k].cost >= 45) keep.push_back(k);
  std::sort(keep.begin(), keep.end(), [&](size_t a, size_t b) { return cl[a].cost != cl[b].cost ? cl[a].cost > cl[b].cost : a < b; });
  if (keep.size() > RT_MAX_GROUPS) keep.resize(RT_MAX_GROUPS);
  std::sort(keep.begin(), keep.end());
  uint32_t ng = 0;
  for (size_t k : keep) {
    ng++;
    for (uint32_t i : cl[k].members) gid[i] = (uint8_t)ng;
    for (int a = 0; a < 3; a++) spheres.push_back((float)cl[k].c[a]);
    spheres.push_back((float)(cl[k].r * 1.002 + 4e-3));     // padded: the skip must be conservative in fp32
  }
  return ng;
}

struct BvhBuild {
  std::vector<BvhNode> nodes;
  std::vector<int8_t> split;                                // per node: axis whose lower centroids went to the first child, -1: none
  std::vector<uint32_t> prims;                              // leaf entries
  std::vector<double> lo, hi;                               // per bounded item
  std::vector<uint32_t> entry, order;                       // per bounded item: encoded entry; working permutation
  static float down(double v) { float f = (float)v; return ((double)f > v) ? nextafterf(f, -INFINITY) : f; }
  static float up(double v) { float f = (float)v; return ((double)f < v) ? nextafterf(f, INFINITY) : f; }
  // nodes in depth-first order; returns the node's index
  uint32_t build(uint32_t b, uint32_t e) {
    double nlo[3] = { INFINITY, INFINITY, INFINITY }, nhi[3] = { -INFINITY, -INFINITY, -INFINITY };
    double clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = b; i < e; i++) {
      const uint32_t it = order[i];
      for (int k = 0; k < 3; k++) {
        nlo[k] = fmin(nlo[k], lo[3 * it + k]); nhi[k] = fmax(nhi[k], hi[3 * it + k]);
        const double c = 0.5 * (lo[3 * it + k] + hi[3 * it + k]);
        clo[k] = fmin(clo[k], c); chi[k] = fmax(chi[k], c);
      }
    }
    const uint32_t me = (uint32_t)nodes.size();
    nodes.push_back(BvhNode()); split.push_back(-1);
    for (int k = 0; k < 3; k++) {
      // padding: the boxes only steer the walk, a box that is too large costs tests, one that is too small loses hits
      const double pad = 1e-4 * (nhi[k] - nlo[k]) + 1e-4;
      nodes[me].lo[k] = down(nlo[k] - pad); nodes[me].hi[k] = up(nhi[k] + pad);
    }
    int axis = 0;
    for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
    if (e - b <= 4 || !(chi[axis] - clo[axis] > 0)) {
      if (e - b > 255) {                                    // many items with one centre: chain of leaves under a split by position
        const uint32_t mid = b + (e - b) / 2;
        nodes[me].leaf = 0;
        build(b, mid); build(mid, e);
        nodes[me].skip = (uint32_t)nodes.size();
        return me;
      }
      nodes[me].leaf = (uint32_t)prims.size() | ((e - b) << 24);
      for (uint32_t i = b; i < e; i++) prims.push_back(entry[order[i]]);
      nodes[me].skip = (uint32_t)nodes.size();
      return me;
    }
    // Split: surface-area heuristic over 16 centroid bins per axis (cost = area x count of either side); the median
    // of the widest axis when the bins cannot separate the items.  The boxes only steer the walk, so any split is exact.
    uint32_t mid = b + (e - b) / 2;
    int used_axis = -1;
    {
      constexpr int NB = 16;
      double best_cost = INFINITY; int best_axis = -1, best_bin = -1;
      for (int ax = 0; ax < 3; ax++) {
        const double c0 = clo[ax], span = chi[ax] - clo[ax];
        if (!(span > 0)) continue;
        double blo[NB][3], bhi[NB][3]; uint32_t cnt[NB] = { 0 };
        for (int i = 0; i < NB; i++) for (int k = 0; k < 3; k++) { blo[i][k] = INFINITY; bhi[i][k] = -INFINITY; }
        auto bin_of = [&](uint32_t it) {
          const double c = 0.5 * (lo[3 * it + ax] + hi[3 * it + ax]);
          int bi = (int)((c - c0) / span * NB);
          return bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
        };
        for (uint32_t i = b; i < e; i++) {
          const uint32_t it = order[i]; const int bi = bin_of(it);
          cnt[bi]++;
          for (int k = 0; k < 3; k++) { blo[bi][k] = fmin(blo[bi][k], lo[3 * it + k]); bhi[bi][k] = fmax(bhi[bi][k], hi[3 * it + k]); }
        }
        auto area = [](const double *l, const double *h) {
          const double dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
          return (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.0;
        };
        double right_area[NB]; uint32_t right_cnt[NB];
        { double l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY }; uint32_t c = 0;
          for (int i = NB - 1; i > 0; i--) { for (int k = 0; k < 3; k++) { l[k] = fmin(l[k], blo[i][k]); h[k] = fmax(h[k], bhi[i][k]); } c += cnt[i]; right_area[i] = area(l, h); right_cnt[i] = c; } }
        double l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY }; uint32_t c = 0;
        for (int i = 0; i < NB - 1; i++) {                    // split between bins i and i + 1
          for (int k = 0; k < 3; k++) { l[k] = fmin(l[k], blo[i][k]); h[k] = fmax(h[k], bhi[i][k]); }
          c += cnt[i];
          if (c == 0 || right_cnt[i + 1] == 0) continue;
          const double cost = area(l, h) * c + right_area[i + 1] * right_cnt[i + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = i; }
        }
      }
      if (best_axis >= 0) {
        const int ax = best_axis; const double c0 = clo[ax], span = chi[ax] - clo[ax];
        auto left = [&](uint32_t it) {
          const double c = 0.5 * (lo[3 * it + ax] + hi[3 * it + ax]);
          int bi = (int)((c - c0) / span * 16);
          bi = bi < 0 ? 0 : (bi >= 16 ? 15 : bi);
          return bi <= best_bin;
        };
        mid = (uint32_t)(std::stable_partition(order.begin() + b, order.begin() + e, left) - order.begin());
        used_axis = ax;
      }
      const uint32_t smaller = (mid - b < e - mid) ? mid - b : e - mid;
      if (best_axis < 0 || mid == b || mid == e || (uint64_t)smaller * 16 < (uint64_t)(e - b)) {       // (bounds the depth of the tree)
        mid = b + (e - b) / 2;
        std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](uint32_t x, uint32_t y) {
          const double cx = lo[3 * x + axis] + hi[3 * x + axis], cy = lo[3 * y + axis] + hi[3 * y + axis];
          return cx < cy || (cx == cy && x < y);
        });
        used_axis = axis;
      }
    }
    nodes[me].leaf = 0; split[me] = (int8_t)used_axis;
    build(b, mid); build(mid, e);
    nodes[me].skip = (uint32_t)nodes.size();
    return me;
  }
  // The subtree of node `me` appended to `out` in depth-first order with, at every split, the child first that a ray of
  // direction octant `oct` (bit k: d[k] < 0) meets first along the split axis.
  void reorder(uint32_t me, unsigned oct, std::vector<BvhNode> &out, size_t base) const {
    const size_t at = out.size();
    out.push_back(nodes[me]);
    if (!nodes[me].leaf) {
      uint32_t first = me + 1, second = nodes[me + 1].skip;
      if (split[me] >= 0 && ((oct >> split[me]) & 1u)) std::swap(first, second);
      reorder(first, oct, out, base); reorder(second, oct, out, base);
    }
    out[at].skip = (uint32_t)(out.size() - base);        // indices count from the copy's own first node
  }
};

// Scenes whose type-sorted copy does not fit the scan kernels' LDS budget (or every scene, with option bvh = 1, for
// tests) get a hierarchy; `order` maps sorted slot -> original index.
static bool rt_wants_bvh(const rt_ctx *ctx, uint32_t n) {
  const int knob = ctx->knobs.bvh;
  // (the same budget the fp32 scan kernels apply when they decide whether to stage the scene in LDS: no scene falls between the two)
  return n > 0 && (knob >= 0 ? knob != 0 : scan_lds_bytes<float>(n) > 56 * 1024);
}
static int rt_build_bvh(rt_ctx *ctx, const rt_scene_desc *d, const std::vector<uint32_t> &order, rt_scene *sc, bool &built) {
  built = false;
  const uint32_t n = d->n_shapes;
  if (!rt_wants_bvh(ctx, n)) return RT_OK;
  if (n >= (1u << 24)) return rt_fail(ctx, RT_ERR_TOO_LARGE, "hierarchy supports up to 2^24 primitives");
  BvhBuild B;
  std::vector<uint32_t> unbounded;
  for (uint32_t s = 0; s < n; s++) {
    const uint32_t i = order[s];
    const int type = d->shape_meta[4 * i], flags = d->shape_meta[4 * i + 2];
    const uint32_t entry = s | ((uint32_t)type << 28);
    double lo[3], hi[3];
    if (rt_prim_box(type, flags, d->shape_geom + (size_t)i * 12, lo, hi) && lo[0] == lo[0] && hi[0] == hi[0]) {
      for (int k = 0; k < 3; k++) { B.lo.push_back(lo[k]); B.hi.push_back(hi[k]); }
      B.entry.push_back(entry);
    } else unbounded.push_back(entry);
  }
  B.order.resize(B.entry.size());
  for (uint32_t i = 0; i < B.order.size(); i++) B.order[i] = i;
  if (!B.order.empty()) B.build(0, (uint32_t)B.order.size());
  auto up = [&](DevMem &m, const void *src, size_t bytes) -> int {
    RT_HIP(ctx, m.alloc(bytes)); if (bytes) RT_HIP(ctx, hipMemcpy(m.p, src, bytes, hipMemcpyHostToDevice)); return RT_OK;
  };
  // Eight copies of the node array, one per octant of ray directions, each threaded near child first: a ray finds its
  // closest hit early and the boxes behind it fail the reach test.  Any order gives the same hit (ties go by index).
  const uint32_t n_nodes = (uint32_t)B.nodes.size();
  const bool octants = !ctx->knobs.bvh_one_order && n_nodes > 1;
  if (octants) {
    std::vector<BvhNode> all; all.reserve((size_t)n_nodes * 8);
    for (unsigned oct = 0; oct < 8; oct++) B.reorder(0, oct, all, all.size());
    if (all.size() != (size_t)n_nodes * 8) return rt_fail(ctx, RT_ERR_INVALID, "hierarchy reordering lost nodes");
    B.nodes.swap(all);
  }
  RT_TRY(up(sc->bvh_nodes, B.nodes.data(), B.nodes.size() * sizeof(BvhNode)));
  RT_TRY(up(sc->bvh_prims, B.prims.data(), B.prims.size() * 4));
  RT_TRY(up(sc->bvh_unbounded, unbounded.data(), unbounded.size() * 4));
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 0) { sc->d32.bvh = sc->bvh_nodes.as<BvhNode>(); sc->d32.bvh_prims = sc->bvh_prims.as<uint32_t>(); sc->d32.unbounded = sc->bvh_unbounded.as<uint32_t>();
                     sc->d32.n_bvh_nodes = n_nodes; sc->d32.bvh_octant_stride = octants ? n_nodes : 0u; sc->d32.n_unbounded = (uint32_t)unbounded.size(); }
    else { sc->d64.bvh = sc->bvh_nodes.as<BvhNode>(); sc->d64.bvh_prims = sc->bvh_prims.as<uint32_t>(); sc->d64.unbounded = sc->bvh_unbounded.as<uint32_t>();
           sc->d64.n_bvh_nodes = n_nodes; sc->d64.bvh_octant_stride = octants ? n_nodes : 0u; sc->d64.n_unbounded = (uint32_t)unbounded.size(); }
  }
  built = true;
  return RT_OK;
}

template <typename T> static int upload_typed(rt_ctx *ctx, const rt_scene_desc *d, const std::vector<uint32_t> &order, DevMem &geom,
                                              DevMem &colour, DevMem &sgeom, DevMem &surface, DevMem &hit, DevScene<T> &S, rt_scene *sc,
                                              const std::vector<uint32_t> &spec_gt) {
  const uint32_t n = d->n_shapes;
  std::vector<T> g((size_t)n * 12), c((size_t)n * 4), sg((size_t)n * 12);
  for (size_t i = 0; i < (size_t)n * 12; i++) g[i] = (T)d->shape_geom[i];
  for (size_t i = 0; i < (size_t)n * 4; i++) c[i] = (T)d->shape_colour[i];
  for (uint32_t s = 0; s < n; s++) {
    const uint32_t i = order[s];
    if (sizeof(T) == 4)           // the fp32 build scans precomputed records instead of the raw geometry slots
      rt_fast_record(d->shape_meta[4 * i], d->shape_meta[4 * i + 2], d->shape_geom + (size_t)i * 12, (float *)&sg[(size_t)s * 12]);
    else
      for (int k = 0; k < 12; k++) sg[(size_t)s * 12 + k] = g[(size_t)i * 12 + k];
  }
  RT_HIP(ctx, geom.alloc(g.size() * sizeof(T))); RT_HIP(ctx, colour.alloc(c.size() * sizeof(T))); RT_HIP(ctx, sgeom.alloc(sg.size() * sizeof(T)));
  RT_HIP(ctx, hipMemcpy(geom.p, g.data(), g.size() * sizeof(T), hipMemcpyHostToDevice));
  RT_HIP(ctx, hipMemcpy(colour.p, c.data(), c.size() * sizeof(T), hipMemcpyHostToDevice));
  RT_HIP(ctx, hipMemcpy(sgeom.p, sg.data(), sg.size() * sizeof(T), hipMemcpyHostToDevice));
  memset(&S, 0, sizeof S);
  S.n_shapes = n; S.n_lights = d->n_lights;
  S.meta = sc->meta.as<uint8_t>(); S.geom = geom.as<T>(); S.colour = colour.as<T>(); S.sgeom = sgeom.as<T>();
  S.sorig = sc->sorig.as<uint32_t>(); S.sflags32 = sc->sflags.as<uint32_t>();
  for (uint32_t l = 0; l < d->n_lights; l++) {
    S.light_shape[l] = d->light_shape[l];
    for (int k = 0; k < 8; k++) S.light_data[l][k] = (T)d->light_data[(size_t)l * 8 + k];
  }
  for (int k = 0; k < 16; k++) S.camera[k] = (T)d->camera[k];
  S.width = d->width; S.height = d->height;
  for (int r = 0; r < 4; r++) for (int k = 0; k < 8; k++) S.mat[r][k] = (T)d->materials[r][k];
  for (int k = 0; k < 8; k++) { S.refl_k[k] = rt_threshold_u24(d->materials[2][k]); S.spec_k[k] = rt_threshold_u24(d->materials[3][k]); }
  S.surface = nullptr; S.spec_gt_k = sc->spec_gt.as<uint32_t>();
  if (d->shape_surface) {
    std::vector<T> sf((size_t)n * 2);
    for (size_t i = 0; i < (size_t)n * 2; i++) sf[i] = (T)d->shape_surface[i];
    RT_HIP(ctx, surface.alloc(sf.size() * sizeof(T)));
    RT_HIP(ctx, hipMemcpy(surface.p, sf.data(), sf.size() * sizeof(T), hipMemcpyHostToDevice));
    S.surface = surface.as<T>();
  }
  // one record per shape with everything shading reads about it (see HitRec)
  std::vector<HitRec<T>> hr(n);
  const T nan = std::numeric_limits<T>::quiet_NaN();
  for (uint32_t i = 0; i < n; i++) {
    HitRec<T> &h = hr[i];
    memset(&h, 0, sizeof h);
    for (int k = 0; k < 12; k++) h.geom[k] = g[(size_t)i * 12 + k];
    for (int k = 0; k < 4; k++) h.colour[k] = c[(size_t)i * 4 + k];
    const uint8_t *m = d->shape_meta + 4 * (size_t)i;
    const bool has_mat = m[1] < 8;
    const int mi = m[1] & 7;
    h.transparent = has_mat ? S.mat[0][mi] : nan; h.refr_index = has_mat ? S.mat[1][mi] : nan;
    h.reflectance = has_mat ? S.mat[2][mi] : nan; h.specular = has_mat ? S.mat[3][mi] : nan;
    h.refl_k = has_mat ? S.refl_k[mi] : 0u; h.spec_k = has_mat ? S.spec_k[mi] : 0u;
    h.meta = (uint32_t)m[0] | ((uint32_t)m[1] << 8) | ((uint32_t)m[2] << 16) | ((uint32_t)m[3] << 24);
    h.spec_gt_k = spec_gt[i];
    h.surf_refl = d->shape_surface ? (T)d->shape_surface[2 * (size_t)i] : nan;
  }
  RT_HIP(ctx, hit.alloc(hr.size() * sizeof(HitRec<T>)));
  RT_HIP(ctx, hipMemcpy(hit.p, hr.data(), hr.size() * sizeof(HitRec<T>), hipMemcpyHostToDevice));
  S.hit = hit.as<HitRec<T>>();
  return RT_OK;
}

int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *d, rt_scene **out) {
  if (!ctx || !d || !out) return RT_ERR_INVALID;
  *out = nullptr;
  if (d->n_shapes == 0 || !d->shape_meta || !d->shape_geom || !d->shape_colour) return rt_fail(ctx, RT_ERR_INVALID, "scene has no shapes");
  if (d->n_lights > RT_MAX_LIGHTS) return rt_fail(ctx, RT_ERR_TOO_LARGE, "at most %d lights", RT_MAX_LIGHTS);
  if (d->n_lights && (!d->light_shape || !d->light_data)) return rt_fail(ctx, RT_ERR_INVALID, "light arrays missing");
  if (d->width == 0 || d->height == 0) return rt_fail(ctx, RT_ERR_INVALID, "canvas size is zero");
  const uint32_t n = d->n_shapes;
  for (uint32_t i = 0; i < n; i++)
    if (d->shape_meta[4 * i] >= RT_NUM_TYPES)                               // rayphoton.js:639-642 throws
      return rt_fail(ctx, RT_ERR_UNKNOWN_SHAPE, "I didn't recognize the shape! (%d)", (int)d->shape_meta[4 * i]);
  for (uint32_t l = 0; l < d->n_lights; l++) {
    uint32_t s = d->light_shape[l];
    if (s >= n || d->shape_meta[4 * s] != RT_DISC || !(d->shape_meta[4 * s + 2] & RT_FLAG_LIGHT))
      return rt_fail(ctx, RT_ERR_INVALID, "light %u does not point at a light disc", l);
  }
  RT_HIP(ctx, hipSetDevice(ctx->device));
  rt_scene *sc = new rt_scene();
  sc->ctx = ctx;
  // type-sorted order, stable in the original index
  std::vector<uint32_t> order(n);
  for (uint32_t i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return d->shape_meta[4 * a] < d->shape_meta[4 * b]; });
  std::vector<uint32_t> sflags(n);
  uint32_t seg[RT_NUM_TYPES + 1] = { 0 };
  for (uint32_t s = 0; s < n; s++) { sflags[s] = d->shape_meta[4 * order[s] + 2]; seg[d->shape_meta[4 * order[s]] + 1]++; }
  for (int t = 0; t < RT_NUM_TYPES; t++) seg[t + 1] += seg[t];
  // `U > specular` (ray.js:910) on the integer draw: k * 2^-24 > s  <=>  k >= floor(s * 2^24) + 1; NaN never
  std::vector<uint32_t> spec_gt(n, (1u << 24) + 1);
  if (d->shape_surface)
    for (uint32_t i = 0; i < n; i++) {
      double sp = d->shape_surface[2 * (size_t)i + 1];
      if (sp == sp) spec_gt[i] = sp < 0 ? 0u : (sp >= 1 ? (1u << 24) + 1 : (uint32_t)floor(sp * 16777216.0) + 1);
    }
  int rc = RT_OK;
  auto up = [&](DevMem &m, const void *src, size_t bytes) -> int {
    RT_HIP(ctx, m.alloc(bytes)); RT_HIP(ctx, hipMemcpy(m.p, src, bytes, hipMemcpyHostToDevice)); return RT_OK;
  };
  // fp32 scan groups (see rt_scan_groups): spheres and, per group, the mask of its sorted slots; none with a hierarchy
  std::vector<uint8_t> gid(n, 0);
  std::vector<float> gspheres(4, 0.f);
  uint32_t n_groups = 0;
  if (!rt_wants_bvh(ctx, n) && ctx->knobs.scan_groups && n <= 64) n_groups = rt_scan_groups(d, gid, gspheres);    // (a scan keeps the slots to test in 64 bits)
  std::vector<uint64_t> gmask(n_groups + 1, 0);
  for (uint32_t s = 0; s < n && s < 64; s++) gmask[gid[order[s]]] |= 1ull << s;
  if (n < 64) gmask[0] |= ~0ull << n;
  if ((rc = up(sc->meta, d->shape_meta, (size_t)n * 4)) || (rc = up(sc->sorig, order.data(), (size_t)n * 4)) ||
      (rc = up(sc->sflags, sflags.data(), (size_t)n * 4)) ||
      (rc = up(sc->gsphere, gspheres.data(), gspheres.size() * 4)) || (rc = up(sc->gmask, gmask.data(), gmask.size() * 8)) ||
      (rc = up(sc->spec_gt, spec_gt.data(), (size_t)n * 4)) ||
      (rc = upload_typed<float>(ctx, d, order, sc->geom32, sc->colour32, sc->sgeom32, sc->surface32, sc->hit32, sc->d32, sc, spec_gt)) ||
      (rc = upload_typed<double>(ctx, d, order, sc->geom64, sc->colour64, sc->sgeom64, sc->surface64, sc->hit64, sc->d64, sc, spec_gt))) {
    delete sc; return rc;
  }
  sc->d32.n_groups = n_groups; sc->d32.gsphere = sc->gsphere.as<float4>(); sc->d32.gmask = sc->gmask.as<uint2>();
  bool has_bvh = false;
  if ((rc = rt_build_bvh(ctx, d, order, sc, has_bvh))) { delete sc; return rc; }
  {   // bounding spheres of the sorted primitives for the primary pass's per-workgroup culling (fp32 build)
    std::vector<float> bs((size_t)n * 4);
    for (uint32_t s = 0; s < n; s++) {
      const uint32_t i = order[s];
      double lo[3], hi[3];
      float *o4 = &bs[(size_t)s * 4];
      if (rt_prim_box(d->shape_meta[4 * i], d->shape_meta[4 * i + 2], d->shape_geom + (size_t)i * 12, lo, hi)) {
        double r2 = 0;
        for (int k = 0; k < 3; k++) { o4[k] = (float)(0.5 * (lo[k] + hi[k])); r2 += 0.25 * (hi[k] - lo[k]) * (hi[k] - lo[k]); }
        o4[3] = (float)(sqrt(r2) * 1.001 + 1e-3);
      } else { o4[0] = o4[1] = o4[2] = 0; o4[3] = INFINITY; }
    }
    if ((rc = up(sc->sbound, bs.data(), bs.size() * 4))) { delete sc; return rc; }
    sc->d32.sbound = sc->sbound.as<float4>();                 // (option no_cull switches the culling off per frame)
  }
  memcpy(sc->d32.seg, seg, sizeof seg); memcpy(sc->d64.seg, seg, sizeof seg);
  // fp32: what every test needs of the camera's position (all rays of a first pass start there), by the device's own code
  if (hipError_t e = sc->sot_cam.alloc((size_t)n * RT_OT_STRIDE * 4); e != hipSuccess) {
    delete sc; return rt_fail(ctx, RT_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  if ((rc = pl_camera_origin_terms(ctx, sc->d32, sc->sot_cam.as<float>()))) { delete sc; return rc; }
  sc->d32.sot_cam = sc->sot_cam.as<float>();
  *out = sc;
  return RT_OK;
}
void rt_scene_free(rt_scene *sc) { if (sc) { (void)hipSetDevice(sc->ctx->device); delete sc; } }

// ------------------------------------------------------------------ geometry kernel surfaces
int rt_intersect_pairs(rt_ctx *ctx, const rt_params *p, uint32_t n, const double *ray_o, const double *ray_d, int normalize_dir,
                       const uint8_t *shape_meta, const double *shape_geom, double *t_out) {
  if (!ctx || !p || !t_out) return RT_ERR_INVALID;
  if (n == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem to, td, tm, tg, tt; const void *o, *d, *m, *g; void *t;
  RT_TRY(rt_stage_in(ctx, ray_o, (size_t)n * 24, to, &o)); RT_TRY(rt_stage_in(ctx, ray_d, (size_t)n * 24, td, &d));
  RT_TRY(rt_stage_in(ctx, shape_meta, (size_t)n * 4, tm, &m)); RT_TRY(rt_stage_in(ctx, shape_geom, (size_t)n * 96, tg, &g));
  RT_TRY(rt_stage_out_begin(ctx, t_out, (size_t)n * 8, tt, &t));
  if (p->precision == RT_PRECISION_F64)
    RT_TRY(pl_intersect_pairs<double>(ctx, p, n, (const double *)o, (const double *)d, normalize_dir, (const uint8_t *)m, (const double *)g, (double *)t));
  else {
    // The fp32 product path: every pair goes through the scan record the upload would precompute for the shape
    // (rt_fast_record) and the distance functions of the fp32 scan kernels.
    std::vector<uint8_t> hm((size_t)n * 4); std::vector<double> hg((size_t)n * 12); std::vector<float> hr((size_t)n * 12);
    RT_HIP(ctx, hipMemcpy(hm.data(), shape_meta, hm.size(), hipMemcpyDefault));
    RT_HIP(ctx, hipMemcpy(hg.data(), shape_geom, hg.size() * 8, hipMemcpyDefault));
    for (uint32_t i = 0; i < n; i++) {
      if (hm[4 * (size_t)i] >= RT_NUM_TYPES) return rt_fail(ctx, RT_ERR_UNKNOWN_SHAPE, "I didn't recognize the shape! (%d)", (int)hm[4 * (size_t)i]);
      rt_fast_record(hm[4 * (size_t)i], hm[4 * (size_t)i + 2], &hg[(size_t)i * 12], &hr[(size_t)i * 12]);
    }
    DevMem trec;
    RT_HIP(ctx, trec.alloc(hr.size() * 4));
    RT_HIP(ctx, hipMemcpyAsync(trec.p, hr.data(), hr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    RT_TRY(pl_intersect_pairs_fast(ctx, p, n, (const double *)o, (const double *)d, normalize_dir, (const uint8_t *)m, trec.as<float>(), (double *)t));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (trec and hr go out of scope)
  }
  return rt_stage_out_end(ctx, t_out, (size_t)n * 8, tt);
}

int rt_intersect(rt_scene *sc, const rt_params *p, int mode, uint32_t n, const double *ray_o, const double *ray_d,
                 const double *source_dist, double *t_out, int32_t *idx_out) {
  if (!sc || !p || !idx_out) return RT_ERR_INVALID;
  rt_ctx *ctx = sc->ctx;
  if (mode < 0 || mode > 2) return rt_fail(ctx, RT_ERR_INVALID, "unknown scan mode %d", mode);
  if (mode == RT_SCAN_ANY && !source_dist) return rt_fail(ctx, RT_ERR_INVALID, "any-hit scan needs source_dist");
  if (n == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem to, td, ts, tt, ti; const void *o, *d, *s = nullptr; void *t, *i;
  RT_TRY(rt_stage_in(ctx, ray_o, (size_t)n * 24, to, &o)); RT_TRY(rt_stage_in(ctx, ray_d, (size_t)n * 24, td, &d));
  if (source_dist) RT_TRY(rt_stage_in(ctx, source_dist, (size_t)n * 8, ts, &s));
  RT_TRY(rt_stage_out_begin(ctx, t_out, (size_t)n * 8, tt, &t)); RT_TRY(rt_stage_out_begin(ctx, idx_out, (size_t)n * 4, ti, &i));
  if (p->precision == RT_PRECISION_F64)
    RT_TRY(pl_intersect_scene<double>(sc, p, mode, n, (const double *)o, (const double *)d, (const double *)s, (double *)t, (int32_t *)i));
  else
    RT_TRY(pl_intersect_scene<float>(sc, p, mode, n, (const double *)o, (const double *)d, (const double *)s, (double *)t, (int32_t *)i));
  RT_TRY(rt_stage_out_end(ctx, t_out, (size_t)n * 8, tt));
  return rt_stage_out_end(ctx, idx_out, (size_t)n * 4, ti);
}

// ------------------------------------------------------------------ explicit rays, points, photons
// stacks as the callers write them (16 bytes a ray: the length, then the materials from the bottom) -> MatStack words
static int pack_stacks(rt_ctx *ctx, const uint8_t *stacks, uint32_t n, std::vector<uint64_t> &out) {
  out.resize(n);
  for (uint32_t i = 0; i < n; i++) {
    const uint8_t *s = stacks + (size_t)i * 16;
    if (s[0] < 1 || s[0] > 15) return rt_fail(ctx, RT_ERR_INVALID, "material stack %u: length %u outside 1..15", i, (unsigned)s[0]);
    uint64_t bits = (uint64_t)s[0] << 56;
    for (int k = 0; k < s[0]; k++) {
      if (s[1 + k] > 7) return rt_fail(ctx, RT_ERR_INVALID, "material stack %u: material %u outside 0..7", i, (unsigned)s[1 + k]);
      bits |= (uint64_t)s[1 + k] << (3 * k);
    }
    out[i] = bits;
  }
  return RT_OK;
}

int rt_trace_rays(rt_scene *sc, const rt_params *p, rt_photon_map *diffuse, rt_photon_map *caustic, uint32_t n, const double *ray_o,
                  const double *ray_d, const double *max_dist, const int32_t *depth, const double *importance, const uint8_t *stacks,
                  uint32_t first_stream, const uint32_t *first_draw, double *rgba_out, rt_trace_stats *stats) {
  if (!sc || !p) return RT_ERR_INVALID;
  rt_ctx *ctx = sc->ctx;
  if (!ray_o || !ray_d || !rgba_out) return rt_fail(ctx, RT_ERR_INVALID, "rt_trace_rays: ray_o, ray_d and rgba_out are required");
  if (p->mode != RT_MODE_PHOTON) return rt_fail(ctx, RT_ERR_INVALID, "rt_trace_rays follows rayphoton.js's traceRay only (mode RT_MODE_PHOTON)");
  if ((diffuse && diffuse->ctx != ctx) || (caustic && caustic->ctx != ctx)) return rt_fail(ctx, RT_ERR_INVALID, "photon map belongs to another context");
  if ((uint64_t)first_stream + n > 0x7FFFFFFFull) return rt_fail(ctx, RT_ERR_INVALID, "rt_trace_rays: first_stream + n must stay below 2^31");
  if (n == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_TRY(rt_synchronize(ctx));                                // (shares the first frame slot's counters)
  DevMem to, td, tm, tp, ti, ts, tf, tr;
  const void *o, *d, *md_ = nullptr, *dp = nullptr, *im = nullptr, *st = nullptr, *fd = nullptr; void *out;
  RT_TRY(rt_stage_in(ctx, ray_o, (size_t)n * 24, to, &o)); RT_TRY(rt_stage_in(ctx, ray_d, (size_t)n * 24, td, &d));
  if (max_dist) RT_TRY(rt_stage_in(ctx, max_dist, (size_t)n * 8, tm, &md_));
  if (depth) RT_TRY(rt_stage_in(ctx, depth, (size_t)n * 4, tp, &dp));
  if (importance) RT_TRY(rt_stage_in(ctx, importance, (size_t)n * 8, ti, &im));
  std::vector<uint64_t> packed;
  if (stacks) {
    if (rt_is_device_ptr(stacks)) return rt_fail(ctx, RT_ERR_INVALID, "rt_trace_rays: stacks must be host memory");
    RT_TRY(pack_stacks(ctx, stacks, n, packed));
    RT_TRY(rt_stage_in(ctx, packed.data(), (size_t)n * 8, ts, &st));
  }
  if (first_draw) RT_TRY(rt_stage_in(ctx, first_draw, (size_t)n * 4, tf, &fd));
  RT_TRY(rt_stage_out_begin(ctx, rgba_out, (size_t)n * 32, tr, &out));
  DevMap md = diffuse ? diffuse->view() : DevMap{}, mc = caustic ? caustic->view() : DevMap{};
  if (p->precision == RT_PRECISION_F64)
    RT_TRY(pl_trace_rays<double>(sc, p, md, mc, n, (const double *)o, (const double *)d, (const double *)md_, (const int32_t *)dp, (const double *)im,
                                 (const uint64_t *)st, first_stream, (const uint32_t *)fd, (double *)out, stats));
  else
    RT_TRY(pl_trace_rays<float>(sc, p, md, mc, n, (const double *)o, (const double *)d, (const double *)md_, (const int32_t *)dp, (const double *)im,
                                (const uint64_t *)st, first_stream, (const uint32_t *)fd, (double *)out, stats));
  return rt_stage_out_end(ctx, rgba_out, (size_t)n * 32, tr);
}

int rt_direct_light(rt_scene *sc, const rt_params *p, uint32_t n, const double *points, const double *normals, const double *local_rgb,
                    uint32_t first_stream, const uint32_t *first_draw, double *rgb_out, uint32_t *draws_out) {
  if (!sc || !p) return RT_ERR_INVALID;
  rt_ctx *ctx = sc->ctx;
  if (!points || !normals || !local_rgb || !rgb_out) return rt_fail(ctx, RT_ERR_INVALID, "rt_direct_light: points, normals, local_rgb and rgb_out are required");
  if (sc->d32.n_lights == 0) return rt_fail(ctx, RT_ERR_INVALID, "rt_direct_light: the scene has no lights");
  if (n == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem tp, tn, tl, tf, tr, tw; const void *pt, *nr, *lc, *fd = nullptr; void *out, *dr = nullptr;
  RT_TRY(rt_stage_in(ctx, points, (size_t)n * 24, tp, &pt)); RT_TRY(rt_stage_in(ctx, normals, (size_t)n * 24, tn, &nr));
  RT_TRY(rt_stage_in(ctx, local_rgb, (size_t)n * 24, tl, &lc));
  if (first_draw) RT_TRY(rt_stage_in(ctx, first_draw, (size_t)n * 4, tf, &fd));
  RT_TRY(rt_stage_out_begin(ctx, rgb_out, (size_t)n * 24, tr, &out));
  if (draws_out) RT_TRY(rt_stage_out_begin(ctx, draws_out, (size_t)n * 4, tw, &dr));
  if (p->precision == RT_PRECISION_F64)
    RT_TRY(pl_direct_light<double>(sc, p, n, (const double *)pt, (const double *)nr, (const double *)lc, first_stream, (const uint32_t *)fd, (double *)out, (uint32_t *)dr));
  else
    RT_TRY(pl_direct_light<float>(sc, p, n, (const double *)pt, (const double *)nr, (const double *)lc, first_stream, (const uint32_t *)fd, (double *)out, (uint32_t *)dr));
  if (draws_out) RT_TRY(rt_stage_out_end(ctx, draws_out, (size_t)n * 4, tw));
  return rt_stage_out_end(ctx, rgb_out, (size_t)n * 24, tr);
}

static int photons_explicit(rt_scene *sc, const rt_params *p, uint32_t n, int phase, uint64_t first_stream, const uint32_t *first_draw, int create,
                            const int32_t *source, const int32_t *depth, const int8_t *caustic, const uint8_t *stacks, int track_only_caustic,
                            double *io, uint8_t *kind_out, uint32_t *draws_out, const char *who) {
  if (!sc || !p) return RT_ERR_INVALID;
  rt_ctx *ctx = sc->ctx;
  if (!io || (!create && !kind_out)) return rt_fail(ctx, RT_ERR_INVALID, "%s: the photon array%s required", who, create ? " is" : " and kind_out are");
  if (phase != 1 && phase != 2) return rt_fail(ctx, RT_ERR_INVALID, "%s: phase must be 1 or 2", who);
  if (sc->d32.n_lights == 0 && create) return rt_fail(ctx, RT_ERR_INVALID, "%s: the scene has no lights", who);
  if (n == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem tf, ts, td, tc, tk, tio, tkind, tdr; const void *fd = nullptr, *src = nullptr, *dp = nullptr, *ca = nullptr, *st = nullptr; void *dio, *dk = nullptr, *dr = nullptr;
  if (first_draw) RT_TRY(rt_stage_in(ctx, first_draw, (size_t)n * 4, tf, &fd));
  if (source) RT_TRY(rt_stage_in(ctx, source, (size_t)n * 4, ts, &src));
  if (depth) RT_TRY(rt_stage_in(ctx, depth, (size_t)n * 4, td, &dp));
  if (caustic) RT_TRY(rt_stage_in(ctx, caustic, (size_t)n, tc, &ca));
  std::vector<uint64_t> packed;
  if (stacks) {
    if (rt_is_device_ptr(stacks)) return rt_fail(ctx, RT_ERR_INVALID, "%s: stacks must be host memory", who);
    RT_TRY(pack_stacks(ctx, stacks, n, packed));
    RT_TRY(rt_stage_in(ctx, packed.data(), (size_t)n * 8, tk, &st));
  }
  // the photon array goes in and comes out
  const bool io_dev = rt_is_device_ptr(io);
  if (io_dev) dio = io;
  else { RT_HIP(ctx, tio.alloc((size_t)n * 88)); dio = tio.p; if (!create) RT_HIP(ctx, hipMemcpyAsync(dio, io, (size_t)n * 88, hipMemcpyHostToDevice, ctx->stream)); }
  if (kind_out) RT_TRY(rt_stage_out_begin(ctx, kind_out, n, tkind, &dk));
  if (draws_out) RT_TRY(rt_stage_out_begin(ctx, draws_out, (size_t)n * 4, tdr, &dr));
  const int domain = phase == 1 ? RT_DOMAIN_EMIT_PHASE1 : RT_DOMAIN_EMIT_PHASE2;
  if (p->precision == RT_PRECISION_F64)
    RT_TRY(pl_photons_explicit<double>(sc, p, n, domain, first_stream, (const uint32_t *)fd, create, (const int32_t *)src, (const int32_t *)dp, (const int8_t *)ca,
                                       (const uint64_t *)st, track_only_caustic, (double *)dio, (uint8_t *)dk, (uint32_t *)dr));
  else
    RT_TRY(pl_photons_explicit<float>(sc, p, n, domain, first_stream, (const uint32_t *)fd, create, (const int32_t *)src, (const int32_t *)dp, (const int8_t *)ca,
                                      (const uint64_t *)st, track_only_caustic, (double *)dio, (uint8_t *)dk, (uint32_t *)dr));
  if (!io_dev) { RT_HIP(ctx, hipMemcpyAsync(io, dio, (size_t)n * 88, hipMemcpyDeviceToHost, ctx->stream)); RT_HIP(ctx, hipStreamSynchronize(ctx->stream)); }
  if (kind_out) RT_TRY(rt_stage_out_end(ctx, kind_out, n, tkind));
  if (draws_out) RT_TRY(rt_stage_out_end(ctx, draws_out, (size_t)n * 4, tdr));
  return RT_OK;
}

int rt_create_photons(rt_scene *sc, const rt_params *p, uint32_t n, int phase, uint64_t first_stream, const int32_t *source, double *photons_out,
                      uint32_t *draws_out) {
  return photons_explicit(sc, p, n, phase, first_stream, nullptr, 1, source, nullptr, nullptr, nullptr, 0, photons_out, nullptr, draws_out, "rt_create_photons");
}

int rt_walk_photons(rt_scene *sc, const rt_params *p, uint32_t n, int phase, uint64_t first_stream, const uint32_t *first_draw, const int32_t *depth,
                    const int8_t *caustic, const uint8_t *stacks, int track_only_caustic, double *photons, uint8_t *kind_out) {
  return photons_explicit(sc, p, n, phase, first_stream, first_draw, 0, nullptr, depth, caustic, stacks, track_only_caustic, photons, kind_out, nullptr, "rt_walk_photons");
}

// ------------------------------------------------------------------ renderer
int rt_trace_tile(rt_scene *sc, const rt_params *p, rt_photon_map *diffuse, rt_photon_map *caustic, int x0, int y0, int w, int h,
                  double *pre_rgba, double *post_rgba, uint8_t *rgba8, rt_trace_stats *stats) {
  if (!sc || !p) return RT_ERR_INVALID;
  rt_ctx *ctx = sc->ctx;
  if (w <= 0 || h <= 0 || x0 < 0 || y0 < 0 || (uint32_t)(x0 + w) > sc->d32.width || (uint32_t)(y0 + h) > sc->d32.height)
    return rt_fail(ctx, RT_ERR_INVALID, "tile [%d,%d)+[%d,%d) outside the %ux%u canvas", x0, y0, w, h, sc->d32.width, sc->d32.height);
  if (p->sub_sample < 1 || p->sub_sample > 64) return rt_fail(ctx, RT_ERR_INVALID, "sub_sample must be in 1..64");
  if (p->mode != RT_MODE_PHOTON && p->mode != RT_MODE_WHITTED) return rt_fail(ctx, RT_ERR_INVALID, "unknown mode %d", p->mode);
  if ((diffuse && diffuse->ctx != ctx) || (caustic && caustic->ctx != ctx)) return rt_fail(ctx, RT_ERR_INVALID, "photon map belongs to another context");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  if (p->super_sample_block < 0 || p->super_sample_block > 65536) return rt_fail(ctx, RT_ERR_INVALID, "super_sample_block must be in 0..65536");
  const int blk = p->super_sample_block > 1 ? p->super_sample_block : 1;
  size_t npx = (size_t)((w + blk - 1) / blk) * (size_t)((h + blk - 1) / blk);      // outputs hold one value per block
  if (p->band_count > 1) {
    if (blk > 1) return rt_fail(ctx, RT_ERR_INVALID, "row bands and super_sample_block cannot be combined");
    if (p->band_rows < 8 || p->band_rows > 4096 || (p->band_rows & (p->band_rows - 1)) || p->band_index < 0 || p->band_index >= p->band_count)
      return rt_fail(ctx, RT_ERR_INVALID, "row bands: band_rows must be a power of two in 8..4096 and 0 <= band_index < band_count");
    npx = (size_t)w * rt_band_rows((uint32_t)h, p->band_rows, p->band_count, p->band_index);      // outputs hold this rank's rows only
    if (npx == 0) return RT_OK;
  }
  DevMem t1, t2, t3; void *pre, *post, *r8;
  RT_TRY(rt_stage_out_begin(ctx, pre_rgba, npx * 32, t1, &pre));
  RT_TRY(rt_stage_out_begin(ctx, post_rgba, npx * 32, t2, &post));
  RT_TRY(rt_stage_out_begin(ctx, rgba8, npx * 4, t3, &r8));
  DevMap md, mc; memset(&md, 0, sizeof md); memset(&mc, 0, sizeof mc);
  if (diffuse) md = diffuse->view();
  if (caustic) mc = caustic->view();
  // A frame that is only queued (no statistics, outputs in device memory) may overlap the one queued before it:
  // with two frames in flight consecutive ones alternate between two streams and two sets of workspaces.  Every
  // other call first waits for what the second stream still holds.
  const bool queued = !stats && !t1.p && !t2.p && !t3.p;
  int slot = 0;
  if (queued && ctx->frames_in_flight > 1) { slot = sc->next_slot % ctx->frames_in_flight; sc->next_slot = (slot + 1) % ctx->frames_in_flight; }
  else RT_TRY(rt_sync_other_streams(ctx));
  if (slot) { ctx->stream = ctx->streams[slot]; ctx->alt_used = true; }
  const int rc = p->precision == RT_PRECISION_F64
      ? pl_trace_tile<double>(sc, slot, p, md, mc, x0, y0, w, h, (double *)pre, (double *)post, (uint8_t *)r8, stats)
      : pl_trace_tile<float>(sc, slot, p, md, mc, x0, y0, w, h, (double *)pre, (double *)post, (uint8_t *)r8, stats);
  ctx->stream = ctx->streams[0];
  if (rc != RT_OK) return rc;
  if (slot) { RT_HIP(ctx, hipEventRecord(ctx->frame_done[slot], ctx->streams[slot])); ctx->frame_pending[slot] = true; }
  if (queued) return RT_OK;                                  // nothing to copy back, the caller synchronises
  RT_TRY(rt_stage_out_end(ctx, pre_rgba, npx * 32, t1));
  RT_TRY(rt_stage_out_end(ctx, post_rgba, npx * 32, t2));
  return rt_stage_out_end(ctx, rgba8, npx * 4, t3);
}

// ------------------------------------------------------------------ emission
int pl_emit_counters(rt_scene *sc, bool reset, uint64_t *segments, uint64_t *errors) {
  rt_ctx *ctx = sc->ctx;
  EmitCounters hc; memset(&hc, 0, sizeof hc);
  if (sc->ws_misc.p) {
    RT_HIP(ctx, hipMemcpyAsync(&hc, sc->ws_misc.p, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
    if (reset) { RT_HIP(ctx, hipMemsetAsync(sc->ws_misc.p, 0, 64, ctx->stream)); RT_HIP(ctx, hipMemsetAsync((char *)sc->ws_misc.p + 256, 0, sizeof hc - 256, ctx->stream)); }   // (not the words at byte 64)
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  for (int k = 0; k < RT_EMIT_CTR_SLOTS; k++) { hc.segments += hc.slots[k * 16]; hc.errors += hc.slots[k * 16 + 1]; hc.overflows += hc.slots[k * 16 + 2]; }
  if (segments) *segments = hc.segments;
  if (errors) *errors = hc.errors;
  return RT_OK;
}

static int emit_batch(rt_scene *sc, const rt_params *p, int phase, uint64_t a0, uint32_t count) {
  return p->precision == RT_PRECISION_F64 ? pl_emit_batch<double>(sc, p, phase, a0, count) : pl_emit_batch<float>(sc, p, phase, a0, count);
}
static int emit_compact(rt_scene *sc, const rt_params *p, uint32_t count, int want, uint64_t a0, uint32_t base, uint32_t cap,
                        double *rec, int64_t *att, uint8_t *kind, uint32_t *n) {
  return p->precision == RT_PRECISION_F64 ? pl_emit_compact<double>(sc, count, want, a0, base, cap, rec, att, kind, n)
                                          : pl_emit_compact<float>(sc, count, want, a0, base, cap, rec, att, kind, n);
}

int rt_emit_photons(rt_scene *sc, const rt_params *p, int phase, uint64_t attempt_begin, uint64_t attempt_end, uint32_t cap,
                    double *rec_out, int64_t *attempt_out, uint8_t *kind_out, uint32_t *n_out, rt_emit_stats *stats) {
  if (!sc || !p || !rec_out || !n_out) return RT_ERR_INVALID;
  rt_ctx *ctx = sc->ctx;
  if (phase != 1 && phase != 2) return rt_fail(ctx, RT_ERR_INVALID, "phase must be 1 or 2");
  if (attempt_end < attempt_begin || attempt_end - attempt_begin > 0x7FFFFFFFull) return rt_fail(ctx, RT_ERR_INVALID, "bad attempt range");
  if (sc->d32.n_lights == 0) return rt_fail(ctx, RT_ERR_INVALID, "No lights! Can't emit photons.");     // rayphoton.js:1355-1358
  *n_out = 0;
  const uint32_t count = (uint32_t)(attempt_end - attempt_begin);
  if (count == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem t1, t2, t3; void *rec, *att, *kind;
  RT_TRY(rt_stage_out_begin(ctx, rec_out, (size_t)cap * 88, t1, &rec));
  RT_TRY(rt_stage_out_begin(ctx, attempt_out, (size_t)cap * 8, t2, &att));
  RT_TRY(rt_stage_out_begin(ctx, kind_out, (size_t)cap, t3, &kind));
  RT_TRY(pl_emit_counters(sc, true, nullptr, nullptr));
  RT_TRY(emit_batch(sc, p, phase, attempt_begin, count));
  uint32_t n = 0;
  RT_TRY(emit_compact(sc, p, count, 0, attempt_begin, 0, cap, (double *)rec, (int64_t *)att, (uint8_t *)kind, &n));
  *n_out = n < cap ? n : cap;
  if (stats) {
    memset(stats, 0, sizeof *stats);
    uint64_t seg = 0, err = 0;
    RT_TRY(pl_emit_counters(sc, true, &seg, &err));
    stats->segments = seg; stats->errors = (int32_t)err;
    if (phase == 1) stats->attempts_phase1 = count; else stats->attempts_phase2 = count;
  }
  RT_TRY(rt_stage_out_end(ctx, rec_out, (size_t)*n_out * 88, t1));
  RT_TRY(rt_stage_out_end(ctx, attempt_out, (size_t)*n_out * 8, t2));
  return rt_stage_out_end(ctx, kind_out, (size_t)*n_out, t3);
}

// ------------------------------------------------------------------ device-resident photon lists
// photonListDiffuse / photonListCaustic (rayphoton.js:1346-1352) while they are being filled: records, the attempt that
// stored each of them, and the caps, all in device memory.  Shards of stored photons (rt_emit_shard; rt_emit.h) are fed
// to rt_lists_apply batch by batch in attempt order; the host sees eight words of state per batch and nothing else.
struct rt_photon_lists {
  rt_ctx *ctx = nullptr;
  int format = RT_SHARD_FULL, precision = RT_PRECISION_F32;
  uint32_t W = 0, cap_d = 0, cap_c = 0;
  DevMem list[2], att[2], blocks, state;
  ListsState host = {};
};

static uint32_t lists_words(int format, int precision) {
  return format == RT_SHARD_PACKED ? 9u : (precision == RT_PRECISION_F64 ? 22u : 11u);
}

int rt_lists_create(rt_ctx *ctx, const rt_params *p, int format, rt_photon_lists **out) {
  if (!ctx || !p || !out) return RT_ERR_INVALID;
  *out = nullptr;
  if (format != RT_SHARD_FULL && format != RT_SHARD_PACKED) return rt_fail(ctx, RT_ERR_INVALID, "unknown record format %d", format);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  rt_photon_lists *L = new rt_photon_lists();
  L->ctx = ctx; L->format = format; L->precision = p->precision == RT_PRECISION_F64 ? RT_PRECISION_F64 : RT_PRECISION_F32;
  L->W = lists_words(format, L->precision); L->cap_d = p->num_photons_diffuse; L->cap_c = p->num_photons_caustic;
  hipError_t e = L->list[0].alloc((size_t)L->cap_d * L->W * 4);
  if (e == hipSuccess) e = L->list[1].alloc((size_t)L->cap_c * L->W * 4);
  if (e == hipSuccess) e = L->att[0].alloc((size_t)L->cap_d * 8);
  if (e == hipSuccess) e = L->att[1].alloc((size_t)L->cap_c * 8);
  if (e == hipSuccess) e = L->state.alloc(sizeof(ListsState));
  if (e == hipSuccess) e = hipMemsetAsync(L->state.p, 0, sizeof(ListsState), ctx->stream);
  if (e != hipSuccess) { delete L; return rt_fail(ctx, RT_ERR_HIP, "photon list allocation failed: %s", hipGetErrorString(e)); }
  *out = L;
  return RT_OK;
}
void rt_lists_free(rt_photon_lists *L) { if (L) { (void)hipSetDevice(L->ctx->device); delete L; } }
uint32_t rt_lists_record_words(const rt_photon_lists *L) { return L ? L->W : 0; }
rt_ctx *rt_lists_ctx(rt_photon_lists *L) { return L->ctx; }

// extent: the merged positions to look at -- n_shards * stride, or fewer when the caller knows that no record lies beyond
// (one shard whose count the host has read)
static int lists_apply(rt_photon_lists *L, int phase, uint64_t tag_base, const uint32_t *shards, const uint32_t *counts, uint32_t n_shards,
                       uint32_t stride, uint32_t extent, rt_lists_info *info) {
  if (!L) return RT_ERR_INVALID;
  rt_ctx *ctx = L->ctx;
  if (phase != 1 && phase != 2) return rt_fail(ctx, RT_ERR_INVALID, "phase must be 1 or 2");
  if (!shards || !counts || n_shards == 0 || stride == 0) return rt_fail(ctx, RT_ERR_INVALID, "rt_lists_apply: shards, counts, n_shards and stride are required");
  if ((uint64_t)n_shards * stride > 0x7FFFFFFFull) return rt_fail(ctx, RT_ERR_TOO_LARGE, "batch of %u x %u records is too large", n_shards, stride);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem ts, tc; const void *dsh, *dct;
  RT_TRY(rt_stage_in(ctx, shards, (size_t)n_shards * stride * (L->W + 1) * 4, ts, &dsh));
  RT_TRY(rt_stage_in(ctx, counts, (size_t)n_shards * 4, tc, &dct));
  const uint32_t total = std::max(1u, std::min(extent, n_shards * stride)), blocks = div_up(total, 256);
  RT_HIP(ctx, L->blocks.reserve(((size_t)blocks * 2 + 2) * 4));
  uint32_t *bd = L->blocks.as<uint32_t>(), *bcq = bd + blocks, *totals = bcq + blocks;
  ListsState *st = L->state.as<ListsState>();
  RT_HIP(ctx, hipMemsetAsync(&st->last_tag, 0xFF, 8, ctx->stream));        // last_tag = pad_ = 0xFFFFFFFF: no attempt has ended the phase
  RT_LAUNCH(ctx, "lists_count", k_lists_count, blocks, 256, (const uint32_t *)dsh, (const uint32_t *)dct, n_shards, stride, L->W, bd, bcq);
  RT_LAUNCH(ctx, "lists_scan", k_scan_blocks, 1, 1024, bd, blocks, totals, (const uint32_t *)nullptr);
  RT_LAUNCH(ctx, "lists_scan", k_scan_blocks, 1, 1024, bcq, blocks, totals + 1, (const uint32_t *)nullptr);
  RT_LAUNCH(ctx, "lists_take", k_lists_take<0>, blocks, 256, (const uint32_t *)dsh, (const uint32_t *)dct, n_shards, stride, L->W, (const uint32_t *)bd,
            (const uint32_t *)bcq, (const uint32_t *)totals, phase, L->cap_d, L->cap_c, (unsigned long long)tag_base, st, L->list[0].as<uint32_t>(),
            L->att[0].as<long long>(), L->list[1].as<uint32_t>(), L->att[1].as<long long>());
  if (phase == 1)
    RT_LAUNCH(ctx, "lists_take", k_lists_take<1>, blocks, 256, (const uint32_t *)dsh, (const uint32_t *)dct, n_shards, stride, L->W, (const uint32_t *)bd,
              (const uint32_t *)bcq, (const uint32_t *)totals, phase, L->cap_d, L->cap_c, (unsigned long long)tag_base, st, L->list[0].as<uint32_t>(),
              L->att[0].as<long long>(), L->list[1].as<uint32_t>(), L->att[1].as<long long>());
  RT_LAUNCH(ctx, "lists_finish", k_lists_finish, 1, 1, (const uint32_t *)totals, phase, L->cap_d, L->cap_c, st);
  RT_HIP(ctx, hipGetLastError());
  static_assert(sizeof(ListsState) <= 64, "the state goes through the context's pinned words");
  RT_HIP(ctx, hipMemcpyAsync(ctx->pinned, st, sizeof(ListsState), hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(&L->host, ctx->pinned, sizeof(ListsState));
  if (info) {
    info->n_diffuse = L->host.n_d; info->n_caustic = L->host.n_c; info->batch_diffuse = L->host.total_d; info->batch_caustic = L->host.total_c;
    info->taken_diffuse = L->host.take_d; info->taken_caustic = L->host.take_c;
    info->phase_ended = L->host.last_tag != 0xFFFFFFFFu;
    info->last_attempt = info->phase_ended ? tag_base + L->host.last_tag : 0;
  }
  return RT_OK;
}

int rt_lists_apply(rt_photon_lists *L, int phase, uint64_t tag_base, const uint32_t *shards, const uint32_t *counts, uint32_t n_shards,
                   uint32_t stride, rt_lists_info *info) {
  return lists_apply(L, phase, tag_base, shards, counts, n_shards, stride, 0xFFFFFFFFu, info);
}

int rt_lists_counts(rt_photon_lists *L, uint32_t *n_diffuse, uint32_t *n_caustic) {
  if (!L) return RT_ERR_INVALID;
  if (n_diffuse) *n_diffuse = L->host.n_d;
  if (n_caustic) *n_caustic = L->host.n_c;
  return RT_OK;
}

// the records of one list as the boundary's 11 doubles (full format) and their (phase, attempt) pairs
int rt_lists_export(rt_photon_lists *L, int which, double *rec_out, int64_t *attempt_out) {
  if (!L || (which != 0 && which != 1)) return RT_ERR_INVALID;
  rt_ctx *ctx = L->ctx;
  if (L->format != RT_SHARD_FULL && rec_out) return rt_fail(ctx, RT_ERR_INVALID, "rt_lists_export: packed lists hold no full records");
  const uint32_t n = which == 0 ? L->host.n_d : L->host.n_c;
  if (n == 0) return RT_OK;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  DevMem t1, t2; void *drec, *datt;
  RT_TRY(rt_stage_out_begin(ctx, rec_out, (size_t)n * 88, t1, &drec));
  RT_TRY(rt_stage_out_begin(ctx, attempt_out, (size_t)n * 16, t2, &datt));
  if (L->precision == RT_PRECISION_F64 || L->format == RT_SHARD_PACKED)
    RT_LAUNCH(ctx, "lists_export", k_lists_export<double>, div_up(n, 256), 256, (const uint32_t *)L->list[which].p, (const long long *)L->att[which].p, n,
              (double *)drec, (long long *)datt);
  else
    RT_LAUNCH(ctx, "lists_export", k_lists_export<float>, div_up(n, 256), 256, (const uint32_t *)L->list[which].p, (const long long *)L->att[which].p, n,
              (double *)drec, (long long *)datt);
  RT_HIP(ctx, hipGetLastError());
  RT_TRY(rt_stage_out_end(ctx, rec_out, (size_t)n * 88, t1));
  return rt_stage_out_end(ctx, attempt_out, (size_t)n * 16, t2);
}

int rt_lists_view(rt_photon_lists *L, int which, const void **records_dev, uint32_t *n, uint32_t *words, int *precision) {
  if (!L || (which != 0 && which != 1)) return RT_ERR_INVALID;
  if (records_dev) *records_dev = L->list[which].p;
  if (n) *n = which == 0 ? L->host.n_d : L->host.n_c;
  if (words) *words = L->W;
  if (precision) *precision = L->precision;
  return RT_OK;
}

// attempts [begin, end) of one phase on this GPU, left as a shard (rt_emit.h) in attempt order
int rt_emit_shard(rt_scene *sc, const rt_params *p, int phase, uint64_t attempt_begin, uint64_t attempt_end, int format, uint64_t tag_base,
                  uint32_t cap, uint32_t *shard_out, uint32_t *n_out_dev, rt_emit_stats *stats) {
  if (!sc || !p || !shard_out || !n_out_dev) return RT_ERR_INVALID;
  rt_ctx *ctx = sc->ctx;
  if (phase != 1 && phase != 2) return rt_fail(ctx, RT_ERR_INVALID, "phase must be 1 or 2");
  if (format != RT_SHARD_FULL && format != RT_SHARD_PACKED) return rt_fail(ctx, RT_ERR_INVALID, "unknown record format %d", format);
  if (attempt_end < attempt_begin || attempt_end - attempt_begin > 0x7FFFFFFFull) return rt_fail(ctx, RT_ERR_INVALID, "bad attempt range");
  if (attempt_begin < tag_base || attempt_end - tag_base > 0x7FFFFFFFull) return rt_fail(ctx, RT_ERR_INVALID, "attempt range outside [tag_base, tag_base + 2^31)");
  if (sc->d32.n_lights == 0) return rt_fail(ctx, RT_ERR_INVALID, "No lights! Can't emit photons.");     // rayphoton.js:1355-1358
  RT_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t count = (uint32_t)(attempt_end - attempt_begin);
  const uint32_t W = lists_words(format, p->precision);
  DevMem t1, t2; void *dsh, *dn;
  RT_TRY(rt_stage_out_begin(ctx, shard_out, (size_t)cap * (W + 1) * 4, t1, &dsh));
  RT_TRY(rt_stage_out_begin(ctx, n_out_dev, 4, t2, &dn));
  if (count == 0) { RT_HIP(ctx, hipMemsetAsync(dn, 0, 4, ctx->stream)); }
  else {
    RT_TRY(pl_emit_counters(sc, true, nullptr, nullptr));
    RT_TRY(emit_batch(sc, p, phase, attempt_begin, count));
    if (p->precision == RT_PRECISION_F64) RT_TRY(pl_emit_shard<double>(sc, count, format, (uint32_t)(attempt_begin - tag_base), cap, (uint32_t *)dsh, (uint32_t *)dn));
    else RT_TRY(pl_emit_shard<float>(sc, count, format, (uint32_t)(attempt_begin - tag_base), cap, (uint32_t *)dsh, (uint32_t *)dn));
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    uint64_t seg = 0, err = 0;
    if (count) RT_TRY(pl_emit_counters(sc, true, &seg, &err));
    stats->segments = seg; stats->errors = (int32_t)err;
    if (phase == 1) stats->attempts_phase1 = count; else stats->attempts_phase2 = count;
  }
  RT_TRY(rt_stage_out_end(ctx, shard_out, (size_t)cap * (W + 1) * 4, t1));
  return rt_stage_out_end(ctx, n_out_dev, 4, t2);
}

// emitManyPhotons (rayphoton.js:1354-1383) in batches on one GPU.  Every batch's stored photons are compacted into a
// shard and the list caps are applied on the device exactly as the sequential loop would: phase 1 ends with the attempt
// that stores the N_D-th diffuse photon, caustic photons of phase-1 attempts up to that one are kept (up to N_C), phase 2
// stops at the N_C-th caustic.  Per batch the host reads back eight words.
static int emit_many_lists(rt_scene *sc, const rt_params *p, int format, rt_photon_lists **lists_out, rt_emit_stats *stats) {
  rt_ctx *ctx = sc->ctx;
  rt_emit_stats st; memset(&st, 0, sizeof st);
  if (stats) *stats = st;
  *lists_out = nullptr;
  if (sc->d32.n_lights == 0) return rt_fail(ctx, RT_ERR_INVALID, "No lights! Can't emit photons.");
  const uint32_t ND = p->num_photons_diffuse, NC = p->num_photons_caustic;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  rt_photon_lists *L = nullptr;
  RT_TRY(rt_lists_create(ctx, p, format, &L));
  struct Guard { rt_photon_lists *&l; bool keep = false; ~Guard() { if (!keep) { rt_lists_free(l); l = nullptr; } } } guard{ L };
  RT_TRY(pl_emit_counters(sc, true, nullptr, nullptr));
  if (!sc->ws_misc.p) { RT_HIP(ctx, sc->ws_misc.alloc(sizeof(EmitCounters))); }
  unsigned long long *seg_exact = (unsigned long long *)((char *)sc->ws_misc.p + 64);
  RT_HIP(ctx, hipMemsetAsync(sc->ws_misc.p, 0, sizeof(EmitCounters), ctx->stream));
  uint32_t nd = 0, nc = 0;
  const uint32_t W = L->W;
  for (int phase = 1; phase <= 2 && !st.capped; phase++) {
    uint64_t attempt = 0;
    const uint64_t cap = 100ull * (phase == 1 ? ND : NC);
    const uint32_t got0 = phase == 1 ? nd : nc;                // what the list held when the phase began
    while (phase == 1 ? nd < ND : nc < NC) {
      const uint32_t need = phase == 1 ? ND - nd : NC - nc, got = (phase == 1 ? nd : nc) - got0;
      // Batch sizes.  Every attempt of a batch is traced, also those behind the attempt that fills the list, so a batch
      // should end just behind it: a first batch of at most 2^18 attempts measures the phase's yield (photons per attempt;
      // 0.2 % off at that size), the next one is sized for what is still missing plus 1 %, and whatever a rare shortfall
      // leaves takes batches of the smallest size.  While nothing has been stored the batches grow as 4 x the need.
      uint64_t batch = std::max<uint64_t>(4ull * need, 65536);
      if (attempt == 0) batch = std::min<uint64_t>(batch, 1u << 18);
      else if (got > 0) batch = std::max<uint64_t>((uint64_t)((double)need * (double)attempt / (double)got * 1.01) + 4096, 65536);
      batch = std::min<uint64_t>(batch, 8u << 20);
      batch = std::min<uint64_t>(batch, cap + 1 - attempt);
      const uint32_t count = (uint32_t)batch;
      RT_TRY(emit_batch(sc, p, phase, attempt, count));
      RT_HIP(ctx, sc->ws_emit_shard.reserve((size_t)count * (W + 1) * 4 + 16));
      uint32_t *shard = sc->ws_emit_shard.as<uint32_t>(), *n_dev = shard + (size_t)count * (W + 1);
      if (p->precision == RT_PRECISION_F64) RT_TRY(pl_emit_shard<double>(sc, count, format, 0u, count, shard, n_dev));
      else RT_TRY(pl_emit_shard<float>(sc, count, format, 0u, count, shard, n_dev));
      // (the caps look at the shard's records only, not at a position per attempt: the count is worth a host round trip)
      uint32_t n_stored = count;
      if (count > (1u << 20)) {                              // (a small batch: looking at every position costs less than the wait)
        RT_HIP(ctx, hipMemcpyAsync(ctx->pinned, n_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        n_stored = std::min(ctx->pinned[0], count);
      }
      rt_lists_info info;
      RT_TRY(lists_apply(L, phase, attempt, shard, n_dev, 1, count, n_stored, &info));
      nd = info.n_diffuse; nc = info.n_caustic;
      const uint64_t last_attempt = info.phase_ended ? info.last_attempt : attempt + count - 1;    // last attempt the sequential loop would run
      RT_LAUNCH(ctx, "sum_segments", k_sum_segments, 256, 256, (const uint8_t *)sc->ws_emit_seg.p, (uint32_t)(last_attempt - attempt + 1), seg_exact);
      attempt = last_attempt + 1;
      if (attempt > cap) { st.capped = 1; break; }      // `attempt++; if (attempt > 100 * N) return` (:1364-1368)
    }
    if (phase == 1) st.attempts_phase1 = attempt; else st.attempts_phase2 = attempt;
  }
  uint64_t seg = 0, err = 0;
  unsigned long long exact = 0;
  RT_HIP(ctx, hipMemcpyAsync(&exact, seg_exact, 8, hipMemcpyDeviceToHost, ctx->stream));
  RT_TRY(pl_emit_counters(sc, true, &seg, &err));
  st.segments = exact; st.errors = (int32_t)err; st.n_diffuse = nd; st.n_caustic = nc;
  if (stats) *stats = st;
  guard.keep = true;
  *lists_out = L;
  // The reference throws here (rayphoton.js:1489).  The fp64 build mirrors that; in fp32 the rim gaps between a
  // cylinder and its caps are epsilon_f32 wide, a photon can slip through one and later meet the inside of the
  // glass with an empty stack: such attempts are dropped and only counted (stats.errors).
  if (err && p->precision == RT_PRECISION_F64)
    return rt_fail(ctx, RT_ERR_MATERIAL_STACK, "photon exited all materials? check consistency of objects? (%llu attempts)", (unsigned long long)err);   // rayphoton.js:1489
  if (st.capped) return rt_fail(ctx, RT_ERR_ATTEMPT_CAP, "I tried many times but only created %u diffuse / %u caustic photons.", nd, nc);
  return RT_OK;
}

int rt_emit_many_lists(rt_scene *sc, const rt_params *p, int format, rt_photon_lists **lists_out, rt_emit_stats *stats) {
  if (!sc || !p || !lists_out) return RT_ERR_INVALID;
  if (format != RT_SHARD_FULL && format != RT_SHARD_PACKED) return rt_fail(sc->ctx, RT_ERR_INVALID, "unknown record format %d", format);
  return emit_many_lists(sc, p, format, lists_out, stats);
}

int rt_emit_many(rt_scene *sc, const rt_params *p, double *diffuse, int64_t *diffuse_attempt, double *caustic,
                 int64_t *caustic_attempt, rt_emit_stats *stats) {
  if (!sc || !p) return RT_ERR_INVALID;
  rt_photon_lists *L = nullptr;
  const int rc = emit_many_lists(sc, p, RT_SHARD_FULL, &L, stats);
  if (!L) return rc;
  int rc2 = rt_lists_export(L, 0, diffuse, diffuse_attempt);
  if (rc2 == RT_OK) rc2 = rt_lists_export(L, 1, caustic, caustic_attempt);
  rt_lists_free(L);
  return rc != RT_OK ? rc : rc2;
}

