// N-API shim: exposes the C ABI of librt355.so (include/rt355.h) to the Node host.
// Thin by design: argument unpacking, typed-array borrowing for the duration of a call,
// opaque handles as napi_external with finalisers (tagged boxes that share their context), and status -> thrown JS Error (the
// reference's convention: `throw new Error`, kdtree.js:6, rayphoton.js:640, :1489).
#include <node_api.h>
#include <stdint.h>
#include <string.h>
#include <set>
#include <string>
#include <vector>
#include "../../../include/rt355.h"

#define NAPI_OK(call)                                                  \
  do { if ((call) != napi_ok) { napi_throw_error(env, nullptr, "N-API call failed: " #call); return nullptr; } } while (0)

static napi_value fail(napi_env env, rt_ctx *ctx, int code) {
  std::string msg = std::string(rt_last_error(ctx)) + " (rt355 status " + std::to_string(code) + ")";
  napi_throw_error(env, std::to_string(code).c_str(), msg.c_str());
  return nullptr;
}

struct Args {
  napi_env env; size_t argc; napi_value argv[16];
  Args(napi_env e, napi_callback_info info) : env(e), argc(16) { napi_get_cb_info(e, info, &argc, argv, nullptr, nullptr); }
  Args(napi_env e, napi_value one) : env(e), argc(1) { argv[0] = one; }          // a single value, for the checks below
  bool has(size_t i) const {
    if (i >= argc) return false;
    napi_valuetype t; napi_typeof(env, argv[i], &t);
    return t != napi_undefined && t != napi_null;
  }
  double num(size_t i, double dflt = 0) const { double v = dflt; if (has(i)) napi_get_value_double(env, argv[i], &v); return v; }
  // A handle of this addon: every box starts with the tag of its kind, so a context passed where a scene is expected
  // (or a handle whose context has been shut down) is a TypeError, not a reinterpreted pointer.
  template <typename T> T *ext(size_t i) const {
    void *p = nullptr;
    if (!has(i)) return nullptr;
    napi_valuetype t; napi_typeof(env, argv[i], &t);
    if (t != napi_external || napi_get_value_external(env, argv[i], &p) != napi_ok || !p) { napi_throw_type_error(env, nullptr, T::what()); return nullptr; }
    T *box = (T *)p;
    if (box->tag != T::TAG) { napi_throw_type_error(env, nullptr, T::what()); return nullptr; }
    if (!box->live()) { napi_throw_error(env, nullptr, "rt355: the handle's context has been shut down"); return nullptr; }
    return box;
  }
  // typed array -> pointer + element count (borrowed for this call only)
  void *typed(size_t i, napi_typedarray_type want, size_t *len = nullptr) const {
    if (!has(i)) { if (len) *len = 0; return nullptr; }
    bool is = false; napi_is_typedarray(env, argv[i], &is);
    if (!is) { napi_throw_type_error(env, nullptr, "typed array expected"); return nullptr; }
    napi_typedarray_type t; size_t n; void *data; napi_value ab; size_t off;
    napi_get_typedarray_info(env, argv[i], &t, &n, &data, &ab, &off);
    if (t != want) { napi_throw_type_error(env, nullptr, "typed array of the wrong element type"); return nullptr; }
    if (len) *len = n;
    return data;
  }
};

static napi_value get_prop(napi_env env, napi_value obj, const char *name) { napi_value v; napi_get_named_property(env, obj, name, &v); return v; }
static double prop_num(napi_env env, napi_value obj, const char *name, double dflt) {
  bool has = false; napi_has_named_property(env, obj, name, &has);
  if (!has) return dflt;
  napi_value v = get_prop(env, obj, name); napi_valuetype t; napi_typeof(env, v, &t);
  if (t != napi_number) return dflt;
  double d; napi_get_value_double(env, v, &d); return d;
}
static void *prop_typed(napi_env env, napi_value obj, const char *name, napi_typedarray_type want, size_t *len) {
  napi_value v = get_prop(env, obj, name);
  bool is = false; napi_is_typedarray(env, v, &is);
  if (!is) { *len = 0; return nullptr; }
  napi_typedarray_type t; void *data; napi_value ab; size_t off;
  napi_get_typedarray_info(env, v, &t, len, &data, &ab, &off);
  return t == want ? data : nullptr;
}

// params object -> rt_params (missing fields keep the reference's defaults)
static void read_params(napi_env env, napi_value o, rt_params *p) {
  rt_default_params(p);
  napi_valuetype t; napi_typeof(env, o, &t);
  if (t != napi_object) return;
  p->epsilon = prop_num(env, o, "epsilon", p->epsilon);
  p->max_trace_dist = prop_num(env, o, "maxTraceDist", p->max_trace_dist);
  p->max_depth = (int32_t)prop_num(env, o, "maxDepth", p->max_depth);
  p->sub_sample = (int32_t)prop_num(env, o, "subSample", p->sub_sample);
  p->num_photons_diffuse = (uint32_t)prop_num(env, o, "numPhotonsDiffuse", p->num_photons_diffuse);
  p->num_photons_caustic = (uint32_t)prop_num(env, o, "numPhotonsCaustic", p->num_photons_caustic);
  p->diffuse_radius = prop_num(env, o, "diffuseRadius", p->diffuse_radius);
  p->caustic_radius = prop_num(env, o, "causticRadius", p->caustic_radius);
  p->standard_lamp_area = prop_num(env, o, "standardLampArea", p->standard_lamp_area);
  p->seed = (uint64_t)prop_num(env, o, "seed", (double)p->seed);
  p->use_direct_light = (int32_t)prop_num(env, o, "useDirectLight", p->use_direct_light);
  p->use_photon_maps = (int32_t)prop_num(env, o, "usePhotonMaps", p->use_photon_maps);
  p->precision = (int32_t)prop_num(env, o, "precision", p->precision);
  p->mode = (int32_t)prop_num(env, o, "mode", p->mode);
  p->shadow_bias = prop_num(env, o, "shadowBias", p->shadow_bias);
  p->epsilon_f32 = prop_num(env, o, "epsilonF32", p->epsilon_f32);
  p->max_bounces = (int32_t)prop_num(env, o, "maxBounces", p->max_bounces);
  p->super_sample_block = (int32_t)prop_num(env, o, "superSampleBlock", p->super_sample_block);
  p->band_rows = (int32_t)prop_num(env, o, "bandRows", p->band_rows);           // one rank's share of a tile (multi-GPU)
  p->band_count = (int32_t)prop_num(env, o, "bandCount", p->band_count);
  p->band_index = (int32_t)prop_num(env, o, "bandIndex", p->band_index);
}

static void set_num(napi_env env, napi_value obj, const char *name, double v) { napi_value n; napi_create_double(env, v, &n); napi_set_named_property(env, obj, name, n); }

// Handles.  Scenes and maps point into their context, and the collector orders nothing: a scene may be finalised long
// after shutdown(ctx).  So the context's box is shared: it knows its live scenes and maps, shutdown() frees them first
// and marks them dead, and the box itself lives until the last handle that refers to it has been finalised.
struct SceneBox; struct MapBox; struct ListsBox; struct DevBox;
struct CtxBox {
  uint64_t tag; rt_ctx *ctx; int refs; std::set<SceneBox *> scenes; std::set<MapBox *> maps; std::set<ListsBox *> lists; std::set<DevBox *> bufs;
  static constexpr uint64_t TAG = 0x7274333535637478ull;
  static const char *what() { return "rt355: a context handle (from init) is expected"; }
  bool live() const { return ctx != nullptr; }
};
struct SceneBox {
  uint64_t tag; CtxBox *cb; rt_scene *scene;
  static constexpr uint64_t TAG = 0x7274333535736365ull;
  static const char *what() { return "rt355: a scene handle (from sceneUpload) is expected"; }
  bool live() const { return scene != nullptr; }
  void drop() { if (scene) { rt_scene_free(scene); scene = nullptr; cb->scenes.erase(this); } }
};
struct MapBox {
  uint64_t tag; CtxBox *cb; rt_photon_map *map;
  static constexpr uint64_t TAG = 0x72743335356d6170ull;
  static const char *what() { return "rt355: a photon map handle (from photonMapBuild) is expected"; }
  bool live() const { return map != nullptr; }
  void drop() { if (map) { rt_photon_map_free(map); map = nullptr; cb->maps.erase(this); } }
};
// photon lists that stay in GPU memory (emitManyLists, listsCreate)
struct ListsBox {
  uint64_t tag; CtxBox *cb; rt_photon_lists *lists;
  static constexpr uint64_t TAG = 0x72743335356c7374ull;
  static const char *what() { return "rt355: a photon lists handle (from emitManyLists or listsCreate) is expected"; }
  bool live() const { return lists != nullptr; }
  void drop() { if (lists) { rt_lists_free(lists); lists = nullptr; cb->lists.erase(this); } }
};
// plain device memory (deviceAlloc): where queued frames, shards and gather inputs live on the Node side
struct DevBox {
  uint64_t tag; CtxBox *cb; void *ptr; size_t bytes;
  static constexpr uint64_t TAG = 0x7274333535627566ull;
  static const char *what() { return "rt355: a device buffer handle (from deviceAlloc) is expected"; }
  bool live() const { return ptr != nullptr; }
  void drop() { if (ptr) { rt_device_free(cb->ctx, ptr); ptr = nullptr; cb->bufs.erase(this); } }
};
static void ctx_close(CtxBox *cb) {                          // frees everything made from the context, then the context
  if (!cb->ctx) return;
  while (!cb->scenes.empty()) (*cb->scenes.begin())->drop();
  while (!cb->maps.empty()) (*cb->maps.begin())->drop();
  while (!cb->lists.empty()) (*cb->lists.begin())->drop();
  while (!cb->bufs.empty()) (*cb->bufs.begin())->drop();
  rt_shutdown(cb->ctx);
  cb->ctx = nullptr;
}
static void ctx_release(CtxBox *cb) { if (--cb->refs == 0) { ctx_close(cb); cb->tag = 0; delete cb; } }
// a box that is already in its context's set -> the JS handle that owns it
template <typename B> static napi_value make_handle(napi_env env, B *box) {
  box->cb->refs++;
  napi_value out;
  auto fin = [](napi_env, void *p, void *) { B *b = (B *)p; CtxBox *cb = b->cb; b->drop(); b->tag = 0; delete b; ctx_release(cb); };
  if (napi_create_external(env, box, fin, nullptr, &out) != napi_ok) { fin(env, box, nullptr); napi_throw_error(env, nullptr, "N-API call failed: napi_create_external"); return nullptr; }
  return out;
}
static bool pending(napi_env env) { bool p = false; napi_is_exception_pending(env, &p); return p; }

// init([deviceId]) -> ctx
static napi_value Init(napi_env env, napi_callback_info info) {
  Args a(env, info);
  int dev = (int)a.num(0, 0);
  rt_ctx *ctx = nullptr;
  int rc = rt_init(&dev, 1, &ctx);
  if (rc != RT_OK) return fail(env, nullptr, rc);
  CtxBox *cb = new CtxBox{ CtxBox::TAG, ctx, 1, {}, {} };
  napi_value out;
  if (napi_create_external(env, cb, [](napi_env, void *p, void *) { ctx_release((CtxBox *)p); }, nullptr, &out) != napi_ok) {
    ctx_release(cb);
    napi_throw_error(env, nullptr, "N-API call failed: napi_create_external");
    return nullptr;
  }
  return out;
}

// shutdown(ctx): frees the context and every scene and map made from it; their handles become unusable (a later call
// with one throws) and their finalisers have nothing left to free
static napi_value Shutdown(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (cb) ctx_close(cb);
  return nullptr;
}

// sceneUpload(ctx, flat) -> scene      flat = what host/scene.js flatten() returns
static napi_value SceneUpload(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  if (!a.has(1)) { napi_throw_type_error(env, nullptr, "sceneUpload(ctx, flat)"); return nullptr; }
  rt_ctx *ctx = cb->ctx;
  napi_value f = a.argv[1];
  rt_scene_desc d; memset(&d, 0, sizeof d);
  size_t n_meta, n_geom, n_col, n_ls, n_ld, n_surf, n;
  d.n_shapes = (uint32_t)prop_num(env, f, "nShapes", 0);
  d.shape_meta = (const uint8_t *)prop_typed(env, f, "shapeMeta", napi_uint8_array, &n_meta);
  d.shape_geom = (const double *)prop_typed(env, f, "shapeGeom", napi_float64_array, &n_geom);
  d.shape_colour = (const double *)prop_typed(env, f, "shapeColour", napi_float64_array, &n_col);
  d.n_lights = (uint32_t)prop_num(env, f, "nLights", 0);
  d.light_shape = (const uint32_t *)prop_typed(env, f, "lightShape", napi_uint32_array, &n_ls);
  d.light_data = (const double *)prop_typed(env, f, "lightData", napi_float64_array, &n_ld);
  d.shape_surface = (const double *)prop_typed(env, f, "shapeSurface", napi_float64_array, &n_surf);   // ray.js shape model, optional
  // the C ABI sees raw pointers only: the arrays must be as long as the counts say
  const size_t ns = d.n_shapes, nl = d.n_lights;
  if (!d.shape_meta || !d.shape_geom || !d.shape_colour || n_meta != ns * 4 || n_geom != ns * 12 || n_col != ns * 4 ||
      (nl > 0 && (!d.light_shape || !d.light_data)) || (d.light_shape && n_ls != nl) || (d.light_data && n_ld != nl * 8) ||
      (d.shape_surface && n_surf != ns * 2)) {
    napi_throw_range_error(env, nullptr, "sceneUpload: flat.shapeMeta / shapeGeom / shapeColour / lightShape / lightData / shapeSurface must hold "
                                         "nShapes*4 / *12 / *4, nLights / nLights*8 and nShapes*2 elements");
    return nullptr;
  }
  const double *cam = (const double *)prop_typed(env, f, "camera", napi_float64_array, &n);
  if (!cam || n < 14) { napi_throw_type_error(env, nullptr, "flat.camera must be a Float64Array(16)"); return nullptr; }
  memcpy(d.camera, cam, sizeof(double) * (n < 16 ? n : 16));
  d.width = (uint32_t)prop_num(env, f, "width", 0); d.height = (uint32_t)prop_num(env, f, "height", 0);
  const double *mats = (const double *)prop_typed(env, f, "materials", napi_float64_array, &n);
  if (!mats || n != 32) { napi_throw_type_error(env, nullptr, "flat.materials must be a Float64Array(32)"); return nullptr; }
  memcpy(d.materials, mats, sizeof d.materials);
  rt_scene *sc = nullptr;
  int rc = rt_scene_upload(ctx, &d, &sc);
  if (rc != RT_OK) return fail(env, ctx, rc);
  SceneBox *box = new SceneBox{ SceneBox::TAG, cb, sc };
  cb->scenes.insert(box);
  return make_handle(env, box);
}

// intersectPairs(ctx, params, rayO, rayD, normalizeDir, shapeMeta, shapeGeom) -> Float64Array t (NaN = undefined)
static napi_value IntersectPairs(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  rt_ctx *ctx = cb->ctx;
  rt_params p; read_params(env, a.argv[1], &p);
  size_t no, nd, nm, ng;
  const double *o = (const double *)a.typed(2, napi_float64_array, &no), *d = (const double *)a.typed(3, napi_float64_array, &nd);
  const uint8_t *meta = (const uint8_t *)a.typed(5, napi_uint8_array, &nm);
  const double *geom = (const double *)a.typed(6, napi_float64_array, &ng);
  if (!o || !d || !meta || !geom) { napi_throw_type_error(env, nullptr, "intersectPairs: bad arguments"); return nullptr; }
  uint32_t n = (uint32_t)(no / 3);
  if (nd != no || nm != (size_t)n * 4 || ng != (size_t)n * 12) { napi_throw_range_error(env, nullptr, "intersectPairs: array sizes disagree"); return nullptr; }
  napi_value ab, out; void *data;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 8, &data, &ab));
  NAPI_OK(napi_create_typedarray(env, napi_float64_array, n, ab, 0, &out));
  int rc = rt_intersect_pairs(ctx, &p, n, o, d, (int)a.num(4, 0), meta, geom, (double *)data);
  if (rc != RT_OK) return fail(env, ctx, rc);
  return out;
}

// intersect(scene, params, mode, rayO, rayD[, sourceDist]) -> {t: Float64Array, idx: Int32Array}
static napi_value Intersect(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  rt_params p; read_params(env, a.argv[1], &p);
  size_t no, nd, ns = 0;
  const double *o = (const double *)a.typed(3, napi_float64_array, &no), *d = (const double *)a.typed(4, napi_float64_array, &nd);
  const double *sd = (const double *)a.typed(5, napi_float64_array, &ns);
  if (!sb) return nullptr;
  if (!o || !d || no != nd) { napi_throw_type_error(env, nullptr, "intersect: bad arguments"); return nullptr; }
  uint32_t n = (uint32_t)(no / 3);
  if (sd && ns != n) { napi_throw_range_error(env, nullptr, "intersect: sourceDist size"); return nullptr; }
  napi_value abt, abi, t, idx, out; void *td, *id;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 8, &td, &abt)); NAPI_OK(napi_create_typedarray(env, napi_float64_array, n, abt, 0, &t));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 4, &id, &abi)); NAPI_OK(napi_create_typedarray(env, napi_int32_array, n, abi, 0, &idx));
  int rc = rt_intersect(sb->scene, &p, (int)a.num(2, 0), n, o, d, sd, (double *)td, (int32_t *)id);
  if (rc != RT_OK) return fail(env, sb->cb->ctx, rc);
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "t", t); napi_set_named_property(env, out, "idx", idx);
  return out;
}

// emitMany(scene, params) -> {diffuse: Float64Array(n*11), caustic, diffuseAttempt: Float64Array(n*2), causticAttempt, stats}
static napi_value EmitMany(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  const uint32_t ND = p.num_photons_diffuse, NC = p.num_photons_caustic;
  std::vector<double> d((size_t)ND * 11), c((size_t)NC * 11);
  std::vector<int64_t> da((size_t)ND * 2), ca((size_t)NC * 2);
  rt_emit_stats st;
  int rc = rt_emit_many(sb->scene, &p, d.data(), da.data(), c.data(), ca.data(), &st);
  if (rc != RT_OK && rc != RT_ERR_ATTEMPT_CAP) return fail(env, sb->cb->ctx, rc);     // the cap only logs in the reference (rayphoton.js:1366)
  napi_value out, stats; NAPI_OK(napi_create_object(env, &out)); NAPI_OK(napi_create_object(env, &stats));
  auto f64 = [&](const char *name, const double *src, size_t n) {
    napi_value ab, ta; void *data; napi_create_arraybuffer(env, n * 8, &data, &ab); memcpy(data, src, n * 8);
    napi_create_typedarray(env, napi_float64_array, n, ab, 0, &ta); napi_set_named_property(env, out, name, ta);
  };
  f64("diffuse", d.data(), (size_t)st.n_diffuse * 11); f64("caustic", c.data(), (size_t)st.n_caustic * 11);
  std::vector<double> tmp;
  tmp.assign(da.begin(), da.begin() + (size_t)st.n_diffuse * 2); f64("diffuseAttempt", tmp.data(), tmp.size());
  tmp.assign(ca.begin(), ca.begin() + (size_t)st.n_caustic * 2); f64("causticAttempt", tmp.data(), tmp.size());
  set_num(env, stats, "attemptsPhase1", (double)st.attempts_phase1); set_num(env, stats, "attemptsPhase2", (double)st.attempts_phase2);
  set_num(env, stats, "segments", (double)st.segments); set_num(env, stats, "nDiffuse", st.n_diffuse); set_num(env, stats, "nCaustic", st.n_caustic);
  set_num(env, stats, "capped", st.capped); set_num(env, stats, "errors", st.errors);
  napi_set_named_property(env, out, "stats", stats);
  return out;
}

static napi_value map_handle(napi_env env, CtxBox *cb, rt_photon_map *m) {
  MapBox *box = new MapBox{ MapBox::TAG, cb, m };
  cb->maps.insert(box);
  return make_handle(env, box);
}

// photonMapBuild(ctx, rec Float64Array(n*11)) -> map
static napi_value PhotonMapBuild(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  rt_ctx *ctx = cb->ctx;
  size_t n = 0;
  const double *rec = (const double *)a.typed(1, napi_float64_array, &n);
  if (n % 11 != 0) { napi_throw_range_error(env, nullptr, "photonMapBuild: rec must hold 11 values per photon"); return nullptr; }
  rt_photon_map *m = nullptr;
  int rc = rt_photon_map_build(ctx, rec, (uint32_t)(n / 11), &m);
  if (rc != RT_OK) return fail(env, ctx, rc);
  return map_handle(env, cb, m);
}

// photonMapNodes(map) -> {perm: Int32Array, dim: Uint8Array}
static napi_value PhotonMapNodes(napi_env env, napi_callback_info info) {
  Args a(env, info);
  MapBox *mb = a.ext<MapBox>(0);
  if (!mb) return nullptr;
  uint32_t n = rt_photon_map_size(mb->map);
  napi_value abp, abd, perm, dim, out; void *pd, *dd;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 4, &pd, &abp)); NAPI_OK(napi_create_typedarray(env, napi_int32_array, n, abp, 0, &perm));
  NAPI_OK(napi_create_arraybuffer(env, n, &dd, &abd)); NAPI_OK(napi_create_typedarray(env, napi_uint8_array, n, abd, 0, &dim));
  int rc = rt_photon_map_nodes(mb->map, (int32_t *)pd, (uint8_t *)dd);
  if (rc != RT_OK) return fail(env, mb->cb->ctx, rc);
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "perm", perm); napi_set_named_property(env, out, "dim", dim);
  return out;
}

// kdRadius(map, queries Float32Array(nq*3), r, cap) -> {count: Uint32Array, idx: Int32Array(nq*cap)}
static napi_value KdRadius(napi_env env, napi_callback_info info) {
  Args a(env, info);
  MapBox *mb = a.ext<MapBox>(0);
  size_t nq3 = 0;
  const float *q = (const float *)a.typed(1, napi_float32_array, &nq3);
  if (!mb) return nullptr;
  if (!q) { napi_throw_type_error(env, nullptr, "kdRadius(map, queries, r, cap)"); return nullptr; }
  uint32_t nq = (uint32_t)(nq3 / 3), cap = (uint32_t)a.num(3, 64);
  napi_value abc, abi, cnt, idx, out; void *cd, *id;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nq * 4, &cd, &abc)); NAPI_OK(napi_create_typedarray(env, napi_uint32_array, nq, abc, 0, &cnt));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nq * cap * 4, &id, &abi)); NAPI_OK(napi_create_typedarray(env, napi_int32_array, (size_t)nq * cap, abi, 0, &idx));
  int rc = rt_kd_radius(mb->map, q, nq, a.num(2, 0), cap, (uint32_t *)cd, (int32_t *)id);
  if (rc != RT_OK) return fail(env, mb->cb->ctx, rc);
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "count", cnt); napi_set_named_property(env, out, "idx", idx);
  return out;
}

// kdKnn(map, queries Float32Array(nq*3), k) -> {idx: Int32Array(nq*k), d2: Float64Array(nq*k)}
static napi_value KdKnn(napi_env env, napi_callback_info info) {
  Args a(env, info);
  MapBox *mb = a.ext<MapBox>(0);
  size_t nq3 = 0;
  const float *q = (const float *)a.typed(1, napi_float32_array, &nq3);
  if (!mb) return nullptr;
  if (!q) { napi_throw_type_error(env, nullptr, "kdKnn(map, queries, k)"); return nullptr; }
  uint32_t nq = (uint32_t)(nq3 / 3), k = (uint32_t)a.num(2, 64);
  napi_value abi, abd, idx, d2, out; void *id, *dd;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nq * k * 4, &id, &abi)); NAPI_OK(napi_create_typedarray(env, napi_int32_array, (size_t)nq * k, abi, 0, &idx));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nq * k * 8, &dd, &abd)); NAPI_OK(napi_create_typedarray(env, napi_float64_array, (size_t)nq * k, abd, 0, &d2));
  int rc = rt_kd_knn(mb->map, q, nq, k, (int32_t *)id, (double *)dd);
  if (rc != RT_OK) return fail(env, mb->cb->ctx, rc);
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "idx", idx); napi_set_named_property(env, out, "d2", d2);
  return out;
}

// kdNearestPath(map, queries Float32Array(nq*3)) -> {idx: Int32Array(nq), dist: Float64Array(nq)}   (kdtree.js:10-62)
static napi_value KdNearestPath(napi_env env, napi_callback_info info) {
  Args a(env, info);
  MapBox *mb = a.ext<MapBox>(0);
  size_t nq3 = 0;
  const float *q = (const float *)a.typed(1, napi_float32_array, &nq3);
  if (!mb) return nullptr;
  if (!q) { napi_throw_type_error(env, nullptr, "kdNearestPath(map, queries)"); return nullptr; }
  uint32_t nq = (uint32_t)(nq3 / 3);
  napi_value abi, abd, idx, dist, out; void *id, *dd;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nq * 4, &id, &abi)); NAPI_OK(napi_create_typedarray(env, napi_int32_array, nq, abi, 0, &idx));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nq * 8, &dd, &abd)); NAPI_OK(napi_create_typedarray(env, napi_float64_array, nq, abd, 0, &dist));
  int rc = rt_kd_nearest_path(mb->map, q, nq, (int32_t *)id, (double *)dd);
  if (rc != RT_OK) return fail(env, mb->cb->ctx, rc);
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "idx", idx); napi_set_named_property(env, out, "dist", dist);
  return out;
}

// gatherRadiance(map, hits Float32Array(nq*9), r, k) -> Float32Array(nq*3)
static napi_value GatherRadiance(napi_env env, napi_callback_info info) {
  Args a(env, info);
  MapBox *mb = a.ext<MapBox>(0);
  size_t n9 = 0;
  const float *h = (const float *)a.typed(1, napi_float32_array, &n9);
  if (!mb) return nullptr;
  if (!h) { napi_throw_type_error(env, nullptr, "gatherRadiance(map, hits, r, k)"); return nullptr; }
  uint32_t nq = (uint32_t)(n9 / 9);
  napi_value ab, out; void *data;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nq * 12, &data, &ab)); NAPI_OK(napi_create_typedarray(env, napi_float32_array, (size_t)nq * 3, ab, 0, &out));
  int rc = rt_gather_radiance(mb->map, h, nq, a.num(2, 0), (uint32_t)a.num(3, 0), (float *)data);
  if (rc != RT_OK) return fail(env, mb->cb->ctx, rc);
  return out;
}

// gatherRadianceAdaptive(map, hits Float32Array(nq*9), r0, grow, minCount, minKeep, maxAttempts) -> {rgb: Float32Array(nq*3), radius: Float32Array(nq)}
static napi_value GatherRadianceAdaptive(napi_env env, napi_callback_info info) {
  Args a(env, info);
  MapBox *mb = a.ext<MapBox>(0);
  size_t n9 = 0;
  const float *h = (const float *)a.typed(1, napi_float32_array, &n9);
  if (!mb) return nullptr;
  if (!h) { napi_throw_type_error(env, nullptr, "gatherRadianceAdaptive(map, hits, r0, grow, minCount, minKeep, maxAttempts)"); return nullptr; }
  uint32_t nq = (uint32_t)(n9 / 9);
  napi_value ab, ab2, rgb, rad, out; void *data, *data2;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nq * 12, &data, &ab)); NAPI_OK(napi_create_typedarray(env, napi_float32_array, (size_t)nq * 3, ab, 0, &rgb));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nq * 4, &data2, &ab2)); NAPI_OK(napi_create_typedarray(env, napi_float32_array, (size_t)nq, ab2, 0, &rad));
  int rc = rt_gather_radiance_adaptive(mb->map, h, nq, a.num(2, 0.1), a.num(3, 3), (uint32_t)a.num(4, 10), (uint32_t)a.num(5, 3), (uint32_t)a.num(6, 10),
                                       (float *)data, (float *)data2);
  if (rc != RT_OK) return fail(env, mb->cb->ctx, rc);
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "rgb", rgb); napi_set_named_property(env, out, "radius", rad);
  return out;
}

// traceTile(scene, params, diffuseMap|null, causticMap|null, x0, y0, w, h) -> {pre, post: Float64Array(w*h*4), rgba8: Uint8Array, stats}
// traceTile(scene, params, diffuseMap|null, causticMap|null, x0, y0, w, h, {pre, post, rgba8: device buffers or absent}) -> undefined:
//   the frame is only QUEUED (rt_trace_tile without stats into device memory); its pixels are complete after synchronize(ctx)
//   or a deviceRead of the buffer, and up to setFramesInFlight(ctx, n) such frames overlap on the GPU
static napi_value TraceTile(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  MapBox *md = a.ext<MapBox>(2), *mc = a.ext<MapBox>(3);
  if (pending(env)) return nullptr;
  if ((md && md->cb != sb->cb) || (mc && mc->cb != sb->cb)) { napi_throw_error(env, nullptr, "traceTile: the maps belong to another context"); return nullptr; }
  int x0 = (int)a.num(4), y0 = (int)a.num(5), w = (int)a.num(6), h = (int)a.num(7);
  if (w <= 0 || h <= 0) { napi_throw_range_error(env, nullptr, "traceTile: empty tile"); return nullptr; }
  const int blk = p.super_sample_block > 1 ? p.super_sample_block : 1;
  const size_t rows = p.band_count > 1 ? rt_band_rows((uint32_t)h, p.band_rows, p.band_count, p.band_index) : (size_t)h;   // (a rank's share of the tile)
  size_t npx = (size_t)((w + blk - 1) / blk) * (size_t)((rows + blk - 1) / blk);      // one value per block
  if (a.has(8)) {
    void *dev[3] = { nullptr, nullptr, nullptr };
    const char *names[3] = { "pre", "post", "rgba8" };
    const size_t need[3] = { npx * 32, npx * 32, npx * 4 };
    for (int i = 0; i < 3; i++) {
      bool has = false; napi_has_named_property(env, a.argv[8], names[i], &has);
      if (!has) continue;
      napi_value v = get_prop(env, a.argv[8], names[i]); napi_valuetype t; napi_typeof(env, v, &t);
      if (t == napi_undefined || t == napi_null) continue;
      Args one(env, v);
      DevBox *b = one.ext<DevBox>(0);
      if (!b) return nullptr;
      if (b->cb != sb->cb) { napi_throw_error(env, nullptr, "traceTile: an output buffer belongs to another context"); return nullptr; }
      if (b->bytes < need[i]) { napi_throw_range_error(env, nullptr, "traceTile: an output buffer is smaller than the tile (32 bytes per pixel for pre / post, 4 for rgba8)"); return nullptr; }
      dev[i] = b->ptr;
    }
    if (!dev[0] && !dev[1] && !dev[2]) { napi_throw_type_error(env, nullptr, "traceTile: the output object names no device buffer (pre, post, rgba8)"); return nullptr; }
    if (npx == 0) return nullptr;
    int rc = rt_trace_tile(sb->scene, &p, md ? md->map : nullptr, mc ? mc->map : nullptr, x0, y0, w, h, (double *)dev[0], (double *)dev[1], (uint8_t *)dev[2], nullptr);
    if (rc != RT_OK) return fail(env, sb->cb->ctx, rc);
    return nullptr;
  }
  napi_value ab1, ab2, ab3, pre, post, r8, out, stats; void *d1, *d2, *d3;
  NAPI_OK(napi_create_arraybuffer(env, npx * 32, &d1, &ab1)); NAPI_OK(napi_create_typedarray(env, napi_float64_array, npx * 4, ab1, 0, &pre));
  NAPI_OK(napi_create_arraybuffer(env, npx * 32, &d2, &ab2)); NAPI_OK(napi_create_typedarray(env, napi_float64_array, npx * 4, ab2, 0, &post));
  NAPI_OK(napi_create_arraybuffer(env, npx * 4, &d3, &ab3)); NAPI_OK(napi_create_typedarray(env, napi_uint8_array, npx * 4, ab3, 0, &r8));
  rt_trace_stats st; memset(&st, 0, sizeof st);
  if (npx) {
    int rc = rt_trace_tile(sb->scene, &p, md ? md->map : nullptr, mc ? mc->map : nullptr, x0, y0, w, h, (double *)d1, (double *)d2, (uint8_t *)d3, &st);
    if (rc != RT_OK) return fail(env, sb->cb->ctx, rc);
  }
  NAPI_OK(napi_create_object(env, &out)); NAPI_OK(napi_create_object(env, &stats));
  napi_set_named_property(env, out, "pre", pre); napi_set_named_property(env, out, "post", post); napi_set_named_property(env, out, "rgba8", r8);
  set_num(env, stats, "rays", (double)st.rays); set_num(env, stats, "shadowRays", (double)st.shadow_rays);
  set_num(env, stats, "gathers", (double)st.gathers); set_num(env, stats, "bounces", (double)st.bounces);
  set_num(env, stats, "tailRays", (double)st.tail_rays); set_num(env, stats, "primTests", (double)st.prim_tests);
  set_num(env, stats, "stackOverflows", (double)st.stack_overflows); set_num(env, stats, "gatherSaturations", (double)st.gather_saturations);
  napi_set_named_property(env, out, "stats", stats);
  return out;
}

// an optional typed-array property of an options object (absent / null / undefined -> nullptr); sets *bad on a wrong type or length
static void *opt_typed(napi_env env, napi_value o, const char *name, napi_typedarray_type want, size_t need, bool *bad) {
  napi_valuetype ot; napi_typeof(env, o, &ot);
  if (ot != napi_object) return nullptr;
  bool has = false; napi_has_named_property(env, o, name, &has);
  if (!has) return nullptr;
  napi_value v = get_prop(env, o, name); napi_valuetype t; napi_typeof(env, v, &t);
  if (t == napi_undefined || t == napi_null) return nullptr;
  size_t n = 0;
  void *p = prop_typed(env, o, name, want, &n);
  if (!p || n != need) { *bad = true; return nullptr; }
  return p;
}
static napi_value make_typed(napi_env env, napi_typedarray_type type, size_t n, size_t elem, void **data) {
  napi_value ab, out;
  if (napi_create_arraybuffer(env, n * elem, data, &ab) != napi_ok || napi_create_typedarray(env, type, n, ab, 0, &out) != napi_ok) return nullptr;
  return out;
}

// traceRays(scene, params, diffuseMap|null, causticMap|null, rayO, rayD, {maxDist, depth, importance, stacks, firstStream, firstDraw}) -> Float64Array(n*4)
static napi_value TraceRays(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  MapBox *md = a.ext<MapBox>(2), *mc = a.ext<MapBox>(3);
  { bool pending = false; napi_is_exception_pending(env, &pending); if (pending) return nullptr; }
  if ((md && md->cb != sb->cb) || (mc && mc->cb != sb->cb)) { napi_throw_error(env, nullptr, "traceRays: the maps belong to another context"); return nullptr; }
  size_t no = 0, nd = 0;
  const double *o = (const double *)a.typed(4, napi_float64_array, &no), *d = (const double *)a.typed(5, napi_float64_array, &nd);
  if (!o || !d || no != nd || no % 3) { napi_throw_range_error(env, nullptr, "traceRays: rayO and rayD must be Float64Arrays of n*3 values"); return nullptr; }
  const uint32_t n = (uint32_t)(no / 3);
  napi_value opt = a.has(6) ? a.argv[6] : nullptr;
  bool bad = false;
  const double *maxd = opt ? (const double *)opt_typed(env, opt, "maxDist", napi_float64_array, n, &bad) : nullptr;
  const int32_t *depth = opt ? (const int32_t *)opt_typed(env, opt, "depth", napi_int32_array, n, &bad) : nullptr;
  const double *imp = opt ? (const double *)opt_typed(env, opt, "importance", napi_float64_array, n, &bad) : nullptr;
  const uint8_t *stacks = opt ? (const uint8_t *)opt_typed(env, opt, "stacks", napi_uint8_array, (size_t)n * 16, &bad) : nullptr;
  const uint32_t *fd = opt ? (const uint32_t *)opt_typed(env, opt, "firstDraw", napi_uint32_array, n, &bad) : nullptr;
  if (bad) { napi_throw_range_error(env, nullptr, "traceRays: an option array has the wrong type or length"); return nullptr; }
  const uint32_t first = opt ? (uint32_t)prop_num(env, opt, "firstStream", 0) : 0;
  void *data; napi_value out = make_typed(env, napi_float64_array, (size_t)n * 4, 8, &data);
  if (!out) { napi_throw_error(env, nullptr, "N-API call failed: typed array"); return nullptr; }
  if (n == 0) return out;
  int rc = rt_trace_rays(sb->scene, &p, md ? md->map : nullptr, mc ? mc->map : nullptr, n, o, d, maxd, depth, imp, stacks, first, fd, (double *)data, nullptr);
  if (rc != RT_OK) return fail(env, sb->cb->ctx, rc);
  return out;
}

// directLight(scene, params, points, normals, localRgb, firstStream, firstDraw|null) -> {rgb: Float64Array(n*3), draws: Uint32Array(n)}
static napi_value DirectLight(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  size_t np_ = 0, nn = 0, nl = 0, nf = 0;
  const double *pt = (const double *)a.typed(2, napi_float64_array, &np_), *nr = (const double *)a.typed(3, napi_float64_array, &nn);
  const double *lc = (const double *)a.typed(4, napi_float64_array, &nl);
  const uint32_t *fd = (const uint32_t *)a.typed(6, napi_uint32_array, &nf);
  if (!pt || !nr || !lc || np_ != nn || np_ != nl || np_ % 3 || (fd && nf != np_ / 3)) { napi_throw_range_error(env, nullptr, "directLight: array sizes disagree"); return nullptr; }
  const uint32_t n = (uint32_t)(np_ / 3);
  void *rd, *dd; napi_value rgb = make_typed(env, napi_float64_array, (size_t)n * 3, 8, &rd), draws = make_typed(env, napi_uint32_array, n, 4, &dd), out;
  if (!rgb || !draws) { napi_throw_error(env, nullptr, "N-API call failed: typed array"); return nullptr; }
  if (n) { int rc = rt_direct_light(sb->scene, &p, n, pt, nr, lc, (uint32_t)a.num(5, 0), fd, (double *)rd, (uint32_t *)dd); if (rc != RT_OK) return fail(env, sb->cb->ctx, rc); }
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "rgb", rgb); napi_set_named_property(env, out, "draws", draws);
  return out;
}

// createPhotons(scene, params, n, phase, firstStream, source Int32Array|null) -> {photons: Float64Array(n*11), draws: Uint32Array(n)}
static napi_value CreatePhotons(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  const uint32_t n = (uint32_t)a.num(2, 0);
  size_t ns = 0;
  const int32_t *src = (const int32_t *)a.typed(5, napi_int32_array, &ns);
  if (src && ns != n) { napi_throw_range_error(env, nullptr, "createPhotons: source must hold n entries"); return nullptr; }
  void *pd, *dd; napi_value ph = make_typed(env, napi_float64_array, (size_t)n * 11, 8, &pd), draws = make_typed(env, napi_uint32_array, n, 4, &dd), out;
  if (!ph || !draws) { napi_throw_error(env, nullptr, "N-API call failed: typed array"); return nullptr; }
  if (n) { int rc = rt_create_photons(sb->scene, &p, n, (int)a.num(3, 1), (uint64_t)a.num(4, 0), src, (double *)pd, (uint32_t *)dd); if (rc != RT_OK) return fail(env, sb->cb->ctx, rc); }
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "photons", ph); napi_set_named_property(env, out, "draws", draws);
  return out;
}

// walkPhotons(scene, params, photons Float64Array(n*11), phase, firstStream, {firstDraw, depth, caustic, stacks, trackOnlyCaustic})
//   -> {photons: Float64Array(n*11) (stored records where kind != 0), kind: Uint8Array(n)}
static napi_value WalkPhotons(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  size_t n11 = 0;
  const double *in = (const double *)a.typed(2, napi_float64_array, &n11);
  if (!in || n11 % 11) { napi_throw_range_error(env, nullptr, "walkPhotons: photons must hold 11 values each"); return nullptr; }
  const uint32_t n = (uint32_t)(n11 / 11);
  napi_value opt = a.has(5) ? a.argv[5] : nullptr;
  bool bad = false;
  const uint32_t *fd = opt ? (const uint32_t *)opt_typed(env, opt, "firstDraw", napi_uint32_array, n, &bad) : nullptr;
  const int32_t *depth = opt ? (const int32_t *)opt_typed(env, opt, "depth", napi_int32_array, n, &bad) : nullptr;
  const int8_t *ca = opt ? (const int8_t *)opt_typed(env, opt, "caustic", napi_int8_array, n, &bad) : nullptr;
  const uint8_t *stacks = opt ? (const uint8_t *)opt_typed(env, opt, "stacks", napi_uint8_array, (size_t)n * 16, &bad) : nullptr;
  if (bad) { napi_throw_range_error(env, nullptr, "walkPhotons: an option array has the wrong type or length"); return nullptr; }
  const int only = opt ? (int)prop_num(env, opt, "trackOnlyCaustic", 0) : 0;
  void *pd, *kd; napi_value ph = make_typed(env, napi_float64_array, n11, 8, &pd), kind = make_typed(env, napi_uint8_array, n, 1, &kd), out;
  if (!ph || !kind) { napi_throw_error(env, nullptr, "N-API call failed: typed array"); return nullptr; }
  if (n) {
    memcpy(pd, in, n11 * 8);
    int rc = rt_walk_photons(sb->scene, &p, n, (int)a.num(3, 1), (uint64_t)a.num(4, 0), fd, depth, ca, stacks, only, (double *)pd, (uint8_t *)kd);
    if (rc != RT_OK) return fail(env, sb->cb->ctx, rc);
  }
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "photons", ph); napi_set_named_property(env, out, "kind", kind);
  return out;
}


// ---- the throughput path: device buffers, photon lists that stay in GPU memory, queued frames, shards ----

static size_t elem_size(napi_typedarray_type t) {
  switch (t) {
    case napi_int8_array: case napi_uint8_array: case napi_uint8_clamped_array: return 1;
    case napi_int16_array: case napi_uint16_array: return 2;
    case napi_int32_array: case napi_uint32_array: case napi_float32_array: return 4;
    default: return 8;
  }
}
// argument i as memory: a typed array of element type `want` (host memory, borrowed for the call; it may sit on a
// SharedArrayBuffer) or a device buffer of context cb -> pointer and size in bytes; throws and returns nullptr otherwise
static void *mem_arg(const Args &a, size_t i, napi_typedarray_type want, CtxBox *cb, size_t *bytes, const char *who) {
  *bytes = 0;
  if (!a.has(i)) { napi_throw_type_error(a.env, nullptr, (std::string(who) + ": a typed array or a device buffer is expected").c_str()); return nullptr; }
  napi_valuetype t; napi_typeof(a.env, a.argv[i], &t);
  if (t == napi_external) {
    DevBox *b = a.ext<DevBox>(i);
    if (!b) return nullptr;
    if (cb && b->cb != cb) { napi_throw_error(a.env, nullptr, (std::string(who) + ": the device buffer belongs to another context").c_str()); return nullptr; }
    *bytes = b->bytes;
    return b->ptr;
  }
  size_t n = 0;
  void *p = a.typed(i, want, &n);
  if (!p && !pending(a.env)) napi_throw_range_error(a.env, nullptr, (std::string(who) + ": empty array").c_str());
  *bytes = n * elem_size(want);
  return p;
}

// deviceAlloc(ctx, bytes) -> buffer
static napi_value DeviceAlloc(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  const double bytes = a.num(1, 0);
  if (!(bytes >= 1) || bytes > 9.0e15) { napi_throw_range_error(env, nullptr, "deviceAlloc(ctx, bytes): bytes must be positive"); return nullptr; }
  void *ptr = nullptr;
  int rc = rt_device_alloc(cb->ctx, (uint64_t)bytes, &ptr);
  if (rc != RT_OK) return fail(env, cb->ctx, rc);
  DevBox *box = new DevBox{ DevBox::TAG, cb, ptr, (size_t)bytes };
  cb->bufs.insert(box);
  return make_handle(env, box);
}
// deviceFree(buffer): now rather than when the collector gets to it
static napi_value DeviceFree(napi_env env, napi_callback_info info) {
  Args a(env, info);
  DevBox *b = a.ext<DevBox>(0);
  if (b) b->drop();
  return nullptr;
}
static napi_value DeviceBytes(napi_env env, napi_callback_info info) {
  Args a(env, info);
  DevBox *b = a.ext<DevBox>(0);
  if (!b) return nullptr;
  napi_value out; napi_create_double(env, (double)b->bytes, &out);
  return out;
}
// deviceRead(buffer, out typed array[, byteOffset]) -> out;  deviceWrite(buffer, src typed array[, byteOffset])
// Both wait for everything the context has queued (a queued frame can be read straight after queueing it).
static napi_value device_io(napi_env env, napi_callback_info info, bool read) {
  Args a(env, info);
  DevBox *b = a.ext<DevBox>(0);
  if (!b) return nullptr;
  bool is = false;
  if (a.has(1)) napi_is_typedarray(env, a.argv[1], &is);
  if (!is) { napi_throw_type_error(env, nullptr, read ? "deviceRead(buffer, out, byteOffset)" : "deviceWrite(buffer, src, byteOffset)"); return nullptr; }
  napi_typedarray_type t; size_t n; void *data; napi_value ab; size_t off;
  napi_get_typedarray_info(env, a.argv[1], &t, &n, &data, &ab, &off);
  const size_t bytes = n * elem_size(t);
  const double at = a.num(2, 0);
  if (at < 0 || at + (double)bytes > (double)b->bytes) { napi_throw_range_error(env, nullptr, "device buffer access out of range"); return nullptr; }
  char *dev = (char *)b->ptr + (size_t)at;
  int rc = read ? rt_device_copy(b->cb->ctx, data, dev, bytes) : rt_device_copy(b->cb->ctx, dev, data, bytes);
  if (rc != RT_OK) return fail(env, b->cb->ctx, rc);
  return read ? a.argv[1] : nullptr;
}
static napi_value DeviceRead(napi_env env, napi_callback_info info) { return device_io(env, info, true); }
static napi_value DeviceWrite(napi_env env, napi_callback_info info) { return device_io(env, info, false); }

// setOption(ctx, name, value) / setFramesInFlight(ctx, n) / join(ctx) / synchronize(ctx) / stream(ctx) -> BigInt (the hipStream_t)
static napi_value SetOption(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  char name[64]; size_t len = 0;
  if (!a.has(1) || napi_get_value_string_utf8(env, a.argv[1], name, sizeof name, &len) != napi_ok) { napi_throw_type_error(env, nullptr, "setOption(ctx, name, value)"); return nullptr; }
  int rc = rt_set_option(cb->ctx, name, a.num(2, 0));
  if (rc != RT_OK) return fail(env, cb->ctx, rc);
  return nullptr;
}
static napi_value SetFramesInFlight(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  int rc = rt_set_frames_in_flight(cb->ctx, (int)a.num(1, 1));
  if (rc != RT_OK) return fail(env, cb->ctx, rc);
  return nullptr;
}
static napi_value Join(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  int rc = rt_join(cb->ctx);
  if (rc != RT_OK) return fail(env, cb->ctx, rc);
  return nullptr;
}
static napi_value Synchronize(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  int rc = rt_synchronize(cb->ctx);
  if (rc != RT_OK) return fail(env, cb->ctx, rc);
  return nullptr;
}
static napi_value Stream(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  napi_value out; NAPI_OK(napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)rt_stream(cb->ctx), &out));
  return out;
}
// bandRows(height, bandRows, bandCount, bandIndex) -> rows of a tile that one rank's interleaved bands hold
static napi_value BandRows(napi_env env, napi_callback_info info) {
  Args a(env, info);
  napi_value out; napi_create_uint32(env, rt_band_rows((uint32_t)a.num(0, 0), (int)a.num(1, 0), (int)a.num(2, 0), (int)a.num(3, 0)), &out);
  return out;
}

static napi_value emit_stats_object(napi_env env, const rt_emit_stats &st) {
  napi_value stats; napi_create_object(env, &stats);
  set_num(env, stats, "attemptsPhase1", (double)st.attempts_phase1); set_num(env, stats, "attemptsPhase2", (double)st.attempts_phase2);
  set_num(env, stats, "segments", (double)st.segments); set_num(env, stats, "nDiffuse", st.n_diffuse); set_num(env, stats, "nCaustic", st.n_caustic);
  set_num(env, stats, "capped", st.capped); set_num(env, stats, "errors", st.errors);
  return stats;
}
static napi_value lists_handle(napi_env env, CtxBox *cb, rt_photon_lists *L) {
  ListsBox *box = new ListsBox{ ListsBox::TAG, cb, L };
  cb->lists.insert(box);
  return make_handle(env, box);
}

// emitManyLists(scene, params, format) -> {lists, stats}: emitManyPhotons with both lists left in GPU memory
// (format 0: full records in the working precision, 1: the 36-byte photons of the map)
static napi_value EmitManyLists(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  rt_photon_lists *L = nullptr; rt_emit_stats st; memset(&st, 0, sizeof st);
  int rc = rt_emit_many_lists(sb->scene, &p, (int)a.num(2, RT_SHARD_FULL), &L, &st);
  if (rc != RT_OK && !(rc == RT_ERR_ATTEMPT_CAP && L)) { if (L) rt_lists_free(L); return fail(env, sb->cb->ctx, rc); }   // the cap only logs in the reference (rayphoton.js:1366)
  napi_value out, lists = lists_handle(env, sb->cb, L);
  if (!lists) return nullptr;
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "lists", lists); napi_set_named_property(env, out, "stats", emit_stats_object(env, st));
  return out;
}
// listsCreate(ctx, params, format) -> lists (empty, with the caps numPhotonsDiffuse / numPhotonsCaustic)
static napi_value ListsCreate(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  rt_photon_lists *L = nullptr;
  int rc = rt_lists_create(cb->ctx, &p, (int)a.num(2, RT_SHARD_FULL), &L);
  if (rc != RT_OK) return fail(env, cb->ctx, rc);
  return lists_handle(env, cb, L);
}
// listsFree(lists): now rather than at collection
static napi_value ListsFree(napi_env env, napi_callback_info info) {
  Args a(env, info);
  ListsBox *lb = a.ext<ListsBox>(0);
  if (lb) lb->drop();
  return nullptr;
}
static napi_value ListsRecordWords(napi_env env, napi_callback_info info) {
  Args a(env, info);
  ListsBox *lb = a.ext<ListsBox>(0);
  if (!lb) return nullptr;
  napi_value out; napi_create_uint32(env, rt_lists_record_words(lb->lists), &out);
  return out;
}
// listsCounts(lists) -> {nDiffuse, nCaustic}
static napi_value ListsCounts(napi_env env, napi_callback_info info) {
  Args a(env, info);
  ListsBox *lb = a.ext<ListsBox>(0);
  if (!lb) return nullptr;
  uint32_t nd = 0, nc = 0;
  int rc = rt_lists_counts(lb->lists, &nd, &nc);
  if (rc != RT_OK) return fail(env, lb->cb->ctx, rc);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  set_num(env, out, "nDiffuse", nd); set_num(env, out, "nCaustic", nc);
  return out;
}
// listsExport(lists, which) -> {records: Float64Array(n*11), attempts: Float64Array(n*2) (phase, attempt)}; full-format lists
static napi_value ListsExport(napi_env env, napi_callback_info info) {
  Args a(env, info);
  ListsBox *lb = a.ext<ListsBox>(0);
  if (!lb) return nullptr;
  const int which = (int)a.num(1, 0);
  uint32_t n = 0, words = 0;
  if (rt_lists_view(lb->lists, which, nullptr, &n, &words, nullptr) != RT_OK) { napi_throw_range_error(env, nullptr, "listsExport(lists, which): which is 0 (diffuse) or 1 (caustic)"); return nullptr; }
  void *rd, *ad; napi_value rec = make_typed(env, napi_float64_array, (size_t)n * 11, 8, &rd), att = make_typed(env, napi_float64_array, (size_t)n * 2, 8, &ad), out;
  if (!rec || !att) { napi_throw_error(env, nullptr, "N-API call failed: typed array"); return nullptr; }
  std::vector<int64_t> tmp((size_t)n * 2);
  int rc = rt_lists_export(lb->lists, which, (double *)rd, tmp.data());
  if (rc != RT_OK) return fail(env, lb->cb->ctx, rc);
  for (size_t i = 0; i < tmp.size(); i++) ((double *)ad)[i] = (double)tmp[i];
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "records", rec); napi_set_named_property(env, out, "attempts", att);
  return out;
}
// listsRecords(lists, which) -> {n, words, precision, records: Float32Array(n*words) | Float64Array(n*words/2)}: the list's records as
// they lie in GPU memory (rt_lists_view), copied out
static napi_value ListsRecords(napi_env env, napi_callback_info info) {
  Args a(env, info);
  ListsBox *lb = a.ext<ListsBox>(0);
  if (!lb) return nullptr;
  const void *dev = nullptr; uint32_t n = 0, words = 0; int prec = 0;
  if (rt_lists_view(lb->lists, (int)a.num(1, 0), &dev, &n, &words, &prec) != RT_OK) { napi_throw_range_error(env, nullptr, "listsRecords(lists, which): which is 0 (diffuse) or 1 (caustic)"); return nullptr; }
  const bool f64 = prec == RT_PRECISION_F64 && words == 22;
  void *data; napi_value rec = f64 ? make_typed(env, napi_float64_array, (size_t)n * 11, 8, &data) : make_typed(env, napi_float32_array, (size_t)n * words, 4, &data), out;
  if (!rec) { napi_throw_error(env, nullptr, "N-API call failed: typed array"); return nullptr; }
  int rc = rt_device_copy(lb->cb->ctx, data, dev, (uint64_t)n * words * 4);
  if (rc != RT_OK) return fail(env, lb->cb->ctx, rc);
  NAPI_OK(napi_create_object(env, &out));
  set_num(env, out, "n", n); set_num(env, out, "words", words); set_num(env, out, "precision", prec);
  napi_set_named_property(env, out, "records", rec);
  return out;
}
// emitShard(scene, params, phase, attemptBegin, attemptEnd, format, tagBase, out Uint32Array | device buffer) -> {n, words, stats}
// One rank's share of a batch of emission attempts: the stored photons of [attemptBegin, attemptEnd), in attempt order, as
// records of words + 1 uint32 in `out` (room for attemptEnd - attemptBegin records always suffices).
static napi_value EmitShard(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  const int phase = (int)a.num(2, 1), format = (int)a.num(5, RT_SHARD_PACKED);
  const double b0 = a.num(3, 0), b1 = a.num(4, 0), tb = a.num(6, 0);
  if (!(b0 >= 0) || !(b1 >= b0) || !(tb >= 0) || b1 > 9.0e15) { napi_throw_range_error(env, nullptr, "emitShard: bad attempt range"); return nullptr; }
  const uint32_t W = format == RT_SHARD_PACKED ? 9u : (p.precision == RT_PRECISION_F64 ? 22u : 11u);
  size_t bytes = 0;
  uint32_t *out = (uint32_t *)mem_arg(a, 7, napi_uint32_array, sb->cb, &bytes, "emitShard");
  if (!out) return nullptr;
  const size_t cap = bytes / ((size_t)(W + 1) * 4);
  if (cap == 0 || cap > 0x7FFFFFFFu) { napi_throw_range_error(env, nullptr, "emitShard: the output holds no record (or more than 2^31)"); return nullptr; }
  uint32_t n = 0; rt_emit_stats st; memset(&st, 0, sizeof st);
  int rc = rt_emit_shard(sb->scene, &p, phase, (uint64_t)b0, (uint64_t)b1, format, (uint64_t)tb, (uint32_t)cap, out, &n, &st);
  if (rc != RT_OK) return fail(env, sb->cb->ctx, rc);
  napi_value res; NAPI_OK(napi_create_object(env, &res));
  set_num(env, res, "n", n); set_num(env, res, "words", W);
  napi_set_named_property(env, res, "stats", emit_stats_object(env, st));
  return res;
}
// listsApply(lists, phase, tagBase, shards Uint32Array | device buffer, counts Uint32Array(nShards), stride) -> info
// One merged batch: shard s starts at record s * stride of `shards` and holds counts[s] records, shards in attempt order.
static napi_value ListsApply(napi_env env, napi_callback_info info) {
  Args a(env, info);
  ListsBox *lb = a.ext<ListsBox>(0);
  if (!lb) return nullptr;
  const double tb = a.num(2, 0), stride = a.num(5, 0);
  size_t bytes = 0, nc = 0;
  const uint32_t *shards = (const uint32_t *)mem_arg(a, 3, napi_uint32_array, lb->cb, &bytes, "listsApply");
  if (!shards) return nullptr;
  const uint32_t *counts = (const uint32_t *)a.typed(4, napi_uint32_array, &nc);
  const uint32_t W = rt_lists_record_words(lb->lists);
  if (!counts || nc == 0 || !(stride >= 1) || !(tb >= 0) || stride > 2147483647.0) { napi_throw_range_error(env, nullptr, "listsApply(lists, phase, tagBase, shards, counts, stride)"); return nullptr; }
  if ((double)nc * stride * (W + 1) * 4 > (double)bytes) { napi_throw_range_error(env, nullptr, "listsApply: shards holds fewer than counts.length * stride records"); return nullptr; }
  for (size_t i = 0; i < nc; i++) if (counts[i] > (uint32_t)stride) { napi_throw_range_error(env, nullptr, "listsApply: a count exceeds the stride"); return nullptr; }
  rt_lists_info li; memset(&li, 0, sizeof li);
  int rc = rt_lists_apply(lb->lists, (int)a.num(1, 1), (uint64_t)tb, shards, counts, (uint32_t)nc, (uint32_t)stride, &li);
  if (rc != RT_OK) return fail(env, lb->cb->ctx, rc);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  set_num(env, out, "nDiffuse", li.n_diffuse); set_num(env, out, "nCaustic", li.n_caustic);
  set_num(env, out, "batchDiffuse", li.batch_diffuse); set_num(env, out, "batchCaustic", li.batch_caustic);
  set_num(env, out, "takenDiffuse", li.taken_diffuse); set_num(env, out, "takenCaustic", li.taken_caustic);
  set_num(env, out, "phaseEnded", li.phase_ended); set_num(env, out, "lastAttempt", (double)li.last_attempt);
  return out;
}
// emitPhotons(scene, params, phase, attemptBegin, attemptEnd) -> {records: Float64Array(n*11), attempts: Float64Array(n), kind: Uint8Array(n), stats}
// every stored photon of the attempt range, no caps (kind 1 diffuse, 2 caustic)
static napi_value EmitPhotons(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  const double b0 = a.num(3, 0), b1 = a.num(4, 0);
  if (!(b0 >= 0) || !(b1 >= b0) || b1 - b0 > 16777216.0) { napi_throw_range_error(env, nullptr, "emitPhotons: the range must hold at most 2^24 attempts"); return nullptr; }
  const uint32_t cap = (uint32_t)(b1 - b0);
  std::vector<double> rec((size_t)cap * 11); std::vector<int64_t> att(cap); std::vector<uint8_t> kind(cap);
  uint32_t n = 0; rt_emit_stats st; memset(&st, 0, sizeof st);
  if (cap) { int rc = rt_emit_photons(sb->scene, &p, (int)a.num(2, 1), (uint64_t)b0, (uint64_t)b1, cap, rec.data(), att.data(), kind.data(), &n, &st); if (rc != RT_OK) return fail(env, sb->cb->ctx, rc); }
  void *rd, *ad, *kd; napi_value r = make_typed(env, napi_float64_array, (size_t)n * 11, 8, &rd), at = make_typed(env, napi_float64_array, n, 8, &ad), k = make_typed(env, napi_uint8_array, n, 1, &kd), out;
  if (!r || !at || !k) { napi_throw_error(env, nullptr, "N-API call failed: typed array"); return nullptr; }
  if (n) { memcpy(rd, rec.data(), (size_t)n * 88); memcpy(kd, kind.data(), n); }
  for (uint32_t i = 0; i < n; i++) ((double *)ad)[i] = (double)att[i];
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "records", r); napi_set_named_property(env, out, "attempts", at); napi_set_named_property(env, out, "kind", k);
  napi_set_named_property(env, out, "stats", emit_stats_object(env, st));
  return out;
}
// photonMapBuildLists(lists, which) -> map: the kd-tree of a list without its photons leaving the GPU
static napi_value PhotonMapBuildLists(napi_env env, napi_callback_info info) {
  Args a(env, info);
  ListsBox *lb = a.ext<ListsBox>(0);
  if (!lb) return nullptr;
  rt_photon_map *m = nullptr;
  int rc = rt_photon_map_build_lists(lb->lists, (int)a.num(1, 0), &m);
  if (rc != RT_OK) return fail(env, lb->cb->ctx, rc);
  return map_handle(env, lb->cb, m);
}
// photonMapBuildPacked(ctx, photons Float32Array(n*9) | device buffer[, n]) -> map
static napi_value PhotonMapBuildPacked(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  size_t bytes = 0;
  const float *ph = (const float *)mem_arg(a, 1, napi_float32_array, cb, &bytes, "photonMapBuildPacked");
  if (!ph) return nullptr;
  const double n = a.has(2) ? a.num(2, 0) : (double)(bytes / 36);
  if (!(n >= 0) || n * 36 > (double)bytes || (!a.has(2) && bytes % 36)) { napi_throw_range_error(env, nullptr, "photonMapBuildPacked: 9 floats per photon"); return nullptr; }
  rt_photon_map *m = nullptr;
  int rc = rt_photon_map_build_packed(cb->ctx, ph, (uint32_t)n, &m);
  if (rc != RT_OK) return fail(env, cb->ctx, rc);
  return map_handle(env, cb, m);
}
static napi_value PhotonMapSize(napi_env env, napi_callback_info info) {
  Args a(env, info);
  MapBox *mb = a.ext<MapBox>(0);
  if (!mb) return nullptr;
  napi_value out; napi_create_uint32(env, rt_photon_map_size(mb->map), &out);
  return out;
}
// kdBuild(ctx, positions Float32Array(n*3)) -> {perm: Int32Array(n), dim: Uint8Array(n)}: the tree alone
static napi_value KdBuild(napi_env env, napi_callback_info info) {
  Args a(env, info);
  CtxBox *cb = a.ext<CtxBox>(0);
  if (!cb) return nullptr;
  size_t n3 = 0;
  const float *pos = (const float *)a.typed(1, napi_float32_array, &n3);
  if (!pos || n3 % 3) { if (!pending(env)) napi_throw_range_error(env, nullptr, "kdBuild(ctx, positions): 3 floats per point"); return nullptr; }
  const uint32_t n = (uint32_t)(n3 / 3);
  void *pd, *dd; napi_value perm = make_typed(env, napi_int32_array, n, 4, &pd), dim = make_typed(env, napi_uint8_array, n, 1, &dd), out;
  if (!perm || !dim) { napi_throw_error(env, nullptr, "N-API call failed: typed array"); return nullptr; }
  int rc = rt_kd_build(cb->ctx, pos, n, (int32_t *)pd, (uint8_t *)dd);
  if (rc != RT_OK) return fail(env, cb->ctx, rc);
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "perm", perm); napi_set_named_property(env, out, "dim", dim);
  return out;
}
// gatherRadianceStatic(map, hits device buffer (nq*9 floats), nq, r, k, rgb device buffer (nq*3 floats)): the gather that
// allocates nothing
static napi_value GatherRadianceStatic(napi_env env, napi_callback_info info) {
  Args a(env, info);
  MapBox *mb = a.ext<MapBox>(0);
  DevBox *hits = a.ext<DevBox>(1), *rgb = a.ext<DevBox>(5);
  if (!mb || !hits || !rgb) { if (!pending(env)) napi_throw_type_error(env, nullptr, "gatherRadianceStatic(map, hitsBuffer, nq, r, k, rgbBuffer)"); return nullptr; }
  if (hits->cb != mb->cb || rgb->cb != mb->cb) { napi_throw_error(env, nullptr, "gatherRadianceStatic: the buffers belong to another context"); return nullptr; }
  const double nq = a.num(2, 0);
  if (!(nq >= 0) || nq * 36 > (double)hits->bytes || nq * 12 > (double)rgb->bytes) { napi_throw_range_error(env, nullptr, "gatherRadianceStatic: nq exceeds the buffers (36 bytes per hit, 12 per result)"); return nullptr; }
  int rc = rt_gather_radiance_static(mb->map, (const float *)hits->ptr, (uint32_t)nq, a.num(3, 0), (uint32_t)a.num(4, 0), (float *)rgb->ptr);
  if (rc != RT_OK) return fail(env, mb->cb->ctx, rc);
  return nullptr;
}

static napi_value FindMedian(napi_env env, napi_callback_info info) {
  Args a(env, info);
  napi_value out; napi_create_uint32(env, rt_find_median((uint32_t)a.num(0, 1)), &out);
  return out;
}

static napi_value ModuleInit(napi_env env, napi_value exports) {
  const struct { const char *name; napi_callback fn; } fns[] = {
    { "init", Init }, { "shutdown", Shutdown }, { "sceneUpload", SceneUpload }, { "intersectPairs", IntersectPairs }, { "intersect", Intersect },
    { "emitMany", EmitMany }, { "photonMapBuild", PhotonMapBuild }, { "photonMapNodes", PhotonMapNodes },
    { "kdRadius", KdRadius }, { "kdKnn", KdKnn }, { "kdNearestPath", KdNearestPath }, { "gatherRadiance", GatherRadiance }, { "gatherRadianceAdaptive", GatherRadianceAdaptive }, { "traceTile", TraceTile },
    { "traceRays", TraceRays }, { "directLight", DirectLight }, { "createPhotons", CreatePhotons }, { "walkPhotons", WalkPhotons },
    { "findMedian", FindMedian },
    { "deviceAlloc", DeviceAlloc }, { "deviceFree", DeviceFree }, { "deviceBytes", DeviceBytes }, { "deviceRead", DeviceRead }, { "deviceWrite", DeviceWrite },
    { "setOption", SetOption }, { "setFramesInFlight", SetFramesInFlight }, { "join", Join }, { "synchronize", Synchronize }, { "stream", Stream }, { "bandRows", BandRows },
    { "emitManyLists", EmitManyLists }, { "listsCreate", ListsCreate }, { "listsFree", ListsFree }, { "listsRecordWords", ListsRecordWords }, { "listsCounts", ListsCounts },
    { "listsExport", ListsExport }, { "listsRecords", ListsRecords }, { "emitShard", EmitShard }, { "listsApply", ListsApply }, { "emitPhotons", EmitPhotons },
    { "photonMapBuildLists", PhotonMapBuildLists }, { "photonMapBuildPacked", PhotonMapBuildPacked }, { "photonMapSize", PhotonMapSize }, { "kdBuild", KdBuild },
    { "gatherRadianceStatic", GatherRadianceStatic },
  };
  for (const auto &f : fns) {
    napi_value fn; napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn);
    napi_set_named_property(env, exports, f.name, fn);
  }
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, ModuleInit)
