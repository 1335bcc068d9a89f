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
}

static void set_num(napi_env env, napi_value obj, const char *name, double v) { napi_value n; napi_create_double(env, v, &n); napi_set_named_property(env, obj, name, n); }

// Handles.  Scenes and maps point into their context, and the collector orders nothing: a scene may be finalised long
// after shutdown(ctx).  So the context's box is shared: it knows its live scenes and maps, shutdown() frees them first
// and marks them dead, and the box itself lives until the last handle that refers to it has been finalised.
struct SceneBox; struct MapBox;
struct CtxBox {
  uint64_t tag; rt_ctx *ctx; int refs; std::set<SceneBox *> scenes; std::set<MapBox *> maps;
  static constexpr uint64_t TAG = 0x7274333535637478ull;
  static const char *what() { return "rt355: a context handle (from init) is expected"; }
  bool live() const { return ctx != nullptr; }
};
struct SceneBox {
  uint64_t tag; CtxBox *cb; rt_scene *scene;
  static constexpr uint64_t TAG = 0x7274333535736365ull;
  static const char *what() { return "rt355: a scene handle (from sceneUpload) is expected"; }
  bool live() const { return scene != nullptr; }
};
struct MapBox {
  uint64_t tag; CtxBox *cb; rt_photon_map *map;
  static constexpr uint64_t TAG = 0x72743335356d6170ull;
  static const char *what() { return "rt355: a photon map handle (from photonMapBuild) is expected"; }
  bool live() const { return map != nullptr; }
};
static void ctx_close(CtxBox *cb) {                          // frees everything made from the context, then the context
  if (!cb->ctx) return;
  for (SceneBox *b : cb->scenes) { rt_scene_free(b->scene); b->scene = nullptr; }
  for (MapBox *b : cb->maps) { rt_photon_map_free(b->map); b->map = nullptr; }
  cb->scenes.clear(); cb->maps.clear();
  rt_shutdown(cb->ctx);
  cb->ctx = nullptr;
}
static void ctx_release(CtxBox *cb) { if (--cb->refs == 0) { ctx_close(cb); cb->tag = 0; delete cb; } }

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
  cb->scenes.insert(box); cb->refs++;
  napi_value out;
  auto fin = [](napi_env, void *p, void *) {
    SceneBox *b = (SceneBox *)p;
    if (b->scene) { rt_scene_free(b->scene); b->cb->scenes.erase(b); }
    ctx_release(b->cb); b->tag = 0; delete b;
  };
  if (napi_create_external(env, box, fin, nullptr, &out) != napi_ok) { fin(env, box, nullptr); napi_throw_error(env, nullptr, "N-API call failed: napi_create_external"); return nullptr; }
  return out;
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
  MapBox *box = new MapBox{ MapBox::TAG, cb, m };
  cb->maps.insert(box); cb->refs++;
  napi_value out;
  auto fin = [](napi_env, void *p, void *) {
    MapBox *b = (MapBox *)p;
    if (b->map) { rt_photon_map_free(b->map); b->cb->maps.erase(b); }
    ctx_release(b->cb); b->tag = 0; delete b;
  };
  if (napi_create_external(env, box, fin, nullptr, &out) != napi_ok) { fin(env, box, nullptr); napi_throw_error(env, nullptr, "N-API call failed: napi_create_external"); return nullptr; }
  return out;
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
static napi_value TraceTile(napi_env env, napi_callback_info info) {
  Args a(env, info);
  SceneBox *sb = a.ext<SceneBox>(0);
  if (!sb) return nullptr;
  rt_params p; read_params(env, a.argv[1], &p);
  MapBox *md = a.ext<MapBox>(2), *mc = a.ext<MapBox>(3);
  { bool pending = false; napi_is_exception_pending(env, &pending); if (pending) return nullptr; }
  if ((md && md->cb != sb->cb) || (mc && mc->cb != sb->cb)) { napi_throw_error(env, nullptr, "traceTile: the maps belong to another context"); return nullptr; }
  int x0 = (int)a.num(4), y0 = (int)a.num(5), w = (int)a.num(6), h = (int)a.num(7);
  if (w <= 0 || h <= 0) { napi_throw_range_error(env, nullptr, "traceTile: empty tile"); return nullptr; }
  const int blk = p.super_sample_block > 1 ? p.super_sample_block : 1;
  size_t npx = (size_t)((w + blk - 1) / blk) * (size_t)((h + blk - 1) / blk);      // one value per block
  napi_value ab1, ab2, ab3, pre, post, r8, out, stats; void *d1, *d2, *d3;
  NAPI_OK(napi_create_arraybuffer(env, npx * 32, &d1, &ab1)); NAPI_OK(napi_create_typedarray(env, napi_float64_array, npx * 4, ab1, 0, &pre));
  NAPI_OK(napi_create_arraybuffer(env, npx * 32, &d2, &ab2)); NAPI_OK(napi_create_typedarray(env, napi_float64_array, npx * 4, ab2, 0, &post));
  NAPI_OK(napi_create_arraybuffer(env, npx * 4, &d3, &ab3)); NAPI_OK(napi_create_typedarray(env, napi_uint8_array, npx * 4, ab3, 0, &r8));
  rt_trace_stats st;
  int rc = rt_trace_tile(sb->scene, &p, md ? md->map : nullptr, mc ? mc->map : nullptr, x0, y0, w, h, (double *)d1, (double *)d2, (uint8_t *)d3, &st);
  if (rc != RT_OK) return fail(env, sb->cb->ctx, rc);
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
  };
  for (const auto &f : fns) {
    napi_value fn; napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn);
    napi_set_named_property(env, exports, f.name, fn);
  }
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, ModuleInit)
