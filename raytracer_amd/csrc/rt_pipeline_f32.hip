// fp32 instantiation of the pipeline: the throughput build (FMA contraction on).
#include "rt_pipeline_impl.h"
RT_INSTANTIATE_PIPELINE(float)

// One (ray, shape) pair per lane through the fp32 PRODUCT intersection code: the precomputed scan record, the origin
// terms and isect_fast<TYPE> -- what scan_fast, scan_fast_visible and the hierarchy walk evaluate.
__global__ void __launch_bounds__(256) k_intersect_pairs_fast(uint32_t n, const double *o, const double *d, int normalize_dir, const uint8_t *meta,
                                                              const float *rec, float eps, double *t_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3<float> ro = mk<float>((float)o[3 * i], (float)o[3 * i + 1], (float)o[3 * i + 2]);
  V3<float> rd = mk<float>((float)d[3 * i], (float)d[3 * i + 1], (float)d[3 * i + 2]);
  if (normalize_dir) rd = normalize<float>(rd, eps);
  float f[RT_GEOM_STRIDE];
  for (int k = 0; k < RT_GEOM_STRIDE; k++) f[k] = rec[(size_t)i * RT_GEOM_STRIDE + k];
  const float t = isect_fast_any((int)meta[4 * i], f, ro, rd, eps, (int)meta[4 * i + 2]);
  t_out[i] = (t > eps) ? (double)t : __builtin_nan("");          // the scans accept a distance iff t > eps
}
int pl_intersect_pairs_fast(rt_ctx *ctx, const rt_params *p, uint32_t n, const double *o, const double *d, int normalize_dir,
                            const uint8_t *meta, const float *fast_rec, double *t_out) {
  RT_LAUNCH(ctx, "intersect_pairs_fast", k_intersect_pairs_fast, div_up(n, 256), 256, n, o, d, normalize_dir, meta, fast_rec,
            (float)working_eps<float>(p), t_out);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

// ---- stand-alone radius gathers through the request queue
// hits: nq x 9 floats (point, normal, albedo 0..255); request i serves query i ("path" i)
__global__ void __launch_bounds__(256) k_requests_from_hits(const float *hits, uint32_t nq, float4 *gq, uint32_t *valid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  valid[i] = 1u;
  const float *h = hits + (size_t)i * 9;
  gq[(size_t)i * 3] = make_float4(h[0], h[1], h[2], __uint_as_float(i));
  gq[(size_t)i * 3 + 1] = make_float4(h[3], h[4], h[5], __uint_as_float(1u));          // the "diffuse" slot of the request; bounce 0
  gq[(size_t)i * 3 + 2] = make_float4(h[6] / 255.0f, h[7] / 255.0f, h[8] / 255.0f, 0.f);
}
__global__ void __launch_bounds__(256) k_accumulators_to_rgb(const long long *gacc, uint32_t n3, float *rgb) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n3) rgb[i] = (float)((double)gacc[i] / RT_GACC_SCALE);
}
int pl_gather_radius_queued(rt_photon_map *m, DevMem &ws, DevMem &sort_tmp, const float *hits_dev, uint32_t nq, double r, float *rgb_dev) {
  rt_ctx *ctx = m->ctx;
  const RtKnobs &K = ctx->knobs;
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const uint32_t items_cap = (uint32_t)std::min<size_t>(std::max<size_t>(nq, 1u << 16), 1u << 22);
  const uint32_t items_shard = (items_cap / RT_ITEM_SHARDS) * 2u;
  const size_t o_hdr = carve(RT_GQ_HEADER_BYTES), o_gq = carve((size_t)nq * 48), o_valid = carve((size_t)nq * 4), o_gacc = carve((size_t)nq * 24), o_kin = carve((size_t)nq * 8), o_kout = carve((size_t)nq * 8),
               o_vin = carve((size_t)nq * 4), o_vout = carve((size_t)nq * 4), o_items = carve((size_t)items_shard * RT_ITEM_SHARDS * 8);
  RT_HIP(ctx, ws.reserve(off));
  char *base = (char *)ws.p;
  TraceCfg C = {};
  C.diffuse_radius = r; C.caustic_radius = r; C.n_paths = nq; C.wave_gather_ratio = (float)K.wave_gather;
  C.gq_count = (uint32_t *)(base + o_hdr); C.gq = (float4 *)(base + o_gq); C.gq_cap = nq; C.gacc = (long long *)(base + o_gacc);
  C.gq_valid = (uint32_t *)(base + o_valid); C.gq_first = nq;         // request i sits in slot i
  RT_HIP(ctx, hipMemsetAsync(C.gq_count, 0, RT_GQ_HEADER_BYTES, ctx->stream));
  RT_HIP(ctx, hipMemsetAsync(C.gacc, 0, (size_t)nq * 24, ctx->stream));
  RT_LAUNCH(ctx, "gather_requests", k_requests_from_hits, div_up(nq, 256), 256, hits_dev, nq, C.gq, C.gq_valid);
  unsigned long long *kin = (unsigned long long *)(base + o_kin), *kout = (unsigned long long *)(base + o_kout);
  uint32_t *vin = (uint32_t *)(base + o_vin), *vout = (uint32_t *)(base + o_vout);
  RT_LAUNCH(ctx, "gather_keys", k_gather_keys, div_up(nq, 256), 256, C, (float)(RT_GATHER_CELLS_PER_RADIUS / r), kin, vin);
  { KernelTimer timer_(ctx, "gather_sort"); RT_TRY(rt_sort_pairs_u64(ctx, sort_tmp, kin, kout, vin, vout, nq)); }
  C.gq_order = vout;
  GatherItems items; items.items = (uint2 *)(base + o_items); items.count = C.gq_count + RT_ITEM_WORDS; items.cap = K.gather_split > 0 ? items_shard : 0; items.tag = 0; items.stage = nullptr; items.stage_n = 0; items.full = false;
  int split = 0;
  if (K.gather_split > 0) while (((uint64_t)m->n >> split) > (uint64_t)K.gather_split && split < 24) split++;
  DevMap none = {};
  RT_LAUNCH(ctx, "gather_deferred", k_gather_deferred, div_up(nq, 256), 256, C, m->view(), none, items, split, 0);
  if (items.cap) RT_LAUNCH(ctx, "gather_items", k_gather_items, 256 * RT_GATHER_ITEM_WGS, 256, C, m->view(), none, items, C.gq_count + RT_ITEM_WORDS * (1u + RT_ITEM_SHARDS));
  RT_LAUNCH(ctx, "gather_to_rgb", k_accumulators_to_rgb, div_up((uint64_t)nq * 3, 256), 256, (const long long *)C.gacc, nq * 3, rgb_dev);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

// ---- origin terms of the sorted primitives for the camera's position, once per uploaded scene
__global__ void __launch_bounds__(64) k_camera_origin_terms(DevScene<float> S, float *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S.n_shapes) return;
  int type = 0;
  while (type < RT_NUM_TYPES - 1 && i >= S.seg[type + 1]) type++;
  float rec[RT_GEOM_STRIDE], ot[RT_OT_STRIDE] = { 0.f, 0.f, 0.f, 0.f, 0.f };
  for (int k = 0; k < RT_GEOM_STRIDE; k++) rec[k] = S.sgeom[(size_t)i * RT_GEOM_STRIDE + k];
  origin_terms_any(type, rec, ld3(S.camera), ot);
  for (int k = 0; k < RT_OT_STRIDE; k++) out[(size_t)i * RT_OT_STRIDE + k] = ot[k];
}
int pl_camera_origin_terms(rt_ctx *ctx, const DevScene<float> &S, float *out) {
  RT_LAUNCH(ctx, "camera_origin_terms", k_camera_origin_terms, div_up(S.n_shapes, 64), 64, S, out);
  RT_HIP(ctx, hipGetLastError());
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}
