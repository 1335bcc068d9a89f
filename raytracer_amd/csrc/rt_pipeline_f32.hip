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
