// Host-side drivers of the templated kernels; instantiated once per working precision
// (rt_pipeline_f32.hip with FMA contraction for throughput, rt_pipeline_f64.hip without it for
// bit parity with the reference's fp64 arithmetic).  All pointers are device pointers.
#pragma once
#include "rt_internal.h"

template <typename T> int pl_intersect_pairs(rt_ctx *ctx, const rt_params *p, uint32_t n, const double *o, const double *d,
                                             int normalize_dir, const uint8_t *meta, const double *geom, double *t_out);
// fp32 only: the same pairs through the records and distance functions the fp32 scan kernels use (rt_fast_record,
// origin_terms, isect_fast); fast_rec: n x 12 floats
int pl_intersect_pairs_fast(rt_ctx *ctx, const rt_params *p, uint32_t n, const double *o, const double *d, int normalize_dir,
                            const uint8_t *meta, const float *fast_rec, double *t_out);
// fp32: the origin terms of the scene's sorted primitives for rays that start at its camera (n_shapes x RT_OT_STRIDE
// floats), computed by the code the scan kernels would run themselves; synchronises the context's stream
int pl_camera_origin_terms(rt_ctx *ctx, const DevScene<float> &S, float *out);
template <typename T> int pl_intersect_scene(rt_scene *sc, const rt_params *p, int mode, uint32_t n, const double *o,
                                             const double *d, const double *source_dist, double *t_out, int32_t *idx_out);
template <typename T> int pl_trace_tile(rt_scene *sc, int slot, const rt_params *p, DevMap md, DevMap mc, int x0, int y0, int w, int h,
                                        double *pre, double *post, uint8_t *rgba8, rt_trace_stats *stats);
// explicit rays / points / photons, one lane each (the host's Scene.traceRay, directLight, createPhoton, emitPhoton)
template <typename T> int pl_trace_rays(rt_scene *sc, const rt_params *p, DevMap md, DevMap mc, uint32_t n, const double *o, const double *d,
                                        const double *max_dist, const int32_t *depth, const double *importance, const uint64_t *stacks,
                                        uint32_t first_stream, const uint32_t *first_draw, double *rgba, rt_trace_stats *stats);
template <typename T> int pl_direct_light(rt_scene *sc, const rt_params *p, uint32_t n, const double *point, const double *normal, const double *local,
                                          uint32_t first_stream, const uint32_t *first_draw, double *rgb, uint32_t *draws);
template <typename T> int pl_photons_explicit(rt_scene *sc, const rt_params *p, uint32_t n, int domain, uint64_t first_stream, const uint32_t *first_draw,
                                              int create, const int32_t *source, const int32_t *depth, const int8_t *caustic, const uint64_t *stacks,
                                              int track_only_caustic, double *io, uint8_t *kind, uint32_t *draws);
// runs attempts [a0, a0+count) of one phase into the scene's emission workspace
template <typename T> int pl_emit_batch(rt_scene *sc, const rt_params *p, int phase, uint64_t a0, uint32_t count);
// ordered compaction of the workspace: records whose kind matches `want` (0 = any stored) go to
// out_rec[out_base + rank]; *n_match = number of matches (device-side scan result, read back)
template <typename T> int pl_emit_compact(rt_scene *sc, uint32_t count, int want, uint64_t a0, uint32_t out_base, uint32_t out_cap,
                                          double *out_rec, int64_t *out_attempt, uint8_t *out_kind, uint32_t *n_match);
template <typename T> int pl_emit_shard(rt_scene *sc, uint32_t count, int format, uint32_t tag_offset, uint32_t cap, uint32_t *shard, uint32_t *n_dev);
int pl_emit_counters(rt_scene *sc, bool reset, uint64_t *segments, uint64_t *errors);
// rt_gather_radiance with k = 0 for many queries on a large map: the queries as gather requests through the queue, sort
// and work items of the frame's photon gathers (rt_trace.h).  ws: grow-only workspace of the map.
int pl_gather_radius_queued(rt_photon_map *m, DevMem &ws, DevMem &sort_tmp, const float *hits_dev, uint32_t nq, double r, float *rgb_dev);
