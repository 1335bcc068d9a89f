// Small vector algebra, templated on the working precision (float for throughput, double for
// the parity build).  Operation order follows the reference's helpers (rayphoton.js:104-112)
// so that the fp64 instantiation, compiled without FMA contraction, reproduces its rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define RT_D __device__ __forceinline__

template <typename T> struct V3 { T x, y, z; };
template <typename T> RT_D V3<T> mk(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename T> RT_D V3<T> ld3(const T *p) { return mk<T>(p[0], p[1], p[2]); }
template <typename T> RT_D V3<T> operator+(V3<T> a, V3<T> b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> RT_D V3<T> operator-(V3<T> a, V3<T> b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> RT_D V3<T> operator*(T k, V3<T> a) { return mk<T>(k * a.x, k * a.y, k * a.z); }
template <typename T> RT_D T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> RT_D V3<T> cross(V3<T> v, V3<T> w) {
  return mk<T>(v.y * w.z - v.z * w.y, v.z * w.x - v.x * w.z, v.x * w.y - v.y * w.x);
}
template <typename T> RT_D T sqlen(V3<T> a) { return dot(a, a); }

RT_D float rt_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }      // one instruction (1 ulp; sqrtf's range scaling costs six more)
RT_D double rt_sqrt(double x) { return sqrt(x); }
RT_D float rt_abs(float x) { return fabsf(x); }
RT_D double rt_abs(double x) { return fabs(x); }
RT_D float rt_fmin(float a, float b) { return fminf(a, b); }
RT_D double rt_fmin(double a, double b) { return fmin(a, b); }
RT_D float rt_fmax(float a, float b) { return fmaxf(a, b); }
RT_D double rt_fmax(double a, double b) { return fmax(a, b); }
RT_D float rt_floor(float x) { return floorf(x); }
RT_D double rt_floor(double x) { return floor(x); }
RT_D float rt_fmod(float x, float y) { return fmodf(x, y); }
RT_D double rt_fmod(double x, double y) { return fmod(x, y); }

// vecNormalize (rayphoton.js:109-110): near-zero vectors become [0,0,1]
template <typename T> RT_D V3<T> normalize(V3<T> a, T eps) {
  T s = sqlen(a);
  if (s < eps) return mk<T>(0, 0, 1);
  if constexpr (sizeof(T) == 4) return __builtin_amdgcn_rsqf(s) * a;
  else return (T(1) / rt_sqrt(s)) * a;
}

template <typename T> struct Limits;
template <> struct Limits<float> { static RT_D float inf() { return __builtin_huge_valf(); } static RT_D float nan() { return __builtin_nanf(""); } };
template <> struct Limits<double> { static RT_D double inf() { return __builtin_huge_val(); } static RT_D double nan() { return __builtin_nan(""); } };
