// Samplers of the photon pipeline: orthonormal frame, cosine-weighted perturbation, uniform
// disc sample, light choice (vecOrthonormal / vecPerturb / discSample / createPhoton,
// rayphoton.js:113-137, :1921-1926, :1326-1344).  Templated on the working precision.
#pragma once
#include "rt_geom.h"
#include "rt_philox.h"

template <typename T> RT_D T two_pi() { return T(2) * T(3.141592653589793); }

// sin and cos of the angle 2 pi u, u in [0, 1): the reference's `Math.sin(2 * Math.PI * Math.random())` pairs
// (rayphoton.js:134-136, :1923-1925).  fp64: exactly that expression (the parity build lands on the reference's bits).
// fp32: the hardware's v_sin_f32 / v_cos_f32, which take their argument in turns -- two instructions instead of the
// library routines' range reduction and polynomials (about a hundred), absolute error below 1e-6, inside the fp32
// build's stated tolerance.
RT_D void sincos_turns(double u, double &s, double &c) { const double phi = two_pi<double>() * u; s = rt_sin(phi); c = rt_cos(phi); }
RT_D void sincos_turns(float u, float &s, float &c) { s = __builtin_amdgcn_sinf(u); c = __builtin_amdgcn_cosf(u); }

template <typename T> RT_D void orthonormal(V3<T> v, T eps, V3<T> &m, V3<T> &n) {
  V3<T> v2 = normalize<T>(v, eps);
  V3<T> m0 = mk<T>(1, 0, 0);
  if (sqlen(cross(v2, m0)) < eps) m0 = mk<T>(0, 1, 0);
  n = normalize<T>(cross(v2, m0), eps);
  m = cross(n, v2);
}

// cosine-weighted direction around v (maxDeviation undefined: lower bound 0). 2 draws.
template <typename T, typename RNG> RT_D V3<T> vec_perturb(V3<T> v, T eps, RNG &rng) {
  V3<T> m, n; orthonormal<T>(v, eps, m, n);
  T x = rng.template next<T>();
  T cosT = rt_sqrt(x);
  T sinT = rt_sqrt(T(1) - cosT * cosT);
  T sinP, cosP; sincos_turns(rng.template next<T>(), sinP, cosP);
  return (cosT * v + (sinT * cosP) * m) + (sinT * sinP) * n;
}

// uniform point on a disc. 2 draws.
template <typename T, typename RNG> RT_D V3<T> disc_sample(V3<T> centre, T radius, V3<T> normal_dir, T eps, RNG &rng) {
  T r = rt_sqrt(rng.template next<T>()) * radius;
  T sinT, cosT; sincos_turns(rng.template next<T>(), sinT, cosT);
  V3<T> m, n; orthonormal<T>(normal_dir, eps, m, n);
  return (centre + (r * cosT) * m) + (r * sinT) * n;
}

template <typename T> struct PhotonState {
  V3<T> o, d;
  T power, r, g, b, a;
  int caustic;      // -1 undefined, 0 false, 1 true (rayphoton.js:652)
};

// createPhoton(source) (rayphoton.js:1326-1344): without a source (< 0) the light is chosen by walking the cumulative
// probabilities; the draw for that choice is made either way.  Always 5 draws.  Returns the light index (clamped
// where the reference would run off the end of the list and throw).
template <typename T, typename RNG> RT_D int create_photon(const DevScene<T> &S, T eps, RNG &rng, PhotonState<T> &ph, int source = -1) {
  T r = rng.template next<T>();
  uint32_t l = 0;
  if (source >= 0) l = (uint32_t)source;
  else while (l < S.n_lights && r > S.light_data[l][2]) { r -= S.light_data[l][2]; l++; }
  if (l >= S.n_lights) l = S.n_lights - 1;
  const T *lg = S.geom + (size_t)S.light_shape[l] * RT_GEOM_STRIDE;
  V3<T> centre = ld3(lg), nrm = ld3(lg + 3);
  ph.o = disc_sample<T>(centre, lg[6], nrm, eps, rng);
  ph.d = normalize<T>(vec_perturb<T>(nrm, eps, rng), eps);     // Photon extends Ray: the ctor normalises
  ph.power = S.light_data[l][0];
  ph.r = S.light_data[l][3]; ph.g = S.light_data[l][4]; ph.b = S.light_data[l][5]; ph.a = S.light_data[l][6];
  ph.caustic = -1;
  return (int)l;
}

// Material stack of a path: up to 16 entries of 3 bits (material ids 0..7) and a length,
// packed in one 64-bit word.  The reference grows a JS array (rayphoton.js:1514-1516, :1724-1727).
struct MatStack {
  uint64_t bits;
  RT_D void init() { bits = (uint64_t)1 << 56; /* [MAT_AIR] : len 1, entry 0 = 0 */ }
  RT_D int len() const { return (int)(bits >> 56); }
  RT_D int at(int i) const { return (int)((bits >> (3 * i)) & 7); }
  RT_D int top() const { return at(len() - 1); }
  RT_D void pop() { bits -= (uint64_t)1 << 56; }
  RT_D bool push(int m) {
    int n = len();
    if (n >= 16) return false;
    bits &= ~((uint64_t)7 << (3 * n));
    bits |= (uint64_t)(m & 7) << (3 * n);
    bits += (uint64_t)1 << 56;
    return true;
  }
};

// Fresnel decision shared by traceRay (rayphoton.js:1679-1732) and emitPhoton (:1484-1521).
// Returns 0 when eta2 is undefined (caller decides what that means), 1 otherwise with the new
// direction in `ndir` (not yet normalised) and the stack updated.  Consumes one draw unless the
// surface reflects totally.
template <typename T, typename RNG>
RT_D int fresnel_bounce(const DevScene<T> &S, int mat, V3<T> d, V3<T> n, T cos1, MatStack &st, RNG &rng,
                        V3<T> &ndir, bool &overflow) {
  // (the indices differ from lane to lane and the table sits in the kernel arguments: selects, not indexed loads)
  T eta1 = pick8<T>(S.mat[1], st.top());
  T eta2;
  if (dot(d, n) < T(0)) eta2 = pick8<T>(S.mat[1], mat);
  else if (st.len() >= 2) eta2 = pick8<T>(S.mat[1], st.at(st.len() - 2));
  else return 0;
  if (!(eta2 == eta2)) return 0;
  T ratio = eta1 / eta2;
  T c2sq = T(1) - ratio * ratio * (T(1) - cos1 * cos1);
  if (c2sq < T(0)) { ndir = d + (T(2) * cos1) * n; return 1; }          // total internal reflection
  T cos2 = rt_sqrt(c2sq);
  T a1 = (eta1 * cos1 - eta2 * cos2) / (eta1 * cos1 + eta2 * cos2);
  T a2 = (eta1 * cos2 - eta2 * cos1) / (eta1 * cos2 + eta2 * cos1);
  T rS = a1 * a1, rP = a2 * a2;
  if (cos1 < T(0)) { rS = T(1) / rS; rP = T(1) / rP; }
  if (rng.template next<T>() < T(0.5) * (rS + rP)) { ndir = d + (T(2) * cos1) * n; return 1; }
  T pm = (cos1 < T(0)) ? T(-1) : T(1);
  ndir = ratio * d + (ratio * cos1 - pm * cos2) * n;
  if (cos1 < T(0)) st.pop();
  else if (!st.push(mat)) overflow = true;
  return 1;
}
