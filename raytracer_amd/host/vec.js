'use strict';
// 3-vector helpers for the host scene layer (plain arrays, like the scene description the
// reference's callers already write).  normalise() keeps the reference's rule that a
// near-zero vector (|v|^2 < 1e-6) becomes [0,0,1] (rayphoton.js:109-110), because composite
// constructors rely on it.
const EPS = 1e-6;
const V = {
  add: (a, b) => [a[0] + b[0], a[1] + b[1], a[2] + b[2]],
  sub: (a, b) => [a[0] - b[0], a[1] - b[1], a[2] - b[2]],
  scale: (k, a) => [k * a[0], k * a[1], k * a[2]],
  dot: (a, b) => a[0] * b[0] + a[1] * b[1] + a[2] * b[2],
  cross: (a, b) => [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]],
  len: (a) => Math.sqrt(V.dot(a, a)),
  normalise(a) {
    const s = V.dot(a, a);
    return s < EPS ? [0, 0, 1] : V.scale(1 / Math.sqrt(s), a);
  },
  neg: (a) => [-1 * a[0], -1 * a[1], -1 * a[2]],
  sum: (...vs) => vs.reduce((acc, v) => V.add(acc, v)),
};
module.exports = V;
