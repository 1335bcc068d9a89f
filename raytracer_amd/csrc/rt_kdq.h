// Device-side photon-map view and the stackless fixed-radius traversal shared by the query
// kernels (rt_kd.hip) and the shading kernel (rt_trace.h).
#pragma once
#include "rt_math.h"

// Photon map in heap order: slot s = pos - 1 of the reference's `nodes[pos]` (kdtree.js:104).
struct DevMap {
  uint32_t n;
  const float4 *node;     // x y z (fp32-rounded position) and the split dimension in w's bits
  const float *pay;       // 6 floats per slot: incoming direction, power * colour
  const double *rec64;    // 11 doubles per slot (origin3 dir3 power rgba) or nullptr
  const int32_t *perm;    // slot -> index in the list the map was built from
};

RT_D int node_dim(float4 nd) { return __float_as_int(nd.w); }

// nearestNeighbours(x, r) (kdtree.js:64-80) without recursion or a stack: the tree is an
// implicit left-balanced heap, so the only state is the current position, its depth and one
// bit per level saying "the right child of the ancestor at this level is still to be
// visited".  visit(slot, nd) is called for every node whose split plane the search ball
// straddles -- the nodes the reference tests for inclusion -- in the reference's pre-order
// (node, left subtree, right subtree).  All comparisons are done in fp64 on the exactly
// converted fp32 node coordinates, with the same operands as the reference's
// `x[dim] + r < p[dim]` and `x[dim] - r > p[dim]`.
template <typename F> RT_D void kd_radius_walk(const DevMap &M, double x0, double x1, double x2, double r, F &&visit) {
  const uint32_t n = M.n;
  uint32_t pos = 1; int depth = 0; uint64_t pending = 0;
  for (;;) {
    if (pos <= n) {
      float4 nd = M.node[pos - 1];
      int dim = node_dim(nd);
      double pd = (dim == 0) ? (double)nd.x : (dim == 1) ? (double)nd.y : (double)nd.z;
      double xd = (dim == 0) ? x0 : (dim == 1) ? x1 : x2;
      if (xd + r < pd) { pos = 2 * pos; depth++; continue; }
      if (xd - r > pd) { pos = 2 * pos + 1; depth++; continue; }
      visit(pos - 1, nd);
      pending |= (uint64_t)1 << depth;
      pos = 2 * pos; depth++;
      continue;
    }
    if (pending == 0) break;
    int k = 63 - __clzll((long long)pending);
    pending &= ~((uint64_t)1 << k);
    pos = ((pos >> (depth - k)) << 1) | 1u;
    depth = k + 1;
  }
}

RT_D double sq_dist(double x0, double x1, double x2, double p0, double p1, double p2) {
#pragma clang fp contract(off)      // three roundings for the products, two for the sums, as in the reference
  double dx = x0 - p0, dy = x1 - p1, dz = x2 - p2;
  return dx * dx + dy * dy + dz * dz;       // vecSqLength(vecMinus(x, point)), kdtree.js:75
}
