"""BASELINE configuration 5 at its size, on one GPU: a caustic photon map of 10^8 photons (emitManyPhotons,
rayphoton.js:1354-1383; makeTree, kdtree.js:82-107; the gathers of traceRay, rayphoton.js:1765-1815).

Nothing of that size can be compared with the CPU oracle photon by photon (its makeTree alone would run for minutes),
so the checks are size-independent properties plus exhaustive fp64 arithmetic done with torch on the GPU:
  * the lists that three simulated ranks produce through the shard / all-gather / cap protocol are the lists of the
    one-GPU loop, record for record (compared on the device),
  * the tree's `perm` is a permutation, and on sampled subtrees every node satisfies the definition of makeTree: split
    axis = widest axis of the subtree's bounding box (ties to the lowest axis), left descendants <= node <= right
    descendants along it, left-balanced sizes (findMedian),
  * k = 64 gathers at sampled hit points equal an exhaustive fp64 search over all photons (by (d^2, input index)),
  * radius gathers served through the request queue equal the lane-by-lane tree walk, and both the exhaustive sum.
"""
import numpy as np
import pytest

from golden_io import load_scene

pytestmark = pytest.mark.gpu

N_CAUSTIC = 100_000_000
N_DIFFUSE = 2_000_000
SEED = 1


@pytest.fixture(scope="module")
def c5():
    import torch
    from raytracer_amd import capi, parallel
    ctx = capi.Context(0)
    sc = ctx.upload(load_scene(10, 1920, 1080))
    p = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=N_DIFFUSE, num_photons_caustic=N_CAUSTIC, seed=SEED)
    lists, st = sc.emit_many_lists(p, capi.SHARD_PACKED, strict=False)
    assert st["rc"] == 0 and st["n_caustic"] == N_CAUSTIC and st["n_diffuse"] == N_DIFFUSE and st["errors"] < 1e-4 * st["attempts_phase2"]
    yield dict(capi=capi, parallel=parallel, torch=torch, ctx=ctx, sc=sc, p=p, lists=lists, stats=st)
    del lists
    ctx.close()


def test_three_simulated_ranks_produce_the_one_gpu_lists(c5):
    """emit_sharded(world_override=3): every round's attempts cut into three ranks' sub-ranges, each emitted into its own
    shard with 32-bit tags relative to the round's first attempt, merged in rank order and capped on the device -- dozens
    of rounds, hundreds of millions of attempts.  The lists must be rt_emit_many_lists's, bit for bit."""
    torch, capi, parallel = c5["torch"], c5["capi"], c5["parallel"]
    eng = parallel.DeviceEngine(c5["sc"], c5["p"], fmt=capi.SHARD_PACKED, device="cuda")
    T = {}
    res = parallel.emit_sharded(eng, batch=3 * (8 << 20), world_override=3, timings=T)
    st = c5["stats"]
    assert res["attempts1"] == st["attempts_phase1"] and res["attempts2"] == st["attempts_phase2"] and not res["capped"]
    assert res["n_diffuse"] == N_DIFFUSE and res["n_caustic"] == N_CAUSTIC and T["rounds"] >= 10
    assert res["attempts2"] > 100_000_000                         # (attempt numbers well beyond a round's 2^31 tag range matter: tags are relative)
    for which in (0, 1):
        a, b = c5["lists"].records_tensor(which), eng.lists.records_tensor(which)
        assert a.shape == b.shape == ((N_DIFFUSE, N_CAUSTIC)[which], 9)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), which          # bit patterns, NaNs and signed zeros included
        ta, tb = c5["lists"].attempts_tensor(which), eng.lists.attempts_tensor(which)
        assert torch.equal(ta, tb)
        # store order = attempt order within a phase, phase 1 before phase 2
        key = ta[:, 0] * (1 << 40) + ta[:, 1]
        assert bool((key[1:] > key[:-1]).all())
        del a, b, ta, tb, key
    del eng
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def c5_map(c5):
    torch = c5["torch"]
    m = c5["lists"].photon_map(1)
    rec = c5["lists"].records_tensor(1)                        # (n, 9) float32: position, direction, power * colour; input order
    yield m, rec
    m.close()
    torch.cuda.empty_cache()


def _subtree_slots(pos, n):
    """heap positions (1-based) of the subtree rooted at `pos` in a complete tree of n nodes, level by level"""
    out, lo, hi = [], pos, pos
    while lo <= n:
        out.append(np.arange(lo, min(hi, n) + 1))
        lo, hi = 2 * lo, 2 * hi + 1
    return np.concatenate(out)


def test_tree_of_1e8_photons_satisfies_the_definition_of_make_tree(c5, c5_map):
    torch = c5["torch"]
    m, rec = c5_map
    n = N_CAUSTIC
    assert m.n == n
    perm, dim = m.nodes(device="cuda")
    seen = torch.zeros(n, dtype=torch.bool, device="cuda")
    assert int(perm.min()) == 0 and int(perm.max()) == n - 1
    seen[perm.long()] = True
    assert bool(seen.all())                                     # a permutation: every photon is exactly one node
    del seen
    assert int(dim.max()) <= 2
    # (1) ordering along a node's split axis against RANDOM DESCENDANTS, 200 000 sampled nodes from every level
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    levels = int(np.floor(np.log2(n)))
    lvl = torch.randint(0, levels, (200_000,), generator=g, device="cuda")
    node = (torch.ones_like(lvl) << lvl) + (torch.rand(lvl.shape, generator=g, device="cuda", dtype=torch.float64) * (torch.ones_like(lvl) << lvl).double()).long()
    node = node.clamp(max=n // 2)                               # nodes with at least one child
    d_node = dim[node - 1].long()
    x_node = rec[perm[node - 1].long(), :].gather(1, d_node[:, None])[:, 0]
    checked = 0
    for side in (0, 1):
        cur = 2 * node + side
        alive = cur <= n
        for step in range(levels + 1):
            if not bool(alive.any()):
                break
            idx = torch.where(alive, cur, torch.ones_like(cur))
            x = rec[perm[idx - 1].long(), :].gather(1, d_node[:, None])[:, 0]
            ok = (x <= x_node) if side == 0 else (x >= x_node)
            assert bool((ok | ~alive).all()), (side, step)
            checked += int(alive.sum())
            cur = 2 * cur + (torch.rand(cur.shape, generator=g, device="cuda") < 0.5).long()
            alive = alive & (cur <= n)
    assert checked > 2_000_000
    # (2) every node of 48 sampled subtrees of 500 .. 4000 photons (and the root's axis) against the definition
    perm_h = None
    rng = np.random.default_rng(9)
    depth = levels - 11                                          # subtrees of 2^11 .. 2^12 photons
    roots = (1 << depth) + rng.choice(1 << depth, 48, replace=False)
    n_nodes = 0
    for r in roots:
        slots = _subtree_slots(int(r), n)
        sl = torch.from_numpy(slots).cuda()
        pts = rec[perm[sl - 1].long(), :3].double().cpu().numpy()
        dm = dim[sl - 1].cpu().numpy()
        where = {int(s): i for i, s in enumerate(slots)}
        stack = [int(r)]
        while stack:
            q = stack.pop()
            sub = _subtree_slots(q, n)
            ii = np.fromiter((where[int(s)] for s in sub), dtype=np.int64, count=sub.size)
            P = pts[ii]
            ext = P.max(axis=0) - P.min(axis=0)
            want = 0
            for k in (1, 2):                                     # kdtree.js:93-100: strictly wider wins, ties keep the lower axis
                if ext[k] > ext[want]:
                    want = k
            dq = int(dm[where[q]])
            assert dq == want, (int(r), q, ext.tolist(), dq)
            xq = pts[where[q], dq]
            for child, cmp in ((2 * q, np.less_equal), (2 * q + 1, np.greater_equal)):
                if child <= n:
                    cs = _subtree_slots(child, n)
                    ci = np.fromiter((where[int(s)] for s in cs), dtype=np.int64, count=cs.size)
                    assert cmp(pts[ci, dq], xq).all(), (int(r), q, child)
                    if cs.size > 1:
                        stack.append(child)
            n_nodes += 1
    assert n_nodes > 20_000
    del perm, dim, perm_h


def _hits(c5, n_q, seed):
    """hit records (position, normal, albedo) at sampled first-hit points of the 1080p frame, whole groups of 64"""
    from raytracer_amd import synth
    capi = c5["capi"]
    hp = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=SEED)
    hits = synth.first_hit_queries(c5["sc"], hp, 1920, 1080)
    rng = np.random.default_rng(seed)
    starts = rng.choice(hits.shape[0] // 64, n_q // 64, replace=False) * 64
    rows = (starts[:, None] + np.arange(64)[None, :]).reshape(-1)
    h = hits[rows].copy()
    nrm = rng.normal(size=(h.shape[0], 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    h[:, 3:6] = nrm                                             # arbitrary normals: the max(0, dir . n) term takes both branches
    h[:, 6:9] = rng.uniform(20, 255, (h.shape[0], 3))
    return np.ascontiguousarray(h.astype(np.float32))


class _Slabs:
    """exhaustive search made affordable: the photons sorted by x once; all photons within distance R of a query lie in
    the slab |x - qx| <= R, a contiguous range of that order"""

    def __init__(self, torch, rec):
        self.torch = torch
        self.x, self.order = torch.sort(rec[:, 0].contiguous())
        self.rec = rec

    def within(self, q, R):
        """input indices and fp64 squared distances (the device's operation order: dx*dx + dy*dy + dz*dz) of the photons of
        the slab [qx - R, qx + R]"""
        torch = self.torch
        lo = int(torch.searchsorted(self.x, torch.tensor([np.float32(q[0] - R)], device="cuda"), right=False))
        hi = int(torch.searchsorted(self.x, torch.tensor([np.float32(q[0] + R)], device="cuda"), right=True))
        idx = self.order[lo:hi]
        P = self.rec[idx, :3].double()
        qd = torch.tensor(q[:3].astype(np.float64), device="cuda")
        d = qd[None, :] - P
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        return idx, d2


def _estimate(torch, rec, idx, hit, area):
    pay = rec[idx].double()
    nrm = torch.tensor(hit[3:6].astype(np.float64), device="cuda")
    w = torch.clamp(pay[:, 3:6] @ nrm, min=0.0)
    s = (pay[:, 6:9] * w[:, None]).sum(dim=0).cpu().numpy()
    return s * hit[6:9].astype(np.float64) / 255.0 / area


def test_knn_gather_on_1e8_photons_equals_the_exhaustive_search(c5, c5_map):
    """k = 64 at 4 096 sampled hit points.  For each query the exhaustive search runs over the x-slab of half-width R around
    it; once the 64th smallest distance found there is <= R no photon outside the slab can be among the nearest, so the
    slab's answer is the answer over all 10^8 photons.  Ties at the k-th distance go to the lowest input index."""
    torch = c5["torch"]
    m, rec = c5_map
    K = 64
    hits = _hits(c5, 4096, seed=3)
    got = m.gather(torch.from_numpy(hits).cuda(), k=K, out=torch.empty((hits.shape[0], 3), dtype=torch.float32, device="cuda")).cpu().numpy()
    assert np.isfinite(got).all()
    slabs = _Slabs(torch, rec)
    exp = np.zeros((hits.shape[0], 3))
    worst_R = 0.0
    for i in range(hits.shape[0]):
        R = 0.004
        while True:
            idx, d2 = slabs.within(hits[i], R)
            if idx.numel() >= K:
                vals, _ = torch.topk(d2, K, largest=False)
                tau = float(vals[-1])
                if tau <= R * R:
                    break
            R *= 2.0
            assert R < 64.0
        worst_R = max(worst_R, R)
        less = d2 < tau
        tie = torch.nonzero(d2 == tau)[:, 0]
        need = K - int(less.sum())
        tie_idx, _ = torch.sort(idx[tie])                       # lowest input indices win the tie
        chosen = torch.cat([idx[less], tie_idx[:need]])
        assert chosen.numel() == K
        exp[i] = _estimate(torch, rec, chosen, hits[i], np.pi * tau)
    scale = max(1.0, float(np.abs(exp).max()))
    assert np.allclose(got, exp, rtol=3e-5, atol=1e-6 * scale), float(np.abs(got - exp).max())
    print(f"k = 64 on 1e8 photons: 4096 queries exhaustive; largest slab half-width {worst_R}")


def test_radius_gathers_on_1e8_photons(c5, c5_map):
    """the reference's own gather (fixed r = 0.15, rayphoton.js:1788-1815) on the 10^8-photon map: served through the request
    queue (sorted, 64 together, subtrees as work items) = the lane-by-lane tree walk to fp32 rounding, and both = the
    exhaustive fp64 sum on a sample"""
    torch, ctx = c5["torch"], c5["ctx"]
    m, rec = c5_map
    r = 0.15
    hits = _hits(c5, 32768, seed=4)
    hd = torch.from_numpy(hits).cuda()
    queued = m.gather(hd, r=r, out=torch.empty((hits.shape[0], 3), dtype=torch.float32, device="cuda")).cpu().numpy()
    with ctx.option("defer_gathers", 0, -1):
        lane = m.gather(hd[:4096].contiguous(), r=r, out=torch.empty((4096, 3), dtype=torch.float32, device="cuda")).cpu().numpy()
    scale = max(1.0, float(np.abs(lane).max()))
    assert np.isfinite(queued).all() and not np.array_equal(queued[:4096], lane)          # (it did take the other route)
    dev = np.abs(queued[:4096] - lane) / (np.abs(lane) + 2e-7 * scale / 5e-5)
    assert dev.max() < 5e-5, (float(dev.max()), np.unravel_index(dev.argmax(), dev.shape), queued[:4096].reshape(-1)[dev.argmax()], lane.reshape(-1)[dev.argmax()])
    slabs = _Slabs(torch, rec)
    n_in = []
    for i in range(0, 4096, 16):
        idx, d2 = slabs.within(hits[i], r)
        inside = d2 < r * r                                      # kdtree.js:75, strict
        exp = _estimate(torch, rec, idx[inside], hits[i], np.pi * r * r)
        n_in.append(int(inside.sum()))
        assert np.allclose(lane[i], exp, rtol=5e-5, atol=2e-7 * scale), (i, n_in[-1])
        assert np.allclose(queued[i], exp, rtol=5e-5, atol=2e-7 * scale), (i, n_in[-1])
    assert max(n_in) > 10_000                                    # dense balls were among the sample
    print(f"radius gather on 1e8 photons: photons in a ball mean {np.mean(n_in):.0f}, max {max(n_in)}")
