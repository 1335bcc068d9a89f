"""GPU parity: the HIP path (through the C ABI, raytracer_amd/capi.py) against the reference's
golden vectors and against the CPU oracle on the same seeded inputs.

fp64 precision mode is held to bits (or to <= 1e-9 where a sum is reassociated); fp32 mode is
held to the stated fp32-vs-fp64 tolerance with discrete outputs exact away from decision
boundaries."""
import numpy as np
import pytest

from golden_io import ALL_PRESETS, RAYJS_PARAMS, load_bundle, load_packaged_preset, load_rayjs_scene, load_scene
from oracle import oracle as ora

pytestmark = pytest.mark.gpu
SEED = 12345


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.fixture(scope="module")
def rt():
    from raytracer_amd import capi
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


# ------------------------------------------------------------------ geometry kernel
def _pairs_inputs():
    b = load_bundle("intersect")
    meta = np.zeros((b["t"].shape[0], 4), np.uint8)
    meta[:, 0] = b["shape_meta"][:, 0]
    meta[:, 2] = b["shape_meta"][:, 1]
    return b, meta


def test_intersect_pairs_f64_bit_exact(rt):
    capi, ctx = rt
    b, meta = _pairs_inputs()
    p = capi.default_params(precision=capi.PRECISION_F64)
    t = ctx.intersect_pairs(p, b["ray_o"], b["ray_d"], meta, b["shape_g"])
    assert same(t, b["t"])
    t = ctx.intersect_pairs(p, b["ray_o"], b["ray_d_raw"], meta, b["shape_g"], normalize_dir=True)
    assert same(t, b["t"])


def test_intersect_pairs_f32_product_path(rt):
    """The reference's 12 288 (ray, shape) pairs through the fp32 PRODUCT code -- rt_fast_record + origin_terms +
    isect_fast<TYPE>, what every fp32 scan evaluates -- per primitive type.
    Stated tolerance: the fp32 build uses epsilon_f32 = 1e-4 where the reference uses 1e-6, so a hit/miss decision may
    differ from the reference's only (a) in the fixture's edge modes that aim at such a decision -- mode 4 (origin
    exactly on the surface: t ~ 0 against epsilon) and mode 6 (direction within 3e-6 of parallel to a planar shape:
    |d.n| against epsilon) -- or (b) where the reference's own answer lies within fp32 rounding of a boundary (hit
    distance under 2e-3, or a miss by less than that margin).  Everywhere else the decision must be the reference's
    and the distance within 2e-5 relative on 99 % of the pairs, 1e-4 on 99.9 % and 2e-3 on all (the quadratic of a
    cylinder or sphere met at a grazing angle cancels)."""
    capi, ctx = rt
    b, meta = _pairs_inputs()
    p = capi.default_params(precision=capi.PRECISION_F32)
    want = b["t"]
    types = load_bundle("intersect").meta["types"]
    per = load_bundle("intersect").meta["per_type"]
    mode = np.arange(want.shape[0]) % 16                   # gen_golden.js: mode = i % 16 within each type's block
    for raw in (False, True):
        t = ctx.intersect_pairs(p, b["ray_o"], b["ray_d_raw"] if raw else b["ray_d"], meta, b["shape_g"], normalize_dir=raw)
        flipped = np.isnan(t) != np.isnan(want)
        edge = (mode == 4) | (mode == 6)
        near_zero = ~np.isnan(want) & (want < 2e-3)        # the reference's own hit sits at the epsilon decision
        report, problems = {}, []
        for ti, name in enumerate(types):
            rows = slice(ti * per, (ti + 1) * per)
            f, e, nz = flipped[rows], edge[rows], near_zero[rows]
            nz = nz | (~np.isnan(want[rows]) & (want[rows] > 1e6))
            report[name] = (int(f.sum()), int((f & ~e & ~nz).sum()))
            # outside the edge modes a decision may differ only for grazing rays (hit point within fp32 rounding of the
            # primitive's rim, a discriminant within rounding of zero): a handful per type
            if (f & ~e & ~nz).sum() > 0.005 * per:
                problems.append(("decisions", name, report[name]))
            both = ~np.isnan(t[rows]) & ~np.isnan(want[rows])
            rel = np.abs(t[rows][both] - want[rows][both]) / np.maximum(1.0, np.abs(want[rows][both]))
            q = (float(np.quantile(rel, 0.99)), float(np.quantile(rel, 0.999)), float(rel.max()))
            if not (q[0] < 2e-5 and q[1] < 1e-4 and q[2] < 2e-3):
                problems.append(("distance", name, q))
        # modes 2 (exactly perpendicular), 3 (parallel to a cylinder's axis: qRoots' a == 0) and 5 (degenerate direction) are
        # decided like the reference, without exception
        # (mode 3: where the reference's a = 1 - (d.axis)^2 comes out as -4.4e-16 instead of 0 it reports a hit of an infinite
        # cylinder 4e8 units away; the fp32 rule |a| < 4e-7 calls those rays parallel, i.e. a miss, like the a == 0 ones)
        absurd = ~np.isnan(want) & (want > 1e6)
        for m in (2, 3, 5):
            if (flipped & (mode == m) & ~near_zero & ~absurd).any():
                problems.append(("mode", m, int((flipped & (mode == m)).sum())))
        assert not problems, (problems, report)
        print("fp32 product path, decisions that differ from the reference per type (all, outside the edge modes):", report)


@pytest.mark.parametrize("preset", [10, 4, 7, 9])
def test_scene_scans_f64_exact(rt, preset):
    capi, ctx = rt
    b = load_bundle(f"closest_preset{preset}")
    sc = ctx.upload(load_scene(preset))
    p = capi.default_params(precision=capi.PRECISION_F64)
    t, idx = sc.intersect(p, capi.SCAN_CLOSEST, b["ray_o"], b["ray_d"])
    assert np.array_equal(idx, b["idx_all"]) and same(t, b["t_all"])
    t, idx = sc.intersect(p, capi.SCAN_CLOSEST_NOLIGHT, b["ray_o"], b["ray_d"])
    assert np.array_equal(idx, b["idx_nolight"]) and same(t, b["t_nolight"])
    _, occ = sc.intersect(p, capi.SCAN_ANY, b["ray_o"], b["ray_d"], b["source_dist"])
    assert np.array_equal(occ, b["occluded"])


# fp32 against the reference's fp64 scans on the golden rays (4096 a preset, half of them starting exactly on a surface),
# counted, not averaged.  Measured on the MI355X: not one hit / miss / occluder decision differs on any of the four presets,
# in any of the three scan modes; the slack of 2 is for a grazing ray of another compiler.  Distances: the rays whose fp32
# distance is off by more than 1e-4 relative are on-surface origins that take the other root of their own sphere or
# cylinder (same shape, other t): 1 on preset 10, none elsewhere.
F32_SCAN_EXPECTED = {10: dict(flips=0, far=1), 4: dict(flips=0, far=0), 7: dict(flips=0, far=0), 9: dict(flips=0, far=0)}


@pytest.mark.parametrize("preset", [10, 4, 7, 9])
def test_scene_scans_f32(rt, preset):
    capi, ctx = rt
    b = load_bundle(f"closest_preset{preset}")
    sc = ctx.upload(load_scene(preset))
    p = capi.default_params(precision=capi.PRECISION_F32)
    exp = F32_SCAN_EXPECTED[preset]
    t, idx = sc.intersect(p, capi.SCAN_CLOSEST, b["ray_o"], b["ray_d"])
    _, idx_nl = sc.intersect(p, capi.SCAN_CLOSEST_NOLIGHT, b["ray_o"], b["ray_d"])
    _, occ = sc.intersect(p, capi.SCAN_ANY, b["ray_o"], b["ray_d"], b["source_dist"])
    flips = [int((idx != b["idx_all"]).sum()), int((idx_nl != b["idx_nolight"]).sum()), int((occ != b["occluded"]).sum())]
    assert max(flips) <= exp["flips"] + 2, flips
    hit = (idx == b["idx_all"]) & (idx >= 0)
    rel = np.abs(t[hit] - b["t_all"][hit]) / np.maximum(1.0, b["t_all"][hit])
    assert int((rel > 1e-4).sum()) <= exp["far"] + 2, int((rel > 1e-4).sum())
    assert np.quantile(rel, 0.99) < 5e-6 and np.median(rel) < 1e-6


def _slot_boundary_scene(n_prims, width=96, height=64):
    """6 planes, spheres and boxes only -- no disc, no annulus: the types behind the last occupied one are empty and, with 64
    primitives, start at sorted slot 64"""
    from raytracer_amd import synth, scene as S
    rng = np.random.default_rng(n_prims)
    b = synth.SceneBuilder()
    for o, n in (((0, 0, 0), (0, 0, 1)), ((0, 0, 9), (0, 0, -1)), ((-7, 0, 0), (1, 0, 0)), ((7, 0, 0), (-1, 0, 0)), ((0, -7, 0), (0, 1, 0)), ((0, 7, 0), (0, -1, 0))):
        b.plane(o, n, (128, 128, 128), S.MAT_PLASTER)
    mats = [S.MAT_PLASTER, S.MAT_COPPER, S.MAT_MIRROR, S.MAT_GLASS]
    for k in range(8):
        a = np.array([rng.uniform(-5, 4), rng.uniform(-5, 4), rng.uniform(0.2, 6)])
        l = rng.uniform(0.5, 1.5, 3)
        b.box(a, (l[0], 0, 0), (0, l[1], 0), (0, 0, l[2]), tuple(rng.integers(32, 256, 3).tolist()), mats[k % 4])
    for k in range(n_prims - 6 - 48):
        b.ball((rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(0.5, 7)), rng.uniform(0.2, 0.8), tuple(rng.integers(32, 256, 3).tolist()), mats[k % 4])
    flat = b.finish((-6, -6, 5), 45, -15, width, height)
    assert flat.n_shapes == n_prims and not (flat.shape_meta[:, 0] >= 6).any()
    return flat


@pytest.mark.parametrize("n_prims", [63, 64, 65])
@pytest.mark.parametrize("groups", [0, 1])
def test_scans_at_the_64_slot_boundary(rt, n_prims, groups):
    """Scenes of 63, 64 and 65 primitives whose last primitive types are empty: the fp32 scans keep the slots to test as a
    64-bit mask, and an empty type behind slot 63 must test nothing (its range is [64, 64)).  fp64 scans and frame against
    the oracle bit for bit; fp32 scans against fp64 (same hit on all but grazing / on-surface rays, no spurious hits:
    a wrongly tested slot would read another type's record) and the fp32 frame within the stated tolerance."""
    capi, ctx = rt
    flat = _slot_boundary_scene(n_prims)
    with ctx.option("scan_groups", groups, 1):
        sc = ctx.upload(flat)
    osc = ora.Scene(flat)
    rng = np.random.default_rng(7)
    n = 20000
    ro = np.column_stack([rng.uniform(-6.5, 6.5, n), rng.uniform(-6.5, 6.5, n), rng.uniform(0.1, 8.5, n)])
    rd = rng.normal(size=(n, 3)); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    ro[n // 2:] = np.repeat(ro[n // 2::64], 64, axis=0)[:n - n // 2] + 0.01 * rng.normal(size=(n - n // 2, 3))       # bundles: waves that skip groups
    rd[n // 2:] = np.repeat(rd[n // 2::64], 64, axis=0)[:n - n // 2] + 0.02 * rng.normal(size=(n - n // 2, 3))
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    sd = np.full(n, 6.0)
    p64, p32 = capi.default_params(precision=capi.PRECISION_F64), capi.default_params(precision=capi.PRECISION_F32)
    for mode in (capi.SCAN_CLOSEST, capi.SCAN_CLOSEST_NOLIGHT, capi.SCAN_ANY):
        t64, i64 = sc.intersect(p64, mode, ro, rd, sd if mode == capi.SCAN_ANY else None)
        to, io = ora.scan_batch(osc, mode, ro, rd, sd if mode == capi.SCAN_ANY else None)
        assert np.array_equal(i64, io), mode
        if mode != capi.SCAN_ANY:
            assert same(t64, to)
        t32, i32 = sc.intersect(p32, mode, ro, rd, sd if mode == capi.SCAN_ANY else None)
        differ = i32 != i64
        # random interior origins: only grazing rays (a hit within fp32 rounding of a rim or of another hit) may differ
        assert differ.mean() < 2e-3, (mode, int(differ.sum()))
        if mode != capi.SCAN_ANY:
            hit = ~differ & (i64 >= 0)
            assert np.abs(t32[hit] - t64[hit]).max() < 2e-3
    q64 = capi.default_params(precision=capi.PRECISION_F64, use_direct_light=0, use_photon_maps=0, seed=3)
    q32 = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=3)
    a = sc.trace_tile(q64)
    pre, post, rays, _ = ora.render(osc, ora.params(seed=3, use_direct_light=0, use_photon_maps=0), None, 0, 0, flat.width, flat.height)
    assert same(a["pre"], pre) and a["stats"]["rays"] == rays
    b = sc.trace_tile(q32)
    err = np.abs(b["pre"][..., :3] - pre[..., :3]).max(axis=-1)
    assert (err <= 1e-3 * 255).mean() >= 0.99, (err <= 1e-3 * 255).mean()


def test_cylinder_rays_near_the_axis_f32(rt):
    """Rays within a fraction of a degree of a cylinder's axis, from inside a finite tube and an infinite one.  The reference
    misses only at a == 0 exactly (qRoots, rayphoton.js:1884-1889).  The fp32 test solves the quadratic through the part of
    the direction across the axis (no cancellation), whose components carry about 1e-7 of rounding, and calls a ray within
    1e-5 rad (0.0006 degrees) of the axis parallel.  Stated tolerance: from 0.01 degrees on every hit / miss decision is the
    fp64 one and the distance is within 1e-3 relative (1e-4 from 0.06 degrees on; measured 3.5e-4 / 2e-5); below 0.0005 degrees the fp32 build
    reports a miss where fp64 hits the wall more than 50 000 radii down the tube -- for a finite cylinder shorter than that
    it is a miss in fp64 too."""
    capi, ctx = rt
    rng = np.random.default_rng(21)
    n = 4096
    axis = np.array([0.3, -0.2, 0.93]); axis /= np.linalg.norm(axis)
    ang = np.radians(np.concatenate([rng.uniform(0.0, 0.0005, n // 4), rng.uniform(0.01, 0.06, n // 4), rng.uniform(0.06, 5.0, n // 2)]))
    perp = np.cross(axis, rng.normal(size=(n, 3))); perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    d = np.cos(ang)[:, None] * axis + np.sin(ang)[:, None] * perp
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    centre = np.array([1.0, 2.0, 3.0])
    o = centre + 0.3 * rng.uniform(-1, 1, (n, 1)) * perp + rng.uniform(-1, 1, (n, 1)) * axis      # inside the tube (radius 0.5)
    for finite in (True, False):
        meta = np.zeros((n, 4), np.uint8)
        meta[:, 0] = 3
        meta[:, 2] = 1 | (8 if finite else 0)
        g = np.zeros((n, 12))
        g[:, 0:3], g[:, 3:6], g[:, 6], g[:, 7] = centre, axis, 0.5, 2000.0
        t64 = ctx.intersect_pairs(capi.default_params(precision=capi.PRECISION_F64), o, d, meta, g)
        t32 = ctx.intersect_pairs(capi.default_params(precision=capi.PRECISION_F32), o, d, meta, g)
        assert same(t64, ora.intersect_batch(o, d, meta, g))
        mid, wide = (ang >= np.radians(0.01)) & (ang < np.radians(0.06)), ang >= np.radians(0.06)
        for sel, tol in ((mid, 1e-3), (wide, 1e-4)):
            flips = np.isnan(t32[sel]) != np.isnan(t64[sel])
            # (a hit within rounding of the tube's end may fall on the other side of `|z| < halfHeight - eps`)
            assert flips.sum() <= 2, (finite, tol, int(flips.sum()))
            both = sel & ~np.isnan(t64) & ~np.isnan(t32)
            assert both.sum() > n // 8
            rel = np.abs(t32[both] - t64[both]) / t64[both]
            print(f"finite={finite} angles {'0.01-0.06' if tol > 1e-4 else '>= 0.06'} deg: relative distance error max {rel.max():.2e}, median {np.median(rel):.2e}")
            assert rel.max() < tol, (finite, tol)
        narrow = ang < np.radians(0.0005)
        assert np.isnan(t32[narrow]).all()                                  # the fp32 rule: parallel, a miss
        if finite:
            assert np.isnan(t64[narrow]).all()                              # ... which is fp64's answer too: the wall lies beyond the tube's ends
        else:
            assert (t64[narrow][~np.isnan(t64[narrow])] > 0.2 / np.sin(np.radians(0.0005))).all()


# ------------------------------------------------------------------ photon emission
@pytest.fixture(scope="module")
def c1_emit(rt):
    capi, ctx = rt
    sc = ctx.upload(load_scene(10))
    p = capi.default_params(precision=capi.PRECISION_F64, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    return sc, p, sc.emit_many(p)


def test_emission_c1_f64_matches_reference(c1_emit):
    _, _, e = c1_emit
    g = load_bundle("photons_c1")
    st = e["stats"]
    assert st["attempts_phase1"] == g.meta["stats"]["attempts1"] and st["attempts_phase2"] == g.meta["stats"]["attempts2"]
    assert st["capped"] == 0 and st["errors"] == 0
    for name in ("diffuse", "caustic"):
        tag = g[name + "_tag"]
        assert np.array_equal(e[name + "_attempt"], tag[:, 1:3].astype(np.int64))     # (phase, attempt) of every stored photon
        assert e[name].shape == g[name].shape
        err = np.abs(e[name] - g[name])
        assert np.nanmax(err) < 1e-12, np.nanmax(err)
        assert same(e[name][:, 7:], g[name][:, 7:])
        assert (e[name] == g[name])[:, :7].mean() > 0.99


def test_emission_sharded_ranges_concatenate(rt):
    """attempt-range sharding (what the multi-GPU path does) gives the single-range result"""
    capi, ctx = rt
    sc = ctx.upload(load_scene(10))
    p = capi.default_params(precision=capi.PRECISION_F64, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    whole = sc.emit_photons(p, 1, 0, 6000)
    parts = [sc.emit_photons(p, 1, a, b) for a, b in ((0, 1000), (1000, 1001), (1001, 4096), (4096, 6000))]
    assert same(np.concatenate([x[0] for x in parts]), whole[0])
    assert np.array_equal(np.concatenate([x[1] for x in parts]), whole[1])
    assert np.array_equal(np.concatenate([x[2] for x in parts]), whole[2])
    # and equals the oracle attempt by attempt
    osc = ora.Scene(load_scene(10))
    op = ora.params(num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    flag, rec = ora.emit_attempts(osc, op, 1, 0, 6000)
    assert np.array_equal(np.nonzero(flag)[0], whole[1])
    assert np.array_equal(flag[flag > 0].astype(np.uint8), whole[2])
    assert np.nanmax(np.abs(rec[flag > 0] - whole[0])) < 1e-12
    assert sc.emit_photons(p, 2, 5, 5)[0].shape[0] == 0       # empty range


def test_emission_f32_statistics(rt):
    """fp32 emission against the reference's lists (10 000 + 10 000 photons), counted.  The streams are keyed by attempt, so
    an attempt is comparable one to one.  Measured on the MI355X: the phases end 4 and 6 attempts away from the reference's
    34 109 / 42 136; 9 998 of the 10 000 diffuse and 9 999 of the 10 000 caustic photons are stored by the same attempts
    (the others: a hit within fp32 rounding of a rim, or a draw within rounding of its threshold, decided the other way --
    and one such attempt shifts every later list position, hence sets, not positions); of the common attempts 191 / 204
    store their photon more than 1e-3 away (a walk through glass amplifies the fp32 rounding of its first hit with every
    bounce), the rest within 2.5e-4 at the 95th percentile."""
    capi, ctx = rt
    sc = ctx.upload(load_scene(10))
    p = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    e = sc.emit_many(p)
    g = load_bundle("photons_c1")
    st = e["stats"]
    assert abs(st["attempts_phase1"] - g.meta["stats"]["attempts1"]) <= 4 + 16
    assert abs(st["attempts_phase2"] - g.meta["stats"]["attempts2"]) <= 6 + 16
    for name, lost, displaced in (("diffuse", 2, 191), ("caustic", 1, 204)):
        a, gt = e[name + "_attempt"], g[name + "_tag"]
        ka = a[:, 0].astype(np.int64) * (1 << 40) + a[:, 1].astype(np.int64)           # (phase, attempt)
        kg = gt[:, 1].astype(np.int64) * (1 << 40) + gt[:, 2].astype(np.int64)
        common, ia, ib = np.intersect1d(ka, kg, return_indices=True)
        assert gt.shape[0] - common.size <= lost + 8, (name, common.size)
        d = np.abs(e[name][ia][:, :3] - g[name][ib][:, :3]).max(axis=1)
        assert int((d > 1e-3).sum()) <= displaced + 60, (name, int((d > 1e-3).sum()))
        assert np.quantile(d, 0.95) < 5e-4


def test_emission_attempt_cap(rt):
    capi, ctx = rt
    flat = load_scene(10)
    # no caustic path exists without specular/transparent shapes: make everything plaster and ask for caustics
    flat.shape_meta = flat.shape_meta.copy()
    flat.shape_meta[flat.shape_meta[:, 1] < 8, 1] = 6
    sc = ctx.upload(flat)
    p = capi.default_params(precision=capi.PRECISION_F64, num_photons_diffuse=50, num_photons_caustic=20, seed=SEED)
    e = sc.emit_many(p, strict=False)
    assert e["stats"]["capped"] == 1 and e["stats"]["rc"] == -5
    assert e["stats"]["attempts_phase2"] == 100 * 20 + 1 and e["stats"]["n_caustic"] == 0 and e["stats"]["n_diffuse"] == 50
    osc = ora.Scene(flat)
    oe = ora.emit_many(osc, ora.params(num_photons_diffuse=50, num_photons_caustic=20, seed=SEED))
    assert oe["capped"] and oe["attempts2"] == e["stats"]["attempts_phase2"] and oe["attempts1"] == e["stats"]["attempts_phase1"]


# ------------------------------------------------------------------ kd-tree
def test_find_median(rt):
    capi, ctx = rt
    fm = load_bundle("kd_c1")["find_median"]
    assert [ctx.find_median(n) for n in range(1, 4097)] == fm.tolist()


def test_kd_build_matches_reference_tree(rt):
    capi, ctx = rt
    ph, kd = load_bundle("photons_c1"), load_bundle("kd_c1")
    for name in ("diffuse", "caustic"):
        perm, dim = ctx.kd_build(ph[name][:, :3].astype(np.float32))
        assert np.array_equal(perm, kd[name + "_perm"]) and np.array_equal(dim, kd[name + "_dim"])
        m = ctx.photon_map(ph[name])
        perm, dim = m.nodes()
        assert np.array_equal(perm, kd[name + "_perm"]) and np.array_equal(dim, kd[name + "_dim"])


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 100, 1000, 4097, 50000])
def test_kd_build_ties_and_ragged_sizes(rt, n):
    """heavy coordinate ties (photons on box faces, duplicated points) exercise the stable-sort rule"""
    capi, ctx = rt
    rng = np.random.default_rng(n)
    pos = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    face = rng.integers(0, 6, n)
    for i in range(3):
        pos[face == 2 * i, i] = -4.0
        pos[face == 2 * i + 1, i] = 4.0
    if n > 10:
        pos[n // 2:n // 2 + n // 10] = pos[0]                  # exact duplicates
        pos[3, 2] = -0.0
        pos[4, 2] = 0.0
    perm, dim = ctx.kd_build(pos)
    operm, odim = ora.kd_build(pos.astype(np.float64))
    assert np.array_equal(perm, operm) and np.array_equal(dim, odim)


def test_kd_build_empty_is_an_error(rt):
    capi, ctx = rt
    with pytest.raises(capi.RtError) as e:                     # kdtree.js:6
        ctx.kd_build(np.zeros((0, 3), np.float32))
    assert e.value.code == -4 and "no data" in str(e.value)
    with pytest.raises(capi.RtError):
        ctx.photon_map(np.zeros((0, 11)))


def test_radius_and_knn_queries_match_reference(rt):
    capi, ctx = rt
    ph, q = load_bundle("photons_c1"), load_bundle("queries_c1")
    maps = {name: ctx.photon_map(ph[name]) for name in ("diffuse", "caustic")}
    queries = q["queries"].astype(np.float32)
    assert np.array_equal(queries.astype(np.float64), q["queries"])
    for key, r in q.meta["radii"].items():
        m = maps[key.split("_")[0]]
        offs, want = q[key + "_offsets"], q[key + "_indices"]
        cap = int(np.diff(offs).max()) + 1
        count, idx = m.radius(queries, r, cap=cap)
        assert np.array_equal(count, np.diff(offs).astype(np.uint32))
        for i in range(queries.shape[0]):
            assert np.array_equal(idx[i, :count[i]], want[offs[i]:offs[i + 1]]), (key, i)   # same order as kdtree.js:64-80
        # a cap below the hit count truncates the list but still reports the full count
        count2, idx2 = m.radius(queries, r, cap=2)
        assert np.array_equal(count2, count) and np.array_equal(idx2[:, :2][count >= 2], idx[:, :2][count >= 2])
    K = q.meta["K"]
    for name in ("diffuse", "caustic"):
        idx, d2 = maps[name].knn(queries, K)
        assert np.array_equal(idx, q[f"knn_{name}_idx"]) and np.array_equal(d2, q[f"knn_{name}_d2"])


def test_nearest_neighbour_and_distance_is_the_reference_s(rt):
    """kdtree.js:10-62 through rt_kd_nearest_path: the reference's own (approximate, one-path) answers on trees of
    10 000 / 9 999 / 2 / 1 photons, bit for bit, [undefined, undefined] as -1 / NaN; lattice maps (distance ties, one-child
    nodes at every even size) against the oracle"""
    capi, ctx = rt
    ph, g = load_bundle("photons_c1"), load_bundle("nearest_c1")
    for name, (which, n) in g.meta["trees"].items():
        m = ctx.photon_map(ph[which][:n])
        n_undef = 0
        for pre in (name + "_", name + "_own_"):
            qs = g[pre + "queries"].astype(np.float32)
            assert np.array_equal(qs.astype(np.float64), g[pre + "queries"])
            idx, dist = m.nearest_path(qs)
            assert np.array_equal(idx, g[pre + "idx"]), pre
            assert np.array_equal(dist, g[pre + "dist"], equal_nan=True), pre
            n_undef += int((idx < 0).sum())
        assert n_undef == g.meta["undefined_results"][name]
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 6, 300, 301):
        rec = np.zeros((n, 11))
        rec[:, :3] = rng.integers(-3, 4, (n, 3))
        rec[:, 6:10] = 1
        m = ctx.photon_map(rec)
        perm, dim = ora.kd_build(rec[:, :3])
        queries = np.concatenate([rng.integers(-4, 5, (200, 3)), rec[:, :3]]).astype(np.float32)
        idx, dist = m.nearest_path(queries)
        for i in range(queries.shape[0]):
            oi, od = ora.kd_nearest(rec[:, :3], perm, dim, queries[i].astype(np.float64))
            assert idx[i] == oi and (dist[i] == od or (oi < 0 and np.isnan(dist[i]))), (n, i)
    # the one-path answer is never closer than the exact nearest photon, and is it for most queries of a real map
    m = ctx.photon_map(ph["diffuse"])
    qs = g["diffuse_queries"].astype(np.float32)
    idx, dist = m.nearest_path(qs)
    _, d2 = m.knn(qs, 1)
    ok = idx >= 0
    assert (dist[ok] >= np.sqrt(d2[ok, 0])).all() and (dist[ok] == np.sqrt(d2[ok, 0])).mean() > 0.3


def test_knn_ties_small_maps_and_k_edge_cases(rt):
    capi, ctx = rt
    rng = np.random.default_rng(5)
    n = 300
    rec = np.zeros((n, 11))
    rec[:, :3] = rng.integers(-3, 4, (n, 3))                   # lattice points: massive distance ties
    rec[:, 3:6] = [0, 0, 1]
    rec[:, 6:10] = 1
    m = ctx.photon_map(rec)
    queries = rng.integers(-3, 4, (64, 3)).astype(np.float32)
    pos = rec[:, :3]
    for k in (1, 5, 64, 128):
        idx, d2 = m.knn(queries, k)
        for i in range(queries.shape[0]):
            oi, od = ora.knn_brute(pos, queries[i].astype(np.float64), k)
            assert np.array_equal(idx[i], oi) and np.array_equal(d2[i], od), (k, i)
    small = ctx.photon_map(rec[:10])
    idx, d2 = small.knn(queries[:4], 16)                       # k > n: tail is -1 / inf
    assert (idx[:, 10:] == -1).all() and np.isinf(d2[:, 10:]).all() and (idx[:, :10] >= 0).all()
    with pytest.raises(capi.RtError):
        m.knn(queries, 0)


def _oracle_gather(pos, rec, hits, idx_lists, r2):
    out = np.zeros((hits.shape[0], 3))
    for i, idx in enumerate(idx_lists):
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size == 0:
            continue
        w = np.maximum(0.0, rec[idx, 3:6] @ hits[i, 3:6].astype(np.float64))
        s = (rec[idx, 7:10] * rec[idx, 6:7] * w[:, None]).sum(axis=0)
        out[i] = s * hits[i, 6:9] / 255 / (np.pi * r2[i])
    return out


def test_gather_radiance_radius_and_knn(rt):
    capi, ctx = rt
    ph, q = load_bundle("photons_c1"), load_bundle("queries_c1")
    rec = ph["diffuse"]
    m = ctx.photon_map(rec)
    nq = q["queries"].shape[0]
    rng = np.random.default_rng(1)
    nrm = rng.normal(size=(nq, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    hits = np.concatenate([q["queries"], nrm, rng.uniform(0, 255, (nq, 3))], axis=1).astype(np.float32)
    offs, want = q["diffuse_wide_offsets"], q["diffuse_wide_indices"]
    r = q.meta["radii"]["diffuse_wide"]
    got = m.gather(hits, r=r)
    exp = _oracle_gather(rec[:, :3], rec, hits, [want[offs[i]:offs[i + 1]] for i in range(nq)], np.full(nq, r * r))
    assert np.allclose(got, exp, rtol=2e-5, atol=1e-6 * max(1.0, np.abs(exp).max()))
    K = q.meta["K"]
    got = m.gather(hits, k=K)
    exp = _oracle_gather(rec[:, :3], rec, hits, q["knn_diffuse_idx"], q["knn_diffuse_d2"][:, -1])
    assert np.allclose(got, exp, rtol=2e-5, atol=1e-6 * max(1.0, np.abs(exp).max()))


# ------------------------------------------------------------------ renderer
def test_image_c1_f64_matches_reference(rt):
    capi, ctx = rt
    g = load_bundle("image_c1_modeB")
    W, H = g.meta["W"], g.meta["H"]
    ph = load_bundle("photons_c1")
    sc = ctx.upload(load_scene(10, W, H))
    md, mc = ctx.photon_map(ph["diffuse"]), ctx.photon_map(ph["caustic"])
    p = capi.default_params(precision=capi.PRECISION_F64, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    out = sc.trace_tile(p, md, mc)
    pre, post = out["pre"], out["post"]
    want_pre, want_post = g["pre_floor"], g["post_floor"]
    err = np.abs(pre[..., :3] - want_pre[..., :3]) / np.maximum(1.0, np.abs(want_pre[..., :3]))
    bad = (err > 1e-9).any(axis=-1)
    # photon positions are fp32 in the device tree: a photon within rounding of a search-ball boundary may flip
    assert bad.sum() <= 8, bad.sum()
    assert same(pre[..., 3][~bad], want_pre[..., 3][~bad])
    post_bad = ~((post == want_post) | (np.isnan(post) & np.isnan(want_post))).all(axis=-1)
    assert post_bad.sum() <= 8, post_bad.sum()
    st = out["stats"]
    assert st["rays"] == 110353 and st["stack_overflows"] == 0       # traceRay calls that scan, counted by the oracle


# ------------------------------------------------------------------ the scene's single-item methods, batched
def _camera_rays(flat, seed, W, H, Y):
    """the camera rays of row Y as traceOnCanvas builds them (rayphoton.js:1589-1609), fp64, draws 0 and 1 of each pixel's stream"""
    cam = np.asarray(flat.camera, np.float64)
    co, cu, cv, cw = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    o = np.tile(co, (W, 1))
    d = np.empty((W, 3))
    for X in range(W):
        a = Y * W + X
        u0, u1 = ora.uniform(seed, 0, a, 0, 0), ora.uniform(seed, 0, a, 0, 1)
        u = ((X + (0 + u0) / 1) * 2 / W) - 1
        v = -(((Y + (0 + u1) / 1) * 2 / H) - 1)
        u *= cam[12]
        v *= cam[13]
        xyz = ((co + u * cu) + v * cv) + (-1.0) * cw
        dv = xyz - co
        s2 = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]
        d[X] = (1.0 / np.sqrt(s2)) * dv
    return o, d


def test_trace_rays_continue_the_pixels_of_a_frame(rt):
    """rt_trace_rays on the camera rays of a frame's rows, keyed with the pixels' streams from draw 2, returns the
    frame's own colours bit for bit (fp64; direct light and both photon maps on)"""
    capi, ctx = rt
    g = load_bundle("image_c1_modeB")
    W, H = g.meta["W"], g.meta["H"]
    ph = load_bundle("photons_c1")
    flat = load_scene(10, W, H)
    sc = ctx.upload(flat)
    md, mc = ctx.photon_map(ph["diffuse"]), ctx.photon_map(ph["caustic"])
    p = capi.default_params(precision=capi.PRECISION_F64, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    frame = sc.trace_tile(p, md, mc)
    rays = 0
    for Y in (0, 97, 151, H - 1):
        o, d = _camera_rays(flat, SEED, W, H, Y)
        got, st = sc.trace_rays(p, o, d, md, mc, first_stream=Y * W, first_draw=np.full(W, 2, np.uint32), stats=True)
        assert same(got, frame["pre"][Y]), Y
        rays += st["rays"]
    assert rays > 4 * W
    # the arguments traceRay takes: a ray of importance below 0.01 is black with alpha 0 (:1627); a short maxDist at the
    # depth limit ends the path at its first hit (:1677); an explicit [MAT_AIR] stack is the default
    o, d = _camera_rays(flat, SEED, W, H, 97)
    got = sc.trace_rays(p, o[:8], d[:8], md, mc, importance=np.full(8, 0.005), first_stream=97 * W, first_draw=np.full(8, 2, np.uint32))
    assert np.array_equal(got, np.zeros((8, 4)))
    base = sc.trace_rays(p, o, d, md, mc, first_stream=97 * W, first_draw=np.full(W, 2, np.uint32))
    again = sc.trace_rays(p, o, d, md, mc, stacks=[[0]] * W, max_dist=np.full(W, p.max_trace_dist), depth=np.zeros(W, np.int32),
                          importance=np.ones(W), first_stream=97 * W, first_draw=np.full(W, 2, np.uint32))
    assert same(again, base)
    cut, st_cut = sc.trace_rays(p, o, d, md, mc, max_dist=np.zeros(W), depth=np.full(W, p.max_depth, np.int32), first_stream=97 * W,
                                first_draw=np.full(W, 2, np.uint32), stats=True)
    assert st_cut["rays"] == W and not same(cut, base)
    with pytest.raises(capi.RtError):
        sc.trace_rays(p, o[:2], d[:2], stacks=[[0], [9]])


def test_direct_light_points_match_the_oracle(rt):
    capi, ctx = rt
    flat = load_scene(10, 320, 240)
    sc, osc = ctx.upload(flat), ora.Scene(flat)
    p = capi.default_params(precision=capi.PRECISION_F64, seed=SEED)
    op = ora.params(seed=SEED)
    rng = np.random.default_rng(8)
    n = 200
    pts = np.column_stack([rng.uniform(-6, 9, n), rng.uniform(-8, 9, n), rng.uniform(0.01, 9, n)])
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[: n // 2] = [0, 0, 1]
    local = rng.uniform(0, 255, (n, 3))
    first_draw = rng.integers(0, 9, n).astype(np.uint32)
    got, draws = sc.direct_light(p, pts, nrm, local, first_stream=5000, first_draw=first_draw)
    lit = 0
    for i in range(n):
        exp, dr = ora.direct_light(osc, op, 0, 5000 + i, 0, int(first_draw[i]), pts[i], nrm[i], local[i])
        assert same(got[i], exp), i
        assert draws[i] == dr == first_draw[i] + 5 * flat.n_lights
        lit += bool(np.any(exp != 0))
    assert lit > n // 10


def test_create_and_walk_photons_are_the_attempts_of_the_bulk_emission(rt):
    """createPhoton then emitPhoton, photon by photon, store what rt_emit_photons stores for the same attempts"""
    capi, ctx = rt
    flat = load_scene(10, 320, 240)
    sc, osc = ctx.upload(flat), ora.Scene(flat)
    p = capi.default_params(precision=capi.PRECISION_F64, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    n = 4000
    for phase in (1, 2):
        made, draws = sc.create_photons(p, n, phase=phase, first_stream=100)
        assert np.all(draws == 5)
        for i in (0, 1, 999):
            l, exp = ora.create_photon(osc, SEED, phase, 100 + i, 0)
            assert same(made[i, :7], exp[:7])
        rec, kind = sc.walk_photons(p, made, phase=phase, first_stream=100, first_draw=draws, track_only_caustic=(phase == 2))
        bulk_rec, bulk_att, bulk_kind, _ = sc.emit_photons(p, phase, 100, 100 + n)
        stored = np.nonzero(kind)[0]
        assert np.array_equal(stored + 100, bulk_att) and np.array_equal(kind[stored], bulk_kind)
        assert same(rec[stored], bulk_rec)
        assert len(stored) > n // 10
    # a named source: every photon starts on that light's disc; the draw that would have chosen the light is still made
    lamp = flat.shape_geom[flat.light_shape[1]]
    made, draws = sc.create_photons(p, 500, first_stream=0, source=np.ones(500, np.int32))
    assert np.all(draws == 5)
    assert np.all(np.linalg.norm(made[:, :3] - lamp[:3], axis=1) <= lamp[6] + 1e-12) and np.all(made[:, 6] == flat.light_data[1][0])
    # a photon handed over in mid-flight: depth 1 inside glass, already caustic
    rec2, kind2 = sc.walk_photons(p, made[:64], first_stream=9000, depth=np.ones(64, np.int32), caustic=np.ones(64, np.int8), stacks=[[0, 1]] * 64)
    assert set(np.unique(kind2)) <= {0, 1, 2}


def test_image_c1_f32_within_tolerance(rt):
    capi, ctx = rt
    g = load_bundle("image_c1_modeB")
    W, H = g.meta["W"], g.meta["H"]
    ph = load_bundle("photons_c1")
    sc = ctx.upload(load_scene(10, W, H))
    md, mc = ctx.photon_map(ph["diffuse"]), ctx.photon_map(ph["caustic"])
    p = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED,
                            shadow_bias=1e-4)
    out = sc.trace_tile(p, md, mc)
    err = np.abs(out["pre"][..., :3] - g["pre_floor"][..., :3]).max(axis=-1)
    ok = err <= 1e-3 * 255
    # stated tolerance: per-pixel radiance within 1e-3*255 on >= 99.5 % of pixels; the rest took another discrete branch
    assert ok.mean() >= 0.995, (ok.mean(), (~ok).sum())


def test_render_is_tile_and_subsample_invariant(rt):
    """size-independent properties at a larger canvas: tiles compose, sub-samples average"""
    capi, ctx = rt
    W, H = 640, 360
    sc = ctx.upload(load_scene(10, W, H))
    p = capi.default_params(precision=capi.PRECISION_F32, use_photon_maps=0, seed=7)
    whole = sc.trace_tile(p)
    parts = [sc.trace_tile(p, x0=x0, y0=y0, w=w, h=h) for (x0, y0, w, h) in
             ((0, 0, 400, 200), (400, 0, 240, 200), (0, 200, 640, 160))]
    assert same(whole["pre"][:200, :400], parts[0]["pre"]) and same(whole["pre"][:200, 400:], parts[1]["pre"])
    assert same(whole["pre"][200:], parts[2]["pre"])
    assert whole["stats"]["rays"] == sum(x["stats"]["rays"] for x in parts)
    p2 = capi.default_params(precision=capi.PRECISION_F64, use_photon_maps=0, seed=7, sub_sample=2)
    osc = ora.Scene(load_scene(10, W, H))
    op = ora.params(seed=7, sub_sample=2, use_photon_maps=0)
    got = sc.trace_tile(p2, x0=100, y0=50, w=64, h=32)
    _, post, rays, _ = ora.render(osc, op, None, 100, 50, 64, 32)
    assert (~same_px(got["post"], post)).sum() <= 2 and got["stats"]["rays"] == rays
    with pytest.raises(capi.RtError):
        sc.trace_tile(p, x0=600, y0=0, w=100, h=10)


def same_px(a, b):
    return ((a == b) | (np.isnan(a) & np.isnan(b))).all(axis=-1)


# ------------------------------------------------------------------ mode A: the older tracer (ray.js)
@pytest.mark.parametrize("preset,rays", [(2, 52993), (3, 59233), (0, 58604), (1, 90165)])
def test_image_mode_a_f64_matches_reference(rt, preset, rays):
    capi, ctx = rt
    g = load_bundle(f"image_rayjs{preset}")
    sc = ctx.upload(load_rayjs_scene(preset))
    p = capi.default_params(precision=capi.PRECISION_F64, mode=capi.MODE_WHITTED, seed=g.meta["seed"], **RAYJS_PARAMS)
    out = sc.trace_tile(p)
    assert same(out["pre"], g["rgba"]) and same(out["post"], g["rgba"])
    assert out["stats"]["rays"] == rays and out["stats"]["stack_overflows"] == 0
    # the same frame with every bounce as a wavefront pass (no tail kernel), and as two tiles
    with ctx.option("tail_threshold", 0, -1):
        wave = sc.trace_tile(p)
    assert same(wave["pre"], g["rgba"]) and wave["stats"]["tail_rays"] == 0 and wave["stats"]["rays"] == rays
    with ctx.option("tail_threshold", 4000000, -1):          # fused first pass, compaction, then the tail kernel
        mixed = sc.trace_tile(p)
    assert same(mixed["pre"], g["rgba"]) and mixed["stats"]["tail_rays"] == rays - g.meta["W"] * g.meta["H"]
    with ctx.option("whole_frame", 1, 0):                    # one launch for the whole frame
        mega = sc.trace_tile(p)
    assert same(mega["pre"], g["rgba"]) and mega["stats"]["rays"] == rays
    H = g.meta["H"]
    top, bot = sc.trace_tile(p, h=50), sc.trace_tile(p, y0=50, h=H - 50)
    assert same(np.concatenate([top["pre"], bot["pre"]]), g["rgba"])


def test_image_mode_a_f32_and_subsamples(rt):
    capi, ctx = rt
    g = load_bundle("image_rayjs2")
    flat = load_rayjs_scene(2)
    sc = ctx.upload(flat)
    p = capi.default_params(precision=capi.PRECISION_F32, mode=capi.MODE_WHITTED, seed=g.meta["seed"], **RAYJS_PARAMS)
    out = sc.trace_tile(p, want=("pre", "rgba8"))
    # stated tolerance: within 1e-3*255 (+ the unit step of Math.round) on >= 99 % of pixels; diffuse bounces
    # that take another discrete branch in fp32 make up the rest
    err = np.abs(out["pre"][..., :3] - g["rgba"][..., :3]).max(axis=-1)
    assert (err <= 1.0 + 1e-3 * 255).mean() >= 0.99, (err <= 1.26).mean()
    assert np.array_equal(out["rgba8"][..., :3], np.nan_to_num(out["pre"][..., :3]).astype(np.uint8))
    # 2x2 sub-samples against the oracle
    p2 = capi.default_params(precision=capi.PRECISION_F64, mode=capi.MODE_WHITTED, seed=5, **{**RAYJS_PARAMS, "sub_sample": 2})
    got = sc.trace_tile(p2, x0=40, y0=30, w=64, h=48)
    op = ora.params(seed=5, **{**RAYJS_PARAMS, "sub_sample": 2})
    _, post, rays = ora.render_whitted(ora.Scene(flat), op, 40, 30, 64, 48)
    assert same(got["post"], post) and got["stats"]["rays"] == rays
    # a scene without the per-shape surface table cannot be traced in this mode
    with pytest.raises(capi.RtError):
        ctx.upload(load_scene(10, 64, 48)).trace_tile(p)


@pytest.mark.parametrize("precision", ["F64", "F32"])
def test_wavefront_and_whole_frame_pipelines_agree(rt, precision):
    """the single-launch pipeline and the multi-pass wavefront pipeline are two schedules of the same paths"""
    capi, ctx = rt
    ph = load_bundle("photons_c1")
    sc = ctx.upload(load_scene(10, 320, 240))
    md, mc = ctx.photon_map(ph["diffuse"]), ctx.photon_map(ph["caustic"])
    for kw in (dict(), dict(use_direct_light=0, use_photon_maps=0)):
        p = capi.default_params(precision=getattr(capi, "PRECISION_" + precision), num_photons_diffuse=10000,
                                num_photons_caustic=10000, seed=SEED, shadow_bias=1e-4, **kw)
        with ctx.option("whole_frame", 1, 0):
            whole = sc.trace_tile(p, md, mc)
        for thr in (0, 1000, 4000000):
            with ctx.option("tail_threshold", thr, -1):
                wave = sc.trace_tile(p, md, mc)
            assert same(wave["pre"], whole["pre"]) and same(wave["post"], whole["post"])
            for k in ("rays", "shadow_rays", "gathers"):
                assert wave["stats"][k] == whole["stats"][k], (k, thr)
        # tail workgroups that hand their last paths on to further launches
        for ho, rounds in ((8, 1), (32, 2), (64, 3)):
            with ctx.option("tail_handoff", ho, 0), ctx.option("tail_rounds", rounds, 1):
                got = sc.trace_tile(p, md, mc)
            assert same(got["pre"], whole["pre"]) and same(got["post"], whole["post"]), (ho, rounds)
            for k in ("rays", "shadow_rays", "gathers", "bounces"):
                assert got["stats"][k] == whole["stats"][k], (k, ho, rounds)
        # the tail kernel with persistent lanes
        for per_lane, ho in ((1, 16), (4, 16), (16, 16), (4, 0), (4, 1), (4, 32), (1, 32)):      # ho: waves down to that many paths hand them to a second launch
            with ctx.option("tail_persistent", per_lane, 0), ctx.option("tail_persistent_handoff", ho, 0):
                got = sc.trace_tile(p, md, mc)
            assert same(got["pre"], whole["pre"]) and same(got["post"], whole["post"]), (per_lane, ho)
            for k in ("rays", "shadow_rays", "gathers", "bounces"):
                assert got["stats"][k] == whole["stats"][k], (k, per_lane, ho)
        # single-bounce passes between the first pass and the tail kernel
        for n in (0, 1, 2):
            with ctx.option("bounce_passes", n, 0):
                for mb in (0, 1, 2, 3):
                    p.max_bounces = mb
                    with ctx.option("whole_frame", 1, 0):
                        ref = whole if mb == 0 else sc.trace_tile(p, md, mc)
                    got = sc.trace_tile(p, md, mc)
                    assert same(got["pre"], ref["pre"]) and same(got["post"], ref["post"]), (n, mb)
                    for k in ("rays", "shadow_rays", "gathers", "bounces"):
                        assert got["stats"][k] == ref["stats"][k], (k, n, mb)
                p.max_bounces = 0


def test_queued_photon_gathers_equal_the_gathers_in_place(rt):
    """fp32 frames with photon maps queue their radius gathers (rayphoton.js:1765-1815) and serve them afterwards, sorted by
    position, 64 at a time, subtrees of the maps as work items of their own: the same photons are summed (the fp64
    `d2 < r^2` decides), in another order -- so the frame equals the one with every gather done where it is shaded to
    fp32 rounding, run after run and schedule after schedule bit for bit, and the counters do not change"""
    capi, ctx = rt
    ctx.set_option("defer_gathers", 1)                      # (by itself the library queues only for maps of 200 000 photons and more)
    try:
        ph = load_bundle("photons_c1")
        sc = ctx.upload(load_scene(10, 320, 240))
        md, mc = ctx.photon_map(ph["diffuse"]), ctx.photon_map(ph["caustic"])
        for kw in (dict(), dict(sub_sample=2), dict(use_direct_light=0)):
            p = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED, shadow_bias=1e-4, **kw)
            with ctx.option("defer_gathers", 0, 1):
                ref = sc.trace_tile(p, md, mc)
            got = sc.trace_tile(p, md, mc)
            again = sc.trace_tile(p, md, mc)
            assert same(got["pre"], again["pre"]) and same(got["post"], again["post"])
            for k in ("rays", "shadow_rays", "gathers", "bounces"):
                assert got["stats"][k] == ref["stats"][k], k
            assert ref["stats"]["gathers"] > 100000
            tol = 2e-5 * np.maximum(1.0, np.abs(ref["pre"][..., :3]))
            assert (np.abs(got["pre"][..., :3] - ref["pre"][..., :3]) <= tol).all()
            assert (got["post"] != ref["post"]).any(axis=-1).mean() < 2e-3         # (a floor that falls the other way)
            assert same(got["pre"][..., 3], ref["pre"][..., 3])
            variants = {}
            for name, opts in (("unsorted, waves together", dict(gather_sort=0)), ("unsorted, lane by lane", dict(gather_sort=0, wave_gather=0)),
                               ("sorted, no work items", dict(gather_split=0)), ("sorted, small work items", dict(gather_split=64)),
                               ("whole frame in one launch", dict(whole_frame=1)), ("persistent tail", dict(tail_persistent=4))):
                import contextlib
                with contextlib.ExitStack() as es:
                    for o, v in opts.items():
                        es.enter_context(ctx.option(o, v, {"gather_sort": 1, "wave_gather": 8, "gather_split": 4096, "whole_frame": 0, "tail_persistent": -1}[o]))
                    variants[name] = sc.trace_tile(p, md, mc)
            for name, v in variants.items():
                assert (np.abs(v["pre"][..., :3] - ref["pre"][..., :3]) <= tol).all(), name
                assert v["stats"]["gathers"] == ref["stats"]["gathers"], name
            # the order of the sums does not depend on the schedule that queued the requests
            for name in ("whole frame in one launch", "persistent tail"):
                assert same(variants[name]["pre"], got["pre"]), name
        # one map only, and a tile that is not a multiple of anything
        p = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED, shadow_bias=1e-4)
        for maps in ((md, None), (None, mc)):
            with ctx.option("defer_gathers", 0, 1):
                ref = sc.trace_tile(p, maps[0], maps[1], x0=17, y0=5, w=123, h=77)
            got = sc.trace_tile(p, maps[0], maps[1], x0=17, y0=5, w=123, h=77)
            assert (np.abs(got["pre"][..., :3] - ref["pre"][..., :3]) <= 2e-5 * np.maximum(1.0, np.abs(ref["pre"][..., :3]))).all()
            assert got["stats"]["gathers"] == ref["stats"]["gathers"]
        # maps large enough for the library to queue by itself (and balls next to the lamps dense enough for work items)
        ctx.set_option("defer_gathers", -1)
        sc2 = ctx.upload(load_scene(10, 640, 360))
        pe = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=300000, num_photons_caustic=300000, seed=SEED)
        lists, _ = sc2.emit_many_lists(pe, capi.SHARD_PACKED, strict=False)
        md2, mc2 = lists.photon_map(0), lists.photon_map(1)
        p2 = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=300000, num_photons_caustic=300000, seed=SEED, shadow_bias=1e-4)
        with ctx.option("defer_gathers", 0, -1):
            ref2 = sc2.trace_tile(p2, md2, mc2)
        got2 = sc2.trace_tile(p2, md2, mc2)
        with ctx.option("gather_split", 0, 4096):
            one2 = sc2.trace_tile(p2, md2, mc2)
        tol2 = 2e-5 * np.maximum(1.0, np.abs(ref2["pre"][..., :3]))
        assert (np.abs(got2["pre"][..., :3] - ref2["pre"][..., :3]) <= tol2).all() and (np.abs(one2["pre"][..., :3] - ref2["pre"][..., :3]) <= tol2).all()
        assert not same(got2["pre"], ref2["pre"])               # (it did take the other route)
        assert got2["stats"]["gathers"] == ref2["stats"]["gathers"] > 400000
        # BASELINE's full size: 1920x1080, 2 M + 2 M emitted photons (balls next to the lamps hold hundreds of thousands)
        sc3 = ctx.upload(load_scene(10, 1920, 1080))
        pe3 = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=1)
        lists3, _ = sc3.emit_many_lists(pe3, capi.SHARD_PACKED, strict=False)
        md3, mc3 = lists3.photon_map(0), lists3.photon_map(1)
        p3 = capi.default_params(precision=capi.PRECISION_F32, num_photons_diffuse=2_000_000, num_photons_caustic=2_000_000, seed=1, shadow_bias=1e-4)
        with ctx.option("defer_gathers", 0, -1):
            ref3 = sc3.trace_tile(p3, md3, mc3, want=("pre",))
        got3 = sc3.trace_tile(p3, md3, mc3, want=("pre",))
        assert (np.abs(got3["pre"][..., :3] - ref3["pre"][..., :3]) <= 3e-5 * np.maximum(1.0, np.abs(ref3["pre"][..., :3]))).all()
        assert got3["stats"]["gathers"] == ref3["stats"]["gathers"] > 4_000_000 and got3["stats"]["rays"] == ref3["stats"]["rays"]
        assert got3["stats"]["gather_saturations"] == 0 and got2["stats"]["gather_saturations"] == 0   # no partial sum left the accumulators' range
        assert same(sc3.trace_tile(p3, md3, mc3, want=("pre",))["pre"], got3["pre"])
        del lists3, md3, mc3, sc3, ref3, got3
        # rt_gather_radiance with k = 0 takes the same queue for many queries on a large map
        from raytracer_amd import synth
        hp = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=SEED)
        hits = synth.first_hit_queries(sc2, hp, 640, 360)
        assert hits.shape[0] > 100000
        for pm_, r_ in ((md2, 0.40), (mc2, 0.15)):
            with ctx.option("defer_gathers", 0, -1):
                lane = pm_.gather(hits, r=r_)
            queued = pm_.gather(hits, r=r_)
            assert not np.array_equal(queued, lane) and np.abs(queued - lane).max() <= 2e-5 * max(1.0, float(np.abs(lane).max()))
            assert np.allclose(queued, lane, rtol=3e-5, atol=1e-7 * float(np.abs(lane).max()))
            assert np.array_equal(pm_.gather(hits, r=r_), queued)                   # the same from run to run
            with ctx.option("defer_gathers", 1, -1):                                # few queries, forced through the queue
                few = pm_.gather(hits[:777], r=r_)
            assert np.allclose(few, lane[:777], rtol=3e-5, atol=1e-7 * float(np.abs(lane).max()))
    finally:
        ctx.set_option("defer_gathers", -1)


# ------------------------------------------------------------------ wave-cooperative k-nearest gather
def _brute_knn_gather(rec, hits, k):
    """fp64 brute force in the oracle's order (d2, then input index), sampled rows only"""
    out = np.zeros((hits.shape[0], 3))
    pos = rec[:, :3]
    for i in range(hits.shape[0]):
        idx, d2 = ora.knn_brute(pos, hits[i, :3].astype(np.float64), k)
        idx = idx[idx >= 0]
        w = np.maximum(0.0, rec[idx, 3:6] @ hits[i, 3:6].astype(np.float64))
        s = (rec[idx, 7:10] * rec[idx, 6:7] * w[:, None]).sum(axis=0)
        out[i] = s * hits[i, 6:9] / 255 / (np.pi * d2[len(idx) - 1])
    return out


def test_knn_gather_wave_kernel_equals_heap_kernel_and_brute_force(rt):
    capi, ctx = rt
    from raytracer_amd import synth
    rec = synth.box_photons(200_000, seed=11)
    hits = synth.box_queries(320, 200)                     # tile-ordered: a wave's 64 queries are an 8x8 pixel patch
    m = ctx.photon_map(rec)
    K = 64
    got = m.gather(hits, k=K)
    with ctx.option("knn_heap", 1, 0):
        heap = m.gather(hits, k=K)
    scale = max(1.0, float(np.abs(heap).max()))
    assert np.allclose(got, heap, rtol=2e-5, atol=1e-6 * scale)          # same photon sets; the sums run in another order
    rows = np.random.default_rng(3).choice(hits.shape[0], 300, replace=False)
    exp = _brute_knn_gather(rec, hits[rows], K)
    assert np.allclose(got[rows], exp, rtol=2e-5, atol=1e-6 * scale)
    # search balls three and eight times too large end in the same sets (candidate lists overflow: units cut into quarters,
    # single queries streamed by one wave each); a thousand times too large sends every query down the single-query path
    for sc_ in (3.0, 8.0, 1000.0):
        with ctx.option("knn_radius_scale", sc_, 0):
            again = m.gather(hits, k=K)
        assert np.allclose(again, heap, rtol=2e-5, atol=1e-6 * scale), sc_
    # queries in random order (a wave's 64 queries are far apart: nearly all of them end as single queries)
    perm = np.random.default_rng(4).permutation(hits.shape[0])[:20000]
    shuffled = m.gather(np.ascontiguousarray(hits[perm]), k=K)
    assert np.allclose(shuffled, heap[perm], rtol=2e-5, atol=1e-6 * scale)
    # ragged tail (nq not a multiple of 64), k = 1 and k = 192
    for k in (1, 192):
        a = m.gather(hits[:1000], k=k)
        with ctx.option("knn_heap", 1, 0):
            b = m.gather(hits[:1000], k=k)
        assert np.allclose(a, b, rtol=2e-5, atol=1e-6 * max(1.0, float(np.abs(b).max()))), k


def test_knn_gather_ties_at_the_kth_distance(rt):
    """photons on a lattice, queries on lattice points and cell centres: many exactly equal distances, so the
    k-th distance is usually a tie group that the lowest input indices must win"""
    capi, ctx = rt
    g = np.stack(np.meshgrid(np.arange(24), np.arange(24), np.arange(4), indexing="ij"), -1).reshape(-1, 3).astype(np.float64) * 0.25
    rng = np.random.default_rng(5)
    rec = np.zeros((g.shape[0], 11))
    rec[:, :3] = g[rng.permutation(g.shape[0])]
    rec[:, 3:6] = [0, 0, 1]
    rec[:, 6] = 1 / 255
    rec[:, 7:10] = rng.uniform(0, 255, (g.shape[0], 3))
    m = ctx.photon_map(rec)
    qx, qy = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
    qpos = np.stack([qx.ravel() * 0.125 + 2.0, qy.ravel() * 0.125 + 2.0, np.full(256, 0.25)], -1)
    hits = np.concatenate([qpos, np.tile([0, 0, 1.0], (256, 1)), np.full((256, 3), 255.0)], axis=1).astype(np.float32)
    for k in (5, 7, 27, 64):
        exp = _brute_knn_gather(rec, hits, k)
        for sc_ in (0, 1000.0):                  # the 64-query path; every query through the single-query path
            with ctx.option("knn_radius_scale", sc_, 0):
                got = m.gather(hits, k=k)
            assert np.allclose(got, exp, rtol=2e-5, atol=1e-6 * max(1.0, float(np.abs(exp).max()))), (k, sc_)


# ------------------------------------------------------------------ hierarchy scans (large scenes)
@pytest.mark.parametrize("preset", [10, 4, 7, 9])
def test_hierarchy_scans_equal_the_golden_scans(rt, preset):
    """option bvh = 1 builds the hierarchy for a small scene too: the fp64 scans through it must still be the reference's"""
    capi, ctx = rt
    b = load_bundle(f"closest_preset{preset}")
    with ctx.option("bvh", 1, -1):
        sc = ctx.upload(load_scene(preset))
    p = capi.default_params(precision=capi.PRECISION_F64)
    t, idx = sc.intersect(p, capi.SCAN_CLOSEST, b["ray_o"], b["ray_d"])
    assert np.array_equal(idx, b["idx_all"]) and same(t, b["t_all"])
    t, idx = sc.intersect(p, capi.SCAN_CLOSEST_NOLIGHT, b["ray_o"], b["ray_d"])
    assert np.array_equal(idx, b["idx_nolight"]) and same(t, b["t_nolight"])
    _, occ = sc.intersect(p, capi.SCAN_ANY, b["ray_o"], b["ray_d"], b["source_dist"])
    assert np.array_equal(occ, b["occluded"])
    # fp32: the same hits as the all-primitives fp32 scan (equal distances may be settled differently)
    flat = ctx.upload(load_scene(preset))
    p32 = capi.default_params(precision=capi.PRECISION_F32)
    t_h, i_h = sc.intersect(p32, capi.SCAN_CLOSEST, b["ray_o"], b["ray_d"])
    t_b, i_b = flat.intersect(p32, capi.SCAN_CLOSEST, b["ray_o"], b["ray_d"])
    assert (i_h == i_b).mean() > 0.999 and same(t_h[i_h == i_b], t_b[i_h == i_b])


def test_hierarchy_frames_equal_all_primitive_frames(rt):
    capi, ctx = rt
    # the reference's golden frames through the hierarchy: mode B with photon maps and direct light, mode A
    g = load_bundle("image_c1_modeB")
    ph = load_bundle("photons_c1")
    with ctx.option("bvh", 1, -1):
        sc = ctx.upload(load_scene(10, g.meta["W"], g.meta["H"]))
        sca = ctx.upload(load_rayjs_scene(3))
    ref = ctx.upload(load_scene(10, g.meta["W"], g.meta["H"]))
    md, mc = ctx.photon_map(ph["diffuse"]), ctx.photon_map(ph["caustic"])
    p = capi.default_params(precision=capi.PRECISION_F64, num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    a, b = sc.trace_tile(p, md, mc), ref.trace_tile(p, md, mc)
    assert same(a["pre"], b["pre"]) and a["stats"]["rays"] == b["stats"]["rays"] == 110353
    assert 0 < a["stats"]["prim_tests"] < b["stats"]["prim_tests"] + b["stats"]["shadow_rays"] * 33
    ga = load_bundle("image_rayjs3")
    pa = capi.default_params(precision=capi.PRECISION_F64, mode=capi.MODE_WHITTED, seed=ga.meta["seed"], **RAYJS_PARAMS)
    assert same(sca.trace_tile(pa)["pre"], ga["rgba"])
    # photon emission through the hierarchy: the reference's lists
    e = sc.emit_many(p)
    assert same(e["diffuse"], ph["diffuse"]) and same(e["caustic"], ph["caustic"])
    # a 10 000-primitive scene: hierarchy (chosen automatically) against the all-primitives scan, both precisions
    from raytracer_amd import synth
    flat = synth.many_primitive_scene(10000, seed=42, width=192, height=108)
    big = ctx.upload(flat)
    with ctx.option("bvh", 0, -1):
        brute = ctx.upload(flat)
    for prec in (capi.PRECISION_F64, capi.PRECISION_F32):
        q = capi.default_params(precision=prec, use_photon_maps=0, seed=3, max_depth=8, shadow_bias=1e-4)
        x, y = big.trace_tile(q), brute.trace_tile(q)
        if prec == capi.PRECISION_F64:
            assert same(x["pre"], y["pre"])
        else:
            assert same_px(x["pre"], y["pre"]).mean() > 0.995
        assert x["stats"]["rays"] == y["stats"]["rays"] or prec == capi.PRECISION_F32
        assert x["stats"]["prim_tests"] * 20 < y["stats"]["prim_tests"]       # > 95 % of the tests avoided
        # one child order for all rays instead of the near-child-first copy per direction octant: same frame, more tests
        with ctx.option("bvh_one_order", 1, 0):
            one = ctx.upload(flat)
        z = one.trace_tile(q)
        assert same(x["pre"], z["pre"]) and x["stats"]["rays"] == z["stats"]["rays"]
        assert x["stats"]["prim_tests"] < z["stats"]["prim_tests"]


@pytest.mark.parametrize("size", [(192, 108), (960, 540)])
def test_gathered_first_pass_equals_the_walk(rt, size):
    """Hierarchy scenes, fp32: the first pass scans, from LDS, the primitives its workgroup's pyramid can reach (lists made
    once per camera); with option cull_lists = 0 every camera ray walks the hierarchy instead.  Both find the closest
    primitive with ties going to the lowest index, so the frames must agree bit for bit -- at a size where most workgroups'
    lists fit their LDS budget and at one where many do not (those workgroups walk)."""
    capi, ctx = rt
    from raytracer_amd import synth
    W, H = size
    big = ctx.upload(synth.many_primitive_scene(10000, seed=42, width=W, height=H))
    for kw in (dict(use_direct_light=0), dict(shadow_bias=1e-4)):
        q = capi.default_params(precision=capi.PRECISION_F32, use_photon_maps=0, seed=3, max_bounces=8, **kw)
        got = big.trace_tile(q, want=("pre", "rgba8"))
        with ctx.option("cull_lists", 0, 1):
            ref = big.trace_tile(q, want=("pre", "rgba8"))
        assert same(got["pre"], ref["pre"]) and np.array_equal(got["rgba8"], ref["rgba8"]), kw
        for k in ("rays", "shadow_rays", "tail_rays", "bounces"):
            assert got["stats"][k] == ref["stats"][k], k
        assert got["stats"]["prim_tests"] != ref["stats"]["prim_tests"]          # (the two routes really differ)
        # a mode A frame of the same scene (ties go to the highest index there)
    flat_a = synth.many_primitive_scene(10000, seed=42, width=W, height=H)
    flat_a.shape_surface = np.tile(np.array([[0.3, 0.5]]), (flat_a.n_shapes, 1))          # ray.js's per-shape reflectance, specular
    big_a = ctx.upload(flat_a)
    qa = capi.default_params(precision=capi.PRECISION_F32, mode=capi.MODE_WHITTED, seed=3, max_trace_dist=100.0, max_depth=6)
    got = big_a.trace_tile(qa, want=("pre",))
    with ctx.option("cull_lists", 0, 1):
        ref = big_a.trace_tile(qa, want=("pre",))
    assert same(got["pre"], ref["pre"])


# ------------------------------------------------------------------ every preset of the reference
@pytest.mark.parametrize("name", ALL_PRESETS)
def test_every_preset_emits_and_renders_like_the_oracle(rt, name):
    """all 13 scenes of rayphoton.js, fp64, full pipeline (emission, both kd-trees, direct light, gathers) at a small
    size against the CPU oracle (which is pinned to the reference on the default scene)"""
    capi, ctx = rt
    W, H, ND, NC = 96, 64, 1500, 1500
    flat = load_packaged_preset(name, W, H)
    sc = ctx.upload(flat)
    # (max_bounces: the reference has no bounce cap -- paths caught between glass surfaces go on until maxDist is used up)
    p = capi.default_params(precision=capi.PRECISION_F64, num_photons_diffuse=ND, num_photons_caustic=NC, seed=SEED, max_bounces=1000000)
    e = sc.emit_many(p, strict=False)
    osc = ora.Scene(flat)
    op = ora.params(seed=SEED, num_photons_diffuse=ND, num_photons_caustic=NC)
    oe = ora.emit_many(osc, op)
    assert same(e["diffuse"], oe["diffuse"]) and same(e["caustic"], oe["caustic"])
    assert e["stats"]["attempts_phase1"] == oe["attempts1"] and e["stats"]["attempts_phase2"] == oe["attempts2"]
    md = ctx.photon_map(e["diffuse"]) if len(e["diffuse"]) else None
    mc = ctx.photon_map(e["caustic"]) if len(e["caustic"]) else None
    out = sc.trace_tile(p, md, mc)
    maps = ora.Maps(oe["diffuse"], oe["caustic"])
    pre, post, rays, _ = ora.render(osc, op, maps, 0, 0, W, H)
    err = np.abs(out["pre"][..., :3] - pre[..., :3]) / np.maximum(1.0, np.abs(pre[..., :3]))
    bad = ~((err <= 1e-9) | (np.isnan(out["pre"][..., :3]) & np.isnan(pre[..., :3]))).all(axis=-1)
    # photon positions are fp32 in the device tree: a photon within rounding of a search-ball boundary may flip
    assert bad.sum() <= 6, (name, int(bad.sum()))
    assert out["stats"]["rays"] == rays


def test_preview_levels_match_reference(rt):
    """super_sample_block = 3 and 9: the reference's traceOnCanvas(x, y, 1) and (x, y, 2) for every block of the C1 frame"""
    capi, ctx = rt
    g = load_bundle("image_c1_scaled")
    ph = load_bundle("photons_c1")
    W, H = g.meta["W"], g.meta["H"]
    sc = ctx.upload(load_scene(10, W, H))
    md, mc = ctx.photon_map(ph["diffuse"]), ctx.photon_map(ph["caustic"])
    for b in (3, 9):
        p = capi.default_params(precision=capi.PRECISION_F64, num_photons_diffuse=10000, num_photons_caustic=10000, seed=g.meta["seed"],
                                super_sample_block=b)
        out = sc.trace_tile(p, md, mc)
        want = g[f"post_block{b}"]
        assert out["post"].shape == want.shape
        assert (~same_px(out["post"], want)).sum() <= 2           # (fp32 photon positions in the device tree, as for the full frame)
        # a tile whose size is not a multiple of the block: the blocks that start inside it
        part = sc.trace_tile(p, md, mc, x0=b * 5, y0=b * 3, w=50, h=31)
        nb_x, nb_y = -(-50 // b), -(-31 // b)
        assert part["post"].shape == (nb_y, nb_x, 4)
        assert (~same_px(part["post"], want[3:3 + nb_y, 5:5 + nb_x])).sum() <= 1


def test_adaptive_radius_gather(rt):
    """the reference's commented-out search (rayphoton.js:1766-1778): triple r until 10 photons, drop results under 3"""
    capi, ctx = rt
    ph, q = load_bundle("photons_c1"), load_bundle("queries_c1")
    rec = ph["caustic"]
    m = ctx.photon_map(rec)
    nq = 400
    rng = np.random.default_rng(2)
    nrm = rng.normal(size=(nq, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    hits = np.concatenate([q["queries"][:nq], nrm, rng.uniform(0, 255, (nq, 3))], axis=1).astype(np.float32)
    got, r_used = m.gather_adaptive(hits, 0.05)
    pos = rec[:, :3].astype(np.float32).astype(np.float64)
    exp, r_exp = np.zeros((nq, 3)), np.zeros(nq)
    for i in range(nq):
        x = hits[i, :3].astype(np.float64)
        d2 = ((x - pos) ** 2).sum(axis=1)             # (same value as the device's sum order up to the last bit: see the margin below)
        r = 0.05
        for attempt in range(10):
            inside = d2 < r * r
            r_exp[i] = r
            if inside.sum() < 10:
                r *= 3
            else:
                break
        if inside.sum() >= 3:
            w = np.maximum(0.0, rec[inside, 3:6] @ hits[i, 3:6].astype(np.float64))
            exp[i] = (rec[inside, 7:10] * rec[inside, 6:7] * w[:, None]).sum(axis=0) * hits[i, 6:9] / 255 / (np.pi * r_exp[i] ** 2)
    assert np.allclose(r_used, r_exp, rtol=1e-6)
    assert np.allclose(got, exp, rtol=2e-5, atol=1e-6 * max(1.0, np.abs(exp).max()))
    assert len(np.unique(r_used)) >= 3               # several radii were in use
    with pytest.raises(capi.RtError):
        m.gather_adaptive(hits, 0.0)


def test_queued_frames_equal_finished_frames(rt):
    """rt_trace_tile without statistics and with device outputs only queues the frame: several frames queued back to
    back (different seeds, one shared workspace) must come out as the same frames rendered one at a time"""
    import torch
    capi, ctx = rt
    W, H = 320, 200
    sc = ctx.upload(load_scene(10, W, H))
    for prec, sub in ((capi.PRECISION_F32, 1), (capi.PRECISION_F64, 2)):
        ps = [capi.default_params(precision=prec, use_photon_maps=0, seed=s, sub_sample=sub) for s in (3, 4, 5)]
        want = [sc.trace_tile(p, want=("post", "rgba8")) for p in ps]
        post = [torch.full((H, W, 4), -1.0, dtype=torch.float64, device="cuda") for _ in ps]
        rgba = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in ps]
        torch.cuda.synchronize()
        for p, a, b in zip(ps, post, rgba):
            assert sc.trace_tile(p, want=(), out_post=a, out_rgba8=b, stats=False)["stats"] is None
        ctx.synchronize()
        for w, a, b in zip(want, post, rgba):
            assert same(a.cpu().numpy(), w["post"]) and np.array_equal(b.cpu().numpy(), w["rgba8"])


def test_two_frames_in_flight_equal_serial_frames(rt):
    """with n frames in flight consecutive queued frames rotate through n streams and workspaces (from three on the
    surviving paths are followed by the tail kernel with persistent lanes): same pixels"""
    import torch
    capi, ctx = rt
    W, H = 320, 200
    sc = ctx.upload(load_scene(10, W, H))
    ps = [capi.default_params(precision=capi.PRECISION_F32, use_photon_maps=0, seed=s) for s in range(11, 17)]
    want = [sc.trace_tile(p, want=("rgba8",))["rgba8"] for p in ps]
    for n in (2, 3, 8):
        out = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in ps]
        torch.cuda.synchronize()
        ctx.set_frames_in_flight(n)
        try:
            for p, b in zip(ps, out):
                sc.trace_tile(p, want=(), out_rgba8=b, stats=False)
            mid = sc.trace_tile(ps[0], want=("rgba8",))          # a finished frame in between waits for the queued ones
            for p, b in zip(ps[:2], out[:2]):
                sc.trace_tile(p, want=(), out_rgba8=b, stats=False)
            ctx.synchronize()
        finally:
            ctx.set_frames_in_flight(1)
        assert np.array_equal(mid["rgba8"], want[0]), n
        for w, b in zip(want, out):
            assert np.array_equal(b.cpu().numpy(), w), n
    with pytest.raises(capi.RtError):
        ctx.set_frames_in_flight(9)


# ------------------------------------------------------------------ the benchmark's sizes, through size-independent properties
def test_full_size_frame_properties(rt):
    """1920x1080 frame of the bench workload: tiles compose, queued == finished, the ray statistics add up, and the
    fp32 build stays within its stated tolerance of the fp64 build (per-pixel radiance within 1e-3*255 on >= 99.5 %)"""
    import torch
    capi, ctx = rt
    W, H = 1920, 1080
    sc = ctx.upload(load_scene(10, W, H))
    p32 = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1)
    p64 = capi.default_params(precision=capi.PRECISION_F64, use_direct_light=0, use_photon_maps=0, seed=1)
    whole = sc.trace_tile(p32, want=("pre", "rgba8"))
    assert whole["stats"]["rays"] == W * H + whole["stats"]["tail_rays"] and whole["stats"]["rays"] > W * H
    tiles = ((0, 0, 1000, 500), (1000, 0, 920, 500), (0, 500, 1920, 580))
    parts = [sc.trace_tile(p32, x0=x, y0=y, w=w, h=h, want=("pre",)) for (x, y, w, h) in tiles]
    for (x, y, w, h), part in zip(tiles, parts):
        assert same(whole["pre"][y:y + h, x:x + w], part["pre"])
    assert whole["stats"]["rays"] == sum(q["stats"]["rays"] for q in parts)
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    sc.trace_tile(p32, want=(), out_rgba8=out, stats=False)
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy(), whole["rgba8"])
    ref = sc.trace_tile(p64, want=("pre",))
    err = np.abs(whole["pre"][..., :3] - ref["pre"][..., :3]).max(axis=-1)
    assert (err <= 1e-3 * 255).mean() >= 0.995
    # the two builds follow the same paths almost everywhere: their segment counts differ by well under a percent
    assert abs(whole["stats"]["rays"] - ref["stats"]["rays"]) < 0.005 * ref["stats"]["rays"]


def test_full_size_knn_gather_properties(rt):
    """1 M photons, one k = 64 gather per pixel of a 1920x1080 frame: a sample of the wave kernel's results equals the
    per-lane heap kernel's, the results do not depend on how the queries are batched, and every result is finite"""
    capi, ctx = rt
    from raytracer_amd import synth
    rec = synth.box_photons(1_000_000, seed=7)
    hits = synth.box_queries(1920, 1080)
    m = ctx.photon_map(rec)
    got = m.gather(hits, k=64)
    assert got.shape == (hits.shape[0], 3) and np.isfinite(got).all() and (got >= 0).all()
    rows = np.arange(0, hits.shape[0] - 64 * 40, hits.shape[0] // 37)[:, None] + np.arange(64 * 40)[None, :]     # whole groups of 64
    rows = rows.reshape(-1)
    sub = m.gather(hits[rows], k=64)
    scale = max(1.0, float(np.abs(got).max()))
    assert np.allclose(sub, got[rows], rtol=2e-5, atol=1e-6 * scale)
    with ctx.option("knn_heap", 1, 0):
        heap = m.gather(hits[rows[:64 * 64]], k=64)
    assert np.allclose(got[rows[:64 * 64]], heap, rtol=2e-5, atol=1e-6 * scale)


# ------------------------------------------------------------------ schedules that must not change a single bit
@pytest.mark.parametrize("size", [(1920, 1080), (1000, 563), (1024, 72)])
def test_culling_and_tile_order_are_exact(rt, size):
    """The first pass's per-workgroup frustum culling (bounding spheres against the pyramid of the workgroup's pixels)
    and the 32x8 tile numbering of the paths are only schedules: the fp32 frame must be the frame of the all-primitives
    scan in raster order, bit for bit, at the benchmark's size, at a size that is no multiple of 32x8 and at one that is."""
    capi, ctx = rt
    W, H = size
    sc = ctx.upload(load_scene(10, W, H))
    for kw in (dict(use_direct_light=0, use_photon_maps=0), dict(use_photon_maps=0)):
        p = capi.default_params(precision=capi.PRECISION_F32, seed=1, shadow_bias=1e-4, **kw)
        ref = sc.trace_tile(p, want=("pre", "rgba8"))
        with ctx.option("no_cull", 1, 0):
            a = sc.trace_tile(p, want=("pre", "rgba8"))
        with ctx.option("raster_order", 1, 0):
            b = sc.trace_tile(p, want=("pre", "rgba8"))
        with ctx.option("no_cull", 1, 0), ctx.option("raster_order", 1, 0):
            c = sc.trace_tile(p, want=("pre", "rgba8"))
        with ctx.option("cull_lists", 0, 1):               # every workgroup culls for itself instead of reading the kept lists
            d = sc.trace_tile(p, want=("pre", "rgba8"))
        with ctx.option("primary_tiles", 1, -1):           # the first pass with persistent workgroups (by itself: from three frames in flight on)
            e_ = sc.trace_tile(p, want=("pre", "rgba8"))
        with ctx.option("primary_tiles", 1, -1), ctx.option("tail_persistent", 4, -1):
            f_ = sc.trace_tile(p, want=("pre", "rgba8"))
        for name, x in (("primary_tiles", e_), ("primary_tiles + persistent tail", f_)):
            assert same(x["pre"], ref["pre"]) and np.array_equal(x["rgba8"], ref["rgba8"]), (name, kw)
            for k in ("rays", "shadow_rays", "tail_rays", "bounces"):
                assert x["stats"][k] == ref["stats"][k], (name, k)
            # (a schedule-dependent count: which paths share a wave after the first pass, and with it the primitive
            # groups their wave skips, follows the order in which the first pass queued them -- 4.6 % seen on the 72-row frame)
            assert abs(x["stats"]["prim_tests"] - ref["stats"]["prim_tests"]) < 0.10 * ref["stats"]["prim_tests"]
        # (the first pass tests the same lists; later segments skip the groups no ray of their wave reaches, and which
        # paths share a wave there is not fixed from run to run)
        assert abs(d["stats"]["prim_tests"] - ref["stats"]["prim_tests"]) < 0.04 * ref["stats"]["prim_tests"]     # (and the two first-pass kernels queue the survivors in different orders)
        for name, x in (("no_cull", a), ("raster_order", b), ("no_cull+raster_order", c), ("cull_lists=0", d)):
            assert same(x["pre"], ref["pre"]) and np.array_equal(x["rgba8"], ref["rgba8"]), (name, kw)
            for k in ("rays", "shadow_rays", "tail_rays", "bounces"):
                assert x["stats"][k] == ref["stats"][k], (name, k)
        # the culled scan really tests fewer primitives
        assert max(ref["stats"]["prim_tests"], a["stats"]["prim_tests"]) < a["stats"]["rays"] * 33
        with ctx.option("scan_groups", 0, 1):              # (no scan groups either: every ray against every primitive)
            brute = ctx.upload(load_scene(10, W, H))
        with ctx.option("no_cull", 1, 0):
            e = brute.trace_tile(p, want=("pre",))
        assert same(e["pre"], ref["pre"]) and e["stats"]["prim_tests"] == e["stats"]["rays"] * 33
        brute.close()
    # a tile whose corner is not tile-aligned inside the big canvas, culled against unculled
    p = capi.default_params(precision=capi.PRECISION_F32, seed=2, use_direct_light=0, use_photon_maps=0)
    w, h = min(W - 37, 480), min(H - 11, 264)
    part = sc.trace_tile(p, x0=37, y0=11, w=w, h=h, want=("pre",))
    with ctx.option("no_cull", 1, 0):
        part_all = sc.trace_tile(p, x0=37, y0=11, w=w, h=h, want=("pre",))
    assert same(part["pre"], part_all["pre"])
    # the kept lists follow the tile's layout: the partial tile again, then the whole frame again
    assert same(sc.trace_tile(p, x0=37, y0=11, w=w, h=h, want=("pre",))["pre"], part["pre"])
    p = capi.default_params(precision=capi.PRECISION_F32, seed=1, shadow_bias=1e-4, use_photon_maps=0)
    assert same(sc.trace_tile(p, want=("pre",))["pre"], ref["pre"])


@pytest.mark.parametrize("name", ALL_PRESETS)
def test_scan_groups_are_exact(rt, name):
    """fp32 scans skip the groups of primitives (box faces, a lamp's shells ...) whose bounding sphere no ray of the wave
    reaches.  That is only a schedule: scans of incoherent random rays in all three modes, the emitted photon lists and a
    frame with direct light must equal, bit for bit, those of the same scene uploaded without groups -- also with the
    scene 200 times larger and far from the origin, where the sphere test's rounding matters most."""
    capi, ctx = rt
    W, H = 160, 96
    rng = np.random.default_rng(99)
    for scale, shift in ((1.0, 0.0), (200.0, 5000.0)):
        flat = load_packaged_preset(name, W, H)
        g = flat.shape_geom.copy()
        t = flat.shape_meta[:, 0]
        g[:, 0:3] = g[:, 0:3] * scale + shift                                   # every type keeps a point in slots 0-2
        g[:, 6] *= scale                                                         # radius / AC.x
        tri = (t == 4) | (t == 5)
        g[tri, 3:6] *= scale; g[tri, 7:9] *= scale                               # AB, rest of AC
        rnd = (t == 2) | (t == 3) | (t == 7)
        g[rnd, 7] *= scale; g[t == 2, 8] *= scale                                # truncMin / halfHeight / inner radius, truncMax
        cam = flat.camera.copy(); cam[0:3] = cam[0:3] * scale + shift
        flat2 = type(flat)(flat.shape_meta, g, flat.shape_colour, flat.light_shape, flat.light_data, cam, W, H, flat.materials, flat.shape_surface)
        with ctx.option("scan_groups", 0, 1):
            plain = ctx.upload(flat2)
        grouped = ctx.upload(flat2)
        p = capi.default_params(precision=capi.PRECISION_F32, seed=SEED, use_photon_maps=0, max_trace_dist=1000.0 * scale,
                                num_photons_diffuse=3000, num_photons_caustic=3000, max_bounces=200)
        n = 20000
        lo, hi = g[:, 0:3].min(axis=0) - 2 * scale, g[:, 0:3].max(axis=0) + 2 * scale
        ro = lo + rng.random((n, 3)) * (hi - lo)
        rd = rng.normal(size=(n, 3)); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
        # half of the rays in bundles of 64 with nearly one direction and origin (waves that do skip groups)
        ro[n // 2:] = np.repeat(ro[n // 2::64], 64, axis=0)[:n - n // 2] + 0.01 * scale * rng.normal(size=(n - n // 2, 3))
        rd[n // 2:] = np.repeat(rd[n // 2::64], 64, axis=0)[:n - n // 2] + 0.02 * rng.normal(size=(n - n // 2, 3))
        rd /= np.linalg.norm(rd, axis=1, keepdims=True)
        for mode in (capi.SCAN_CLOSEST, capi.SCAN_CLOSEST_NOLIGHT, capi.SCAN_ANY):
            sd = np.full(n, 30.0 * scale) if mode == capi.SCAN_ANY else None
            ta, ia = plain.intersect(p, mode, ro, rd, sd)
            tb, ib = grouped.intersect(p, mode, ro, rd, sd)
            assert np.array_equal(ia, ib) and same(ta, tb), (name, scale, mode)
        a, b = plain.trace_tile(p, want=("pre",)), grouped.trace_tile(p, want=("pre",))
        assert same(a["pre"], b["pre"]) and a["stats"]["rays"] == b["stats"]["rays"] and a["stats"]["shadow_rays"] == b["stats"]["shadow_rays"], (name, scale)
        if scale == 1.0:
            ea, eb = plain.emit_many(p, strict=False), grouped.emit_many(p, strict=False)
            assert same(ea["diffuse"], eb["diffuse"]) and same(ea["caustic"], eb["caustic"]), name
            assert ea["stats"]["attempts_phase1"] == eb["stats"]["attempts_phase1"] and ea["stats"]["attempts_phase2"] == eb["stats"]["attempts_phase2"]
        plain.close(); grouped.close()


# ------------------------------------------------------------------ BASELINE config 4: the 10 000-primitive scene
def test_config4_scene_matches_the_oracle(rt):
    """the synthetic 10 000-primitive scene (hierarchy chosen automatically), fp64, direct light on, paths of up to 8
    levels of glass: every pixel of a small canvas against the CPU oracle's all-primitives loops"""
    capi, ctx = rt
    from raytracer_amd import synth
    W, H = 96, 54
    flat = synth.many_primitive_scene(10000, seed=42, width=W, height=H)
    sc = ctx.upload(flat)
    p = capi.default_params(precision=capi.PRECISION_F64, use_photon_maps=0, seed=3, max_depth=8, max_bounces=1000000)
    out = sc.trace_tile(p)
    assert 0 < out["stats"]["prim_tests"] < 0.05 * out["stats"]["rays"] * flat.n_shapes      # the hierarchy is in use
    osc = ora.Scene(flat)
    op = ora.params(seed=3, use_photon_maps=0, max_depth=8)
    pre, post, rays, _ = ora.render(osc, op, None, 0, 0, W, H)
    # (a path's local terms are summed front to back on the GPU, inside-out in the recursion: <= 1e-9 relative)
    err = np.abs(out["pre"][..., :3] - pre[..., :3]) / np.maximum(1.0, np.abs(pre[..., :3]))
    assert ((err <= 1e-9) | (np.isnan(out["pre"][..., :3]) & np.isnan(pre[..., :3]))).all(), float(np.nanmax(err))
    assert same(out["pre"][..., 3], pre[..., 3]) and (~same_px(out["post"], post)).sum() <= 1
    assert out["stats"]["rays"] == rays
    # emission through the hierarchy against the oracle's loops, attempt by attempt
    got = sc.emit_photons(p, 1, 0, 3000)
    flag, rec = ora.emit_attempts(osc, ora.params(seed=3, max_depth=8), 1, 0, 3000)
    assert np.array_equal(np.nonzero(flag)[0], got[1]) and np.array_equal(flag[flag > 0].astype(np.uint8), got[2])
    assert same(rec[flag > 0], got[0])


def test_config4_full_size_frame_properties(rt):
    """3840x2160 frame of the 10 000-primitive scene, fp32, paths capped at 8 segments: tiles compose, a queued frame
    equals a finished one, the statistics add up, and the hierarchy's frame equals the all-primitives frame on a crop"""
    import torch
    capi, ctx = rt
    from raytracer_amd import synth
    W, H = 3840, 2160
    flat = synth.many_primitive_scene(10000, seed=42, width=W, height=H)
    sc = ctx.upload(flat)
    p = capi.default_params(precision=capi.PRECISION_F32, use_direct_light=0, use_photon_maps=0, seed=1, max_bounces=8)
    whole = sc.trace_tile(p, want=("rgba8",))
    st = whole["stats"]
    assert st["rays"] > W * H and st["bounces"] <= 8 and st["stack_overflows"] == 0
    assert st["prim_tests"] < 0.01 * st["rays"] * flat.n_shapes
    tiles = ((0, 0, 2000, 1000), (2000, 0, 1840, 1000), (0, 1000, 3840, 1160))
    parts = [sc.trace_tile(p, x0=x, y0=y, w=w, h=h, want=("rgba8",)) for (x, y, w, h) in tiles]
    for (x, y, w, h), part in zip(tiles, parts):
        assert np.array_equal(whole["rgba8"][y:y + h, x:x + w], part["rgba8"])
    assert st["rays"] == sum(q["stats"]["rays"] for q in parts)
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    sc.trace_tile(p, want=(), out_rgba8=out, stats=False)
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy(), whole["rgba8"])
    with ctx.option("bvh", 0, -1):
        brute = ctx.upload(flat)
    crop = (1700, 900, 256, 128)
    a = sc.trace_tile(p, x0=crop[0], y0=crop[1], w=crop[2], h=crop[3], want=("pre",))
    b = brute.trace_tile(p, x0=crop[0], y0=crop[1], w=crop[2], h=crop[3], want=("pre",))
    # (fp32: the hierarchy settles equal distances on the original index, the all-primitives scan on the type-sorted slot)
    assert same_px(a["pre"], b["pre"]).mean() > 0.999


# ------------------------------------------------------------------ kd-tree at the benchmark's size
def test_kd_build_1m_matches_the_oracle(rt):
    """1 000 000 photons on the faces of a box (the benchmark's map: heavy coordinate ties): perm and dim of every node
    against the oracle's makeTree"""
    capi, ctx = rt
    from raytracer_amd import synth
    rec = synth.box_photons(1_000_000, seed=7)
    pos = rec[:, :3].astype(np.float32)
    perm, dim = ctx.kd_build(pos)
    operm, odim = ora.kd_build(pos.astype(np.float64))
    assert np.array_equal(perm, operm) and np.array_equal(dim, odim)
    m = ctx.photon_map(rec)
    perm2, dim2 = m.nodes()
    assert np.array_equal(perm2, operm) and np.array_equal(dim2, odim)


def test_join_orders_the_main_stream_behind_frames_in_flight(rt):
    """with two frames in flight a consumer that only waits on rt_stream() must see finished frames after rt_join()"""
    import torch
    capi, ctx = rt
    W, H = 640, 360
    sc = ctx.upload(load_scene(10, W, H))
    ps = [capi.default_params(precision=capi.PRECISION_F32, use_photon_maps=0, seed=s) for s in range(21, 27)]
    want = [sc.trace_tile(p, want=("rgba8",))["rgba8"] for p in ps]
    out = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in ps]
    copies = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in ps]
    torch.cuda.synchronize()
    main = torch.cuda.ExternalStream(ctx.stream())
    ctx.set_frames_in_flight(3)
    try:
        for p, b in zip(ps, out):
            sc.trace_tile(p, want=(), out_rgba8=b, stats=False)
        ctx.join()
        with torch.cuda.stream(main):                         # the consumer's work, ordered behind rt_stream() only
            for b, c in zip(out, copies):
                c.copy_(b, non_blocking=True)
        main.synchronize()                                    # (not rt_synchronize: only the main stream is waited for)
        got = [c.cpu().numpy() for c in copies]
    finally:
        ctx.synchronize()
        ctx.set_frames_in_flight(1)
    for w, g in zip(want, got):
        assert np.array_equal(g, w)


# ------------------------------------------------------------------ device-resident photon lists, shards, row bands
def test_emission_segment_count_is_the_sequential_loops(c1_emit):
    """stats.segments = emitPhoton calls of exactly the attempts the sequential loop runs (not of whole batches)"""
    sc, p, e = c1_emit
    osc = ora.Scene(load_scene(10))
    oe = ora.emit_many(osc, ora.params(num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED))
    assert e["stats"]["segments"] == oe["segments"]


def test_device_lists_and_maps_equal_the_host_path(rt):
    """rt_emit_many_lists keeps the lists on the GPU: exported, they are rt_emit_many's lists; a photon map built straight
    from a list (full or packed 36-byte records) is the map built from the exported records"""
    capi, ctx = rt
    sc = ctx.upload(load_scene(10))
    hits = load_bundle("queries_c1")["queries"][:512]
    rng = np.random.default_rng(4)
    nrm = rng.normal(size=(hits.shape[0], 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    hits = np.concatenate([hits, nrm, rng.uniform(0, 255, (hits.shape[0], 3))], axis=1).astype(np.float32)
    for prec in (capi.PRECISION_F64, capi.PRECISION_F32):
        p = capi.default_params(precision=prec, num_photons_diffuse=6000, num_photons_caustic=3000, seed=SEED)
        whole = sc.emit_many(p)
        full, st = sc.emit_many_lists(p, capi.SHARD_FULL)
        assert st == whole["stats"]
        for which, name in ((0, "diffuse"), (1, "caustic")):
            rec, att = full.export(which)
            assert same(rec, whole[name]) and np.array_equal(att, whole[name + "_attempt"])
        packed, st2 = sc.emit_many_lists(p, capi.SHARD_PACKED)
        assert st2 == whole["stats"] and packed.words == 9
        for which, name in ((0, "diffuse"), (1, "caustic")):
            _, att = packed.export(which, records=False)
            assert np.array_equal(att, whole[name + "_attempt"])
            ref = ctx.photon_map(whole[name])
            rperm, rdim = ref.nodes()
            r = 0.4 if which == 0 else 0.15
            want = ref.gather(hits, r=r)
            for lst in (full, packed):
                m = lst.photon_map(which)
                perm, dim = m.nodes()
                assert np.array_equal(perm, rperm) and np.array_equal(dim, rdim)
                assert np.array_equal(m.gather(hits, r=r), want) and np.array_equal(m.gather(hits, k=16), ref.gather(hits, k=16))
            # the packed records themselves, through the host: rt_photon_map_build_packed
            w = whole[name]
            pk = np.concatenate([w[:, :6], w[:, 7:10] * w[:, 6:7]], axis=1).astype(np.float32)
            m = ctx.photon_map_packed(pk)
            assert np.array_equal(m.nodes()[0], rperm) and np.array_equal(m.gather(hits, r=r), want)


def test_sharded_emission_through_device_lists(c1_emit):
    """the multi-GPU emission protocol on one GPU: every batch is cut into 3 ranks' attempt sub-ranges, each emitted into
    its own shard (device memory), the shards applied in rank order -- the lists are rt_emit_many's"""
    import torch
    from raytracer_amd import parallel
    sc, p, whole = c1_emit
    capi = __import__("raytracer_amd.capi", fromlist=["x"])
    for fmt in (capi.SHARD_FULL, capi.SHARD_PACKED):
        eng = parallel.DeviceEngine(sc, p, fmt=fmt, device="cuda")
        res = parallel.emit_sharded(eng, batch=7001, world_override=3)
        assert res["attempts1"] == whole["stats"]["attempts_phase1"] and res["attempts2"] == whole["stats"]["attempts_phase2"]
        for which, name in ((0, "diffuse"), (1, "caustic")):
            rec, att = eng.lists.export(which, records=fmt == capi.SHARD_FULL)
            assert np.array_equal(att, whole[name + "_attempt"])
            if rec is not None:
                assert same(rec, whole[name])


@pytest.mark.parametrize("size,band,count", [((640, 360), 8, 3), ((1000, 563), 16, 8), ((320, 40), 32, 2)])
def test_row_bands_assemble_to_the_whole_frame(rt, size, band, count):
    """rt_params.band_*: every rank's interleaved row bands, put back together, are the undivided frame bit for bit"""
    capi, ctx = rt
    W, H = size
    sc = ctx.upload(load_scene(10, W, H))
    for prec, kw in ((capi.PRECISION_F32, dict(use_direct_light=0, use_photon_maps=0)), (capi.PRECISION_F64, dict(use_photon_maps=0, sub_sample=2))):
        if prec == capi.PRECISION_F64 and W > 700:
            continue
        p = capi.default_params(precision=prec, seed=9, shadow_bias=1e-4, **kw)
        whole = sc.trace_tile(p, want=("pre", "rgba8"))
        got = np.full_like(whole["pre"], -1.0)
        got8 = np.zeros_like(whole["rgba8"])
        rays = 0
        for idx in range(count):
            q = capi.default_params(precision=prec, seed=9, shadow_bias=1e-4, band_rows=band, band_count=count, band_index=idx, **kw)
            rows = np.array([y for y in range(H) if (y // band) % count == idx], dtype=np.int64)
            assert ctx.band_rows(H, band, count, idx) == rows.size
            part = sc.trace_tile(q, want=("pre", "rgba8"))
            if rows.size == 0:
                continue
            assert part["pre"].shape[0] == rows.size
            got[rows] = part["pre"]
            got8[rows] = part["rgba8"]
            rays += part["stats"]["rays"]
        assert same(got, whole["pre"]) and np.array_equal(got8, whole["rgba8"]) and rays == whole["stats"]["rays"]
    with pytest.raises(capi.RtError):
        sc.trace_tile(capi.default_params(band_rows=12, band_count=2, band_index=0))
    # the multi-GPU protocol's batched form: every rank's bands of several frames queued back to back (frames in flight)
    import torch
    from raytracer_amd import parallel
    p = capi.default_params(precision=capi.PRECISION_F32, seed=9, use_direct_light=0, use_photon_maps=0)
    whole8 = sc.trace_tile(p, want=("rgba8",))["rgba8"]
    eng = parallel.DeviceEngine(sc, p, device="cuda", with_lists=False)
    for idx in range(count):
        rows = parallel.band_rows_of(H, band, count, idx)
        parts = eng.render_bands_batch(W, H, band, count, idx, 5)
        assert tuple(parts.shape) == (5, rows.size, W, 4)
        for f in parts.cpu().numpy():
            assert np.array_equal(f, whole8[rows])
