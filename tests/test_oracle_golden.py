"""Pins the CPU oracle (oracle/rt_oracle.c) to fixtures produced by running the reference
itself (oracle/harness/gen_golden.js -> tests/golden/).  CPU only."""
import numpy as np
import pytest

from golden_io import RAYJS_PARAMS, load_bundle, load_rayjs_scene, load_scene
from oracle import oracle as ora

SEED = 12345
AUX = 3


def same(a, b):
    """bit-equality that treats NaN == NaN"""
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert ora.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ora.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ora.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_intersect_dist_all_types_bit_exact():
    b = load_bundle("intersect")
    meta = np.zeros((b["t"].shape[0], 4), np.uint8)
    meta[:, 0] = b["shape_meta"][:, 0]
    meta[:, 2] = b["shape_meta"][:, 1]
    t = ora.intersect_batch(b["ray_o"], b["ray_d"], meta, b["shape_g"])
    assert same(t, b["t"])
    # Ray ctor normalisation (rayphoton.js:484-487) from the raw directions
    t2 = ora.intersect_batch(b["ray_o"], b["ray_d_raw"], meta, b["shape_g"], normalize_dir=True)
    assert same(t2, b["t"])
    per = b.meta["per_type"]
    for k in range(8):       # every type has both hits and misses in the fixture
        tk = b["t"][k * per:(k + 1) * per]
        assert np.isnan(tk).any() and (~np.isnan(tk)).any()


@pytest.mark.parametrize("preset", [10, 4, 7, 9])
def test_closest_and_any_hit_scans(preset):
    b = load_bundle(f"closest_preset{preset}")
    sc = ora.Scene(load_scene(preset))
    t, idx = ora.scan_batch(sc, 0, b["ray_o"], b["ray_d"])
    assert np.array_equal(idx, b["idx_all"]) and same(t, b["t_all"])
    t, idx = ora.scan_batch(sc, 1, b["ray_o"], b["ray_d"])
    assert np.array_equal(idx, b["idx_nolight"]) and same(t, b["t_nolight"])
    _, occ = ora.scan_batch(sc, 2, b["ray_o"], b["ray_d"], b["source_dist"])
    assert np.array_equal(occ, b["occluded"])
    assert (b["idx_all"] >= 0).sum() > 1000 and b["occluded"].sum() > 100


def test_samplers_create_photon_perturb_disc():
    b = load_bundle("samples_preset10")
    sc = ora.Scene(load_scene(10))
    cp = b["create_photon"]
    for i in range(cp.shape[0]):
        _, out = ora.create_photon(sc, SEED, AUX, 3000, i)
        assert out[7] == cp[i, 7] == 5          # always 5 draws (rayphoton.js:1326-1344)
        assert same(out[:7], cp[i, :7])
    vp = b["vec_perturb"]
    for i in range(vp.shape[0]):
        first = 0 if i % 4 < 2 else 3
        out = ora.vec_perturb(SEED, AUX, 3001, i, first, vp[i, :3])
        assert same(out, vp[i, 3:])
    ds = b["disc_sample"]
    for i in range(ds.shape[0]):
        out = ora.disc_sample(SEED, AUX, 3002, i, 4, [1, 2, 3], ds[i, 3], ds[i, :3])
        assert same(out, ds[i, 4:])


@pytest.fixture(scope="module")
def c1_emission():
    sc = ora.Scene(load_scene(10))
    p = ora.params(num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    return ora.emit_many(sc, p)


def test_emission_c1_lists_match_reference(c1_emission):
    g = load_bundle("photons_c1")
    e = c1_emission
    assert not e["capped"]
    assert e["attempts1"] == g.meta["stats"]["attempts1"] and e["attempts2"] == g.meta["stats"]["attempts2"]
    assert e["draws"] == g.meta["stats"]["draws"]
    for name in ("diffuse", "caustic"):
        tag = g[name + "_tag"]
        assert np.array_equal(e[name + "_tag"], tag.astype(np.int64))      # isCaustic, phase, attempt
        got, want = e[name], g[name]
        assert got.shape == want.shape
        assert same(got, want)      # bit for bit, incl. the sin/cos of vecPerturb / discSample (fdlibm restated)


def test_find_median_table():
    g = load_bundle("kd_c1")
    fm = g["find_median"]
    assert [ora.find_median(n) for n in range(1, 4097)] == fm.tolist()


def test_kd_build_and_queries_match_reference():
    ph = load_bundle("photons_c1")
    kd = load_bundle("kd_c1")
    q = load_bundle("queries_c1")
    trees = {}
    for name in ("diffuse", "caustic"):
        pos = ph[name][:, :3].astype(np.float32).astype(np.float64)      # fp32-rounded, as the fixture was built
        perm, dim = ora.kd_build(pos)
        assert np.array_equal(perm, kd[name + "_perm"]) and np.array_equal(dim, kd[name + "_dim"])
        trees[name] = (pos, perm, dim)
    for key, r in q.meta["radii"].items():
        pos, perm, dim = trees[key.split("_")[0]]
        offs, want = q[key + "_offsets"], q[key + "_indices"]
        for i in range(q["queries"].shape[0]):
            got = ora.kd_radius(pos, perm, dim, q["queries"][i], r)
            assert np.array_equal(got, want[offs[i]:offs[i + 1]]), (key, i)
    for name in ("diffuse", "caustic"):
        pos = trees[name][0]
        for i in range(0, q["queries"].shape[0], 8):
            idx, d2 = ora.knn_brute(pos, q["queries"][i], q.meta["K"])
            assert np.array_equal(idx, q[f"knn_{name}_idx"][i]) and np.array_equal(d2, q[f"knn_{name}_d2"][i])


def test_nearest_neighbour_and_distance_matches_reference():
    """kdtree.js:10-62 run by the reference on trees of 10 000 / 9 999 / 2 / 1 photons: the closest node of one
    root-to-leaf path, [undefined, undefined] where the path leaves through a missing upper child"""
    ph = load_bundle("photons_c1")
    g = load_bundle("nearest_c1")
    assert g.meta["undefined_results"]["diffuse"] >= 1 and g.meta["undefined_results"]["two"] > 100
    for name, (which, n) in g.meta["trees"].items():
        pos = ph[which][:n, :3].astype(np.float32).astype(np.float64)
        perm, dim = ora.kd_build(pos)
        for pre in (name + "_", name + "_own_"):
            qs, want_i, want_d = g[pre + "queries"], g[pre + "idx"], g[pre + "dist"]
            step = 1 if qs.shape[0] < 4000 else 3
            for i in range(0, qs.shape[0], step):
                idx, dist = ora.kd_nearest(pos, perm, dim, qs[i])
                assert idx == want_i[i] and (dist == want_d[i] or (idx < 0 and np.isnan(dist) and np.isnan(want_d[i]))), (pre, i)
        undef = np.flatnonzero(g[name + "_own_idx"] < 0)
        for i in undef:                                # (with step 3 the single undefined result of a big tree could be skipped)
            assert ora.kd_nearest(pos, perm, dim, g[name + "_own_queries"][i])[0] == -1


def test_kd_build_empty_raises():
    with pytest.raises(ValueError):                  # kdtree.js:6
        ora.kd_build(np.zeros((0, 3)))


def test_image_c1_mode_b(c1_emission):
    """Full C1 frame: oracle emission -> oracle trees -> oracle render vs the reference's image."""
    g = load_bundle("image_c1_modeB")
    W, H = g.meta["W"], g.meta["H"]
    sc = ora.Scene(load_scene(10, W, H))
    # feed the reference's own photon lists so that ulp drift in emission does not leak in
    ph = load_bundle("photons_c1")
    maps = ora.Maps(ph["diffuse"], ph["caustic"])
    p = ora.params(num_photons_diffuse=10000, num_photons_caustic=10000, seed=SEED)
    pre, post, rays, tests = ora.render(sc, p, maps, 0, 0, W, H)
    want_pre, want_post = g["pre_floor"], g["post_floor"]
    assert same(pre, want_pre)          # every pixel, bit for bit (fp64, same operation order)
    assert same(post, want_post)
    assert rays > W * H


@pytest.mark.parametrize("preset,rays", [(2, 52993), (3, 59233), (0, 58604), (1, 90165)])
def test_image_mode_a_whitted(preset, rays):
    """the older tracer (ray.js:801-993): every pixel of three presets, bit for bit"""
    g = load_bundle(f"image_rayjs{preset}")
    osc = ora.Scene(load_rayjs_scene(preset))
    op = ora.params(seed=g.meta["seed"], **RAYJS_PARAMS)
    raw, post, n = ora.render_whitted(osc, op, 0, 0, g.meta["W"], g.meta["H"])
    assert same(raw, g["rgba"]) and n == rays
    assert same(post, g["rgba"])                 # one sub-sample: putPixel gets floor(colour / 1)


def test_image_c1_preview_levels(c1_emission):
    """traceOnCanvas(x, y, scale > 0) (rayphoton.js:1589-1596): one jittered sample per 3x3 and per 9x9 block"""
    g = load_bundle("image_c1_scaled")
    ph = load_bundle("photons_c1")
    W, H = g.meta["W"], g.meta["H"]
    osc = ora.Scene(load_scene(10, W, H))
    maps = ora.Maps(ph["diffuse"], ph["caustic"])
    for b in (3, 9):
        op = ora.params(seed=g.meta["seed"], num_photons_diffuse=10000, num_photons_caustic=10000, super_sample_block=b)
        _, post, _, _ = ora.render(osc, op, maps, 0, 0, W, H)
        assert same(post, g[f"post_block{b}"])
