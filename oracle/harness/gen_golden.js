'use strict';
// TEST INFRASTRUCTURE (oracle harness) -- runs only in the build container (needs the
// reference checkout); writes data-only fixtures into tests/golden/.
//
//   node oracle/harness/gen_golden.js [scenes] [intersect] [closest] [samples] [nearest] [c1] [c1image]
//
// Every number written here is produced by evaluating the reference's own functions
// (rayphoton.js / kdtree.js) under a keyed Philox Math.random; the only logic restated in
// this file is (a) the 10-line attempt loop of emitManyPhotons (rayphoton.js:1361-1381), so
// that each attempt can be given its own RNG stream, (b) the 5-line closest/any-hit scan
// loops (rayphoton.js:1644-1650, :1469-1475, :1854-1857) and (c) the primary-ray set-up of
// traceOnCanvas (rayphoton.js:1597-1609) so that pre-floor radiance can be captured.

const path = require('path');
const fs = require('fs');
const { loadRayphoton, stubCtx } = require('./ref_loader');
const { flattenScene, flattenShape } = require('./flatten_ref');
const { writeBundle } = require('./bundle');
const PH = require('./philox');

const OUT = path.resolve(__dirname, '../../tests/golden');
const SEED = 12345;
const W = 320, H = 240;

function buildScene(preset, opts) {
  const L = loadRayphoton(Object.assign({ seed: SEED, subSample: 1 }, opts || {}));
  const { ctx, px } = stubCtx(W, H);
  const scene = new L.ex.Scene(ctx);
  scene.emitManyPhotons = function () {};          // emission is driven by the keyed loop below
  L.rng.setStream(PH.DOMAIN_AUX, 1000 + (typeof preset === 'number' ? preset : 99), 0);
  scene.loadPreset(preset);
  return { L, scene, px, ctx };
}

// ---------------------------------------------------------------- scenes
function genScenes() {
  for (const preset of [10, 4, 7, 9, 2, 0]) {
    const { L, scene } = buildScene(preset);
    const flat = flattenScene(scene, L.ex, preset);
    fs.writeFileSync(path.join(OUT, `scene_preset${preset}_flat.json`), JSON.stringify(flat));
    console.log(`scene preset ${preset}: ${flat.shapes.length} shapes, ${flat.lights.length} lights`);
  }
}

// every preset of rayphoton.js (:731-1305) as a flat scene description, for the host's constructor check and the
// all-preset parity tests.  The RNG-built presets (5, 9, 'giacometti') are the instances drawn from stream
// (AUX, 1000 + id; 1099 for the named ones).
function genPresets() {
  const dir = OUT;
  for (const preset of [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 'rmt', 'giacometti']) {
    let built;
    try { built = buildScene(preset); } catch (e) { console.log(`preset ${preset}: reference threw ${e.message}`); continue; }
    const flat = flattenScene(built.scene, built.L.ex, preset);
    fs.writeFileSync(path.join(dir, `scene_preset${preset}_flat.json`), JSON.stringify(flat));
    console.log(`preset ${preset}: ${flat.shapes.length} shapes, ${flat.lights.length} lights`);
  }
}

// ---------------------------------------------------------------- intersect
function genIntersect() {
  const L = loadRayphoton({ seed: SEED });
  const ex = L.ex, rng = L.rng;
  rng.setStream(PH.DOMAIN_AUX, 1, 0);
  const R = () => rng.next();
  const U = (a, b) => a + (b - a) * R();
  const V = (s) => [U(-s, s), U(-s, s), U(-s, s)];
  const COL = ex.COL_WHITE;
  const makers = {
    plane: () => new ex.Plane(V(5), V(1), COL, 6),
    sphere: () => new ex.Sphere(V(5), U(0.2, 3), COL, 6),
    hemisphere: () => {
      const r = U(0.3, 3);
      const k = R();
      const tmin = (k < 0.3) ? 0 : U(-0.8, 0.8) * r;
      const tmax = (R() < 0.5) ? undefined : tmin + U(0.05, 1) * r;
      return new ex.Hemisphere(V(5), r, V(1), tmin, tmax, R() < 0.5, COL, 6);
    },
    cylinder: () => new ex.Cylinder(V(5), (R() < 0.15) ? [0, 0, 1] : V(1), U(0.2, 2), (R() < 0.25) ? undefined : U(0.2, 3), R() < 0.5, COL, 6),
    triangle: () => new ex.Triangle(V(5), V(3), V(3), COL, 6),
    square: () => new ex.Square(V(5), V(3), V(3), COL, 6),
    disc: () => new ex.Disc(V(5), U(0.2, 3), V(1), COL, 6),
    annulus: () => { const ro = U(0.5, 3); return new ex.Annulus(V(5), ro, ro * U(0.1, 0.9), V(1), COL, 6); },
  };
  const anchor = (s) => s.centre || s.origin || s.vtxA;
  const size = (s) => s.radius || s.outerRadius || 2;
  const PER_TYPE = 1536;
  const types = Object.keys(makers);
  const n = PER_TYPE * types.length;
  const shapeG = new Float64Array(n * 12), shapeMeta = new Int32Array(n * 2);
  const rayO = new Float64Array(n * 3), rayDraw = new Float64Array(n * 3), rayD = new Float64Array(n * 3);
  const tOut = new Float64Array(n);
  let k = 0, hits = 0;
  for (const type of types) {
    for (let i = 0; i < PER_TYPE; i++, k++) {
      const s = makers[type]();
      let o = V(8), d;
      const mode = i % 16;
      const a = anchor(s), sz = size(s);
      const target = [a[0] + U(-1.2, 1.2) * sz, a[1] + U(-1.2, 1.2) * sz, a[2] + U(-1.2, 1.2) * sz];
      d = [target[0] - o[0], target[1] - o[1], target[2] - o[2]];
      if (mode === 1) {                       // origin inside / near the anchor
        o = [a[0] + U(-0.3, 0.3) * sz, a[1] + U(-0.3, 0.3) * sz, a[2] + U(-0.3, 0.3) * sz]; d = V(1);
      } else if (mode === 2 && (s.normalDir || s.axis)) {   // exactly perpendicular to the normal / axis
        const nn = s.normalDir || s.axis;
        const m = [1, 0, 0];
        const c = [nn[1] * m[2] - nn[2] * m[1], nn[2] * m[0] - nn[0] * m[2], nn[0] * m[1] - nn[1] * m[0]];
        d = (c[0] * c[0] + c[1] * c[1] + c[2] * c[2] < 1e-6) ? [0, 1, 0] : c;
      } else if (mode === 3 && s.axis) {      // parallel to the cylinder axis (qRoots a==0 quirk, rayphoton.js:1884-1889)
        d = (R() < 0.5) ? s.axis.slice() : [0, 0, 1];
      } else if (mode === 4) {                // origin exactly on the surface: shoot a ray, restart from the hit
        const r0 = new ex.Ray(o, d), t0 = r0.intersectDist(s);
        if (t0 !== undefined) { o = [o[0] + t0 * r0.dir[0], o[1] + t0 * r0.dir[1], o[2] + t0 * r0.dir[2]]; d = V(1); }
      } else if (mode === 5) {                // degenerate direction (vecNormalize -> [0,0,1], rayphoton.js:109-110)
        d = [1e-4 * U(-1, 1), 1e-4 * U(-1, 1), 1e-4 * U(-1, 1)];
      } else if (mode === 6 && s.normalDir) { // almost parallel to a planar shape
        const nn = s.normalDir, t = V(1);
        const dot = t[0] * nn[0] + t[1] * nn[1] + t[2] * nn[2];
        const eps = U(-3e-6, 3e-6);
        d = [t[0] - (dot - eps) * nn[0], t[1] - (dot - eps) * nn[1], t[2] - (dot - eps) * nn[2]];
      }
      const ray = new ex.Ray(o, d);
      const t = ray.intersectDist(s);
      const f = flattenShape(s, 0, null);
      for (let j = 0; j < 12; j++) shapeG[k * 12 + j] = f.g[j];
      shapeMeta[k * 2] = f.typeId; shapeMeta[k * 2 + 1] = f.flags;
      for (let j = 0; j < 3; j++) { rayO[k * 3 + j] = o[j]; rayDraw[k * 3 + j] = d[j]; rayD[k * 3 + j] = ray.dir[j]; }
      tOut[k] = (t === undefined) ? NaN : t;
      if (t !== undefined) hits++;
    }
  }
  writeBundle(path.join(OUT, 'intersect'), {
    shape_g: { data: shapeG, shape: [n, 12] }, shape_meta: { data: shapeMeta, shape: [n, 2] },
    ray_o: { data: rayO, shape: [n, 3] }, ray_d_raw: { data: rayDraw, shape: [n, 3] },
    ray_d: { data: rayD, shape: [n, 3] }, t: tOut,
  }, { per_type: PER_TYPE, types, note: 't = NaN encodes undefined (miss)' });
  console.log(`intersect: ${n} pairs, ${hits} hits`);
}

// ---------------------------------------------------------------- closest / any-hit
function closestScan(scene, ray, skipLights) {      // rayphoton.js:1642-1650 / :1467-1475
  let min = Infinity, minS = -1;
  for (let s = 0; s < scene.shapes.length; s++) {
    const t = ray.intersectDist(scene.shapes[s]);
    if (t > 1e-6 && t < min && !(skipLights && scene.shapes[s].isLight)) { min = t; minS = s; }
  }
  return [min, minS];
}
function anyHitScan(scene, ray, sourceDist) {        // rayphoton.js:1853-1857
  for (let s2 = 0; s2 < scene.shapes.length; s2++) {
    const t = ray.intersectDist(scene.shapes[s2]);
    if (t > 1e-6 && t < sourceDist - 1e-6 && !scene.shapes[s2].isLight) return 1;
  }
  return 0;
}
function primaryRay(L, scene, x, y) {                // rayphoton.js:1597-1609 with SUB_SAMPLE = 1, scale 0
  const cam = scene.camera;
  let u = ((x + (0 + L.rng.next()) / 1) * 2 / scene.canvasWidth) - 1;
  let v = -(((y + (0 + L.rng.next()) / 1) * 2 / scene.canvasHeight) - 1);
  u *= cam.fovScaleWidth; v *= cam.fovScaleHeight;
  const xyz = cam.toXYZ([u, v, -1]);
  const o = cam.origin;
  return new L.ex.Ray(o, [xyz[0] - o[0], xyz[1] - o[1], xyz[2] - o[2]]);
}

function genClosest() {
  const N = 4096;
  for (const preset of [10, 4, 7, 9]) {
    const { L, scene } = buildScene(preset);
    const rng = L.rng;
    const rayO = new Float64Array(N * 3), rayD = new Float64Array(N * 3), srcDist = new Float64Array(N);
    const tAll = new Float64Array(N), tNoLight = new Float64Array(N);
    const idxAll = new Int32Array(N), idxNoLight = new Int32Array(N), occ = new Int32Array(N);
    for (let i = 0; i < N; i++) {
      rng.setStream(PH.DOMAIN_AUX, 2000 + preset, i);
      let ray = primaryRay(L, scene, Math.floor(rng.next() * W), Math.floor(rng.next() * H));
      if (i % 2 === 1) {                      // secondary-like ray: start on the first hit, leave in a random direction
        const [t0, s0] = closestScan(scene, ray, false);
        if (s0 >= 0) {
          const p = [ray.origin[0] + t0 * ray.dir[0], ray.origin[1] + t0 * ray.dir[1], ray.origin[2] + t0 * ray.dir[2]];
          ray = new L.ex.Ray(p, [2 * rng.next() - 1, 2 * rng.next() - 1, 2 * rng.next() - 1]);
        }
      }
      const [ta, sa] = closestScan(scene, ray, false);
      const [tn, sn] = closestScan(scene, ray, true);
      const sd = 0.5 + 14.5 * rng.next();
      for (let j = 0; j < 3; j++) { rayO[i * 3 + j] = ray.origin[j]; rayD[i * 3 + j] = ray.dir[j]; }
      tAll[i] = ta; idxAll[i] = sa; tNoLight[i] = tn; idxNoLight[i] = sn; srcDist[i] = sd;
      occ[i] = anyHitScan(scene, ray, sd);
    }
    writeBundle(path.join(OUT, `closest_preset${preset}`), {
      ray_o: { data: rayO, shape: [N, 3] }, ray_d: { data: rayD, shape: [N, 3] },
      t_all: tAll, idx_all: idxAll, t_nolight: tNoLight, idx_nolight: idxNoLight,
      source_dist: srcDist, occluded: occ,
    }, { preset, note: 't = Infinity, idx = -1 encode a miss' });
    console.log(`closest preset ${preset}: ${N} rays`);
  }
}

// ---------------------------------------------------------------- samplers
function genSamples() {
  const { L, scene } = buildScene(10);
  const N = 512;
  const out = new Float64Array(N * 8), pert = new Float64Array(N * 6), disc = new Float64Array(N * 7);
  for (let i = 0; i < N; i++) {
    L.rng.setStream(PH.DOMAIN_AUX, 3000, i);
    const p = scene.createPhoton();
    out.set([p.origin[0], p.origin[1], p.origin[2], p.dir[0], p.dir[1], p.dir[2], p.power, L.rng.draw], i * 8);
    L.rng.setStream(PH.DOMAIN_AUX, 3001, i);
    const v = (i % 4 === 0) ? [1, 0, 0] : (i % 4 === 1) ? [0, 0, -1]
      : [2 * L.rng.next() - 1, 2 * L.rng.next() - 1, 2 * L.rng.next() - 1];
    const w = L.ex.vecPerturb(v);
    pert.set([v[0], v[1], v[2], w[0], w[1], w[2]], i * 6);
    L.rng.setStream(PH.DOMAIN_AUX, 3002, i);
    const nrm = [2 * L.rng.next() - 1, 2 * L.rng.next() - 1, 2 * L.rng.next() - 1];
    const rad = 0.25 + L.rng.next();
    const q = L.ex.discSample([1, 2, 3], rad, nrm);
    disc.set([nrm[0], nrm[1], nrm[2], rad, q[0], q[1], q[2]], i * 7);
  }
  writeBundle(path.join(OUT, 'samples_preset10'), {
    create_photon: { data: out, shape: [N, 8] }, vec_perturb: { data: pert, shape: [N, 6] },
    disc_sample: { data: disc, shape: [N, 7] },
  }, { seed: SEED, note: 'create_photon rows: origin3 dir3 power draws; streams (AUX,3000,i); vec_perturb rows: v3 out3 streams (AUX,3001,i) -- v drawn first for i%4>=2; disc_sample rows: n3 radius out3, centre [1,2,3]' });
  console.log('samples written');
}

// ---------------------------------------------------------------- C1: emission, kd, queries, image
function emitKeyed(L, scene, ND, NC) {                 // restates rayphoton.js:1361-1381 with per-attempt streams
  const ex = L.ex, rng = L.rng;
  let curPhase = 0, curAttempt = 0;
  const origStore = scene.storePhoton.bind(scene);
  scene.storePhoton = function (photon) {
    const nd = scene.photonListDiffuse.length, nc = scene.photonListCaustic.length;
    origStore(photon);
    if (scene.photonListDiffuse.length > nd || scene.photonListCaustic.length > nc) {
      photon.__attempt = curAttempt; photon.__phase = curPhase;
    }
  };
  const stats = { attempts1: 0, attempts2: 0, draws: 0 };
  const before = rng.total;
  let attempt = 0;
  curPhase = 1;
  while (scene.photonListDiffuse.length < ND) {
    rng.setStream(PH.DOMAIN_EMIT_PHASE1, attempt >>> 0, Math.floor(attempt / 4294967296));
    curAttempt = attempt;
    const photon = scene.createPhoton();
    scene.emitPhoton(photon, 0, false, [ex.MAT_AIR]);
    attempt++;
    if (attempt > 100 * ND) break;
  }
  stats.attempts1 = attempt;
  attempt = 0; curPhase = 2;
  while (scene.photonListCaustic.length < NC) {
    rng.setStream(PH.DOMAIN_EMIT_PHASE2, attempt >>> 0, Math.floor(attempt / 4294967296));
    curAttempt = attempt;
    const photon = scene.createPhoton();
    scene.emitPhoton(photon, 0, true, [ex.MAT_AIR]);
    attempt++;
    if (attempt > 100 * NC) break;
  }
  stats.attempts2 = attempt;
  stats.draws = rng.total - before;
  return stats;
}

function photonArrays(list) {
  const n = list.length;
  const rec = new Float64Array(n * 11), tag = new Int32Array(n * 3);
  list.forEach((p, i) => {
    rec.set([p.origin[0], p.origin[1], p.origin[2], p.dir[0], p.dir[1], p.dir[2], p.power,
      p.colour.r, p.colour.g, p.colour.b, (p.colour.a === undefined) ? NaN : p.colour.a], i * 11);
    tag.set([p.isCaustic ? 1 : 0, p.__phase, p.__attempt], i * 3);
  });
  return { rec: { data: rec, shape: [n, 11] }, tag: { data: tag, shape: [n, 3] } };
}

function treeArrays(tree) {                            // nodes[pos] -> (input index, dim), pos = 1..N
  const n = tree.nodes.length - 1;
  const perm = new Int32Array(n), dim = new Uint8Array(n);
  for (let pos = 1; pos <= n; pos++) { perm[pos - 1] = tree.nodes[pos].photon.__idx; dim[pos - 1] = tree.nodes[pos].dim; }
  return { perm, dim };
}

function genC1(withImage, withScaled) {
  const ND = 10000, NC = 10000;
  const { L, scene, px } = buildScene(10, { numDiffuse: ND, numCaustic: NC });
  const ex = L.ex;
  let t0 = Date.now();
  const stats = emitKeyed(L, scene, ND, NC);
  console.log(`c1 emission: ${JSON.stringify(stats)} in ${Date.now() - t0} ms; lists ${scene.photonListDiffuse.length}/${scene.photonListCaustic.length}`);
  const dArr = photonArrays(scene.photonListDiffuse), cArr = photonArrays(scene.photonListCaustic);
  writeBundle(path.join(OUT, 'photons_c1'), {
    diffuse: dArr.rec, diffuse_tag: dArr.tag, caustic: cArr.rec, caustic_tag: cArr.tag,
  }, { seed: SEED, ND, NC, stats, preset: 10,
    note: 'rows: origin3 dir3 power colour_rgba; tag rows: isCaustic phase attempt. Lists are in store order, BEFORE any kd-tree build (kdtree.js:102 sorts in place).' });

  // kd-trees on fp32-rounded positions (the device photon record holds fp32 positions)
  const rounded = (list) => list.map((p, i) => ({ origin: p.origin.map(Math.fround), __idx: i }));
  const fm = new Int32Array(4096);
  const kdOut = {};
  const trees = {};
  for (const [name, list] of [['diffuse', scene.photonListDiffuse], ['caustic', scene.photonListCaustic]]) {
    const tree = new ex.KdTree(rounded(list));
    trees[name] = tree;
    const ta = treeArrays(tree);
    kdOut[`${name}_perm`] = ta.perm; kdOut[`${name}_dim`] = ta.dim;
    if (tree.nodes.length !== list.length + 1) throw new Error('tree not hole-free');
    for (let n = 1; n <= 4096; n++) fm[n - 1] = tree.findMedian(n);
  }
  kdOut.find_median = fm;
  writeBundle(path.join(OUT, 'kd_c1'), kdOut, { note: 'perm[pos-1] = index into the (fp32-rounded) input list of nodes[pos].photon; find_median[n-1] = findMedian(n)' });

  // queries: fp32-rounded first-hit points of camera rays
  const NQ = 1024;
  const q = new Float64Array(NQ * 3);
  for (let i = 0; i < NQ; i++) {
    L.rng.setStream(PH.DOMAIN_AUX, 4000, i);
    const ray = primaryRay(L, scene, Math.floor(L.rng.next() * W), Math.floor(L.rng.next() * H));
    const [t, s] = closestScan(scene, ray, true);
    const p = (s >= 0) ? [ray.origin[0] + t * ray.dir[0], ray.origin[1] + t * ray.dir[1], ray.origin[2] + t * ray.dir[2]] : [0, 0, 0];
    for (let j = 0; j < 3; j++) q[i * 3 + j] = Math.fround(p[j]);
  }
  const qOut = { queries: { data: q, shape: [NQ, 3] } };
  for (const [name, r] of [['diffuse', 0.40], ['caustic', 0.15], ['diffuse_wide', 1.0]]) {
    const tree = trees[name.split('_')[0]];
    const offs = new Int32Array(NQ + 1), idx = [];
    for (let i = 0; i < NQ; i++) {
      const res = tree.nearestNeighbours([q[i * 3], q[i * 3 + 1], q[i * 3 + 2]], r);
      for (const p of res) idx.push(p.__idx);
      offs[i + 1] = idx.length;
    }
    qOut[`${name}_offsets`] = offs; qOut[`${name}_indices`] = Int32Array.from(idx);
    console.log(`radius ${name} r=${r}: ${idx.length} hits over ${NQ} queries`);
  }
  // k-NN brute force (the reference has no k-NN; order = (d2, input index))
  const K = 64;
  for (const [name, list] of [['diffuse', scene.photonListDiffuse], ['caustic', scene.photonListCaustic]]) {
    const pos = list.map((p) => p.origin.map(Math.fround));
    const kidx = new Int32Array(NQ * K), kd2 = new Float64Array(NQ * K);
    const d2 = new Float64Array(pos.length), order = new Int32Array(pos.length);
    for (let i = 0; i < NQ; i++) {
      const x = [q[i * 3], q[i * 3 + 1], q[i * 3 + 2]];
      for (let j = 0; j < pos.length; j++) {
        const dx = x[0] - pos[j][0], dy = x[1] - pos[j][1], dz = x[2] - pos[j][2];
        d2[j] = dx * dx + dy * dy + dz * dz; order[j] = j;
      }
      const sorted = Array.from(order).sort((a, b) => (d2[a] - d2[b]) || (a - b));
      for (let k = 0; k < K; k++) { kidx[i * K + k] = sorted[k]; kd2[i * K + k] = d2[sorted[k]]; }
    }
    qOut[`knn_${name}_idx`] = { data: kidx, shape: [NQ, K] }; qOut[`knn_${name}_d2`] = { data: kd2, shape: [NQ, K] };
  }
  writeBundle(path.join(OUT, 'queries_c1'), qOut, { K, radii: { diffuse: 0.40, caustic: 0.15, diffuse_wide: 1.0 },
    note: 'radius results are photon indices in the reference pre-order (kdtree.js:64-80) as CSR; knn = brute force ordered by (d2, index)' });

  if (!withImage && !withScaled) return;
  // image: the scene's own trees (built on the unrounded lists, as the reference does)
  scene.photonKdTDiffuse = new ex.KdTree(scene.photonListDiffuse);
  scene.photonKdTCaustic = new ex.KdTree(scene.photonListCaustic);
  if (withScaled) {
    // the progressive preview levels: traceOnCanvas(x, y, scale) traces ONE jittered sample for the block of
    // SUPER_SAMPLE_BASE ** scale pixels at (x, y) (rayphoton.js:1589-1596, :1911); stream (RENDER, pixel of the block's corner, 0)
    const outS = {}, metaS = {};
    for (const scale of [1, 2]) {
      const sblk = ex.SUPER_SAMPLE_BASE ** scale, nbx = Math.ceil(W / sblk), nby = Math.ceil(H / sblk);
      const postS = new Float64Array(nby * nbx * 4), drawsS = new Int32Array(nby * nbx);
      for (let by = 0; by < nby; by++) for (let bx = 0; bx < nbx; bx++) {
        const x = bx * sblk, y = by * sblk, pi = y * W + x;
        L.rng.setStream(PH.DOMAIN_RENDER, pi, 0);
        scene.traceOnCanvas(x, y, scale);
        drawsS[by * nbx + bx] = L.rng.draw;
        for (let c = 0; c < 4; c++) postS[(by * nbx + bx) * 4 + c] = px[pi * 4 + c];
      }
      outS[`post_block${sblk}`] = { data: postS, shape: [nby, nbx, 4] };
      outS[`draws_block${sblk}`] = { data: drawsS, shape: [nby, nbx] };
      metaS[`block${sblk}`] = [nby, nbx];
      console.log(`scaled preview, block ${sblk}: ${nbx}x${nby} samples`);
    }
    writeBundle(path.join(OUT, 'image_c1_scaled'), outS, { seed: SEED, W, H, ND, NC, preset: 10, blocks: metaS,
      note: 'what putPixel received from traceOnCanvas(x, y, scale) for every block corner (x, y) of the C1 frame, scale 1 and 2' });
  }
  if (!withImage) return;
  const pre = new Float64Array(W * H * 4), post = new Float64Array(W * H * 4), draws = new Int32Array(W * H);
  t0 = Date.now();
  for (let y = 0; y < H; y++) {
    for (let x = 0; x < W; x++) {
      const pi = y * W + x;
      L.rng.setStream(PH.DOMAIN_RENDER, pi, 0);
      scene.traceOnCanvas(x, y, 0);
      draws[pi] = L.rng.draw;
      L.rng.setStream(PH.DOMAIN_RENDER, pi, 0);
      const ray = primaryRay(L, scene, x, y);
      const c = scene.traceRay(ray, ex.MAX_TRACE_DIST, 0, 1, [ex.MAT_AIR]);
      pre.set([c.r, c.g, c.b, (c.a === undefined) ? NaN : c.a], pi * 4);
    }
    if (y % 40 === 0) console.log(`  image row ${y} (${Date.now() - t0} ms)`);
  }
  post.set(px);
  writeBundle(path.join(OUT, 'image_c1_modeB'), {
    pre_floor: { data: pre, shape: [H, W, 4] }, post_floor: { data: post, shape: [H, W, 4] },
    draws: { data: draws, shape: [H, W] },
  }, { seed: SEED, W, H, ND, NC, preset: 10,
    note: 'pre_floor = raw traceRay result (rayphoton.js:1626) for stream (RENDER, pixel, 0); post_floor = what putPixel received (rayphoton.js:1618-1623); draws = RNG draws consumed per pixel' });
  console.log(`c1 image done in ${Date.now() - t0} ms`);
}

// ---------------------------------------------------------------- nearestNeighbourAndDistance (kdtree.js:10-62)
function genNearest() {
  const ND = 10000, NC = 10000;
  const { L, scene } = buildScene(10, { numDiffuse: ND, numCaustic: NC });
  emitKeyed(L, scene, ND, NC);                         // the lists of photons_c1 again (same seed, same streams)
  const rounded = (list, n) => list.slice(0, n).map((p, i) => ({ origin: p.origin.map(Math.fround), __idx: i }));
  const trees = { diffuse: ['diffuse', 10000], caustic9999: ['caustic', 9999], two: ['diffuse', 2], one: ['diffuse', 1] };
  // queries: every 7th diffuse photon position, the photons around the middle of the list, and points drawn in the
  // bounding box of the diffuse map (keyed AUX stream 4100)
  const dl = rounded(scene.photonListDiffuse, ND);
  const pts = [];
  for (let i = 0; i < ND; i += 7) pts.push(dl[i].origin);
  const lo = [Infinity, Infinity, Infinity], hi = [-Infinity, -Infinity, -Infinity];
  for (const p of dl) for (let j = 0; j < 3; j++) { lo[j] = Math.min(lo[j], p.origin[j]); hi[j] = Math.max(hi[j], p.origin[j]); }
  for (let i = 0; i < 2048; i++) {
    L.rng.setStream(PH.DOMAIN_AUX, 4100, i);
    pts.push([0, 1, 2].map((j) => Math.fround(lo[j] + (hi[j] - lo[j]) * (1.2 * L.rng.next() - 0.1))));
  }
  const out = {}, undef = {};
  for (const [name, [which, n]] of Object.entries(trees)) {
    const list = rounded(which === 'diffuse' ? scene.photonListDiffuse : scene.photonListCaustic, n);
    const owner = new Map(list.map((p) => [p.origin, p.__idx]));
    const tree = new L.ex.KdTree(list);
    // every photon of the tree is a query too for the tree's own positions (reaches the node with one child)
    const qs = pts.concat(list.filter((_, i) => i % 3 === 0 || Math.abs(i - n / 2) < 4).map((p) => p.origin));
    const q = new Float64Array(qs.length * 3), idx = new Int32Array(qs.length), dist = new Float64Array(qs.length);
    let nu = 0;
    qs.forEach((x, i) => {
      q.set(x, i * 3);
      const [point, d] = tree.nearestNeighbourAndDistance(x);
      if (point === undefined) { idx[i] = -1; dist[i] = NaN; nu++; } else { idx[i] = owner.get(point); dist[i] = d; }
    });
    // ... and each tree position itself, which is how the node with a single child gets visited
    const own = new Float64Array(n * 3), oidx = new Int32Array(n), odist = new Float64Array(n);
    for (let pos = 1; pos <= n; pos++) {
      const x = tree.nodes[pos].photon.origin;
      own.set(x, (pos - 1) * 3);
      const [point, d] = tree.nearestNeighbourAndDistance(x);
      if (point === undefined) { oidx[pos - 1] = -1; odist[pos - 1] = NaN; nu++; } else { oidx[pos - 1] = owner.get(point); odist[pos - 1] = d; }
    }
    out[`${name}_queries`] = { data: q, shape: [qs.length, 3] }; out[`${name}_idx`] = idx; out[`${name}_dist`] = dist;
    out[`${name}_own_queries`] = { data: own, shape: [n, 3] }; out[`${name}_own_idx`] = oidx; out[`${name}_own_dist`] = odist;
    undef[name] = nu;
    console.log(`nearest ${name}: ${qs.length} + ${n} queries, ${nu} undefined`);
  }
  writeBundle(path.join(OUT, 'nearest_c1'), out, { trees, undefined_results: undef,
    note: 'nearestNeighbourAndDistance(x) (kdtree.js:10-62) on trees over the first n fp32-rounded photons of the lists of photons_c1: idx = input index of the returned point (-1 = [undefined, undefined]), dist = the returned distance (NaN = undefined); own_* = the tree nodes positions themselves as queries, in node order' });
}

PH.selfTest();
fs.mkdirSync(OUT, { recursive: true });
const want = new Set(process.argv.slice(2));
const all = want.size === 0;
if (all || want.has('scenes')) genScenes();
if (want.has('presets')) genPresets();
if (all || want.has('intersect')) genIntersect();
if (all || want.has('closest')) genClosest();
if (all || want.has('samples')) genSamples();
if (all || want.has('nearest')) genNearest();
if (all || want.has('c1') || want.has('c1image') || want.has('c1scaled')) genC1(all || want.has('c1image'), all || want.has('c1scaled'));
