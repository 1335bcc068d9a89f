'use strict';
// GPU check of the Node host: the reference's default scene driven through Scene / KdTree / Ray
// exactly as a caller of the reference would, compared with the fixtures the reference produced
// (tests/golden).  Prints one line per check; exit code 0 iff all pass.
const rt = require('..');
const { loadBundle } = require('./bundle');

let failures = 0;
function check(name, ok, detail) { console.log(`${ok ? 'PASS' : 'FAIL'} ${name}${detail ? ' -- ' + detail : ''}`); if (!ok) failures++; }
const same = (a, b) => a === b || (Number.isNaN(a) && Number.isNaN(b));

const W = 320, H = 240;
const filled = new Float64Array(W * H * 4).fill(NaN);
const ctx = { width: W, height: H, fillStyle: '', fillRect(x, y) {
  const m = /rgba\(([^,]+),([^,]+),([^,]+),([^)]+)\)/.exec(this.fillStyle);
  for (let c = 0; c < 4; c++) filled[(y * W + x) * 4 + c] = Number(m[c + 1]);
} };
const log = console.log; console.log = () => {};         // the tree-build line of kdtree.js:3
const scene = new rt.Scene(ctx, { numPhotonsDiffuse: 10000, numPhotonsCaustic: 10000, subSample: 1, seed: 12345, precision: rt.PRECISION.F64 });
scene.loadPreset(10);
console.log = log;

// emission: attempts and stored lists
const g = loadBundle('photons_c1');
check('emission attempts', scene.emitStats.attemptsPhase1 === g.meta.stats.attempts1 && scene.emitStats.attemptsPhase2 === g.meta.stats.attempts2,
  `${scene.emitStats.attemptsPhase1}/${scene.emitStats.attemptsPhase2}`);
let worst = 0;
for (let i = 0; i < g.diffuse.length; i++) if (!Number.isNaN(g.diffuse[i])) worst = Math.max(worst, Math.abs(scene._recD[i] - g.diffuse[i]));
for (let i = 0; i < g.caustic.length; i++) if (!Number.isNaN(g.caustic[i])) worst = Math.max(worst, Math.abs(scene._recC[i] - g.caustic[i]));
check('photon lists', scene._recD.length === g.diffuse.length && worst < 1e-12, `max abs diff ${worst}`);
check('photon objects', scene.photonListDiffuse.length === 10000 && scene.photonListCaustic[0].isCaustic === true && scene.photonListDiffuse[0].origin.length === 3);

// kd-tree: nodes and radius query order
const kd = loadBundle('kd_c1');
const nodes = scene.photonKdTDiffuse.nodes;
let nodeOk = nodes.length === 10001;
for (let pos = 1; pos <= 10000 && nodeOk; pos++) nodeOk = nodes[pos].dim === kd.diffuse_dim[pos - 1] && nodes[pos].photon === scene.photonKdTDiffuse.photonAt(kd.diffuse_perm[pos - 1]);
check('KdTree.nodes', nodeOk);
const q = loadBundle('queries_c1');
let radiusOk = true;
for (let i = 0; i < 64 && radiusOk; i++) {
  const x = [q.queries[i * 3], q.queries[i * 3 + 1], q.queries[i * 3 + 2]];
  const got = scene.photonKdTDiffuse.nearestNeighbours(x, 1.0);
  const lo = q.diffuse_wide_offsets[i], hi = q.diffuse_wide_offsets[i + 1];
  radiusOk = got.length === hi - lo;
  for (let k = 0; k < got.length && radiusOk; k++) radiusOk = got[k] === scene.photonKdTDiffuse.photonAt(q.diffuse_wide_indices[lo + k]);
}
check('KdTree.nearestNeighbours order', radiusOk);
{
  // adaptive-radius gather: the radius used holds at least 10 photons (or is the last attempt), one step less does not
  const tree = scene.photonKdTDiffuse, nq = 32;
  const hits = new Float32Array(nq * 9);
  for (let i = 0; i < nq; i++) {
    for (let c = 0; c < 3; c++) hits[i * 9 + c] = q.queries[i * 3 + c];
    hits[i * 9 + 5] = 1; hits[i * 9 + 6] = 255; hits[i * 9 + 7] = 255; hits[i * 9 + 8] = 255;
  }
  const r0 = 0.02, ad = tree.gatherAdaptive(hits, r0);
  let adOk = ad.rgb.length === nq * 3 && ad.radius.length === nq;
  for (let i = 0; i < nq && adOk; i++) {
    const x = [hits[i * 9], hits[i * 9 + 1], hits[i * 9 + 2]];
    const steps = Math.round(Math.log(ad.radius[i] / r0) / Math.log(3));
    const radii = [r0];
    for (let k = 1; k <= steps; k++) radii.push(radii[k - 1] * 3);           // the library's own sequence of radii
    const r = radii[steps];
    const inside = tree.nearestNeighbours(x, r).length;
    adOk = Math.abs(ad.radius[i] - r) <= 1e-5 * r && (inside >= 10 || steps === 9);     // (10 attempts: steps 0 .. 9)
    if (adOk && steps > 0) adOk = tree.nearestNeighbours(x, radii[steps - 1]).length < 10;
  }
  check('KdTree.gatherAdaptive radius', adOk);
}
let fmOk = true;
for (let n = 1; n <= 4096 && fmOk; n++) fmOk = scene.photonKdTDiffuse.findMedian(n) === kd.find_median[n - 1];
check('KdTree.findMedian', fmOk);
// nearestNeighbourAndDistance: the reference's one-path answer (never closer than the exact one; undefined for a
// query that leaves a two-photon tree through the missing upper child)
{
  const tree = scene.photonKdTDiffuse;
  let ok = true, same = 0, worstD = 0, closer = 0;
  for (let i = 0; i < 64; i++) {
    const x = scene.photonListDiffuse[i * 7].origin.map((v) => v + 0.01 * i);
    const [p, d] = tree.nearestNeighbourAndDistance(x), [pe, de] = tree.nearestExact(x);
    ok = ok && p !== undefined;
    if (p === undefined) continue;
    if (d < de) closer++;
    worstD = Math.max(worstD, Math.abs(Math.hypot(...[0, 1, 2].map((j) => Math.fround(p[j]) - Math.fround(x[j]))) - d));   // (fp32 positions and queries)
    if (p === pe) same++;
  }
  check('KdTree.nearestNeighbourAndDistance', ok && closer === 0 && worstD < 1e-12 && same > 8, `defined ${ok}, closer than exact ${closer}, |d - hypot| ${worstD}, same as exact ${same}/64`);
  const two = new rt.KdTree([{ origin: [0, 0, 0], dir: [0, 0, 1], power: 1, colour: { r: 1, g: 1, b: 1 } },
    { origin: [1, 0, 0], dir: [0, 0, 1], power: 1, colour: { r: 1, g: 1, b: 1 } }]);
  const below = two.nearestNeighbourAndDistance([0.2, 0, 0]), above = two.nearestNeighbourAndDistance([5, 0, 0]);
  check('KdTree.nearestNeighbourAndDistance undefined', below[0] !== undefined && above[0] === undefined && above[1] === undefined);
}
let threw = false;
try { new rt.KdTree([]); } catch (e) { threw = /no data/.test(e.message); }
check('KdTree([]) throws', threw);

// Ray.intersectDist on single pairs
const ib = loadBundle('intersect');
const names = ['plane', 'sphere', 'hemisphere', 'cylinder', 'triangle', 'square', 'disc', 'annulus'];
let isectOk = true;
for (let i = 0; i < ib.t.length && isectOk; i += 97) {
  const gg = ib.shape_g.subarray(i * 12, i * 12 + 12), type = names[ib.shape_meta[i * 2]], flags = ib.shape_meta[i * 2 + 1];
  const shape = { type, convex: !!(flags & 1), _material: 6 };
  const v = (o) => [gg[o], gg[o + 1], gg[o + 2]];
  if (type === 'plane') Object.assign(shape, { origin: v(0), normalDir: v(3) });
  else if (type === 'sphere') Object.assign(shape, { centre: v(0), radius: gg[6] });
  else if (type === 'hemisphere') Object.assign(shape, { centre: v(0), normalDir: v(3), radius: gg[6], truncateMin: gg[7], truncateMax: (flags & 4) ? gg[8] : undefined });
  else if (type === 'cylinder') Object.assign(shape, { centre: v(0), axis: v(3), radius: gg[6], halfHeight: (flags & 8) ? gg[7] : undefined });
  else if (type === 'triangle' || type === 'square') Object.assign(shape, { vtxA: v(0), edgeAB: v(3), edgeAC: v(6), normalDir: v(9) });
  else if (type === 'disc') Object.assign(shape, { centre: v(0), normalDir: v(3), radius: gg[6] });
  else Object.assign(shape, { centre: v(0), normalDir: v(3), outerRadius: gg[6], innerRadius: gg[7] });
  const ray = new rt.Ray([ib.ray_o[i * 3], ib.ray_o[i * 3 + 1], ib.ray_o[i * 3 + 2]], [ib.ray_d_raw[i * 3], ib.ray_d_raw[i * 3 + 1], ib.ray_d_raw[i * 3 + 2]]);
  const t = ray.intersectDist(shape);
  isectOk = same(t === undefined ? NaN : t, ib.t[i]);
}
check('Ray.intersectDist', isectOk);

// image: whole frame in one launch, then the tile API with fillRect
const img = loadBundle('image_c1_modeB');
const out = scene.render(0, 0, W, H);
let bad = 0;
for (let p = 0; p < W * H; p++) {
  for (let c = 0; c < 3; c++) {
    const a = out.pre[p * 4 + c], b = img.pre_floor[p * 4 + c];
    if (Math.abs(a - b) / Math.max(1, Math.abs(b)) > 1e-9) { bad++; break; }
  }
}
check('Scene.render radiance', bad <= 8, `${bad} of ${W * H} pixels differ, ${out.stats.rays} rays`);
scene.traceTile([96, 64], 64, 0);
scene.traceOnCanvas(5, 7, 0);
let tileBad = 0;
for (let y = 64; y < 128; y++) for (let x = 96; x < 160; x++) for (let c = 0; c < 4; c++) if (!same(filled[(y * W + x) * 4 + c], img.post_floor[(y * W + x) * 4 + c])) { tileBad++; break; }
for (let c = 0; c < 4; c++) if (!same(filled[(7 * W + 5) * 4 + c], img.post_floor[(7 * W + 5) * 4 + c])) tileBad++;
check('Scene.traceTile / traceOnCanvas -> fillRect', tileBad <= 2, `${tileBad} pixels differ`);

// preview levels and the progressive order: every pixel gets covered, coarse tiles before the tiles they contain
{
  const gs = loadBundle('image_c1_scaled');
  filled.fill(NaN);
  scene.traceTile([0, 0], 9 * 36, 2);                       // the whole frame as 9x9 blocks
  let sbad = 0;
  for (let by = 0; by < 27; by++) for (let bx = 0; bx < 36; bx++) {
    for (let c = 0; c < 4; c++) if (!same(filled[((by * 9) * W + bx * 9) * 4 + c], gs.post_block9[(by * 36 + bx) * 4 + c])) { sbad++; break; }
  }
  check('Scene.traceTile(scale 2) -> fillRect', sbad <= 1, `${sbad} blocks differ`);
  const order = scene.progressiveTiles();
  const seen = new Uint8Array(W * H); let orderOk = true;
  for (const t of order) {
    for (let y = t.region[1]; y < Math.min(H, t.region[1] + t.size); y++) for (let x = t.region[0]; x < Math.min(W, t.region[0] + t.size); x++) {
      if (t.scale < seen[y * W + x] - 1 && false) orderOk = false;
      if (seen[y * W + x] && seen[y * W + x] - 1 < t.scale) orderOk = false;   // a coarser tile after a finer one over the same pixel
      seen[y * W + x] = t.scale + 1;
    }
  }
  check('Scene.progressiveTiles order', orderOk && seen.every((v) => v === 1), `${order.length} tiles`);
}

// the older tracer (ray.js) from its flat scene file, fp64: every pixel of the reference's frame
{
  const path = require('path');
  const ga = loadBundle('image_rayjs3');
  const wa = ga.meta.W, ha = ga.meta.H;
  const sa = new rt.WhittedScene({ width: wa, height: ha }, path.join(__dirname, '..', '..', '..', 'tests', 'golden', 'scene_rayjs3_flat.json'),
    { seed: ga.meta.seed, precision: rt.PRECISION.F64 });
  const oa = sa.render(0, 0, wa, ha);
  let badA = 0;
  for (let i = 0; i < wa * ha * 4; i++) if (!same(oa.pre[i], ga.rgba[i])) badA++;
  check('WhittedScene.render (ray.js preset 3)', badA === 0, `${badA} values differ, ${oa.stats.rays} rays`);
  // the same scene built with the older tracer's constructors instead of read from the file
  const sb = new rt.WhittedScene({ width: wa, height: ha }, null, { seed: ga.meta.seed, precision: rt.PRECISION.F64 });
  sb.loadPreset(3);
  const ob = sb.render(0, 0, wa, ha);
  let badB = 0;
  for (let i = 0; i < wa * ha * 4; i++) if (!same(ob.pre[i], ga.rgba[i])) badB++;
  check('WhittedScene.loadPreset(3) by constructors', badB === 0, `${badB} values differ`);
  // a preset of rayphoton.js that is built from random numbers renders (the scene is the fixture's: flatten_check.js)
  const s9 = new rt.Scene({ width: 64, height: 48 }, { numPhotonsDiffuse: 500, numPhotonsCaustic: 500, subSample: 1, seed: 12345 });
  const quiet = console.log; console.log = () => {};
  s9.loadPreset(9);
  console.log = quiet;
  const o9 = s9.render(0, 0, 64, 48);
  check("Scene.loadPreset(9) (random hues) renders", s9.shapes.length === 43 && o9.stats.rays >= 64 * 48 && o9.rgba8.some((v) => v > 0), `${o9.stats.rays} rays`);
  const os = require('os'), fs = require('fs');
  const ppm = path.join(os.tmpdir(), 'rt355_check.ppm');
  rt.writePPM(ppm, oa.rgba8, wa, ha);
  check('writePPM', fs.statSync(ppm).size === 15 + wa * ha * 3, `${fs.statSync(ppm).size} bytes`);
  fs.unlinkSync(ppm);
}

// the single-item methods: traceRay on a pixel's camera ray continues that pixel's stream and returns the frame's colour;
// createPhoton + emitPhoton on an attempt's stream store what the bulk emission stored for that attempt
{
  const RNG = require('../rng'), V = require('../vec');
  const whole = scene.render(0, 0, W, H);
  const cam = scene.camera;
  let badRay = 0;
  for (const [X, Y] of [[0, 0], [160, 120], [77, 201], [319, 239], [200, 60]]) {
    const s = new RNG.Stream(12345, RNG.DOMAIN.RENDER, Y * W + X, 0);
    let u = ((X + (0 + s.next()) / 1) * 2 / W) - 1, v = -(((Y + (0 + s.next()) / 1) * 2 / H) - 1);
    u *= cam.fovScaleWidth; v *= cam.fovScaleHeight;
    const xyz = cam.toXYZ([u, v, -1]);
    const c = scene.traceRay(new rt.Ray(cam.origin, V.sub(xyz, cam.origin)), scene.params.maxTraceDist, 0, 1, [rt.MAT.AIR], { id: Y * W + X, draw: 2 });
    const o = (Y * W + X) * 4;
    if (!same(c.r, whole.pre[o]) || !same(c.g, whole.pre[o + 1]) || !same(c.b, whole.pre[o + 2]) || !same(c.a === undefined ? NaN : c.a, whole.pre[o + 3])) badRay++;
  }
  check('Scene.traceRay continues a pixel of the frame', badRay === 0, `${badRay} of 5 differ`);
  const black = scene.traceRay(new rt.Ray(cam.origin, [0, 1, 0]), 20, 0, 0.001, [rt.MAT.AIR]);
  check('Scene.traceRay importance cut', black.r === 0 && black.a === 0);
  const dl = scene.directLight([0, 0, 0.01], [0, 0, 1], { r: 255, g: 255, b: 255 }, { id: 777, draw: 0 });
  check('Scene.directLight', Number.isInteger(dl.r) && dl.a === 1 && scene.lastDraw === 5 * scene.lights.length, `rgb ${dl.r},${dl.g},${dl.b}`);
  // attempts 0 .. 39 of phase 1, one createPhoton + emitPhoton each (the lists are full: make room for what they store)
  scene.params.numPhotonsDiffuse += 8; scene.params.numPhotonsCaustic += 8;
  let okWalk = true, storedN = 0;
  for (let a = 0; a < 40; a++) {
    const ph = scene.createPhoton(undefined, { id: a, phase: 1 });
    okWalk = okWalk && ph.stream.draw === 5 && ph.isCaustic === undefined;
    if (scene.emitPhoton(ph, 0, false, [rt.MAT.AIR])) storedN++;
  }
  scene.params.numPhotonsDiffuse -= 8; scene.params.numPhotonsCaustic -= 8;
  check('Scene.createPhoton / emitPhoton / storePhoton', okWalk && storedN > 0 && storedN <= 40, `${storedN} of 40 attempts stored a photon`);
  const lamp1 = scene.shapes[scene.lights[1]];
  const fromLamp = scene.createPhoton(1);
  check('Scene.createPhoton(source)', V.len(V.sub(fromLamp.origin, lamp1.centre)) <= lamp1.radius + 1e-12 && fromLamp.power === lamp1.wattage);
}

// handles: wrong kinds and inconsistent flat scenes are rejected before the C ABI sees them; after shutdown(ctx) the
// scenes and maps made from the context are dead (a call throws, a late finaliser frees nothing twice)
{
  const path = require('path');
  const native = require(path.join(__dirname, '..', 'napi', 'rt355.node'));
  const throws = (fn, what) => { try { fn(); return false; } catch (e) { return what ? what.test(String(e && e.name) + ' ' + String(e && e.message)) : true; } };
  const ctx2 = native.init(0);
  const sc2 = new rt.Scene({ width: 64, height: 48 }, { numPhotonsDiffuse: 100, numPhotonsCaustic: 100 });
  require('../presets')['10'](sc2, rt);
  sc2.initLights();
  const flat = sc2.flatten();
  const dev2 = native.sceneUpload(ctx2, flat);
  const rec = new Float64Array(11 * 64); for (let i = 0; i < 64; i++) { rec[i * 11] = i; rec[i * 11 + 5] = 1; rec[i * 11 + 6] = 1; }
  const map2 = native.photonMapBuild(ctx2, rec);
  check('handle kinds are checked', throws(() => native.sceneUpload(dev2, flat), /TypeError/) && throws(() => native.photonMapNodes(dev2), /TypeError/) &&
        throws(() => native.traceTile(ctx2, {}, null, null, 0, 0, 8, 8), /TypeError/) && throws(() => native.photonMapBuild({}, rec), /TypeError/), '');
  const short = Object.assign({}, flat, { shapeGeom: flat.shapeGeom.subarray(0, flat.shapeGeom.length - 12) });
  check('sceneUpload checks array lengths', throws(() => native.sceneUpload(ctx2, short), /RangeError/) &&
        throws(() => native.sceneUpload(ctx2, Object.assign({}, flat, { nLights: flat.nLights + 1 })), /RangeError/), '');
  native.shutdown(ctx2);
  check('handles die with their context', throws(() => native.photonMapNodes(map2), /shut down/) && throws(() => native.traceTile(dev2, {}, null, null, 0, 0, 8, 8), /shut down/) &&
        throws(() => native.sceneUpload(ctx2, flat), /shut down/), '');
  if (global.gc) global.gc();
}

// the throughput path: photon lists that stay on the GPU, trees built from them there, frames queued into device buffers,
// shards and their merge, and the multi-GPU example with two workers on this one GPU
(async () => {
  const path = require('path');
  const native = rt.addon(), ctxN = rt.context();
  const bits = (a) => new Uint32Array(a.buffer, a.byteOffset, a.byteLength / 4);
  const sameBits = (a, b) => { const x = bits(a), y = bits(b); if (x.length !== y.length) return false; for (let i = 0; i < x.length; i++) if (x[i] !== y[i]) return false; return true; };
  {
    // emitManyPhotons left the lists on the GPU; what the host path (emitMany: arrays across the boundary) returns is the same
    const host = native.emitMany(scene._scene(), scene.params);
    check('device-resident lists = host lists', native.listsCounts(scene.photonLists).nDiffuse === 10000 && sameBits(host.diffuse, scene._recD) && sameBits(host.caustic, scene._recC));
    // the 36-byte format: position, direction, power * colour of the same photons
    const packed = native.emitManyLists(scene._scene(), scene.params, rt.SHARD.PACKED);
    const pr = native.listsRecords(packed.lists, 0);
    let worstP = 0;
    for (let i = 0; i < pr.n; i++) {
      const r = scene._recD.subarray(i * 11, i * 11 + 11), q = pr.records.subarray(i * 9, i * 9 + 9);
      const want = [r[0], r[1], r[2], r[3], r[4], r[5], r[6] * r[7], r[6] * r[8], r[6] * r[9]];
      for (let c = 0; c < 9; c++) worstP = Math.max(worstP, Math.abs(q[c] - want[c]) / Math.max(1, Math.abs(want[c])));
    }
    check('packed lists hold the same photons', pr.n === 10000 && pr.words === 9 && worstP < 1e-6, `max rel diff ${worstP}`);
    // trees: from the full list on the GPU (the scene's), from the packed list, from packed photons handed over as an array
    const a = native.photonMapNodes(scene.photonKdTDiffuse.map), b = native.photonMapNodes(native.photonMapBuildLists(packed.lists, 0));
    const c = native.photonMapNodes(native.photonMapBuildPacked(ctxN, pr.records)), d = native.kdBuild(ctxN, Float32Array.from({ length: pr.n * 3 }, (_, i) => pr.records[Math.floor(i / 3) * 9 + i % 3]));
    check('trees from lists / packed photons / positions agree', sameBits(a.perm, b.perm) && sameBits(a.perm, c.perm) && sameBits(a.perm, d.perm) && a.dim.every((v, i) => v === b.dim[i] && v === c.dim[i] && v === d.dim[i]) &&
      native.photonMapSize(scene.photonKdTDiffuse.map) === 10000);
    native.listsFree(packed.lists);
    // shards: two sub-ranges of one batch, merged, give the lists of the loop (first 4096 attempts of phase 1, no cap reached)
    const W1 = 10, lo = new Uint32Array(2048 * W1), hi = new rt.DeviceBuffer(2048 * W1 * 4);
    const s0 = native.emitShard(scene._scene(), scene.params, 1, 0, 2048, rt.SHARD.PACKED, 0, lo), s1 = native.emitShard(scene._scene(), scene.params, 1, 2048, 4096, rt.SHARD.PACKED, 0, hi.handle);
    const m = Math.max(s0.n, s1.n, 1), merged = new Uint32Array(2 * m * W1);
    merged.set(lo.subarray(0, s0.n * W1), 0); merged.set(hi.read(new Uint32Array(s1.n * W1)), m * W1);
    const ls = native.listsCreate(ctxN, scene.params, rt.SHARD.PACKED);
    const info = native.listsApply(ls, 1, 0, merged, Uint32Array.of(s0.n, s1.n), m);
    const all = native.emitPhotons(scene._scene(), scene.params, 1, 0, 4096);
    const nD = all.kind.reduce((k, v) => k + (v === 1), 0);
    const got = native.listsRecords(ls, 0);
    let worstS = 0, j = 0;
    for (let i = 0; i < all.kind.length && j < got.n; i++) {
      if (all.kind[i] !== 1) continue;
      const r = all.records.subarray(i * 11, i * 11 + 11), q = got.records.subarray(j * 9, j * 9 + 9);
      const want = [r[0], r[1], r[2], r[3], r[4], r[5], r[6] * r[7], r[6] * r[8], r[6] * r[9]];
      for (let c = 0; c < 9; c++) worstS = Math.max(worstS, Math.abs(q[c] - want[c]) / Math.max(1, Math.abs(want[c])));
      j++;
    }
    check('emitShard + listsApply = the stored photons of the same attempts', info.nDiffuse === nD && got.n === nD && j === nD && info.nDiffuse + info.nCaustic === s0.n + s1.n && info.phaseEnded === 0 && worstS < 1e-6,
      `${s0.n} + ${s1.n} stored photons, ${nD} diffuse, max rel diff ${worstS}`);
    hi.free(); native.listsFree(ls);
  }
  {
    // frames queued into device buffers (three in flight) are the finished frame
    const done = scene.render(0, 0, W, H);
    const frames = scene.renderFrames(4, null, 3);
    check('queued frames = finished frame', frames.length === 4 && frames.every((f) => sameBits(f, done.rgba8)));
    const pre = new rt.DeviceBuffer(W * H * 32);
    scene.renderQueued(0, 0, W, H, { pre });
    const preHost = pre.read(new Float64Array(W * H * 4));
    check('queued frame into a pre-clamp buffer', sameBits(preHost, done.pre));
    const small = new rt.DeviceBuffer(64);
    let refused = false;
    try { scene.renderQueued(0, 0, W, H, { rgba8: small }); } catch (e) { refused = /smaller than the tile/.test(e.message); }
    check('queued frame into a short buffer is refused', refused);
    pre.free(); small.free();
    native.join(ctxN); native.synchronize(ctxN);
    let unknown = false;
    try { native.setOption(ctxN, 'no_such_switch', 1); } catch (e) { unknown = /unknown option/.test(e.message); }
    native.setOption(ctxN, 'knn_verbose', 0);
    check('context calls', unknown && typeof native.stream(ctxN) === 'bigint' && native.bandRows(100, 16, 2, 1) === 48 && native.bandRows(100, 16, 2, 0) === 52);
  }
  {
    // the gather that allocates nothing, on device buffers, against the ordinary one
    const nq = 512, hits = new Float32Array(nq * 9);
    for (let i = 0; i < nq; i++) {
      const ph = scene._recD.subarray((i * 19) * 11, (i * 19) * 11 + 3);
      hits.set([ph[0], ph[1], ph[2], 0, 0, 1, 200, 180, 160], i * 9);
    }
    const hd = new rt.DeviceBuffer(nq * 36).write(hits), od = new rt.DeviceBuffer(nq * 12);
    const map = scene.photonKdTDiffuse.map;
    let okG = true;
    for (const [r, k] of [[0.4, 0], [0, 50]]) {
      native.gatherRadianceStatic(map, hd.handle, nq, r, k, od.handle);
      okG = okG && sameBits(od.read(new Float32Array(nq * 3)), native.gatherRadiance(map, hits, r, k));
    }
    check('gatherRadianceStatic on device buffers', okG);
    hd.free(); od.free();
  }
  {
    // two workers, each with its own context, on this GPU: lists, trees and frame of the one-GPU job
    const multi = require(path.join(__dirname, '..', 'examples', 'multi_gpu.js'));
    const out = await multi.run({ devices: [0, 0], preset: 10, width: 256, height: 160, batch: 1 << 17, check: true,
      options: { numPhotonsDiffuse: 20000, numPhotonsCaustic: 30000, seed: 7 } });
    check('multi_gpu.js: two workers = one GPU', out.check.ok, JSON.stringify(out.check) + ` rounds ${out.workers[0].rounds}`);
  }
  process.exit(failures ? 1 : 0);
})().catch((e) => { console.log('FAIL uncaught: ' + (e && e.stack || e)); process.exit(1); });
