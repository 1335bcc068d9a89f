'use strict';
// TEST INFRASTRUCTURE (oracle harness) -- fixtures for the older Whitted-style tracer ray.js ("mode A").
//   node oracle/harness/gen_golden_modeA.js
// Loads the unmodified ray.js in its own vm context (its top-level names clash with rayphoton.js),
// builds presets 2 and 3 (loadPreset always throws at the end, ray.js:768, after the scene is
// complete), and renders them with one keyed RNG stream per pixel.  Restated here: only the
// primary-ray set-up of traceOnCanvas (ray.js:816-829) so that the unrounded traceRay result
// can be captured per pixel.

const path = require('path');
const fs = require('fs');
const { loadRayJs } = require('./ref_loader');
const { flattenShape } = require('./flatten_ref');
const { writeBundle } = require('./bundle');
const PH = require('./philox');

const OUT = path.resolve(__dirname, '../../tests/golden');
const SEED = 12345, W = 160, H = 120;
const PROC = { 0: { 0: 1, 8: 2 }, 1: { 0: 5 }, 2: { 0: 3 }, 3: { 0: 4 } };

for (const preset of [2, 3, 0, 1]) {
  const L = loadRayJs({ seed: SEED, subSample: 1 });
  const ex = L.ex;
  const scene = new ex.Scene({ width: W, height: H });
  try { scene.loadPreset(preset); } catch (e) { /* ReferenceError: lights is not defined (ray.js:768) */ }
  const cam = scene.camera;
  const shapes = scene.shapes.map((s, i) => {
    const f = flattenShape(Object.assign(Object.create(Object.getPrototypeOf(s)), s, { _material: s.material }), i, PROC[preset]);
    if (s.type === 'cylinder') f.flags |= 16;            // ray.js cylinders: infinite, first root only (ray.js:443-461)
    f.reflectance = s.reflectance; f.specular = s.specular;
    return f;
  });
  const flat = {
    preset: 'rayjs' + preset, width: W, height: H, shapes, lights: [],
    camera: { origin: cam.origin, u: cam.ONBu, v: cam.ONBv, w: cam.ONBw, fovScaleWidth: cam.fovScaleWidth, fovScaleHeight: cam.fovScaleHeight },
    materials: { transparent: [1, 0, 1, 1, 0, 0, 0, 0], refrIndex: [1.0, null, 1.5, 1.33, null, null, null, null],
      reflectance: [null, null, null, null, null, null, null, null], specular: [null, null, null, null, null, null, null, null] },
    params: { maxTraceDist: ex.MAX_TRACE_DIST, maxDepth: ex.MAX_DEPTH },
  };
  fs.writeFileSync(path.join(OUT, `scene_rayjs${preset}_flat.json`), JSON.stringify(flat));

  const img = new Float64Array(W * H * 4), draws = new Int32Array(W * H);
  let errors = 0;
  const t0 = Date.now();
  for (let y = 0; y < H; y++) for (let x = 0; x < W; x++) {
    const pi = y * W + x;
    L.rng.setStream(PH.DOMAIN_RENDER, pi, 0);
    let u = ((x + (0 + L.rng.next()) / 1) * 2 / W) - 1;
    let v = -(((y + (0 + L.rng.next()) / 1) * 2 / H) - 1);
    u *= cam.fovScaleWidth; v *= cam.fovScaleHeight;
    const xyz = cam.toXYZ([u, v, -1]);
    const ray = new ex.Ray(cam.origin, [xyz[0] - cam.origin[0], xyz[1] - cam.origin[1], xyz[2] - cam.origin[2]]);
    let c;
    try { c = scene.traceRay(ray, ex.MAX_TRACE_DIST, ex.MAX_DEPTH, 1, [ex.MAT_AIR]); } catch (e) { c = undefined; }
    if (c === undefined) { errors++; img.set([NaN, NaN, NaN, NaN], pi * 4); }
    else img.set([c.r, c.g, c.b, c.a === undefined ? NaN : c.a], pi * 4);
    draws[pi] = L.rng.draw;
  }
  writeBundle(path.join(OUT, `image_rayjs${preset}`), { rgba: { data: img, shape: [H, W, 4] }, draws: { data: draws, shape: [H, W] } },
    { seed: SEED, W, H, preset, errors, note: 'raw traceRay result (ray.js:844-993) per pixel, stream (RENDER, pixel, 0); NaN rows = traceRay returned undefined / threw (ray.js:905)' });
  console.log(`ray.js preset ${preset}: ${shapes.length} shapes, ${errors} error pixels, ${Date.now() - t0} ms`);
}
