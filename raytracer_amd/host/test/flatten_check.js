'use strict';
// CPU check (no GPU, no addon): every preset of the reference -- the thirteen scenes of rayphoton.js and the four of
// ray.js -- built with this host's constructors flattens to exactly the records the reference's own scene objects
// flatten to (tests/golden/scene_preset*_flat.json, scene_rayjs*_flat.json; the presets that draw random numbers as the
// instance of stream (seed 12345, AUX, 1000 + id), 1099 for the named ones).
const fs = require('fs');
const path = require('path');
const S = require('../scene');
const W = require('../whitted');
const RNG = require('../rng');
const presets = require('../presets');

const golden = path.resolve(__dirname, '../../../tests/golden');
const same = (a, b) => (a === b) || ((a === undefined || Number.isNaN(a)) && (b === null || b === undefined || Number.isNaN(b)));
let failures = 0;

function compare(label, f, want, surface) {
  const bad = [];
  if (f.nShapes !== want.shapes.length) bad.push(`shape count ${f.nShapes} vs ${want.shapes.length}`);
  want.shapes.forEach((w, i) => {
    if (i >= f.nShapes) return;
    const m = f.shapeMeta.subarray(i * 4, i * 4 + 4);
    if (m[0] !== w.typeId || m[1] !== w.material || m[2] !== w.flags || m[3] !== w.proc) bad.push(`shape ${i} meta ${Array.from(m)} vs ${[w.typeId, w.material, w.flags, w.proc]}`);
    for (let k = 0; k < 12; k++) if (!same(f.shapeGeom[i * 12 + k], w.g[k])) bad.push(`shape ${i} (${w.type}) g[${k}] ${f.shapeGeom[i * 12 + k]} vs ${w.g[k]}`);
    for (let k = 0; k < 4; k++) if (!same(f.shapeColour[i * 4 + k], w.colour[k])) bad.push(`shape ${i} colour[${k}] ${f.shapeColour[i * 4 + k]} vs ${w.colour[k]}`);
    if (surface && (!f.shapeSurface || !same(f.shapeSurface[2 * i], w.reflectance) || !same(f.shapeSurface[2 * i + 1], w.specular))) bad.push(`shape ${i} surface`);
  });
  if (f.nLights !== want.lights.length) bad.push(`light count ${f.nLights} vs ${want.lights.length}`);
  want.lights.forEach((w, i) => {
    if (i >= f.nLights) return;
    const d = f.lightData.subarray(i * 8, i * 8 + 8);
    const exp = [w.wattage, w.area, w.prob, ...w.colour];
    if (f.lightShape[i] !== w.shape) bad.push(`light ${i} shape`);
    exp.forEach((v, k) => { if (!same(d[k], v)) bad.push(`light ${i} field ${k}: ${d[k]} vs ${v}`); });
  });
  const cam = [...want.camera.origin.slice(0, 3), ...want.camera.u, ...want.camera.v, ...want.camera.w, want.camera.fovScaleWidth, want.camera.fovScaleHeight];
  cam.forEach((v, k) => { if (!same(f.camera[k], v)) bad.push(`camera[${k}] ${f.camera[k]} vs ${v}`); });
  const mats = [...want.materials.transparent, ...want.materials.refrIndex, ...want.materials.reflectance, ...want.materials.specular];
  mats.forEach((v, k) => { if (!same(f.materials[k], typeof v === 'boolean' ? Number(v) : v)) bad.push(`materials[${k}] ${f.materials[k]} vs ${v}`); });
  console.log(`${label}: ${bad.length ? 'MISMATCH ' + bad.slice(0, 5).join('; ') : 'ok'} (${f.nShapes} shapes, ${f.nLights} lights)`);
  failures += bad.length;
}

for (const id of ['0', '1', '2', '3', '4', '5', '6', '7', '8', '9', '10', 'rmt', 'giacometti']) {
  const want = JSON.parse(fs.readFileSync(path.join(golden, `scene_preset${id}_flat.json`), 'utf8'));
  const stream = new RNG.Stream(12345, RNG.DOMAIN.AUX, 1000 + (/^\d+$/.test(id) ? Number(id) : 99), 0);
  const scene = { shapes: [], lights: [], canvasWidth: want.width, canvasHeight: want.height, random: () => stream.next() };
  presets[id](scene, S);
  let total = 0;                                            // initLights (rayphoton.js:1313-1324)
  for (const l of scene.lights) { const s = scene.shapes[l]; s.intensity = s.wattage * s.area; total += s.intensity; }
  for (const l of scene.lights) { const s = scene.shapes[l]; s.prob = s.intensity / total; }
  compare(`preset ${id}`, S.flatten(scene), want, false);
}
for (const id of ['0', '1', '2', '3']) {
  const want = JSON.parse(fs.readFileSync(path.join(golden, `scene_rayjs${id}_flat.json`), 'utf8'));
  const scene = { shapes: [], lights: [], lamps: [], canvasWidth: want.width, canvasHeight: want.height, materials: W.MATERIALS };
  W.presets[id](scene);
  compare(`ray.js preset ${id}`, S.flatten(scene), want, true);
}
process.exit(failures ? 1 : 0);
