'use strict';
// CPU check (no GPU, no addon): this host's presets, built with this host's constructors, flatten to
// exactly the records the reference's scene objects flatten to (tests/golden/scene_preset*_flat.json).
const fs = require('fs');
const path = require('path');
const S = require('../scene');
const presets = require('../presets');

const golden = path.resolve(__dirname, '../../../tests/golden');
let failures = 0;
for (const id of ['10', '7', '4', '2', '0']) {
  const want = JSON.parse(fs.readFileSync(path.join(golden, `scene_preset${id}_flat.json`), 'utf8'));
  const scene = { shapes: [], lights: [], canvasWidth: want.width, canvasHeight: want.height };
  presets[id](scene, S);
  let total = 0;                                            // initLights (rayphoton.js:1313-1324)
  for (const l of scene.lights) { const s = scene.shapes[l]; s.intensity = s.wattage * s.area; total += s.intensity; }
  for (const l of scene.lights) { const s = scene.shapes[l]; s.prob = s.intensity / total; }
  const f = S.flatten(scene);
  const same = (a, b) => (a === b) || (Number.isNaN(a) && (b === null || Number.isNaN(b)));
  const bad = [];
  if (f.nShapes !== want.shapes.length) bad.push(`shape count ${f.nShapes} vs ${want.shapes.length}`);
  want.shapes.forEach((w, i) => {
    const m = f.shapeMeta.subarray(i * 4, i * 4 + 4);
    if (m[0] !== w.typeId || m[1] !== w.material || m[2] !== w.flags || m[3] !== w.proc) bad.push(`shape ${i} meta ${Array.from(m)} vs ${[w.typeId, w.material, w.flags, w.proc]}`);
    for (let k = 0; k < 12; k++) if (!same(f.shapeGeom[i * 12 + k], w.g[k])) bad.push(`shape ${i} (${w.type}) g[${k}] ${f.shapeGeom[i * 12 + k]} vs ${w.g[k]}`);
    for (let k = 0; k < 4; k++) if (!same(f.shapeColour[i * 4 + k], w.colour[k])) bad.push(`shape ${i} colour[${k}]`);
  });
  want.lights.forEach((w, i) => {
    const d = f.lightData.subarray(i * 8, i * 8 + 8);
    const exp = [w.wattage, w.area, w.prob, ...w.colour];
    if (f.lightShape[i] !== w.shape) bad.push(`light ${i} shape`);
    exp.forEach((v, k) => { if (!same(d[k], v)) bad.push(`light ${i} field ${k}: ${d[k]} vs ${v}`); });
  });
  const cam = [...want.camera.origin, ...want.camera.u, ...want.camera.v, ...want.camera.w, want.camera.fovScaleWidth, want.camera.fovScaleHeight];
  cam.forEach((v, k) => { if (!same(f.camera[k], v)) bad.push(`camera[${k}] ${f.camera[k]} vs ${v}`); });
  const mats = [...want.materials.transparent, ...want.materials.refrIndex, ...want.materials.reflectance, ...want.materials.specular];
  mats.forEach((v, k) => { if (!same(f.materials[k], v)) bad.push(`materials[${k}]`); });
  console.log(`preset ${id}: ${bad.length ? 'MISMATCH ' + bad.slice(0, 5).join('; ') : 'ok'} (${f.nShapes} shapes, ${f.nLights} lights)`);
  failures += bad.length;
}
process.exit(failures ? 1 : 0);
