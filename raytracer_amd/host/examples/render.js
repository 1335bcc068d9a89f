'use strict';
// Headless render of a preset (or of a flat scene file) to a PPM image -- what index.html does in the browser,
// with the canvas replaced by a file.
//   node raytracer_amd/host/examples/render.js [--preset 10 | --flat scene.json [--whitted]] [--size 1200x700]
//        [--spp 2] [--photons 20000,50000] [--seed 1] [--out frame.ppm]
const rt = require('..');

const arg = (name, dflt) => { const i = process.argv.indexOf('--' + name); return i > 0 ? process.argv[i + 1] : dflt; };
const [W, H] = arg('size', '1200x700').split('x').map(Number);        // index.html:13
const [nd, nc] = arg('photons', '20000,50000').split(',').map(Number);
const options = { subSample: Number(arg('spp', '1')), seed: Number(arg('seed', '12345')), numPhotonsDiffuse: nd, numPhotonsCaustic: nc };
const ctx = { width: W, height: H };
let scene;
if (arg('flat')) {
  scene = process.argv.includes('--whitted') ? new rt.WhittedScene(ctx, arg('flat'), options) : new rt.FlatFileScene(ctx, arg('flat'), options);
  if (!process.argv.includes('--whitted')) scene.emitManyPhotons();
} else {
  scene = new rt.Scene(ctx, options);
  scene.loadPreset(arg('preset', '10'));
}
const t0 = Date.now();
const out = scene.render(0, 0, W, H);
console.log(`${W}x${H}: ${out.stats.rays} rays, ${out.stats.shadowRays} shadow rays, ${out.stats.gathers} gathers in ${Date.now() - t0} ms`);
rt.writePPM(arg('out', 'frame.ppm'), out.rgba8, W, H);
