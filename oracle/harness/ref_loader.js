'use strict';
// TEST INFRASTRUCTURE (oracle harness) -- not part of the shipped product path.
//
// Loads the unmodified reference scripts (rayphoton.js + kdtree.js, or ray.js) from
// RT_REFERENCE_DIR (default /root/reference) into a fresh `vm` context whose Math.random is a
// keyed Philox stream, and returns the classes the scripts declare.  The reference text is
// only read and evaluated here; it is never written anywhere.
//
// The three `const` sizes the parity runs need to change (rayphoton.js:11,16,17) are
// rewritten in the text before evaluation because emitPhoton divides by them
// (rayphoton.js:1551) and storePhoton caps on them (rayphoton.js:1346-1352).

const fs = require('fs');
const path = require('path');
const vm = require('vm');
const { KeyedRng } = require('./philox');

const REF_DIR = process.env.RT_REFERENCE_DIR || '/root/reference';

function replaceConst(src, name, value) {
  const re = new RegExp(`const ${name} = [^;]+;`);
  if (!re.test(src)) throw new Error(`const ${name} not found in reference text`);
  return src.replace(re, `const ${name} = ${value};`);
}

function makeContext(rng, log) {
  const mathCopy = {};
  for (const k of Object.getOwnPropertyNames(Math)) mathCopy[k] = Math[k];
  mathCopy.random = () => rng.next();
  const quiet = { log: (...a) => { if (log) log.push(a.join(' ')); } };
  return vm.createContext({ console: quiet, Math: mathCopy });
}

// opts: { seed, numDiffuse, numCaustic, subSample }
function loadRayphoton(opts) {
  opts = opts || {};
  const rng = new KeyedRng(opts.seed === undefined ? 12345 : opts.seed);
  let src = fs.readFileSync(path.join(REF_DIR, 'rayphoton.js'), 'utf8') + '\n' +
            fs.readFileSync(path.join(REF_DIR, 'kdtree.js'), 'utf8');
  if (opts.numDiffuse !== undefined) src = replaceConst(src, 'NUM_PHOTONS_DIFFUSE', opts.numDiffuse);
  if (opts.numCaustic !== undefined) src = replaceConst(src, 'NUM_PHOTONS_CAUSTIC', opts.numCaustic);
  if (opts.subSample !== undefined) src = replaceConst(src, 'SUB_SAMPLE', opts.subSample);
  src += '\n;({Scene, KdTree, Ray, Photon, Camera, qRoots, vecPerturb, vecOrthonormal, discSample,' +
         ' Plane, Sphere, Hemisphere, Cylinder, Triangle, Square, Disc, Annulus,' +
         ' Box, Prism, Cuboctahedron, Ball, Can, Tube, Halfball, Bowl, Spotlight,' +
         ' matTransparent, matRefrIndex, matReflectance, matSpecular,' +
         ' NUM_PHOTONS_DIFFUSE, NUM_PHOTONS_CAUSTIC, SUB_SAMPLE, MAX_TRACE_DIST, MAX_DEPTH,' +
         ' EPSILON, DIFFUSE_RADIUS, CAUSTIC_RADIUS, COL_WHITE, MAT_AIR, SUPER_SAMPLE_BASE});';
  const log = [];
  const ctx = makeContext(rng, log);
  const ex = vm.runInContext(src, ctx, { filename: 'reference:rayphoton+kdtree' });
  return { ex, rng, log };
}

function loadRayJs(opts) {
  opts = opts || {};
  const rng = new KeyedRng(opts.seed === undefined ? 12345 : opts.seed);
  let src = fs.readFileSync(path.join(REF_DIR, 'ray.js'), 'utf8');
  if (opts.subSample !== undefined) src = replaceConst(src, 'SUB_SAMPLE', opts.subSample);
  src += '\n;({Scene, Ray, Camera, qRoots, vecPerturb, SUB_SAMPLE, MAX_TRACE_DIST, MAX_DEPTH, EPSILON, MAT_AIR});';
  const log = [];
  const ctx = makeContext(rng, log);
  const ex = vm.runInContext(src, ctx, { filename: 'reference:ray.js' });
  return { ex, rng, log };
}

// Canvas stub: records the last fillRect colour per pixel (rayphoton.js:1917-1918).
function stubCtx(width, height) {
  const px = new Float64Array(width * height * 4).fill(NaN);
  const ctx = {
    width, height, fillStyle: '',
    fillRect(x, y, w, h) {
      const m = /rgba\(([^,]+),([^,]+),([^,]+),([^)]+)\)/.exec(this.fillStyle);
      if (!m) return;
      const o = (y * width + x) * 4;
      for (let c = 0; c < 4; c++) px[o + c] = Number(m[c + 1]);
    },
    createImageData(w, h) { return { data: new Uint8ClampedArray(w * h * 4) }; },
    putImageData() {}, beginPath() {}, moveTo() {}, lineTo() {}, stroke() {}, arc() {},
  };
  return { ctx, px };
}

module.exports = { loadRayphoton, loadRayJs, stubCtx, REF_DIR };
