'use strict';
// TEST INFRASTRUCTURE (oracle harness).  Turns the reference's live `scene` object graph
// (rayphoton.js:335-481 primitives, :316-328 light fields, :656-713 camera) into the flat
// record layout the C ABI takes (include/rt355.h).  Geometry slot assignment g[0..11]:
//   plane       p=g0-2 n=g3-5
//   sphere      c=g0-2                r=g6
//   hemisphere  c=g0-2 n=g3-5         r=g6 truncMin=g7 truncMax=g8
//   cylinder    c=g0-2 axis=g3-5      r=g6 halfHeight=g7
//   triangle    A=g0-2 AB=g3-5 AC=g6-8 n=g9-11      (square: same)
//   disc        c=g0-2 n=g3-5         r=g6
//   annulus     c=g0-2 n=g3-5         outer=g6 inner=g7

const TYPE_ID = { plane: 0, sphere: 1, hemisphere: 2, cylinder: 3, triangle: 4, square: 5, disc: 6, annulus: 7 };
const FLAG_CONVEX = 1, FLAG_LIGHT = 2, FLAG_TRUNC_MAX = 4, FLAG_HALF_HEIGHT = 8;

// procedural colour ids (rayphoton.js:746-755, :756-768, :784-792/:820-829, :871-874/:889-892)
const PROC_BY_PRESET = { 0: { 0: 1, 8: 2 }, 1: { 0: 3 }, 2: { 0: 3 }, 3: { 0: 4 }, 4: { 0: 4 } };

function num(x) { return (x === undefined || Number.isNaN(x)) ? null : x; }

function flattenShape(s, index, procMap) {
  const g = new Array(12).fill(0);
  let flags = 0;
  const put = (o, v) => { g[o] = v[0]; g[o + 1] = v[1]; g[o + 2] = v[2]; };
  switch (s.type) {
    case 'plane': put(0, s.origin); put(3, s.normalDir); break;
    case 'sphere': put(0, s.centre); g[6] = s.radius; break;
    case 'hemisphere':
      put(0, s.centre); put(3, s.normalDir); g[6] = s.radius; g[7] = s.truncateMin;
      if (s.truncateMax !== undefined) { g[8] = s.truncateMax; flags |= FLAG_TRUNC_MAX; }
      break;
    case 'cylinder':
      put(0, s.centre); put(3, s.axis); g[6] = s.radius;
      if (s.halfHeight !== undefined) { g[7] = s.halfHeight; flags |= FLAG_HALF_HEIGHT; }
      break;
    case 'triangle': case 'square':
      put(0, s.vtxA); put(3, s.edgeAB); put(6, s.edgeAC); put(9, s.normalDir); break;
    case 'disc': put(0, s.centre); put(3, s.normalDir); g[6] = s.radius; break;
    case 'annulus': put(0, s.centre); put(3, s.normalDir); g[6] = s.outerRadius; g[7] = s.innerRadius; break;
    default: throw new Error(`unknown shape type ${s.type}`);
  }
  if (s.convex) flags |= FLAG_CONVEX;
  if (s.isLight) flags |= FLAG_LIGHT;
  const c = s.baseColour;
  return {
    type: s.type, typeId: TYPE_ID[s.type],
    material: (s._material === undefined) ? 255 : s._material,
    flags, proc: (procMap && procMap[index]) || 0, g,
    colour: [c.r, c.g, c.b, num(c.a)],
  };
}

function flattenScene(scene, ex, preset) {
  const procMap = PROC_BY_PRESET[preset];
  const shapes = scene.shapes.map((s, i) => flattenShape(s, i, procMap));
  const lights = scene.lights.map((l) => {
    const s = scene.shapes[l], c = s.lightColour;
    return { shape: l, wattage: s.wattage, area: s.area, intensity: s.intensity, prob: s.prob,
      colour: [c.r, c.g, c.b, num(c.a)] };
  });
  const cam = scene.camera;
  const arr = (t) => Array.from(t);
  return {
    preset: String(preset), width: scene.canvasWidth, height: scene.canvasHeight,
    shapes, lights,
    camera: { origin: cam.origin.slice(0, 3), u: cam.ONBu, v: cam.ONBv, w: cam.ONBw,
      fovScaleWidth: cam.fovScaleWidth, fovScaleHeight: cam.fovScaleHeight },
    materials: {
      transparent: arr(ex.matTransparent).map((b) => (b ? 1 : 0)),
      refrIndex: Array.from({ length: 8 }, (_, i) => num(ex.matRefrIndex[i])),
      reflectance: Array.from({ length: 8 }, (_, i) => num(ex.matReflectance[i])),
      specular: Array.from({ length: 8 }, (_, i) => num(ex.matSpecular[i])),
    },
  };
}

module.exports = { flattenScene, flattenShape, TYPE_ID, FLAG_CONVEX, FLAG_LIGHT, FLAG_TRUNC_MAX, FLAG_HALF_HEIGHT };
