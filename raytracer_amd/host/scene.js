'use strict';
// Host scene layer: the scene / primitive / material vocabulary callers of the reference
// write against (constructor names and positional signatures of rayphoton.js:184-481,
// :656-713), re-implemented over plain records, plus flatten() which lays a scene out as the
// typed arrays the C ABI takes (include/rt355.h, slot table in raytracer_amd/scene.py).
//
// Composites push primitives into a shapes[] list and shrink by LITTLE_SPACE exactly like the
// reference's (Box :184-210, Prism :212-231, Cuboctahedron :233-270, Ball :272-276, Can
// :278-284, Tube :286-293, Halfball :295-305, Bowl :307-314, Spotlight :316-328), so a scene
// description written for the reference builds the same geometry here.

const V = require('./vec');

const LITTLE_SPACE = 1e-3;
const DEG = Math.PI / 180;

const MAT = { AIR: 0, GLASS: 1, WATER: 2, MIRROR: 3, COPPER: 4, LINOLEUM: 5, PLASTER: 6, SPECTRALON: 7 };
// material tables (rayphoton.js:64-100); undefined stays undefined and crosses the ABI as NaN
const MATERIALS = {
  transparent: [true, true, true, false, false, false, false, false],
  refrIndex: [1.0, 1.5, 1.33, undefined, 0.64, undefined, undefined, undefined],
  reflectance: [undefined, undefined, undefined, 1, 0.7, 0.6, 0.2, 1],
  specular: [undefined, undefined, undefined, 1, 0.5, 0.2, 0, 0],
};

const rgba = (r, g, b, a) => ({ r, g, b, a });
const COL = {
  BLACK: rgba(0, 0, 0, 0), WHITE: rgba(255, 255, 255, 1), DARK_GREY: rgba(64, 64, 64, 1),
  VERY_DARK_GREY: rgba(16, 16, 16, 1), GREY: rgba(128, 128, 128, 1), SILVER: rgba(192, 192, 192, 1),
  FIRE_ENGINE_RED: rgba(200, 10, 10, 1), AMETHYST: rgba(150, 100, 200, 1), DEEP_BLUE: rgba(8, 8, 64, 1),
  SKY_BLUE: rgba(128, 128, 224, 1), WARM_GREY: rgba(144, 128, 128, 1),
  ORANGE_ORANGE: { r: 224, g: 124, b: 32 }, GRAPEFRUIT_YELLOW: { r: 248, g: 210, b: 112 },   // no alpha, as in rayphoton.js:39-41
  SCHOOL_BUS_YELLOW: { r: 255, g: 216, b: 1 },
  LIME_GREEN: rgba(112, 160, 1, 1), ROBINS_EGG_BLUE: rgba(1, 180, 180, 1), DEEP_PINK: rgba(255, 32, 144, 1),
  RAW_UMBER: rgba(112, 68, 17, 1), ENGLISH_WALNUT: rgba(68, 48, 40, 1), VANILLA: rgba(243, 229, 171, 1),
  CHOCOLATE: rgba(32, 16, 8, 1),
};

const TYPE_ID = { plane: 0, sphere: 1, hemisphere: 2, cylinder: 3, triangle: 4, square: 5, disc: 6, annulus: 7 };
const FLAG = { CONVEX: 1, LIGHT: 2, TRUNC_MAX: 4, HALF_HEIGHT: 8, FIRST_ROOT: 16 };
const PROC = { NONE: 0, STRIPE_CHECKER: 1, EIGHT_BALL: 2, TILE_GRID: 3, ROTATED_CHECKER: 4, STRIPE_CHECKER3: 5 };

// ---------------------------------------------------------------- primitives
class Shape {
  constructor(type, defaults, baseColour, material) {
    this.type = type;
    this.baseColour = baseColour || defaults;
    this.procColour = PROC.NONE;      // closures cannot cross to the GPU: presets name one of the PROC patterns instead
    this.material = material;
  }
  set material(m) {                   // rayphoton.js:339-346
    this._material = m;
    this.transparent = MATERIALS.transparent[m];
    this.refrIndex = MATERIALS.refrIndex[m];
    this.reflectance = MATERIALS.reflectance[m];
    this.specular = MATERIALS.specular[m];
    this.isLight = false;
  }
  get material() { return this._material; }
}
class Plane extends Shape {
  constructor(origin, normalDir, baseColour, material) {
    super('plane', COL.DEEP_BLUE, baseColour, material);
    this.origin = origin; this.normalDir = V.normalise(normalDir);
  }
}
class Sphere extends Shape {
  constructor(centre, radius, baseColour, material) {
    super('sphere', COL.FIRE_ENGINE_RED, baseColour, material);
    this.centre = centre; this.radius = radius;
  }
}
class Hemisphere extends Shape {      // normalDir points to the half that exists
  constructor(centre, radius, normalDir, truncateMin, truncateMax, convex, baseColour, material) {
    super('hemisphere', COL.FIRE_ENGINE_RED, baseColour, material);
    Object.assign(this, { centre, radius, normalDir: V.normalise(normalDir), truncateMin, truncateMax, convex });
  }
}
class Cylinder extends Shape {
  constructor(centre, axis, radius, halfHeight, convex, baseColour, material) {
    super('cylinder', COL.WHITE, baseColour, material);
    Object.assign(this, { centre, axis: V.normalise(axis), radius, halfHeight, convex });
  }
}
class Triangle extends Shape {
  constructor(vtxA, edgeAB, edgeAC, baseColour, material) {
    super('triangle', COL.LIME_GREEN, baseColour, material);
    Object.assign(this, { vtxA, edgeAB, edgeAC, normalDir: V.normalise(V.cross(edgeAB, edgeAC)) });
  }
}
class Square extends Shape {          // a parallelogram
  constructor(vtxA, edgeAB, edgeAC, baseColour, material) {
    super('square', COL.DEEP_PINK, baseColour, material);
    Object.assign(this, { vtxA, edgeAB, edgeAC, normalDir: V.normalise(V.cross(edgeAB, edgeAC)) });
  }
}
class Disc extends Shape {
  constructor(centre, radius, normalDir, baseColour, material) {
    super('disc', COL.DEEP_PINK, baseColour, material);
    Object.assign(this, { centre, radius, normalDir: V.normalise(normalDir) });
  }
}
class Annulus extends Shape {
  constructor(centre, outerRadius, innerRadius, normalDir, baseColour, material) {
    super('annulus', COL.DEEP_PINK, baseColour, material);
    Object.assign(this, { centre, outerRadius, innerRadius, normalDir: V.normalise(normalDir) });
  }
}

// ---------------------------------------------------------------- composites
// shared by Box / Prism / Cuboctahedron: corner moved inwards along the three edges
function insetCorner(vtx, edges, k) {
  return edges.reduce((p, e) => V.add(p, V.scale(k, V.normalise(e))), vtx);
}
const shorten = (e, k) => V.sub(e, V.scale(2 * k, V.normalise(e)));

class Box {                            // parallelepiped: six Squares
  constructor(shapes, vtxA, edgeAB, edgeAC, edgeAD, baseColour, material, nudge) {
    const k = ((nudge === undefined || nudge) ? 1 : 0) * LITTLE_SPACE;
    const a = insetCorner(vtxA, [edgeAB, edgeAC, edgeAD], k);
    const ab = shorten(edgeAB, k), ac = shorten(edgeAC, k), ad = shorten(edgeAD, k);
    const opp = V.add(V.add(V.add(a, ab), ac), ad);
    const ba = V.neg(ab), ca = V.neg(ac), da = V.neg(ad);
    for (const [v, e1, e2] of [[a, ac, ab], [a, ad, ac], [a, ab, ad], [opp, ba, ca], [opp, ca, da], [opp, da, ba]]) {
      shapes.push(new Square(v, e1, e2, baseColour, material));
    }
  }
}
class Prism {                          // triangle ABC extruded along AD
  constructor(shapes, vtxA, edgeAB, edgeAC, edgeAD, baseColour, material) {
    const a = insetCorner(vtxA, [edgeAB, edgeAC, edgeAD], LITTLE_SPACE);
    const ab = shorten(edgeAB, LITTLE_SPACE), ac = shorten(edgeAC, LITTLE_SPACE), ad = shorten(edgeAD, LITTLE_SPACE);
    const opp = V.add(V.add(a, ac), ad);
    const ca = V.neg(ac), cb = V.add(ca, ab), da = V.neg(ad);
    shapes.push(new Triangle(a, ac, ab, baseColour, material));
    shapes.push(new Square(a, ad, ac, baseColour, material));
    shapes.push(new Square(a, ab, ad, baseColour, material));
    shapes.push(new Triangle(opp, ca, cb, baseColour, material));
    shapes.push(new Square(opp, cb, da, baseColour, material));
  }
}
class Cuboctahedron {                  // cube with its eight corners cut: six Squares, eight Triangles
  constructor(shapes, choppedVtxA, edgeAB, edgeAC, edgeAD, baseColourSquare, baseColourTriangle, material) {
    const a = insetCorner(choppedVtxA, [edgeAB, edgeAC, edgeAD], LITTLE_SPACE);
    const half = (e) => V.scale(0.5 - LITTLE_SPACE / V.len(e), e);
    const full = (e) => V.scale(1 - 2 * LITTLE_SPACE / V.len(e), e);
    const ds = V.add(a, half(edgeAB)), dw = V.add(a, half(edgeAC));
    const dn = V.add(ds, full(edgeAC)), de = V.add(dw, full(edgeAB));
    const sw = V.add(a, half(edgeAD));
    const nw = V.add(sw, full(edgeAC)), ne = V.add(nw, full(edgeAB)), se = V.add(sw, full(edgeAB));
    const un = V.add(dn, full(edgeAD)), ue = V.add(de, full(edgeAD)), us = V.add(ds, full(edgeAD)), uw = V.add(dw, full(edgeAD));
    const face = (Cls, p, q, r, col) => shapes.push(new Cls(p, V.sub(q, p), V.sub(r, p), col, material));
    for (const [p, q, r] of [[ds, dw, de], [ne, dn, un], [se, de, ue], [sw, ds, us], [nw, dw, uw], [us, ue, uw]]) face(Square, p, q, r, baseColourSquare);
    for (const [p, q, r] of [[dn, ne, de], [de, se, ds], [ds, sw, dw], [dw, nw, dn], [un, ue, ne], [ue, us, se], [us, uw, sw], [uw, un, nw]]) face(Triangle, p, q, r, baseColourTriangle);
  }
}
class Ball { constructor(shapes, centre, radius, baseColour, material) { shapes.push(new Sphere(centre, radius - LITTLE_SPACE, baseColour, material)); } }
class Can {
  constructor(shapes, centre, axis, radius, halfHeight, baseColourAround, baseColourCaps, material) {
    const r = radius - LITTLE_SPACE, h = halfHeight - LITTLE_SPACE, ax = V.normalise(axis);
    shapes.push(new Cylinder(centre, axis, r, h, true, baseColourAround, material));
    shapes.push(new Disc(V.add(centre, V.scale(h, ax)), r, axis, baseColourCaps, material));
    shapes.push(new Disc(V.add(centre, V.scale(-h, ax)), r, V.neg(axis), baseColourCaps, material));
  }
}
class Tube {
  constructor(shapes, centre, axis, outerRadius, innerRadius, halfHeight, baseColourAround, baseColourCaps, material) {
    const ro = outerRadius - LITTLE_SPACE, ri = innerRadius + LITTLE_SPACE, h = halfHeight - LITTLE_SPACE, ax = V.normalise(axis);
    shapes.push(new Cylinder(centre, axis, ro, h, true, baseColourAround, material));
    shapes.push(new Cylinder(centre, axis, ri, h, false, baseColourAround, material));
    shapes.push(new Annulus(V.add(centre, V.scale(h, ax)), ro, ri, axis, baseColourCaps, material));
    shapes.push(new Annulus(V.add(centre, V.scale(-h, ax)), ro, ri, V.neg(axis), baseColourCaps, material));
  }
}
class Halfball {                       // normalDir points away from the dome (disc on top)
  constructor(shapes, centre, radius, normalDir, truncateMin, truncateMax, baseColour, material) {
    const tMin = (truncateMin === undefined) ? LITTLE_SPACE : truncateMin + LITTLE_SPACE;
    const tMax = (truncateMax === undefined) ? undefined : truncateMax - LITTLE_SPACE;
    const r = radius - LITTLE_SPACE, n = V.normalise(normalDir);
    shapes.push(new Hemisphere(centre, r, V.neg(normalDir), tMin, tMax, true, baseColour, material));
    shapes.push(new Disc(V.add(centre, V.scale(-1 * tMin, n)), Math.sqrt(r ** 2 - tMin ** 2), normalDir, baseColour, material));
    if (truncateMax < radius) {
      shapes.push(new Disc(V.add(centre, V.scale(-1 * tMax, n)), Math.sqrt(r ** 2 - tMax ** 2), V.neg(normalDir), baseColour, material));
    }
  }
}
class Bowl {                           // normalDir points towards the rim
  constructor(shapes, centre, outerRadius, innerRadius, normalDir, baseColour, material) {
    const ro = outerRadius - LITTLE_SPACE, ri = innerRadius + LITTLE_SPACE;
    shapes.push(new Hemisphere(centre, ro, V.neg(normalDir), 0, undefined, true, baseColour, material));
    shapes.push(new Hemisphere(centre, ri, V.neg(normalDir), 0, undefined, false, baseColour, material));
    shapes.push(new Annulus(centre, ro, ri, normalDir, baseColour, material));
  }
}
class Spotlight {                      // copper shade (a Bowl) around a light-emitting Disc
  constructor(shapes, lights, centre, radius, dir, wattage, colour) {
    new Bowl(shapes, centre, 1.3 * radius, 1.125 * radius, dir, COL.RAW_UMBER, MAT.COPPER);
    const lamp = new Disc(V.add(centre, V.scale(-0.5 * radius, V.normalise(dir))), radius, dir);
    Object.assign(lamp, { isLight: true, wattage, lightColour: colour || COL.WHITE, area: Math.PI * radius ** 2 });
    lights.push(shapes.push(lamp) - 1);
  }
}

// ---------------------------------------------------------------- camera (rayphoton.js:671-713)
class Camera {
  constructor(origin, gazeTheta, gazePhi, width, height, fieldOfView) {
    Object.assign(this, { origin, width, height });
    this._gazeTheta = gazeTheta; this._gazePhi = gazePhi;
    this.findOrthonormalBasis();
    this.fieldOfView = fieldOfView || 60;
  }
  set gazeTheta(t) { this._gazeTheta = t; this.findOrthonormalBasis(); }
  set gazePhi(p) { this._gazePhi = p; this.findOrthonormalBasis(); }
  set fieldOfView(fov) {
    this._fieldOfView = fov;
    this.fovRadians = (fov / 2) * DEG;
    this.fovScaleWidth = Math.tan(this.fovRadians);
    this.fovScaleHeight = this.fovScaleWidth * this.height / this.width;
  }
  findOrthonormalBasis() {
    const th = (this._gazeTheta || 0) * DEG, ph = (this._gazePhi || 0) * DEG;
    this.gazeDir = V.normalise([Math.cos(th) * Math.cos(ph), Math.sin(th) * Math.cos(ph), Math.sin(ph)]);
    this.ONBw = V.neg(this.gazeDir);
    this.ONBu = V.normalise(V.cross([0, 0, 1], this.ONBw));
    this.ONBv = V.cross(this.ONBw, this.ONBu);
  }
  toUVW(xyz) { const t = V.sub(xyz, this.origin); return [V.dot(t, this.ONBu), V.dot(t, this.ONBv), V.dot(t, this.ONBw)]; }
  toXYZ(uvw) { return V.add(V.add(V.add(this.origin, V.scale(uvw[0], this.ONBu)), V.scale(uvw[1], this.ONBv)), V.scale(uvw[2], this.ONBw)); }
}

// ---------------------------------------------------------------- flatten
function flattenShape(s) {
  const g = new Array(12).fill(0);
  let flags = 0;
  const put = (o, v) => { g[o] = v[0]; g[o + 1] = v[1]; g[o + 2] = v[2]; };
  switch (s.type) {
    case 'plane': put(0, s.origin); put(3, s.normalDir); break;
    case 'sphere': put(0, s.centre); g[6] = s.radius; break;
    case 'hemisphere':
      put(0, s.centre); put(3, s.normalDir); g[6] = s.radius; g[7] = s.truncateMin;
      if (s.truncateMax !== undefined) { g[8] = s.truncateMax; flags |= FLAG.TRUNC_MAX; }
      break;
    case 'cylinder':
      put(0, s.centre); put(3, s.axis); g[6] = s.radius;
      if (s.halfHeight !== undefined) { g[7] = s.halfHeight; flags |= FLAG.HALF_HEIGHT; }
      break;
    case 'triangle': case 'square': put(0, s.vtxA); put(3, s.edgeAB); put(6, s.edgeAC); put(9, s.normalDir); break;
    case 'disc': put(0, s.centre); put(3, s.normalDir); g[6] = s.radius; break;
    case 'annulus': put(0, s.centre); put(3, s.normalDir); g[6] = s.outerRadius; g[7] = s.innerRadius; break;
    default: throw new Error(`I didn't recognize the shape! (${s.type})`);     // rayphoton.js:640
  }
  if (s.convex) flags |= FLAG.CONVEX;
  if (s.isLight) flags |= FLAG.LIGHT;
  if (s.firstRootOnly) flags |= FLAG.FIRST_ROOT;              // the older tracer's cylinder (ray.js:443-461)
  return { typeId: TYPE_ID[s.type], material: (s._material === undefined) ? 255 : s._material, flags, proc: s.procColour || 0, g };
}

// scene -> the typed arrays of rt_scene_desc.  `undefined` numbers become NaN.
function flatten(scene) {
  const n = scene.shapes.length, nl = scene.lights.length;
  const nan = (x) => (x === undefined ? NaN : x);
  const out = {
    nShapes: n, shapeMeta: new Uint8Array(n * 4), shapeGeom: new Float64Array(n * 12), shapeColour: new Float64Array(n * 4),
    nLights: nl, lightShape: new Uint32Array(nl), lightData: new Float64Array(nl * 8),
    camera: new Float64Array(16), width: scene.canvasWidth, height: scene.canvasHeight, materials: new Float64Array(32),
  };
  scene.shapes.forEach((s, i) => {
    const f = flattenShape(s);
    out.shapeMeta.set([f.typeId, f.material, f.flags, f.proc], i * 4);
    out.shapeGeom.set(f.g, i * 12);
    const c = s.baseColour;
    out.shapeColour.set([c.r, c.g, c.b, nan(c.a)], i * 4);
  });
  scene.lights.forEach((l, i) => {
    const s = scene.shapes[l], c = s.lightColour;
    out.lightShape[i] = l;
    out.lightData.set([s.wattage, s.area, nan(s.prob), c.r, c.g, c.b, nan(c.a), 0], i * 8);
  });
  const cam = scene.camera;
  out.camera.set([...cam.origin.slice(0, 3), ...cam.ONBu, ...cam.ONBv, ...cam.ONBw, cam.fovScaleWidth, cam.fovScaleHeight]);
  const mats = scene.materials || MATERIALS;
  ['transparent', 'refrIndex', 'reflectance', 'specular'].forEach((k, r) => {
    for (let m = 0; m < 8; m++) { const v = mats[k][m]; out.materials[r * 8 + m] = (typeof v === 'boolean') ? (v ? 1 : 0) : nan(v === null ? undefined : v); }
  });
  // the older tracer keeps reflectance / specular on the shape (ray.js:244-385)
  if (scene.shapes.some((s) => s.reflectance !== undefined || s.specular !== undefined)) {
    out.shapeSurface = new Float64Array(n * 2);
    scene.shapes.forEach((s, i) => { out.shapeSurface[2 * i] = nan(s.reflectance); out.shapeSurface[2 * i + 1] = nan(s.specular); });
  }
  return out;
}

// The flat scene files (tests/golden/scene_*_flat.json; the same layout FlatScene.from_flat reads in Python) -> the
// typed arrays of rt_scene_desc.  null stands for `undefined`.
function flattenFromJSON(d, width, height) {
  const n = d.shapes.length, nl = d.lights.length;
  const nan = (x) => (x === undefined || x === null ? NaN : x);
  const W = width || d.width, H = height || d.height;
  const out = {
    nShapes: n, shapeMeta: new Uint8Array(n * 4), shapeGeom: new Float64Array(n * 12), shapeColour: new Float64Array(n * 4),
    nLights: nl, lightShape: new Uint32Array(nl), lightData: new Float64Array(nl * 8),
    camera: new Float64Array(16), width: W, height: H, materials: new Float64Array(32),
  };
  d.shapes.forEach((s, i) => {
    out.shapeMeta.set([s.typeId, s.material, s.flags, s.proc || 0], i * 4);
    out.shapeGeom.set(s.g, i * 12);
    out.shapeColour.set(s.colour.map(nan), i * 4);
  });
  d.lights.forEach((l, i) => {
    out.lightShape[i] = l.shape;
    out.lightData.set([l.wattage, l.area, nan(l.prob), ...l.colour.map(nan), 0], i * 8);
  });
  const c = d.camera;
  // fovScaleHeight follows the canvas (rayphoton.js:694)
  out.camera.set([...c.origin.slice(0, 3), ...c.u, ...c.v, ...c.w, c.fovScaleWidth, (W === d.width && H === d.height) ? c.fovScaleHeight : c.fovScaleWidth * H / W]);
  const mats = d.materials || MATERIALS;
  ['transparent', 'refrIndex', 'reflectance', 'specular'].forEach((k, r) => {
    for (let m = 0; m < 8; m++) { const v = mats[k][m]; out.materials[r * 8 + m] = (typeof v === 'boolean') ? (v ? 1 : 0) : nan(v); }
  });
  if (n && d.shapes[0].reflectance !== undefined) {
    out.shapeSurface = new Float64Array(n * 2);
    d.shapes.forEach((s, i) => { out.shapeSurface[2 * i] = nan(s.reflectance); out.shapeSurface[2 * i + 1] = nan(s.specular); });
  }
  return out;
}

module.exports = {
  LITTLE_SPACE, MAT, MATERIALS, COL, TYPE_ID, FLAG, PROC,
  Shape, Plane, Sphere, Hemisphere, Cylinder, Triangle, Square, Disc, Annulus,
  Box, Prism, Cuboctahedron, Ball, Can, Tube, Halfball, Bowl, Spotlight, Camera,
  flatten, flattenShape, flattenFromJSON,
};
