'use strict';
// The older tracer's scene vocabulary (ray.js:12-63, :104-385): the same eight primitives and composites as
// rayphoton.js, but a surface is described on the shape itself -- every constructor ends in
// (baseColour, reflectance, specular, material) -- the materials are AIR, OPAQUE, GLASS, WATER, the colour table is
// its own, and each primitive class has its own default colour and specular share.  The geometry of the composites
// is that of scene.js (the two scripts build them alike); this module only re-dresses the primitives they make.
// Its four presets (ray.js:573-762) follow.

const S = require('./scene');
const V = require('./vec');

const MAT = { AIR: 0, OPAQUE: 1, GLASS: 2, WATER: 3 };
const MATERIALS = {
  transparent: [true, false, true, true, false, false, false, false], refrIndex: [1.0, undefined, 1.5, 1.33, undefined, undefined, undefined, undefined],
  reflectance: new Array(8).fill(undefined), specular: new Array(8).fill(undefined),
};
const rgba = (r, g, b, a) => ({ r, g, b, a });
const COL = {
  BLACK: rgba(0, 0, 0, 0), WHITE: rgba(255, 255, 255, 1), DARK_GREY: rgba(64, 64, 64, 1), VERY_DARK_GREY: rgba(16, 16, 16, 1),
  GREY: rgba(128, 128, 128, 1), SILVER: rgba(192, 192, 192, 1), RED: rgba(192, 0, 0, 1), LIME_GREEN: rgba(112, 160, 0, 1),
  YELLOW: rgba(240, 224, 8, 1), MAUVE: rgba(64, 32, 112, 1), DEEP_BLUE: rgba(8, 8, 64, 1), SKY_BLUE: rgba(128, 128, 224, 1),
  WARM_GREY: rgba(144, 128, 128, 1),
  ORANGE_ORANGE: { r: 224, g: 124, b: 32 }, GRAPEFRUIT_YELLOW: { r: 248, g: 210, b: 112 }, DEEP_PINK: { r: 255, g: 32, b: 144 },
  COPPER: { r: 174, g: 105, b: 56 },                         // (no alpha, ray.js:25-28)
};
// per class: default colour and default specular share (ray.js:244-385); the default reflectance is 0.5 throughout
const CLASS_DEFAULTS = {
  cylinder: [COL.WHITE, 0.8], sphere: [COL.RED, 0.5], hemisphere: [COL.RED, 0.5], plane: [COL.DEEP_BLUE, 0.1],
  triangle: [COL.LIME_GREEN, 0.2], square: [COL.DEEP_PINK, 0.35], disc: [COL.DEEP_PINK, 0.5], annulus: [COL.DEEP_PINK, 0.5],
};

// gives a primitive of scene.js the older tracer's surface: `material` becomes a plain field (assigning it later, as
// the presets do, must not reset reflectance / specular from a table)
function dress(shape, baseColour, reflectance, specular, material) {
  const [colour, spec] = CLASS_DEFAULTS[shape.type];
  Object.defineProperty(shape, 'material', { get() { return this._material; }, set(m) { this._material = m; }, enumerable: true });
  shape.material = (material === undefined || material === null) ? MAT.OPAQUE : material;
  shape.baseColour = baseColour || colour;
  shape.reflectance = (reflectance === undefined || reflectance === null) ? 0.5 : reflectance;
  shape.specular = (specular === undefined || specular === null) ? spec : specular;
  shape.transparent = undefined; shape.refrIndex = undefined;
  return shape;
}

const Plane = function (origin, normalDir, c, r, s, m) { return dress(new S.Plane(origin, normalDir), c, r, s, m); };
const Sphere = function (centre, radius, c, r, s, m) { return dress(new S.Sphere(centre, radius), c, r, s, m); };
const Hemisphere = function (centre, radius, normalDir, tMin, tMax, convex, c, r, s, m) { return dress(new S.Hemisphere(centre, radius, normalDir, tMin, tMax, convex), c, r, s, m); };
const Triangle = function (a, ab, ac, c, r, s, m) { return dress(new S.Triangle(a, ab, ac), c, r, s, m); };
const Square = function (a, ab, ac, c, r, s, m) { return dress(new S.Square(a, ab, ac), c, r, s, m); };
const Disc = function (centre, radius, normalDir, c, r, s, m) { return dress(new S.Disc(centre, radius, normalDir), c, r, s, m); };
const Annulus = function (centre, outer, inner, normalDir, c, r, s, m) { return dress(new S.Annulus(centre, outer, inner, normalDir), c, r, s, m); };
// an unbounded cylinder of which only the first root is tested (ray.js:443-461); `height` is kept but unused there too
const Cylinder = function (centre, axis, height, radius, c, r, s, m) {
  const shape = dress(new S.Cylinder(centre, axis, radius, undefined, true), c, r, s, m);
  shape.height = height; shape.firstRootOnly = true;
  return shape;
};

// composites: built by scene.js, then every primitive they made is dressed (colours chosen per primitive type where a
// composite takes two)
function composite(Ctor, shapes, geometry, colourOf, r, s, m) {
  const made = [];
  new Ctor(made, ...geometry);
  for (const shape of made) shapes.push(dress(shape, colourOf(shape), r, s, m));
}
class Box { constructor(shapes, a, ab, ac, ad, c, r, s, m) { composite(S.Box, shapes, [a, ab, ac, ad], () => c, r, s, m); } }
class Prism { constructor(shapes, a, ab, ac, ad, c, r, s, m) { composite(S.Prism, shapes, [a, ab, ac, ad], () => c, r, s, m); } }
class Cuboctahedron {
  constructor(shapes, a, ab, ac, ad, cSquare, cTriangle, r, s, m) { composite(S.Cuboctahedron, shapes, [a, ab, ac, ad], (p) => (p.type === 'square' ? cSquare : cTriangle), r, s, m); }
}
class Ball { constructor(shapes, centre, radius, c, r, s, m) { composite(S.Ball, shapes, [centre, radius], () => c, r, s, m); } }
class Halfball { constructor(shapes, centre, radius, normalDir, tMin, tMax, c, r, s, m) { composite(S.Halfball, shapes, [centre, radius, normalDir, tMin, tMax], () => c, r, s, m); } }
class Bowl { constructor(shapes, centre, outer, inner, normalDir, c, r, s, m) { composite(S.Bowl, shapes, [centre, outer, inner, normalDir], () => c, r, s, m); } }
// a copper shade; the lamp itself is a Disc that is only listed, never drawn (ray.js:219-226)
class Spotlight {
  constructor(shapes, lamps, centre, radius, dir, wattage) {
    new Bowl(shapes, centre, 1.25 * radius, radius, dir, COL.COPPER, 0.7, 0.7, MAT.OPAQUE);
    lamps.push(Object.assign(Disc(V.add(centre, V.scale(-0.1 * radius, V.normalise(dir))), radius, dir, COL.WHITE, 0.1, 0.7, MAT.OPAQUE), { wattage }));
  }
}

// ray.js's camera looks along a direction with a given up vector (ray.js:538-552); the flattened form is the same
class Camera {
  constructor(origin, gazeDir, up, width, height, fieldOfView) {
    Object.assign(this, { origin, width, height, up });
    this.fieldOfView = fieldOfView || 45;
    this.fovScaleWidth = Math.tan(Math.PI / 180 * (this.fieldOfView / 2));
    this.fovScaleHeight = this.fovScaleWidth * height / width;
    this.ONBw = V.normalise(V.scale(-1, gazeDir));
    this.ONBu = V.normalise(V.cross(up, this.ONBw));
    this.ONBv = V.cross(this.ONBw, this.ONBu);
  }
}

const dome = () => [                       // the open-ended rooms of presets 0 and 1
  Plane([0, 0, 0], [0, 0, 1], COL.WHITE, 0.2), Plane([0, 80, 0], [0, -1, 0], COL.SKY_BLUE), Plane([0, -80, 0], [0, 1, 0], COL.VERY_DARK_GREY),
  Plane([80, 0, 0], [-1, 0, 0], COL.VERY_DARK_GREY), Plane([-80, 0, 0], [1, 0, 0], COL.VERY_DARK_GREY), Plane([0, 0, 80], [1, 0, -1], COL.VERY_DARK_GREY)];
const pen = (half, wall) => [                                // floor and four low-reflectance walls of presets 2 and 3
  Plane([0, 0, 0], [0, 0, 1], COL.GREY, 0.6), Plane([0, half, 0], [0, -1, 0], COL.VERY_DARK_GREY, 0.02), Plane([0, -half, 0], [0, 1, 0], COL.VERY_DARK_GREY, 0.02),
  Plane([half, 0, 0], [-1, 0, 0], wall, 0.02), Plane([-half, 0, 0], [1, 0, 0], wall, 0.02)];

const presets = {
  '0'(scene) {
    const sh = scene.shapes = dome().concat([
      Sphere([0.3, 0, 0.9], 0.4, COL.RED, 0.90), Sphere([-1.1, 1.2, 0.25], 0.25, COL.MAUVE, 0.45),
      Sphere([0.5, 2.5, 1.25], 1.25, COL.BLACK, 0.3), Sphere([-1.5, -2.5, 0.8], 0.8, COL.ORANGE_ORANGE, 0.25)]);
    sh[0].procColour = S.PROC.STRIPE_CHECKER;
    sh[6].material = MAT.GLASS;
    sh[8].procColour = S.PROC.EIGHT_BALL;
    scene.camera = new Camera([-0.3, -4, 1], [0, 1, -0.1], [0, 0, 1], scene.canvasWidth, scene.canvasHeight);
  },
  '1'(scene) {
    const sh = scene.shapes = dome().concat([
      Sphere([0.3, 0, 0.5], 0.45, COL.RED, 0.50), Sphere([-1.1, 1.2, 0.25], 0.25, COL.MAUVE, 0.45),
      Sphere([0.5, 7, 1.25], 1.25, COL.BLACK, 0.3), Sphere([-1.5, -2.5, 0.8], 0.8, COL.ORANGE_ORANGE, 0.25),
      Plane([0, 4.25, 0], [0, -1, 0], COL.WHITE, 0.95), Plane([0, 4.35, 0], [0, 1, 0], COL.WHITE, 0.95)]);     // a pane of glass
    sh[0].procColour = S.PROC.STRIPE_CHECKER3;
    for (const i of [9, 10, 11]) sh[i].material = MAT.GLASS;
    scene.camera = new Camera([-0.3, -6, 3], [0, 1, -0.2], [0, 0, 1], scene.canvasWidth, scene.canvasHeight);
  },
  '2'(scene) {
    const sh = scene.shapes = pen(12, COL.MAUVE);
    sh[0].procColour = S.PROC.TILE_GRID;
    new Spotlight(sh, scene.lamps, [0, -2, 1.5], 1, [0, 0, 1], 40);
    new Spotlight(sh, scene.lamps, [0, 10, 1.5], 1, [0, -1, 0], 40);
    scene.camera = new Camera([-2.8, -9, 2.5], [0.3, 1, -0.12], [0, 0, 1], scene.canvasWidth, scene.canvasHeight);
  },
  '3'(scene) {
    const sh = scene.shapes = pen(16, COL.DARK_GREY);
    // (positional: 0.97 is the reflectance and the water material's number lands in `specular`, ray.js:741)
    new Halfball(sh, [0, -1, 1], 0.9, [0, 0, 1], 0.3, undefined, COL.WHITE, 0.97, MAT.WATER);
    new Bowl(sh, [-2.3, 1, 1], 1, 0.8, [0, 0, 1], COL.DEEP_BLUE, 0.3);
    new Ball(sh, [-2.3, 1, 0.7], 0.5, COL.GRAPEFRUIT_YELLOW, 0.3);
    new Ball(sh, [0.35, -0.8, 2.8], 0.3, COL.ORANGE_ORANGE, 0.3);
    new Ball(sh, [-0.4, 3.5, 2], 2, COL.COPPER, 0.6);
    new Cuboctahedron(sh, [-0.7, -2, 0], [2.5, 0, 0], [0, 2.5, 0], [0, 0, 2.5], COL.DEEP_PINK, COL.DARK_GREY, 0.3);
    sh[0].procColour = S.PROC.ROTATED_CHECKER;
    scene.camera = new Camera([-3.3, -8, 4.5], [0.4, 1, -0.4], [0, 0, 1], scene.canvasWidth, scene.canvasHeight);
  },
};

module.exports = { MAT, MATERIALS, COL, CLASS_DEFAULTS, Plane, Sphere, Hemisphere, Cylinder, Triangle, Square, Disc, Annulus,
  Box, Prism, Cuboctahedron, Ball, Halfball, Bowl, Spotlight, Camera, presets };
