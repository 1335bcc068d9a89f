'use strict';
// Preset scene descriptions for the Node host.  These are data: the same arrangements of
// primitives, materials, lights and camera that the reference hard-codes in Scene.loadPreset
// (rayphoton.js:731-1305: presets 0 .. 10 -- 10 is the program's default -- 'rmt' and 'giacometti'),
// written against this host's constructors.
// Per-shape albedo closures of the reference become `procColour` ids (see scene.js PROC).
// Presets 5, 9 and 'giacometti' place or colour things at random: they draw from scene.random(), the
// host's keyed stream, in the order the reference calls Math.random().

function room(S, planes) {
  return planes.map(([o, n, c, m]) => new S.Plane(o, n, c, m));
}
// the 8 x 18 x 8 room of presets 5 .. 9 (floor, four walls, ceiling)
function studio(S) {
  const { COL, MAT } = S;
  return room(S, [
    [[0, 0, 0], [0, 0, 1], COL.SILVER, MAT.LINOLEUM], [[0, 4, 0], [0, -1, 0], COL.GRAPEFRUIT_YELLOW, MAT.PLASTER],
    [[0, -14, 0], [0, 1, 0], COL.GREY, MAT.PLASTER], [[4, 0, 0], [-1, 0, 0], COL.SKY_BLUE, MAT.PLASTER],
    [[-4, 0, 0], [1, 0, 0], COL.FIRE_ENGINE_RED, MAT.PLASTER], [[0, 0, 8], [0, 0, -1], COL.DARK_GREY, MAT.PLASTER]]);
}

// fully saturated colour of hue h in degrees (randomSaturatedColour, rayphoton.js:162-182, and the same table inlined
// in presets 5 and 8)
function saturated(h) {
  const c = Math.floor(256 * (1 - Math.abs((h / 60) % 2 - 1)));
  const rgb = [[255, c, 0], [c, 255, 0], [0, 255, c], [0, c, 255], [c, 0, 255], [255, 0, c]][Math.floor(h / 60)];
  if (!rgb) throw new Error('impossible HSV?');
  return { r: rgb[0], g: rgb[1], b: rgb[2], a: 1 };
}
// randomPointOnSphere (rayphoton.js:139-147).  The rejection test raises each coordinate to ITS OWN power (x ** x, not
// x ** 2): a negative non-integer coordinate makes the sum NaN, which ends the loop, so the accepted points are not
// uniform on the sphere.  It decides the geometry of 'giacometti', hence it is kept.
function pointOnSphere(rand, V) {
  let x, y, z;
  do { x = 2 * rand() - 1; y = 2 * rand() - 1; z = 2 * rand() - 1; } while (Math.pow(x, x) + Math.pow(y, y) + Math.pow(z, z) > 1);
  return V.normalise([x, y, z]);
}
function randomBasis(rand, V) {                             // randomONB, rayphoton.js:148-158
  const u = pointOnSphere(rand, V);
  let w;
  do { w = V.cross(u, pointOnSphere(rand, V)); } while (V.dot(w, w) < 1e-6);
  w = V.normalise(w);
  return [u, V.cross(w, u), w];
}

module.exports = {
  '0'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = room(S, [
      [[0, 0, 0], [0, 0, 1], COL.WHITE, MAT.LINOLEUM], [[0, 80, 0], [0, -1, 0], COL.SKY_BLUE, MAT.PLASTER],
      [[0, -80, 0], [0, 1, 0], COL.VERY_DARK_GREY, MAT.PLASTER], [[80, 0, 0], [-1, 0, 0], COL.VERY_DARK_GREY, MAT.PLASTER],
      [[-80, 0, 0], [1, 0, 0], COL.VERY_DARK_GREY, MAT.PLASTER], [[0, 0, 80], [1, 0, -1], COL.VERY_DARK_GREY, MAT.PLASTER]]);
    new S.Ball(sh, [0.3, 0, 0.4], 0.4, COL.FIRE_ENGINE_RED, MAT.GLASS);
    new S.Ball(sh, [-1.1, 1.2, 0.25], 0.25, COL.AMETHYST, MAT.PLASTER);
    new S.Ball(sh, [0.5, 2.5, 1.25], 1.25, COL.BLACK, MAT.COPPER);
    new S.Ball(sh, [-1.5, -2.5, 0.8], 0.8, COL.ORANGE_ORANGE, MAT.PLASTER);
    sh[0].procColour = S.PROC.STRIPE_CHECKER;
    sh[8].procColour = S.PROC.EIGHT_BALL;
    new S.Spotlight(sh, scene.lights, [0, 0, 10], 1.0, [0, 0, -1], 40);
    new S.Spotlight(sh, scene.lights, [-4, -4, 10], 0.5, [1, 1, -1], 80);
    scene.camera = new S.Camera([-1, -3, 2], 80, -20, scene.canvasWidth, scene.canvasHeight);
  },

  '2'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = room(S, [
      [[0, 0, 0], [0, 0, 1], COL.GREY, MAT.LINOLEUM], [[0, 12, 0], [0, -1, 0], COL.SKY_BLUE, MAT.PLASTER],
      [[0, -12, 0], [0, 1, 0], COL.SKY_BLUE, MAT.PLASTER], [[12, 0, 0], [-1, 0, 0], COL.SKY_BLUE, MAT.PLASTER],
      [[-12, 0, 0], [1, 0, 0], COL.SKY_BLUE, MAT.PLASTER], [[0, 0, 12], [0, 0, -1], COL.GREY, MAT.PLASTER]]);
    sh[0].procColour = S.PROC.TILE_GRID;
    const wood = COL.ENGLISH_WALNUT;
    new S.Box(sh, [-1, -1, 0.9], [2, 0, 0], [0, 2, 0], [0, 0, 0.1], wood, MAT.PLASTER);          // table top
    for (const [x, y] of [[0.9, -1], [0.9, 0.9], [-1, 0.9], [-1, -1]]) new S.Box(sh, [x, y, 0], [0.1, 0, 0], [0, 0.1, 0], [0, 0, 1], wood, MAT.PLASTER);
    new S.Bowl(sh, [0.3, -0.1, 1.5], 0.5, 0.45, [0, 0, 1], COL.WHITE, MAT.GLASS);
    new S.Halfball(sh, [0.3, -0.1, 1.5], 0.45, [0, 0, 1], 0.2, undefined, COL.WHITE, MAT.WATER);
    new S.Ball(sh, [-0.7, 0.1, 1.15], 0.15, COL.ORANGE_ORANGE, MAT.PLASTER);
    new S.Ball(sh, [-0.5, -0.5, 1.25], 0.25, COL.GRAPEFRUIT_YELLOW, MAT.PLASTER);
    new S.Ball(sh, [0.38, -0.12, 1.30], 0.05, COL.ROBINS_EGG_BLUE, MAT.PLASTER);
    new S.Can(sh, [0.6, -0.7, 1.2], [0, 0, 1], 0.2, 0.1, COL.CHOCOLATE, COL.FIRE_ENGINE_RED, MAT.COPPER);
    new S.Box(sh, [-1.25, 2.75, 0], [0.9, 0.2, 0], [-0.2, 0.9, 0], [0, 0, 0.5], COL.LIME_GREEN, MAT.PLASTER);
    new S.Spotlight(sh, scene.lights, [0, -2, 8.5], 1, [0, 0.1, -1], 60);
    scene.camera = new S.Camera([-1.1, -1.6, 2], 54, -32, scene.canvasWidth, scene.canvasHeight);
  },

  '4'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = room(S, [
      [[0, 0, 0], [0, 0, 1], COL.GREY, MAT.LINOLEUM], [[0, 12, 0], [0, -1, 0], COL.SKY_BLUE, MAT.PLASTER],
      [[0, -24, 0], [0, 1, 0], COL.SKY_BLUE, MAT.PLASTER], [[12, 0, 0], [-1, 0, 0], COL.SKY_BLUE, MAT.PLASTER],
      [[-12, 0, 0], [1, 0, 0], COL.SKY_BLUE, MAT.PLASTER], [[0, 0, 12], [0, 0, -1], COL.GREY, MAT.PLASTER]]);
    sh[0].procColour = S.PROC.ROTATED_CHECKER;
    new S.Ball(sh, [0, 0, 2], 2, COL.RAW_UMBER, MAT.COPPER);
    new S.Ball(sh, [3.2, 0.4, 1], 1, COL.LIME_GREEN, MAT.PLASTER);
    new S.Prism(sh, [-3, -2.5, 0], [9, -2.6, 0], [9, 0.2, 0], [0, 0, 2.2], COL.WHITE, MAT.GLASS);
    new S.Spotlight(sh, scene.lights, [-1, -3, 8.5], 1.0, [0, 0, -1], 40);
    new S.Spotlight(sh, scene.lights, [-6, 0, 6.5], 0.5, [1, 0, -1], 40);
    new S.Box(sh, [2, 3, 0], [2, -1, 0], [0.1, 0.2, 0], [0, 0, 5], COL.WHITE, MAT.MIRROR);
    new S.Box(sh, [4.5, 1.5, 0], [1, -2, 0], [0.2, 0.1, 0], [0, 0, 5], COL.WHITE, MAT.MIRROR);
    scene.camera = new S.Camera([-2, -13, 5.5], 68, -17, scene.canvasWidth, scene.canvasHeight);
  },

  '7'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = room(S, [
      [[0, 0, 0], [0, 0, 1], COL.SILVER, MAT.LINOLEUM], [[0, 4, 0], [0, -1, 0], COL.GRAPEFRUIT_YELLOW, MAT.PLASTER],
      [[0, -14, 0], [0, 1, 0], COL.GREY, MAT.PLASTER], [[4, 0, 0], [-1, 0, 0], COL.SKY_BLUE, MAT.PLASTER],
      [[-4, 0, 0], [1, 0, 0], COL.FIRE_ENGINE_RED, MAT.PLASTER], [[0, 0, 8], [0, 0, -1], COL.DARK_GREY, MAT.PLASTER]]);
    new S.Box(sh, [-3.5, 1.5, 1.5], [7, 0, 0], [0, 2, 0], [0, 0, 3], COL.WHITE, MAT.GLASS);       // glass tank ...
    new S.Box(sh, [-3.4, 1.6, 1.6], [6.8, 0, 0], [0, 1.8, 0], [0, 0, 2.8], COL.WHITE, MAT.AIR);   // ... hollowed by an air box
    new S.Box(sh, [-3, 2.25, 0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 1.5], COL.RAW_UMBER, MAT.PLASTER);
    new S.Box(sh, [3, 2.25, 0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 1.5], COL.RAW_UMBER, MAT.PLASTER);
    new S.Ball(sh, [-2, 2.5, 2.3], 0.7, COL.AMETHYST, MAT.COPPER);
    new S.Ball(sh, [0, 2.5, 2.3], 0.7, COL.ROBINS_EGG_BLUE, MAT.COPPER);
    new S.Ball(sh, [2, 2.5, 2.3], 0.7, COL.LIME_GREEN, MAT.COPPER);
    new S.Spotlight(sh, scene.lights, [-1.5, 0.5, 6.5], 1, [0.1, 0.3, -1], 40);
    scene.camera = new S.Camera([-2, -12.5, 3], 84, -2, scene.canvasWidth, scene.canvasHeight, 35);
  },

  '10'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = room(S, [
      [[0, 0, 0], [0, 0, 1], COL.SILVER, MAT.LINOLEUM], [[0, 10, 0], [0, -1, 0], COL.GRAPEFRUIT_YELLOW, MAT.PLASTER],
      [[10, 0, 0], [-1, 0, 0], COL.ORANGE_ORANGE, MAT.PLASTER]]);
    new S.Box(sh, [10, -5, 3], [0, 4, 0], [-0.2, 0, 0], [0, 0, 8], COL.WHITE, MAT.MIRROR);
    new S.Tube(sh, [-3, 4, 4], [3, 1, 0], 4, 3.5, 1, COL.DEEP_BLUE, COL.DEEP_BLUE, MAT.COPPER);
    new S.Ball(sh, [1, 4, 3], 3, COL.WHITE, MAT.PLASTER);
    new S.Prism(sh, [4, 0, 0], [3, -1, 0], [2, 3, 0], [0, 0, 5], COL.LIME_GREEN, MAT.LINOLEUM);
    new S.Can(sh, [5, -4, 4], [0, 0, 1], 1.5, 4, COL.WHITE, COL.WHITE, MAT.GLASS);
    new S.Ball(sh, [5, -4.5, 6], 0.3, COL.WHITE, MAT.AIR);                                          // bubbles in the glass
    new S.Ball(sh, [5.75, -3.5, 6.5], 0.4, COL.WHITE, MAT.AIR);
    new S.Ball(sh, [4.75, -4.25, 5.25], 0.2, COL.WHITE, MAT.AIR);
    new S.Spotlight(sh, scene.lights, [-1.5, 0.5, 12.5], 1, [0.1, 0.3, -1], 40);
    new S.Spotlight(sh, scene.lights, [5.5, -6.5, 12.5], 0.25, [0.2, 0.4, -1], 200);
    scene.camera = new S.Camera([-8, -13, 11], 55, -17, scene.canvasWidth, scene.canvasHeight);
  },
  '1'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = room(S, [
      [[0, 0, 0], [0, 0, 1], COL.GREY, MAT.LINOLEUM], [[0, 18, 0], [0, -1, 0], COL.VERY_DARK_GREY, MAT.PLASTER],
      [[0, -24, 0], [0, 1, 0], COL.VERY_DARK_GREY, MAT.PLASTER], [[12, 0, 0], [-1, 0, 0], COL.VERY_DARK_GREY, MAT.PLASTER],
      [[-12, 0, 0], [1, 0, 0], COL.VERY_DARK_GREY, MAT.PLASTER], [[0, 0, 12], [0, 0, -1], COL.GREY, MAT.PLASTER]]);
    sh[0].procColour = S.PROC.TILE_GRID;
    const trio = [[COL.RAW_UMBER, MAT.COPPER], [COL.WHITE, MAT.GLASS], [COL.AMETHYST, MAT.PLASTER]];
    trio.forEach(([c, m], i) => new S.Box(sh, [-5 + 3 * i, 8, 0], [1.5, -0.3, 0], [0.3, 1.5, 0], [0, 0, 0.5], c, m));
    [[COL.WHITE, MAT.GLASS], [COL.AMETHYST, MAT.PLASTER], [COL.RAW_UMBER, MAT.COPPER]].forEach(([c, m], i) => new S.Ball(sh, [-5 + 3 * i, 10, 1], 1, c, m));
    new S.Ball(sh, [-1, 14, 2], 2, COL.WHITE, MAT.MIRROR);
    new S.Spotlight(sh, scene.lights, [0, 0, 10], 1, [0, 0, -1], 40);
    new S.Spotlight(sh, scene.lights, [-6, 10, 4.5], 0.5, [1, 0, -1], 40);          // (white: the red sits outside the call, rayphoton.js:805)
    new S.Spotlight(sh, scene.lights, [-3, 10, 5.5], 0.5, [0, 0, -1], 40, COL.LIME_GREEN);
    new S.Spotlight(sh, scene.lights, [0, 10, 4.5], 0.5, [-1, 0, -1], 40, COL.DEEP_BLUE);
    scene.camera = new S.Camera([-2, -6, 5], 90, -10, scene.canvasWidth, scene.canvasHeight);
  },

  '3'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = room(S, [[[0, 0, 0], [0, 0, 1], COL.GREY, MAT.LINOLEUM]]);
    new S.Bowl(sh, [-2.3, 1, 1], 1, 0.8, [0, 0, 1], COL.DEEP_BLUE, MAT.COPPER);
    new S.Ball(sh, [-2.3, 1, 0.7], 0.5, COL.GRAPEFRUIT_YELLOW, MAT.PLASTER);
    new S.Ball(sh, [0.35, -0.8, 2.8], 0.3, COL.ORANGE_ORANGE, MAT.PLASTER);
    new S.Ball(sh, [-0.4, 3.5, 2], 2, COL.RAW_UMBER, MAT.PLASTER);
    new S.Cuboctahedron(sh, [-0.7, -2, 0], [2.5, 0, 0], [0, 2.5, 0], [0, 0, 2.5], COL.DEEP_PINK, COL.DARK_GREY, MAT.PLASTER);
    sh[0].procColour = S.PROC.ROTATED_CHECKER;
    new S.Spotlight(sh, scene.lights, [0, -2, 8.5], 1, [0, 0.1, -1], 40);
    scene.camera = new S.Camera([-3, -7, 4.5], 68, -16, scene.canvasWidth, scene.canvasHeight);
  },

  // five plaster blocks of random place, turn, height and hue; a mirror ball on the tallest, a glass ball on the shortest
  '5'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = studio(S), rand = () => scene.random();
    let tall = [0, 0, 0], short = [0, 0, Infinity];
    for (let b = 0; b < 5; b++) {
      const cx = 6 * rand() - 3, cy = 6 * rand() - 3, theta = Math.PI / 2 * rand();
      const cosT = Math.cos(theta), sinT = Math.sin(theta), base = 0, height = 0.1 + 3 * rand();
      if (base + height > tall[2]) tall = [cx, cy, base + height];
      if (base + height < short[2]) short = [cx, cy, base + height];
      const colour = saturated(360 * rand());
      new S.Box(sh, [cx - (cosT + sinT) / 2, cy - (-sinT + cosT) / 2, base], [cosT, -sinT, 0], [sinT, cosT, 0], [0, 0, height], colour, MAT.PLASTER);
    }
    new S.Ball(sh, [tall[0], tall[1], tall[2] + 0.5], 0.5, COL.WHITE, MAT.MIRROR);
    new S.Ball(sh, [short[0], short[1], short[2] + 0.5], 0.5, COL.WHITE, MAT.GLASS);
    new S.Spotlight(sh, scene.lights, [-1.5, -1, 6.75], 1, [0.5, 0, -1], 20);
    new S.Spotlight(sh, scene.lights, [1.5, 1, 6.75], 1, [-0.5, 0, -1], 20);
    scene.camera = new S.Camera([0, -10, 6], 90, -20, scene.canvasWidth, scene.canvasHeight);
  },

  '6'(scene, S) {                                           // (the reference's loop of random balls runs zero times, rayphoton.js:1000)
    const { COL, MAT } = S, sh = scene.shapes = studio(S);
    new S.Ball(sh, [-2.5, 2.5, 1], 1, COL.WHITE, MAT.GLASS);
    new S.Ball(sh, [0, 2.5, 1], 1, COL.ROBINS_EGG_BLUE, MAT.PLASTER);
    new S.Ball(sh, [2.5, 2.5, 1], 1, COL.RAW_UMBER, MAT.COPPER);
    new S.Spotlight(sh, scene.lights, [-1.5, 0.5, 6.5], 1, [0.1, 0.3, -1], 40);
    new S.Box(sh, [-3.75, -3.5, -0.1], [7.5, 4, 0], [-0.2, 0.375, 0], [0, 0, 4], COL.WHITE, MAT.GLASS);
    scene.camera = new S.Camera([0, -13.5, 4.5], 90, -10, scene.canvasWidth, scene.canvasHeight);
  },

  // five balls in a row, hues 72 degrees apart, each behind a glass pane turned 22.5 degrees more than the last
  '8'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = studio(S), DEG = Math.PI / 180;
    for (let b = -2; b <= 2; b++) {
      const cx = 1.5 * b, cy = 3;
      new S.Ball(sh, [cx, cy, 0.6], 0.6, saturated(72 * (b + 2)), MAT.PLASTER);
      const cosT = Math.cos(22.5 * (b + 2) * DEG), sinT = Math.sin(22.5 * (b + 2) * DEG);
      new S.Box(sh, [cx + (-1.4 * cosT + 0.04 * sinT) / 2, cy - 2 + (-1.4 * sinT - 0.04 * cosT) / 2, 0], [1.4 * cosT, 1.4 * sinT, 0],
        [-0.04 * sinT, 0.04 * cosT, 0], [0, 0, 4], COL.WHITE, MAT.GLASS);
    }
    new S.Spotlight(sh, scene.lights, [-1.5, 0.5, 6.5], 1.0, [0.1, 0.3, -1], 40);
    scene.camera = new S.Camera([-1, -8.6, 2, 5], 85, -5, scene.canvasWidth, scene.canvasHeight, 55);   // (a four-element origin, rayphoton.js:1121)
  },

  // steps with six balls of random hue on top; three glass balls: solid, thick-walled, thin-walled
  '9'(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = studio(S);
    for (const [y, dy, dz] of [[1.5, 2, 3], [0.5, 1, 2], [-0.5, 1, 1]]) new S.Box(sh, [-3.5, y, 1.5], [7, 0, 0], [0, dy, 0], [0, 0, dz], COL.RAW_UMBER, MAT.PLASTER);
    for (let b = 0; b <= 20; b += 4) new S.Ball(sh, [-3.4 + 6.8 * b / 20, 1, 3.65], 0.45, saturated(360 * scene.random()), MAT.PLASTER);
    new S.Ball(sh, [-2.5, -2, 1], 1, COL.WHITE, MAT.GLASS);
    new S.Ball(sh, [0, -2, 1], 1, COL.WHITE, MAT.GLASS);
    new S.Ball(sh, [0, -2, 1], 0.5, COL.WHITE, MAT.AIR);
    new S.Ball(sh, [2.5, -2, 1], 1, COL.WHITE, MAT.GLASS);
    new S.Ball(sh, [2.5, -2, 1], 0.98, COL.WHITE, MAT.AIR);
    new S.Spotlight(sh, scene.lights, [-1.5, 0.5, 6.5], 1, [0.1, 0.3, -1], 40);
    new S.Spotlight(sh, scene.lights, [0.5, -8.5, 0.5], 1, [-0.1, 1, 0.2], 40);
    scene.camera = new S.Camera([-2, -13.6, 3.0], 85, -1, scene.canvasWidth, scene.canvasHeight);
  },

  // a bedroom: walls and window wall of un-nudged boxes, two glass panes, a bed, a bowl of water
  rmt(scene, S) {
    const { COL, MAT } = S, sh = scene.shapes = room(S, [
      [[0, 0, 0], [0, 0, 1], COL.ENGLISH_WALNUT, MAT.PLASTER], [[0, 25, 0], [0, -1, 0], COL.SKY_BLUE, MAT.PLASTER],
      [[0, -13, 0], [0, 1, 0], COL.SKY_BLUE, MAT.PLASTER], [[-9, 0, 0], [-1, 0, 0], COL.SKY_BLUE, MAT.PLASTER],
      [[-9, 0, 0], [1, 0, 0], COL.SKY_BLUE, MAT.PLASTER], [[0, 0, 100], [0, 0, -1], COL.GREY, MAT.PLASTER]]);
    const wall = (a, ab, ac, ad, c) => new S.Box(sh, a, ab, ac, ad, c, MAT.PLASTER, 0);
    wall([-8, -8, 0], [0, 20, 0], [-1, 0, 0], [0, 0, 10], COL.VANILLA);
    wall([8, -8, 0], [1, 0, 0], [0, 20, 0], [0, 0, 10], COL.VANILLA);
    wall([-8, -8, 10], [16, 0, 0], [0, 20, 0], [0, 0, 10], COL.WARM_GREY);
    for (const [a, w, h] of [[[-8, 12, 0], 16, 5], [[-8, 12, 8], 16, 2], [[-8, 12, 0], 8, 10], [[1.5, 12, 0], 1.5, 10], [[4.5, 12, 0], 3.5, 10]]) {
      wall(a, [w, 0, 0], [0, 1, 0], [0, 0, h], COL.WARM_GREY);
    }
    for (const x of [0, 3]) new S.Box(sh, [x, 12.25, 5], [1.5, 0, 0], [0, 0.5, 0], [0, 0, 3], COL.WHITE, MAT.GLASS);      // window panes
    const wood = COL.ENGLISH_WALNUT;
    new S.Box(sh, [-1, 3, 3.8], [7, 0, 0], [0, 3, 0], [0, 0, 0.2], wood, MAT.PLASTER);                                   // bed
    for (const [x, y] of [[5.7, 3], [5.7, 5.7], [-1, 5.7], [-1, 3]]) new S.Box(sh, [x, y, 0], [0.3, 0, 0], [0, 0.3, 0], [0, 0, 3.8], wood, MAT.PLASTER);
    new S.Box(sh, [4.5, 2.9, 4], [1.4, 0, 0], [0, 2.8, 0], [0, 0, 0.2], COL.WHITE, MAT.PLASTER);                         // pillow
    new S.Box(sh, [-4, 10, 0], [1.2, 0.3, 0], [-0.3, 1.2, 0], [0, 0, 5], COL.AMETHYST, MAT.COPPER);
    new S.Ball(sh, [-3.5, 10.6, 5.4], 0.4, COL.LIME_GREEN, MAT.MIRROR);
    new S.Bowl(sh, [2, 4, 5], 1, 0.8, [0, 0, 1], COL.WHITE, MAT.GLASS);
    new S.Halfball(sh, [2, 4, 5], 0.8, [0, 0, 1], 0.3, undefined, COL.WHITE, MAT.WATER);
    new S.Spotlight(sh, scene.lights, [0, 5, 8.5], 1, [0, 0, -1], 15);
    new S.Spotlight(sh, scene.lights, [-8, 14, 10], 2, [5, 12, 0], 30, COL.GRAPEFRUIT_YELLOW);
    new S.Spotlight(sh, scene.lights, [-8, 22, 10], 2, [10, -10, 0], 45, COL.GRAPEFRUIT_YELLOW);
    scene.camera = new S.Camera([-2.3, -7.6, 6.75], 85, -4, scene.canvasWidth, scene.canvasHeight);
  },

  // a plinth in a corner with four random solids (ball, can, box or prism) hovering over it
  giacometti(scene, S) {
    const { COL, MAT } = S, V = require('./vec'), rand = () => scene.random();
    const sh = scene.shapes = room(S, [
      [[0, 0, 0], [0, 0, 1], COL.WHITE, MAT.LINOLEUM], [[0, 8, 0], [0, -1, 0], COL.GRAPEFRUIT_YELLOW, MAT.PLASTER],
      [[12, 0, 0], [-1, 0, 0], COL.ORANGE_ORANGE, MAT.PLASTER]]);
    new S.Box(sh, [12, 8, 0], [-80, 0, 0], [0, -0.1, 0], [0, 0, 1], COL.WHITE, MAT.PLASTER);       // skirting boards
    new S.Box(sh, [12, 8, 0], [-0.1, 0, 0], [0, -80, 0], [0, 0, 1], COL.WHITE, MAT.PLASTER);
    new S.Box(sh, [0, 1, 0], [6, 0, 0], [0, 6, 0], [0, 0, 0.5], COL.DARK_GREY, MAT.PLASTER);
    new S.Box(sh, [1, 2, 0.5], [4, 0, 0], [0, 4, 0], [0, 0, 0.5], COL.DARK_GREY, MAT.PLASTER);
    new S.Box(sh, [2.5, 3.5, 1], [1, 0, 0], [0, 1, 0], [0, 0, 4], COL.WHITE, MAT.PLASTER);
    for (let s = 0; s < 4; s++) {
      const centre = [3 + (2 * rand() - 1), 4 + (2 * rand() - 1), 6 + (4 * rand() - 2)];
      const kind = Math.floor(4 * rand()), colour = COL.WHITE, material = MAT.LINOLEUM;
      if (kind === 0) {
        new S.Ball(sh, centre, 0.5 * rand() + 0.5, colour, material);
      } else if (kind === 1) {
        const axis = pointOnSphere(rand, V), radius = 0.5 * rand() + 0.5, halfHeight = 0.5 * rand() + 0.5;
        new S.Can(sh, centre, axis, radius, halfHeight, colour, colour, material);
      } else {
        const basis = randomBasis(rand, V), lengths = [1.5 * rand() + 0.5, 1.5 * rand() + 0.5, 2.5 * rand() + 0.5];
        let corner = centre;
        for (let i = 0; i < 3; i++) corner = V.sub(corner, V.scale(0.5 * lengths[i], basis[i]));
        const edges = basis.map((e, i) => V.scale(lengths[i], e));
        if (kind === 2) new S.Box(sh, corner, edges[0], edges[1], edges[2], colour, material);
        else new S.Prism(sh, corner, edges[0], edges[1], edges[2], colour, material);
      }
    }
    new S.Spotlight(sh, scene.lights, [6, 5, 18], 0.5, [-0.2, 0.1, -1], 25);
    new S.Spotlight(sh, scene.lights, [-12, 5, 18], 0.5, [0.6, 0.1, -1], 25);
    new S.Spotlight(sh, scene.lights, [2, -15, 4], 0.5, [0, 1, 0.2], 40);
    scene.camera = new S.Camera([-12, -4, 7.5], 31, -12, scene.canvasWidth, scene.canvasHeight);
  },
};
module.exports.helpers = { saturated, pointOnSphere, randomBasis };
