'use strict';
// Preset scene descriptions for the Node host.  These are data: the same arrangements of
// primitives, materials, lights and camera that the reference hard-codes in Scene.loadPreset
// (rayphoton.js:731-774 preset 0, :811-854 preset 2, :880-906 preset 4, :1045-1067 preset 7,
// :1161-1181 preset 10 -- the program's default), written against this host's constructors.
// Per-shape albedo closures of the reference become `procColour` ids (see scene.js PROC).

function room(S, planes) {
  return planes.map(([o, n, c, m]) => new S.Plane(o, n, c, m));
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
};
