'use strict';
// Node host of the MI355X ray-tracing / photon-mapping core: the reference's JavaScript-visible
// surface (Scene, Ray, Photon, KdTree; rayphoton.js:483-488, :647-654, :715-727, kdtree.js:1-8)
// kept as the API, with every hot path running in HIP behind the N-API shim
// (napi/rt355_napi.cc -> librt355.so).  Nothing here computes intersections, photon walks,
// tree builds or gathers in JavaScript; if the addon or a GPU is missing the calls throw.

const path = require('path');
const S = require('./scene');
const presets = require('./presets');
const RNG = require('./rng');
const W = require('./whitted');

let native = null;
function addon() {
  if (!native) {
    try { native = require(path.join(__dirname, 'napi', 'rt355.node')); }
    catch (e) { throw new Error(`rt355.node is not built (make -C raytracer_amd/host/napi): ${e.message}`); }
  }
  return native;
}

let sharedCtx = null;
function context(device) {
  if (!sharedCtx) sharedCtx = addon().init(device || 0);
  return sharedCtx;
}

const PRECISION = { F32: 0, F64: 1 };
// record formats of device-resident photon lists and shards: FULL = the 11 record values in the working precision,
// PACKED = position, direction, power * colour as 9 floats (what the photon map keeps)
const SHARD = { FULL: 0, PACKED: 1 };

// Plain GPU memory: where a queued frame's pixels, a shard or a batch of gather points live until somebody reads them.
class DeviceBuffer {
  constructor(bytes, ctx) { this.bytes = bytes; this.handle = addon().deviceAlloc(ctx || context(), bytes); }
  // copies out.byteLength bytes from byteOffset into `out` (any typed array); waits for queued frames first
  read(out, byteOffset) { return addon().deviceRead(this.handle, out, byteOffset || 0); }
  write(src, byteOffset) { addon().deviceWrite(this.handle, src, byteOffset || 0); return this; }
  free() { addon().deviceFree(this.handle); }
}

// the reference's top-of-file constants (rayphoton.js:6-23) as overridable defaults
const DEFAULTS = {
  epsilon: 1e-6, maxTraceDist: 20, maxDepth: 5, subSample: 6, numPhotonsDiffuse: 20000, numPhotonsCaustic: 50000,
  causticRadius: 0.15, diffuseRadius: 0.40, standardLampArea: 100, seed: 12345, precision: PRECISION.F32,
  superSampleBase: 3, superSampleLevels: 10,                  // rayphoton.js:14-15
};

class Ray {
  constructor(origin, dir) { this.origin = origin; this.dir = require('./vec').normalise(dir); }
  // distance to one shape, or undefined (rayphoton.js:490-644) -- one-pair launch of the geometry kernel
  intersectDist(shape, params) {
    const f = S.flattenShape(shape);
    const t = addon().intersectPairs(context(), Object.assign({ precision: PRECISION.F64 }, params || {}),
      Float64Array.from(this.origin), Float64Array.from(this.dir), 0,
      Uint8Array.from([f.typeId, f.material, f.flags, f.proc]), Float64Array.from(f.g));
    return Number.isNaN(t[0]) ? undefined : t[0];
  }
}

class Photon extends Ray {
  constructor(origin, dir, power, colour, isCaustic) {
    super(origin, dir);
    this.power = power; this.colour = colour || S.COL.WHITE; this.isCaustic = isCaustic;
  }
}

// photon list <-> the 11-double records of the C ABI (origin3 dir3 power rgba)
function photonsToRecords(photons) {
  const rec = new Float64Array(photons.length * 11);
  photons.forEach((p, i) => {
    const c = p.colour || S.COL.WHITE;
    rec.set([p.origin[0], p.origin[1], p.origin[2], p.dir[0], p.dir[1], p.dir[2], p.power === undefined ? 0 : p.power,
      c.r, c.g, c.b, c.a === undefined ? NaN : c.a], i * 11);
  });
  return rec;
}
function recordsToPhotons(rec, isCaustic) {
  const out = new Array(rec.length / 11);
  for (let i = 0; i < out.length; i++) {
    const r = rec.subarray(i * 11, i * 11 + 11);
    const p = Object.create(Photon.prototype);          // records are already normalised: skip the ctor's renormalisation
    p.origin = [r[0], r[1], r[2]]; p.dir = [r[3], r[4], r[5]]; p.power = r[6];
    p.colour = { r: r[7], g: r[8], b: r[9], a: r[10] }; p.isCaustic = isCaustic;
    out[i] = p;
  }
  return out;
}

class KdTree {
  // photons: array of {origin, dir, power, colour} (anything with .origin works for pure queries),
  // a Float64Array of 11-double records, or {lists, which, n}: one of the two device-resident lists of an emission
  // (the tree is then built where the photons are; records and photon objects cross to the host only if somebody looks)
  constructor(data) {
    const fromLists = data && data.lists !== undefined;
    const n = fromLists ? data.n : (ArrayBuffer.isView(data) ? data.length / 11 : data.length);
    console.log(`building a k-d tree with ${n} elements`);                 // kdtree.js:3
    if (n === 0) throw new Error('tried to build a k-d tree with no data!');   // kdtree.js:6
    this._n = n;
    this._nodes = null;
    if (fromLists) {
      this.photons = null; this._records = null; this._lists = data.lists; this._which = data.which;
      this.map = addon().photonMapBuildLists(data.lists, data.which);
      return;
    }
    this.photons = ArrayBuffer.isView(data) ? null : data;
    this._records = ArrayBuffer.isView(data) ? data : photonsToRecords(data);
    this.map = addon().photonMapBuild(context(), this._records);
  }
  // the 11-double records (of a device-resident list: exported on first use; such lists must be in the FULL format)
  get records() {
    if (!this._records) this._records = addon().listsExport(this._lists, this._which).records;
    return this._records;
  }
  get length() { return this._n; }
  photonAt(i) {
    if (!this.photons) this.photons = recordsToPhotons(this.records, undefined);
    return this.photons[i];
  }
  // nodes[pos] = {dim, photon}, 1-based implicit heap (kdtree.js:104), materialised on first use
  get nodes() {
    if (!this._nodes) {
      const { perm, dim } = addon().photonMapNodes(this.map);
      this._nodes = [undefined];
      for (let s = 0; s < perm.length; s++) this._nodes.push({ dim: dim[s], photon: this.photonAt(perm[s]) });
    }
    return this._nodes;
  }
  // all photons within r of x, in the reference's pre-order (kdtree.js:64-80)
  nearestNeighbours(x, r, pos) {
    if (pos !== undefined && pos !== 1) throw new Error('nearestNeighbours: sub-tree queries (pos != 1) are not offered by the device tree');
    return this.nearestNeighboursBatch(Float32Array.from(x), r)[0];
  }
  // many queries in one launch: queries = Float32Array(nq*3) -> array of photon lists
  nearestNeighboursBatch(queries, r, cap) {
    const nq = queries.length / 3;
    cap = cap || 64;
    let res = addon().kdRadius(this.map, queries, r, cap);
    let most = 0;
    for (let i = 0; i < nq; i++) most = Math.max(most, res.count[i]);
    if (most > cap) { cap = most; res = addon().kdRadius(this.map, queries, r, cap); }
    const out = [];
    for (let i = 0; i < nq; i++) {
      const list = [];
      for (let k = 0; k < res.count[i]; k++) list.push(this.photonAt(res.idx[i * cap + k]));
      out.push(list);
    }
    return out;
  }
  // k nearest photons ordered by (squared distance, input index): [{photon, index, distSq}]
  kNearest(x, k) {
    const res = addon().kdKnn(this.map, Float32Array.from(x), k);
    const out = [];
    for (let i = 0; i < k && res.idx[i] >= 0; i++) out.push({ photon: this.photonAt(res.idx[i]), index: res.idx[i], distSq: res.d2[i] });
    return out;
  }
  // The adaptive-radius estimate the reference sketches in a commented block (rayphoton.js:1766-1778, :1789-1801):
  // start at r0, triple the radius (up to 10 times) while fewer than 10 photons are inside, give nothing below 3.
  // hits = Float32Array(nq*9): position, normal, albedo.  -> {rgb: Float32Array(nq*3), radius: Float32Array(nq)}
  gatherAdaptive(hits, r0, grow, minCount, minKeep, maxAttempts) {
    return addon().gatherRadianceAdaptive(this.map, hits, r0, grow === undefined ? 3 : grow, minCount === undefined ? 10 : minCount,
      minKeep === undefined ? 3 : minKeep, maxAttempts === undefined ? 10 : maxAttempts);
  }
  // [point, distance] as the reference computes them (kdtree.js:10-62): the closest photon of the ONE root-to-leaf path
  // the query descends -- the far side of a split is never searched (`if (false)`, :34, :50), so the answer is
  // approximate -- and [undefined, undefined] where that path leaves the tree through a missing upper child (:13).
  // The device tree holds fp32 positions: the distance is the one to the photon's rounded position.
  nearestNeighbourAndDistance(x, pos) {
    if (pos !== undefined && pos !== 1) throw new Error('nearestNeighbourAndDistance: sub-tree queries are not offered by the device tree');
    const res = addon().kdNearestPath(this.map, Float32Array.from(x));
    return res.idx[0] < 0 ? [undefined, undefined] : [this.photonAt(res.idx[0]).origin, res.dist[0]];
  }
  // the exact nearest photon: [point, distance]
  nearestExact(x) {
    const best = this.kNearest(x, 1)[0];
    return best ? [best.photon.origin, Math.sqrt(best.distSq)] : [undefined, undefined];
  }
  findMedian(n) {
    if (n === 0) throw new Error(':O');                                      // kdtree.js:110
    return addon().findMedian(n);
  }
}

class Scene {
  // ctx: anything with width/height and (optionally) fillStyle/fillRect, like the canvas context
  // the reference draws into (rayphoton.js:716-727, :1917-1918)
  constructor(ctx, options) {
    this.ctx = ctx;
    this.canvasWidth = ctx.width; this.canvasHeight = ctx.height;
    this.shapes = []; this.lights = [];
    this.photonListDiffuse = []; this.photonListCaustic = []; this.photonListBad = [];
    this.photonKdTDiffuse = []; this.photonKdTCaustic = [];
    this.params = Object.assign({}, DEFAULTS, options || {});
    this._device = null;
  }

  // All thirteen scenes of rayphoton.js:731-1305, built with this host's constructors (presets.js).  Those that place
  // or colour things at random draw from random(): the keyed stream (seed, AUX, 1000 + id; 1099 for the named presets),
  // so a preset is the same scene on every run and on every rank.
  loadPreset(def, emit) {
    const build = presets[String(def)];
    if (build && String(def) !== 'helpers') {
      this.lights = [];
      this._random = new RNG.Stream(this.params.seed, RNG.DOMAIN.AUX, 1000 + (typeof def === 'number' ? def : 99), 0);
      build(this, S);
    } else { console.log('No such scene!'); }                                // rayphoton.js:1306
    this.initLights();
    if (emit !== false) this.emitManyPhotons();          // (false: a rank of a multi-GPU job emits its share itself)
  }
  // the next number of the scene's own random stream (what Math.random() is to the reference's scene code)
  random() {
    if (!this._random) this._random = new RNG.Stream(this.params.seed, RNG.DOMAIN.AUX, 999, 0);
    return this._random.next();
  }

  initLights() {                                                              // rayphoton.js:1313-1324
    let total = 0;
    for (const l of this.lights) { const s = this.shapes[l]; s.intensity = s.wattage * s.area; total += s.intensity; }
    for (const l of this.lights) { const s = this.shapes[l]; s.prob = s.intensity / total; }
    this._device = null;
  }

  flatten() { return S.flatten(this); }

  _scene() {                  // uploads on first use; call initLights()/invalidate() after editing shapes
    if (!this._device) this._device = addon().sceneUpload(context(), this.flatten());
    return this._device;
  }
  invalidate() { this._device = null; }

  // emitManyPhotons (rayphoton.js:1354-1383): both phases on the GPU, then both kd-trees.  The two photon lists stay in
  // GPU memory (scene.photonLists) and the trees are built from them there; photonListDiffuse / photonListCaustic and the
  // record arrays are fetched when somebody reads them.  params.listFormat = SHARD.PACKED keeps 36-byte photons only
  // (enough for the maps and for rendering; the photon objects and `records` are then not available).
  emitManyPhotons() {
    if (this.lights.length === 0) { console.log("No lights! Can't emit photons."); return; }
    const format = this.params.listFormat === undefined ? SHARD.FULL : this.params.listFormat;
    const res = addon().emitManyLists(this._scene(), this.params, format);
    this.emitStats = res.stats;
    this.photonLists = res.lists;
    const lazyRecords = (which, n) => { let rec = null; return () => (rec = rec || (n ? addon().listsExport(res.lists, which).records : new Float64Array(0))); };
    const recD = lazyRecords(0, res.stats.nDiffuse), recC = lazyRecords(1, res.stats.nCaustic);
    Object.defineProperty(this, '_recD', { configurable: true, get: recD });
    Object.defineProperty(this, '_recC', { configurable: true, get: recC });
    let listD = null, listC = null;     // photon object lists are materialised only if somebody looks
    Object.defineProperty(this, 'photonListDiffuse', { configurable: true, get: () => (listD = listD || recordsToPhotons(recD(), false)) });
    Object.defineProperty(this, 'photonListCaustic', { configurable: true, get: () => (listC = listC || recordsToPhotons(recC(), true)) });
    if (res.stats.capped) {
      console.log(`I tried many times but only created ${res.stats.nDiffuse} diffuse / ${res.stats.nCaustic} caustic photons.`);
      if (res.stats.nDiffuse < this.params.numPhotonsDiffuse) return;         // rayphoton.js:1365-1368: no trees
    }
    this.photonKdTDiffuse = new KdTree({ lists: res.lists, which: 0, n: res.stats.nDiffuse });
    if (!res.stats.capped) this.photonKdTCaustic = new KdTree({ lists: res.lists, which: 1, n: res.stats.nCaustic });
  }

  // pixels [x0,x0+w) x [y0,y0+h): {pre, post: Float64Array(w*h*4), rgba8: Uint8Array, stats}
  render(x0, y0, w, h, overrides) {
    const [md, mc] = this._maps();
    return addon().traceTile(this._scene(), Object.assign({}, this.params, overrides || {}), md, mc, x0, y0, w, h);
  }
  // The same tile, only QUEUED on the GPU: out = {rgba8, pre, post}, DeviceBuffers (any subset; 4 bytes a pixel for rgba8,
  // 32 for pre / post).  Returns at once; the pixels are there after Scene.synchronize() or a read() of the buffer.
  // Frames queued one after the other run back to back without the host in between, and after
  // Scene.setFramesInFlight(n) up to n of them overlap (give those separate buffers).
  renderQueued(x0, y0, w, h, out, overrides) {
    const [md, mc] = this._maps();
    const dev = {};
    for (const k of ['pre', 'post', 'rgba8']) if (out[k]) dev[k] = out[k].handle || out[k];
    addon().traceTile(this._scene(), Object.assign({}, this.params, overrides || {}), md, mc, x0, y0, w, h, dev);
  }
  // n frames of the whole canvas, queued with up to inFlight of them overlapping -> [Uint8Array(w*h*4)] (rgba8).
  // perFrame(i) may return parameter overrides for frame i (a seed per frame, say).
  renderFrames(n, perFrame, inFlight) {
    const w = this.canvasWidth, h = this.canvasHeight, ctx = context();
    const bufs = [];
    for (let i = 0; i < n; i++) bufs.push(new DeviceBuffer(w * h * 4, ctx));
    addon().setFramesInFlight(ctx, Math.max(1, Math.min(inFlight || 3, n, 8)));
    try {
      for (let i = 0; i < n; i++) this.renderQueued(0, 0, w, h, { rgba8: bufs[i] }, perFrame ? perFrame(i) : undefined);
      addon().synchronize(ctx);
      return bufs.map((b) => b.read(new Uint8Array(w * h * 4)));
    } finally {
      addon().setFramesInFlight(ctx, 1);
      bufs.forEach((b) => b.free());
    }
  }
  static setFramesInFlight(n) { addon().setFramesInFlight(context(), n); }
  static synchronize() { addon().synchronize(context()); }

  // ---- the reference's single-item methods: batches of one on the GPU entry points that take explicit items.
  // The reference draws from Math.random(); here every call names a keyed stream and a position in it: `stream` is
  // {id, draw} (id: a pixel's y * width + x for rays and points, an attempt number for photons), by default the next
  // id of a counter of the scene and draw 0.  The stream's position after the call is left in scene.lastDraw where
  // the entry point reports it.
  _nextStream() { this._streamCounter = (this._streamCounter || 0) + 1; return 0x40000000 + this._streamCounter; }
  static _stackBytes(materialStack) {
    const st = materialStack || [S.MAT.AIR], out = new Uint8Array(16);
    out[0] = st.length; out.set(st, 1);
    return out;
  }
  _maps() {
    return [this.photonKdTDiffuse instanceof KdTree ? this.photonKdTDiffuse.map : null, this.photonKdTCaustic instanceof KdTree ? this.photonKdTCaustic.map : null];
  }

  // traceRay(ray, maxDist, depth, importance, materialStack) (rayphoton.js:1626-1838) -> {r, g, b, a}
  traceRay(ray, maxDist, depth, importance, materialStack, stream) {
    const [md, mc] = this._maps();
    const st = stream || { id: this._nextStream(), draw: 0 };
    const c = addon().traceRays(this._scene(), this.params, md, mc, Float64Array.from(ray.origin.slice(0, 3)), Float64Array.from(ray.dir), {
      maxDist: Float64Array.of(maxDist === undefined ? this.params.maxTraceDist : maxDist), depth: Int32Array.of(depth || 0),
      importance: Float64Array.of(importance === undefined ? 1 : importance), stacks: Scene._stackBytes(materialStack),
      firstStream: st.id, firstDraw: Uint32Array.of(st.draw || 0) });
    return { r: c[0], g: c[1], b: c[2], a: Number.isNaN(c[3]) ? undefined : c[3] };
  }

  // directLight(point, normal, localColour) (rayphoton.js:1841-1880) -> {r, g, b, a: 1}
  directLight(point, normal, localColour, stream) {
    const st = stream || { id: this._nextStream(), draw: 0 };
    const res = addon().directLight(this._scene(), this.params, Float64Array.from(point), Float64Array.from(normal),
      Float64Array.of(localColour.r, localColour.g, localColour.b), st.id, Uint32Array.of(st.draw || 0));
    this.lastDraw = res.draws[0];
    return { r: res.rgb[0], g: res.rgb[1], b: res.rgb[2], a: 1 };
  }

  // createPhoton(source) (rayphoton.js:1326-1344): a photon leaving a light -- light number `source`, or one chosen by
  // intensity.  The photon remembers its stream so that emitPhoton(photon) carries on where createPhoton stopped: the
  // two calls together are one emission attempt (stream.id = the attempt, stream.phase = 1 or 2).
  createPhoton(source, stream) {
    const st = stream || { id: this._nextStream(), phase: 1 };
    const res = addon().createPhotons(this._scene(), this.params, 1, st.phase || 1, st.id, source === undefined ? null : Int32Array.of(source));
    const p = recordsToPhotons(res.photons, undefined)[0];
    p.stream = { id: st.id, phase: st.phase || 1, draw: res.draws[0] };
    return p;
  }

  // emitPhoton(photon, depth, trackOnlyCaustic, materialStack) (rayphoton.js:1461-1557): follows the photon to its end;
  // what the reference would hand to storePhoton goes to this.storePhoton.  Returns the stored photon, if any.
  emitPhoton(photon, depth, trackOnlyCaustic, materialStack) {
    const st = photon.stream || { id: this._nextStream(), phase: trackOnlyCaustic ? 2 : 1, draw: 0 };
    const res = addon().walkPhotons(this._scene(), this.params, photonsToRecords([photon]), st.phase || 1, st.id, {
      firstDraw: Uint32Array.of(st.draw || 0), depth: Int32Array.of(depth || 0),
      caustic: Int8Array.of(photon.isCaustic === undefined ? -1 : (photon.isCaustic ? 1 : 0)),
      stacks: Scene._stackBytes(materialStack), trackOnlyCaustic: trackOnlyCaustic ? 1 : 0 });
    if (!res.kind[0]) return undefined;
    const stored = recordsToPhotons(res.photons, res.kind[0] === 2)[0];
    this.storePhoton(stored);
    return stored;
  }

  // storePhoton (rayphoton.js:1346-1352): the two lists, each up to its cap
  storePhoton(photon) {
    if (photon.isCaustic) { if (this.photonListCaustic.length < this.params.numPhotonsCaustic) this.photonListCaustic.push(photon); }
    else if (this.photonListDiffuse.length < this.params.numPhotonsDiffuse) this.photonListDiffuse.push(photon);
  }

  // traceTile / traceOnCanvas (rayphoton.js:1573-1624) at superSampleScale 0: one launch per tile,
  // then the same fillStyle / fillRect calls per pixel that putPixel makes (:1910-1919)
  traceTile(tile, tileSize, superSampleScale) {
    const x0 = tile[0], y0 = tile[1];
    const w = Math.min(tileSize, this.canvasWidth - x0), h = Math.min(tileSize, this.canvasHeight - y0);
    if (w <= 0 || h <= 0) return;
    // scale > 0: one jittered sample stands for a block of base ** scale pixels (rayphoton.js:1574-1596), drawn as one
    // filled square (:1911)
    const s = superSampleScale ? this.params.superSampleBase ** superSampleScale : 1;
    const out = this.render(x0, y0, w, h, s > 1 ? { superSampleBlock: s } : undefined);
    if (this.ctx && typeof this.ctx.fillRect === 'function') {
      const wb = Math.ceil(w / s), hb = Math.ceil(h / s);
      for (let y = 0; y < hb; y++) for (let x = 0; x < wb; x++) {
        const o = (y * wb + x) * 4, p = out.post;
        this.ctx.fillStyle = 'rgba(' + p[o] + ',' + p[o + 1] + ',' + p[o + 2] + ',' + p[o + 3] + ')';
        this.ctx.fillRect(x0 + x * s, y0 + y * s, s, s);
      }
    }
    return out;
  }
  traceOnCanvas(canvasX, canvasY, superSampleScale) { return this.traceTile([canvasX, canvasY], 1, superSampleScale); }

  // The order in which the reference's superSampleTiles (rayphoton.js:2108-2244) draws: coarse preview levels first,
  // and inside a level the tiles where the photon density changes most (shadow edges, caustics), never a tile before
  // the larger tile that covers it.  Returns [{region: [x, y], size, scale, photonDiff}] in drawing order.  (The
  // reference breaks ties with an engine-dependent random shuffle; here ties keep row-major order.)
  progressiveTiles() {
    const B = this.params.superSampleBase, L = this.params.superSampleLevels;
    const tileSize = Math.max(1, Math.floor(50 / this.params.subSample));
    const tx = Math.floor(this.canvasWidth / tileSize) + 1, ty = Math.floor(this.canvasHeight / tileSize) + 1;
    const count = new Float64Array(tx * ty);
    for (const rec of [this._recD, this._recC]) {
      if (!rec) continue;
      for (let i = 0; i < rec.length; i += 11) {
        const pr = this.projectToCanvas([rec[i], rec[i + 1], rec[i + 2]]);
        if (!pr) continue;
        const x = Math.floor(pr.x / tileSize), y = Math.floor(pr.y / tileSize);
        if (x >= 0 && x < tx && y >= 0 && y < ty) count[y * tx + x]++;
      }
    }
    const tiles = [];
    for (let scale = 0; scale <= L; scale++) {
      const step = B ** scale;
      if (step > Math.max(tx, ty) * B) break;                 // (levels coarser than the canvas repeat the same single tile)
      for (let y = 0; y < ty; y += step) for (let x = 0; x < tx; x += step) {
        let diff = 0;
        for (let j = -2; j <= 2; j++) for (let i = -2; i <= 2; i++) {
          if ((i || j) && y + j >= 0 && y + j < ty && x + i >= 0 && x + i < tx) diff += (count[y * tx + x] - count[(y + j) * tx + (x + i)]) ** 2 / (i * i + j * j);
        }
        tiles.push({ region: [x * tileSize, y * tileSize], size: tileSize * step, scale, photonDiff: diff, _gx: x, _gy: y });
      }
      // fold this level's counts into the corner tile of every group of the next level
      const next = step * B;
      for (let y = 0; y < ty; y += next) for (let x = 0; x < tx; x += next) {
        for (let bj = 0; bj < B; bj++) for (let bi = 0; bi < B; bi++) {
          const i = bi * step, j = bj * step;
          if ((i || j) && x + i < tx && y + j < ty) count[y * tx + x] += count[(y + j) * tx + (x + i)];
        }
      }
    }
    // drawing order: a tile is ready once the tile covering it one level up has been drawn; the ready tile with the
    // largest scale, then the largest photonDiff, goes next
    const key = (t) => `${t.scale}:${t._gx}:${t._gy}`;
    const byKey = new Map(tiles.map((t) => [key(t), t]));
    const top = Math.max(...tiles.map((t) => t.scale));
    const children = new Map();
    let ready = [];
    for (const t of tiles) {
      if (t.scale === top) { ready.push(t); continue; }
      const up = B ** (t.scale + 1);
      const parent = byKey.get(`${t.scale + 1}:${Math.floor(t._gx / up) * up}:${Math.floor(t._gy / up) * up}`);
      if (!parent) { ready.push(t); continue; }
      if (!children.has(parent)) children.set(parent, []);
      children.get(parent).push(t);
    }
    const order = [];
    const better = (a, b) => (a.scale !== b.scale ? a.scale > b.scale : a.photonDiff > b.photonDiff);
    while (ready.length) {
      let bi = 0;
      for (let i = 1; i < ready.length; i++) if (better(ready[i], ready[bi])) bi = i;
      const t = ready.splice(bi, 1)[0];
      order.push(t);
      for (const c of children.get(t) || []) ready.push(c);
    }
    return order.map(({ region, size, scale, photonDiff }) => ({ region, size, scale, photonDiff }));
  }
  // draws the whole canvas in that order; onTile(tile, result) after each tile
  renderProgressive(onTile) {
    for (const t of this.progressiveTiles()) { const r = this.traceTile(t.region, t.size, t.scale); if (onTile) onTile(t, r); }
  }

  projectToCanvas(xyz) {                                                      // rayphoton.js:1559-1571
    const uvw = this.camera.toUVW(xyz);
    if (uvw[2] >= 0) return undefined;
    const u = (-uvw[0] / uvw[2]) / this.camera.fovScaleWidth, v = (-uvw[1] / uvw[2]) / this.camera.fovScaleHeight;
    return { x: (u + 1) * this.canvasWidth / 2 - 1, y: (-v + 1) * this.canvasHeight / 2 - 1 };
  }
}

// A scene given as a flat description file (tests/golden/scene_*_flat.json and anything written in that layout)
// instead of constructor calls.  Same rendering / emission methods as Scene.
class FlatFileScene extends Scene {
  constructor(ctx, flat, options) {
    super(ctx, options);
    this._flatSource = typeof flat === 'string' ? JSON.parse(require('fs').readFileSync(flat, 'utf8')) : flat;
    this.lights = this._flatSource.lights.map((l) => l.shape);
  }
  flatten() { return S.flattenFromJSON(this._flatSource, this.canvasWidth, this.canvasHeight); }
  initLights() { this._device = null; }                      // the file carries intensities and probabilities
}

// The older tracer, ray.js ("mode A"): no lights, no photons; shapes carry reflectance / specular themselves and its
// four materials are AIR, OPAQUE, GLASS, WATER (ray.js:39-61).  Constants of ray.js:8-12 as defaults.
const WHITTED_DEFAULTS = { mode: 1, maxTraceDist: 100, maxDepth: 20, subSample: 1 };
const WHITTED_MATERIALS = {
  transparent: [true, false, true, true, false, false, false, false], refrIndex: [1.0, undefined, 1.5, 1.33, undefined, undefined, undefined, undefined],
  reflectance: new Array(8).fill(undefined), specular: new Array(8).fill(undefined),
};
class WhittedScene extends FlatFileScene {
  // flat: a flat description whose shapes carry `reflectance` and `specular`, or null to fill `shapes` with this
  // host's primitives (give each shape .reflectance / .specular, cylinders .firstRootOnly = true) and set `camera`
  constructor(ctx, flat, options) {
    super(ctx, flat || { shapes: [], lights: [] }, Object.assign({}, WHITTED_DEFAULTS, options || {}));
    this.materials = WHITTED_MATERIALS;
    this._fromFile = !!flat;
  }
  flatten() { return this._fromFile ? super.flatten() : S.flatten(this); }
  emitManyPhotons() { /* ray.js has no photon pass */ }
  // the four scenes of ray.js:573-762, built with the older tracer's constructors (whitted.js)
  loadPreset(def) {
    const build = W.presets[String(def)];
    if (!build) return;                                                      // (ray.js:760: nothing happens)
    this.lamps = [];
    build(this);
    this._fromFile = false; this._device = null;
  }
}

// Binary PPM (P6) of an rgba8 frame as render() returns it; alpha is dropped.
function writePPM(file, rgba8, width, height) {
  const fs = require('fs');
  const head = Buffer.from(`P6\n${width} ${height}\n255\n`, 'ascii');
  const body = Buffer.alloc(width * height * 3);
  for (let p = 0; p < width * height; p++) { body[3 * p] = rgba8[4 * p]; body[3 * p + 1] = rgba8[4 * p + 1]; body[3 * p + 2] = rgba8[4 * p + 2]; }
  fs.writeFileSync(file, Buffer.concat([head, body]));
}

module.exports = Object.assign({}, S, { Whitted: W, Scene, FlatFileScene, WhittedScene, Ray, Photon, KdTree, DeviceBuffer, PRECISION, SHARD, DEFAULTS, WHITTED_DEFAULTS,
  context, addon, photonsToRecords, recordsToPhotons, writePPM });
