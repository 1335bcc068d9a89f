'use strict';
// One scene over several GPUs from Node: one worker thread and one context per device, shared memory between them.
//
//   node raytracer_amd/host/examples/multi_gpu.js --devices 0,1,2,3 [--preset 10] [--size 1920x1080] [--spp 1]
//        [--photons 200000,500000] [--seed 12345] [--batch 1048576] [--out frame.ppm] [--check]
//
// emitManyPhotons (rayphoton.js:1354-1383) is cut into batches of attempts; every worker emits a contiguous sub-range of
// a batch on its GPU (emitShard: the stored photons, 36 bytes + a tag each, in attempt order), the shards meet in a
// SharedArrayBuffer, and every worker applies the reference's list caps to the identical merged batch on its own GPU
// (listsApply).  All workers therefore hold the same two photon lists and build the same two kd-trees, with nothing
// else exchanged.  The frame is split into interleaved bands of 16 rows (bandRows / bandCount / bandIndex); every worker
// traces its bands and writes them into the shared frame.  Lists and pixels do not depend on the number of workers:
// every attempt and every pixel draws from its own keyed random stream.  (A device may be named twice; two workers then
// share that GPU -- how the test of this file runs on a one-GPU machine.)  --check repeats the job on the first device
// alone and compares the lists bit for bit and the frame to one 8-bit level.
const { Worker, isMainThread, parentPort, workerData } = require('worker_threads');
const BAND = 16;

// all workers wait here for each other (word 0 counts arrivals, word 1 is the round)
function barrier(sync, world) {
  const round = Atomics.load(sync, 1);
  if (Atomics.add(sync, 0, 1) === world - 1) {
    Atomics.store(sync, 0, 0);
    Atomics.add(sync, 1, 1);
    Atomics.notify(sync, 1);
  } else {
    while (Atomics.load(sync, 1) === round) {
      if (Atomics.wait(sync, 1, round, 120000) === 'timed-out') throw new Error('a worker did not reach the barrier (did it fail?)');
    }
  }
}

function buildScene(rt, job, emit) {
  const scene = new rt.Scene({ width: job.width, height: job.height }, Object.assign({ listFormat: rt.SHARD.PACKED }, job.options));
  const quiet = console.log; console.log = () => {};
  try { scene.loadPreset(job.preset, emit); } finally { console.log = quiet; }
  return scene;
}

function workerMain() {
  const job = workerData, rank = job.rank, world = job.world;
  const rt = require('..');
  const native = rt.addon();
  const ctx = rt.context(job.device);
  const scene = buildScene(rt, job, false);
  const dev = scene._scene(), params = scene.params;
  const sync = new Int32Array(job.sync), counts = new Uint32Array(job.counts), shared = new Uint32Array(job.shards);
  const lists = native.listsCreate(ctx, params, rt.SHARD.PACKED);
  const W1 = native.listsRecordWords(lists) + 1;                  // words per shard record: payload + tag
  const S = job.stride;                                           // records of shared memory per worker
  const local = new Uint32Array(S * W1), merged = new Uint32Array(world * S * W1);
  const ND = params.numPhotonsDiffuse, NC = params.numPhotonsCaustic;
  let nd = 0, nc = 0, capped = false, rounds = 0, segments = 0;
  const attempts = { 1: 0, 2: 0 };
  const t0 = Date.now();
  for (const phase of [1, 2]) {
    let a = 0;
    const limit = 100 * (phase === 1 ? ND : NC);                  // rayphoton.js:1365, :1377
    while (phase === 1 ? nd < ND : nc < NC) {
      // about four times the attempts the list still needs (an attempt stores at most one photon, most store none)
      const need = phase === 1 ? ND - nd : NC - nc;
      const b = Math.min(job.batch, Math.max(4 * need, 65536 * world), limit + 1 - a);
      const lo = a + Math.floor(b * rank / world), hi = a + Math.floor(b * (rank + 1) / world);   // sub-ranges in rank order = attempt order
      const sh = native.emitShard(dev, params, phase, lo, hi, rt.SHARD.PACKED, a, local);
      segments += sh.stats.segments;
      shared.set(local.subarray(0, sh.n * W1), rank * S * W1);
      Atomics.store(counts, rank, sh.n);
      barrier(sync, world);                                       // every shard of the round is in shared memory
      const cnt = new Uint32Array(world);
      let m = 1;
      for (let k = 0; k < world; k++) { cnt[k] = Atomics.load(counts, k); m = Math.max(m, cnt[k]); }
      for (let k = 0; k < world; k++) merged.set(shared.subarray(k * S * W1, (k * S + cnt[k]) * W1), k * m * W1);
      barrier(sync, world);                                       // ... and has been read by everybody: the next round may overwrite it
      const info = native.listsApply(lists, phase, a, merged, cnt, m);
      nd = info.nDiffuse; nc = info.nCaustic;
      a = (info.phaseEnded ? info.lastAttempt : a + b - 1) + 1;
      rounds++;
      if (a > limit) { capped = true; break; }
    }
    attempts[phase] = a;
    if (capped) break;
  }
  const tEmit = Date.now();
  const quiet = console.log; console.log = () => {};
  if (nd) scene.photonKdTDiffuse = new rt.KdTree({ lists, which: 0, n: nd });
  if (nc && !capped) scene.photonKdTCaustic = new rt.KdTree({ lists, which: 1, n: nc });
  console.log = quiet;
  const tBuild = Date.now();
  // this worker's bands of the frame, then into the shared frame row by row
  const band = world > 1 ? { bandRows: BAND, bandCount: world, bandIndex: rank } : {};
  const part = scene.render(0, 0, job.width, job.height, band);
  const frame = new Uint8Array(job.frame), rowBytes = job.width * 4;
  let src = 0;
  for (let y = 0; y < job.height; y++) {
    if (world > 1 && Math.floor(y / BAND) % world !== rank) continue;
    frame.set(part.rgba8.subarray(src, src + rowBytes), y * rowBytes);
    src += rowBytes;
  }
  if (rank === 0 && job.listD) {                                  // the lists as they lie on this GPU, for --check
    const d = native.listsRecords(lists, 0), c = native.listsRecords(lists, 1);
    new Float32Array(job.listD).set(d.records); new Float32Array(job.listC).set(c.records);
  }
  parentPort.postMessage({ rank, nd, nc, attempts, capped, rounds, segments, rays: part.stats.rays,
    emitMs: tEmit - t0, buildMs: tBuild - tEmit, renderMs: Date.now() - tBuild });
  native.shutdown(ctx);
}

// run({devices, preset, width, height, options, batch, check}) -> Promise of {frame: Uint8Array, workers: [...], check}
function run(opts) {
  const devices = opts.devices || [0], world = devices.length;
  const options = Object.assign({ subSample: 1, seed: 12345, numPhotonsDiffuse: 20000, numPhotonsCaustic: 50000 }, opts.options || {});
  const batch = opts.batch || (1 << 20);
  const job = { world, preset: opts.preset === undefined ? 10 : opts.preset, width: opts.width || 640, height: opts.height || 360, options, batch };
  job.stride = Math.ceil(Math.max(batch, 65536 * world) / world) + 1;
  job.sync = new SharedArrayBuffer(8);
  job.counts = new SharedArrayBuffer(4 * world);
  job.shards = new SharedArrayBuffer(world * job.stride * 10 * 4);             // packed records: 9 words + tag
  job.frame = new SharedArrayBuffer(job.width * job.height * 4);
  if (opts.check) {
    job.listD = new SharedArrayBuffer(options.numPhotonsDiffuse * 36);
    job.listC = new SharedArrayBuffer(options.numPhotonsCaustic * 36);
  }
  const workers = devices.map((device, rank) => new Promise((resolve, reject) => {
    const w = new Worker(__filename, { workerData: Object.assign({ rank, device }, job) });
    let result = null;
    w.on('message', (m) => { result = m; });
    w.on('error', reject);
    w.on('exit', (code) => (result && code === 0 ? resolve(result) : reject(new Error(`worker ${rank} ended with code ${code}`))));
  }));
  return Promise.all(workers).then((results) => {
    const out = { frame: new Uint8Array(job.frame), workers: results, width: job.width, height: job.height };
    if (!opts.check) return out;
    // the same job on the first device alone
    const rt = require('..');
    const native = rt.addon();
    rt.context(devices[0]);
    const one = buildScene(rt, Object.assign({}, job, { device: devices[0] }), true);
    const d = native.listsRecords(one.photonLists, 0), c = native.listsRecords(one.photonLists, 1);
    const img = one.render(0, 0, job.width, job.height);
    const sameBits = (a, b, n) => { const x = new Uint32Array(a.buffer, a.byteOffset, n), y = new Uint32Array(b.buffer, b.byteOffset, n); for (let i = 0; i < n; i++) if (x[i] !== y[i]) return false; return true; };
    const md = new Float32Array(job.listD), mc = new Float32Array(job.listC), r0 = results[0];
    // (fp32 frames sum the photons of a gather in an order that depends on which gathers are queued together: a channel
    // may differ by one 8-bit level between the split and the whole frame; anything more is a failure)
    let badPixels = 0, offByOne = 0;
    for (let i = 0; i < out.frame.length; i++) {
      const d8 = Math.abs(out.frame[i] - img.rgba8[i]);
      if (d8 > 1) badPixels++; else if (d8 === 1) offByOne++;
    }
    out.check = {
      counts: r0.nd === d.n && r0.nc === c.n && r0.attempts[1] === one.emitStats.attemptsPhase1 && r0.attempts[2] === one.emitStats.attemptsPhase2,
      lists: r0.nd === d.n && r0.nc === c.n && sameBits(md, d.records, d.n * 9) && sameBits(mc, c.records, c.n * 9),
      agree: results.every((r) => r.nd === r0.nd && r.nc === r0.nc && r.attempts[1] === r0.attempts[1] && r.attempts[2] === r0.attempts[2]),
      badPixels, offByOne,
    };
    out.check.ok = out.check.counts && out.check.lists && out.check.agree && badPixels === 0;
    return out;
  });
}

if (!isMainThread) workerMain();
else if (require.main === module) {
  const arg = (name, dflt) => { const i = process.argv.indexOf('--' + name); return i > 0 ? process.argv[i + 1] : dflt; };
  const [width, height] = arg('size', '1920x1080').split('x').map(Number);
  const [nd, nc] = arg('photons', '200000,500000').split(',').map(Number);
  const preset = arg('preset', '10');
  const t0 = Date.now();
  run({ devices: arg('devices', '0').split(',').map(Number), preset: /^\d+$/.test(preset) ? Number(preset) : preset, width, height,
    options: { subSample: Number(arg('spp', '1')), seed: Number(arg('seed', '12345')), numPhotonsDiffuse: nd, numPhotonsCaustic: nc },
    batch: Number(arg('batch', String(1 << 20))), check: process.argv.includes('--check') }).then((out) => {
    for (const r of out.workers) console.log(`worker ${r.rank}: ${r.nd} diffuse / ${r.nc} caustic photons after ${r.attempts[1]} + ${r.attempts[2]} attempts in ${r.rounds} rounds, ` +
      `emit ${r.emitMs} ms, trees ${r.buildMs} ms, bands ${r.renderMs} ms (${r.rays} rays)`);
    console.log(`${width}x${height} on ${out.workers.length} worker(s) in ${Date.now() - t0} ms`);
    if (out.check) console.log(`check against one GPU: ${out.check.ok ? 'PASS' : 'FAIL'} ${JSON.stringify(out.check)}`);
    if (arg('out')) require('..').writePPM(arg('out'), out.frame, width, height);
    process.exit(out.check && !out.check.ok ? 1 : 0);
  }).catch((e) => { console.error(e); process.exit(1); });
}

module.exports = { run };
