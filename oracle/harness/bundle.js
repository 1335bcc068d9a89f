'use strict';
// TEST INFRASTRUCTURE (oracle harness).  Tiny array-bundle writer: <base>.json is a manifest
// {arrays: {name: {dtype, shape, offset, nbytes}}, meta: {...}} and <base>.bin holds the raw
// little-endian data.  tests/golden_io.py reads the same format with numpy.

const fs = require('fs');

const DTYPES = new Map([
  [Float64Array, 'f8'], [Float32Array, 'f4'], [Int32Array, 'i4'], [Uint32Array, 'u4'],
  [Uint8Array, 'u1'], [Int8Array, 'i1'],
]);

function writeBundle(base, arrays, meta) {
  const manifest = { arrays: {}, meta: meta || {} };
  const chunks = [];
  let offset = 0;
  for (const name of Object.keys(arrays)) {
    let a = arrays[name], shape;
    if (!ArrayBuffer.isView(a)) { shape = a.shape; a = a.data; }
    const dtype = DTYPES.get(a.constructor);
    if (!dtype) throw new Error(`unsupported array type for ${name}`);
    shape = shape || [a.length];
    const n = shape.reduce((p, q) => p * q, 1);
    if (n !== a.length) throw new Error(`shape mismatch for ${name}: ${shape} vs ${a.length}`);
    const buf = Buffer.from(a.buffer, a.byteOffset, a.byteLength);
    manifest.arrays[name] = { dtype, shape, offset, nbytes: buf.length };
    chunks.push(buf);
    offset += buf.length;
    const pad = (8 - (offset % 8)) % 8;
    if (pad) { chunks.push(Buffer.alloc(pad)); offset += pad; }
  }
  fs.writeFileSync(base + '.bin', Buffer.concat(chunks));
  fs.writeFileSync(base + '.json', JSON.stringify(manifest));
}

module.exports = { writeBundle };
