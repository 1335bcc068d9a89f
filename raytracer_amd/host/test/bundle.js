'use strict';
// reads the array bundles under tests/golden (<name>.json manifest + <name>.bin)
const fs = require('fs');
const path = require('path');
const GOLDEN = path.resolve(__dirname, '../../../tests/golden');
const CT = { f8: Float64Array, f4: Float32Array, i4: Int32Array, u4: Uint32Array, u1: Uint8Array, i1: Int8Array };
function loadBundle(name) {
  const man = JSON.parse(fs.readFileSync(path.join(GOLDEN, name + '.json'), 'utf8'));
  const raw = fs.readFileSync(path.join(GOLDEN, name + '.bin'));
  const buf = raw.buffer.slice(raw.byteOffset, raw.byteOffset + raw.byteLength);
  const out = { meta: man.meta };
  for (const [k, a] of Object.entries(man.arrays)) {
    const T = CT[a.dtype];
    out[k] = new T(buf, a.offset, a.nbytes / T.BYTES_PER_ELEMENT);
  }
  return out;
}
module.exports = { loadBundle, GOLDEN };
