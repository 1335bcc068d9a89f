'use strict';
// Keyed random numbers of the host: Philox4x32-10 (Salmon et al., SC'11; Random123 constants) under the stream
// convention the device code uses (raytracer_amd/csrc/rt_philox.h):
//   key = (seed_lo, seed_hi), counter = (draw >> 2, a, b, domain), U = (word[draw & 3] >> 8) * 2^-24.
// The reference draws from Math.random(); here every consumer names its stream, so a scene built from random numbers
// (presets 5, 9 and 'giacometti') is the same scene on every run and on every rank.

const DOMAIN = { RENDER: 0, EMIT_PHASE1: 1, EMIT_PHASE2: 2, AUX: 3 };

// 32 x 32 -> 64 bit product as [high, low] words, from 16-bit limbs (every partial product stays below 2^53)
function mul32(a, b) {
  const a0 = a & 0xFFFF, a1 = a >>> 16, b0 = b & 0xFFFF, b1 = b >>> 16;
  const p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
  const carry = (p00 >>> 16) + (p01 & 0xFFFF) + (p10 & 0xFFFF);
  return [(p11 + (p01 >>> 16) + (p10 >>> 16) + (carry >>> 16)) >>> 0, (((carry & 0xFFFF) << 16) | (p00 & 0xFFFF)) >>> 0];
}

function philox(counter, key) {
  let [c0, c1, c2, c3] = counter, [k0, k1] = key;
  for (let round = 0; round < 10; round++) {
    const [h0, l0] = mul32(0xD2511F53, c0), [h1, l1] = mul32(0xCD9E8D57, c2);
    [c0, c1, c2, c3] = [(h1 ^ c1 ^ k0) >>> 0, l1, (h0 ^ c3 ^ k1) >>> 0, l0];
    k0 = (k0 + 0x9E3779B9) >>> 0; k1 = (k1 + 0xBB67AE85) >>> 0;
  }
  return [c0, c1, c2, c3];
}

class Stream {
  constructor(seed, domain, a, b) {
    this.key = [seed >>> 0, Math.floor(seed / 4294967296) >>> 0];
    this.domain = domain >>> 0; this.a = a >>> 0; this.b = (b || 0) >>> 0;
    this.draw = 0; this.words = null; this.block = -1;
  }
  next() {                                                  // the next U in [0, 1)
    const block = this.draw >>> 2;
    if (block !== this.block) { this.words = philox([block, this.a, this.b, this.domain], this.key); this.block = block; }
    return (this.words[this.draw++ & 3] >>> 8) / 16777216;
  }
}

module.exports = { DOMAIN, Stream, philox };
