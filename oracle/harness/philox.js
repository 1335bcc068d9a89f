'use strict';
// TEST INFRASTRUCTURE (oracle harness) -- not part of the shipped product path.
//
// Philox4x32-10 counter-based generator (Salmon et al., "Parallel random numbers: as easy
// as 1, 2, 3", SC'11; constants as published in Random123).  The same function is restated
// in oracle/rt_oracle.c and raytracer_amd/csrc/rt_philox.h; all three are pinned by the
// Random123 known-answer vectors below (selfTest()).
//
// Stream convention shared by every side:
//   key     = (seed_lo, seed_hi)
//   counter = (draw >>> 2, streamA, streamB, domain)
//   U       = (word[draw & 3] >>> 8) * 2^-24        (exactly representable in fp32 and fp64)

const M0 = 0xD2511F53, M1 = 0xCD9E8D57, W0 = 0x9E3779B9, W1 = 0xBB67AE85;

function mulhilo(a, b) {            // 32x32 -> [hi, lo], all unsigned
  const al = a & 0xFFFF, ah = a >>> 16, bl = b & 0xFFFF, bh = b >>> 16;
  const ll = al * bl, lh = al * bh, hl = ah * bl, hh = ah * bh;
  const mid = (ll >>> 16) + (lh & 0xFFFF) + (hl & 0xFFFF);
  const lo = ((mid & 0xFFFF) << 16 | (ll & 0xFFFF)) >>> 0;
  const hi = (hh + (lh >>> 16) + (hl >>> 16) + (mid >>> 16)) >>> 0;
  return [hi, lo];
}

function philox4x32_10(c0, c1, c2, c3, k0, k1) {
  for (let r = 0; r < 10; r++) {
    const [hi0, lo0] = mulhilo(M0, c0);
    const [hi1, lo1] = mulhilo(M1, c2);
    const n0 = (hi1 ^ c1 ^ k0) >>> 0, n1 = lo1, n2 = (hi0 ^ c3 ^ k1) >>> 0, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 = (k0 + W0) >>> 0; k1 = (k1 + W1) >>> 0;
  }
  return [c0, c1, c2, c3];
}

const DOMAIN_RENDER = 0, DOMAIN_EMIT_PHASE1 = 1, DOMAIN_EMIT_PHASE2 = 2, DOMAIN_AUX = 3;

class KeyedRng {
  constructor(seed) {
    this.k0 = seed >>> 0; this.k1 = Math.floor(seed / 4294967296) >>> 0;
    this.a = 0; this.b = 0; this.domain = DOMAIN_AUX; this.draw = 0; this.total = 0;
    this.block = null; this.blockIdx = -1;
  }
  setStream(domain, a, b) {
    this.domain = domain >>> 0; this.a = a >>> 0; this.b = (b || 0) >>> 0;
    this.draw = 0; this.blockIdx = -1;
  }
  next() {
    const bi = this.draw >>> 2;
    if (bi !== this.blockIdx) {
      this.block = philox4x32_10(bi, this.a, this.b, this.domain, this.k0, this.k1);
      this.blockIdx = bi;
    }
    const w = this.block[this.draw & 3];
    this.draw++; this.total++;
    return (w >>> 8) / 16777216;
  }
}

function selfTest() {
  const hex = (v) => v.map((x) => ('00000000' + x.toString(16)).slice(-8)).join(' ');
  const kat = [
    [[0, 0, 0, 0], [0, 0], '6627e8d5 e169c58d bc57ac4c 9b00dbd8'],
    [[0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff], [0xffffffff, 0xffffffff], '408f276d 41c83b0e a20bc7c6 6d5451fd'],
    [[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], 'd16cfe09 94fdcceb 5001e420 24126ea1'],
  ];
  for (const [c, k, want] of kat) {
    const got = hex(philox4x32_10(c[0], c[1], c[2], c[3], k[0], k[1]));
    if (got !== want) throw new Error(`philox KAT mismatch: got ${got}, want ${want}`);
  }
  return true;
}

module.exports = { philox4x32_10, KeyedRng, selfTest,
  DOMAIN_RENDER, DOMAIN_EMIT_PHASE1, DOMAIN_EMIT_PHASE2, DOMAIN_AUX };
if (require.main === module) { selfTest(); console.log('philox KAT ok'); }
