"use strict";
// Half packing used by Scene.setData.  Bit-for-bit the behaviour of src/utils.ts:16-48, including its quirks
// (SURVEY.md D2): the mantissa is truncated, not rounded; |x| >= 32768 and NaN map to +-Inf; and for f32
// exponents below 81 the reference's `frac >> (113 - exp)` shifts by (113 - exp) mod 32 because that is what
// JavaScript's >> does -- written out explicitly here.
const f32 = new Float32Array(1);
const i32 = new Int32Array(f32.buffer);

function floatToHalf(value) {
    f32[0] = value;               // f64 -> f32, round to nearest even
    const bits = i32[0];
    const sign = (bits >>> 31) & 1;
    const exp = (bits >>> 23) & 0xff;
    let frac = bits & 0x007fffff;
    let outExp;
    if (exp === 0) {
        outExp = 0;               // f32 zero/denormal: mantissa bits pass through truncated
    } else if (exp < 113) {       // below the half normal range
        outExp = 0;
        frac = (frac | 0x00800000) >> ((113 - exp) & 31);
        if (frac & 0x01000000) { outExp = 1; frac = 0; }
    } else if (exp < 142) {
        outExp = exp - 112;
    } else {                      // >= 2^15, Inf, NaN
        outExp = 31;
        frac = 0;
    }
    return (sign << 15) | (outExp << 10) | (frac >> 13);
}

// x in the low 16 bits, y in the high 16 bits (src/utils.ts:46-48)
function packHalf2x16(x, y) { return (floatToHalf(x) | (floatToHalf(y) << 16)) >>> 0; }

module.exports = { floatToHalf, packHalf2x16 };
