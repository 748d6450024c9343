#!/usr/bin/env python3
"""Generate the polynomial coefficients used by include/rpt_strict_math.h.

Run here (needs mpmath); the output is pasted into the header.  The header's
functions are the project's bit-reproducible stand-in for the platform libm
that Rust's f32::{sin,cos,tan,powf,log2} defer to
(reference call sites: rust-pathtracer/src/tracer.rs:181,239,248-251,266-267,
329-330,401; scene.rs:33; camera/pinhole.rs:43).
"""
import mpmath as mp
import struct

mp.mp.dps = 60

def f64hex(x):
    d = float(x)
    return "%s /* %s */" % (d.hex(), repr(d))

def f32(x):
    return struct.unpack("<f", struct.pack("<f", float(x)))[0]

# ---- log2(1+f)/f on f in [sqrt(1/2)-1, sqrt(2)-1] --------------------------
a = mp.sqrt(mp.mpf(1) / 2) - 1
b = mp.sqrt(2) - 1
g = lambda f: (mp.log(1 + f, 2) / f) if f != 0 else 1 / mp.log(2)
for deg in (12, 13, 14):
    coeffs, err = mp.chebyfit(g, [a, b], deg + 1, error=True)
    # coeffs: highest degree first
    cd = [mp.mpf(float(c)) for c in coeffs]
    # measured error of the double-rounded polynomial for log2(1+f) = f*P(f)
    worst = mp.mpf(0)
    for i in range(4001):
        f = a + (b - a) * i / 4000
        p = mp.polyval(cd, f) * f
        e = abs(p - mp.log(1 + f, 2))
        worst = max(worst, e)
    print("log2 deg", deg, "fit err", mp.nstr(err, 5), "abs err of f*P(f):", mp.nstr(worst, 5))
    if deg == 13:
        print("LOG2 coefficients (c0..c13, ascending):")
        for c in reversed(coeffs):
            print("   ", f64hex(c) + ",")

# ---- 2^r on r in [-0.5, 0.5] ------------------------------------------------
h = lambda r: mp.mpf(2) ** r
for deg in (8, 9, 10):
    coeffs, err = mp.chebyfit(h, [-0.5, 0.5], deg + 1, error=True)
    print("exp2 deg", deg, "fit err", mp.nstr(err, 5))
    if deg == 9:
        print("EXP2 coefficients (c0..c9, ascending):")
        for c in reversed(coeffs):
            print("   ", f64hex(c) + ",")

# ---- pi/2 split for Cody-Waite (f32) ---------------------------------------
pio2 = mp.pi / 2
# hi has its last mantissa bit cleared so small multiples are exact even without fma
hi = f32(pio2)
hi_bits = struct.unpack("<I", struct.pack("<f", hi))[0] & 0xFFFFF000
hi = struct.unpack("<f", struct.pack("<I", hi_bits))[0]
mid = f32(pio2 - mp.mpf(hi))
mid_bits = struct.unpack("<I", struct.pack("<f", mid))[0] & 0xFFFFF000
mid = struct.unpack("<f", struct.pack("<I", mid_bits))[0]
lo = f32(pio2 - mp.mpf(hi) - mp.mpf(mid))
print("PIO2_HI  = %s (%r)" % (float(hi).hex(), hi))
print("PIO2_MID = %s (%r)" % (float(mid).hex(), mid))
print("PIO2_LO  = %s (%r)" % (float(lo).hex(), lo))
print("residual:", mp.nstr(pio2 - mp.mpf(hi) - mp.mpf(mid) - mp.mpf(lo), 5))
print("2/pi f32 = %s" % float(f32(2 / mp.pi)).hex())

# ---- sin / cos minimax on [-pi/4, pi/4] in f32 ------------------------------
# sin(r) = r + r^3 * S(r^2);   cos(r) = 1 + r^2 * C(r^2)
q = (mp.pi / 4) ** 2 * mp.mpf("1.02")
S = lambda u: ((mp.sin(mp.sqrt(u)) / mp.sqrt(u) - 1) / u) if u != 0 else mp.mpf(-1) / 6
C = lambda u: ((mp.cos(mp.sqrt(u)) - 1) / u) if u != 0 else mp.mpf(-1) / 2
for name, fn, deg in (("SIN", S, 3), ("COS", C, 4)):
    coeffs, err = mp.chebyfit(fn, [mp.mpf(0), q], deg + 1, error=True)
    print(name, "deg", deg, "fit err", mp.nstr(err, 5))
    for c in reversed(coeffs):
        print("    %s /* %r */," % (float(f32(c)).hex(), f32(c)))
