#!/usr/bin/env python3
"""A rigorous forward-error bound for the fp32 OFDM transform and differential product of the parity guard's PROVEN level
(dabtools_amd/csrc/device_types.hpp: kGuardCProven, kGuardProdProven; DESIGN.md section 3 carries the derivation in prose).

The transform (fft_core.hpp + k_fused.hip / k_fft.hip; both kernels run the same butterflies, factors and roundings):
    2048 = 8 (stage A, index a) x 8 (B, index b) x 8 (C, index c) x 4 (D, index d), decimation in frequency,
    dft8 = three levels of complex additions, the odd half turned by exp(-i pi/4), exp(-3i pi/4) between levels one and two,
    a table factor (cmul) on outputs 1..7 of stages A, B, C; dft4 = two levels of additions, no factor.
Arithmetic model (IEEE binary32, round to nearest, u = 2^-24; no overflow: |values| < 2^19; no underflow: see note at the end):
    (R1) z = fl(a +- b), complex:                |z - (a +- b)| <= u |a +- b|          (add_mi / sub_mi: the factor +-1 of the fused multiply-add is exact)
    (R2) z = the turn by exp(-i pi/4) etc., written as fma(d.yx, (h, -h), fl(d h)) with h^ = fl(sqrt(1/2)):
                                                 |z - d exp(..)| <= (h^ + 1 + |h^ - h| / (h u)) u |d| (1 + u)   = 1.9941 u |d|
    (R3) z = cmul(a, w^) = (fma(a.x, w.x, fl(-a.y w.y)), fma(a.x, w.y, fl(a.y w.x))), w^ = (float(cos), float(-sin)) of double values:
         |w^ - w| <= sqrt(2) 2^-25 = u / sqrt(2);  |z - a w| <= (1 + 1 + 1 / sqrt(2)) u |a| (1 + 2u)             = 2.7072 u |a|
Propagation.  With F_s the exact butterfly matrices and T_s the exact diagonal factors, DFT = F_D T_C F_C T_B F_B T_A F_A.  Let v_s be the COMPUTED
vector after stage s and eps_s = v_s - T_s F_s v_(s-1) the error stage s adds to its (computed) input.  Then, exactly,
    v_D - DFT x = eps_D + F_D eps_C + F_D T_C F_C eps_B + F_D T_C F_C T_B F_B eps_A ,
and row k of each of these exact operators has entries of modulus 1 on the cone of bin k (256, 32, 4, 1 elements) and 0 elsewhere:
    |e_k| <= sum_(cone_A(k)) |eps_A| + sum_(cone_B(k)) |eps_B| + sum_(cone_C(k)) |eps_C| + |eps_D(k)| .
Inside a radix-8 block with computed inputs v_0..v_7, S = sum |v_i|, S_o = |v_1| + |v_3| + |v_5| + |v_7| (the same exact-identity argument, one level
at a time; every intermediate result is at most S (1 + 6u)):
    general block (stages B, C):   |eps(q)| <= u (1 + 8u) [3 S (three levels of additions) + 1.9941 S_o (q odd) + 2.7072 S (q > 0)]   <= 7.7013 u S (1 + 8u)
    stage A (inputs are integers of magnitude <= 128: every addition that involves no turned value is exact):
                                   |eps(q)| <= u (1 + 8u) [1.9941 S_o (the two turns) + S_o (a5 +- a7) + S (last addition, q odd) + 2.7072 S (q > 0)] <= 6.7013 u S
    stage D (dft4):                |eps(q)| <= u (1 + 2u) 2 S4 .
Cone sums: sum over the blocks of cone_s(k) of S = the l1 norm of the cone's input values; by Cauchy-Schwarz and |partial DFT sum of m terms|^2 <= m sum |terms|^2
it is <= sqrt(2048) |x|_2 (1 + 8u) for every stage (A: |x|_1 <= sqrt(2048) |x|_2; B: 16 x sqrt(8); C: sqrt(32) x 8; D: 2 x sqrt(512)).
    |e_k| <= u sqrt(2048) (kA + kB + kC + kD) (1 + 1e-5) |x|_2 .
The k's above are the worst outputs' (q odd).  Bin k = a' + 8 b' + 64 c' + 512 d' is output a' of its stage-A blocks, b' of its stage-B blocks, c' of its stage-C blocks,
so ITS bound takes kA(a') = 0 | 2.7072 (even) | 6.7013 (odd), kB(b'), kC(c') = 3 | 5.7072 | 7.7013: bin_bound(k) below, 0.33 .. 1 of the worst bin's and 0.79 of it on
average over the carriers' bins -- the proven guard level lists per bin (device_types.hpp: guard_bin_scale), and the model check below holds every bin against its own bound.
The bound is attained up to its constant by a single tone (all partial sums towards its bin add coherently); for an OFDM symbol of 1536 carriers the partial
sums add incoherently and the measured errors are ~ 80 x smaller -- which is why the MEASURED level's band is 13 x narrower.

Differential product (device_types.hpp: diff_re / diff_im = fma(c.x, p.x, fl(c.y p.y)) etc.) on the computed bins c^, p^:
    |fl(..) - Re / Im(c^ conj(p^))| <= u |c.y p.y| (1 + u) + u |Re / Im| <= 2 u (1 + u) |c^|_2 |p^|_2 ,
    |Re / Im(c^ conj p^) - Re / Im(c conj p)| <= |e_c| |p^| + |c^| |e_p| + |e_c| |e_p| ;
the guard's threshold (guard_threshold) is |c^|_1 dp + |p^|_1 dc + prod |c^|_1 |p^|_1 + dc dp with dc, dp >= the bins' error bounds and |.|_1 >= |.|_2.

Underflow: the smallest non-zero table entry is |float(cos(pi/2))| = 6.1e-17, the smallest non-zero value an addition can leave is one unit in the last place
of a value >= 2^-24 ...: products stay above 1e-30 >> 2^-126, so (R1)-(R3) hold without a denormal term (and gfx950 keeps fp32 denormals).

The second half of this script checks the algebra numerically: an fp32 model of the very same operation order (numpy; fused multiply-adds modelled as the
fp64 result rounded once more to fp32) against fp64 on adversarial inputs (single tones at odd bins, clipped tones, two-tone cancellations, random OFDM) --
`worst_model_error / bound` must stay <= 1 and shows how far from attained the bound is.  The GPU's own arithmetic is audited by tools/decision_audit.py."""
import json
import math
import sys

import numpy as np

U = 2.0 ** -24
H = math.sqrt(0.5)
HF = float(np.float32(H))


def constants():
    turn = (HF + 1.0 + abs(HF - H) / (H * U)) * (1 + U)          # (R2)
    cm = (2.0 + 1.0 / math.sqrt(2.0)) * (1 + 2 * U)              # (R3)
    hi = 1 + 8 * U
    # general radix-8 block, worst output (q odd): three addition levels (each <= S), the turned pair, the table factor
    k_general = (3.0 + turn + cm) * hi
    # stage A: additions of integers are exact.  q odd: the two turns (<= turn * S_o), b5 / d57 = a5 +- a7 (one rounding, |.| <= S_o), the last addition
    # (<= S), the table factor (<= cm * S); S_o <= S.  q even: only the table factor.
    k_a = (turn + 1.0 + 1.0 + cm) * hi
    k_d = 2.0 * (1 + 2 * U)
    c_bin = U * math.sqrt(2048.0) * (k_a + 2 * k_general + k_d) * (1 + 1e-5)
    c_prod = 2 * U * (1 + U)
    # per output index q of a stage (the bin's digits: k = a' + 8 b' + 64 c' + 512 d'): q = 0 takes no table factor, an even q no turned value
    per_q_a = [0.0 if q == 0 else ((turn + 2.0 + cm) * hi if q & 1 else cm * hi) for q in range(8)]
    per_q_bc = [3.0 * hi if q == 0 else ((3.0 + turn + cm) * hi if q & 1 else (3.0 + cm) * hi) for q in range(8)]
    return {"u": U, "turn": turn, "cmul": cm, "kappa_A": k_a, "kappa_B": k_general, "kappa_C": k_general, "kappa_D": k_d, "bin_bound": c_bin, "prod_bound": c_prod,
            "kappa_A_by_digit": per_q_a, "kappa_BC_by_digit": per_q_bc}


def bin_bound(k, c=None):
    """the bound for raw bin k by itself: its stage terms by its index digits (<= constants()["bin_bound"], the worst bin's)"""
    c = c or constants()
    kap = c["kappa_A_by_digit"][k & 7] + c["kappa_BC_by_digit"][(k >> 3) & 7] + c["kappa_BC_by_digit"][(k >> 6) & 7] + c["kappa_D"]
    return U * math.sqrt(2048.0) * kap * (1 + 1e-5)


# ---- fp32 model of the kernels' operation order -----------------------------------------------------------------------------
f32 = np.float32


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


class C32:
    """vectors of complex numbers held as two float32 arrays; every operation rounds where the kernels round"""

    def __init__(self, x, y):
        self.x, self.y = x.astype(f32), y.astype(f32)

    def __add__(self, o): return C32(self.x + o.x, self.y + o.y)
    def __sub__(self, o): return C32(self.x - o.x, self.y - o.y)
    def add_mi(self, d): return C32(self.x + d.y, self.y - d.x)      # b + (-i) d
    def sub_mi(self, d): return C32(self.x - d.y, self.y + d.x)
    def z(self): return self.x.astype(np.float64) + 1j * self.y.astype(np.float64)


def turn45(d, sign):
    h = f32(HF)
    if sign > 0:    # * exp(-i pi/4): h (x + y, y - x) = fma(d.yx, (h, -h), d * h)
        return C32(fma32(d.y, np.full_like(d.y, h), d.x * h), fma32(d.x, np.full_like(d.x, -h), d.y * h))
    return C32(fma32(d.y, np.full_like(d.y, h), d.x * -h), fma32(d.x, np.full_like(d.x, -h), d.y * -h))   # * exp(-3i pi/4)


def cmul32(a, wx, wy):
    tx, ty = -(a.y * wy), a.y * wx          # v_pk_mul with neg_lo: (-a.y b.y, a.y b.x), one rounding each
    return C32(fma32(a.x, wx, tx), fma32(a.x, wy, ty))


def dft8(v):
    a0, a4 = v[0] + v[4], v[0] - v[4]
    a1, d5 = v[1] + v[5], v[1] - v[5]
    a2, a6 = v[2] + v[6], v[2] - v[6]
    a3, d7 = v[3] + v[7], v[3] - v[7]
    a5, a7 = turn45(d5, +1), turn45(d7, -1)
    b0, b2, b1, d13 = a0 + a2, a0 - a2, a1 + a3, a1 - a3
    b4, b6, b5, d57 = a4.add_mi(a6), a4.sub_mi(a6), a5 + a7, a5 - a7
    return [b0 + b1, b4 + b5, b2.add_mi(d13), b6.add_mi(d57), b0 - b1, b4 - b5, b2.sub_mi(d13), b6.sub_mi(d57)]


def dft4(x):
    d0, d2, d1, d13 = x[0] + x[2], x[0] - x[2], x[1] + x[3], x[1] - x[3]
    return [d0 + d1, d2.add_mi(d13), d0 - d1, d2.sub_mi(d13)]


def twf():
    k = np.arange(2048)
    a = 2 * np.pi * k / 2048
    return np.cos(a).astype(f32), (-np.sin(a)).astype(f32)


def fft2048_model(x):
    """x: complex128 array of 2048 integer-valued samples -> fp32-model bins (complex128 view of the float32 results), natural order"""
    wx, wy = twf()
    X = C32(x.real, x.imag)
    # stage A: n = 256 a + t, t = 0..255: outputs P[a'][t] = W_2048^(a' t) sum_a x[256 a + t] W_8^(a a')
    t = np.arange(256)
    v = [C32(X.x[256 * a + t], X.y[256 * a + t]) for a in range(8)]
    v = dft8(v)
    P = [v[0]] + [cmul32(v[q], wx[(t * q) % 2048], wy[(t * q) % 2048]) for q in range(1, 8)]
    # stage B: t = 32 b + t1 (t1 = 4 c + d): Q[a'][b'][t1] = W_256^(b' t1) sum_b P[a'][32 b + t1] W_8^(b b')
    t1 = np.arange(32)
    Q = []
    for q in range(8):
        v = dft8([C32(P[q].x[32 * b + t1], P[q].y[32 * b + t1]) for b in range(8)])
        Q.append([v[0]] + [cmul32(v[q2], wx[(8 * t1 * q2) % 2048], wy[(8 * t1 * q2) % 2048]) for q2 in range(1, 8)])
    # stage C: t1 = 4 c + d: R[a'][b'][c'][d] = W_32^(c' d) sum_c Q[a'][b'][4 c + d] W_8^(c c');  stage D: DFT-4 over d; bin = a' + 8 b' + 64 c' + 512 d'
    out = np.zeros(2048, dtype=np.complex128)
    d = np.arange(4)
    for q in range(8):
        for q2 in range(8):
            v = dft8([C32(Q[q][q2].x[4 * c + d], Q[q][q2].y[4 * c + d]) for c in range(8)])
            R = [v[0]] + [cmul32(v[q3], wx[(64 * d * q3) % 2048], wy[(64 * d * q3) % 2048]) for q3 in range(1, 8)]
            for q3 in range(8):
                y = dft4([C32(R[q3].x[i:i + 1], R[q3].y[i:i + 1]) for i in range(4)])
                for k3 in range(4):
                    out[q + 8 * q2 + 64 * q3 + 512 * k3] = y[k3].z()[0]
    return out


def adversarial_inputs(rng):
    n = np.arange(2048)
    yield "zeros", np.zeros(2048, complex)
    yield "constant 127+127i", np.full(2048, 127 + 127j)
    for k in (1, 73, 585, 767, 1281, 1535, 2047, 511 + 512, 512, 1536, 520, 576, 8, 64, 2, 16, 1288):      # (the second half: bins whose own bound is small -- even or zero index digits)
        tone = 127 * np.exp(2j * np.pi * k * n / 2048)
        yield "tone %d, rounded to int8" % k, np.round(tone.real) + 1j * np.round(tone.imag)
        yield "tone %d, clipped square wave" % k, 127 * np.sign(np.round(tone.real)) + 127j * np.sign(np.round(tone.imag))
        half = np.where(n < 1024, 1.0, -1.0)
        yield "tone %d, second half negated (coherent partial sums, cancelling last stage)" % k, np.round(tone.real * half) + 1j * np.round(tone.imag * half)
    for trial in range(8):
        ph = np.exp(0.5j * np.pi * rng.integers(0, 4, 2048))
        spec = np.zeros(2048, complex)
        spec[1:769] = ph[1:769]
        spec[1280:] = ph[1280:]
        x = np.fft.ifft(spec) * 2048
        x *= 40 / np.sqrt(np.mean(np.abs(x) ** 2))
        x = np.clip(np.round(x.real), -128, 127) + 1j * np.clip(np.round(x.imag), -128, 127)
        yield "random OFDM symbol %d" % trial, x
    for trial in range(4):
        yield "uniform random int8 %d" % trial, rng.integers(-128, 128, 2048) + 1j * rng.integers(-128, 128, 2048)


def main():
    c = constants()
    rng = np.random.default_rng(6)
    worst, worst_per_bin, rows = 0.0, 0.0, []
    bounds_by_bin = np.array([bin_bound(k, c) for k in range(2048)])
    for name, x in adversarial_inputs(rng):
        norm = float(np.sqrt(np.sum(np.abs(x) ** 2)))
        if norm == 0:
            continue
        got = fft2048_model(x)
        want = np.fft.fft(x.astype(np.complex128))          # fp64: error 1e-16 |x|_2 sqrt-ish, nothing at this scale
        per_bin = np.abs(got - want) / norm
        err = float(np.max(per_bin))
        worst = max(worst, err)
        worst_per_bin = max(worst_per_bin, float(np.max(per_bin / bounds_by_bin)))      # every bin against ITS OWN bound
        rows.append({"input": name, "max_bin_error_over_l2": err, "fraction_of_bound": err / c["bin_bound"], "worst_fraction_of_the_bins_own_bound": float(np.max(per_bin / bounds_by_bin))})
    # differential product: random pairs, fp32 model vs exact
    a = rng.standard_normal((4, 1 << 20)).astype(f32)
    re = fma32(a[0], a[2], a[1] * a[3])
    exact = a[0].astype(np.float64) * a[2] + a[1].astype(np.float64) * a[3]
    l2 = np.hypot(a[0].astype(np.float64), a[1]) * np.hypot(a[2].astype(np.float64), a[3])
    worst_prod = float(np.max(np.abs(re - exact) / l2))
    out = {"what": "rigorous forward-error bound of the fp32 OFDM transform (radix 8.8.8.4) and of the differential product; see the docstring",
           "constants": c, "per_bin_bound": {"mean_over_inband_bins_of_bound_over_worst_bins": float(np.mean([bounds_by_bin[k] for k in list(range(1, 769)) + list(range(1280, 2048))]) / c["bin_bound"]),
                            "smallest_over_worst": float(np.min(bounds_by_bin) / c["bin_bound"])},
           "model_check": {"worst_bin_error_over_l2": worst, "worst_over_bound": worst / c["bin_bound"], "worst_over_the_bins_own_bound": worst_per_bin, "cases": rows,
                                           "worst_product_rounding_over_l2l2": worst_prod, "product_over_bound": worst_prod / c["prod_bound"]}}
    print(json.dumps(out, indent=1))
    return 0 if worst <= c["bin_bound"] and worst_per_bin <= 1.0 and worst_prod <= c["prod_bound"] else 1


if __name__ == "__main__":
    sys.exit(main())
