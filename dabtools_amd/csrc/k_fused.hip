// k_fused.hip — K2 and K2b in one pass (optional variant, hard decisions): the 2048-point transforms of k_fft.hip with the
// differential demodulation, hard QPSK demap, frequency de-interleave (input_sdr.c:132-162) and the time-de-interleaving
// scatter of demap_kernel<true, 1> applied to the bins while they are still in registers.  The complex64 spectra are never
// written: per TF the kernel reads 311,296 B of IQ and writes 28,800 B of bits, so it is NOT the HBM-roofline stage
// (reported separately from the K2 roofline number, never instead of it).  Output bits are identical to
// ofdm_fft_kernel + demap_kernel: same butterflies, same products, same comparisons.
// This file is compiled three times (Makefile): DABHIP_FUSED_GUARD = 1 (default) -> launch_ofdm_demap_fused_guarded, with the
// parity guard's test in the symbol loop; = 0 -> launch_ofdm_demap_fused_plain, the kernel without any of it (also the AFC
// variant); DABHIP_FUSED_SOFT = 1 -> launch_ofdm_demap_fused_soft: signed 4-bit soft values instead of hard decisions.
#include <hip/hip_runtime.h>

#ifndef DABHIP_FUSED_SOFT
#define DABHIP_FUSED_SOFT 0          // third build: 4-bit soft decisions (launch_ofdm_demap_fused_soft), no guard
#endif
// Fourth build (round 5, test / calibration only): the guarded kernel as it ships -- the same source lines do the arithmetic -- that ALSO leaves every
// bin it holds and every differential product it decides on in global memory (launch_ofdm_demap_fused_audit), so that dabhip_stage_decision_audit can
// hold the kernel the default decode runs against fp64 transforms: error of its bins, error of its products, every decision, every list entry.
#ifndef DABHIP_FUSED_AUDIT
#define DABHIP_FUSED_AUDIT 0
#endif
#ifndef DABHIP_FUSED_GUARD
#define DABHIP_FUSED_GUARD (!DABHIP_FUSED_SOFT)
#endif
#define DABHIP_FUSED_ENERGY (DABHIP_FUSED_GUARD || DABHIP_FUSED_SOFT)   // the symbol energies are needed by both

#include "dab_tables.hpp"
#include "device_types.hpp"
#include "fft_core.hpp"
#include "kernels.hpp"

namespace dabhip {
namespace {

// The transform of k_fft.hip's fft2048_store (same butterflies, same factors, same roundings) with the bins left in registers and only TWO
// trips through LDS and two workgroup barriers per symbol.  Index digits: n = 256 a + 32 b + 4 c + d, bin k = a' + 8 b' + 64 c' + 512 d'.
//   stage A (thread = (b, c, d) = tid):                              DFT-8 over a, factors W_2048^(a' tid)      -> P[a'][tid]        barrier
//   stage B (thread = (a' = tid >> 5, (c, d) = tid & 31)):           DFT-8 over b, factors W_256^(b' (4c + d))   -> Q (below)         barrier
//   stage C (lane = a' | b'_0 << 3 | d << 4, wave = b' >> 1):         DFT-8 over c, factors W_32^(c' d)
//   stage D: DFT-4 over d.  Its four inputs sit in four lanes of the SAME wave that differ in lane bits 4 and 5 only, so the last exchange is
//   a transposition of (lane bits 4, 5) with (register bits 0, 1): eight v_permlane16_swap + eight v_permlane32_swap (gfx950), no LDS, no barrier.
// Afterwards the thread holds x[k3] = bin a' + 8 b' + 64 c' + 512 k3 with c' = (tid >> 4 & 3), y[k3] = the same with c' + 4 (fused_bin below).
// Any bin-to-thread assignment serves this kernel: the decisions go wherever the de-interleaver table says (ak[] in the kernel's prologue).
// With no third trip the two exchange arrays keep their roles for every symbol: P is rewritten by the next symbol's stage A after this symbol's
// second barrier (its last readers, stage B, came before it), Q by the next stage B after the next first barrier (its readers, stage C, before it).
// Round 3: 3 barriers + 3 trips -> 2 + 2.
constexpr int kQStride = 258;                           // [a'] stride of Q in float2: 8 x 32 + 2 (bank arithmetic in fft2048_rest)
constexpr int kPSize = 8 * 256, kQSize = 8 * kQStride;  // float2 each
__device__ __forceinline__ int fused_bin(int tid, int m)   // raw bin of x[m >> 1] (m even) / y[m >> 1] (m odd) of thread tid
{
  return (tid & 7) + 8 * (((tid >> 3) & 1) | ((tid >> 6) << 1)) + 64 * (((tid >> 4) & 3) + 4 * (m & 1)) + 512 * (m >> 1);
}
__device__ __forceinline__ void fft2048_first(float2 (&v)[8], float2* bufP, const Twiddles& tw)
{
  const int tid = threadIdx.x;
  dft8(v);
#pragma unroll
  for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], tw.s1[q - 1]);
#pragma unroll
  for (int q = 0; q < 8; ++q) bufP[q * 256 + tid] = v[q];
  __syncthreads();
}
// The transposition of stage D (see above) on the eight complex registers of a thread: (lane bit 4 <-> register bit 0) by v_permlane16_swap (rows 1, 3
// of the first operand <-> rows 0, 2 of the second), then (lane bit 5 <-> register bit 1) by v_permlane32_swap (upper half of the first <-> lower half of
// the second); tools/ubench/lane_swap_check.hip prints what the two do, valu_rates.hip their rate (8.3 clocks each, a pair of moves' worth).  One asm
// block: the s_nop covers the two wait states a swap needs after a VALU write of its operands (the compiler cannot see inside); every later swap
// reads registers written at least three instructions earlier.  (The builtins of ROCm 7.2 miscompile this pattern: chained swaps of the same
// registers came out with results folded together -- hence the asm.)
__device__ __forceinline__ void transpose_lanes45(float2 (&v)[8])
{
  asm volatile(
      "s_nop 1\n"
      "v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7\n"
      "v_permlane16_swap_b32 %8, %10\n v_permlane16_swap_b32 %9, %11\n v_permlane16_swap_b32 %12, %14\n v_permlane16_swap_b32 %13, %15\n"
      "v_permlane32_swap_b32 %0, %4\n v_permlane32_swap_b32 %1, %5\n v_permlane32_swap_b32 %2, %6\n v_permlane32_swap_b32 %3, %7\n"
      "v_permlane32_swap_b32 %8, %12\n v_permlane32_swap_b32 %9, %13\n v_permlane32_swap_b32 %10, %14\n v_permlane32_swap_b32 %11, %15\n"
      "s_nop 1"
      : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y),
        "+v"(v[4].x), "+v"(v[4].y), "+v"(v[5].x), "+v"(v[5].y), "+v"(v[6].x), "+v"(v[6].y), "+v"(v[7].x), "+v"(v[7].y));
}
__device__ __forceinline__ void fft2048_rest(float2 (&v)[8], const float2* bufP, float2* bufQ, const Twiddles& tw, float2 (&x)[4], float2 (&y)[4])
{
  const int tid = threadIdx.x;
  {
    // lane l of a half-wave takes (c, d) = (l & 7, l >> 3): its P reads are a permutation of 32 consecutive places, and its Q stores come out in lane order
    const int q = tid >> 5, l = tid & 31, t1 = 4 * (l & 7) + (l >> 3);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = bufP[q * 256 + t1 + 32 * r];
    dft8(v);
#pragma unroll
    for (int q2 = 1; q2 < 8; ++q2) v[q2] = cmul(v[q2], tw.s2[q2 - 1]);
    // Q[a'][b' >> 1][d][c][b' & 1]: the pair b' = 2 p, 2 p + 1 of one (c, d) leaves as one 16-byte store, eight consecutive lanes fill a 128-byte row
    float4* dst = reinterpret_cast<float4*>(bufQ) + q * (kQStride / 2) + l;
#pragma unroll
    for (int p = 0; p < 4; ++p) dst[32 * p] = make_float4(v[2 * p].x, v[2 * p].y, v[2 * p + 1].x, v[2 * p + 1].y);
  }
  __syncthreads();
  {
    // place of (a', b', c, d) in float2: 258 a' + 64 (b' >> 1) + 16 d + 2 c + (b' & 1).  A read (fixed c) of the 32 lanes (a', b'_0, d_0) covers the 32 bank
    // pairs 2 a' + b'_0 + 16 d_0, and its 16-lane groups (a', b'_0) 16 different ones modulo 16: free of conflicts as ds_read_b64 and as ds_read2_b64
    const int q = tid & 7, q2 = ((tid >> 3) & 1) | ((tid >> 6) << 1), t2 = (tid >> 4) & 3;
    const float2* src = bufQ + q * kQStride + 64 * (q2 >> 1) + 16 * t2 + (q2 & 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = src[2 * r];
    dft8(v);
#pragma unroll
    for (int q3 = 1; q3 < 8; ++q3) v[q3] = cmul(v[q3], tw.s3[q3]);
  }
  transpose_lanes45(v);                                // register r now holds d = r & 3 of c' = (lane bits 4, 5) + (r & 4)
#pragma unroll
  for (int k3 = 0; k3 < 4; ++k3) { x[k3] = v[k3]; y[k3] = v[4 + k3]; }
  dft4(x[0], x[1], x[2], x[3]);
  dft4(y[0], y[1], y[2], y[3]);
}

// Hard decisions of one symbol against the previous one, for the 8 bins of this thread (input_sdr.c:132-162), written as
// 0/1 BYTES into `dec` at the place where the output word of an MSC symbol wants them: byte 32 t + b = bit b of word t of the
// 16 planes i & 15 of 6 words each (the layout of demap_kernel<true, 1>): bit i of the symbol's 3072 sits at
// A(i) = ((i & 15) 6 + (i >> 9)) 32 + ((i >> 4) & 31), and the second bit of a carrier, i + 1536, at A(i) + 96.  ak[m] = A of
// the first bit of this thread's bin m, fixed for the kernel's life (-1: no carrier).  Every one of the 3072 places is written
// by exactly one carrier, so the array needs no clearing.  The three FIC symbols leave in natural order: their flush gathers
// from the same layout (flush_symbol), so the symbol loop knows only one set of addresses.
// Parity guard (k_parity.hip): a decision whose margin is inside the fp32 error band (dc, dp = error bounds of this and of the
// previous symbol's bins) is listed for the fp64 re-decision that follows this kernel.  The exact per-bin test runs inline and
// straight-line, its outcome OR-ed into one flag per thread; the (rare) thread with a hit repeats the tests to find the bins.
// The test itself is device_types.hpp's guard_threshold; |bin|_1 of both symbols is one instruction each (carrying the previous
// symbol's costs eight registers, which the kernel does not have).
// Measured alternatives from the time the kernel ran three waves per SIMD, all slower (7.0 .. 13 ms against 6.5): the list append
// inline per bin; a symbol-wide threshold in the loop with the exact test in a cold block (inlined, looped over a select chain, as
// a real call, or on an LDS parking area); a wave-level ballot; v_min3 chains; candidate records filtered by a second kernel;
// exponent bytes examined at flush time; a per-thread threshold (largest component of the thread's own bins) with the exact test
// in a cold block; the guard flag as a template parameter; stage-2 twiddles from LDS to free registers.  What did pay (6.5 ->
// 4.9 ms): one decision layout for FIC and MSC symbols (no per-symbol address selects), the flag instead of a per-bin mask, the
// symbol energy summed with DPP adds instead of __shfl_xor steps (1 ms by itself), cmul in two packed instructions, four waves.
// Round 3, at four waves per SIMD: the per-thread threshold again (one threshold from the largest |bin|_1 of the thread's eight bins and of
// the previous symbol's, 22 instructions fewer per symbol and thread, exact test cold): 4.43 .. 4.48 ms against 4.40 .. 4.47 -- the kernel
// is not bound by its instruction count (57 % of the SIMDs' issue time; barriers and LDS round trips are the rest).
struct FusedGuard {
  GuardArgs g;
  unsigned frame;        // index of this TF in the frame list
  uint64_t fast_base;    // address of the frame buffer's byte 0 in the IQ stream for the symbols read in place (fused_symbols<true>); 0 for those read through the view
};
#if DABHIP_FUSED_AUDIT
__device__ float2* g_audit_bins;     // [frame][76][2048] by raw bin
__device__ float2* g_audit_prod;     // [frame][76][2048] by raw bin: (re, im) of cur conj(prev) as the kernel computed them (symbol 0: unused)
__device__ __forceinline__ void audit_dump_bins(const float2 (&x)[4], const float2 (&y)[4], unsigned frame, int sym)
{
  float2* dst = g_audit_bins + (static_cast<size_t>(frame) * kSymbolsPerTf + sym) * 2048;
#pragma unroll
  for (int m = 0; m < 8; ++m) dst[fused_bin(threadIdx.x, m)] = (m & 1) ? y[m >> 1] : x[m >> 1];
}
#endif
#if DABHIP_FUSED_GUARD
__device__ __forceinline__ float l1norm(const float2 v)
{
  float r;
  asm("v_add_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(v.x), "v"(v.y));   // one instruction (the compiler pairs two of these into v_and x 4 + v_pk_add)
  return r;
}
#endif
#if !DABHIP_FUSED_SOFT
// Which of the thread's eight bins carry a carrier is known at compile time for all but two: x[2] (bins 1024 + 0..255) never does, y[1] (bins
// 768 + 0..255) only in thread 0 (bin 768, the last carrier of the upper half), and x[0] of thread 0 is the DC bin -- ak[] says so at run time.
__device__ __forceinline__ bool bin_in_use(int m) { return m != 4; }
#if DABHIP_FUSED_GUARD
// Guarded build.  The decision bytes are the SIGN bits of re and im (one shift each; the flush inverts the words of the second bits): that equals the
// reference's comparisons (input_sdr.c:157-158) for every value except an exact zero -- and a decision with a zero in it is always listed and re-decided in
// fp64 (the test below), as is every decision inside the error band.  The in-loop test is one threshold per thread, made of the largest |bin|_1 among the
// thread's bins of this symbol (maxc) and of the previous one (maxp; guard_threshold grows with both), against the smallest |re|, |im| of the thread: three
// instructions per bin (|bin|_1, max, min3) instead of seven.  Only a thread that trips it repeats the exact per-bin test (rare), so the LIST is the one
// the per-bin rule makes, plus the zeros.  Three tiers since round 6: the thread-wide threshold in the loop; the flat per-bin rule on |.|_1 for the thread that trips
// it; and, at the proven guard level (GuardArgs::per_bin), the bound's per-bin form on |.|_2 (guard_bin_threshold) for each candidate of the flat rule, where the
// entry is about to be appended -- every tier a subset of the one before, so the two hot ones stay as cheap as they were.
__device__ __forceinline__ float bins_l1max(const float2 (&x)[4], const float2 (&y)[4])
{
  float mx = 0.0f;
#pragma unroll
  for (int m = 0; m < 8; ++m)
    if (bin_in_use(m)) mx = fmaxf(mx, l1norm((m & 1) ? y[m >> 1] : x[m >> 1]));
  return mx;
}
__device__ __forceinline__ void decide(const float2 (&x)[4], const float2 (&y)[4], const float2 (&px)[4], const float2 (&py)[4],
                                       const int (&ak)[8], uint8_t* dec, const FusedGuard& guard, int sym, float dc, float dp, float& maxp)
{
  float lo = __builtin_inff(), maxc = 0.0f;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    if (!bin_in_use(m)) continue;
    if (m == 3 && threadIdx.x != 0) continue;               // y[1]: thread 0 only (wave-uniform everywhere else)
    const float2 cur = (m & 1) ? y[m >> 1] : x[m >> 1], prev = (m & 1) ? py[m >> 1] : px[m >> 1];
    const float re = diff_re(cur.x, cur.y, prev.x, prev.y);   // Re(cur conj(prev))
    const float im = diff_im(cur.x, cur.y, prev.x, prev.y);   // -Im(cur conj(prev)), as stored at input_sdr.c:139-143
    if (m == 0 && ak[0] < 0) continue;                      // the DC bin (thread 0)
#if DABHIP_FUSED_AUDIT
    g_audit_prod[(static_cast<size_t>(guard.frame) * kSymbolsPerTf + sym) * 2048 + fused_bin(threadIdx.x, m)] = make_float2(re, im);
#endif
#ifndef DABHIP_PROBE_NOSCATTER
    dec[ak[m]] = static_cast<uint8_t>(__builtin_bit_cast(unsigned, re) >> 31);        // 1 = "not re > 0" (input_sdr.c:157), zeros aside
    dec[ak[m] + 96] = static_cast<uint8_t>(__builtin_bit_cast(unsigned, im) >> 31);   // 0 = "im > 0" (input_sdr.c:158): inverted by the flush
#else
    // measurement build only (wrong output): the decisions without their 12 one-byte LDS stores per thread and symbol -- the most ANY re-arrangement of
    // the de-interleaver scatter could save (tools/build_variant.sh noscatter -DDABHIP_PROBE_NOSCATTER; profiles/r05_fused_scatter_probe.txt)
    asm volatile("" : : "v"(__builtin_bit_cast(unsigned, re) >> 31), "v"(__builtin_bit_cast(unsigned, im) >> 31), "v"(ak[m]));
#endif
    asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(lo) : "v"(re), "v"(im));
    maxc = fmaxf(maxc, l1norm(cur));
  }
#ifndef DABHIP_PROBE_NOTEST
  const bool any = !(lo > guard_threshold(maxc, maxp, dc, dp, guard.g.prod) * 1.000001f);      // (>= every per-bin threshold below, whichever form: |.|_1 >= |.|_2, scale <= 1)
#else
  const bool any = false;
#endif
  maxp = maxc;
  if (any) {                                              // rare: which bins?  bin m of thread t is raw bin fused_bin(t, m)
    unsigned hits = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (!bin_in_use(m)) continue;
      const float2 cur = (m & 1) ? y[m >> 1] : x[m >> 1], prev = (m & 1) ? py[m >> 1] : px[m >> 1];
      const float re = diff_re(cur.x, cur.y, prev.x, prev.y), im = diff_im(cur.x, cur.y, prev.x, prev.y);
      const float least = fminf(fabsf(re), fabsf(im));
      hits |= (ak[m] >= 0 && (least < guard_threshold(l1norm(cur), l1norm(prev), dc, dp, guard.g.prod) || !(least > 0.0f)) ? 1u : 0u) << m;
    }
    while (hits) {
      const unsigned m = __ffs(hits) - 1;
      hits &= hits - 1;
      if (guard.g.delta == nullptr) continue;             // (never in this build: the engine runs the plain kernel when the guard is off)
      if (guard.g.per_bin) {
        // The proven level lists by the per-bin form of the bound (device_types.hpp: guard_bin_threshold -- the bin's own stage terms, |.|_2 norms): a subset of what the
        // flat rule above found.  Rarest path of the kernel: the bin's values are picked by a select chain (m is a run-time value here).
        float cx = 0.0f, cy = 0.0f, qx = 0.0f, qy = 0.0f;
#pragma unroll
        for (unsigned j = 0; j < 8; ++j) {
          const float2 c = (j & 1) ? y[j >> 1] : x[j >> 1], q = (j & 1) ? py[j >> 1] : px[j >> 1];
          cx = m == j ? c.x : cx;
          cy = m == j ? c.y : cy;
          qx = m == j ? q.x : qx;
          qy = m == j ? q.y : qy;
        }
        const float least = fminf(fabsf(diff_re(cx, cy, qx, qy)), fabsf(diff_im(cx, cy, qx, qy)));
        if (!(least < guard_bin_threshold(cx, cy, qx, qy, fused_bin(threadIdx.x, static_cast<int>(m)), dc, dp, guard.g.prod) || !(least > 0.0f))) continue;
      }
      const unsigned at = atomicAdd(guard.g.counter, 1u);
      const unsigned k = static_cast<unsigned>(fused_bin(threadIdx.x, static_cast<int>(m)));
      // (the re-decision reads the two symbols' samples: their address goes along, so that it does not have to walk frame list, descriptor and view first)
      const uint64_t win = guard.fast_base ? guard.fast_base + 2u * static_cast<unsigned>(kNullSamples + kSymSamples * sym + kCpSamples) : 0u;
      if (at < guard.g.cap) guard.g.list[at] = make_uint4(guard.frame, (static_cast<unsigned>(sym) << 16) | k, static_cast<unsigned>(win), static_cast<unsigned>(win >> 32));
    }
  }
}
#else
// Plain build (no guard, also the software-AFC variant): the reference's comparisons as they stand.
__device__ __forceinline__ void decide(const float2 (&x)[4], const float2 (&y)[4], const float2 (&px)[4], const float2 (&py)[4],
                                       const int (&ak)[8], uint8_t* dec, const FusedGuard&, int, float, float, float&)
{
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    if (!bin_in_use(m)) continue;
    const float2 cur = (m & 1) ? y[m >> 1] : x[m >> 1], prev = (m & 1) ? py[m >> 1] : px[m >> 1];
    const float re = diff_re(cur.x, cur.y, prev.x, prev.y);   // Re(cur conj(prev))
    const float im = diff_im(cur.x, cur.y, prev.x, prev.y);   // -Im(cur conj(prev)), as stored at input_sdr.c:139-143
    if (ak[m] >= 0) {                                     // bins without a carrier (DC, guard bands) decide nothing
      dec[ak[m]] = (re > 0.0f) ? 0 : 1;                   // input_sdr.c:157
      dec[ak[m] + 96] = (im > 0.0f) ? 1 : 0;              // input_sdr.c:158
    }
  }
}
#endif
#else
// 4-bit soft values (extension): round(scale x) clamped to +-7, positive = "bit 0"; x = Re for the first
// bit and Im(cur conj(prev)) = -im for the second; scale = soft_scale(dc, dp) (device_types.hpp).  One byte per value, placed
// like the hard decisions' bytes: value i of the symbol's 3072 at S(i) = (i & 15) 192 + (i >> 4) (plane i & 15, place i >> 4),
// the second value of a carrier at S(i) + 96; ak[m] = S of the first value of bin m.  FIC symbols are gathered by their flush.
__device__ __forceinline__ void decide(const float2 (&x)[4], const float2 (&y)[4], const float2 (&px)[4], const float2 (&py)[4],
                                       const int (&ak)[8], uint8_t* dec, float scale)
{
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const float2 cur = (m & 1) ? y[m >> 1] : x[m >> 1], prev = (m & 1) ? py[m >> 1] : px[m >> 1];
    const float re = diff_re(cur.x, cur.y, prev.x, prev.y);   // Re(cur conj(prev))
    const float im = diff_im(cur.x, cur.y, prev.x, prev.y);   // -Im(cur conj(prev)), as stored at input_sdr.c:139-143
    if (ak[m] >= 0) {
      // round-to-nearest-even by the 1.5 x 2^23 trick: the sum's bit pattern is the constant's plus the rounded integer (two's complement), so the clamp to
      // +-7 is ONE v_med3_i32 on the bits and the nibble their low four -- mul, add, med3, and instead of mul, rndne, cvt, med3, and (same values: the sum
      // rounds exactly as v_rndne does for |x| < 2^22, and anything larger lands beyond the clamp either way)
      constexpr float kMagic = 12582912.0f;
      constexpr int kBits = 0x4B400000;
      // (product and sum rounded separately, as round(x) of the rounded product was: no contraction into one fused multiply-add)
      const int b0 = __builtin_bit_cast(int, __fadd_rn(__fmul_rn(re, scale), kMagic)), b1 = __builtin_bit_cast(int, __fadd_rn(__fmul_rn(-im, scale), kMagic));
      dec[ak[m]] = static_cast<uint8_t>(max(kBits - 7, min(kBits + 7, b0)) & 15);
      dec[ak[m] + 96] = static_cast<uint8_t>(max(kBits - 7, min(kBits + 7, b1)) & 15);
    }
  }
}
#endif

// sum of |x_n|^2 over the symbol this workgroup is about to transform: every wave leaves its part in esum[0..3] BEFORE the
// first barrier of the transform; symbol_bound() reads them after it.  All sums are exact (integers < 2^24 in fp32, then int).
// The wave's sum runs on DPP adds (no LDS traffic, no lane-address arithmetic: as __shfl_xor steps it cost 1 ms per 16 k TF).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_take(float v)      // the value of the lane the DPP control selects, 0 where the row mask excludes this lane
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, kRowMask, 0xf, false));
}
__device__ __forceinline__ void symbol_energy_part(const float2 (&v)[8], int* esum)
{
  v2f acc = V(v[0]) * V(v[0]);
#pragma unroll
  for (int r = 1; r < 8; ++r) acc = vfma(V(v[r]), V(v[r]), acc);
  float e = acc.x + acc.y;
  e += dpp_take<0xB1, 0xf>(e);                          // quad_perm [1,0,3,2]
  e += dpp_take<0x4E, 0xf>(e);                          // quad_perm [2,3,0,1]
  e += dpp_take<0x141, 0xf>(e);                         // row_half_mirror
  e += dpp_take<0x140, 0xf>(e);                         // row_mirror: every lane holds the sum of its row of 16
  e += dpp_take<0x142, 0xa>(e);                         // row_bcast:15 into rows 1 and 3
  e += dpp_take<0x143, 0xc>(e);                         // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
  const int total = static_cast<int>(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), 63)));
  esum[threadIdx.x >> 6] = total;                       // every lane writes the same word
}
__device__ __forceinline__ float symbol_bound(const int* esum, float c)
{
  const int4 p = *reinterpret_cast<const int4*>(esum);
  return c * sqrtf(static_cast<float>(p.x + p.y + p.z + p.w));
}

#if !DABHIP_FUSED_SOFT
// 32 decision bytes (0 / 1) -> one output word, by thread t < 96: v_dot4_u32_u8 with the weights 1, 2, 4, 8 / 16, 32, 64, 128 gathers eight of them
// into a byte (the multiply-and-shift this replaces cost four times as much: v_mul_lo_u32 is a quarter-rate instruction).
__device__ __forceinline__ uint32_t pack_word(const uint8_t* dec, int t)
{
  const uint4* p = reinterpret_cast<const uint4*>(dec + 32 * t);
  const uint4 lo = p[0], hi = p[1];
  const uint32_t d[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t w = 0;
#pragma unroll
  for (int k = 3; k >= 0; --k) {
    const uint32_t byte = __builtin_amdgcn_udot4(d[2 * k], 0x08040201u, __builtin_amdgcn_udot4(d[2 * k + 1], 0x80402010u, 0u, false), false);
    w = (w << 8) | byte;
  }
  return w;
}
#endif

// A symbol whose window reaches into the stale tail of the reference's frame buffer (symbol 75 after a negative timing
// shift, frames after a coarse resync): its samples come through the view, one at a time, via an LDS staging row, so that
// this rare path adds no register pressure to the symbol loop.  The row is the decision array that is idle at that moment (the
// one the symbol in flight will fill at its end; its previous content left two barriers ago), used in two halves of 2 KB;
// every thread reads back only what it wrote itself.
__device__ __forceinline__ void load_symbol_view(const uint8_t* stream, const FrameView& view, int sym, uint8_t* idle_dec, unsigned (&raw)[8])
{
  const int tid = threadIdx.x;
  const int start = 2 * (kNullSamples + kSymSamples * sym + kCpSamples);
  uint16_t* stage = reinterpret_cast<uint16_t*>(idle_dec);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      const int p = start + 2 * (tid + 256 * (4 * half + r));
      stage[tid + 256 * r] = static_cast<uint16_t>(view_byte(stream, view, p) | (view_byte(stream, view, p + 1) << 8));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) raw[4 * half + r] = stage[tid + 256 * r];
  }
}

struct FusedOut {
  uint32_t* fic_row;     // this TF's 288 FIC words
  uint32_t* msc;         // logical CIF rows
  int cif_row;
};

#if DABHIP_FUSED_SOFT
// 3072 value bytes -> 384 words of eight nibbles (all 256 threads: words t and t + 256); rows are four times as long.
// MSC symbols: word w is bytes 8 w .. 8 w + 7 (plane w / 24).  FIC symbols want natural order: value b of word w is i = 8 w + b,
// whose byte sits at S(i) = (8 (w & 1) + b) 192 + (w >> 1): eight single-byte reads (three symbols per TF, in part 0 only).
__device__ __forceinline__ void flush_symbol(const uint8_t* dec, int sym, const FusedOut& o)
{
  for (int w = threadIdx.x; w < 384; w += kThreads) {
    if (sym <= 3) {
      const uint8_t* src = dec + 8 * (w & 1) * 192 + (w >> 1);
      uint32_t bits = 0;
#pragma unroll
      for (int b = 0; b < 8; ++b) bits |= static_cast<uint32_t>(src[192 * b]) << (4 * b);
      o.fic_row[(sym - 1) * 384 + w] = bits;
    } else {
      const uint2 x = *reinterpret_cast<const uint2*>(dec + 8 * w);
      const uint32_t y0 = (x.x | (x.x >> 4)) & 0x00ff00ffu, y1 = (x.y | (x.y >> 4)) & 0x00ff00ffu;   // nibbles of bytes 0,1 and 2,3 joined
      const uint32_t bits = ((y0 | (y0 >> 8)) & 0xffffu) | (((y1 | (y1 >> 8)) & 0xffffu) << 16);
      const int q = (sym - 4) / 18, sidx = (sym - 4) % 18;
      const int r = w / 24, wq = w % 24;
      const int delay = static_cast<int>(__brev(static_cast<unsigned>(r)) >> 28);   // map[r], misc.c:32
      o.msc[static_cast<size_t>(o.cif_row + q - delay) * (1728 * 4) + r * (108 * 4) + sidx * 24 + wq] = bits;
    }
  }
}
#else
// 96 output words of 32 decision bytes each.  MSC symbols: word t is bytes 32 t .. 32 t + 31 (plane t / 6, word t % 6).  FIC
// symbols want natural order: bit b of word t is decision i = 32 t + b, whose byte sits at A(i) (see decide) =
// (b & 15) 192 + (t >> 4) 32 + (2 t & 31) + (b >> 4): 16 two-byte reads (three symbols per TF, in part 0 only).
__device__ __forceinline__ void flush_symbol(const uint8_t* dec, int sym, const FusedOut& o)
{
  const int tid = threadIdx.x;
  if (tid >= 96) return;
  if (sym <= 3) {
    const uint8_t* src = dec + (tid >> 4) * 32 + ((2 * tid) & 31);
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t u = *reinterpret_cast<const uint16_t*>(src + 192 * j);
      bits |= ((u & 1u) << j) | ((u >> 8) << (16 + j));
    }
#if DABHIP_FUSED_GUARD
    if (tid >= 48) bits = ~bits;                          // second bits of the carriers: the bytes hold the sign of im (decide)
#endif
    o.fic_row[(sym - 1) * 96 + tid] = bits;
  } else {
    uint32_t bits = pack_word(dec, tid);
    const int q = (sym - 4) / 18, sidx = (sym - 4) % 18;
    const int r = tid / 6, wq = tid % 6;
#if DABHIP_FUSED_GUARD
    if (wq >= 3) bits = ~bits;                            // words 3..5 of a plane: the second bits (sign of im, see decide)
#endif
    const int delay = static_cast<int>(__brev(static_cast<unsigned>(r)) >> 28);   // map[r], misc.c:32
    o.msc[static_cast<size_t>(o.cif_row + q - delay) * 1728 + r * 108 + sidx * 6 + wq] = bits;
  }
}
#endif

// symbols [sym_begin, sym_end): transform, and from the second one on demap against the one before
template <bool kFast, bool kNco>
__device__ __forceinline__ void fused_symbols(GlobalU16 fast_src, const uint8_t* stream, const FrameView& view, int sym_begin, int sym_end,
                                              bool have_prev, float2 (&px)[4], float2 (&py)[4], float2* exA, float2* exB, uint8_t* decA,
                                              uint8_t* decB, const Twiddles& tw, const int (&qk)[8], uint32_t nco_inc, const FusedOut& out,
                                              const FusedGuard& guard, int* esum, float& dprev, float& maxp)
{
  if (sym_begin >= sym_end) return;
#if DABHIP_FUSED_SOFT
  constexpr bool guarded = true;                        // "energies wanted": the soft scale is made of them
#elif DABHIP_FUSED_GUARD
  const bool guarded = guard.g.delta != nullptr;
#else
  constexpr bool guarded = false;
#endif
  bool have_out = false;                                // decisions of the previous symbol wait in the other dec array
  unsigned raw[8];
  if (kFast) load_symbol<true>(fast_src, stream, view, sym_begin, raw);
  else load_symbol_view(stream, view, sym_begin, decA, raw);
  // two symbols per trip: the decision arrays swap roles every symbol, so the trip body sees them at fixed places
  for (int sym = sym_begin; sym < sym_end; sym += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s = sym + h;
      if (s < sym_end) {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = sample_f32(raw[r]);
        if (kNco && nco_inc) derotate(v, nco_inc, kNullSamples + kSymSamples * s + kCpSamples + static_cast<int>(threadIdx.x));
        // prefetch under the transform; unconditional (a branch here makes the compiler wait for the loads at once)
        if (kFast) load_symbol<true>(fast_src, stream, view, min(s + 1, sym_end - 1), raw);
        else load_symbol_view(stream, view, min(s + 1, sym_end - 1), h ? decB : decA, raw);
        float2 x[4], y[4];
#ifndef DABHIP_PROBE_NOENERGY
        if (guarded) symbol_energy_part(v, esum + 4 * h);
#endif
        fft2048_first(v, exA, tw);
#ifndef DABHIP_PROBE_NOENERGY
        const float dcur = guarded ? symbol_bound(esum + 4 * h, DABHIP_FUSED_SOFT ? kSoftNormC : guard.g.c) : 0.0f;
#else
        const float dcur = 1.0e-3f;
#endif
        // the barrier just passed also orders the previous symbol's decisions: they leave now, one symbol late, so that
        // the wait for the NEXT prefetch (issued above) never has to drain a store issued right before it
        if (have_out) flush_symbol(h ? decA : decB, s - 1, out);
        fft2048_rest(v, exA, exB, tw, x, y);
#if DABHIP_FUSED_AUDIT
        audit_dump_bins(x, y, guard.frame, s);
#endif
#if DABHIP_FUSED_SOFT
        if (have_prev) decide(x, y, px, py, qk, h ? decB : decA, soft_scale(dcur, dprev));
#else
        if (have_prev) decide(x, y, px, py, qk, h ? decB : decA, guard, s, dcur, dprev, maxp);
#if DABHIP_FUSED_GUARD
        else maxp = bins_l1max(x, y);                     // the run's reference symbol
#endif
#endif
        dprev = dcur;
        have_out = have_prev;
        have_prev = true;
#pragma unroll
        for (int k3 = 0; k3 < 4; ++k3) { px[k3] = x[k3]; py[k3] = y[k3]; }
      }
    }
  }
  if (have_out) {                                       // the run's last symbol ((sym_end - 1 - sym_begin) & 1 picks its array)
    __syncthreads();
    flush_symbol(((sym_end - 1 - sym_begin) & 1) ? decB : decA, sym_end - 1, out);
  }
}

// grid = nparts * nframes: the data symbols [sym_a, sym_b) of every frame (1 <= sym_a: symbol 0 is the phase reference only), split over
// nparts workgroups per frame; a workgroup transforms the symbol before its first data symbol once more as differential reference.
// Four workgroups per CU (16 waves, 128 VGPRs each, 39.9 KB of LDS each): the symbol loop is a chain of LDS round trips and barriers,
// and a fourth wave per SIMD hides more of them than the handful of spilled registers costs (guarded 5.3 -> 4.9 ms per 16 k TF, plain
// 4.5 -> 4.1; measured with three: DABHIP_FUSED_WG_PER_CU=3).
#ifndef DABHIP_FUSED_WG_PER_CU
#define DABHIP_FUSED_WG_PER_CU 4
#endif
template <bool kNco>
__global__ __launch_bounds__(kThreads, DABHIP_FUSED_WG_PER_CU) void ofdm_demap_kernel(const uint8_t* const* __restrict__ iq, const CallDesc* __restrict__ descs,
                                                                 int max_calls, const int2* __restrict__ frames, int first,
                                                                 const float2* __restrict__ tw_global, const int* __restrict__ frame_slot,
                                                                 const int* __restrict__ frame_cif_row,
                                                                 const uint16_t* __restrict__ qpsk_of_carrier,
                                                                 uint32_t* __restrict__ fic_bits, uint32_t* __restrict__ msc_bits, const GuardArgs gargs,
                                                                 int sym_a, int sym_b, int nparts)
{
  __shared__ __attribute__((aligned(16))) float2 exA[kPSize];        // P: stage A -> B
  __shared__ __attribute__((aligned(16))) float2 exB[kQSize];        // Q: stage B -> C
  __shared__ __attribute__((aligned(16))) uint8_t decA[kBitsPerSym], decB[kBitsPerSym];   // 3072 B each
  __shared__ FrameView view;
  __shared__ float2 tw3[4 * 8];
  __shared__ __attribute__((aligned(16))) int esum[8];  // per-wave parts of the symbol energy, two symbols in flight
  const int tid = threadIdx.x;
  const int j = blockIdx.x / nparts, part = blockIdx.x % nparts;   // the engine launches the FIC symbols of all frames first, then the MSC symbols
  const int2 fr = frames[first + j];
  const CallDesc* desc = descs + static_cast<size_t>(fr.x) * max_calls + fr.y;
  const uint8_t* stream = iq[fr.x];
  if (tid == 0) view = desc->view;
  if (tid < 32) tw3[tid] = tw_global[64 * (tid >> 3) * (tid & 7)];
  Twiddles tw;
  {
    const int t1 = 4 * (tid & 7) + ((tid >> 3) & 3), t2 = (tid >> 4) & 3;   // (c, d) of this thread in stage B, d in stage C (fft2048_rest)
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      tw.s1[q - 1] = tw_global[tid * q];
      tw.s2[q - 1] = tw_global[8 * t1 * q];
    }
    tw.s3 = tw3 + 8 * t2;
  }
  // QPSK symbol index of each of this thread's 8 bins (frequency de-interleaver, dab_tables.c:164); -1 = no carrier there.
  // Raw bin k: carriers 768..1535 sit at k = 1..768, carriers 0..767 at k = 1280..2047 (input_sdr.c:146-162 on the shifted array).
  int qk[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = fused_bin(tid, m);
    const int c = (k >= 1 && k <= 768) ? k + 767 : (k >= 1280 ? k - 1280 : -1);
    qk[m] = c >= 0 ? qpsk_of_carrier[c] : -1;
#if DABHIP_FUSED_SOFT
    if (c >= 0) qk[m] = (qk[m] & 15) * 192 + (qk[m] >> 4);                              // where the value byte goes (decide)
#else
    if (c >= 0) qk[m] = ((qk[m] & 15) * 6 + (qk[m] >> 9)) * 32 + ((qk[m] >> 4) & 31);   // where the decision byte goes (decide)
#endif
  }
  const FusedOut out{fic_bits + static_cast<size_t>(frame_slot[first + j]) * (DABHIP_FUSED_SOFT ? 288 * 4 : 288), msc_bits, frame_cif_row[first + j]};
  const int seg_end0 = desc->view.seg_end[0];
  const int64_t seg_src0 = desc->view.seg_src[0];
  const int nco_hz = desc->nco_hz;
  __syncthreads();

  int nfast = 0;                                        // symbols whose window lies inside what this call read (see fft_block)
  if (seg_src0 >= 0) {
    const int avail = (seg_end0 - 4096) / 2 - kNullSamples - kCpSamples;
    if (seg_end0 >= 4096 && avail >= 0) nfast = min(kSymbolsPerTf, avail / kSymSamples + 1);
  }
  GlobalU16 src = reinterpret_cast<GlobalU16>(reinterpret_cast<uintptr_t>(stream + (seg_src0 >= 0 ? seg_src0 : 0)));
  const uint32_t nco_inc = nco_hz ? static_cast<uint32_t>(static_cast<int64_t>(llrint(nco_hz * (4294967296.0 / 2048000.0)))) : 0u;

  const int per = (sym_b - sym_a + nparts - 1) / nparts;             // data symbols per workgroup
  const int sym_begin = sym_a + part * per - 1, sym_end = min(sym_b, sym_a + (part + 1) * per);   // the symbol before the first one is its reference
  // fast run (an even number of symbols, so that the LDS buffers end where they started), then the rest through the view
  int fast_end = max(sym_begin, min(sym_end, nfast));
  fast_end -= (fast_end - sym_begin) & 1;
  float2 px[4], py[4];                                  // the previous symbol's bins
  // the guard re-decides from the raw samples: with the software AFC's NCO in the path there is no such exact reference
  FusedGuard guard{gargs, static_cast<unsigned>(first + j), 0};
  if (nco_inc) guard.g.delta = nullptr;
  float dprev = 0.0f;                                   // error bound of the previous symbol's bins
  float maxp = 0.0f;                                    // largest |bin|_1 of the previous symbol among this thread's bins (guarded build)
  guard.fast_base = reinterpret_cast<uintptr_t>(stream + (seg_src0 >= 0 ? seg_src0 : 0));
  fused_symbols<true, kNco>(src, stream, view, sym_begin, fast_end, false, px, py, exA, exB, decA, decB, tw, qk, nco_inc, out, guard, esum, dprev, maxp);
  guard.fast_base = 0;
  fused_symbols<false, kNco>(src, stream, view, fast_end, sym_end, fast_end > sym_begin, px, py, exA, exB, decA, decB, tw, qk, nco_inc, out, guard, esum, dprev, maxp);
}

}  // namespace

#if DABHIP_FUSED_SOFT
hipError_t launch_ofdm_demap_fused_soft(bool afc, const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                                        const float2* tw, const int* frame_slot, const int* frame_cif_row, const uint16_t* qpsk_of_carrier,
                                        uint32_t* fic_bits, uint32_t* msc_bits, hipStream_t stream, int sym_a, int sym_b, int nparts)
{
  if (nframes <= 0 || nparts <= 0) return hipSuccess;
  const GuardArgs guard{};
  if (afc)
    hipLaunchKernelGGL(ofdm_demap_kernel<true>, dim3(nparts * nframes), dim3(kThreads), 0, stream, iq, descs, max_calls, frames, first, tw, frame_slot,
                       frame_cif_row, qpsk_of_carrier, fic_bits, msc_bits, guard, sym_a, sym_b, nparts);
  else
    hipLaunchKernelGGL(ofdm_demap_kernel<false>, dim3(nparts * nframes), dim3(kThreads), 0, stream, iq, descs, max_calls, frames, first, tw, frame_slot,
                       frame_cif_row, qpsk_of_carrier, fic_bits, msc_bits, guard, sym_a, sym_b, nparts);
  return hipGetLastError();
}
#elif DABHIP_FUSED_AUDIT
hipError_t launch_ofdm_demap_fused_audit(const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                                         const float2* tw, const int* frame_slot, const int* frame_cif_row, const uint16_t* qpsk_of_carrier,
                                         uint32_t* fic_bits, uint32_t* msc_bits, const GuardArgs& guard, hipStream_t stream, int sym_a, int sym_b, int nparts,
                                         float2* dump_bins, float2* dump_prod)
{
  if (nframes <= 0 || nparts <= 0) return hipSuccess;
  hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_audit_bins), &dump_bins, sizeof dump_bins, 0, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_audit_prod), &dump_prod, sizeof dump_prod, 0, hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ofdm_demap_kernel<false>, dim3(nparts * nframes), dim3(kThreads), 0, stream, iq, descs, max_calls, frames, first, tw, frame_slot,
                     frame_cif_row, qpsk_of_carrier, fic_bits, msc_bits, guard, sym_a, sym_b, nparts);
  return hipGetLastError();
}
#elif DABHIP_FUSED_GUARD
hipError_t launch_ofdm_demap_fused_guarded(const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                                           const float2* tw, const int* frame_slot, const int* frame_cif_row, const uint16_t* qpsk_of_carrier,
                                           uint32_t* fic_bits, uint32_t* msc_bits, const GuardArgs& guard, hipStream_t stream, int sym_a, int sym_b, int nparts)
{
  if (nframes <= 0 || nparts <= 0) return hipSuccess;
  hipLaunchKernelGGL(ofdm_demap_kernel<false>, dim3(nparts * nframes), dim3(kThreads), 0, stream, iq, descs, max_calls, frames, first, tw, frame_slot,
                     frame_cif_row, qpsk_of_carrier, fic_bits, msc_bits, guard, sym_a, sym_b, nparts);
  return hipGetLastError();
}
#else
hipError_t launch_ofdm_demap_fused_plain(bool afc, const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                                         const float2* tw, const int* frame_slot, const int* frame_cif_row, const uint16_t* qpsk_of_carrier,
                                         uint32_t* fic_bits, uint32_t* msc_bits, hipStream_t stream, int sym_a, int sym_b, int nparts)
{
  if (nframes <= 0 || nparts <= 0) return hipSuccess;
  const GuardArgs guard{};
  if (afc)
    hipLaunchKernelGGL(ofdm_demap_kernel<true>, dim3(nparts * nframes), dim3(kThreads), 0, stream, iq, descs, max_calls, frames, first, tw, frame_slot,
                       frame_cif_row, qpsk_of_carrier, fic_bits, msc_bits, guard, sym_a, sym_b, nparts);
  else
    hipLaunchKernelGGL(ofdm_demap_kernel<false>, dim3(nparts * nframes), dim3(kThreads), 0, stream, iq, descs, max_calls, frames, first, tw, frame_slot,
                       frame_cif_row, qpsk_of_carrier, fic_bits, msc_bits, guard, sym_a, sym_b, nparts);
  return hipGetLastError();
}
#endif

}  // namespace dabhip
