// k_parity.hip — the parity guard of the OFDM stage, and the audit that calibrates it.
//
// The reference demaps with sign tests on fp64 FFTW spectra (input_sdr.c:132-162: bit = Re(cur conj(prev)) <= 0 etc.).  K2 /
// the fused OFDM kernel transform in fp32, so a decision whose |Re| or |Im| lies inside the fp32 error of the product can come
// out differently.  The demapping kernels therefore list every decision with
//     min(|Re|, |Im|)  <  |cur| d(l-1) + |prev| d(l) + P |cur| |prev| + d(l) d(l-1),     d(l) = C sqrt(sum_n |x_n|^2)
// (device_types.hpp: guard_threshold; d(l) bounds the error of any bin of symbol l's fp32 transform, P the rounding of the product) at one of two levels:
// the MEASURED one (C, P = 5e-6, 5e-7: 4.5 x the worst errors the audit below has measured; |.|_1 norms; one band for all bins) or the PROVEN one, the default
// (C, P = 6.6e-5, 1.25e-7: >= a rigorous forward-error bound of this transform and product; |.|_2 norms; the band scaled per bin by the
// bin's own stage terms, guard_bin_scale), and exact_decide_kernel re-decides the listed carriers from the int8 samples in fp64 by direct summation (relative
// error 1e-13; bins 512 / 1536 and their products exactly) and patches the two bits.  Nothing on clean input, 1e-5 .. 1e-4 of the decisions at 5 dB: the output is
// the one exact arithmetic gives.
//
// decision_audit_kernel is test / calibration infrastructure behind dabhip_stage_decision_audit: fp64 transforms of every
// symbol on the GPU (fft64.hpp), compared bin by bin and decision by decision with what the fp32 stage produced.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "dab_tables.hpp"
#include "device_types.hpp"
#include "fft64.hpp"
#include "kernels.hpp"

namespace dabhip {
namespace {

__device__ __forceinline__ int pview_byte(const uint8_t* stream, const FrameView& v, int p) { return frame_byte(stream, v, p); }
__device__ __forceinline__ int prail(int byte) { return static_cast<int>(static_cast<int8_t>(static_cast<uint8_t>(byte - 127))); }

// ---- d(l) for symbols [sym0, sym0 + nsym) of every frame: one wave per symbol -------------------------------------
__global__ __launch_bounds__(256) void symbol_delta_kernel(const uint8_t* const* __restrict__ iq, const CallDesc* __restrict__ descs,
                                                           int max_calls, const int2* __restrict__ frames, int first, int nframes,
                                                           int nsym, float* __restrict__ delta, int delta_stride, float delta_c)
{
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= nframes * nsym) return;
  const int j = w / nsym, l = w % nsym;
  const int2 fr = frames[first + j];
  const CallDesc* desc = descs + static_cast<size_t>(fr.x) * max_calls + fr.y;
  const uint8_t* stream = iq[fr.x];
  const int start = 2 * (kNullSamples + kSymSamples * l + kCpSamples);
  int acc = 0;
  const int seg_end0 = desc->view.seg_end[0];
  const int64_t seg_src0 = desc->view.seg_src[0];
  if (seg_src0 >= 0 && start + 4096 <= seg_end0) {        // the window lies inside what this call read: contiguous 2-byte loads
    const uint16_t* src = reinterpret_cast<const uint16_t*>(stream + seg_src0 + start);
#pragma unroll 8
    for (int n = lane; n < 2048; n += 64) {
      const unsigned w = src[n];
      const int a = prail(w & 0xff), b = prail(w >> 8);
      acc += a * a + b * b;
    }
  } else {
    for (int n = lane; n < 2048; n += 64) {
      const int p = start + 2 * n;
      const int a = prail(pview_byte(stream, desc->view, p)), b = prail(pview_byte(stream, desc->view, p + 1));
      acc += a * a + b * b;
    }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
  if (lane == 0) delta[static_cast<size_t>(first + j) * delta_stride + l] = delta_c * sqrtf(static_cast<float>(acc));
}

// ---- re-decide the flagged carriers in fp64: one wave per entry -----------------------------------------------------
// entry = GuardArgs::list's {frame index into `frames`, symbol << 16 | raw bin, address of the symbol's samples or 0}.  X_l[k] = sum_n x_n exp(-2 pi i n k / 2048) by
// direct summation.  This is the general form, sample by sample through the frame's view (windows that reach into the stale tail); exact_bin_pair below is the common case.
__device__ __forceinline__ void exact_bin(const uint8_t* stream, const FrameView& view, int l, int k, const double2* __restrict__ tw2048,
                                          double* xr, double* xi)
{
  const int lane = threadIdx.x & 63;
  const int start = 2 * (kNullSamples + kSymSamples * l + kCpSamples);
  double sr = 0, si = 0;
  if (view.seg_src[0] >= 0 && start + 4096 <= view.seg_end[0]) {
    // the window lies inside what this call read (all but the frames after a resync): 2-byte loads, all 32 of a lane in flight at once.  At 5 dB a step
    // lists 26,000 decisions, and as a chain of byte loads through the view and 2048 scattered table reads per bin this kernel took 0.9 ms of it; now 0.15.
    // The factor of sample n = lane + 64 i is split, exp(2 pi i k lane / 2048) exp(2 pi i (64 k) i / 2048): one table entry per lane and one per i that is
    // the same in every lane (a scalar load), instead of 2048 scattered 16-byte reads per bin -- their product carries one rounding more (1e-16).
    const uint16_t* src = reinterpret_cast<const uint16_t*>(stream + view.seg_src[0] + start);
    unsigned raw[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) raw[i] = src[lane + 64 * i];
    const double2 wl = tw2048[(lane * k) & 2047];
    const int k64 = __builtin_amdgcn_readfirstlane((64 * k) & 2047);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const double2 wi = tw2048[(k64 * i) & 2047];
      const double wx = wl.x * wi.x - wl.y * wi.y, wy = wl.x * wi.y + wl.y * wi.x;
      const double a = prail(raw[i] & 0xff), b = prail(raw[i] >> 8);
      sr += a * wx + b * wy;
      si += b * wx - a * wy;
    }
  } else {
    for (int n = lane; n < 2048; n += 64) {
      const int p = start + 2 * n;
      const double a = prail(pview_byte(stream, view, p)), b = prail(pview_byte(stream, view, p + 1));
      const double2 w = tw2048[(n * k) & 2047];             // exp(+2 pi i nk / 2048); the forward transform uses the conjugate
      sr += a * w.x + b * w.y;
      si += b * w.x - a * w.y;
    }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) { sr += __shfl_xor(sr, s); si += __shfl_xor(si, s); }
  *xr = sr;
  *xi = si;
}

// The common case, three times as fast (round 6: the proven guard level lists 13 x as many decisions, 325,000 per step at 5 dB): both symbols' windows lie
// inside what the call read.  ONE entry per wave, lanes 0..31 sum bin k of symbol l, lanes 32..63 of symbol l - 1.  A lane owns the samples
//     n = 16 (h + 32 i) + j,   h = lane & 31, i = 0..3, j = 0..15:   exp(-2 pi i n k / 2048) = W^(16 h k) (-i)^(i k) W^(j k),  W = exp(-2 pi i / 2048):
//   * its four 32-byte pieces (two 16-byte loads each; the half-wave reads 1 KB at a stretch) differ by quarter turns (-i)^(i k), which are EXACT on the int8
//     samples: z_j = sum_i (-i)^(i k) x[16 (h + 32 i) + j] in integers (|z_j| <= 512), two sign-extending byte adds per sample, the turn chosen at compile
//     time from k & 3 (four instances, one wave-uniform branch);
//   * the rail values (int8)(byte - 127) of the four bytes of a word come from two masked adds: even bytes from (w & 0x00ff00ff) + 0x00810081, odd bytes from
//     (w & 0xff00ff00) + 0x81008100 -- each byte's carry lands in a cleared neighbour, and 255 wraps to -128 as in the reference (input_sdr.c:60-63);
//   * sum_j z_j W^(j k): 16 factors that are the same in every lane (scalar loads), 60 fused multiply-adds; one factor W^(16 h k) per lane; a 32-lane sum.
// 300 vector instructions per entry instead of 1200; the same fp64 accuracy (fewer roundings, if anything: the inner sums are exact).
__device__ __forceinline__ uint4 load16(const uint8_t* p)
{
  uint4 v;
  __builtin_memcpy(&v, p, 16);              // (2-byte aligned: the hardware's unaligned global access mode serves it)
  return v;
}
template <int kRot>
__device__ __forceinline__ void turn_add(int& zr, int& zi, int a, int b)      // z += (-i)^kRot (a + i b)
{
  if (kRot == 0) { zr += a; zi += b; }
  else if (kRot == 1) { zr += b; zi -= a; }
  else if (kRot == 2) { zr -= a; zi -= b; }
  else { zr -= b; zi += a; }
}
template <int K4, int I>
__device__ __forceinline__ void exact_piece(const uint8_t* src, int (&zr)[16], int (&zi)[16])
{
  const uint4 a = load16(src + 1024 * I), b = load16(src + 1024 * I + 16);
  const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  constexpr int rot = (I * K4) & 3;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const unsigned ev = (w[d] & 0x00ff00ffu) + 0x00810081u, od = (w[d] & 0xff00ff00u) + 0x81008100u;
    turn_add<rot>(zr[2 * d], zi[2 * d], static_cast<int8_t>(ev), static_cast<int8_t>(od >> 8));
    turn_add<rot>(zr[2 * d + 1], zi[2 * d + 1], static_cast<int8_t>(ev >> 16), static_cast<int8_t>(od >> 24));
  }
}
// after the call: lanes 0..31 hold X_l[k], lanes 32..63 X_(l-1)[k] (cur / prev = the first byte of the two symbols' windows)
template <int K4>
__device__ __forceinline__ void exact_bin_pair(const uint8_t* cur, const uint8_t* prev, int k, const double2* __restrict__ tw2048, double* xr_out, double* xi_out)
{
  const int lane = threadIdx.x & 63, h = lane & 31;
  const uint8_t* src = (lane < 32 ? cur : prev) + 32 * h;
  int zr[16], zi[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) zr[j] = zi[j] = 0;
  exact_piece<K4, 0>(src, zr, zi);
  exact_piece<K4, 1>(src, zr, zi);
  exact_piece<K4, 2>(src, zr, zi);
  exact_piece<K4, 3>(src, zr, zi);
  double sr = zr[0], si = zi[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) {
    const double2 w = tw2048[(k * j) & 2047];                 // exp(+2 pi i j k / 2048), the same in every lane; the forward transform uses the conjugate
    const double a = zr[j], b = zi[j];
    sr = __builtin_fma(a, w.x, sr);
    sr = __builtin_fma(b, w.y, sr);
    si = __builtin_fma(b, w.x, si);
    si = __builtin_fma(-a, w.y, si);
  }
  const double2 wl = tw2048[(16 * h * k) & 2047];
  double xr = __builtin_fma(sr, wl.x, si * wl.y), xi = __builtin_fma(si, wl.x, -(sr * wl.y));
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) { xr += __shfl_xor(xr, s); xi += __shfl_xor(xi, s); }
  *xr_out = xr;
  *xi_out = xi;
}

// Eight waves per SIMD (64 registers): an entry is a chain of dependent loads (list entry -> samples -> output word), and the resident waves are what
// hides it -- with four, the 325,000 entries of a 5 dB step at the proven level took 0.79 ms, most of it waiting.  The entry itself carries the address of
// its samples (GuardArgs::list), and the next entry is fetched while this one is summed.
__global__ __launch_bounds__(256, 8) void exact_decide_kernel(const uint4* __restrict__ list, const unsigned* __restrict__ counter, unsigned cap,
                                                              const uint8_t* const* __restrict__ iq, const CallDesc* __restrict__ descs, int max_calls,
                                                              const int2* __restrict__ frames, const double2* __restrict__ tw2048,
                                                              const uint16_t* __restrict__ qpsk_of_carrier,
                                                              const int* __restrict__ frame_slot,
                                                              const int* __restrict__ frame_cif_row, int planar, uint32_t* __restrict__ fic_bits,
                                                              uint32_t* __restrict__ msc_bits, int sample_by_sample)
{
  const unsigned n = min(counter[0], cap);
  const int lane = threadIdx.x & 63;
  const unsigned stride = gridDim.x * 4;
  unsigned e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n) return;
  uint4 ent = list[e];
  for (; e < n; e += stride) {
    const uint4 next = e + stride < n ? list[e + stride] : ent;            // in flight while this entry is summed
    const int f = __builtin_amdgcn_readfirstlane(static_cast<int>(ent.x)), l = __builtin_amdgcn_readfirstlane(static_cast<int>(ent.y >> 16));
    const int k = __builtin_amdgcn_readfirstlane(static_cast<int>(ent.y & 0x7ffu));     // raw bin
    uint64_t win = (static_cast<uint64_t>(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ent.w)))) << 32) |
                   static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ent.z)));
    if (sample_by_sample) win = 0;                        // test knob (DABHIP_EXACT_SAMPLEWISE=1): every entry through the general form
    ent = next;
    // where the two bits go (independent of the sums: these loads fly beside the samples')
    const int c = (k >= 1 && k <= 768) ? k + 767 : k - 1280;
    const int q = qpsk_of_carrier[c];
    const int slot_or_row = l <= 3 ? frame_slot[f] : frame_cif_row[f];
    double cr, ci, pr, pi;
    if (win) {
      const uint8_t* cur = reinterpret_cast<const uint8_t*>(static_cast<uintptr_t>(win));
      const uint8_t* prev = cur - 2 * kSymSamples;
      double xr, xi;
      switch (k & 3) {
        case 0: exact_bin_pair<0>(cur, prev, k, tw2048, &xr, &xi); break;
        case 1: exact_bin_pair<1>(cur, prev, k, tw2048, &xr, &xi); break;
        case 2: exact_bin_pair<2>(cur, prev, k, tw2048, &xr, &xi); break;
        default: exact_bin_pair<3>(cur, prev, k, tw2048, &xr, &xi); break;
      }
      cr = xr;
      ci = xi;
      pr = __shfl(xr, 32);
      pi = __shfl(xi, 32);
    } else {
      // no address: an entry of the two-kernel stage (which decides on spectra), or a symbol read through the view (its window reaches into the stale
      // tail of the frame buffer: frames after a resync) -- frame list, descriptor, view; in place where the windows allow it, else sample by sample
      const int2 fr = frames[f];
      const CallDesc* desc = descs + static_cast<size_t>(fr.x) * max_calls + fr.y;
      const uint8_t* stream = iq[fr.x];
      const int start = 2 * (kNullSamples + kSymSamples * l + kCpSamples);
      if (!sample_by_sample && desc->view.seg_src[0] >= 0 && start + 4096 <= desc->view.seg_end[0]) {
        const uint8_t* cur = stream + desc->view.seg_src[0] + start;
        const uint8_t* prev = cur - 2 * kSymSamples;
        double xr, xi;
        switch (k & 3) {
          case 0: exact_bin_pair<0>(cur, prev, k, tw2048, &xr, &xi); break;
          case 1: exact_bin_pair<1>(cur, prev, k, tw2048, &xr, &xi); break;
          case 2: exact_bin_pair<2>(cur, prev, k, tw2048, &xr, &xi); break;
          default: exact_bin_pair<3>(cur, prev, k, tw2048, &xr, &xi); break;
        }
        cr = xr;
        ci = xi;
        pr = __shfl(xr, 32);
        pi = __shfl(xi, 32);
      } else {
        exact_bin(stream, desc->view, l, k, tw2048, &cr, &ci);
        exact_bin(stream, desc->view, l - 1, k, tw2048, &pr, &pi);
      }
    }
#ifdef DABHIP_EXACT_DEBUG
    {
      const int2 fr = frames[f];
      const CallDesc* desc = descs + static_cast<size_t>(fr.x) * max_calls + fr.y;
      double a0, a1, a2, a3;
      exact_bin(iq[fr.x], desc->view, l, k, tw2048, &a0, &a1);
      exact_bin(iq[fr.x], desc->view, l - 1, k, tw2048, &a2, &a3);
      if (lane == 0) {
        const double re1 = cr * pr + ci * pi, im1 = cr * pi - ci * pr, re2 = a0 * a2 + a1 * a3, im2 = a0 * a3 - a1 * a2;
        if ((re1 > 0.0) != (re2 > 0.0) || (im1 > 0.0) != (im2 > 0.0))
          printf("MISMATCH f=%d l=%d k=%d pair: c=(%.17g,%.17g) p=(%.17g,%.17g) re=%.6g im=%.6g | samplewise: c=(%.17g,%.17g) p=(%.17g,%.17g) re=%.6g im=%.6g\n", f, l, k, cr, ci, pr,
                 pi, re1, im1, a0, a1, a2, a3, re2, im2);
      }
    }
#endif
    if (lane != 0) continue;
    const double re = cr * pr + ci * pi;                  // Re(cur conj(prev)); the reference divides by |prev|^2 > 0 (input_sdr.c:135-143)
    const double im = cr * pi - ci * pr;                  // the imaginary part as stored there
    const unsigned b0 = (re > 0.0) ? 0u : 1u, b1 = (im > 0.0) ? 1u : 0u;      // input_sdr.c:157-158
    const int pos[2] = {q, 1536 + q};
    const unsigned bit[2] = {b0, b1};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = pos[h];
      uint32_t* word;
      unsigned sh;
      if (l <= 3) {
        word = fic_bits + static_cast<size_t>(slot_or_row) * 288 + (l - 1) * 96 + (i >> 5);
        sh = i & 31;
      } else if (planar) {                               // layout of demap_kernel<true, 1> / the fused kernel
        const int qc = (l - 4) / 18, sidx = (l - 4) % 18, r = i & 15, u = i >> 4;
        const int delay = static_cast<int>(__brev(static_cast<unsigned>(r)) >> 28);
        word = msc_bits + static_cast<size_t>(slot_or_row + qc - delay) * 1728 + r * 108 + sidx * 6 + (u >> 5);
        sh = u & 31;
      } else {
        word = msc_bits + static_cast<size_t>(slot_or_row) * 1728 + (l - 4) * 96 + (i >> 5);
        sh = i & 31;
      }
      if (bit[h]) atomicOr(word, 1u << sh);
      else atomicAnd(word, ~(1u << sh));
    }
  }
}

// ---- list overflow: every decision of the launch again, in fp64 ------------------------------------------------------------
// More decisions inside the fp32 error band than the list holds (input that synchronises but has many near-zero products: strongly
// notched, narrowband or near-DC frames) used to make the decode fail.  Now the launch's frames are simply decided again in full, by
// fp64 transforms of their symbols (the audit's transform): slow -- 76 double-precision 2048-point transforms per frame -- but only ever
// run for a launch whose list overflowed, and the bits are then those of exact arithmetic like everywhere else.  A fixed small grid:
// the normal case (no overflow) costs one counter read per workgroup.  One workgroup per frame at a time, symbols [sym_a - 1, sym_b)
// in order (the previous symbol's bins stay in LDS as the differential reference), the symbol's 96 output words built in LDS.
__global__ __launch_bounds__(kFft64Threads) void exact_decide_all_kernel(const unsigned* __restrict__ counter, unsigned cap, const uint8_t* const* __restrict__ iq,
                                                                        const CallDesc* __restrict__ descs, int max_calls, const int2* __restrict__ frames,
                                                                        int first, int nframes, int sym_a, int sym_b, const double2* __restrict__ tw2048,
                                                                        const uint16_t* __restrict__ qpsk_of_carrier, const int* __restrict__ frame_slot,
                                                                        const int* __restrict__ frame_cif_row, int planar, int skip_fic,
                                                                        uint32_t* __restrict__ fic_bits, uint32_t* __restrict__ msc_bits)
{
  if (counter[0] <= cap) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2* A = reinterpret_cast<double2*>(smem);
  double2* P = A + 2048;
  double2* tw = P + 2048;                                  // 1024 twiddles
  __shared__ uint32_t words[96];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += kFft64Threads) tw[lds_at(i)] = tw2048[i];
  for (int j = blockIdx.x; j < nframes; j += gridDim.x) {
    const int f = first + j;
    const int2 fr = frames[f];
    const CallDesc* desc = descs + static_cast<size_t>(fr.x) * max_calls + fr.y;
    const uint8_t* stream = iq[fr.x];
    for (int l = sym_a - 1; l < sym_b; ++l) {
      double2* cur = (l & 1) ? P : A;
      const double2* prev = (l & 1) ? A : P;
      if (tid < 96) words[tid] = 0u;
      __syncthreads();
      const int start = 2 * (kNullSamples + kSymSamples * l + kCpSamples);
      for (int n = tid; n < 2048; n += kFft64Threads)
        cur[lds_at(n)] = make_double2(prail(pview_byte(stream, desc->view, start + 2 * n)), prail(pview_byte(stream, desc->view, start + 2 * n + 1)));
      __syncthreads();
      dft_dif<11, 3, 3, 3, 2>(cur, 1, -1.0, tw);            // ends with a barrier
      if (l < sym_a || (skip_fic && l <= 3)) continue;      // the reference symbol of the run; FIC symbols another launch owns
      for (int c = tid; c < kCarriers; c += kFft64Threads) {
        const int k = c < 768 ? c + 1280 : c - 767;         // raw bin of carrier c
        const double2 x = cur[lds_at(brev(k, 11))], p = prev[lds_at(brev(k, 11))];
        const double re = x.x * p.x + x.y * p.y, im = x.x * p.y - x.y * p.x;      // input_sdr.c:135-143
        const unsigned b0 = (re > 0.0) ? 0u : 1u, b1 = (im > 0.0) ? 1u : 0u;      // input_sdr.c:157-158
        const int q = qpsk_of_carrier[c];
        const int pos[2] = {q, 1536 + q};
        const unsigned bit[2] = {b0, b1};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = pos[h];
          // word index inside the symbol's 96 output words: natural order, or plane i & 15, bit i >> 4 of that plane (demap_kernel<true, 1>)
          const int w = (l <= 3 || !planar) ? (i >> 5) : (i & 15) * 6 + (i >> 9), sh = (l <= 3 || !planar) ? (i & 31) : ((i >> 4) & 31);
          if (bit[h]) atomicOr(&words[w], 1u << sh);
        }
      }
      __syncthreads();
      if (tid < 96) {
        if (l <= 3) fic_bits[static_cast<size_t>(frame_slot[f]) * 288 + (l - 1) * 96 + tid] = words[tid];
        else if (planar) {
          const int qc = (l - 4) / 18, sidx = (l - 4) % 18, r = tid / 6, wq = tid % 6;
          const int delay = static_cast<int>(__brev(static_cast<unsigned>(r)) >> 28);
          msc_bits[static_cast<size_t>(frame_cif_row[f] + qc - delay) * 1728 + r * 108 + sidx * 6 + wq] = words[tid];
        } else {
          msc_bits[static_cast<size_t>(frame_cif_row[f]) * 1728 + (l - 4) * 96 + tid] = words[tid];
        }
      }
      __syncthreads();
    }
  }
}

// ---- audit: fp64 transforms of every symbol vs the fp32 stage ---------------------------------------------------------
// One 512-thread workgroup per frame (contiguous 393216-byte frames).  out[]: see dabhip_stage_decision_audit.
struct AuditOut {
  unsigned long long decisions, disagree, disagree_outside_guard, flagged;
  unsigned max_bin_err_bits;        // float bits of max |X32 - X64| / sqrt(sum |x|^2)
  unsigned max_dec_err_bits;        // float bits of max |v32 - v64| / (|cur|_1 s(l-1) + |prev|_1 s(l)), v = Re or Im of the product
  unsigned max_prod_err_bits;       // float bits of max over decisions of the part of that error the bin errors do not explain / (|cur|_1 |prev|_1)
  unsigned pad;
};

// kFused = false: the two-kernel stage -- spectra = K2's output (fftshifted), products recomputed as K2b computes them, bits in natural order (frame j: FIC
// slot j, MSC row 4 j).  kFused = true (round 5): the fused kernel's audit build -- spectra = its bins by RAW bin index, prods = the products it decided
// on, bits where that kernel puts them: FIC slot j in natural order, MSC scattered over the planar logical CIF rows from row row_lead + 4 j on.
template <bool kFused>
__global__ __launch_bounds__(kFft64Threads) void decision_audit_kernel(const uint8_t* __restrict__ frames_iq, const float2* __restrict__ spectra,
                                                                      const float2* __restrict__ prods, int row_lead,
                                                                      const uint32_t* __restrict__ fic_bits, const uint32_t* __restrict__ msc_bits,
                                                                      const double2* __restrict__ tw2048, const uint16_t* __restrict__ qpsk_of_carrier,
                                                                      AuditOut* __restrict__ out, float guard_c, float guard_prod, int per_bin)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2* A = reinterpret_cast<double2*>(smem);           // current symbol, bit-reversed after the transform
  double2* P = A + 2048;                                   // previous symbol
  double2* tw = P + 2048;                                  // 1024 twiddles
  __shared__ float s_energy[2];
  __shared__ int s_acc;
  const int j = blockIdx.x, tid = threadIdx.x;
  const uint8_t* frame = frames_iq + static_cast<size_t>(j) * kTfBytes;
  const float2* spec = spectra + static_cast<size_t>(j) * (kSymbolsPerTf * 2048);
  for (int i = tid; i < 1024; i += kFft64Threads) tw[lds_at(i)] = tw2048[i];
  unsigned long long n_dec = 0, n_dis = 0, n_out = 0, n_flag = 0;
  float m_bin = 0, m_dec = 0, m_prod = 0;
  for (int l = 0; l < kSymbolsPerTf; ++l) {
    double2* cur = (l & 1) ? P : A;
    double2* prev = (l & 1) ? A : P;
    if (tid == 0) s_acc = 0;
    __syncthreads();
    const int start = 2 * (kNullSamples + kSymSamples * l + kCpSamples);
    int e = 0;
    for (int n = tid; n < 2048; n += kFft64Threads) {
      const int a = prail(frame[start + 2 * n]), b = prail(frame[start + 2 * n + 1]);
      cur[lds_at(n)] = make_double2(a, b);
      e += a * a + b * b;
    }
    atomicAdd(&s_acc, e);
    __syncthreads();
    if (tid == 0) s_energy[l & 1] = sqrtf(static_cast<float>(s_acc));
    dft_dif<11, 3, 3, 3, 2>(cur, 1, -1.0, tw);            // ends with a barrier
    const float s_cur = s_energy[l & 1], s_prev = s_energy[(l & 1) ^ 1];
    for (int c = tid; c < kCarriers; c += kFft64Threads) {
      const int k = c < 768 ? c + 1280 : c - 767;         // raw bin of carrier c
      const int ks = (k + 1024) & 2047;                   // fftshifted index
      const double2 x64 = cur[lds_at(brev(k, 11))];
      const float2 x32 = spec[l * 2048 + (kFused ? k : ks)];
      if (s_cur > 0) m_bin = fmaxf(m_bin, static_cast<float>(hypot(x32.x - x64.x, x32.y - x64.y)) / s_cur);
      if (l == 0) continue;
      const double2 p64 = prev[lds_at(brev(k, 11))];
      const float2 p32 = spec[(l - 1) * 2048 + (kFused ? k : ks)];
      const double re64 = x64.x * p64.x + x64.y * p64.y, im64 = x64.x * p64.y - x64.y * p64.x;
      float re32 = diff_re(x32.x, x32.y, p32.x, p32.y), im32 = diff_im(x32.x, x32.y, p32.x, p32.y);
      if (kFused) {                                       // what the kernel itself computed (its own instruction sequence, its own roundings)
        const float2 pr = prods[(static_cast<size_t>(j) * kSymbolsPerTf + l) * 2048 + k];
        re32 = pr.x;
        im32 = pr.y;
      }
      const float n1c = fabsf(x32.x) + fabsf(x32.y), n1p = fabsf(p32.x) + fabsf(p32.y);
      const float unit = n1c * s_prev + n1p * s_cur;
      const float t = per_bin ? guard_bin_threshold(x32.x, x32.y, p32.x, p32.y, k, guard_c * s_cur, guard_c * s_prev, guard_prod)
                              : guard_threshold(n1c, n1p, guard_c * s_cur, guard_c * s_prev, guard_prod);   // the kernels' own test
      // (the fused kernel decides by sign bits and therefore also lists every product with an exact zero in it: k_fused.hip, decide)
      const bool flagged = fminf(fabsf(re32), fabsf(im32)) < t || (kFused && !(fminf(fabsf(re32), fabsf(im32)) > 0.0f));
      n_flag += flagged ? 1 : 0;
      const int q = qpsk_of_carrier[c];
      unsigned got0, got1;
      if (!kFused || l <= 3) {
        const uint32_t* row = l <= 3 ? fic_bits + static_cast<size_t>(j) * 288 + (l - 1) * 96 : msc_bits + static_cast<size_t>(4 * j) * 1728 + (l - 4) * 96;
        got0 = (row[q >> 5] >> (q & 31)) & 1u;
        got1 = (row[(1536 + q) >> 5] >> ((1536 + q) & 31)) & 1u;
      } else {
        // decision i of the symbol (i = q, 1536 + q): plane r = i & 15 of logical row (first row of the frame + CIF - map[r]), word i >> 9 of the
        // symbol's six in that plane, bit (i >> 4) & 31 (k_fused.hip: flush_symbol; misc.c:29-39)
        const int cif = (l - 4) / 18, sidx = (l - 4) % 18;
        auto bit_of = [&](int i) {
          const int r = i & 15, delay = static_cast<int>(__brev(static_cast<unsigned>(r)) >> 28);
          const size_t row = static_cast<size_t>(row_lead + 4 * j + cif - delay);
          return (msc_bits[row * 1728 + r * 108 + sidx * 6 + (i >> 9)] >> ((i >> 4) & 31)) & 1u;
        };
        got0 = bit_of(q);
        got1 = bit_of(1536 + q);
      }
      const unsigned want0 = (re64 > 0.0) ? 0u : 1u, want1 = (im64 > 0.0) ? 1u : 0u;
      const int bad = static_cast<int>(got0 != want0) + static_cast<int>(got1 != want1);
      n_dec += 2;
      n_dis += bad;
      // would the guard have caught it?  (audited with the guard OFF: a disagreement on a carrier the rule does not flag is a miss)
      if (bad && !flagged) n_out += bad;
      if (unit > 0) {
        const float err = fmaxf(fabsf(static_cast<float>(re32 - re64)), fabsf(static_cast<float>(im32 - im64)));
        m_dec = fmaxf(m_dec, err / unit);
        // what the bin errors cannot explain, relative to |cur|_1 |prev|_1: rounding of the fp32 product itself
        const float dx = static_cast<float>(hypot(x32.x - x64.x, x32.y - x64.y)), dp = static_cast<float>(hypot(p32.x - p64.x, p32.y - p64.y));
        const float rest = err - (n1p * dx + n1c * dp);
        if (n1c * n1p > 0) m_prod = fmaxf(m_prod, rest / (n1c * n1p));
      }
    }
    __syncthreads();
  }
  atomicAdd(&out->decisions, n_dec);
  atomicAdd(&out->disagree, n_dis);
  atomicAdd(&out->disagree_outside_guard, n_out);
  atomicAdd(&out->flagged, n_flag);
  atomicMax(&out->max_bin_err_bits, __float_as_uint(m_bin));      // non-negative floats order like their bit patterns
  atomicMax(&out->max_dec_err_bits, __float_as_uint(m_dec));
  atomicMax(&out->max_prod_err_bits, __float_as_uint(fmaxf(m_prod, 0.0f)));
}

}  // namespace

hipError_t launch_symbol_delta(const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                               int nsym, float* delta, int delta_stride, float delta_c, hipStream_t stream)
{
  if (nframes <= 0) return hipSuccess;
  hipLaunchKernelGGL(symbol_delta_kernel, dim3((nframes * nsym + 3) / 4), dim3(256), 0, stream, iq, descs, max_calls, frames, first, nframes,
                     nsym, delta, delta_stride, delta_c);
  return hipGetLastError();
}

hipError_t launch_exact_decide(const uint4* list, const unsigned* counter, unsigned cap, const uint8_t* const* iq, const CallDesc* descs,
                               int max_calls, const int2* frames, const double2* tw2048, const uint16_t* qpsk_of_carrier, const uint16_t* /*carrier_of_qpsk*/,
                               const int* frame_slot, const int* frame_cif_row, bool planar, uint32_t* fic_bits, uint32_t* msc_bits, hipStream_t stream)
{
  static const int samplewise = std::getenv("DABHIP_EXACT_SAMPLEWISE") ? std::atoi(std::getenv("DABHIP_EXACT_SAMPLEWISE")) : 0;
  hipLaunchKernelGGL(exact_decide_kernel, dim3(2048), dim3(256), 0, stream, list, counter, cap, iq, descs, max_calls, frames, tw2048,
                     qpsk_of_carrier, frame_slot, frame_cif_row, planar ? 1 : 0, fic_bits, msc_bits, samplewise);
  return hipGetLastError();
}

hipError_t launch_exact_decide_all(const unsigned* counter, unsigned cap, const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames,
                                   int first, int nframes, int sym_a, int sym_b, const double2* tw2048, const uint16_t* qpsk_of_carrier,
                                   const int* frame_slot, const int* frame_cif_row, bool planar, bool skip_fic, uint32_t* fic_bits, uint32_t* msc_bits,
                                   hipStream_t stream)
{
  if (nframes <= 0) return hipSuccess;
  static std::once_flag once[64];
  static hipError_t result[64];
  const size_t lds = sizeof(double2) * (2048 + 2048 + 1024);
  const hipError_t attr = once_per_device(once, result, []() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(exact_decide_all_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  });
  if (attr != hipSuccess) return attr;
  hipLaunchKernelGGL(exact_decide_all_kernel, dim3(std::min(nframes, 512)), dim3(kFft64Threads), lds, stream, counter, cap, iq, descs, max_calls, frames, first,
                     nframes, sym_a, sym_b, tw2048, qpsk_of_carrier, frame_slot, frame_cif_row, planar ? 1 : 0, skip_fic ? 1 : 0, fic_bits, msc_bits);
  return hipGetLastError();
}

hipError_t launch_decision_audit(const uint8_t* frames_iq, int nframes, const float2* spectra, const uint32_t* fic_bits, const uint32_t* msc_bits,
                                 const double2* tw2048, const uint16_t* qpsk_of_carrier, void* out, hipStream_t stream, const float2* fused_prods, int row_lead, int guard_level)
{
  if (nframes <= 0) return hipSuccess;
  static std::once_flag once[64];
  static hipError_t result[64];
  const size_t lds = sizeof(double2) * (2048 + 2048 + 1024);
  const hipError_t attr = once_per_device(once, result, []() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decision_audit_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(decision_audit_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    return e;
  });
  if (attr != hipSuccess) return attr;
  if (fused_prods)
    hipLaunchKernelGGL(decision_audit_kernel<true>, dim3(nframes), dim3(kFft64Threads), lds, stream, frames_iq, spectra, fused_prods, row_lead, fic_bits, msc_bits,
                       tw2048, qpsk_of_carrier, static_cast<AuditOut*>(out), guard_c_of(guard_level), guard_prod_of(guard_level), guard_level >= 2 ? 1 : 0);
  else
    hipLaunchKernelGGL(decision_audit_kernel<false>, dim3(nframes), dim3(kFft64Threads), lds, stream, frames_iq, spectra, nullptr, 0, fic_bits, msc_bits, tw2048,
                       qpsk_of_carrier, static_cast<AuditOut*>(out), guard_c_of(guard_level), guard_prod_of(guard_level), guard_level >= 2 ? 1 : 0);
  return hipGetLastError();
}

}  // namespace dabhip
