// k_fft.hip — K2: the OFDM stage.  76 x 2048-point forward DFTs per transmission frame,
// cu8 IQ in, fftshifted complex64 spectra out (replaces the FFTW loop of
// input_sdr.c:115-130 incl. the u8->s8 conversion of :60-63), and K2b: differential
// demodulation + hard QPSK demap + frequency de-interleave (input_sdr.c:132-162) into
// bit-packed FIC / MSC symbol rows.
//
// K2 is the HBM-roofline stage: per TF it must read 311,296 B of IQ and write 1,245,184 B
// of spectra (1,556,480 B algorithmic) for 8.6 MFLOP, i.e. 5.5 flop/B -- far below the
// machine balance, no dense contraction, no MFMA.  One 256-thread workgroup transforms one
// symbol at a time entirely in registers + LDS: radix 8 x 8 x 8 x 4 decimation in
// frequency, three LDS exchanges with conflict-free (padded) layouts, per-thread twiddles
// kept in registers, next symbol's IQ prefetched under the current transform, coalesced
// 2-byte global loads and 16-byte (1 KiB per wave) stores.
#include <hip/hip_runtime.h>

#include "dab_tables.hpp"
#include "device_types.hpp"
#include "fft_core.hpp"
#include "kernels.hpp"

namespace dabhip {
namespace {

// One 2048-point transform by the whole workgroup.  v holds x[tid + 256 r] on entry.
// bufP / bufQ alternate roles from symbol to symbol, which removes the barrier that would
// otherwise separate the last LDS read of one symbol from the first LDS write of the next.
__device__ __forceinline__ void fft2048_store(float2 (&v)[8], float2* bufP, float2* bufQ, const Twiddles& tw, float2* __restrict__ out)
{
  const int tid = threadIdx.x;
  // stage 1: radix 8 over r, twiddle W_2048^(t q), exchange so that each thread owns one q
  dft8(v);
#pragma unroll
  for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], tw.s1[q - 1]);
#pragma unroll
  for (int q = 0; q < 8; ++q) bufP[q * 256 + tid] = v[q];
  __syncthreads();
  {
    const int q = tid >> 5, t1 = tid & 31;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = bufP[q * 256 + t1 + 32 * r];
    dft8(v);
#pragma unroll
    for (int q2 = 1; q2 < 8; ++q2) v[q2] = cmul(v[q2], tw.s2[q2 - 1]);
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) bufQ[q * kEx2Stride + q2 * 32 + t1] = v[q2];
  }
  __syncthreads();
  {
    const int q = tid & 7, t2 = (tid >> 3) & 3, q2 = tid >> 5;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = bufQ[q * kEx2Stride + q2 * 32 + t2 + 4 * r];
    dft8(v);
#pragma unroll
    for (int q3 = 1; q3 < 8; ++q3) v[q3] = cmul(v[q3], tw.s3[q3]);
#pragma unroll
    for (int q3 = 0; q3 < 8; ++q3) bufP[t2 * kEx3Stride + q + 8 * q2 + 64 * q3] = v[q3];
  }
  __syncthreads();
  // stage 4: radix 4 over the last index t2.  Each thread takes the adjacent pair p = 2 tid,
  // 2 tid + 1 so that bins k = p + 512 k3 leave as 16-byte stores (1 KiB per wave), fftshifted.
  {
    const float4* src = reinterpret_cast<const float4*>(bufP);
    const float4 a0 = src[tid], a1 = src[kEx3Stride / 2 + tid], a2 = src[kEx3Stride + tid], a3 = src[3 * kEx3Stride / 2 + tid];
    float2 x0 = make_float2(a0.x, a0.y), x1 = make_float2(a1.x, a1.y), x2 = make_float2(a2.x, a2.y), x3 = make_float2(a3.x, a3.y);
    float2 y0 = make_float2(a0.z, a0.w), y1 = make_float2(a1.z, a1.w), y2 = make_float2(a2.z, a2.w), y3 = make_float2(a3.z, a3.w);
    dft4(x0, x1, x2, x3);
    dft4(y0, y1, y2, y3);
    typedef float __attribute__((ext_vector_type(4))) vfloat4;
    vfloat4* dst = reinterpret_cast<vfloat4*>(out);        // element i holds bins 2i, 2i+1
#ifdef DABHIP_PROBE_NOSTORE
    if (x0.x + y1.y != 1.2345e30f) return;                // probe: the transform without its stores
#endif
    __builtin_nontemporal_store(vfloat4{x0.x, x0.y, y0.x, y0.y}, &dst[(tid + 512) & 1023]);          // k3 = 0: bin p       -> p + 1024
    __builtin_nontemporal_store(vfloat4{x1.x, x1.y, y1.x, y1.y}, &dst[(tid + 256 + 512) & 1023]);    // k3 = 1: bin p + 512
    __builtin_nontemporal_store(vfloat4{x2.x, x2.y, y2.x, y2.y}, &dst[(tid + 512 + 512) & 1023]);    // k3 = 2: bin p + 1024
    __builtin_nontemporal_store(vfloat4{x3.x, x3.y, y3.x, y3.y}, &dst[(tid + 768 + 512) & 1023]);    // k3 = 3: bin p + 1536
  }
  // no barrier here: the next symbol starts by writing the OTHER buffer (roles swap)
}

// Transform symbols [sym_begin, sym_end).  `parity` tells which LDS buffer plays which role first and is
// returned advanced, so that a fast run can be followed by a slow run without an extra barrier.
// Parity guard (k_parity.hip): the energy of the symbol about to be transformed.  Every wave leaves its part in LDS before
// the transform's barriers; thread 0 adds them up after the transform and stores the symbol's error bound.
struct EnergyOut {
  float* esum;           // LDS, 2 symbols x 4 waves
  float* delta_tf;       // this frame's 76 bounds
  float c;               // the guard level's constant (GuardArgs::c)
};
__device__ __forceinline__ void energy_part(const float2 (&v)[8], float* esum4)
{
  float e = 0.0f;
#pragma unroll
  for (int r = 0; r < 8; ++r) e += v[r].x * v[r].x + v[r].y * v[r].y;
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) e += __shfl_xor(e, sft);
  if ((threadIdx.x & 63) == 0) esum4[threadIdx.x >> 6] = e;
}
__device__ __forceinline__ void energy_store(const float* esum4, float* delta_tf, int sym, float c)
{
  // the wave parts are exact integers (< 2^24); added up as integers so that every kernel arrives at the very same float
  if (threadIdx.x == 0)
    delta_tf[sym] = c * sqrtf(static_cast<float>(static_cast<int>(esum4[0]) + static_cast<int>(esum4[1]) + static_cast<int>(esum4[2]) + static_cast<int>(esum4[3])));
}

template <bool kFast, bool kEnergy>
__device__ __forceinline__ int transform_symbols(GlobalU16 fast_src, const uint8_t* stream, const FrameView& view, int sym_begin,
                                                 int sym_end, int parity, float2* exA, float2* exB, const Twiddles& tw,
                                                 float2* __restrict__ out_tf, uint32_t nco_inc, const EnergyOut& eo)
{
  if (sym_begin >= sym_end) return parity;
  unsigned raw[8];
  load_symbol<kFast>(fast_src, stream, view, sym_begin, raw);
  for (int sym = sym_begin; sym < sym_end; ++sym) {
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = sample_f32(raw[r]);
    if (nco_inc) derotate(v, nco_inc, kNullSamples + kSymSamples * sym + kCpSamples + static_cast<int>(threadIdx.x));
    if (sym + 1 < sym_end) load_symbol<kFast>(fast_src, stream, view, sym + 1, raw);   // prefetch under the transform
    if (kEnergy) energy_part(v, eo.esum + 4 * (sym & 1));
    if (parity) fft2048_store(v, exB, exA, tw, out_tf + static_cast<size_t>(sym) * 2048);
    else fft2048_store(v, exA, exB, tw, out_tf + static_cast<size_t>(sym) * 2048);
    if (kEnergy) energy_store(eo.esum + 4 * (sym & 1), eo.delta_tf, sym, eo.c);
    parity ^= 1;
  }
  return parity;
}

// One workgroup: kSyms consecutive OFDM symbols (sym0 ..) of one transmission frame -> out_tf[sym * 2048 ..].
template <int kSyms, bool kEnergy>
__device__ __forceinline__ void fft_block(const uint8_t* stream, const FrameView& view, const int seg_end0, const int64_t seg_src0,
                                          const int sym0, float2* __restrict__ out_tf, const float2* __restrict__ tw_global,
                                          float2* exA, float2* exB, const int nco_hz, const EnergyOut& eo)
{
  const int tid = threadIdx.x;
  __shared__ float2 tw3[4 * 8];
  if (tid < 32) tw3[tid] = tw_global[64 * (tid >> 3) * (tid & 7)];
  Twiddles tw;
  {
    const int t1 = tid & 31, t2 = (tid >> 3) & 3;
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      tw.s1[q - 1] = tw_global[tid * q];
      tw.s2[q - 1] = tw_global[8 * t1 * q];
    }
    tw.s3 = tw3 + 8 * t2;
  }
  __syncthreads();
  // Symbols whose window lies inside what this call read from the stream take the contiguous path; the
  // rest (symbol 75 after a negative timing shift, frames right after a coarse resync) read through the view.
  const int sym_end = sym0 + kSyms;
  int nfast = 0;
  if (seg_src0 >= 0) {
    const int avail = (seg_end0 - 4096) / 2 - kNullSamples - kCpSamples;      // start sample of the last fitting window, relative
    if (seg_end0 >= 4096 && avail >= 0) nfast = min(kSymbolsPerTf, avail / kSymSamples + 1);
  }
  const int fast_end = max(sym0, min(sym_end, nfast));
  GlobalU16 src = reinterpret_cast<GlobalU16>(reinterpret_cast<uintptr_t>(stream + (seg_src0 >= 0 ? seg_src0 : 0)));
  // the common case gets a compile-time, EVEN trip count (buffer roles end where they started): fully unrolled,
  // prefetches hoisted.  For 19 symbols that is 18; the 19th is the only one that can straddle the stale tail.
  constexpr int kFixed = kSyms & ~1;
  int done = sym0;
  // software AFC (off in parity mode: nco_hz == 0): per-sample phase step as a 32-bit fraction of a turn
  const uint32_t nco_inc = nco_hz ? static_cast<uint32_t>(static_cast<int64_t>(llrint(nco_hz * (4294967296.0 / 2048000.0)))) : 0u;
  if (nco_inc == 0 && fast_end - sym0 >= kFixed) {
    unsigned raw[8];
    load_symbol<true>(src, stream, view, sym0, raw);
#pragma unroll
    for (int i = 0; i < kFixed; ++i) {
      float2 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = sample_f32(raw[r]);
      if (i + 1 < kFixed) load_symbol<true>(src, stream, view, sym0 + i + 1, raw);
      if (kEnergy) energy_part(v, eo.esum + 4 * ((sym0 + i) & 1));
      if (i & 1) fft2048_store(v, exB, exA, tw, out_tf + static_cast<size_t>(sym0 + i) * 2048);
      else fft2048_store(v, exA, exB, tw, out_tf + static_cast<size_t>(sym0 + i) * 2048);
      if (kEnergy) energy_store(eo.esum + 4 * ((sym0 + i) & 1), eo.delta_tf, sym0 + i, eo.c);
    }
    done = sym0 + kFixed;
  }
  const int parity = transform_symbols<true, kEnergy>(src, stream, view, done, max(done, fast_end), 0, exA, exB, tw, out_tf, nco_inc, eo);
  transform_symbols<false, kEnergy>(nullptr, stream, view, max(done, fast_end), sym_end, parity, exA, exB, tw, out_tf, nco_inc, eo);
}

// grid = (4 * nframes); frame j of the launch is frames[first + j] = {stream, call}
// kEnergy: also leave the parity guard's per-symbol error bounds (delta[(first + j) * 76 + symbol]); the plain variant is the
// one the roofline figure is measured on.
template <bool kEnergy>
__global__ __launch_bounds__(kThreads, 4) void ofdm_fft_kernel(const uint8_t* const* __restrict__ iq,
                                                               const CallDesc* __restrict__ descs, int max_calls,
                                                               const int2* __restrict__ frames, int first,
                                                               float2* __restrict__ spectra,
                                                               const float2* __restrict__ tw_global, float* __restrict__ delta, float delta_c)
{
  __shared__ __attribute__((aligned(16))) float2 exA[kExSize];
  __shared__ __attribute__((aligned(16))) float2 exB[kExSize];
  __shared__ FrameView view;
  __shared__ float esum[kEnergy ? 8 : 1];
  const int j = blockIdx.x >> 2, part = blockIdx.x & 3;
  const int2 fr = frames[first + j];
  const CallDesc* desc = descs + static_cast<size_t>(fr.x) * max_calls + fr.y;
  if (threadIdx.x == 0) view = desc->view;
  const EnergyOut eo{esum, kEnergy ? delta + static_cast<size_t>(first + j) * kSymbolsPerTf : nullptr, delta_c};
  fft_block<kSymPerBlock, kEnergy>(iq[fr.x], view, desc->view.seg_end[0], desc->view.seg_src[0], part * kSymPerBlock,
                                   spectra + static_cast<size_t>(j) * (kSymbolsPerTf * 2048), tw_global, exA, exB, desc->nco_hz, eo);
}

// FIC pre-pass: only the phase reference symbol and the three FIC symbols (0..3) of every frame, so that the FIC
// can be decoded -- and the host control plane run -- while the full OFDM stage is still on the GPU.
// grid = nframes; output [frame][4][2048].
__global__ __launch_bounds__(kThreads, 4) void fic_fft_kernel(const uint8_t* const* __restrict__ iq,
                                                              const CallDesc* __restrict__ descs, int max_calls,
                                                              const int2* __restrict__ frames, int first,
                                                              float2* __restrict__ spectra4,
                                                              const float2* __restrict__ tw_global)
{
  __shared__ __attribute__((aligned(16))) float2 exA[kExSize];
  __shared__ __attribute__((aligned(16))) float2 exB[kExSize];
  __shared__ FrameView view;
  const int j = blockIdx.x;
  const int2 fr = frames[first + j];
  const CallDesc* desc = descs + static_cast<size_t>(fr.x) * max_calls + fr.y;
  if (threadIdx.x == 0) view = desc->view;
  fft_block<4, false>(iq[fr.x], view, desc->view.seg_end[0], desc->view.seg_src[0], 0, spectra4 + static_cast<size_t>(j) * (4 * 2048), tw_global,
                      exA, exB, desc->nco_hz, EnergyOut{nullptr, nullptr, 0.0f});
}

// ---- K2b ----------------------------------------------------------------------------------
// grid = kDemapGroups * nframes: workgroup g handles kDemapSyms consecutive data symbols of frame g / kDemapGroups
// (it re-reads one previous symbol as the differential reference).  Carrier c (0..1535, ascending frequency) sits at fftshifted bin
// 256 + c (c < 768) or 257 + c.
//
// MSC output layout (kPlanar): the time de-interleaver (misc.c:29-39) reads bit i of the CIF
// that lies map[i & 15] CIFs after the oldest one.  Instead of gathering from 16 rows later,
// every transmitted CIF n is scattered here: its bits with i & 15 == r form "plane r" of
// LOGICAL row n - map[r] (3456 bits = 108 words per plane, plane after plane: a symbol then leaves as 16 runs
// of 6 consecutive words instead of 96 isolated ones).  A complete logical row is then exactly the reference's cif_time_deinterleaved:
// out[i] = bit ((i >> 4) & 31) of row word ((i >> 9) * 16 + (i & 15)).
// kBits = 1: hard decisions (the reference); kBits = 4: signed 4-bit soft values (extension):
// value = round(7 x / mean|x|) clamped to +-7 (soft_scale, device_types.hpp) with x = Re(cur conj(prev)) for the first bit and Im(cur conj(prev)) for
// the second, i.e. positive = "bit 0"; the mean is taken over the 3072 components of the OFDM symbol.
template <bool kPlanar, int kBits>
__global__ __launch_bounds__(kThreads) void demap_kernel(const float2* __restrict__ spectra, int syms_per_tf, int group_syms,
                                                         int groups_per_tf, int first,
                                                         const int* __restrict__ frame_slot,
                                                         const int* __restrict__ frame_cif_row,
                                                         const uint16_t* __restrict__ qpsk_of_carrier,
                                                         uint32_t* __restrict__ fic_bits, uint32_t* __restrict__ msc_bits, const GuardArgs guard)
{
  // The decisions of a symbol are written as bytes where the output word wants them: byte kPer t + b = value b of output
  // word t (kPer = values per word), so the packing threads read their word's bytes contiguously (the straightforward
  // "bits[i]" array made 96 lanes hit 4 LDS banks: 75 % of this kernel's LDS cycles were bank conflicts).  Two arrays
  // alternate from symbol to symbol: one barrier per symbol.
  constexpr int kPer = 32 / kBits;                        // received values per 32-bit word
  __shared__ __attribute__((aligned(16))) uint8_t dec[2][kBitsPerSym];
  const int tid = threadIdx.x;
  const int j = blockIdx.x / groups_per_tf, grp = blockIdx.x % groups_per_tf;
  const int slot = frame_slot[first + j];                 // TF slot (FIC rows, FIB records)
  const int cif_row = frame_cif_row[first + j];           // row of this TF's first CIF
  const float2* tf = spectra + static_cast<size_t>(j) * (syms_per_tf * 2048);
  int bin[6], qk[6];
  float2 prev[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    const int c = tid + 256 * m;
    bin[m] = c < 768 ? 256 + c : 257 + c;
    qk[m] = qpsk_of_carrier[c];
    prev[m] = tf[(group_syms * grp) * 2048 + bin[m]];
  }
  for (int l = group_syms * grp + 1; l <= group_syms * grp + group_syms; ++l) {
    uint8_t* d = dec[l & 1];
    const bool natural = l <= 3 || !kPlanar;
    // Batch path (kPlanar): the FIC symbols were demapped -- and guarded -- by the pre-pass, and the FIC decode may be reading
    // those rows on the side stream right now (Engine::fic_decode_slots_async), so this launch neither stores nor lists them.
    const bool fic_done = kPlanar && l <= 3;
    float re[6], im[6];
    // parity guard (k_parity.hip; hard decisions only): error bounds of this symbol's and the previous symbol's bins
    const bool guarded = kBits == 1 && guard.delta != nullptr && !fic_done;
    const float dc = guarded ? guard.delta[static_cast<size_t>(first + j) * guard.delta_stride + l] : 0.0f;
    const float dp = guarded ? guard.delta[static_cast<size_t>(first + j) * guard.delta_stride + l - 1] : 0.0f;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const float2 cur = tf[l * 2048 + bin[m]];
      re[m] = diff_re(cur.x, cur.y, prev[m].x, prev[m].y);   // Re(cur conj(prev))
      im[m] = diff_im(cur.x, cur.y, prev[m].x, prev[m].y);   // -Im(cur conj(prev)), as stored at input_sdr.c:139-143
      if (guarded) {
        const float n1c = fabsf(cur.x) + fabsf(cur.y), n1p = fabsf(prev[m].x) + fabsf(prev[m].y);
        const float thr = guard.per_bin ? guard_bin_threshold(cur.x, cur.y, prev[m].x, prev[m].y, (bin[m] + 1024) & 2047, dc, dp, guard.prod)
                                    : guard_threshold(n1c, n1p, dc, dp, guard.prod);
        if (fminf(fabsf(re[m]), fabsf(im[m])) < thr) {
          const unsigned at = atomicAdd(guard.counter, 1u);
          if (at < guard.cap) guard.list[at] = make_uint4(static_cast<unsigned>(first + j), (static_cast<unsigned>(l) << 16) | static_cast<unsigned>((bin[m] + 1024) & 2047), 0u, 0u);
        }
      }
      prev[m] = cur;
    }
    // soft decisions: the scale comes from the two symbols' sample energies (soft_scale, device_types.hpp), like in the one-kernel stage
    const float scale = kBits != 1 ? soft_scale(guard.delta[static_cast<size_t>(first + j) * guard.delta_stride + l],
                                                guard.delta[static_cast<size_t>(first + j) * guard.delta_stride + l - 1])
                                   : 0.0f;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      // position i of a value inside the symbol -> its byte: natural order i, or plane i & 15, value i >> 4 of that plane
      const int i0 = qk[m], i1 = 1536 + qk[m];
      const int a0 = natural ? i0 : ((i0 & 15) * (6 * kBits) + (i0 >> 4) / kPer) * kPer + (i0 >> 4) % kPer;
      const int a1 = natural ? i1 : ((i1 & 15) * (6 * kBits) + (i1 >> 4) / kPer) * kPer + (i1 >> 4) % kPer;
      if (kBits == 1) {
        d[a0] = (re[m] > 0.0f) ? 0 : 1;                  // input_sdr.c:157
        d[a1] = (im[m] > 0.0f) ? 1 : 0;                  // input_sdr.c:158
      } else {
        const int q0 = max(-7, min(7, __float2int_rn(re[m] * scale)));     // > 0: first bit is 0
        const int q1 = max(-7, min(7, __float2int_rn(-im[m] * scale)));    // stored im > 0 means bit 1
        d[a0] = static_cast<uint8_t>(q0 & 15);
        d[a1] = static_cast<uint8_t>(q1 & 15);
      }
    }
    __syncthreads();
    for (int t = tid; t < 96 * kBits; t += kThreads) {
      uint32_t w;
      if (kBits == 1) {
        const uint4* p = reinterpret_cast<const uint4*>(d + 32 * t);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        w = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) w |= ((x[k] * 0x01020408u) >> 24 & 15u) << (4 * k);   // bytes (0/1) b0..b3 -> b0 | b1<<1 | b2<<2 | b3<<3
      } else {
        const uint2 x = *reinterpret_cast<const uint2*>(d + 8 * t);
        const uint32_t y0 = (x.x | (x.x >> 4)) & 0x00ff00ffu, y1 = (x.y | (x.y >> 4)) & 0x00ff00ffu;   // nibbles of bytes 0,1 and 2,3 joined
        w = ((y0 | (y0 >> 8)) & 0xffffu) | (((y1 | (y1 >> 8)) & 0xffffu) << 16);
      }
      if (natural) {
        if (l <= 3) { if (!fic_done) fic_bits[static_cast<size_t>(slot) * (288 * kBits) + (l - 1) * (96 * kBits) + t] = w; }
        else msc_bits[static_cast<size_t>(cif_row) * (1728 * kBits) + (l - 4) * (96 * kBits) + t] = w;
      } else {
        const int q = (l - 4) / 18, sidx = (l - 4) % 18;          // CIF within the TF, symbol within the CIF
        const int r = t / (6 * kBits), wq = t % (6 * kBits);      // plane (i & 15) and word within this symbol's 192 values
        const int delay = static_cast<int>(__brev(static_cast<unsigned>(r)) >> 28);   // map[r], misc.c:32
        msc_bits[static_cast<size_t>(cif_row + q - delay) * (1728 * kBits) + r * (108 * kBits) + sidx * 6 * kBits + wq] = w;
      }
    }
    // no second barrier: the next symbol writes the other array, and two barriers lie between two uses of one array
  }
}

}  // namespace

hipError_t launch_ofdm_fft(const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first,
                           int nframes, float2* spectra, const float2* tw, hipStream_t stream, float* delta, float delta_c)
{
  if (nframes <= 0) return hipSuccess;
  if (delta)
    hipLaunchKernelGGL(ofdm_fft_kernel<true>, dim3(4 * nframes), dim3(kThreads), 0, stream, iq, descs, max_calls, frames, first, spectra, tw, delta, delta_c);
  else
    hipLaunchKernelGGL(ofdm_fft_kernel<false>, dim3(4 * nframes), dim3(kThreads), 0, stream, iq, descs, max_calls, frames, first, spectra, tw, delta, delta_c);
  return hipGetLastError();
}

hipError_t launch_demap(bool planar, int soft_bits, const float2* spectra, int first, int nframes, const int* frame_slot,
                        const int* frame_cif_row, const uint16_t* qpsk_of_carrier, uint32_t* fic_bits, uint32_t* msc_bits,
                        const GuardArgs& guard, hipStream_t stream)
{
  if (nframes <= 0) return hipSuccess;
  const dim3 grid(kDemapGroups * nframes), block(kThreads);
  if (planar && soft_bits)
    hipLaunchKernelGGL((demap_kernel<true, 4>), grid, block, 0, stream, spectra, kSymbolsPerTf, kDemapSyms, kDemapGroups, first, frame_slot,
                       frame_cif_row, qpsk_of_carrier, fic_bits, msc_bits, guard);
  else if (planar)
    hipLaunchKernelGGL((demap_kernel<true, 1>), grid, block, 0, stream, spectra, kSymbolsPerTf, kDemapSyms, kDemapGroups, first, frame_slot,
                       frame_cif_row, qpsk_of_carrier, fic_bits, msc_bits, guard);
  else
    hipLaunchKernelGGL((demap_kernel<false, 1>), grid, block, 0, stream, spectra, kSymbolsPerTf, kDemapSyms, kDemapGroups, first, frame_slot,
                       frame_cif_row, qpsk_of_carrier, fic_bits, msc_bits, guard);
  return hipGetLastError();
}

// FIC pre-pass: 4-symbol spectra -> FIC rows only
hipError_t launch_fic_prepass(int soft_bits, const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first,
                              int nframes, float2* spectra4, const float2* tw, const int* frame_slot, const uint16_t* qpsk_of_carrier,
                              uint32_t* fic_bits, const GuardArgs& guard, hipStream_t stream)
{
  if (nframes <= 0) return hipSuccess;
  hipLaunchKernelGGL(fic_fft_kernel, dim3(nframes), dim3(kThreads), 0, stream, iq, descs, max_calls, frames, first, spectra4, tw);
  if (soft_bits)
    hipLaunchKernelGGL((demap_kernel<false, 4>), dim3(nframes), dim3(kThreads), 0, stream, spectra4, 4, 3, 1, first, frame_slot, frame_slot,
                       qpsk_of_carrier, fic_bits, static_cast<uint32_t*>(nullptr), guard);
  else
    hipLaunchKernelGGL((demap_kernel<false, 1>), dim3(nframes), dim3(kThreads), 0, stream, spectra4, 4, 3, 1, first, frame_slot, frame_slot,
                       qpsk_of_carrier, fic_bits, static_cast<uint32_t*>(nullptr), guard);
  return hipGetLastError();
}

}  // namespace dabhip
