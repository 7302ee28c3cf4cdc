// fft_core.hpp — register/LDS building blocks of the 2048-point OFDM transform shared by k_fft.hip (K2) and
// k_fused.hip (the fused FFT + demap variant): radix-8/4 butterflies, per-thread twiddles, symbol loads.
#pragma once

#include <hip/hip_runtime.h>

#include "dab_tables.hpp"
#include "device_types.hpp"

namespace dabhip {
namespace {

constexpr int kThreads = 256;
constexpr int kSymPerBlock = 19;                 // 76 symbols = 4 workgroups x 19
constexpr int kEx2Stride = 260;                  // [q] stride of exchange 2 (8*32 + 4: bank skew)
constexpr int kEx3Stride = 520;                  // [t''] stride of exchange 3 (512 + 8)
constexpr int kExSize = 2080;                    // float2 per exchange buffer
constexpr int kDemapSyms = 5;                    // data symbols per demap workgroup (75 = 5 x 15)
constexpr int kDemapGroups = 75 / kDemapSyms;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // * (-i)

// 8-point forward DFT in registers: v[q] <- sum_r v[r] exp(-2 pi i r q / 8)
__device__ __forceinline__ void dft8(float2 (&v)[8])
{
  constexpr float h = 0.70710678118654752440f;
  float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
  float2 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
  float2 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
  float2 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  a5 = make_float2(h * (a5.x + a5.y), h * (a5.y - a5.x));      // * exp(-i pi/4)
  a6 = mul_mi(a6);                                              // * exp(-i pi/2)
  a7 = make_float2(h * (a7.y - a7.x), -h * (a7.x + a7.y));     // * exp(-3i pi/4)
  float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = mul_mi(csub(a1, a3));
  float2 b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = cadd(a5, a7), b7 = mul_mi(csub(a5, a7));
  v[0] = cadd(b0, b1); v[4] = csub(b0, b1); v[2] = cadd(b2, b3); v[6] = csub(b2, b3);
  v[1] = cadd(b4, b5); v[5] = csub(b4, b5); v[3] = cadd(b6, b7); v[7] = csub(b6, b7);
}

__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3)
{
  const float2 d0 = cadd(x0, x2), d2 = csub(x0, x2), d1 = cadd(x1, x3), d3 = mul_mi(csub(x1, x3));
  x0 = cadd(d0, d1); x2 = csub(d0, d1); x1 = cadd(d2, d3); x3 = csub(d2, d3);
}

__device__ __forceinline__ int view_byte(const uint8_t* stream, const FrameView& v, int p)
{
  int i = 0;
  while (i < v.nseg - 1 && p >= v.seg_end[i]) ++i;
  const int64_t s = v.seg_src[i];
  return s < 0 ? 0 : stream[s + p];
}
__device__ __forceinline__ float rail(int byte) { return static_cast<float>(static_cast<int8_t>(static_cast<uint8_t>(byte - 127))); }

// software AFC: de-rotate the 8 samples of this thread by exp(-2 pi i nco n / fs); phase kept as a 32-bit fraction of a
// turn (inc = nco / fs * 2^32 per sample), so it never loses precision over the 196,608 samples of a frame
__device__ __forceinline__ void derotate(float2 (&v)[8], uint32_t inc, int first_sample)
{
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t turns = 0u - inc * static_cast<uint32_t>(first_sample + 256 * r);
    float sn, cs;
    sincospif(static_cast<float>(static_cast<int32_t>(turns)) * (1.0f / 2147483648.0f), &sn, &cs);   // angle = pi * turns / 2^31
    v[r] = make_float2(v[r].x * cs - v[r].y * sn, v[r].x * sn + v[r].y * cs);
  }
}

// Twiddles of the three inter-stage multiplications depend only on the thread index, so
// each thread keeps its 21 factors in registers for all symbols it transforms.
struct Twiddles {
  float2 s1[7], s2[7];
  const float2* s3;      // stage-3 factors W_32^(t2 q3) depend on 2 bits of the thread index only: a 4 x 8 table in LDS
};

typedef const __attribute__((address_space(1))) uint16_t* GlobalU16;

template <bool kFast>
__device__ __forceinline__ void load_symbol(GlobalU16 fast_src, const uint8_t* stream, const FrameView& view, int sym, unsigned (&raw)[8])
{
  const int tid = threadIdx.x;
  const int start = 2 * (kNullSamples + kSymSamples * sym + kCpSamples);   // byte offset in the frame buffer
  if (kFast) {
    GlobalU16 src = fast_src + (start >> 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) raw[r] = src[tid + 256 * r];
  } else {   // window reaches into the stale tail of the reference's frame buffer
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int p = start + 2 * (tid + 256 * r);
      raw[r] = static_cast<unsigned>(view_byte(stream, view, p)) | (static_cast<unsigned>(view_byte(stream, view, p + 1)) << 8);
    }
  }
}

}  // namespace
}  // namespace dabhip
