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
#ifndef DABHIP_EX2_STRIDE
#define DABHIP_EX2_STRIDE 258
#endif
#ifndef DABHIP_EX3_STRIDE
#define DABHIP_EX3_STRIDE 520
#endif
constexpr int kEx2Stride = DABHIP_EX2_STRIDE;    // [q] stride of exchange 2 (8*32 + 2: the 16 lanes (q, t'' & 1) of a read land in 16 different bank pairs; 260 measured 6.2e8 conflict cycles per 16 launches, 258 none)
constexpr int kEx3Stride = DABHIP_EX3_STRIDE;    // [t''] stride of exchange 3 (512 + 8)
constexpr int kExSize = (8 * kEx2Stride > 4 * kEx3Stride ? 8 * kEx2Stride : 4 * kEx3Stride);   // float2 per exchange buffer
constexpr int kDemapSyms = 5;                    // data symbols per demap workgroup (75 = 5 x 15)
constexpr int kDemapGroups = 75 / kDemapSyms;

// Complex arithmetic on 2-element vectors: the compiler maps these onto the packed fp32 instructions of CDNA3/4
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), swizzles and sign flips folded into their op_sel / neg modifiers.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f V(float2 a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ float2 F(v2f a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// b + (-i) d and b - (-i) d: the quarter turn rides in the multiplier (1, -1), no separate negation
__device__ __forceinline__ v2f add_mi(v2f b, v2f d) { return vfma(d.yx, v2f{1.0f, -1.0f}, b); }
__device__ __forceinline__ v2f sub_mi(v2f b, v2f d) { return vfma(d.yx, v2f{-1.0f, 1.0f}, b); }
// complex product a b = a.xx b + ((a.yy (-1, 1)) b.yx)
__device__ __forceinline__ v2f vmul(v2f a, v2f b) { return vfma(a.yy * v2f{-1.0f, 1.0f}, b.yx, a.xx * b); }
// complex product, (fma(a.x, b.x, -(a.y b.y)), fma(a.x, b.y, a.y b.x)), in one of two forms with the SAME roundings (so kernels may
// differ in the form and still agree bit for bit):
//   DABHIP_CMUL_SCALAR: two multiplies and two fused multiply-adds of the plain kind (2.6 clocks each per wave on gfx950);
//   default: two packed instructions (5.0 .. 5.6 clocks each), the swap and the sign riding in op_sel / neg_lo -- the compiler's own
//   rendering of vmul() spends a third packed instruction on the (-1, 1) factor (tools/ubench/valu_rates.hip for the rates).
#ifdef DABHIP_CMUL_SCALAR
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
  return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}
#elif defined(DABHIP_CMUL_VMUL)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return F(vmul(V(a), V(b))); }   // probe: different roundings
#else
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
  const v2f va = V(a), vb = V(b);
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(va), "v"(vb));        // (-a.y b.y, a.y b.x)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));   // (a.x b.x, a.x b.y) + t
  return F(r);
}
#endif
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return F(V(a) + V(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return F(V(a) - V(b)); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // * (-i)

// 8-point forward DFT in registers: v[q] <- sum_r v[r] exp(-2 pi i r q / 8)
__device__ __forceinline__ void dft8(float2 (&vv)[8])
{
  constexpr float h = 0.70710678118654752440f;
  v2f v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = V(vv[i]);
  const v2f a0 = v[0] + v[4], a4 = v[0] - v[4];
  const v2f a1 = v[1] + v[5], d5 = v[1] - v[5];
  const v2f a2 = v[2] + v[6], a6 = v[2] - v[6];
  const v2f a3 = v[3] + v[7], d7 = v[3] - v[7];
  const v2f a5 = vfma(d5.yx, v2f{h, -h}, d5 * h);               // * exp(-i pi/4):  h (x + y, y - x)
  const v2f a7 = vfma(d7.yx, v2f{h, -h}, d7 * -h);              // * exp(-3i pi/4): h (y - x, -x - y)
  const v2f b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, d13 = a1 - a3;
  const v2f b4 = add_mi(a4, a6), b6 = sub_mi(a4, a6), b5 = a5 + a7, d57 = a5 - a7;   // a6 enters turned by -i
  vv[0] = F(b0 + b1); vv[4] = F(b0 - b1); vv[2] = F(add_mi(b2, d13)); vv[6] = F(sub_mi(b2, d13));
  vv[1] = F(b4 + b5); vv[5] = F(b4 - b5); vv[3] = F(add_mi(b6, d57)); vv[7] = F(sub_mi(b6, d57));
}

__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3)
{
  const v2f d0 = V(x0) + V(x2), d2 = V(x0) - V(x2), d1 = V(x1) + V(x3), d13 = V(x1) - V(x3);
  x0 = F(d0 + d1); x2 = F(d0 - d1); x1 = F(add_mi(d2, d13)); x3 = F(sub_mi(d2, d13));
}

__device__ __forceinline__ int view_byte(const uint8_t* stream, const FrameView& v, int p) { return frame_byte(stream, v, p); }
__device__ __forceinline__ float rail(int byte) { return static_cast<float>(static_cast<int8_t>(static_cast<uint8_t>(byte - 127))); }
// one IQ sample, I in byte 0 and Q in byte 1 of w (upper half clear): (float)(int8)(byte - 127) each (input_sdr.c:60-63).
// - 127 = + 0x81 mod 256, added without a carry into the byte that is read; the conversion sign-extends that byte itself
// (two instructions per component: the cast chain above compiles to v_add_u16 + v_bfe_i32 + v_cvt_f32_i32, three of the slower kind)
__device__ __forceinline__ float2 sample_f32(unsigned w)
{
  const unsigned ta = w + 0x81u, tb = w + 0x8100u;
  float2 v;
  asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(v.x) : "v"(ta));
  asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(v.y) : "v"(tb));
  return v;
}

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
