// fft64.hpp — double-precision DFTs held in LDS, shared by the sync scan (k_sync.hip: the reference's FFTW calls in
// sdr_sync.c are fp64) and by the decision audit (k_parity.hip: the fp64 yardstick the fp32 OFDM stage is measured against).
#pragma once

#include <hip/hip_runtime.h>

namespace dabhip {
namespace {

constexpr int kFft64Threads = 512;   // threads of the workgroup that runs these transforms

__device__ __forceinline__ unsigned brev(unsigned x, int bits) { return __brev(x) >> (32 - bits); }

// Where element i of a transform buffer (or of the twiddle table) sits in LDS.  The later levels of a transform walk the buffer
// with strides of 4, 32, ... elements of 16 bytes and the twiddle table with strides of 8 .. 128: left in place, the eight lanes
// that share an LDS cycle would meet in the same banks (two thirds of K1's LDS cycles were bank conflicts).  XOR-ing the low
// three index bits with bits 3..5 and 6..8 keeps aligned groups of eight together and spreads every one of those strides.
// Buffers start at multiples of 512 elements, so the swizzle commutes with the buffer offsets.
__device__ __forceinline__ int lds_at(int i) { return i ^ ((i >> 3) & 7) ^ ((i >> 6) & 7); }

// nbatch independent DFTs of size N = 1 << LOGN stored back to back in LDS, radix-2 decimation in frequency, in
// place; X[k] ends up at index brev(k).  sign = -1 forward.  Element i of buf and of tw2048 (an LDS copy of exp(2 pi i k / 2048),
// k < 1024) is at lds_at(i).
// The scan is a chain of dependent LDS round trips with only 2 waves per SIMD to hide them, so L consecutive
// radix-2 levels are fused: a thread holds the 2^L points it needs in registers, runs the L levels on them (the same
// butterflies, in the same order of operations as level-by-level radix 2) and meets the others at ONE barrier.
// V = double2 (the reference's arithmetic) or float2 (sync_verify_kernel's first pass, whose arg-max is re-done in double when it is
// not clear-cut): the same butterflies in the same order.
template <class V>
struct ScalarOf;
template <>
struct ScalarOf<double2> { using type = double; };
template <>
struct ScalarOf<float2> { using type = float; };

template <int LOGN, int L, class V>
__device__ __forceinline__ void dif_levels(V* buf, int nbatch, int s, typename ScalarOf<V>::type sign, const V* tw2048)
{
  using R = typename ScalarOf<V>::type;
  constexpr int N = 1 << LOGN, RADIX = 1 << L;
  const int ms = N >> s;                         // size of the sub-transforms at level s
  const int q = ms >> L;                         // distance between the points one thread holds
  constexpr int per_batch = N >> L;
  for (int idx = threadIdx.x; idx < nbatch * per_batch; idx += kFft64Threads) {
    const int batch = idx / per_batch, w = idx % per_batch;
    const int blk = w / q, k = w % q;
    const int e0 = batch * N + blk * ms + k;
    V r[RADIX];
#pragma unroll
    for (int j = 0; j < RADIX; ++j) r[j] = buf[lds_at(e0 + j * q)];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int span = RADIX >> (l + 1);         // partner distance in units of q
      const int half = ms >> (l + 1);
      const int twstep = 1024 / half;
#pragma unroll
      for (int j = 0; j < RADIX; ++j) {
        if (j & span) continue;
        const int pos = k + (j & (span - 1)) * q;          // index of the butterfly inside its sub-transform
        const V A = r[j], B = r[j + span];
        V tw = tw2048[lds_at(pos * twstep)];
        tw.y *= sign;
        const R dr = A.x - B.x, di = A.y - B.y;
        r[j].x = A.x + B.x;
        r[j].y = A.y + B.y;
        r[j + span].x = dr * tw.x - di * tw.y;
        r[j + span].y = dr * tw.y + di * tw.x;
      }
    }
#pragma unroll
    for (int j = 0; j < RADIX; ++j) buf[lds_at(e0 + j * q)] = r[j];
  }
  __syncthreads();
}

template <int LOGN, int... Ls, class V>
__device__ __forceinline__ void dft_dif(V* buf, int nbatch, typename ScalarOf<V>::type sign, const V* tw2048)
{
  int s = 0;
  ((dif_levels<LOGN, Ls>(buf, nbatch, s, sign, tw2048), s += Ls), ...);
}

}  // namespace
}  // namespace dabhip
