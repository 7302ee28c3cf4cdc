// k_sync.hip — K1: per-stream synchronisation scan.
//
// Replaces, for every 262144-byte call of sdr_demod (input_sdr.c:27-112): the FIFO
// bookkeeping of sdr_fifo.c:26-61 (in closed form, as views into the resident IQ stream),
// dab_coarse_time_sync (sdr_sync.c:34-68), dab_fine_time_sync (:71-202),
// dab_coarse_freq_sync_2 (:205-258) and dab_fine_freq_corr (:259-302).
//
// The chain is sequential per stream (the timing correction found in TF n positions TF
// n+1) and independent across streams, so one 512-thread workgroup owns one stream for
// the whole scan: no inter-workgroup traffic, B workgroups in flight.  Arithmetic is fp64
// like the reference's FFTW calls (two 2048-point DFTs, one 1536-point and 29 128-point
// inverse DFTs per TF); arg-max decisions use the reference's float compare, first hit wins.
// Kernels: sync_scan_kernel (the chain; <true> = only what the next call depends on, the rest left to
// sync_verify32_kernel / sync_verify_kernel / sync_carry_kernel, which run over all calls at once),
// sync_ahead_kernel (small batches: the chain's estimators for every remaining call at every start
// position near the predicted one, so that the chain only looks them up).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "dab_tables.hpp"
#include "device_types.hpp"
#include "fft64.hpp"
#include "fifo_view.hpp"
#include "kernels.hpp"

namespace dabhip {
namespace {

constexpr int kThreads = kFft64Threads;   // 512; one workgroup per stream: more threads = shorter butterfly stages

// measurement build only (tools/sync_times.py, -DDABHIP_SYNC_TIMES=1): time stamps of the chain's phases, stream 0, the first 96 calls
#ifndef DABHIP_SYNC_TIMES
#define DABHIP_SYNC_TIMES 0
#endif
// measurement build only (tools/build_variant_sync.sh notail "-DDABHIP_K1_TAIL=0"): the chain without its tail-byte work (wrong results after any short
// read) -- what carrying the frame buffer's last 1536 bytes costs per call
#ifndef DABHIP_K1_TAIL
#define DABHIP_K1_TAIL 1
#endif
#if DABHIP_SYNC_TIMES
__device__ unsigned long long g_sync_times[96 * 16];
__device__ volatile int g_sync_call;     // volatile: read back through the vector path (a scalar load may be served a stale line)
#define SYNC_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_sync_call < 96) g_sync_times[g_sync_call * 16 + (i)] = wall_clock64(); } while (0)
#else
#define SYNC_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ int view_byte(const uint8_t* stream, const FrameView& v, int p) { return frame_byte(stream, v, p); }
// one IQ sample (I at the even byte p, Q at p + 1): segment boundaries and sources are even, so both bytes
// come from the same segment and one 2-byte load fetches them
// nco_hz != 0 (software AFC only): the sample is de-rotated by exp(-2 pi i nco n / fs), n = sample index in the frame
__device__ __forceinline__ double2 view_sample(const uint8_t* stream, const FrameView& v, int p, int nco_hz)
{
  const unsigned w = frame_u16(stream, v, p);
  const double2 x = make_double2(static_cast<int8_t>(static_cast<uint8_t>((w & 0xff) - 127)), static_cast<int8_t>(static_cast<uint8_t>((w >> 8) - 127)));
  if (nco_hz == 0) return x;
  double sn, cs;
  sincospi(-2.0 * nco_hz * (p >> 1) / 2048000.0, &sn, &cs);
  return make_double2(x.x * cs - x.y * sn, x.x * sn + x.y * cs);
}
// u8 -> s8 with DC offset 127 and int8 wrap (input_sdr.c:60-63)
__device__ __forceinline__ int rail(int byte) { return static_cast<int>(static_cast<int8_t>(static_cast<uint8_t>(byte - 127))); }

// Buffer bytes [p0, p1) of the frame almost always lie inside what this call read from the stream (segment 0 of the view);
// the loads below then go straight to the stream, unrolled, instead of through the per-sample segment search (whose loop
// keeps the compiler from overlapping the loads: one memory latency per sample).
__device__ __forceinline__ const uint8_t* contiguous_window(const uint8_t* stream, const FrameView& v, int p0, int p1)
{
  return (v.seg_src[0] >= 0 && p1 <= v.seg_end[0]) ? stream + v.seg_src[0] + p0 : nullptr;
}
__device__ __forceinline__ double2 sample_of(unsigned w)
{
  return make_double2(static_cast<int8_t>(static_cast<uint8_t>((w & 0xff) - 127)), static_cast<int8_t>(static_cast<uint8_t>((w >> 8) - 127)));
}
// the same loads issued early (registers), for use after some other work: the chain kernel fetches the samples of the fine time
// search while the null-symbol test runs
template <int kCount>
struct Prefetched {
  bool ok;
  unsigned w[kCount / kThreads];
};
template <int kCount>
__device__ __forceinline__ Prefetched<kCount> prefetch_samples(const uint8_t* stream, const FrameView& view, int p0, int nco)
{
  Prefetched<kCount> pf;
  const uint8_t* win = nco == 0 ? contiguous_window(stream, view, p0, p0 + 2 * kCount) : nullptr;
  pf.ok = win != nullptr;
  if (pf.ok) {
    const uint16_t* src = reinterpret_cast<const uint16_t*>(win);
#pragma unroll
    for (int i = 0; i < kCount / kThreads; ++i) pf.w[i] = src[threadIdx.x + i * kThreads];
  }
  return pf;
}
// count (a multiple of kThreads) samples starting at buffer byte p0 -> transform buffer dst, elements 0 .. count (fft64.hpp: lds_at)
template <int kCount>
__device__ __forceinline__ void load_samples(const uint8_t* stream, const FrameView& view, int p0, int nco, double2* dst)
{
  const int tid = threadIdx.x;
  const uint8_t* win = nco == 0 ? contiguous_window(stream, view, p0, p0 + 2 * kCount) : nullptr;
  if (win) {
    const uint16_t* src = reinterpret_cast<const uint16_t*>(win);
    unsigned w[kCount / kThreads];
#pragma unroll
    for (int i = 0; i < kCount / kThreads; ++i) w[i] = src[tid + i * kThreads];
#pragma unroll
    for (int i = 0; i < kCount / kThreads; ++i) dst[lds_at(tid + i * kThreads)] = sample_of(w[i]);
  } else {
    for (int n = tid; n < kCount; n += kThreads) dst[lds_at(n)] = view_sample(stream, view, p0 + 2 * n, nco);
  }
}

constexpr int kWaves = kThreads / 64;
struct Red {
  float fv[kWaves];
  int iv[kWaves];
  double dv[kWaves];
};

// Wave-level reductions on DPP moves (no LDS round trips: as __shfl_xor steps, i.e. ds_bpermute, the twelve dependent permutes of
// one arg-max were half a microsecond of every call of the chain).  The value of the lane a DPP control selects; lanes the row mask
// excludes keep their own.  After the six steps lane 63 holds the wave's result.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_from(int own)
{
  return __builtin_amdgcn_update_dpp(own, own, kCtrl, kRowMask, 0xf, false);
}
template <typename F>
__device__ __forceinline__ void wave_reduce_steps(F&& step)
{
  step(std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
  step(std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
  step(std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});   // row_half_mirror
  step(std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});   // row_mirror: rows of 16 done
  step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1 and 3
  step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 into rows 2 and 3
}

// arg-max with the reference's semantics (strict '>' scanning upwards: lowest index wins ties);
// DPP steps inside the wave, one LDS round across the waves
__device__ void block_argmax(Red& r, float v, int idx, float* out_v, int* out_i)
{
  wave_reduce_steps([&](auto ctrl, auto mask) {
    const float ov = __builtin_bit_cast(float, dpp_from<decltype(ctrl)::value, decltype(mask)::value>(__builtin_bit_cast(int, v)));
    const int oi = dpp_from<decltype(ctrl)::value, decltype(mask)::value>(idx);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  });
  if ((threadIdx.x & 63) == 63) { r.fv[threadIdx.x >> 6] = v; r.iv[threadIdx.x >> 6] = idx; }
  __syncthreads();
  float bv = r.fv[0];
  int bi = r.iv[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) {
    const float ov = r.fv[w];
    const int oi = r.iv[w];
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  *out_v = bv;
  *out_i = bi;
  __syncthreads();
}

__device__ int block_sum_int(Red& r, int v)
{
  wave_reduce_steps([&](auto ctrl, auto mask) {
    v += __builtin_amdgcn_update_dpp(0, v, decltype(ctrl)::value, decltype(mask)::value, 0xf, false);
  });
  if ((threadIdx.x & 63) == 63) r.iv[threadIdx.x >> 6] = v;
  __syncthreads();
  int out = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) out += r.iv[w];
  __syncthreads();
  return out;
}

// (the order of the additions inside a wave is a fixed tree, as it was with the shuffle steps: estimate only, sdr_sync.c:259-302)
__device__ double block_sum_double(Red& r, double v)
{
  wave_reduce_steps([&](auto ctrl, auto mask) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits), decltype(ctrl)::value, decltype(mask)::value, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits >> 32), decltype(ctrl)::value, decltype(mask)::value, 0xf, false);
    v += __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));   // 0.0 where the row mask excludes the lane
  });
  if ((threadIdx.x & 63) == 63) r.dv[threadIdx.x >> 6] = v;
  __syncthreads();
  double out = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) out += r.dv[w];
  __syncthreads();
  return out;
}

// multiply by conj(PRS value): quarter turns q = 0..3 <-> 1, j, -1, -j
__device__ __forceinline__ double2 mul_conj_prs(double2 x, int q)
{
  switch (q & 3) {
    case 0: return x;
    case 1: return make_double2(x.y, -x.x);
    case 2: return make_double2(-x.x, -x.y);
    default: return make_double2(-x.y, x.x);
  }
}

struct Shared {
  StreamState st;
  Red red;
  int status, do_sync, fifo_count, fresh;
  int coarse_fs;
  double fine_fs;
  int2 ahead;            // what the look-ahead pass's table holds for this call's read (sync_call), x < 0: nothing
};

// LDS: A (2048 points) and the batch buffer Bf lie back to back; the coarse frequency search correlates all 29 offsets in ONE
// batch of 29 x 128 = 3712 points laid over both (the 156 spectrum bins it needs are set aside first), the fine time search
// uses 3 x 512 points of Bf.  78 KB in all: two workgroups per CU.
constexpr int kBatchPoints = 29 * 128 - 2048;              // 1664 >= 3 x 512
constexpr int kSpecBins = 128 + 28;                        // spectrum bins touched by the 29 offsets

// ---- the estimators of sdr_sync.c, each run by the whole workgroup on LDS buffers ---------------------------------------
// A: 2048 points; Bf: kBatchPoints; tw: exp(2 pi i k / 2048), k < 1024.  All return the same value in every thread.

// dab_coarse_time_sync (sdr_sync.c:34-68) -> byte shift, 0 = the null symbol is where it should be.  In two parts, so that the caller can
// put the stream's tail bytes in place between them (the search reads the whole frame buffer, the test only its first 5320 bytes):
// the null-symbol energy test (sdr_sync.c:40-46) ...
// (this thread's share, to be summed over the workgroup: the caller issues other loads between the two)
__device__ __forceinline__ int null_symbol_energy_part(const uint8_t* stream, const FrameView& view)
{
  const int tid = threadIdx.x;
  int e = 0;
  if (const uint8_t* win = contiguous_window(stream, view, 0, 20 * 266)) {
    if (tid < 266) e = abs(rail(win[20 * tid]));
  } else {
    for (int n = tid; n < 266; n += kThreads) e += abs(rail(view_byte(stream, view, 20 * n)));
  }
  return e;
}
// ... and the search for the null symbol (sdr_sync.c:47-68)
__device__ int coarse_time_search(const uint8_t* stream, const FrameView& view, Red& red, uint8_t* env)
{
  const int tid = threadIdx.x;
  // envelope a[n] = |real[10 n]|, window sums of 266 taps, first minimum
  constexpr int kEnv = (kTfSamples - kNullSamples) / 10 + 266;   // 19661
  constexpr int kWin = (kTfSamples - kNullSamples) / 10;        // 19395 windows examined
  for (int n = tid; n < kEnv; n += kThreads) env[n] = static_cast<uint8_t>(abs(rail(view_byte(stream, view, 20 * n))));
  __syncthreads();
  const int per = (kWin + kThreads - 1) / kThreads;
  const int m0 = tid * per, m1 = min(m0 + per, kWin);
  float best = 9999999.0f;
  int bestm = 0x7fffffff;
  if (m0 < m1) {
    int s = 0;
    for (int q = 0; q < 266; ++q) s += env[m0 + q];
    for (int m = m0; m < m1; ++m) {
      if (static_cast<float>(s) < best) { best = static_cast<float>(s); bestm = m; }
      s += env[m + 266] - env[m];
    }
  }
  // arg-min, lowest index wins ties: arg-max of the negated value
  float bv;
  int bi;
  block_argmax(red, -best, bestm, &bv, &bi);
  const int coarse = (-bv < 9999999.0f) ? bi * 20 : 0;
  __syncthreads();                                         // env (aliases the batch buffer) is free again
  return coarse;
}

// What a thread of the fine time search reads from tables at places that depend on its index only: the PRS quarter turns of its three carriers and the
// radix-3 factors of its three correlation lags.  Fetched ONCE per kernel: inside the chain they were two rounds of global loads per call (0.9 of 8.6 us,
// tools/sync_times.py).
struct FineTimeTables {
  int q[3];
  double2 w1[3], w2[3];
};
__device__ __forceinline__ FineTimeTables fine_time_tables(const double2* __restrict__ tw1536, const uint8_t* __restrict__ prs_q)
{
  static_assert(kCarriers == 3 * kThreads, "three carriers per thread");
  FineTimeTables t;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = threadIdx.x + j * kThreads;
    t.q[j] = prs_q[i];
    t.w1[j] = tw1536[i];
    t.w2[j] = tw1536[(2 * i) % 1536];
  }
  return t;
}

// dab_fine_time_sync (sdr_sync.c:71-202) -> signed byte shift
__device__ int fine_time_sync(const uint8_t* stream, const FrameView& view, int nco, double2* A, double2* Bf, const double2* tw,
                              const FineTimeTables& tab, Red& red, const Prefetched<2048>& pf)
{
  const int tid = threadIdx.x;
  if (pf.ok) {
#pragma unroll
    for (int i = 0; i < 2048 / kThreads; ++i) A[lds_at(tid + i * kThreads)] = sample_of(pf.w[i]);
  } else {
    load_samples<2048>(stream, view, 2 * (kNullSamples + kCpSamples), nco, A);
  }
  __syncthreads();
  SYNC_STAMP(3);
  dft_dif<11, 3, 3, 3, 2>(A, 1, -1.0, tw);
  SYNC_STAMP(4);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = tid + j * kThreads;
    const int bin = i < 768 ? i + 1280 : i - 765;
    const double2 c = mul_conj_prs(A[lds_at(brev(bin, 11))], tab.q[j]);
    Bf[lds_at((i % 3) * 512 + i / 3)] = c;       // decimate by 3 for the 3 x 512 inverse DFT
  }
  __syncthreads();
  SYNC_STAMP(5);
  dft_dif<9, 3, 3, 3>(Bf, 3, +1.0, tw);
  SYNC_STAMP(6);
  float fv = -99999.0f;
  int fi = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int kk = tid + j * kThreads;
    const int r = brev(kk & 511, 9);
    const double2 f0 = Bf[lds_at(r)], f1 = Bf[lds_at(512 + r)], f2 = Bf[lds_at(1024 + r)];
    const double2 w1 = tab.w1[j], w2 = tab.w2[j];
    const double xr = f0.x + (f1.x * w1.x - f1.y * w1.y) + (f2.x * w2.x - f2.y * w2.y);
    const double xi = f0.y + (f1.x * w1.y + f1.y * w1.x) + (f2.x * w2.y + f2.y * w2.x);
    const float mag = static_cast<float>(sqrt(xr * xr + xi * xi));
    if (mag > fv) { fv = mag; fi = kk; }     // kk ascending per thread: first maximum kept
  }
  float gv;
  int gi;
  SYNC_STAMP(7);
  block_argmax(red, fv, fi, &gv, &gi);
  SYNC_STAMP(8);
  return gi < 768 ? gi * 2 + 16 : (gi - 1536) * 2;
}

// input_sdr.c:90-104 + dab_coarse_freq_sync_2 (sdr_sync.c:205-258) -> carrier offset -14 .. 14
__device__ int coarse_freq_sync(const uint8_t* stream, const FrameView& view, int nco, int fine, double2* A, double2* spec, const double2* tw,
                                const uint8_t* __restrict__ prs_q, Red& red)
{
  const int tid = threadIdx.x;
  load_samples<2048>(stream, view, 2 * (kNullSamples + kCpSamples + 1 + fine), nco, A);
  __syncthreads();
  dft_dif<11, 3, 3, 3, 2>(A, 1, -1.0, tw);
  // offset kk = -14 .. 14 correlates bins 14 + kk + 256 + s (s < 128) of the fftshifted spectrum: 256 .. 411
  for (int j = tid; j < kSpecBins; j += kThreads) spec[j] = A[lds_at(brev((256 + j + 1024) & 2047, 11))];
  __syncthreads();
  double2* W = A;                                          // 29 x 128 points over A and the batch buffer behind it
  for (int idx = tid; idx < 29 * 128; idx += kThreads) {
    const int o = idx / 128, s = idx % 128;                // o = kk + 14
    W[lds_at(idx)] = mul_conj_prs(spec[o + s], prs_q[14 + s]);
  }
  __syncthreads();
  dft_dif<7, 3, 2, 2>(W, 29, +1.0, tw);
  // per-offset maximum |.|, then first maximum over offsets
  float cv = -99999.0f;
  int ci = 0x7fffffff;
  for (int idx = tid; idx < 29 * 128; idx += kThreads) {
    const double2 x = W[lds_at(idx)];
    const float mag = static_cast<float>(sqrt(x.x * x.x + x.y * x.y));
    if (mag > cv) { cv = mag; ci = idx / 128; }            // idx ascending per thread
  }
  float hv;
  int hi;
  block_argmax(red, cv, ci, &hv, &hi);
  return hi - 14;
}

// ---- the same search in single precision: sync_verify_kernel's first pass ---------------------------------------------------
// The reference's result is the OFFSET (0 .. 28) that holds the largest of the 29 x 128 magnitudes (float compare of double values,
// first maximum wins).  In fp32 the same offset comes out whenever that maximum stands clear of every OTHER offset's maximum by far
// more than the rounding of the transforms (1e-6 of the values' scale): the pass below demands 1 % and otherwise leaves the call to
// the fp64 pass (a locked signal's true offset correlates ~100 x higher than the others; noise-only frames are not clear-cut and
// simply take the fp64 path).  Half the LDS bytes and twice the instruction rate of the fp64 transforms.
constexpr int kVerifyAgain = 0x7fff0000;                   // coarse_freq_shift marker: "decide this call in double"
__device__ __forceinline__ float2 mul_conj_prs32(float2 x, int q)
{
  switch (q & 3) {
    case 0: return x;
    case 1: return make_float2(x.y, -x.x);
    case 2: return make_float2(-x.x, -x.y);
    default: return make_float2(-x.y, x.x);
  }
}
__device__ int coarse_freq_sync32(const uint8_t* stream, const FrameView& view, int fine, float2* A, float2* spec, const float2* tw,
                                  const uint8_t* __restrict__ prs_q, Red& red, bool* clear)
{
  const int tid = threadIdx.x;
  {
    const int p0 = 2 * (kNullSamples + kCpSamples + 1 + fine);
    if (const uint8_t* win = contiguous_window(stream, view, p0, p0 + 2 * 2048)) {
      const uint16_t* src = reinterpret_cast<const uint16_t*>(win);
      unsigned w[2048 / kThreads];
#pragma unroll
      for (int i = 0; i < 2048 / kThreads; ++i) w[i] = src[tid + i * kThreads];
#pragma unroll
      for (int i = 0; i < 2048 / kThreads; ++i) {
        const double2 x = sample_of(w[i]);
        A[lds_at(tid + i * kThreads)] = make_float2(static_cast<float>(x.x), static_cast<float>(x.y));
      }
    } else {
      for (int n = tid; n < 2048; n += kThreads) {
        const double2 x = view_sample(stream, view, p0 + 2 * n, 0);
        A[lds_at(n)] = make_float2(static_cast<float>(x.x), static_cast<float>(x.y));
      }
    }
  }
  __syncthreads();
  dft_dif<11, 3, 3, 3, 2>(A, 1, -1.0f, tw);
  for (int j = tid; j < kSpecBins; j += kThreads) spec[j] = A[lds_at(brev((256 + j + 1024) & 2047, 11))];
  __syncthreads();
  float2* W = A;                                           // 29 x 128 points
  for (int idx = tid; idx < 29 * 128; idx += kThreads) {
    const int o = idx / 128, s = idx % 128;
    W[lds_at(idx)] = mul_conj_prs32(spec[o + s], prs_q[14 + s]);
  }
  __syncthreads();
  dft_dif<7, 3, 2, 2>(W, 29, +1.0f, tw);
  // a thread's points idx = tid + 512 k lie in offsets tid / 128 + 4 k: eight different ones, their magnitudes kept for the second question
  constexpr int kPer = (29 * 128 + kThreads - 1) / kThreads;
  static_assert(kThreads % 128 == 0, "one offset per point of a thread");
  float mags[kPer];
  float cv = -99999.0f;
  int ci = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int idx = tid + k * kThreads;
    mags[k] = -1.0f;
    if (idx < 29 * 128) {
      const float2 x = W[lds_at(idx)];
      mags[k] = sqrtf(x.x * x.x + x.y * x.y);
      if (mags[k] > cv) { cv = mags[k]; ci = idx / 128; }  // idx ascending per thread
    }
  }
  float hv;
  int hi;
  block_argmax(red, cv, ci, &hv, &hi);
  float ov = 0.0f;                                         // the largest magnitude of any OTHER offset
#pragma unroll
  for (int k = 0; k < kPer; ++k)
    if ((tid + k * kThreads) / 128 != hi) ov = fmaxf(ov, mags[k]);
  float rv;
  int ri;
  block_argmax(red, ov, tid, &rv, &ri);
  *clear = hv > 0.0f && (hv - rv) > 0.01f * hv;
  return hi - 14;
}

// dab_fine_freq_corr (sdr_sync.c:259-302): estimate only, in Hz
__device__ double fine_freq_corr(const uint8_t* stream, const FrameView& view, int nco, Red& red)
{
  double acc = 0;
  const uint8_t* win = nco == 0 ? contiguous_window(stream, view, 2 * kNullSamples, 2 * (kNullSamples + 2048 + kCpSamples)) : nullptr;
  for (int n = threadIdx.x; n < kCpSamples; n += kThreads) {
    const int pl = 2 * (kNullSamples + 2048 + n), pr = 2 * (kNullSamples + n);
    double2 l, r;
    if (win) {
      const uint16_t* src = reinterpret_cast<const uint16_t*>(win);
      l = sample_of(src[2048 + n]);
      r = sample_of(src[n]);
    } else {
      l = view_sample(stream, view, pl, nco);
      r = view_sample(stream, view, pr, nco);
    }
    const double lr = l.x, li = l.y, rr = r.x, ri = r.y;
    acc += atan2(-lr * ri + li * rr, lr * rr + li * ri);
  }
  acc = block_sum_double(red, acc);
  return acc / 504 / (2 * M_PI) * 1000;
}

struct SyncLds {
  double2* A;
  double2* Bf;
  double2* tw;
  double2* spec;
  Shared* sh;
};
__device__ __forceinline__ SyncLds sync_lds(unsigned char* smem, const double2* __restrict__ tw2048, bool fill = true)
{
  SyncLds l;
  l.A = reinterpret_cast<double2*>(smem);                  // 2048: main DFT buffer
  l.Bf = l.A + 2048;                                        // batch buffer (also the envelope bytes of the coarse time search)
  l.tw = l.Bf + kBatchPoints;                               // 1024: LDS copy of the twiddle table
  l.spec = l.tw + 1024;                                     // kSpecBins
  l.sh = reinterpret_cast<Shared*>(l.spec + kSpecBins);
  if (fill)
    for (int i = threadIdx.x; i < 1024; i += kThreads) l.tw[lds_at(i)] = tw2048[i];
  return l;
}

// fifo_call (fifo_view.hpp) on the workgroup's shared state, by the first wave instead of one thread: lane i looks at segment i of the old view (12 at most), a
// ballot says which survive the read and a prefix count where each goes; the scalar part is computed by every lane alike.  One thread walking the LDS-resident
// state, building the new view through a dynamically indexed local copy, took well over a microsecond of every call of the chain (tools/sync_times.py).
// Same results as fifo_call: the host replay of that function is what the CPU suite holds against the reference's sdr_fifo.c, and every GPU trace test
// compares fifo_count and the views' effects call by call.
// chunk: bytes this call appended; tail_slot: where this call's copy of the stream's tail bytes goes (device_types.hpp: kTailBytes), or null
__device__ __forceinline__ void fifo_call_wave(Shared& sh, int chunk, const uint8_t* tail_slot)
{
  const int lane = threadIdx.x;                          // caller: threadIdx.x < 64
  StreamState& st = sh.st;
  const int64_t fed = st.fed + chunk;
  int64_t consumed = st.consumed, count = fed - consumed;
  int status = 0, do_sync = 0, fresh = 0;
  int startup = st.startup_delay;
  if (count >= 3 * kTfSamples) {
    const int shift = st.coarse_timeshift + st.fine_timeshift;
    const int64_t consumed0 = consumed;
    int len, skipped = 0;
    if (shift > 0) {
      consumed += shift;
      count -= shift;
      len = count < kTfBytes ? static_cast<int>(count) : kTfBytes;
      skipped = shift;
    } else {
      len = kTfBytes + shift;
    }
    const bool extra = skipped > len;                    // the skipped bytes beyond a short frame stay visible
    const int n0 = extra ? 2 : 1, covered = extra ? skipped : len;
    // old entries, one per lane (read before anything is written: one instruction stream)
    const int old_n = st.view.nseg;
    const int old_end = lane < kMaxSeg ? st.view.seg_end[lane] : 0;
    const int prev_end = lane > 0 && lane < kMaxSeg ? st.view.seg_end[lane - 1] : 0;
    const int64_t old_src = lane < kMaxSeg ? st.view.seg_src[lane] : -1;
    // (only what still shows BELOW the tail bytes: those travel as bytes, fifo_view.hpp)
    const bool keep = lane < old_n && lane < kMaxSeg && old_end > covered && covered < kTailStart && prev_end < kTailStart;
    fresh = n0;
    const unsigned long long mask = __ballot(keep);
    const int pos = n0 + __popcll(mask & ((1ull << lane) - 1ull));
    const int total = n0 + __popcll(mask);
    const int n = total < kMaxSeg ? total : kMaxSeg;
    if (keep && pos < kMaxSeg) { st.view.seg_end[pos] = old_end; st.view.seg_src[pos] = old_src; }
    if (lane >= n && lane < kMaxSeg) { st.view.seg_end[lane] = kTfBytes; st.view.seg_src[lane] = -1; }
    if (lane == 0) {
      st.view.seg_end[0] = len;
      st.view.seg_src[0] = consumed;
      if (extra) { st.view.seg_end[1] = skipped; st.view.seg_src[1] = consumed0; }   // buffer[p] = stream[consumed0 + p] for p < shift
      st.view.nseg = n;
      st.view.tail = tail_slot;
      if (total > kMaxSeg) st.overflow = 1;
    }
    consumed += len;
    count -= len;
    status = 1;
    if (startup <= 0) ++startup;
    else do_sync = 1;
  }
  if (lane == 0) {
    st.fed = fed;
    st.consumed = consumed;
    st.startup_delay = startup;
    sh.status = status;
    sh.do_sync = do_sync;
    sh.fifo_count = static_cast<int>(count);
    sh.fresh = fresh;
    sh.coarse_fs = 0;
  }
}

// What one call of sdr_demod needs besides the stream's state: fixed for a workgroup's life
struct CallEnv {
  const uint8_t* stream;
  int b, max_calls, kdesc0, afc;
  CallDesc* descs;
  int2* info;
  SyncTails tails;
  SyncLds lds;
  const uint8_t* prs_q;
  // the look-ahead pass's table (sync_ahead_kernel) -- by value: a pointer to the kernel's argument struct sends every use through scratch memory
  bool ahead;                // there is one
  const int2* ahead_table;
  const int64_t* ahead_src0;
  int* ahead_hits;
  int ahead_nspec, ahead_nhyp;
  int kahead0;               // the call its entry 0 stands for
};
// what a call's read leaves in the stream's tail bytes (thread t < 384 holds bytes 4 t .. 4 t + 3 in tail_word): the two halves of the update, so that the
// caller can issue the loads behind those it waits for first and take them up behind that wait
struct TailUpdate {
  unsigned lo = 0, hi = 0;
  bool take_lo = false, take_hi = false;
};
constexpr int kTailWords = kTailBytes / 4;
__device__ __forceinline__ TailUpdate tail_load(const uint8_t* stream, const Shared& sh, bool frame_read)
{
  TailUpdate u;
  const int tid = threadIdx.x;
  if (frame_read && tid < kTailWords) {
    const int64_t s0 = read_source(sh.st.view, sh.fresh, kTailStart + 4 * tid), s1 = read_source(sh.st.view, sh.fresh, kTailStart + 4 * tid + 2);
    u.take_lo = s0 >= 0;
    u.take_hi = s1 >= 0;
    if (u.take_lo) u.lo = *reinterpret_cast<const uint16_t*>(stream + s0);
    if (u.take_hi) u.hi = *reinterpret_cast<const uint16_t*>(stream + s1);
  }
  return u;
}
__device__ __forceinline__ void tail_commit(const TailUpdate& u, bool frame_read, uint32_t& tail_word, uint8_t* tail_slot)
{
  const int tid = threadIdx.x;
  if (frame_read && tid < kTailWords) {
    if (u.take_lo) tail_word = (tail_word & 0xffff0000u) | u.lo;
    if (u.take_hi) tail_word = (tail_word & 0x0000ffffu) | (u.hi << 16);
    if (tail_slot) reinterpret_cast<uint32_t*>(tail_slot)[tid] = tail_word;
  }
}

// (the look-ahead pass: sync_ahead_kernel below)
constexpr int kAheadWindowEnd = 2 * (kNullSamples + kCpSamples) + 2 * 2048;   // the estimators read buffer bytes [0, 5320) and [6320, 10416)
// A call's row of the table and its predicted start position travel one call ahead of their use, in registers of the stream's first wave (lane h holds
// the entry of start position h): the chain's dependent path then has no global load in it (two in a row cost every call 1.5 us).
struct AheadRow {
  int64_t src0 = -1;
  int2 entry = make_int2(-1, 0);
};
__device__ __forceinline__ AheadRow ahead_fetch(const int2* __restrict__ table, const int64_t* __restrict__ src0, int nspec, int nhyp, int b, int j)   // by the first wave
{
  AheadRow r;
  if (j >= 0 && j < nspec) {
    const size_t slot = static_cast<size_t>(b) * nspec + j;
    r.src0 = src0[slot];
    if (static_cast<int>(threadIdx.x) < nhyp) r.entry = table[slot * nhyp + threadIdx.x];
  }
  return r;
}
// the entry for a read that began where this call's did (the view fifo_call_wave just left), by the first wave; the same value in all its lanes
__device__ __forceinline__ int2 ahead_lookup(int nhyp, const AheadRow& row, const FrameView& view)
{
  const int2 miss = make_int2(-1, 0);
  const int64_t src = view.seg_src[0];
  if (row.src0 < 0 || src < 0 || view.seg_end[0] < kAheadWindowEnd) return miss;
  const int64_t d = src - row.src0;
  const int half = nhyp / 2;
  if ((d & 1) != 0 || d < -2 * half || d > 2 * half) return miss;
  const int h = static_cast<int>(d / 2) + half;
  return make_int2(__shfl(row.entry.x, h), __shfl(row.entry.y, h));
}
// kChainOnly = false: sdr_demod's synchronisation as the reference runs it, call after call.
// kChainOnly = true : only what the NEXT call depends on -- FIFO bookkeeping, coarse time, fine time.  The coarse frequency
//   offset is assumed to come out within +-1 carrier (so the frame is demodulated and no resync is forced) and left, with the
//   fine frequency estimate, to sync_verify_kernel, which runs over all transmission frames in parallel; a stream whose
//   assumption fails is scanned again in full (Engine::scan_streams).  Not with the software AFC, where the NCO of the next
//   frame depends on both estimates.
// ONE call of sdr_demod (input_sdr.c:27-112) on the workgroup's shared state: call k of the stream, descriptor k - kdesc0.  The body of the chain's loop
// (sync_scan_kernel).  Ends with a barrier.
// a call's update of the tail bytes whose loads are still in flight: taken up at the start of the next call (TailPending below)
struct TailPending {
  TailUpdate upd;
  uint8_t* slot = nullptr;
  bool valid = false;
};
template <bool kChainOnly>
__device__ __forceinline__ void sync_call(const CallEnv& env, Shared& sh, int k, const FineTimeTables& fine_tab, uint32_t& tail_word, AheadRow& row, TailPending& pend)
{
  const int tid = threadIdx.x;
  if (pend.valid) {                                       // (the same in every thread) the previous call's tail bytes: their loads have had a whole call's time
    tail_commit(pend.upd, true, tail_word, pend.slot);
    pend.valid = false;
  }
  const uint8_t* const stream = env.stream;
  double2* const A = env.lds.A;
  double2* const Bf = env.lds.Bf;
  double2* const tw = env.lds.tw;
  uint8_t* const env_bytes = reinterpret_cast<uint8_t*>(Bf);     // 19661 bytes, coarse search only
  SYNC_STAMP(0);
  // ---- FIFO bookkeeping: input_sdr.c:36-55 over sdr_fifo.c:43-61 (fifo_view.hpp) -------
  uint8_t* const tail_slot = env.tails.images ? env.tails.images + (static_cast<size_t>(env.b) * env.max_calls + (k - env.kdesc0)) * kTailBytes : nullptr;
  if (tid < 64) {
    fifo_call_wave(sh, env.tails.chunk, tail_slot);
    if (env.ahead) {
      // what the look-ahead pass (sync_ahead_kernel) holds for a read that began where this one did; then the next call's row on its way
      const int2 found = sh.do_sync ? ahead_lookup(env.ahead_nhyp, row, sh.st.view) : make_int2(-1, 0);
      if (tid == 0) sh.ahead = found;
      row = ahead_fetch(env.ahead_table, env.ahead_src0, env.ahead_nspec, env.ahead_nhyp, env.b, k + 1 - env.kahead0);
    }
  }
  __syncthreads();

  // what this call's read leaves in the tail bytes: the loads are issued behind those the chain waits for first (below), taken up -- and the frame's
  // copy written -- behind that wait
  const bool frame_read = DABHIP_K1_TAIL && sh.status != 0;
  TailUpdate tail_upd;

  const int nco = env.afc ? sh.st.tuner_hz : 0;
  if (sh.do_sync) {
    const FrameView& view = sh.st.view;
    // {null-symbol energy, fine time shift} out of the look-ahead pass's table, x < 0: nothing there (or no pass: never with the software AFC)
    const int2 ahead = env.ahead ? sh.ahead : make_int2(-1, 0);
    const bool hit = ahead.x >= 0;                        // (the same in every thread)
    if (hit && tid == 0) atomicAdd(env.ahead_hits, 1);   // statistics ("sync_spec_calls")
    Prefetched<2048> pf;
    pf.ok = false;
    if (!hit) pf = prefetch_samples<2048>(stream, view, 2 * (kNullSamples + kCpSamples), nco);   // for the fine time search
    SYNC_STAMP(1);
    const int force = sh.st.force_timesync;
    const int energy_part = hit ? 0 : null_symbol_energy_part(stream, view);                   // input_sdr.c:64-74
    tail_upd = tail_load(stream, sh, frame_read);
    int energy = ahead.x;
    if (hit) __syncthreads();                             // (force has been read by everyone before it is cleared below)
    else energy = block_sum_int(sh.red, energy_part);
    // With the estimators out of the table nothing in this call waits for memory but the tail bytes' loads, and nothing in it reads the tail bytes unless the
    // coarse search runs: the update is then taken up at the start of the next call (or behind the loop) instead of stalling this one for a memory latency.
    if (hit && frame_read && energy < 5000 && force == 0) {
      pend.upd = tail_upd;
      pend.slot = tail_slot;
      pend.valid = true;
    } else {
      tail_commit(tail_upd, frame_read, tail_word, tail_slot);
    }
    int coarse = 0;
    if (energy >= 5000 || force != 0) {
      __syncthreads();                                  // the search reads the whole frame buffer, tail bytes included
      coarse = coarse_time_search(stream, view, sh.red, env_bytes);
    }
    if (tid == 0) { sh.st.coarse_timeshift = coarse; sh.st.force_timesync = 0; }
    __syncthreads();
    SYNC_STAMP(2);
    if (coarse == 0) {
      const int fine = hit ? ahead.y : fine_time_sync(stream, view, nco, A, Bf, tw, fine_tab, sh.red, pf);   // input_sdr.c:84
      if (tid == 0) sh.st.fine_timeshift = fine;
      if (kChainOnly) {
        if (tid == 0) sh.status = 2;                    // assumed; sync_verify_kernel checks it
      } else {
        const int cfs = coarse_freq_sync(stream, view, nco, fine, A, env.lds.spec, tw, env.prs_q, sh.red);   // input_sdr.c:90-104
        if (tid == 0) sh.coarse_fs = cfs;
        if (abs(cfs) > 1) {
          if (tid == 0) sh.st.force_timesync = 1;       // input_sdr.c:105-109
        } else {
          const double ffs = fine_freq_corr(stream, view, nco, sh.red);                        // input_sdr.c:112
          if (tid == 0) {
            sh.fine_fs = ffs;
            sh.status = 2;
          }
        }
      }
    }
  } else {
    tail_upd = tail_load(stream, sh, frame_read);
    tail_commit(tail_upd, frame_read, tail_word, tail_slot);   // a frame that is read and dropped (input_sdr.c:51-55) still overwrites the buffer
  }
  __syncthreads();
  if (tid >= 64 && tid < 64 + static_cast<int>(sizeof(FrameView) / 4)) {     // the view, word by word, by the second wave (one thread copying 152 bytes out of LDS was a microsecond)
    CallDesc& d = env.descs[static_cast<size_t>(env.b) * env.max_calls + (k - env.kdesc0)];
    reinterpret_cast<uint32_t*>(&d.view)[tid - 64] = reinterpret_cast<const uint32_t*>(&sh.st.view)[tid - 64];
  }
  if (tid == 0) {
    CallDesc& d = env.descs[static_cast<size_t>(env.b) * env.max_calls + (k - env.kdesc0)];
    d.status = sh.status;
    d.ordinal = sh.status == 2 ? sh.st.next_ordinal++ : -1;
    if (env.info) env.info[static_cast<size_t>(env.b) * env.max_calls + (k - env.kdesc0)] = make_int2(d.status, d.ordinal);   // what the host lays the frames out with
    d.coarse_timeshift = sh.st.coarse_timeshift;
    d.fine_timeshift = sh.st.fine_timeshift;
    d.coarse_freq_shift = sh.coarse_fs;
    d.fifo_count = sh.fifo_count;
    d.fine_freq_shift = sh.fine_fs;
    d.nco_hz = nco;
    if (env.afc) {
      // the tuner feedback of demod_thread_fn (dab2eti.c:76-103), applied to the NCO instead of the tuner; it runs
      // after EVERY call, also those that produced no frame (coarse 0, fine estimate stale), exactly as there
      StreamState& st = sh.st;
      const int c = sh.coarse_fs;
      if (abs(c) > 1) st.tuner_hz += c < 0 ? -1000 : 1000;
      if (abs(c) == 1) {
        st.rng = st.rng * 1103515245u + 12345u;
        const int step = static_cast<int>((st.rng >> 16) % 1000u);
        st.tuner_hz += c < 0 ? -step : step;
      }
      // dab2eti.c:97-98 in its integer semantics: abs() is the int one (the double is truncated first: |ffs| >= 51), and
      // "frequency = frequency + ffs/3" stores a double into the unsigned tuner frequency, i.e. floors the sum
      if (c == 0 && abs(static_cast<int>(sh.fine_fs)) > 50) st.tuner_hz += static_cast<int>(floor(sh.fine_fs / 3));
    }
  }
  __syncthreads();
  SYNC_STAMP(9);
}

// spec (the look-ahead schedule, Engine::scan_streams): call_limit >= 0: at most that many calls in this launch; ctl != nullptr: the descriptor numbering
// of a scan made of several launches -- the launch with record_base set writes its first call there, the others read it; lookup: the look-ahead pass's table
// stands for the calls from where the stream stands at this launch
template <bool kChainOnly>
__global__ __launch_bounds__(kThreads) void sync_scan_kernel(const uint8_t* const* __restrict__ iq,
                                                             const int64_t* __restrict__ nbytes,
                                                             const StreamState* __restrict__ states_in, StreamState* __restrict__ states,
                                                             const int* __restrict__ stream_list,
                                                             CallDesc* __restrict__ descs, int2* __restrict__ info, int max_calls, int call_begin,
                                                             int call_end, const double2* __restrict__ tw2048,
                                                             const double2* __restrict__ tw1536,
                                                             const uint8_t* __restrict__ prs_q, int afc, SyncTails tails, SpecArgs spec)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = stream_list ? stream_list[blockIdx.x] : blockIdx.x, tid = threadIdx.x;
  const int64_t total_calls = nbytes[b] / kChunkBytes;
  int kend = call_end < 0 ? static_cast<int>(total_calls) : min(call_end, static_cast<int>(total_calls));
  // call_begin < 0: continue where the stream's state stands (calls already fed), descriptors numbered from there
  const int kfirst = call_begin >= 0 ? call_begin : static_cast<int>(states_in[b].fed / kChunkBytes);
  int kdesc0 = call_begin >= 0 ? 0 : kfirst;
  if (spec.ctl) {                                          // a scan made of several launches: the first one records the numbering, the others read it
    if (spec.record_base) { if (tid == 0) { spec.ctl[b] = kdesc0; if (blockIdx.x == 0) spec.ctl[gridDim.x] = 0; } }   // (and the count of table hits cleared)
    else kdesc0 = spec.ctl[b];
  }
  if (spec.call_limit >= 0) kend = min(kend, kfirst + spec.call_limit);
  if (spec.call_limit >= 0 && kfirst >= kend) return;     // a short chain with nothing to do: states and tails stay as they are
  const SyncLds lds = sync_lds(smem, tw2048);
  Shared& sh = *lds.sh;
  const CallEnv env{iq[b], b, max_calls, kdesc0, afc, descs, info, tails, lds, prs_q, spec.lookup != 0, spec.table, spec.src0, spec.ctl + spec.nstreams, spec.nspec, spec.nhyp, kfirst};
  if (tid == 0) { sh.st = states_in[b]; sh.fine_fs = sh.st.fine_freq_shift; }
  const FineTimeTables fine_tab = fine_time_tables(tw1536, prs_q);
  // The stream's tail bytes -- the last kTailBytes of the reference's frame buffer, which a short read leaves as they were (sdr_fifo.c:56-59) --
  // travel in registers: thread t < 384 holds bytes 4 t .. 4 t + 3.  (tails.state_in == nullptr: a caller without tail state, stage tests)
  uint32_t tail_word = 0;
  if (tails.state_in && tid < kTailWords) tail_word = reinterpret_cast<const uint32_t*>(tails.state_in + static_cast<size_t>(b) * kTailBytes)[tid];
  __syncthreads();

  AheadRow row;
  TailPending pend;
  if (env.ahead && tid < 64) row = ahead_fetch(env.ahead_table, env.ahead_src0, env.ahead_nspec, env.ahead_nhyp, b, 0);
  for (int k = kfirst; k < kend; ++k) {
#if DABHIP_SYNC_TIMES
    if (blockIdx.x == 0 && tid == 0) g_sync_call = k - kfirst;
#endif
    sync_call<kChainOnly>(env, sh, k, fine_tab, tail_word, row, pend);
  }
  if (pend.valid) tail_commit(pend.upd, true, tail_word, pend.slot);
  if (tid == 0) { sh.st.fine_freq_shift = sh.fine_fs; states[b] = sh.st; }
  if (tails.state_out && tid < kTailWords) reinterpret_cast<uint32_t*>(tails.state_out + static_cast<size_t>(b) * kTailBytes)[tid] = tail_word;
}

// ---- the look-ahead pass -------------------------------------------------------------------------------------------------------------------
// The chain is sequential because a call's time shifts position the NEXT read -- but what a call computes from its frame (the null-symbol energy and the
// fine time search: all that the chain-only scan needs) is a function of the samples at the START of the frame buffer, i.e. of the stream position the
// read began at, and a locked receiver's reads begin within a few samples of where a receiver that never corrected anything would begin them (the
// reference's fine time search settles into a limit cycle of +20, -8, -8, -2 bytes on an ideal channel; it never returns "0, 0, 0").  So this pass runs,
// for every call the stream has left and every start position within +-(nhyp / 2) samples of the predicted one, the two estimators at once -- grid (nhyp,
// nspec, nstreams), all on the device together -- and leaves them in a table keyed by (call, start position); the chain launch behind it (sync_call) looks
// its call's start position up and skips the estimators on a hit.  The table holds what the chain would have computed, bit for bit (same code, same
// samples, same order of operations), so the results are the chain's in every case; a call whose read begins outside the window (a coarse correction,
// a drifting sample clock), that reads short of the estimators' samples, or that the prediction did not expect to read at all, is computed by the chain
// as before.  Worth it where the chain leaves most of the device idle (Engine::scan_streams: small batches): one ensemble of 64 TF: sync stage 0.58 -> 0.29 ms (profiles/r05_batch_curve*.json).
// The predicted start positions: the stream's calls replayed from where it stands with every time shift 0 (the pending one applied first).  Getting to
// "the state before call j" is bookkeeping only: calls up to the first predicted-zero shift go through the chain's own FIFO code (fifo_call_wave: a
// pending shift, a first frame still to be dropped), from there on a read is a full frame at the read pointer (fifo_view.hpp with shift 0), the counters
// repeat every three calls (3 x 262144 = 2 x 393216) and whole periods are skipped in one step.
__global__ __launch_bounds__(kThreads) void sync_ahead_kernel(const uint8_t* const* __restrict__ iq, const int64_t* __restrict__ nbytes,
                                                              const StreamState* __restrict__ states, const double2* __restrict__ tw2048,
                                                              const double2* __restrict__ tw1536, const uint8_t* __restrict__ prs_q, int chunk, SpecArgs spec)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.z, j = blockIdx.y, h = blockIdx.x, tid = threadIdx.x;
  const int total_calls = static_cast<int>(nbytes[b] / kChunkBytes);
  const int kcur = static_cast<int>(states[b].fed / kChunkBytes);
  if (j >= min(spec.nspec, total_calls - kcur)) return;
  const SyncLds lds = sync_lds(smem, tw2048, false);       // (the twiddle table only once the call turns out to have work: a third of the calls read no frame)
  Shared& sh = *lds.sh;
  static_assert(sizeof(StreamState) % 4 == 0 && sizeof(StreamState) / 4 <= kThreads, "copied word by word");
  if (tid < static_cast<int>(sizeof(StreamState) / 4)) reinterpret_cast<uint32_t*>(&sh.st)[tid] = reinterpret_cast<const uint32_t*>(states + b)[tid];
  __syncthreads();

  // ---- the j calls before this one, as predicted: every frame demodulated, no time shift ----
  int i = 0;
  while (i < j) {
    const bool lean = sh.st.coarse_timeshift + sh.st.fine_timeshift == 0 && sh.st.startup_delay > 0 && chunk == kChunkBytes;
    __syncthreads();                                       // every thread has read the state before it changes
    if (!lean) {
      // the chain's own bookkeeping for this call (a pending shift: short, skipping or dry read; the first frame, which is dropped), then the predicted outcome
      if (tid < 64) fifo_call_wave(sh, chunk, nullptr);
      __syncthreads();
      if (tid == 0 && sh.do_sync) {
        sh.st.coarse_timeshift = sh.st.fine_timeshift = sh.st.force_timesync = 0;
        ++sh.st.next_ordinal;
      }
      __syncthreads();
      ++i;
      continue;
    }
    // from here to call j: fifo_call with shift 0 -- a full frame from the read pointer once 1.5 frames are queued (fifo_view.hpp)
    int64_t fed = sh.st.fed, consumed = sh.st.consumed;
    fifo_skip_unshifted(fed, consumed, j - i);
    i = j;
    if (tid == 0) { sh.st.fed = fed; sh.st.consumed = consumed; }   // (the view a full read leaves shows through nothing: this call's read rewrites the buffer)
    __syncthreads();
  }

  // ---- this call's read, and the estimators at start position (predicted + 2 (h - nhyp / 2)) ----
  // (one start position per workgroup: as a loop over several of them the kernel takes 200 registers instead of 105 and loses its second workgroup per CU)
  if (tid < 64) fifo_call_wave(sh, chunk, nullptr);
  __syncthreads();
  const bool expected = sh.do_sync && sh.st.view.seg_src[0] >= 0 && sh.st.view.seg_end[0] >= kAheadWindowEnd;
  const int64_t src0 = expected ? sh.st.view.seg_src[0] : -1;
  const size_t slot = static_cast<size_t>(b) * spec.nspec + j;
  if (h == 0 && tid == 0) spec.src0[slot] = src0;
  if (!expected) return;
  const int64_t src = src0 + 2 * (h - spec.nhyp / 2);
  int2 out = make_int2(-1, 0);
  if (src >= 0 && src + kAheadWindowEnd <= nbytes[b]) {   // (workgroup-uniform)
    const FineTimeTables fine_tab = fine_time_tables(tw1536, prs_q);
    for (int n = tid; n < 1024; n += kThreads) lds.tw[lds_at(n)] = tw2048[n];
    __syncthreads();                                       // the view has been read by everyone
    if (tid == 0) {
      sh.st.view.nseg = 1;
      sh.st.view.seg_end[0] = kTfBytes;
      sh.st.view.seg_src[0] = src;
      sh.st.view.tail = nullptr;
    }
    __syncthreads();
    const FrameView& view = sh.st.view;
    const Prefetched<2048> pf = prefetch_samples<2048>(iq[b], view, 2 * (kNullSamples + kCpSamples), 0);
    const int energy = block_sum_int(sh.red, null_symbol_energy_part(iq[b], view));                 // input_sdr.c:64-74
    const int fine = fine_time_sync(iq[b], view, 0, lds.A, lds.Bf, lds.tw, fine_tab, sh.red, pf);   // input_sdr.c:84
    out = make_int2(energy, fine);
  }
  if (tid == 0) spec.table[slot * spec.nhyp + h] = out;
}

// The estimators the chain-only scan left out, for every call it assumed demodulated: grid (max_calls, nstreams).
// Fills coarse_freq_shift and fine_freq_shift of the call's descriptor; an offset beyond +-1 carrier breaks the assumption:
// the first such call of a stream is recorded in violation[stream].
// only_marked: the second pass behind sync_verify32_kernel -- only the calls that one left undecided (kVerifyAgain).
__global__ __launch_bounds__(kThreads) void sync_verify_kernel(const uint8_t* const* __restrict__ iq, CallDesc* __restrict__ descs, int max_calls,
                                                               int nstreams, const double2* __restrict__ tw2048, const uint8_t* __restrict__ prs_q,
                                                               int* __restrict__ violation, int only_marked)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int total = max_calls * nstreams;
  if (only_marked && violation[nstreams] == 0) return;     // the fp32 pass left nothing (its count of marked calls sits behind the per-stream entries)
  if (static_cast<int>(gridDim.x) == total && (descs[blockIdx.x].status != 2 || (only_marked && descs[blockIdx.x].coarse_freq_shift != kVerifyAgain)))
    return;                                                // nothing to do: skip the table fill as well
  const SyncLds lds = sync_lds(smem, tw2048);
  Shared& sh = *lds.sh;
  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int b = w / max_calls;
    CallDesc& d = descs[w];
    if (d.status != 2 || (only_marked && d.coarse_freq_shift != kVerifyAgain)) continue;   // the same for every thread of the workgroup
    __syncthreads();                                       // the previous call's readers of sh.st.view are done
    {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&d.view);
      uint32_t* dst = reinterpret_cast<uint32_t*>(&sh.st.view);
      if (threadIdx.x < sizeof(FrameView) / 4) dst[threadIdx.x] = src[threadIdx.x];
    }
    const int fine = d.fine_timeshift;
    __syncthreads();
    const uint8_t* stream = iq[b];
    const int cfs = coarse_freq_sync(stream, sh.st.view, 0, fine, lds.A, lds.spec, lds.tw, prs_q, sh.red);
    if (abs(cfs) > 1) {
      if (threadIdx.x == 0) { d.coarse_freq_shift = cfs; atomicMin(violation + b, w % max_calls); }
      continue;
    }
    const double ffs = fine_freq_corr(stream, sh.st.view, 0, sh.red);
    if (threadIdx.x == 0) { d.coarse_freq_shift = cfs; d.fine_freq_shift = ffs; }
  }
}

// First pass of the verification in single precision (see coarse_freq_sync32): one workgroup per call.  LDS: 3712 + 1024 + 156 float2
// and the small shared block = 39.5 KB, three workgroups per CU.  Calls whose arg-max is not clear-cut get the marker and are left to
// sync_verify_kernel(only_marked); the fine frequency estimate is the fp64 routine as before.
constexpr int kVerify32Points = 29 * 128;
__global__ __launch_bounds__(kThreads, 6) void sync_verify32_kernel(const uint8_t* const* __restrict__ iq, CallDesc* __restrict__ descs, int max_calls,
                                                                    int nstreams, const double2* __restrict__ tw2048, const uint8_t* __restrict__ prs_q,
                                                                    int* __restrict__ violation, int distrust)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int w = blockIdx.x;
  CallDesc& d = descs[w];
  if (d.status != 2) return;
  float2* A = reinterpret_cast<float2*>(smem);
  float2* tw = A + kVerify32Points;
  float2* spec = tw + 1024;
  Shared& sh = *reinterpret_cast<Shared*>(spec + kSpecBins + (kSpecBins & 1));
  for (int i = threadIdx.x; i < 1024; i += kThreads) {
    const double2 t = tw2048[i];
    tw[lds_at(i)] = make_float2(static_cast<float>(t.x), static_cast<float>(t.y));
  }
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&d.view);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&sh.st.view);
    if (threadIdx.x < sizeof(FrameView) / 4) dst[threadIdx.x] = src[threadIdx.x];
  }
  const int b = w / max_calls, fine = d.fine_timeshift;
  __syncthreads();
  const uint8_t* stream = iq[b];
  bool clear;
  const int cfs = coarse_freq_sync32(stream, sh.st.view, fine, A, spec, tw, prs_q, sh.red, &clear);
  if (!clear || distrust) {                                // distrust: test mode, every call goes on to the fp64 pass
    if (threadIdx.x == 0) { d.coarse_freq_shift = kVerifyAgain; atomicAdd(violation + nstreams, 1); }   // the counter behind the per-stream entries
    return;
  }
  if (abs(cfs) > 1) {
    if (threadIdx.x == 0) { d.coarse_freq_shift = cfs; atomicMin(violation + b, w % max_calls); }
    return;
  }
  const double ffs = fine_freq_corr(stream, sh.st.view, 0, sh.red);
  if (threadIdx.x == 0) { d.coarse_freq_shift = cfs; d.fine_freq_shift = ffs; }
}

// fine_freq_shift is only recomputed by calls that demodulate (input_sdr.c:112); every other call still shows the last value
// (sdr->fine_freq_shift persists).  One WAVE per stream carries it through the descriptors of a chain-only scan, 64 calls at a time: every
// lane reads its call, a ballot says which calls demodulated, and a call that did not takes the value of the nearest earlier one that did
// (a readlane by index), or the value carried in.  (One thread per stream walked the 96 descriptors one dependent load after the other: 39 us.)
__global__ __launch_bounds__(256) void sync_carry_kernel(CallDesc* __restrict__ descs, int max_calls, const int64_t* __restrict__ nbytes, const int* __restrict__ calls_before,
                                                         StreamState* __restrict__ states, const int* __restrict__ violation, int nstreams)
{
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= nstreams || violation[b] != 0x7f7f7f7f) return;   // streams that broke the assumption are scanned again in full
  const int ncalls = static_cast<int>(nbytes[b] / kChunkBytes) - calls_before[b];
  double carried = states[b].fine_freq_shift;              // the chain-only scan left the incoming value untouched
  for (int k0 = 0; k0 < ncalls; k0 += 64) {
    const int k = k0 + lane;
    CallDesc* d = descs + static_cast<size_t>(b) * max_calls + k;
    const bool valid = k < ncalls, demod = valid && d->status == 2;
    const double own = demod ? d->fine_freq_shift : 0.0;
    const unsigned long long mask = __ballot(demod);
    const unsigned long long earlier = mask & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));   // demodulated calls at or before this lane
    const int src = earlier ? 63 - __clzll(earlier) : -1;
    const long long bits = __builtin_bit_cast(long long, own);
    const int lo = __shfl(static_cast<int>(bits), src < 0 ? 0 : src), hi = __shfl(static_cast<int>(bits >> 32), src < 0 ? 0 : src);
    const double v = src < 0 ? carried : __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
    if (valid && !demod) d->fine_freq_shift = v;
    // what the next 64 calls start from: the value of this group's last demodulated call, if any
    if (mask) {
      const int last = 63 - __clzll(mask);
      const int l2 = __shfl(static_cast<int>(bits), last), h2 = __shfl(static_cast<int>(bits >> 32), last);
      carried = __builtin_bit_cast(double, (static_cast<long long>(h2) << 32) | static_cast<unsigned>(l2));
    }
  }
  if (lane == 0) states[b].fine_freq_shift = carried;
}

}  // namespace

size_t sync_scan_lds_bytes() { return sizeof(double2) * (2048 + kBatchPoints + 1024 + kSpecBins) + sizeof(Shared); }
static size_t sync_verify32_lds_bytes() { return sizeof(float2) * (kVerify32Points + 1024 + kSpecBins + (kSpecBins & 1)) + sizeof(Shared); }

static hipError_t sync_attr()
{
  static std::once_flag once[64];
  static hipError_t result[64];
  return once_per_device(once, result, []() {
    const int lds = static_cast<int>(sync_scan_lds_bytes());
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sync_scan_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sync_scan_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sync_ahead_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sync_verify_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sync_verify32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sync_verify32_lds_bytes()));
    return e;
  });
}

// Everything a scan needs on the device before its first kernel, in ONE launch: the per-stream arrays straight out of the caller's page-locked
// host arrays (a few KB over the link), the descriptors and call infos cleared, the violation marks set, the incoming states kept for a rescan.
// (Nine small copy-engine operations before: each costs 5 .. 10 us of idle GPU between two of them.)
namespace {
__global__ __launch_bounds__(256) void scan_setup_kernel(ScanSetupArgs a)
{
  const size_t tid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x, step = static_cast<size_t>(gridDim.x) * blockDim.x;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (size_t i = tid; i < a.desc_vec; i += step) a.descs[i] = zero;
  for (size_t i = tid; i < a.info_vec; i += step) a.info[i] = zero;
  constexpr size_t kStateWords = sizeof(StreamState) / 4;
  const size_t nstate_words = static_cast<size_t>(a.nstreams) * kStateWords;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(a.h_states ? a.h_states : a.states);      // host array (fresh decode) or what the device holds (session)
  for (size_t i = tid; i < nstate_words; i += step) {
    const uint32_t w = src[i];
    if (a.h_states) reinterpret_cast<uint32_t*>(a.states)[i] = w;
    if (a.states_prev) reinterpret_cast<uint32_t*>(a.states_prev)[i] = w;
  }
  // the streams' tail bytes (device_types.hpp: kTailBytes): all zero at the start of a capture (sdr_init callocs the frame buffer, input_sdr.c:167-186);
  // the incoming ones kept for a rescan like the states
  if (a.tail_state) {
    const size_t ntail_words = static_cast<size_t>(a.nstreams) * (kTailBytes / 4);
    for (size_t i = tid; i < ntail_words; i += step) {
      const uint32_t w = a.h_states ? 0u : reinterpret_cast<const uint32_t*>(a.tail_state)[i];
      if (a.h_states) reinterpret_cast<uint32_t*>(a.tail_state)[i] = 0u;
      if (a.tail_state_prev) reinterpret_cast<uint32_t*>(a.tail_state_prev)[i] = w;
    }
  }
  for (size_t b = tid; b < static_cast<size_t>(a.nstreams); b += step) {
    a.iq_ptrs[b] = a.h_ptrs[b];
    a.nbytes[b] = a.h_nbytes[b];
    if (a.calls_before) a.calls_before[b] = a.h_calls_before[b];
    if (a.viol) a.viol[b] = 0x7f7f7f7f;
  }
  if (tid == 0 && a.viol) a.viol[a.nstreams] = 0;
}
}  // namespace
// A few small page-locked host arrays to the device (and a small region cleared) in one launch: the frame lists behind the layout
namespace {
__global__ __launch_bounds__(256) void host_words_kernel(HostWordsArgs a)
{
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    for (uint32_t i = tid; i < a.nwords[k]; i += step) a.dst[k][i] = a.src[k][i];
  for (uint32_t i = tid; i < a.nzero; i += step) a.zero[i] = 0;
}
}  // namespace
hipError_t launch_host_words(const HostWordsArgs& a, hipStream_t stream)
{
  uint32_t most = a.nzero;
  for (int k = 0; k < 4; ++k) most = most > a.nwords[k] ? most : a.nwords[k];
  if (most == 0) return hipSuccess;
  const int blocks = static_cast<int>(most / 256 + 1 > 256 ? 256 : most / 256 + 1);
  hipLaunchKernelGGL(host_words_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_scan_setup(const ScanSetupArgs& a, hipStream_t stream)
{
  static_assert(sizeof(StreamState) % 4 == 0, "copied word by word");
  const size_t most = a.desc_vec > a.info_vec ? a.desc_vec : a.info_vec;
  const int blocks = static_cast<int>(most / 256 / 4 < 64 ? 64 : (most / 256 / 4 > 2048 ? 2048 : most / 256 / 4));
  hipLaunchKernelGGL(scan_setup_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_sync_scan(const uint8_t* const* iq, const int64_t* nbytes, StreamState* states, CallDesc* descs, int2* info,
                            int nstreams, int max_calls, int call_begin, int call_end, const double2* tw2048,
                            const double2* tw1536, const uint8_t* prs_q, int afc, hipStream_t stream, bool chain_only,
                            const StreamState* states_in, const int* stream_list, SyncTails tails, SpecArgs spec)
{
  if (tails.chunk < 0) tails.chunk = kChunkBytes;
  if (nstreams <= 0) return hipSuccess;
  hipError_t e = sync_attr();
  if (e != hipSuccess) return e;
  const size_t lds = sync_scan_lds_bytes();
  if (!states_in) states_in = states;
  if (chain_only)
    hipLaunchKernelGGL(sync_scan_kernel<true>, dim3(nstreams), dim3(kThreads), lds, stream, iq, nbytes, states_in, states, stream_list, descs, info,
                       max_calls, call_begin, call_end, tw2048, tw1536, prs_q, afc, tails, spec);
  else
    hipLaunchKernelGGL(sync_scan_kernel<false>, dim3(nstreams), dim3(kThreads), lds, stream, iq, nbytes, states_in, states, stream_list, descs, info,
                       max_calls, call_begin, call_end, tw2048, tw1536, prs_q, afc, tails, spec);
  return hipGetLastError();
}

hipError_t launch_sync_ahead(const uint8_t* const* iq, const int64_t* nbytes, const StreamState* states, int nstreams, const double2* tw2048, const double2* tw1536,
                             const uint8_t* prs_q, hipStream_t stream, const SpecArgs& spec)
{
  if (nstreams <= 0 || spec.nspec <= 0 || spec.nhyp <= 0) return hipSuccess;
  hipError_t e = sync_attr();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(sync_ahead_kernel, dim3(spec.nhyp, spec.nspec, nstreams), dim3(kThreads), sync_scan_lds_bytes(), stream, iq, nbytes, states, tw2048, tw1536, prs_q,
                     kChunkBytes, spec);
  return hipGetLastError();
}

hipError_t launch_sync_verify(const uint8_t* const* iq, const int64_t* nbytes, const int* calls_before, StreamState* states, CallDesc* descs,
                              int nstreams, int max_calls, const double2* tw2048, const uint8_t* prs_q, int* violation, bool carry_only, hipStream_t stream)
{
  if (nstreams <= 0 || max_calls <= 0) return hipSuccess;
  hipError_t e = sync_attr();
  if (e != hipSuccess) return e;
  if (!carry_only) {
    const int blocks = max_calls * nstreams;                // one call per workgroup measured better than persistent ones (0.59 vs 0.63 ms)
    // fp32 first, fp64 for what that pass left undecided (DABHIP_VERIFY_FP32=0: everything in fp64, as before round 3)
    // (= 2: test mode, the fp32 pass runs but hands every call on)
    static const int mode = std::getenv("DABHIP_VERIFY_FP32") ? std::atoi(std::getenv("DABHIP_VERIFY_FP32")) : 1;
    const bool fp32_first = mode != 0;
    if (fp32_first)
      hipLaunchKernelGGL(sync_verify32_kernel, dim3(blocks), dim3(kThreads), sync_verify32_lds_bytes(), stream, iq, descs, max_calls, nstreams, tw2048, prs_q, violation,
                         mode == 2 ? 1 : 0);
    // behind the fp32 pass the fp64 kernel normally finds nothing to do: a small persistent grid looks through the descriptors then
    hipLaunchKernelGGL(sync_verify_kernel, dim3(fp32_first ? std::min(blocks, 1024) : blocks), dim3(kThreads), sync_scan_lds_bytes(), stream, iq, descs, max_calls,
                       nstreams, tw2048, prs_q, violation, fp32_first ? 1 : 0);
  } else {
    hipLaunchKernelGGL(sync_carry_kernel, dim3((nstreams + 3) / 4), dim3(256), 0, stream, descs, max_calls, nbytes, calls_before, states, violation, nstreams);
  }
  return hipGetLastError();
}


#if DABHIP_SYNC_TIMES
}  // namespace dabhip
extern "C" int dabhip_debug_sync_times(unsigned long long* out)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dabhip::g_sync_times), sizeof(unsigned long long) * 96 * 16) == hipSuccess ? 0 : -1;
}
namespace dabhip {
#endif
}  // namespace dabhip
