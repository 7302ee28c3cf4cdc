// k_sync.hip — K1: per-stream synchronisation scan.
//
// Replaces, for every 262144-byte call of sdr_demod (input_sdr.c:27-112): the FIFO
// bookkeeping of sdr_fifo.c:26-61 (in closed form, as views into the resident IQ stream),
// dab_coarse_time_sync (sdr_sync.c:34-68), dab_fine_time_sync (:71-202),
// dab_coarse_freq_sync_2 (:205-258) and dab_fine_freq_corr (:259-302).
//
// The chain is sequential per stream (the timing correction found in TF n positions TF
// n+1) and independent across streams, so one 256-thread workgroup owns one stream for
// the whole scan: no inter-workgroup traffic, B workgroups in flight.  Arithmetic is fp64
// like the reference's FFTW calls (two 2048-point DFTs, one 1536-point and 29 128-point
// inverse DFTs per TF); arg-max decisions use the reference's float compare, first hit wins.
#include <hip/hip_runtime.h>

#include "dab_tables.hpp"
#include "device_types.hpp"
#include "fft64.hpp"
#include "fifo_view.hpp"
#include "kernels.hpp"

namespace dabhip {
namespace {

constexpr int kThreads = kFft64Threads;   // 512; one workgroup per stream: more threads = shorter butterfly stages

__device__ __forceinline__ int view_byte(const uint8_t* stream, const FrameView& v, int p)
{
  int i = 0;
  while (i < v.nseg - 1 && p >= v.seg_end[i]) ++i;
  const int64_t s = v.seg_src[i];
  return s < 0 ? 0 : stream[s + p];
}
// one IQ sample (I at the even byte p, Q at p + 1): segment boundaries and sources are even, so both bytes
// come from the same segment and one 2-byte load fetches them
// nco_hz != 0 (software AFC only): the sample is de-rotated by exp(-2 pi i nco n / fs), n = sample index in the frame
__device__ __forceinline__ double2 view_sample(const uint8_t* stream, const FrameView& v, int p, int nco_hz)
{
  int i = 0;
  while (i < v.nseg - 1 && p >= v.seg_end[i]) ++i;
  const int64_t s = v.seg_src[i];
  const unsigned w = s < 0 ? 0u : *reinterpret_cast<const uint16_t*>(stream + s + p);
  const double2 x = make_double2(static_cast<int8_t>(static_cast<uint8_t>((w & 0xff) - 127)), static_cast<int8_t>(static_cast<uint8_t>((w >> 8) - 127)));
  if (nco_hz == 0) return x;
  double sn, cs;
  sincospi(-2.0 * nco_hz * (p >> 1) / 2048000.0, &sn, &cs);
  return make_double2(x.x * cs - x.y * sn, x.x * sn + x.y * cs);
}
// u8 -> s8 with DC offset 127 and int8 wrap (input_sdr.c:60-63)
__device__ __forceinline__ int rail(int byte) { return static_cast<int>(static_cast<int8_t>(static_cast<uint8_t>(byte - 127))); }

constexpr int kWaves = kThreads / 64;
struct Red {
  float fv[kWaves];
  int iv[kWaves];
  double dv[kWaves];
};

// arg-max with the reference's semantics (strict '>' scanning upwards: lowest index wins ties);
// wave shuffles first, one LDS round across the waves
__device__ void block_argmax(Red& r, float v, int idx, float* out_v, int* out_i)
{
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const float ov = __shfl_xor(v, m);
    const int oi = __shfl_xor(idx, m);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { r.fv[threadIdx.x >> 6] = v; r.iv[threadIdx.x >> 6] = idx; }
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
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
  if ((threadIdx.x & 63) == 0) r.iv[threadIdx.x >> 6] = v;
  __syncthreads();
  int out = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) out += r.iv[w];
  __syncthreads();
  return out;
}

__device__ double block_sum_double(Red& r, double v)
{
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
  if ((threadIdx.x & 63) == 0) r.dv[threadIdx.x >> 6] = v;
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
  int status, do_sync, fifo_count;
  int coarse_fs;
  double fine_fs;
};

__global__ __launch_bounds__(kThreads) void sync_scan_kernel(const uint8_t* const* __restrict__ iq,
                                                             const int64_t* __restrict__ nbytes,
                                                             StreamState* __restrict__ states,
                                                             CallDesc* __restrict__ descs, int2* __restrict__ info, int max_calls, int call_begin,
                                                             int call_end, const double2* __restrict__ tw2048,
                                                             const double2* __restrict__ tw1536,
                                                             const uint8_t* __restrict__ prs_q, int afc)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2* A = reinterpret_cast<double2*>(smem);   // 2048: main DFT buffer
  double2* Bf = A + 2048;                            // 3712: 3 x 512 or 29 x 128 batch buffer
  double2* tw = Bf + 29 * 128;                       // 1024: exp(2 pi i k / 2048), LDS copy of the twiddle table
  Shared& sh = *reinterpret_cast<Shared*>(tw + 1024);
  uint8_t* env = reinterpret_cast<uint8_t*>(Bf);     // 19660 bytes, coarse search only

  const int b = blockIdx.x, tid = threadIdx.x;
  const uint8_t* stream = iq[b];
  const int64_t total_calls = nbytes[b] / kChunkBytes;
  const int kend = call_end < 0 ? static_cast<int>(total_calls) : min(call_end, static_cast<int>(total_calls));
  // call_begin < 0: continue where the stream's state stands (calls already fed), descriptors numbered from there
  const int kfirst = call_begin >= 0 ? call_begin : static_cast<int>(states[b].fed / kChunkBytes);
  const int kdesc0 = call_begin >= 0 ? 0 : kfirst;
  if (tid == 0) { sh.st = states[b]; sh.fine_fs = sh.st.fine_freq_shift; }
  for (int i = tid; i < 1024; i += kThreads) tw[i] = tw2048[i];
  __syncthreads();

  for (int k = kfirst; k < kend; ++k) {
    // ---- FIFO bookkeeping: input_sdr.c:36-55 over sdr_fifo.c:43-61 (fifo_view.hpp) -------
    if (tid == 0) {
      const FifoCall fc = fifo_call(sh.st);
      sh.status = fc.status;
      sh.do_sync = fc.do_sync;
      sh.fifo_count = fc.fifo_count;
      sh.coarse_fs = 0;
    }
    __syncthreads();

    const int nco = afc ? sh.st.tuner_hz : 0;
    if (sh.do_sync) {
      const FrameView& view = sh.st.view;
      // ---- coarse time: sdr_sync.c:34-68 ------------------------------------------------
      int e = 0;
      for (int n = tid; n < 266; n += kThreads) e += abs(rail(view_byte(stream, view, 20 * n)));
      e = block_sum_int(sh.red, e);
      int coarse = 0;
      if (!(e < 5000 && sh.st.force_timesync == 0)) {
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
        block_argmax(sh.red, -best, bestm, &bv, &bi);
        coarse = (-bv < 9999999.0f) ? bi * 20 : 0;
      }
      __syncthreads();
      if (tid == 0) { sh.st.coarse_timeshift = coarse; sh.st.force_timesync = 0; }
      __syncthreads();

      if (coarse == 0) {
        // ---- fine time: sdr_sync.c:71-202 -------------------------------------------------
        for (int n = tid; n < 2048; n += kThreads) {
          const int p = 2 * (kNullSamples + kCpSamples + n);
          A[n] = view_sample(stream, view, p, nco);
        }
        __syncthreads();
        dft_dif<11, 3, 3, 3, 2>(A, 1, -1.0, tw);
        for (int i = tid; i < kCarriers; i += kThreads) {
          const int bin = i < 768 ? i + 1280 : i - 765;
          const double2 c = mul_conj_prs(A[brev(bin, 11)], prs_q[i]);
          Bf[(i % 3) * 512 + i / 3] = c;       // decimate by 3 for the 3 x 512 inverse DFT
        }
        __syncthreads();
        dft_dif<9, 3, 3, 3>(Bf, 3, +1.0, tw);
        float fv = -99999.0f;
        int fi = 0x7fffffff;
        for (int kk = tid; kk < kCarriers; kk += kThreads) {
          const int r = brev(kk & 511, 9);
          const double2 f0 = Bf[r], f1 = Bf[512 + r], f2 = Bf[1024 + r];
          const double2 w1 = tw1536[kk], w2 = tw1536[(2 * kk) % 1536];
          const double xr = f0.x + (f1.x * w1.x - f1.y * w1.y) + (f2.x * w2.x - f2.y * w2.y);
          const double xi = f0.y + (f1.x * w1.y + f1.y * w1.x) + (f2.x * w2.y + f2.y * w2.x);
          const float mag = static_cast<float>(sqrt(xr * xr + xi * xi));
          if (mag > fv) { fv = mag; fi = kk; }     // kk ascending per thread: first maximum kept
        }
        float gv;
        int gi;
        block_argmax(sh.red, fv, fi, &gv, &gi);
        const int fine = gi < 768 ? gi * 2 + 16 : (gi - 1536) * 2;
        if (tid == 0) sh.st.fine_timeshift = fine;

        // ---- coarse frequency: input_sdr.c:90-109, sdr_sync.c:205-258 ---------------------
        for (int n = tid; n < 2048; n += kThreads) {
          const int p = 2 * (kNullSamples + kCpSamples + 1 + fine + n);
          A[n] = view_sample(stream, view, p, nco);
        }
        __syncthreads();
        dft_dif<11, 3, 3, 3, 2>(A, 1, -1.0, tw);
        for (int idx = tid; idx < 29 * 128; idx += kThreads) {
          const int kk = idx / 128 - 14, s = idx % 128;
          const int shifted = 14 + kk + 256 + s;              // index into the fftshifted spectrum
          const int bin = (shifted + 1024) & 2047;
          Bf[idx] = mul_conj_prs(A[brev(bin, 11)], prs_q[14 + s]);
        }
        __syncthreads();
        dft_dif<7, 3, 2, 2>(Bf, 29, +1.0, tw);
        // per-offset maximum |.|, then first maximum over offsets
        float cv = -99999.0f;
        int ci = 0x7fffffff;
        for (int idx = tid; idx < 29 * 128; idx += kThreads) {
          const double2 x = Bf[idx];
          const float mag = static_cast<float>(sqrt(x.x * x.x + x.y * x.y));
          if (mag > cv) { cv = mag; ci = idx / 128; }        // idx ascending per thread
        }
        float hv;
        int hi;
        block_argmax(sh.red, cv, ci, &hv, &hi);
        const int cfs = hi - 14;
        if (tid == 0) sh.coarse_fs = cfs;
        if (abs(cfs) > 1) {
          if (tid == 0) sh.st.force_timesync = 1;
        } else {
          // ---- fine frequency (estimate only): sdr_sync.c:259-302 -------------------------
          double acc = 0;
          for (int n = tid; n < kCpSamples; n += kThreads) {
            const int pl = 2 * (kNullSamples + 2048 + n), pr = 2 * (kNullSamples + n);
            const double2 l = view_sample(stream, view, pl, nco), r = view_sample(stream, view, pr, nco);
            const double lr = l.x, li = l.y, rr = r.x, ri = r.y;
            acc += atan2(-lr * ri + li * rr, lr * rr + li * ri);
          }
          acc = block_sum_double(sh.red, acc);
          if (tid == 0) {
            sh.fine_fs = acc / 504 / (2 * M_PI) * 1000;
            sh.status = 2;
          }
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      CallDesc& d = descs[static_cast<size_t>(b) * max_calls + (k - kdesc0)];
      d.status = sh.status;
      d.ordinal = sh.status == 2 ? sh.st.next_ordinal++ : -1;
      if (info) info[static_cast<size_t>(b) * max_calls + (k - kdesc0)] = make_int2(d.status, d.ordinal);   // what the host lays the frames out with
      d.coarse_timeshift = sh.st.coarse_timeshift;
      d.fine_timeshift = sh.st.fine_timeshift;
      d.coarse_freq_shift = sh.coarse_fs;
      d.fifo_count = sh.fifo_count;
      d.fine_freq_shift = sh.fine_fs;
      d.nco_hz = nco;
      d.view = sh.st.view;
      if (afc) {
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
  }
  if (tid == 0) { sh.st.fine_freq_shift = sh.fine_fs; states[b] = sh.st; }
}

}  // namespace

size_t sync_scan_lds_bytes() { return sizeof(double2) * (2048 + 29 * 128 + 1024) + sizeof(Shared); }

hipError_t launch_sync_scan(const uint8_t* const* iq, const int64_t* nbytes, StreamState* states, CallDesc* descs, int2* info,
                            int nstreams, int max_calls, int call_begin, int call_end, const double2* tw2048,
                            const double2* tw1536, const uint8_t* prs_q, int afc, hipStream_t stream)
{
  static bool attr_set = false;
  const size_t lds = sync_scan_lds_bytes();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sync_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(sync_scan_kernel, dim3(nstreams), dim3(kThreads), lds, stream, iq, nbytes, states, descs, info, max_calls,
                     call_begin, call_end, tw2048, tw1536, prs_q, afc);
  return hipGetLastError();
}

}  // namespace dabhip
