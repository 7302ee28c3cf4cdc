// k_synth.hip — device-side synthetic Mode-I modulator (the reference has no transmitter).
//
// The bit content of every transmission frame (FIBs, payload, convolutional code, puncturing, time interleaving:
// SymbolBits in synth.cpp) is prepared on the host cores, 28,800 bytes per TF; everything sample-sized happens here:
// differential QPSK phase accumulation, carrier mapping, the 76 inverse 2048-point DFTs, cyclic prefixes, carrier
// offset, AWGN and cu8 quantisation, written straight into the caller's device buffers (393,216 bytes per TF).
// Same signal as dabhip_synth_generate (synth.cpp); samples may differ by one LSB where fp32 and fp64 round apart.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dabhip.h"
#include "dab_tables.hpp"
#include "synth.hpp"

namespace dabhip {
void set_error(const std::string& msg);

namespace {

struct ModStream {
  uint8_t* out;           // device cu8 buffer of this stream
  const uint32_t* bits;   // packed data-symbol bits [tf][75][96 words], bit i of a symbol at word i >> 5, bit i & 31
  uint64_t seed;
  double cfo_turns;       // carrier offset in turns per sample
  float amplitude, noise_rms;
  int32_t skip_samples, ntf;
};

__host__ __device__ inline uint64_t mix64(uint64_t z)
{
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
// the noise keys of synth.cpp (domain 3): one counter per emitted sample, two uniforms per counter
__host__ __device__ inline uint64_t noise_key(uint64_t seed, uint64_t ctr, uint64_t which) { return mix64(mix64(mix64(seed ^ mix64(3)) + ctr) + which); }

constexpr int kThreads = 256;
constexpr int kSymsPerBlock = 19;   // 4 workgroups per TF
__constant__ float kCos8[8] = {1.0f, 0.70710678f, 0.0f, -0.70710678f, -1.0f, -0.70710678f, 0.0f, 0.70710678f};

__device__ __forceinline__ void emit(const ModStream& st, long long g, float xr, float xi)
{
  if (st.cfo_turns != 0.0) {
    double t = st.cfo_turns * static_cast<double>(g);
    t -= floor(t);
    double sn, cs;
    sincospi(2.0 * t, &sn, &cs);
    const float r = static_cast<float>(xr * cs - xi * sn);
    xi = static_cast<float>(xr * sn + xi * cs);
    xr = r;
  }
  if (g < st.skip_samples) return;
  const long long j = g - st.skip_samples;
  double a = xr, b = xi;
  if (st.noise_rms > 0.0f) {
    const uint64_t ka = noise_key(st.seed, static_cast<uint64_t>(j), 0), kb = noise_key(st.seed, static_cast<uint64_t>(j), 1);
    const double u1 = (static_cast<double>(ka >> 11) + 1.0) / 9007199254740993.0;
    const double u2 = static_cast<double>(kb >> 11) / 9007199254740992.0;
    const double r = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    a += st.noise_rms * r * cs;
    b += st.noise_rms * r * sn;
  }
  a = floor(127.0 + a + 0.5);
  b = floor(127.0 + b + 0.5);
  a = a < 1 ? 1 : (a > 254 ? 254 : a);
  b = b < 1 ? 1 : (b > 254 ? 254 : b);
  const unsigned v = static_cast<unsigned>(a) | (static_cast<unsigned>(b) << 8);
  *reinterpret_cast<uint16_t*>(st.out + 2 * j) = static_cast<uint16_t>(v);
}

__global__ __launch_bounds__(kThreads) void modulate_kernel(const ModStream* __restrict__ streams, const uint8_t* __restrict__ prs_q,
                                                            const uint16_t* __restrict__ qpsk_of_carrier, const float2* __restrict__ twf)
{
  __shared__ float2 buf[2048];
  const ModStream st = streams[blockIdx.y];
  const int tf = blockIdx.x >> 2, part = blockIdx.x & 3, tid = threadIdx.x;
  if (tf >= st.ntf) return;
  const long long tf_base = static_cast<long long>(tf) * kTfSamples;
  const uint32_t* tfbits = st.bits + static_cast<size_t>(tf) * 75 * 96;

  // each thread owns 6 carriers; differential phase in eighth turns
  int ph[6], nq[6], pos[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int k = tid + kThreads * j;
    nq[j] = qpsk_of_carrier[k];
    ph[j] = 2 * prs_q[k];
    const int bin = k < 768 ? 1280 + k : k - 767;
    pos[j] = static_cast<int>(__brev(static_cast<unsigned>(bin)) >> 21);    // decimation in time: bit-reversed input order
  }
  auto advance = [&](int l) {        // symbol l >= 1 adds the QPSK increment of its bit pair to the phase of symbol l - 1
    const uint32_t* row = tfbits + static_cast<size_t>(l - 1) * 96;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const unsigned b0 = (row[nq[j] >> 5] >> (nq[j] & 31)) & 1u, b1 = (row[(1536 + nq[j]) >> 5] >> (nq[j] & 31)) & 1u;
      const unsigned code = b0 | (b1 << 1);                    // (b0,b1): 00 -> 45 deg, 10 -> 135, 01 -> 315, 11 -> 225
      ph[j] = (ph[j] + ((0x5731u >> (4 * code)) & 7u)) & 7;
    }
  };
  const int l0 = part * kSymsPerBlock;
  for (int l = 1; l < l0; ++l) advance(l);

  if (part == 0)
    for (int n = tid; n < kNullSamples; n += kThreads) emit(st, tf_base + n, 0.0f, 0.0f);

  for (int l = l0; l < l0 + kSymsPerBlock; ++l) {
    if (l > 0) advance(l);
    __syncthreads();                                           // the previous symbol has been read out
#pragma unroll
    for (int u = 0; u < 8; ++u) buf[tid + kThreads * u] = make_float2(0.0f, 0.0f);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 6; ++j) buf[pos[j]] = make_float2(st.amplitude * kCos8[ph[j]], st.amplitude * kCos8[(ph[j] + 6) & 7]);   // sin = cos(x - 90 deg)
    __syncthreads();
    // unnormalised inverse DFT, radix 2, in place: x[n] = sum X[k] exp(+2 pi i k n / 2048)
    for (int len = 2; len <= 2048; len <<= 1) {
      const int half = len >> 1, step = 2048 / len;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + kThreads * u;
        const int k = idx & (half - 1), a = ((idx - k) << 1) + k, b = a + half;
        const float2 w = twf[k * step];                        // (cos, -sin): conjugate for the synthesis direction
        const float2 A = buf[a], B = buf[b];
        const float tr = B.x * w.x + B.y * w.y, ti = B.y * w.x - B.x * w.y;
        buf[a] = make_float2(A.x + tr, A.y + ti);
        buf[b] = make_float2(A.x - tr, A.y - ti);
      }
      __syncthreads();
    }
    const long long sym_base = tf_base + kNullSamples + static_cast<long long>(kSymSamples) * l;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = tid + kThreads * u;
      const float2 x = buf[n];
      emit(st, sym_base + kCpSamples + n, x.x, x.y);
      if (n >= 2048 - kCpSamples) emit(st, sym_base + n - (2048 - kCpSamples), x.x, x.y);     // cyclic prefix
    }
  }
}

bool ok(hipError_t e, const char* what)
{
  if (e == hipSuccess) return true;
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return false;
}

template <class T>
struct Scoped {
  T* p = nullptr;
  ~Scoped() { if (p) (void)hipFree(p); }
  bool alloc(size_t n) { return hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)) == hipSuccess; }
};

}  // namespace

int synth_generate_device(const dabhip_synth_cfg* cfgs, int nstreams, int ntf, uint8_t* const* iq, int device)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device: the device modulator needs a GPU"); return -1; }
  if (!cfgs || !iq || nstreams <= 0 || ntf <= 0 || device < 0 || device >= ndev) { set_error("synth_generate_device: bad arguments"); return -1; }
  for (int i = 0; i < nstreams; ++i) {
    if (!synth_validate(cfgs[i])) return -1;
    if (synth_channel_active(cfgs[i])) { set_error("synth_generate_device: the channel stages (dabhip_channel_cfg) exist in the host generator only"); return -1; }
  }
  if (!ok(hipSetDevice(device), "hipSetDevice")) return -1;

  std::vector<float2> twf(2048);
  for (int k = 0; k < 2048; ++k) twf[k] = make_float2(static_cast<float>(std::cos(2 * M_PI * k / 2048)), static_cast<float>(-std::sin(2 * M_PI * k / 2048)));
  std::vector<uint8_t> prs(prs_quarter_turns().begin(), prs_quarter_turns().end());
  std::vector<uint16_t> qpsk(carrier_to_qpsk().begin(), carrier_to_qpsk().end());
  Scoped<float2> d_twf;
  Scoped<uint8_t> d_prs;
  Scoped<uint16_t> d_qpsk;
  const size_t words_per_stream = static_cast<size_t>(ntf) * 75 * 96;
  // chunks of streams bound the staging memory (28,800 bytes per TF and stream)
  const int chunk = static_cast<int>(std::max<size_t>(1, std::min<size_t>(nstreams, (size_t(256) << 20) / (words_per_stream * 4))));
  Scoped<uint32_t> d_bits;
  Scoped<ModStream> d_streams;
  if (!d_twf.alloc(2048) || !d_prs.alloc(prs.size()) || !d_qpsk.alloc(qpsk.size()) || !d_bits.alloc(words_per_stream * chunk) || !d_streams.alloc(chunk)) {
    set_error("synth_generate_device: hipMalloc failed");
    return -1;
  }
  if (!ok(hipMemcpy(d_twf.p, twf.data(), twf.size() * sizeof(float2), hipMemcpyHostToDevice), "table upload") ||
      !ok(hipMemcpy(d_prs.p, prs.data(), prs.size(), hipMemcpyHostToDevice), "table upload") ||
      !ok(hipMemcpy(d_qpsk.p, qpsk.data(), qpsk.size() * 2, hipMemcpyHostToDevice), "table upload"))
    return -1;

  std::vector<uint32_t> h_bits(words_per_stream * chunk);
  std::vector<ModStream> h_streams(chunk);
  const int nthreads = std::max(1, std::min<int>(static_cast<int>(std::thread::hardware_concurrency()), 32));
  for (int first = 0; first < nstreams; first += chunk) {
    const int n = std::min(chunk, nstreams - first);
    // bit content on the host cores, one stream per task
    std::vector<std::thread> workers;
    for (int w = 0; w < std::min(nthreads, n); ++w)
      workers.emplace_back([&, w]() {
        std::vector<uint8_t> symbits(static_cast<size_t>(kBitsPerSym) * 75);
        for (int i = w; i < n; i += std::min(nthreads, n)) {
          SymbolBits gen(cfgs[first + i]);
          uint32_t* dst = h_bits.data() + words_per_stream * i;
          for (int tf = 0; tf < ntf; ++tf) {
            gen.next_tf(symbits.data());
            for (int wd = 0; wd < 75 * 96; ++wd) {
              uint32_t x = 0;
              const uint8_t* b = symbits.data() + 32 * static_cast<size_t>(wd);
              for (int k = 0; k < 32; ++k) x |= static_cast<uint32_t>(b[k] & 1u) << k;
              dst[static_cast<size_t>(tf) * 75 * 96 + wd] = x;
            }
          }
        }
      });
    for (auto& t : workers) t.join();
    for (int i = 0; i < n; ++i) {
      const dabhip_synth_cfg& c = cfgs[first + i];
      h_streams[i] = ModStream{iq[first + i], d_bits.p + words_per_stream * i, c.seed, c.cfo_hz / 2048000.0, static_cast<float>(c.amplitude),
                               static_cast<float>(synth_noise_rms(c)), c.skip_samples, ntf};
    }
    if (!ok(hipMemcpy(d_bits.p, h_bits.data(), words_per_stream * n * 4, hipMemcpyHostToDevice), "bit upload") ||
        !ok(hipMemcpy(d_streams.p, h_streams.data(), sizeof(ModStream) * n, hipMemcpyHostToDevice), "stream upload"))
      return -1;
    hipLaunchKernelGGL(modulate_kernel, dim3(4 * ntf, n), dim3(kThreads), 0, nullptr, d_streams.p, d_prs.p, d_qpsk.p, d_twf.p);
    if (!ok(hipGetLastError(), "modulate launch") || !ok(hipDeviceSynchronize(), "modulate")) return -1;
  }
  return 0;
}

}  // namespace dabhip
