// k2_traffic_probe.cpp — what the memory system gives a kernel with K2's traffic and nothing else: per TF it reads
// 311,296 B of cu8 samples (2-byte loads, coalesced) and writes 1,245,184 B of complex64 (16-byte stores, 1 KiB per wave
// instruction, non-temporal or plain), with a trivial conversion in between.  The achievable ceiling for K2's roofline.
//   hipcc --offload-arch=gfx950 -O3 -o k2_traffic_probe tools/k2_traffic_probe.cpp && ./k2_traffic_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>

typedef float __attribute__((ext_vector_type(4))) vfloat4;

// the same with the next symbol's loads issued before the current symbol's stores (what K2 does)
template <int kSpb>
__global__ __launch_bounds__(256, 4) void probe_prefetch(const uint16_t* __restrict__ in, vfloat4* __restrict__ out)
{
  const size_t sym0 = static_cast<size_t>(blockIdx.x) * kSpb;
  unsigned raw[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) raw[r] = in[sym0 * 2048 + threadIdx.x + 256 * r];
#pragma unroll
  for (int s = 0; s < kSpb; ++s) {
    unsigned cur[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) cur[r] = raw[r];
    if (s + 1 < kSpb) {
      const uint16_t* src = in + (sym0 + s + 1) * 2048;
#pragma unroll
      for (int r = 0; r < 8; ++r) raw[r] = src[threadIdx.x + 256 * r];
    }
    vfloat4* dst = out + (sym0 + s) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const vfloat4 v = {static_cast<float>(cur[2 * k] & 0xff), static_cast<float>(cur[2 * k] >> 8), static_cast<float>(cur[2 * k + 1] & 0xff),
                         static_cast<float>(cur[2 * k + 1] >> 8)};
      __builtin_nontemporal_store(v, &dst[threadIdx.x + 256 * k]);
    }
  }
}

template <bool kNonTemporal>
__global__ __launch_bounds__(256) void probe(const uint16_t* __restrict__ in, vfloat4* __restrict__ out, int symbols_per_block)
{
  // one "symbol" = 2048 samples in, 2048 float2 out; thread handles samples tid + 256 r and output elements tid + 256 k
  const size_t sym0 = static_cast<size_t>(blockIdx.x) * symbols_per_block;
  for (int s = 0; s < symbols_per_block; ++s) {
    const uint16_t* src = in + (sym0 + s) * 2048;
    unsigned raw[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) raw[r] = src[threadIdx.x + 256 * r];
    vfloat4* dst = out + (sym0 + s) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const vfloat4 v = {static_cast<float>(raw[2 * k] & 0xff), static_cast<float>(raw[2 * k] >> 8), static_cast<float>(raw[2 * k + 1] & 0xff),
                         static_cast<float>(raw[2 * k + 1] >> 8)};
      if (kNonTemporal) __builtin_nontemporal_store(v, &dst[threadIdx.x + 256 * k]);
      else dst[threadIdx.x + 256 * k] = v;
    }
  }
}

int main()
{
  const int tfs = 4096, symbols = tfs * 76;
  uint16_t* in;
  vfloat4* out;
  if (hipMalloc(&in, static_cast<size_t>(symbols) * 4096) != hipSuccess || hipMalloc(&out, static_cast<size_t>(symbols) * 16384) != hipSuccess) return 1;
  (void)hipMemset(in, 1, static_cast<size_t>(symbols) * 4096);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int nt = 0; nt < 2; ++nt)
    for (int spb : {19, 4, 1}) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(a);
        if (nt) probe<true><<<symbols / spb, 256>>>(in, out, spb);
        else probe<false><<<symbols / spb, 256>>>(in, out, spb);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep) best = ms < best ? ms : best;
      }
      const double bytes = static_cast<double>(symbols) * (4096 + 16384);
      std::printf("%s stores, %2d symbols per workgroup: %.3f ms per 4096 TF = %.0f GB/s (%.1f %% of 8 TB/s)\n", nt ? "non-temporal" : "plain       ", spb, best,
                  bytes / best / 1e6, bytes / best / 1e6 / 80);
    }
  for (int v = 0; v < 2; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      (void)hipEventRecord(a);
      if (v == 0) probe_prefetch<19><<<symbols / 19, 256>>>(in, out);
      else probe_prefetch<4><<<symbols / 4, 256>>>(in, out);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      if (rep) best = ms < best ? ms : best;
    }
    const double bytes = static_cast<double>(symbols) * (4096 + 16384);
    std::printf("non-temporal stores, %2d symbols per workgroup, next symbol prefetched, 4 workgroups per CU: %.3f ms = %.0f GB/s (%.1f %% of 8 TB/s)\n", v ? 4 : 19, best,
                bytes / best / 1e6, bytes / best / 1e6 / 80);
  }
  return 0;
}
