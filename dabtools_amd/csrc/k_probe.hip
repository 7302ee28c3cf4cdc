// k_probe.hip — what HBM delivers to a bare streaming kernel on THIS device, measured beside K2 by bench.py: the yardstick the
// K2 roofline figure is read against (the 8 TB/s of the data sheet is not reachable by any kernel).  Three mixes over the same
// footprint: fill (write only), copy (1 read : 1 written), and K2's own mix (1 byte read per 4 bytes written, 16-byte nontemporal
// stores like ofdm_fft_kernel's).  Not part of the data path.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace dabhip {
namespace {

typedef float __attribute__((ext_vector_type(4))) vfloat4;

__global__ __launch_bounds__(256) void probe_fill_kernel(vfloat4* __restrict__ dst, size_t n)
{
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    __builtin_nontemporal_store(vfloat4{static_cast<float>(i), 1.0f, 2.0f, 3.0f}, &dst[i]);
}
__global__ __launch_bounds__(256) void probe_copy_kernel(vfloat4* __restrict__ dst, const vfloat4* __restrict__ src, size_t n)
{
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
__global__ __launch_bounds__(256) void probe_mix_kernel(vfloat4* __restrict__ dst, const unsigned* __restrict__ src, size_t n)
{
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned w = src[i];
    __builtin_nontemporal_store(vfloat4{static_cast<float>(w & 255u), static_cast<float>((w >> 8) & 255u), static_cast<float>((w >> 16) & 255u),
                                        static_cast<float>(w >> 24)}, &dst[i]);
  }
}

}  // namespace

// gbs[0..2] = fill, copy, K2 mix in GB/s (bytes moved / time), each the best of three grid sizes; `bytes` = size of the written buffer
int stream_ceiling(int device, size_t bytes, int reps, double* gbs)
{
  if (hipSetDevice(device) != hipSuccess) return -1;
  bytes &= ~static_cast<size_t>(4095);
  vfloat4 *a = nullptr, *b = nullptr;
  if (bytes == 0 || reps <= 0 || hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); return -1; }
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 2, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const size_t n = bytes / 16;
  const double moved[3] = {1.0 * bytes, 2.0 * bytes, 1.25 * bytes};
  int rc = 0;
  for (int mode = 0; mode < 3; ++mode) {
    gbs[mode] = 0;
    for (int blocks : {256 * 8, 256 * 32, 256 * 128}) {
      for (int r = -1; r < reps; ++r) {                    // r = -1: untimed
        if (r == 0) (void)hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(probe_fill_kernel, dim3(blocks), dim3(256), 0, 0, a, n);
        if (mode == 1) hipLaunchKernelGGL(probe_copy_kernel, dim3(blocks), dim3(256), 0, 0, a, b, n);
        if (mode == 2) hipLaunchKernelGGL(probe_mix_kernel, dim3(blocks), dim3(256), 0, 0, a, reinterpret_cast<const unsigned*>(b), n);
      }
      (void)hipEventRecord(e1, 0);
      float ms = 0;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0) { rc = -1; break; }
      const double rate = moved[mode] * reps / (ms * 1e-3) / 1e9;
      if (rate > gbs[mode]) gbs[mode] = rate;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  return rc;
}

}  // namespace dabhip
