// What HBM delivers to simple streaming kernels on this part: fill (write only), copy (read + write) and a K2-like mix
// (1 byte read per 4 bytes written).  hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_rates.hip -o variants/hbm_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(4))) vfloat4;

template <bool kNt>
__global__ __launch_bounds__(256) void fill_kernel(vfloat4* dst, size_t n)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const vfloat4 v = {float(i), 1.0f, 2.0f, 3.0f};
    if (kNt) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
  }
}
template <bool kNt>
__global__ __launch_bounds__(256) void copy_kernel(vfloat4* dst, const vfloat4* src, size_t n)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const vfloat4 v = kNt ? __builtin_nontemporal_load(&src[i]) : src[i];
    if (kNt) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
  }
}
// K2's mix: every thread reads 4 bytes and writes 16
__global__ __launch_bounds__(256) void mix_kernel(vfloat4* dst, const unsigned* src, size_t n)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned w = src[i];
    const vfloat4 v = {float(w & 255), float((w >> 8) & 255), float((w >> 16) & 255), float(w >> 24)};
    __builtin_nontemporal_store(v, &dst[i]);
  }
}
template <typename F>
static void timeit(const char* name, double bytes, F&& launch)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms / 5, bytes / (ms / 5) * 1e-6);
}
int main()
{
  const size_t bytes = size_t(4) << 30, n = bytes / 16;
  vfloat4 *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  for (int blocks : {256 * 8, 256 * 32, 256 * 128}) {
    printf("grid %d x 256\n", blocks);
    timeit("fill, plain stores", bytes, [&] { hipLaunchKernelGGL(fill_kernel<false>, dim3(blocks), dim3(256), 0, 0, a, n); });
    timeit("fill, nontemporal stores", bytes, [&] { hipLaunchKernelGGL(fill_kernel<true>, dim3(blocks), dim3(256), 0, 0, a, n); });
    timeit("copy, plain", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    timeit("copy, nontemporal", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_kernel<true>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    timeit("mix 1 read : 4 written (K2)", 1.25 * bytes, [&] { hipLaunchKernelGGL(mix_kernel, dim3(blocks), dim3(256), 0, 0, a, reinterpret_cast<const unsigned*>(b), n); });
  }
  timeit("hipMemsetAsync", bytes, [&] { hipMemsetAsync(a, 0, bytes, 0); });
  timeit("hipMemcpyAsync D2D", 2.0 * bytes, [&] { hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
