// VERDICT r5 item 8, the bounded probe of the idle matrix pipe: can a matrix instruction stream run BESIDE the packed-fp32 VALU stream of the OFDM
// transform (k_fused.hip is VALU-issue-bound, the matrix cores idle)?  Three loops per wave, timed with s_memtime over many iterations, 1..4 waves per SIMD:
//   valu      : 48 independent v_pk_fma_f32 per iteration (a radix-8 stage with its factors is ~42 packed instructions per thread)
//   mfma      : 4 x v_mfma_f32_16x16x4_f32 per iteration (one 16 x 16 real = 8 x 8 complex DFT matrix applied to 16 columns: K = 16 = 4 instructions;
//               fp32 in, fp32 accumulate: bit-exact fmaf chains, no precision question) -- and 4 x v_mfma_f32_16x16x32_bf16 (the int8 samples are exact in bf16)
//   both      : the two interleaved in ONE wave's instruction stream
// If `both` ~ max(valu, mfma) the pipes overlap; if ~ valu + mfma they do not.  What the transform could gain is bounded by the VALU work of the stage moved:
// stage A's butterflies are 28 of the ~335 VALU instructions per thread and transform (8 %), its factors cannot move (they differ per column).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_coissue.hip -o /tmp/mfma_coissue && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int kMode>   // 1 = valu, 2 = mfma f32, 3 = both (f32), 4 = mfma bf16, 5 = both (bf16)
__global__ __launch_bounds__(256) void loop_kernel(float* out, long long* cycles, int iters)
{
  v2f a[12];
  for (int i = 0; i < 12; ++i) a[i] = v2f{threadIdx.x * 1.0f + i, 0.5f * i};
  const v2f m = v2f{1.0001f, 0.9999f}, c = v2f{0.001f, -0.001f};
  v4f acc[4] = {v4f{0, 0, 0, 0}, v4f{0, 0, 0, 0}, v4f{0, 0, 0, 0}, v4f{0, 0, 0, 0}};
  const float fa = threadIdx.x * 0.25f, fb = 1.0f + threadIdx.x * 0.125f;
  v8bf ba, bb;
  for (int i = 0; i < 8; ++i) { ba[i] = static_cast<__bf16>(1.0f + i); bb[i] = static_cast<__bf16>(0.5f * i); }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (kMode == 2 || kMode == 3) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[q], 0, 0, 0);
      if (kMode == 4 || kMode == 5) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[q], 0, 0, 0);
      if (kMode == 1 || kMode == 3 || kMode == 5) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 12; ++i) s += a[i].x + a[i].y;
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int kMode>
double run(int blocks, int threads, int iters)
{
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipLaunchKernelGGL(loop_kernel<kMode>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 16);
  hipLaunchKernelGGL(loop_kernel<kMode>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double sum = 0;
  for (long long v : h) sum += static_cast<double>(v);
  hipFree(out);
  hipFree(cyc);
  return sum / blocks / iters;          // s_memtime ticks per iteration (one per shader clock on gfx950 as measured: 48 v_pk_fma_f32 = 248 ticks)
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, iters = 20000;
  std::printf("# %s, %d CUs; s_memtime ticks (shader clocks) per iteration of {48 v_pk_fma_f32} / {4 mfma} / both, one wave's stream; waves per SIMD 1, 2, 4\n", p.name, cus);
  std::printf("# waves/SIMD  valu      mfma_f32  both_f32  mfma_bf16 both_bf16   both_f32/(valu+mfma)  both_f32/max\n");
  for (int wps : {1, 2, 4}) {
    const int threads = 256, blocks = cus * wps;      // one 4-wave workgroup per CU and wave-per-SIMD step
    const double v = run<1>(blocks, threads, iters), mf = run<2>(blocks, threads, iters), bf = run<3>(blocks, threads, iters), mb = run<4>(blocks, threads, iters),
                 bb = run<5>(blocks, threads, iters);
    std::printf("%10d  %8.4f  %8.4f  %8.4f  %8.4f  %8.4f   %8.3f  %8.3f\n", wps, v, mf, bf, mb, bb, bf / (v + mf), bf / (v > mf ? v : mf));
  }
  return 0;
}
