// What v_permlane16_swap / v_permlane32_swap (gfx950) do to a pair of registers, lane by lane: a = lane, b = 100 + lane before.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/lane_swap_check.hip -o variants/lane_swap_check
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(unsigned* out)
{
  const unsigned lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + lane] = s[0];
  out[192 + lane] = s[1];
}

int main()
{
  unsigned* d;
  unsigned h[256];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const char* names[4] = {"swap16 [0]", "swap16 [1]", "swap32 [0]", "swap32 [1]"};
  for (int v = 0; v < 4; ++v) {
    printf("%s:", names[v]);
    for (int l = 0; l < 64; l += 8) printf(" %u", h[64 * v + l]);
    printf("\n");
  }
  return 0;
}
