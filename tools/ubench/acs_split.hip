// VERDICT r5 item 4, the bounded version: what a trellis step costs when a code word's 64 path metrics are spread over 1, 2 or 4 lanes.
// The lane form (k_decode.hip) holds all 64 metrics of a code word in one lane: 32 packed registers, per step 64 v_add_u32 + 32 v_pk_max_u16 (+ ~21 others).
// Spread over 2 (4) lanes a lane holds 16 (8) registers and per step does 32 (16) adds -- half of them with a DPP quad_perm source, the partner lane's
// predecessor metrics -- and 16 (8) packed maxima.  This kernel runs exactly that instruction stream (dependent from step to step like the real recursion,
// operands otherwise arbitrary) and reports shader clocks per trellis step and wave at 1, 2, 4 waves per SIMD: the latency a mid-size batch is bound by
// (8 .. 32 streams: one lane-form wave per SIMD walks 4,614 dependent steps, 1.45 ms).  Not a decoder: no branch-metric look-up, records or layout algebra.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/acs_split.hip -o /tmp/acs_split && /tmp/acs_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int kRegs, bool kDpp>
__global__ __launch_bounds__(256) void acs_kernel(unsigned* out, long long* cycles, int steps)
{
  unsigned m[kRegs], n[kRegs];
  for (int i = 0; i < kRegs; ++i) m[i] = (threadIdx.x * 2654435761u + i * 40503u) & 0x0fff0fffu;
  const unsigned bl = 0x00110013u, bh = 0x00120010u;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int r = 0; r < kRegs / 2; ++r) {
      // one butterfly pair: two candidates for each of two result registers, one packed max each (viterbi.c:404-421 in the pair layouts of k_decode.hip)
      unsigned a0, a1, a2, a3;
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(a0) : "v"(m[r]), "v"(bl));
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(a2) : "v"(m[r]), "v"(bh));
      if (kDpp) {   // the high predecessors live in the partner lane
        asm volatile("v_add_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a1) : "v"(m[kRegs / 2 + r]), "v"(bh));
        asm volatile("v_add_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a3) : "v"(m[kRegs / 2 + r]), "v"(bl));
      } else {
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(a1) : "v"(m[kRegs / 2 + r]), "v"(bh));
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(a3) : "v"(m[kRegs / 2 + r]), "v"(bl));
      }
      asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(n[2 * r]) : "v"(a0), "v"(a1));
      asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(n[2 * r + 1]) : "v"(a2), "v"(a3));
    }
#pragma unroll
    for (int i = 0; i < kRegs; ++i) m[i] = n[i] & 0x3fff3fffu;       // (keeps the values bounded; the real kernel clears its tags in the re-pairing permute)
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  unsigned acc = 0;
  for (int i = 0; i < kRegs; ++i) acc ^= m[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int kRegs, bool kDpp>
double run(int blocks, int steps)
{
  unsigned* out;
  long long* cyc;
  hipMalloc(&out, sizeof(unsigned) * blocks * 256);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipLaunchKernelGGL((acs_kernel<kRegs, kDpp>), dim3(blocks), dim3(256), 0, 0, out, cyc, 64);
  hipLaunchKernelGGL((acs_kernel<kRegs, kDpp>), dim3(blocks), dim3(256), 0, 0, out, cyc, steps);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double sum = 0;
  for (long long v : h) sum += static_cast<double>(v);
  hipFree(out);
  hipFree(cyc);
  return sum / blocks / steps;
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, steps = 20000;
  std::printf("# %s, %d CUs; shader clocks (s_memtime ticks) per trellis step and wave: the add-compare-select stream of a code word held in 1 / 2 / 4 lanes\n", p.name, cus);
  std::printf("# (the AND that bounds the values is part of the stream: 32 / 16 / 8 more instructions; the lane form's real step is 117 instructions, 96 of them these)\n");
  std::printf("# waves/SIMD   1 lane (32 regs)   2 lanes (16 regs, DPP)   4 lanes (8 regs, DPP)   2 lanes, no DPP   4 lanes, no DPP\n");
  for (int wps : {1, 2, 4}) {
    const int blocks = cus * wps;
    std::printf("%10d   %14.1f   %20.1f   %19.1f   %15.1f   %15.1f\n", wps, run<32, false>(blocks, steps), run<16, true>(blocks, steps), run<8, true>(blocks, steps),
                run<16, false>(blocks, steps), run<8, false>(blocks, steps));
  }
  return 0;
}
