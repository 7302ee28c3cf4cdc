// valu_rate.cpp — issue rate of a few VALU opcodes on gfx950 (cycles per wave64 instruction per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.cpp -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(unsigned* out, int iters)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = blockIdx.x | 1;
  for (int i = 0; i < iters; ++i) {
#define BODY(ins) asm volatile(REP16(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n") \
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
    if (OP == 0) BODY("v_add_u32");
    if (OP == 1) BODY("v_pk_add_u16");
    if (OP == 2) BODY("v_pk_max_i16");
    if (OP == 3) BODY("v_max_i32");
    if (OP == 4) BODY("v_add_f32");
    if (OP == 5) BODY("v_pk_sub_i16");
    if (OP == 6) BODY("v_pk_lshrrev_b16");
    if (OP == 7) BODY("v_and_b32");
    if (OP == 8) BODY("v_max_f32");
#define BODY3(ins) asm volatile(REP16(ins " %0, %0, %8, %1\n" ins " %1, %1, %8, %2\n" ins " %2, %2, %8, %3\n" ins " %3, %3, %8, %4\n" ins " %4, %4, %8, %5\n" ins " %5, %5, %8, %6\n" ins " %6, %6, %8, %7\n" ins " %7, %7, %8, %0\n") \
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
    if (OP == 10) BODY("v_lshrrev_b32");
    if (OP == 11) BODY3("v_and_or_b32");
    if (OP == 12) BODY3("v_lshl_or_b32");
    if (OP == 13) BODY3("v_or3_b32");
    if (OP == 14) BODY3("v_perm_b32");
    if (OP == 15) BODY3("v_bfi_b32");
    if (OP == 16) BODY("v_bcnt_u32_b32");
    if (OP == 17) BODY("v_sub_u32");
    if (OP == 18) BODY("v_xor_b32");
    if (OP == 19) BODY3("v_alignbit_b32");
    if (OP == 20) BODY("v_mul_u32_u24");
    if (OP == 21) BODY3("v_bfe_u32");
    if (OP == 22) BODY3("v_lshl_add_u32");
    if (OP == 23) BODY3("v_add3_u32");
    if (OP == 24) BODY("v_pk_ashrrev_i16");
    if (OP == 25) BODY3("v_xad_u32");
    if (OP == 26) BODY3("v_sad_u8");
    if (OP == 27) BODY("v_min_u32");
    if (OP == 28) BODY3("v_max3_i32");
    if (OP == 29) BODY3("v_pk_mad_u16");
    if (OP == 30) BODY("v_max_i16");
    if (OP == 31) BODY("v_add_u16");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
void run(const char* name, unsigned* d)
{
  const int blocks = 256 * 8, threads = 256, iters = 2000;   // 8 waves per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<OP><<<blocks, threads>>>(d, 10);
  hipEventRecord(a);
  k<OP><<<blocks, threads>>>(d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double insts_per_simd = double(blocks) * 4 /*waves*/ * iters * 128 / 1024.0;
  printf("%-18s %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}

int main()
{
  unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_u32", d); run<1>("v_pk_add_u16", d); run<2>("v_pk_max_i16", d); run<3>("v_max_i32", d); run<4>("v_add_f32", d);
  run<5>("v_pk_sub_i16", d); run<6>("v_pk_lshrrev_b16", d); run<7>("v_and_b32", d); run<8>("v_max_f32", d);
  run<10>("v_lshrrev_b32", d); run<11>("v_and_or_b32", d); run<12>("v_lshl_or_b32", d); run<13>("v_or3_b32", d); run<14>("v_perm_b32", d);
  run<15>("v_bfi_b32", d); run<16>("v_bcnt_u32_b32", d); run<17>("v_sub_u32", d); run<18>("v_xor_b32", d); run<19>("v_alignbit_b32", d);
  run<20>("v_mul_u32_u24", d); run<21>("v_bfe_u32", d); run<22>("v_lshl_add_u32", d); run<23>("v_add3_u32", d); run<24>("v_pk_ashrrev_i16", d);
  run<25>("v_xad_u32", d); run<26>("v_sad_u8", d); run<27>("v_min_u32", d); run<28>("v_max3_i32", d); run<29>("v_pk_mad_u16", d);
  run<30>("v_max_i16", d); run<31>("v_add_u16", d);
  return 0;
}
