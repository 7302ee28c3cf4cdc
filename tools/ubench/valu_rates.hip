// Issue rates of the VALU instructions the hot kernels lean on (wave64, gfx950): wave-instructions per cycle and SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o variants/valu_rates && variants/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int kOp>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters)
{
  v2f a[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { a[i] = v2f{threadIdx.x * 1.0f + i, 0.5f * i}; u[i] = threadIdx.x * 7u + i; }
  const v2f m = v2f{1.0001f, 0.9999f}, c = v2f{0.001f, -0.001f};
  const unsigned k = 0x00030001u;
  const unsigned kh = 0x3c004000u;                  // two f16: 1.0, 2.0
  if (kOp == 36 || kOp == 37 || kOp == 44 || kOp == 45 || kOp == 53) asm volatile("s_mov_b64 vcc, 0x5555" : : : "vcc");
  if (kOp == 49) asm volatile("s_mov_b64 vcc, exec" : : : "vcc");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#define OP(i)                                                                                                   \
  if (kOp == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));                     \
  if (kOp == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                        \
  if (kOp == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                    \
  if (kOp == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));                                    \
  if (kOp == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));                                   \
  if (kOp == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                       \
  if (kOp == 6) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                   \
  if (kOp == 7) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                   \
  if (kOp == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(k), "v"(0x07060100u));               \
  if (kOp == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(k));                             \
  if (kOp == 10) asm volatile("v_min_f32 %0, |%0|, |%1|" : "+v"(a[i].x) : "v"(c.x));                             \
  if (kOp == 11) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i].x));                                            \
  if (kOp == 12) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x));                                               \
  if (kOp == 13) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&a[i])) : "v"(*reinterpret_cast<const double*>(&m))); \
  if (kOp == 14) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&a[i])) : "v"(*reinterpret_cast<const double*>(&m))); \
  if (kOp == 15) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i].x)); \
  if (kOp == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[2:3]" : "+v"(u[i]) : "v"(k));                       \
  if (kOp == 17) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i].x), "v"(c.x) : "vcc");                        \
  if (kOp == 18) asm volatile("v_cmp_lt_f32_e64 s[4:5], %0, %1" : : "v"(a[i].x), "v"(c.x) : "s4", "s5");            \
  if (kOp == 19) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                       \
  if (kOp == 20) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(u[i]));                                             \
  if (kOp == 21) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(k));                                   \
  if (kOp == 22) asm volatile("v_bfe_i32 %0, %0, 0, 8" : "+v"(u[i]));                                              \
  if (kOp == 23) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "+v"(u[i])); \
  if (kOp == 24) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));                                   \
  if (kOp == 25) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                       \
  if (kOp == 26) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(k));                                           \
  if (kOp == 27) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(k));                                  \
  if (kOp == 28) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));                        \
  if (kOp == 29) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));                                   \
  if (kOp == 30) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(k));                              \
  if (kOp == 31) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                       \
  if (kOp == 32) asm volatile("v_add_f32_e64 %0, |%0|, |%1|" : "+v"(a[i].x) : "v"(c.x));                           \
  if (kOp == 33) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(m), "v"(c)); \
  if (kOp == 34) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(k));                               \
  if (kOp == 35) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(u[i]) : "v"(k));                                  \
  if (kOp == 36) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(k) : );                           \
  if (kOp == 37) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i]) : "v"(k), "v"(threadIdx.x));            \
  if (kOp == 38) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));                                   \
  if (kOp == 39) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                        \
  if (kOp == 40) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));                                             \
  if (kOp == 41) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                       \
  if (kOp == 42) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                       \
  if (kOp == 43) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(k) : "vcc");                \
  if (kOp == 44) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(k));                          \
  if (kOp == 45) asm volatile("v_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(u[i]));                                     \
  if (kOp == 46) asm volatile("v_cndmask_b32_e64 %0, 0, 1, s[2:3]" : "=v"(u[i]));                                  \
  if (kOp == 47) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(a[i].x), "v"(c.x), "v"(k) : "vcc"); \
  if (kOp == 48) asm volatile("v_cmp_lt_f32_e64 s[4:5], %1, %2\n v_cndmask_b32_e64 %0, %0, %3, s[4:5]" : "+v"(u[i]) : "v"(a[i].x), "v"(c.x), "v"(k) : "s4", "s5"); \
  if (kOp == 49) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(k));                              \
  if (kOp == 50) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) & 7]));                  \
  if (kOp == 51) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) & 7]));                  \
  if (kOp == 52) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); \
  if (kOp == 53) asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(k)); \
  if (kOp == 54) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[i ^ 1]));                          \
  if (kOp == 55) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(*reinterpret_cast<unsigned long long*>(&a[i])) : "v"(k & 7u)); \
  if (kOp == 56) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(u[i]) : "v"(k));                              \
  if (kOp == 57) asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i].x) : "v"(c.x), "v"(m.x));               \
  if (kOp == 58) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(u[i]) : "v"(k));                               \
  if (kOp == 59) asm volatile("v_add_lshl_u32 %0, %0, %1, 16" : "+v"(u[i]) : "v"(k));                             \
  if (kOp == 60) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(kh));                                  \
  if (kOp == 61) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i]) : "v"(kh));                                  \
  if (kOp == 62) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(u[i]) : "v"(kh));                                  \
  if (kOp == 63) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                   \
  if (kOp == 64) asm volatile("v_max3_u16 %0, %0, %1, %1" : "+v"(u[i]) : "v"(k));                                 \
  if (kOp == 65) asm volatile("v_max_u16 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                      \
  if (kOp == 66) asm volatile("v_max_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(u[i]) : "v"(k)); \
  if (kOp == 67) asm volatile("v_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(kh));                                     \
  if (kOp == 68) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(u[i]) : "v"(kh));                              \
  if (kOp == 69) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                   \
  if (kOp == 70) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                   \
  if (kOp == 71) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(u[i]) : "v"(k));                               \
  if (kOp == 72) asm volatile("v_add_u16 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                      \
  if (kOp == 73) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[i]) : "v"(k));                                      \
  if (kOp == 74) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(k));                                 \
  if (kOp == 75) asm volatile("v_pk_lshrrev_b16 %0, 1, %0" : "+v"(u[i]));                                          \
  if (kOp == 76) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %1" : "+v"(u[i]) : "v"(kh));
      REP8(OP)
#undef OP
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int kOp>
double run(const char* name, float* out)
{
  const int blocks = 256 * 8, iters = 2000;          // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<kOp>, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<kOp>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winst = double(blocks) * 4 * iters * 64;                 // wave-instructions
  const double per_simd_clk = winst / (ms * 1e-3) / (256.0 * 4) / 2.4e9;  // at 2.4 GHz
  printf("%-14s %8.3f ms  %7.1f G wave-inst/s  %.3f per SIMD and clock (2.4 GHz) = %.2f clocks each\n", name, ms, winst / ms * 1e-6, per_simd_clk, 1.0 / per_simd_clk);
  return ms;
}

int main(int argc, char** argv)
{
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  if (argc > 1 && argv[1][0] == 'p') {               // "p16": the packed 16-bit forms an add-compare-select could run on (round 5)
    run<5>("v_add_u32", out);
    run<6>("v_pk_add_u16", out);
    run<7>("v_pk_max_u16", out);
    run<69>("v_pk_min_u16", out);
    run<70>("v_pk_sub_u16", out);
    run<63>("v_pk_max_i16", out);
    run<60>("v_pk_max_f16", out);
    run<62>("v_pk_min_f16", out);
    run<61>("v_pk_add_f16", out);
    run<68>("v_pk_fma_f16", out);
    run<76>("v_pk_maximum3_f16", out);
    run<71>("v_pk_mad_u16", out);
    run<75>("v_pk_lshrrev_b16", out);
    run<64>("v_max3_u16", out);
    run<65>("v_max_u16 (VOP2)", out);
    run<66>("v_max_u16 sdwa hi", out);
    run<67>("v_max_f16 (VOP2)", out);
    run<72>("v_add_u16 (VOP2)", out);
    run<25>("v_max_u32", out);
    run<73>("v_max_i32", out);
    run<74>("v_max3_i32", out);
    run<38>("v_max_f32", out);
    run<8>("v_perm_b32", out);
    return 0;
  }
  if (argc > 1) {                                    // "lanes": only the cross-lane moves (round 3)
    run<26>("v_mov_b32", out);
    run<50>("permlane32_swap", out);
    run<51>("permlane16_swap", out);
    run<54>("permlane32_swap pair", out);
    run<52>("mov_dpp row_ror:8", out);
    run<53>("cndmask_dpp ror:8", out);
    run<55>("v_lshrrev_b64", out);
    run<56>("v_dot4_u32_u8", out);
    run<57>("v_min3_f32 abs", out);
    run<58>("v_lshl_or_b32", out);
    run<59>("v_add_lshl_u32", out);
    return 0;
  }
  run<0>("v_fma_f32", out);
  run<1>("v_pk_fma_f32", out);
  run<2>("v_pk_add_f32", out);
  run<3>("v_pk_mul_f32", out);
  run<4>("v_add_f32", out);
  run<5>("v_add_u32", out);
  run<6>("v_pk_add_u16", out);
  run<7>("v_pk_max_u16", out);
  run<8>("v_perm_b32", out);
  run<9>("v_cndmask_b32", out);
  run<10>("v_min_f32 abs", out);
  run<11>("v_cvt_f32_i32", out);
  run<12>("v_sqrt_f32", out);
  run<13>("v_mul_f64", out);
  run<14>("v_fma_f64", out);
  run<15>("v_add_f32_dpp", out);
  run<16>("cndmask e64 sgpr", out);
  run<17>("v_cmp vcc", out);
  run<18>("v_cmp e64 sgpr", out);
  run<19>("v_and_b32", out);
  run<20>("v_lshrrev_b32", out);
  run<21>("v_or3_b32", out);
  run<22>("v_bfe_i32", out);
  run<23>("cvt_f32_i32 sdwa", out);
  run<24>("v_mul_f32", out);
  run<25>("v_max_u32", out);
  run<26>("v_mov_b32", out);
  run<27>("v_add3_u32", out);
  run<28>("v_fmac_f32", out);
  run<29>("v_sub_f32", out);
  run<30>("cndmask vcc again", out);
  run<31>("v_xor_b32", out);
  run<32>("v_add_f32 abs", out);
  run<33>("pk_fma opsel", out);
  run<34>("v_lshl_add_u32", out);
  run<35>("v_bfi_b32", out);
  run<36>("cndmask vcc set", out);
  run<37>("cndmask vcc nodep", out);
  run<38>("v_max_f32", out);
  run<39>("v_or_b32", out);
  run<40>("v_lshlrev_b32", out);
  run<41>("v_sub_u32", out);
  run<42>("v_min_u32", out);
  run<43>("v_addc_co_u32", out);
  run<44>("cndmask e64 vcc", out);
  run<45>("cndmask e64 0,1,vcc", out);
  run<46>("cndmask e64 0,1,sgpr", out);
  run<47>("cmp+cndmask vcc (x2)", out);
  run<48>("cmp+cndmask sgpr (x2)", out);
  run<49>("cndmask vcc=exec", out);
  return 0;
}
