// k_decode.hip — K3/K4: channel decoding of the FIC and of every MSC sub-channel, and K5:
// ETI frame completion.
//
//  regroup_kernel / fic_group_kernel  logical CIF rows (demap_kernel folded misc.c:29-39 into its scatter)
//                  resp. FIC blocks -> natural bit order, 64 records interleaved word by word.
//  viterbi_fused_kernel  FIC and MSC decoder with the de-puncturing (depuncture.c:45-132) fused into its load.
//  viterbi_kernel  the same decoder fed with explicit per-step symbols: the S1 seam (dabhip_viterbi_batch).
//  viterbi_kernel  K=7 rate-1/4 maximum-likelihood decoder with the decisions of the
//                  reference's scalar viterbi() (viterbi.c:352-451): one LANE per code word,
//                  all 64 path metrics of that code word live in the lane's VGPRs (32 packed
//                  int16 pairs), so the add-compare-select is packed VALU work with no
//                  cross-lane traffic; 64 code words of equal length per wave.  Followed in the same kernel by the chain-back,
//                  energy-dispersal descrambling (misc.c:41-58) and MSB-first byte packing.
//  eti_finish_kernel  header + FIBs + EOF CRC + trailer of each ETI frame (misc.c:218-296).
//
// Metric equivalence (bit-exactness argument).  The reference adds, per transmitted symbol,
// +3 if the hypothesis agrees with the hard bit and -7 if not, and 0 for an erasure
// (mettab from gen_met(amp=1,noise=1,bias=0,scale=4), viterbi.c:455-462).  That is
// 10*agree - 7 per transmitted symbol; the number of transmitted symbols up to a step is
// the same for every path, so every compare "m1 > m0" at a step equals the compare of the
// agreement counts.  The kernel therefore accumulates agreement counts (0..4 per step);
// decisions, ties (strict '>' keeps the low predecessor) and the start condition (state 0
// reachable, others far below) are identical.
//
// Survivor records ("tagged max").  Metrics are held x16; within a block of four trellis steps the candidate of
// the LOW predecessor carries the tag 1 << k at step k (k = 0..3).  Both candidates being otherwise multiples
// of 16 plus the tags of earlier steps (< 1 << k), "high + garbage_h > low + garbage_l + tag" holds exactly
// when the high predecessor is strictly better, so one packed max both selects the survivor and leaves, in the
// low nibble of every state, the four decisions ALONG THAT STATE'S SURVIVOR PATH.  That nibble is all the
// chain-back needs to step four trellis steps back, so it is collected once per block (8 words per lane per 4
// steps, the same volume as one bit per state and step) and the nibbles are cleared with the re-pairing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <utility>

#include "dab_tables.hpp"
#include "device_types.hpp"
#include "kernels.hpp"

namespace dabhip {
namespace {

// ---------------------------------------------------------------------------------------
// branch code words: bit j of code(i) = parity(i & poly_j), i = 7-bit register with the
// low predecessor's oldest bit 0 (viterbi.c:373-381).  poly 0 == poly 3, so bit 3 == bit 0
// and only 8 distinct code words exist; code(i ^ 1) = code(i) ^ 7 on the three distinct bits.
__host__ __device__ constexpr unsigned parity_u(unsigned x)
{
  x ^= x >> 4;
  x ^= x >> 2;
  x ^= x >> 1;
  return x & 1u;
}
__host__ __device__ constexpr unsigned branch_code3(unsigned i)   // bits 0..2 of the code word
{
  return parity_u(i & 0x6d) | (parity_u(i & 0x4f) << 1) | (parity_u(i & 0x53) << 2);
}

// Path metrics are agreement counts held as packed int16 pairs: 32 VGPRs for the 64 states.
// In layout L(tau) (tau = 0..3, delta = 1 << tau) register r (0..15) holds states (k, k ^ delta) for the
// r-th k < 32 whose bit tau is clear, register 16 + r the same pair + 32.  With both members of a
// pair in one register, the butterflies of k and k ^ delta run as ONE packed add/max each:
//   E = max(X + B[c], Y + B[c^7])   -> new states (2k, 2(k^delta))      (pair difference 2 delta)
//   O = max(X + B[c^7], Y + B[c])   -> new states (2k+1, 2(k^delta)+1)
// so a step in L(tau) leaves its results in L(tau+1) with no data movement; after tau = 3 the
// registers are re-paired to L(0) with one v_perm_b32 each.  Four steps = one 32-bit word of
// trellis input.
typedef short __attribute__((ext_vector_type(2))) pk16;
typedef unsigned __attribute__((ext_vector_type(4))) vuint4;
// Survivor records are written once and read once, megabytes apart in time: nontemporal both ways (Viterbi stage 5.5 -> 5.25 ms)
#ifndef DABHIP_VIT_NOSTORE      // measurement builds only (tools/build_variant_decode.sh): the forward pass without its record traffic
#define DABHIP_VIT_NOSTORE 0
#endif
#ifndef DABHIP_VIT_WAVES        // waves per SIMD the fused decoder is compiled for (register budget 512 / waves)
#define DABHIP_VIT_WAVES 4
#endif
__device__ __forceinline__ void rec_store(uint4* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
  __builtin_nontemporal_store(vuint4{a, b, c, d}, reinterpret_cast<vuint4*>(p));
}
__device__ __forceinline__ uint4 rec_load(const uint4* p) { return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const vuint4*>(p))); }
typedef unsigned short __attribute__((ext_vector_type(2))) upk16;

__device__ __forceinline__ pk16 as_pk(uint32_t x) { return __builtin_bit_cast(pk16, x); }
__device__ __forceinline__ upk16 as_upk(uint32_t x) { return __builtin_bit_cast(upk16, x); }
constexpr int kMetricShift = 4;                           // metrics x16: four tag bits below every metric
__device__ __forceinline__ uint32_t as_u32(pk16 x) { return __builtin_bit_cast(uint32_t, x); }

__host__ __device__ constexpr int expand_bit(int r, int tau) { return ((r >> tau) << (tau + 1)) | (r & ((1 << tau) - 1)); }
__host__ __device__ constexpr int compress_bit(int k, int tau) { return ((k >> (tau + 1)) << tau) | (k & ((1 << tau) - 1)); }

template <int kTau, int kR>
__device__ __forceinline__ void butterfly_pair(const pk16 (&p)[32], pk16 (&n)[32], const pk16 (&bl)[8], const pk16 (&bh)[8])
{
  constexpr int j = expand_bit(kR, kTau);                 // butterfly index of the low half; high half is j ^ delta
  constexpr unsigned c = branch_code3(2 * j);
  const uint32_t x = as_u32(p[kR]), y = as_u32(p[16 + kR]);
  // Metrics are non-negative and < 2^16, so the two halves of a register can be advanced by ONE 32-bit add
  // without a carry crossing over (v_add_u32 issues at twice the rate of the packed 16-bit forms on gfx950).
  // bl = branch metric + tag (low predecessor), bh = branch metric (high predecessor).
  const upk16 t0 = as_upk(x + as_u32(bl[c])), t1 = as_upk(y + as_u32(bh[c ^ 7]));      // into states 2j, 2j'      (viterbi.c:404-414)
  const upk16 t2 = as_upk(x + as_u32(bl[c ^ 7])), t3 = as_upk(y + as_u32(bh[c]));      // into states 2j+1, 2j'+1  (viterbi.c:415-421)
  // the high predecessor wins iff it is strictly better (viterbi.c:411: "if (m1 > m0)"): the tag decides ties
  const pk16 e = as_pk(__builtin_bit_cast(uint32_t, __builtin_elementwise_max(t0, t1)));
  const pk16 o = as_pk(__builtin_bit_cast(uint32_t, __builtin_elementwise_max(t2, t3)));
  // results already form the pairs of the next layout
  constexpr int k_e = 2 * j, k_o = 2 * j + 1;
  if (kTau < 3) {
    n[(k_e >= 32 ? 16 : 0) + compress_bit(k_e & 31, kTau + 1)] = e;
    n[(k_o >= 32 ? 16 : 0) + compress_bit(k_o & 31, kTau + 1)] = o;
  } else {                                                // delta = 8 -> pairs (k, k ^ 16): park as L(4)
    n[(k_e >= 32 ? 16 : 0) + (k_e & 15)] = e;
    n[(k_o >= 32 ? 16 : 0) + (k_o & 15)] = o;
  }
}

template <int kTau, int... kRs>
__device__ __forceinline__ void all_pairs(const pk16 (&p)[32], pk16 (&n)[32], const pk16 (&bl)[8], const pk16 (&bh)[8],
                                          std::integer_sequence<int, kRs...>)
{
  (butterfly_pair<kTau, kRs>(p, n, bl, bh), ...);
}

// branch metrics of the 8 distinct code words (bit3 = bit0) for one trellis step, already x16 (kMetricShift)
// hard decisions: agreement count with the received nibble v under the "transmitted" mask m (sb = v | m << 4)
__device__ __forceinline__ void branch_metrics_hard(unsigned sb, int (&bm)[8])
{
  const unsigned v = sb & 15u, m = (sb >> 4) & 15u;
  const int ntx = __popc(m) << kMetricShift;
#pragma unroll
  for (unsigned c = 0; c < 4; ++c) {
    const unsigned cw = c | ((c & 1u) << 3);
    bm[c] = __popc(~(v ^ cw) & m) << kMetricShift;
    bm[c ^ 7] = ntx - bm[c];
  }
}
// (soft decisions: acs_step_soft below builds its packed metric words directly.  Four signed 4-bit values s_j per step, > 0: bit 0
// more likely, 0: punctured; metric = 28 + sum_j (c_j ? -s_j : +s_j), so that it stays non-negative)
template <int kTau>
__device__ __forceinline__ void acs_step_bm(const int (&bm)[8], const pk16 (&p)[32], pk16 (&n)[32])
{
  constexpr unsigned gamma = branch_code3(2u << kTau);    // code difference between the two members of a pair
  constexpr uint32_t tag = 0x00010001u << kTau;
  pk16 bl[8], bh[8];
#pragma unroll
  for (unsigned c = 0; c < 8; ++c) {
    const uint32_t b = static_cast<uint32_t>(bm[c]) | (static_cast<uint32_t>(bm[c ^ gamma]) << 16);
    bh[c] = as_pk(b);
    bl[c] = as_pk(b + tag);
  }
  all_pairs<kTau>(p, n, bl, bh, std::make_integer_sequence<int, 16>{});
}

template <int kTau>
__device__ __forceinline__ void acs_step(unsigned sb, const pk16 (&p)[32], pk16 (&n)[32])
{
  int bm[8];
  branch_metrics_hard(sb, bm);
  acs_step_bm<kTau>(bm, p, n);
}

// Soft decisions: the packed branch-metric words of a step straight from its four 4-bit values.  u_j = s_j + 8 (0..15) as four
// bytes; the metric of code word c (0..3), (28 + sum_j sigma_cj s_j) << 4, is ONE dot product with the sign pattern scaled by 16
// (bit 0 of c flips s0 and s3, bit 1 flips s1; the offsets fold the +8s in); the complementary code words are (56 << 4) minus that,
// also in packed form (both halves stay non-negative, so a 32-bit subtract serves both).  29 instructions instead of 43.
template <int kTau>
__device__ __forceinline__ void acs_step_soft(unsigned nib16, const pk16 (&p)[32], pk16 (&n)[32])
{
  constexpr unsigned gamma = branch_code3(2u << kTau);    // code difference between the two members of a pair
  constexpr uint32_t tag = 0x00010001u << kTau;
  constexpr uint32_t kAll = static_cast<uint32_t>(56 << kMetricShift) * 0x00010001u;
  const uint32_t a = nib16 & 0x0f0fu, b = (nib16 >> 4) & 0x0f0fu;
  const int u4 = static_cast<int>(__builtin_amdgcn_perm(b, a, 0x05010400u) ^ 0x08080808u);     // bytes (s0, s1, s2, s3) + 8
  constexpr int kSign[4] = {0x10101010, static_cast<int>(0xf01010f0u), 0x1010f010, static_cast<int>(0xf010f0f0u)};   // 16 sigma_c, byte j = value j
  constexpr int kOff[4] = {(28 - 32) * 16, (28 - 0) * 16, (28 - 16) * 16, (28 + 16) * 16};                             // 16 (28 - 8 sum_j sigma_cj)
  uint32_t lo[4], inv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    lo[c] = static_cast<uint32_t>(__builtin_amdgcn_sdot4(u4, kSign[c], kOff[c], false));
    inv[c] = static_cast<uint32_t>(56 << kMetricShift) - lo[c];
  }
  pk16 bl[8], bh[8];
#pragma unroll
  for (unsigned c = 0; c < 4; ++c) {
    const unsigned cp = c ^ gamma;
    const uint32_t hi = cp < 4 ? lo[cp] : inv[cp ^ 7];
    const uint32_t w = lo[c] | (hi << 16), wc = kAll - w;
    bh[c] = as_pk(w);
    bl[c] = as_pk(w + tag);
    bh[c ^ 7] = as_pk(wc);
    bl[c ^ 7] = as_pk(wc + tag);
  }
  all_pairs<kTau>(p, n, bl, bh, std::make_integer_sequence<int, 16>{});
}

// Soft decisions, branch metrics from two LDS tables (viterbi_fused_kernel<4>; DABHIP_SOFT_LUT=0 keeps the dot-product form above).  The
// metric of code c, (28 + sg0 (s0 + s3) + sg1 s1 + s2) << 4 with sg0 = -1 for odd c, sg1 = -1 for c & 2, is the sum of a part of (s0, s3)
// and a part of (s1, s2), each made non-negative by its share of the 28: A_c = (14 + sg0 (s0 + s3)) << 4, B_c = (14 + sg1 s1 + s2) << 4;
// the complementary code's metric (56 << 4) - (A_c + B_c) splits the same way, (28 << 4) - A_c and (28 << 4) - B_c.  A table row holds the
// four packed words of codes 0..3 for one layout tau (low half: code c, high half: code c ^ gamma(tau)), indexed by the two raw nibbles:
// one 16-byte read per table and step, then 4 adds, instead of a byte permute, four dot products and a dozen shifts, ors and subtracts.
#ifndef DABHIP_SOFT_LUT
#define DABHIP_SOFT_LUT 1
#endif
struct SoftLut {
  uint4 a[4][256];       // [tau][s0 | s3 << 4]
  uint4 b[4][256];       // [tau][s1 | s2 << 4]
};
__device__ __forceinline__ void build_soft_lut(SoftLut* lut)
{
  for (int e = threadIdx.x; e < 2 * 4 * 256; e += blockDim.x) {
    const int which = e >> 10, tau = (e >> 8) & 3, idx = e & 255;
    // the two signed 4-bit values of this row (the demapper clamps to +-7; a -8 that could only come from corrupted input counts as -7, so that every share stays >= 0)
    const int v0 = max(((idx & 15) ^ 8) - 8, -7), v1 = max(((idx >> 4) ^ 8) - 8, -7);
    const unsigned gamma = tau == 0 ? branch_code3(2u) : tau == 1 ? branch_code3(4u) : tau == 2 ? branch_code3(8u) : branch_code3(16u);
    uint32_t w[4];
#pragma unroll
    for (unsigned c = 0; c < 4; ++c) {
      auto part = [&](unsigned code) -> uint32_t {          // this table's share of the metric of `code` (0..7)
        const unsigned q = code < 4 ? code : code ^ 7;      // complementary codes: the share's mirror image
        const int sg0 = (q & 1) ? -1 : 1, sg1 = (q & 2) ? -1 : 1;
        const int share = which == 0 ? 14 + sg0 * (v0 + v1) : 14 + sg1 * v0 + v1;     // table a: (s0, s3); table b: (s1, s2)
        return static_cast<uint32_t>((code < 4 ? share : 28 - share) << kMetricShift);
      };
      w[c] = part(c) | (part(c ^ gamma) << 16);
    }
    (which == 0 ? lut->a[tau][idx] : lut->b[tau][idx]) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __syncthreads();
}
template <int kTau>
__device__ __forceinline__ void acs_step_soft_lut(unsigned nib16, const SoftLut* lut, const pk16 (&p)[32], pk16 (&n)[32])
{
  constexpr uint32_t tag = 0x00010001u << kTau;
  constexpr uint32_t kAll = static_cast<uint32_t>(56 << kMetricShift) * 0x00010001u;
  const uint4 ta = lut->a[kTau][(nib16 & 15u) | ((nib16 >> 8) & 0xf0u)], tb = lut->b[kTau][(nib16 >> 4) & 0xffu];
  const uint32_t w[4] = {ta.x + tb.x, ta.y + tb.y, ta.z + tb.z, ta.w + tb.w};
  pk16 bl[8], bh[8];
#pragma unroll
  for (unsigned c = 0; c < 4; ++c) {
    const uint32_t wc = kAll - w[c];
    bh[c] = as_pk(w[c]);
    bl[c] = as_pk(w[c] + tag);
    bh[c ^ 7] = as_pk(wc);
    bl[c ^ 7] = as_pk(wc + tag);
  }
  all_pairs<kTau>(p, n, bl, bh, std::make_integer_sequence<int, 16>{});
}
template <int kTau>
__device__ __forceinline__ void acs_step_soft_any(unsigned nib16, const SoftLut* lut, const pk16 (&p)[32], pk16 (&n)[32])
{
  if (DABHIP_SOFT_LUT) acs_step_soft_lut<kTau>(nib16, lut, p, n);
  else acs_step_soft<kTau>(nib16, p, n);
}

// Hard decisions with the de-puncturing fused in (viterbi_fused_kernel<1>): a step receives the first n = 0..4 bits of its group of
// four (wave-uniform n), so only 16 + 8 + 4 + 2 + 1 = 31 (mask, value) pairs occur.  Their eight packed branch-metric words
// per step type sit in an LDS table (8 tag bits x 4 parts x 32 rows x 16 B = 16 KB, shared by the 4 waves of a workgroup); a step
// fetches its row with four 16-byte reads instead of ~34 VALU instructions.  Rows of one n are 16 bytes apart = disjoint
// banks, equal rows broadcast: the reads are conflict-free.
__device__ __forceinline__ unsigned lut_row_base(int n) { return n == 4 ? 0u : (0x10181c1eu >> (8 * n)) & 0xffu; }   // n = 4, 3, 2, 1, 0 -> row 0, 16, 24, 28, 30

// lut[tag bit 0..7][part][row]: parts 0, 1 = the words of the LOW predecessor's candidates (metric + tag, codes 0..3 and 4..7),
// parts 2, 3 = those of the high predecessor's (metric only); x256 (byte tags, see acs8_lut); layout tau = tag bit & 3
typedef uint4 MetricLut[4][32];
__device__ __forceinline__ void build_metric_lut(MetricLut* lut)
{
  for (int e = threadIdx.x; e < 8 * 4 * 32; e += blockDim.x) {
    const int tagbit = e >> 7, part = (e >> 5) & 3, row = e & 31, tau = tagbit & 3, half = part & 1;
    const int n = row < 16 ? 4 : row < 24 ? 3 : row < 28 ? 2 : row < 30 ? 1 : 0;
    const unsigned v = row < 30 ? static_cast<unsigned>(row) - lut_row_base(n) : 0u, m = (1u << n) - 1u;
    int bm[8];
    branch_metrics_hard(v | (m << 4), bm);
    const unsigned gamma = tau == 0 ? branch_code3(2u) : tau == 1 ? branch_code3(4u) : tau == 2 ? branch_code3(8u) : branch_code3(16u);
    const uint32_t tag = part < 2 ? (0x00010001u << tagbit) : 0u;
    uint32_t w[4];
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
      const unsigned c = 4u * half + k;
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (unsigned q = 0; q < 8; ++q) {                // bm[] is indexed with run-time values: select, do not index
        lo = (q == c) ? static_cast<uint32_t>(bm[q]) : lo;
        hi = (q == (c ^ gamma)) ? static_cast<uint32_t>(bm[q]) : hi;
      }
      w[k] = ((lo | (hi << 16)) << (8 - kMetricShift)) + tag;
    }
    lut[tagbit][part][row] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __syncthreads();
}

template <int kTau, int kTagBit>
__device__ __forceinline__ void acs_step_lut(unsigned row, const MetricLut* lut, const pk16 (&p)[32], pk16 (&n)[32])
{
  static_assert((kTagBit & 3) == kTau, "the tag bit selects the table of its layout");
  const uint4 l0 = lut[kTagBit][0][row], l1 = lut[kTagBit][1][row], h0 = lut[kTagBit][2][row], h1 = lut[kTagBit][3][row];
  const pk16 bl[8] = {as_pk(l0.x), as_pk(l0.y), as_pk(l0.z), as_pk(l0.w), as_pk(l1.x), as_pk(l1.y), as_pk(l1.z), as_pk(l1.w)};
  const pk16 bh[8] = {as_pk(h0.x), as_pk(h0.y), as_pk(h0.z), as_pk(h0.w), as_pk(h1.x), as_pk(h1.y), as_pk(h1.z), as_pk(h1.w)};
  all_pairs<kTau>(p, n, bl, bh, std::make_integer_sequence<int, 16>{});
}

// L(4) (pairs (k, k^16)) -> L(0) (pairs (k, k^1)): one byte permute per register; the tag nibbles are cleared here
__device__ __forceinline__ void repair_layout(const pk16 (&n)[32], pk16 (&p)[32])
{
#pragma unroll
  for (int side = 0; side < 2; ++side)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int a = side * 16 + ((2 * q) & 15);           // register holding state 2q (low half if 2q < 16)
      const uint32_t sel = (2 * q < 16) ? 0x05040100u : 0x07060302u;
      p[side * 16 + q] = as_pk(__builtin_amdgcn_perm(as_u32(n[a + 1]), as_u32(n[a]), sel) & 0xfff0fff0u);
    }
}

// Survivor record of a block: the tag nibbles of the 64 states, 8 words per lane.  Register R (in the L(4) numbering:
// low half = state 32 (R >> 4) + (R & 15), high half = that + 16) lands in word R >> 2, nibble
// 4 (R & 1) + 2 half + ((R >> 1) & 1).  After a partial block (layouts L(1..3)) only state 0 is read back, and state 0
// is the low half of register 0 in every layout.
__device__ __forceinline__ void survivor_record(const pk16 (&n)[32], uint4* rec)
{
  uint32_t d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t pa = __builtin_amdgcn_perm(as_u32(n[4 * i + 1]), as_u32(n[4 * i]), 0x06040200u);       // low bytes of 4 halves
    const uint32_t pb = __builtin_amdgcn_perm(as_u32(n[4 * i + 3]), as_u32(n[4 * i + 2]), 0x06040200u);
    d[i] = (pa & 0x0f0f0f0fu) | ((pb << 4) & 0xf0f0f0f0u);                                               // one v_bfi_b32
  }
  rec_store(rec, d[0], d[1], d[2], d[3]);
  rec_store(rec + 64, d[4], d[5], d[6], d[7]);
}
__device__ __forceinline__ uint32_t in_vgpr(uint32_t x)
{
  asm volatile("" : "+v"(x));   // keeps the 8-way select below a v_cndmask tree (the optimiser would index a scratch copy instead)
  return x;
}
// ---------------------------------------------------------------------------------------
// regroup: logical CIF rows (16 planes of 108 x kBits words each, one row per ETI frame) ->
// de-interleaved natural bit order, 64 frames interleaved word by word:
//   grouped[(tile * 1728 + w) * 64 + lane] = bits 32 w .. 32 w + 31 of cif_time_deinterleaved
// of frame job_ids[64 tile + lane].  The Viterbi kernel (lane = frame) then reads its received
// bits with fully coalesced 256-byte loads.  Thread = (frame lane, 4 consecutive plane words): 16 loads of
// 16 bytes (one per plane) in, 64 words out; a workgroup consumes whole 64-byte lines of every plane.
// One step of a bit-matrix transposition over 16 words: index bit kWordBit of the word number changes places with index bit kPosBit of the bit
// position (the masked swap of the classic 32 x 32 transpose).
template <int kWordBit, int kPosBit>
__device__ __forceinline__ void swap_index_bits(uint32_t (&w)[16])
{
  constexpr uint32_t s = 1u << kPosBit;
  constexpr uint32_t m = kPosBit == 0 ? 0x55555555u : kPosBit == 1 ? 0x33333333u : kPosBit == 2 ? 0x0f0f0f0fu : kPosBit == 3 ? 0x00ff00ffu : 0x0000ffffu;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i & (1 << kWordBit)) continue;
    uint32_t& a = w[i];
    uint32_t& b = w[i | (1 << kWordBit)];
    const uint32_t t = ((a >> s) ^ b) & m;
    b ^= t;
    a ^= t << s;
  }
}
// 16 plane words (plane p, bits b = 2 o + h) -> 16 output words (word o, bit 16 h + p): as an exchange of index bits, word number
// (p3 p2 p1 p0) and bit position (o3 o2 o1 o0 h) become word number (o3 o2 o1 o0) and bit position (h p3 p2 p1 p0).  Eight masked-swap
// steps walk h up the bit position while every p_i drops into its place and every o_i into the word number: 24 operations per output word
// instead of the 80 of a bit-by-bit gather.  (The kernel's time did not move, 0.32 ms: it was never bound by these instructions but by its
// 16-byte reads from 64 different rows per wave-instruction -- 0.7 GB through HBM at about two thirds of the copy rate.)
__device__ __forceinline__ void planes_to_words(uint32_t (&w)[16])
{
  swap_index_bits<0, 0>(w);   // word bit 0: p0 <-> h
  swap_index_bits<0, 1>(w);   //             h  <-> o0
  swap_index_bits<1, 1>(w);   // word bit 1: p1 <-> h
  swap_index_bits<1, 2>(w);   //             h  <-> o1
  swap_index_bits<2, 2>(w);
  swap_index_bits<2, 3>(w);
  swap_index_bits<3, 3>(w);
  swap_index_bits<3, 4>(w);
}

template <int kBits>
__global__ __launch_bounds__(256) void regroup_kernel(const int* __restrict__ job_ids, const DecodeJob* __restrict__ jobs,
                                                      const int* __restrict__ stream_cif_base, const uint32_t* __restrict__ rows,
                                                      uint32_t* __restrict__ grouped, int ntiles)
{
  constexpr int kRowWords = 1728 * kBits, kBlocks = 108 * kBits;   // blocks of 16 words (one word of each plane)
  // XCD-aware order: workgroups go to the eight XCDs (each with its own L2) round robin by their linear index, and the kPerTile workgroups of a tile read
  // neighbouring 64-byte halves of the same 128-byte lines of the tile's 64 rows -- so a tile's workgroups take linear indices 8 apart and meet in ONE L2
  // (as a (x, tile) grid the halves of a line were fetched by two XCDs: 0.69 GB fetched for 0.35 GB of rows)
  constexpr int kPerTile = (27 * kBits + 3) / 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = (slot / kPerTile) * 8 + xcd, lane = threadIdx.x & 63;
  if (tile >= ntiles) return;
  const int wb4 = (slot % kPerTile) * 4 + (threadIdx.x >> 6);    // group of 4 consecutive plane words
  if (wb4 * 4 >= kBlocks) return;
  const int jid = job_ids[tile * 64 + lane];
  if (jid < 0) return;
  const DecodeJob job = jobs[jid];
  // rows are plane-major: plane p (the bits i with i & 15 == p) occupies words [p * kBlocks, (p + 1) * kBlocks)
  const uint32_t* row = rows + (static_cast<size_t>(stream_cif_base[job.stream]) + job.cif) * kRowWords + wb4 * 4;
  uint4 v[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) v[p] = *reinterpret_cast<const uint4*>(row + p * kBlocks);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t pw[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) pw[p] = j == 0 ? v[p].x : j == 1 ? v[p].y : j == 2 ? v[p].z : v[p].w;
    uint32_t* dst = grouped + (static_cast<size_t>(tile) * kRowWords + (wb4 * 4 + j) * 16) * 64 + lane;
    if (kBits == 1) {          // output word o: bit k <- plane k & 15, bit 2 o + (k >> 4) of that plane's word
      planes_to_words(pw);
#pragma unroll
      for (int o = 0; o < 16; ++o) dst[static_cast<size_t>(o) * 64] = pw[o];
      continue;
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      uint32_t w = 0;
      {                        // 4-bit soft values: output word o = values of planes 8 (o & 1) .. + 7 at index o >> 1 of their words
#pragma unroll
        for (int k = 0; k < 8; ++k) w |= ((pw[8 * (o & 1) + k] >> (4 * (o >> 1))) & 15u) << (4 * k);
      }
      dst[static_cast<size_t>(o) * 64] = w;
    }
  }
}

// FIC blocks are already in natural order: only interleave 64 of them word by word for the fused decoder
__global__ __launch_bounds__(256) void fic_group_kernel(const uint32_t* __restrict__ fic_rows, int first_block, int nblocks, int block_words,
                                                        uint32_t* __restrict__ grouped)
{
  const int tile = blockIdx.y, lane = threadIdx.x & 63;
  const int b = tile * 64 + lane;
  if (b >= nblocks) return;
  for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < block_words; w += gridDim.x * 4)
    grouped[(static_cast<size_t>(tile) * block_words + w) * 64 + lane] = fic_rows[static_cast<size_t>(first_block + b) * block_words + w];
}

// ---------------------------------------------------------------------------------------
// shared by both Viterbi kernels: 4 trellis steps = one pass through the register layouts = one survivor record
__device__ __forceinline__ void acs4(uint32_t ww, pk16 (&pm)[32], pk16 (&pn)[32], pk16 (&pl4)[32], uint4* rec)
{
  acs_step<0>(ww & 0xff, pm, pn);
  acs_step<1>((ww >> 8) & 0xff, pn, pm);
  acs_step<2>((ww >> 16) & 0xff, pm, pn);
  acs_step<3>(ww >> 24, pn, pl4);
  survivor_record(pl4, rec);
  repair_layout(pl4, pm);
}

// x0 .. x3: the received 4-bit values of four steps, each step's in the low 16 bits of its own word (values not received: zero nibbles)
__device__ __forceinline__ void acs4_soft(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, const SoftLut* lut, pk16 (&pm)[32], pk16 (&pn)[32], pk16 (&pl4)[32], uint4* rec)
{
  acs_step_soft_any<0>(x0, lut, pm, pn);
  acs_step_soft_any<1>(x1, lut, pn, pm);
  acs_step_soft_any<2>(x2, lut, pm, pn);
  acs_step_soft_any<3>(x3, lut, pn, pl4);
  survivor_record(pl4, rec);
  repair_layout(pl4, pm);
}

// the last r = 1..3 steps of a code word whose length is not a multiple of four: no re-pairing, the record is
// only read for state 0
__device__ __forceinline__ void acs_tail(uint32_t ww, int r, pk16 (&pm)[32], pk16 (&pn)[32], uint4* rec)
{
  acs_step<0>(ww & 0xff, pm, pn);
  if (r == 1) { survivor_record(pn, rec); return; }
  acs_step<1>((ww >> 8) & 0xff, pn, pm);
  if (r == 2) { survivor_record(pm, rec); return; }
  acs_step<2>((ww >> 16) & 0xff, pm, pn);
  survivor_record(pn, rec);
}
__device__ __forceinline__ void acs_tail_soft(uint32_t x0, uint32_t x1, uint32_t x2, int r, const SoftLut* lut, pk16 (&pm)[32], pk16 (&pn)[32], uint4* rec)
{
  acs_step_soft_any<0>(x0, lut, pm, pn);
  if (r == 1) { survivor_record(pn, rec); return; }
  acs_step_soft_any<1>(x1, lut, pn, pm);
  if (r == 2) { survivor_record(pm, rec); return; }
  acs_step_soft_any<2>(x2, lut, pm, pn);
  survivor_record(pn, rec);
}

// ---- hard decisions, byte tags (viterbi_fused_kernel<1>) --------------------------------------
// Metrics x256: the tag of step k = 0..7 of a block of EIGHT steps is 1 << k, so the low BYTE of every state collects the
// eight decisions along its survivor path.  Records are then plain byte gathers (16 v_perm per 8 steps, no masking or
// merging), and the re-pairing permute that follows a record clears the tags for free (selector 0x0c = constant zero).
template <bool kClear>
__device__ __forceinline__ void repair_layout8(const pk16 (&n)[32], pk16 (&p)[32])
{
#pragma unroll
  for (int side = 0; side < 2; ++side)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int a = side * 16 + ((2 * q) & 15);           // register holding state 2q (low half if 2q < 16)
      const uint32_t sel = (2 * q < 16) ? (kClear ? 0x050c010cu : 0x05040100u) : (kClear ? 0x070c030cu : 0x07060302u);
      p[side * 16 + q] = as_pk(__builtin_amdgcn_perm(as_u32(n[a + 1]), as_u32(n[a]), sel));
    }
}
// register R (L(4) numbering, see survivor_record) lands in word R >> 1, byte 2 (R & 1) + half
__device__ __forceinline__ void survivor_record8(const pk16 (&n)[32], uint4* rec)
{
  uint32_t d[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = __builtin_amdgcn_perm(as_u32(n[2 * i + 1]), as_u32(n[2 * i]), 0x06040200u);
#pragma unroll
  for (int j = 0; j < 4; ++j) rec_store(rec + 64 * j, d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3]);
}
// 8 trellis steps from 8 table rows (one byte each): two passes through the register layouts, one survivor record
__device__ __forceinline__ void acs8_lut(uint32_t rows0, uint32_t rows1, const MetricLut* lut, pk16 (&pm)[32], pk16 (&pn)[32],
                                         pk16 (&pl4)[32], uint4* rec)
{
  acs_step_lut<0, 0>(rows0 & 0xff, lut, pm, pn);
  acs_step_lut<1, 1>((rows0 >> 8) & 0xff, lut, pn, pm);
  acs_step_lut<2, 2>((rows0 >> 16) & 0xff, lut, pm, pn);
  acs_step_lut<3, 3>(rows0 >> 24, lut, pn, pl4);
  repair_layout8<false>(pl4, pm);
  acs_step_lut<0, 4>(rows1 & 0xff, lut, pm, pn);
  acs_step_lut<1, 5>((rows1 >> 8) & 0xff, lut, pn, pm);
  acs_step_lut<2, 6>((rows1 >> 16) & 0xff, lut, pm, pn);
  acs_step_lut<3, 7>(rows1 >> 24, lut, pn, pl4);
  survivor_record8(pl4, rec);
  repair_layout8<true>(pl4, pm);
}
// the last r = 1..7 steps of a code word: the record is only read for state 0 (low half of register 0 in every layout)
__device__ __forceinline__ void acs8_tail_lut(uint32_t rows0, uint32_t rows1, int r, const MetricLut* lut, pk16 (&pm)[32],
                                              pk16 (&pn)[32], pk16 (&pl4)[32], uint4* rec)
{
  if (r >= 4) {
    acs_step_lut<0, 0>(rows0 & 0xff, lut, pm, pn);
    acs_step_lut<1, 1>((rows0 >> 8) & 0xff, lut, pn, pm);
    acs_step_lut<2, 2>((rows0 >> 16) & 0xff, lut, pm, pn);
    acs_step_lut<3, 3>(rows0 >> 24, lut, pn, pl4);
    repair_layout8<false>(pl4, pm);
    if (r == 4) { survivor_record8(pm, rec); return; }
    acs_step_lut<0, 4>(rows1 & 0xff, lut, pm, pn);
    if (r == 5) { survivor_record8(pn, rec); return; }
    acs_step_lut<1, 5>((rows1 >> 8) & 0xff, lut, pn, pm);
    if (r == 6) { survivor_record8(pm, rec); return; }
    acs_step_lut<2, 6>((rows1 >> 16) & 0xff, lut, pm, pn);
    survivor_record8(pn, rec);
    return;
  }
  acs_step_lut<0, 0>(rows0 & 0xff, lut, pm, pn);
  if (r == 1) { survivor_record8(pn, rec); return; }
  acs_step_lut<1, 1>((rows0 >> 8) & 0xff, lut, pn, pm);
  if (r == 2) { survivor_record8(pm, rec); return; }
  acs_step_lut<2, 2>((rows0 >> 16) & 0xff, lut, pm, pn);
  survivor_record8(pn, rec);
}

// chain back over byte-tag records: block b of 8 steps at my_rec[256 b + 64 j], j = 0..3
__device__ __forceinline__ void chain_back8(const uint4* my_rec, int nsteps, const uint32_t* __restrict__ prbs_words, uint32_t* dst)
{
  unsigned state = 0;
  uint32_t acc = 0;
  auto consume = [&](unsigned tags, int t0, int k_hi) {    // steps t0 + k_hi .. t0, newest first
#pragma unroll
    for (int k = 7; k >= 0; --k) {
      const int t = t0 + k;
      if (k <= k_hi && t >= 6) {                           // steps 0..5 only flush the encoder's initial zeros
        const unsigned bit = ((tags >> k) & 1u) ^ 1u;      // tag set = low predecessor survived = decision 0
        state = (state | (bit << 6)) >> 1;
        const int i = t - 6;                               // data bit index
        acc |= bit << (8 * ((i >> 3) & 3) + (7 - (i & 7)));
        if ((i & 31) == 0) {
          dst[i >> 5] = acc ^ prbs_words[i >> 5];
          acc = 0;
        }
      }
    }
  };
  const int nfull = nsteps >> 3, r = nsteps & 7;
  if (r) consume(my_rec[static_cast<size_t>(nfull) * 256].x & 255u, 8 * nfull, r - 1);
  // Only the word of a record that holds the current state's byte is fetched (its 16-byte part = state bits 5 and 3): a chain of
  // dependent loads, one memory round trip per 8 steps, but 45 % of the records' sectors instead of all of them -- the stage is
  // bound by the record traffic, and a wave that waits here leaves the SIMD to the forward passes of the others (5.42 -> 5.06 ms;
  // fetching whole records 2 .. 8 blocks ahead, which hides the latency instead, made no difference at all).  Round 3, from per-wave time
  // stamps (profiles/r03_viterbi_tail.json): a 4614-step wave spends 1.07 ms here (577 dependent loads at HBM latency under load), a
  // quarter of all resident wave-time is chain-back -- yet with the records kept in L2 the whole launch gains only 0.3 ms, without any
  // chain-back 0.45.  A lane's record in ONE 64-byte line ([block][lane][part], read back whole, 2 or 4 blocks ahead) was 2.8 x slower
  // (14 ms): 16-byte stores at a 64-byte stride.  Also measured (round 3) and dropped: the launch as PERSISTENT waves that take groups off a counter and advance
  // the chain-back of the group before by one record per eight steps of the next group's forward pass (the load issued at the end of a block, consumed a block
  // later): 5.5 ms against 4.85.  The four words of chain state live across the forward pass's blocks, the kernel has no register to spare (12 spilled), and a
  // scratch reload inside the block waits -- loads and stores retire in order on one counter -- for the record stores issued just before it.
  // A whole block per step: its decisions d_k = !tag_k (k = 0 .. 7, step 8 b + k), bit-reversed: r8 = d_0 .. d_7 from the top.  Walking
  // back from step 8 b + 7 to 8 b leaves state (d_0 .. d_5) = r8 >> 2; data bit i = step - 6 goes MSB-first into byte i >> 3: d_6, d_7
  // are the top two bits of byte b, d_0 .. d_5 the low six bits of byte b - 1 -- the new state itself.
  for (int b = nfull - 1; b >= 0; --b) {
    const unsigned reg = ((state >> 5) << 4) | (state & 15u), half = (state >> 4) & 1u, i = reg >> 1, byte = 2u * (reg & 1u) + half;   // survivor_record8's layout
    const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(my_rec + static_cast<size_t>(b) * 256 + 64 * (i >> 2)) + (i & 3u));
    const unsigned tags = (w >> (8u * byte)) & 255u;
    if (b >= 1) {
      const unsigned r8 = __brev(~tags & 0xffu) >> 24;
      state = r8 >> 2;
      acc |= ((r8 & 3u) << 6) << (8 * (b & 3));
      if ((b & 3) == 0) {
        dst[b >> 2] = acc ^ prbs_words[b >> 2];
        acc = 0;
      }
      acc |= state << (8 * ((b - 1) & 3));
    } else {
      consume(tags, 0, 7);                                 // steps 0..5 only flush the encoder's initial zeros
    }
  }
}

// chain back from state 0 (viterbi.c:438-450), descramble (misc.c:41-58), pack MSB first.  my_rec: this lane's
// records, block b at my_rec[128 b] and my_rec[128 b + 64].
__device__ __forceinline__ void chain_back(const uint4* my_rec, int nsteps, const uint32_t* __restrict__ prbs_words, uint32_t* dst)
{
  unsigned state = 0;
  uint32_t acc = 0;
  auto consume = [&](unsigned nib, int t0, int k_hi) {     // steps t0 + k_hi .. t0, newest first
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      const int t = t0 + k;
      if (k <= k_hi && t >= 6) {                           // steps 0..5 only flush the encoder's initial zeros
        const unsigned bit = ((nib >> k) & 1u) ^ 1u;       // tag set = low predecessor survived = decision 0
        state = (state | (bit << 6)) >> 1;
        const int i = t - 6;                               // data bit index
        acc |= bit << (8 * ((i >> 3) & 3) + (7 - (i & 7)));
        if ((i & 31) == 0) {
          dst[i >> 5] = acc ^ prbs_words[i >> 5];
          acc = 0;
        }
      }
    }
  };
  const int nfull = nsteps >> 2, r = nsteps & 3;
  if (r) consume(my_rec[static_cast<size_t>(nfull) * 128].x & 15u, 4 * nfull, r - 1);
  // as in chain_back8: only the 16-byte half of a record that holds the state's nibble is fetched (half = word index >> 2).  (Round 3: two blocks per memory
  // round trip -- both halves of record b - 1 fetched with the selected half of record b -- measured the same, 6.55 against 6.51 .. 6.66 ms.)
  // Block 0 (steps 0..3) only flushes the encoder's initial zeros: nothing to read.
  for (int b = nfull - 1; b >= 1; --b) {
    const unsigned reg = ((state >> 5) << 4) | (state & 15u), half = (state >> 4) & 1u;
    const unsigned i = reg >> 2, nib = 4u * (reg & 1u) + 2u * half + ((reg >> 1) & 1u);     // survivor_record's layout
    const uint4 v = rec_load(my_rec + static_cast<size_t>(b) * 128 + 64 * (i >> 2));
    const uint32_t lo = (i & 1u) ? in_vgpr(v.y) : in_vgpr(v.x), hi = (i & 1u) ? in_vgpr(v.w) : in_vgpr(v.z);
    consume((((i & 2u) ? hi : lo) >> (4u * nib)) & 15u, 4 * b, 3);
  }
}

template <int kBits>
struct MetricScale {
  // state 0 is kept at kBase at every re-base; the other states stay within 6 steps' worth of branch metric of it
  // (24 agreements hard, 6 x 56 soft), x16, plus tags: kBase exceeds that, so nothing goes negative.  The same gap
  // is the start condition (viterbi.c:387-389: 0 vs -999999; it only has to exceed what six steps can collect).
  // key 8 = hard decisions with byte tags (metrics x256, viterbi_fused_kernel<1>): state 0 at 64 agreements, the others
  // within 24 of it, growth <= 4 per step: 64 + 24 + 32 x 4 = 216 < 256
  static constexpr uint32_t kBase = kBits == 1 ? 1024u : kBits == 8 ? 16384u : 8192u;
  // growth per step <= 16 x 4 (hard) / 16 x 56 (soft) / 256 x 4 (byte tags): re-base long before the unsigned 16-bit range ends
  static constexpr int kRebaseSteps = kBits == 1 ? 256 : 32;
};

template <int kBits>
__device__ __forceinline__ void init_metrics(pk16 (&pm)[32])
{
  // L(0): register r = states (2r, 2r+1), register 16 + r = states (32 + 2r, 33 + 2r)
#pragma unroll
  for (int r = 0; r < 32; ++r) pm[r] = as_pk(0u);
  pm[0] = as_pk(MetricScale<kBits>::kBase);
}

template <int kBits>
__device__ __forceinline__ void rebase_metrics(pk16 (&pm)[32])
{
  const uint32_t s0 = (as_u32(pm[0]) & 0xffffu) - MetricScale<kBits>::kBase;
  const uint32_t base = s0 | (s0 << 16);                  // every half is >= s0: no borrow crosses
#pragma unroll
  for (int r = 0; r < 32; ++r) pm[r] = as_pk(as_u32(pm[r]) - base);
}

// one wave (64 lanes) per group of <= 64 equal-length code words; trellis input = one byte per step (gather_kernel)
__global__ __launch_bounds__(64) void viterbi_kernel(const WaveGroup* __restrict__ groups, const int* __restrict__ job_ids,
                                                     const CodewordPlan* __restrict__ plans, const uint4* __restrict__ steps,
                                                     uint2* __restrict__ decisions, const uint32_t* __restrict__ prbs_words,
                                                     uint8_t* __restrict__ out, int record_stride)
{
  const int lane = threadIdx.x;
  const WaveGroup grp = groups[blockIdx.x];
  const int nsteps = grp.nsteps;
  const uint4* my_steps = steps + grp.step_base * 64 + lane;
  uint4* my_rec = reinterpret_cast<uint4*>(decisions + grp.dec_base * 64) + lane;

  pk16 pm[32], pn[32], pl4[32];
  init_metrics<1>(pm);
  const int n16 = (nsteps + 15) >> 4;
  uint4 pack = my_steps[0];
  for (int t16 = 0; t16 < n16; ++t16) {
    // fetch the next 16 steps before this block's record stores are issued: the wait for it then
    // does not have to drain those stores (loads and stores retire in order on one counter)
    const uint4 next = my_steps[static_cast<size_t>(min(t16 + 1, n16 - 1)) * 64];
    const uint32_t w[4] = {pack.x, pack.y, pack.z, pack.w};
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int t = 16 * t16 + 4 * q;
      if (t >= nsteps) break;
      const uint32_t ww = (q == 0) ? w[0] : (q == 1) ? w[1] : (q == 2) ? w[2] : w[3];
      uint4* rec = my_rec + static_cast<size_t>(t >> 2) * 128;
      if (t + 4 <= nsteps) acs4(ww, pm, pn, pl4, rec);
      else acs_tail(ww, nsteps - t, pm, pn, rec);
    }
    if ((t16 & 15) == 15) rebase_metrics<1>(pm);
    pack = next;
  }
  if (lane >= grp.count) return;
  const CodewordPlan pl = plans[grp.plan];
  const int record = job_ids ? job_ids[grp.first + lane] : grp.first + lane;
  chain_back(my_rec, nsteps, prbs_words, reinterpret_cast<uint32_t*>(out + static_cast<size_t>(record) * record_stride + pl.out_offset));
}

// Decoder with the de-puncturing fused into the load (depuncture.c:45-132): lane = output record (ETI frame or FIC
// block), received bits come from lane-interleaved rows (regroup_kernel / fic_group_kernel), 32 bits per coalesced load.
// Puncturing vectors keep the FIRST n bits of every group of four (n = 1..4), so a step's input is simply the next n
// received values of the stream; n is wave-uniform.  kBits = 1: hard bits (the reference's behaviour);
// kBits = 4: signed 4-bit soft values (extension).
#ifndef DABHIP_VIT_TIMES         // measurement build only (tools/vit_tail.py): start / end / chain-back time stamps of every wave
#define DABHIP_VIT_TIMES 0
#endif
#if DABHIP_VIT_TIMES
__device__ unsigned long long g_vit_times[4 * 32768];     // {start, forward pass done, end, nsteps << 8 | XCC} per wave-group of the LAST big launch
#endif

template <int kBits>
__global__ __launch_bounds__(256, DABHIP_VIT_WAVES) void viterbi_fused_kernel(const WaveGroup* __restrict__ groups, int ngroups, const int* __restrict__ job_ids,
                                                               const CodewordPlan* __restrict__ plans,
                                                               const uint32_t* __restrict__ grouped, int row_words,
                                                               uint2* __restrict__ decisions, const uint32_t* __restrict__ prbs_words,
                                                               uint8_t* __restrict__ out, int record_stride)
{
  // four independent waves per workgroup (one wave-group of code words each); they only share the branch-metric table
  // (two per workgroup -- finer dispatch granularity for the launch's tail -- measured the same, 5.01 .. 5.05 against 5.02 .. 5.07 ms; one: 5.8)
  constexpr size_t kLutBytes = kBits == 1 ? 8 * sizeof(MetricLut) : (DABHIP_SOFT_LUT ? sizeof(SoftLut) : 16);
  __shared__ __attribute__((aligned(16))) unsigned char lut_raw[kLutBytes];
  MetricLut* lut = reinterpret_cast<MetricLut*>(lut_raw);
  SoftLut* soft_lut = reinterpret_cast<SoftLut*>(lut_raw);
  if (kBits == 1) build_metric_lut(lut);
  else if (DABHIP_SOFT_LUT) build_soft_lut(soft_lut);
  // the wave's index as a SCALAR (readfirstlane): its group and plan then come through scalar loads, and everything the loop derives from them -- step
  // counts, puncturing masks, bit counts, the fifo's fill and the word index -- lives in SGPRs with SALU arithmetic and scalar branches (as vector values
  // they cost two dozen VGPRs, spills with scratch reloads at every refill, and exec-mask branches around wave-uniform conditions)
  const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(4 * blockIdx.x + (threadIdx.x >> 6));
  if (g >= ngroups) return;
  const WaveGroup grp = groups[g];
  const CodewordPlan pl = plans[grp.plan];
  const int nsteps = grp.nsteps;             // 32 x blocks + 6: the tail unit holds 6 steps
#if DABHIP_VIT_TIMES
  const bool stamp = ngroups > 2048 && g < 32768 && lane == 0;
  if (stamp) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_vit_times[4 * g] = wall_clock64();
    g_vit_times[4 * g + 3] = (static_cast<unsigned long long>(nsteps) << 8) | (xcc & 15u);
  }
#endif
  uint4* my_rec = reinterpret_cast<uint4*>(decisions + grp.dec_base * 64) + lane;

  // received words of this lane's record: tile = grp.first / 64 (job lists are padded to tiles of 64)
  // (a plan whose first word lies outside the row cannot arrive any more: the control plane drops such multiplexes, control_plane.hpp StreamFault;
  // the clamp keeps even a corrupted plan inside the tile's rows)
  const int word0 = min((pl.start_bit * kBits) >> 5, row_words - 1);
  const uint32_t* src = grouped + (static_cast<size_t>(grp.first >> 6) * row_words + word0) * 64 + lane;
  const int last_word = row_words - 1 - word0;
  uint64_t fifo = 0;
  int have = 0;                              // wave-uniform number of valid bits in fifo
  // (one word ahead; two -- hard -- and six -- soft -- words ahead measured slower, 5.04 against 4.88 and 6.78 against 6.63 ms)
  uint32_t nextw = src[0];
  int widx = 1;
  auto refill = [&]() {
    fifo |= static_cast<uint64_t>(nextw) << have;
    have += 32;
    nextw = src[static_cast<size_t>(min(widx, last_word)) * 64];
    ++widx;
  };

  constexpr int kScale = kBits == 1 ? 8 : kBits;     // hard decisions run on byte tags here
  pk16 pm[32], pn[32], pl4[32];
  init_metrics<kScale>(pm);
  int t = 0;
  for (int seg = 0; seg < 5; ++seg) {
    const uint32_t mask = seg < 4 ? pl.mask[seg] : (puncture_mask(8) & 0x00ffffffu);
    const int units = seg < 4 ? 4 * pl.blocks[seg] : 1;     // units of 8 trellis steps (= 32 mother-code bits)
    const int need = __popc(mask);
    // values taken by each of the 8 steps of a unit, 3 bits per step
    uint32_t counts = 0;
    for (int g = 0; g < 8; ++g) counts |= static_cast<uint32_t>(__popc((mask >> (4 * g)) & 15u)) << (3 * g);
    for (int u = 0; u < units; ++u) {
      if (kBits == 1) {
        if (have < need) refill();           // at most one refill per unit: need <= 32
        uint32_t ww[2] = {0, 0};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int n = (counts >> (3 * g)) & 7;
          const uint32_t m = (1u << n) - 1u;
          const uint32_t row = (static_cast<uint32_t>(fifo) & m) + lut_row_base(n);     // table row of (value, mask)
          fifo >>= n;
          ww[g >> 2] |= row << (8 * (g & 3));
        }
        have -= need;
        // (measurement build DABHIP_VIT_NOSTORE: the MSC code words' records all land on the same four blocks, i.e. stay in L2 --
        // the forward pass without its HBM traffic; the decoded MSC bytes are garbage then, the FIC is untouched)
        const size_t blk = (DABHIP_VIT_NOSTORE && nsteps > 800) ? static_cast<size_t>((t >> 3) & 3) : static_cast<size_t>(t >> 3);
        if (t + 8 <= nsteps) acs8_lut(ww[0], ww[1], lut, pm, pn, pl4, my_rec + blk * 256);
        else if (t < nsteps) acs8_tail_lut(ww[0], ww[1], nsteps - t, lut, pm, pn, pl4, my_rec + blk * 256);
      } else {
        // a step's values straight out of the fifo's low word (4 n <= 16 bits; n wave-uniform): one 32-bit and, one 64-bit shift per step -- collecting
        // them in 64-bit words first, to be taken apart again by the steps, cost seven instructions more per step (161 -> 150 per step)
        uint32_t x[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int nb = 4 * ((counts >> (3 * g)) & 7);        // bits of this step: 4 per received value
          if (have < nb) refill();
          x[g] = static_cast<uint32_t>(fifo) & ((1u << nb) - 1u);
          fifo >>= nb;
          have -= nb;
        }
        if (t + 4 <= nsteps) acs4_soft(x[0], x[1], x[2], x[3], soft_lut, pm, pn, pl4, my_rec + static_cast<size_t>(t >> 2) * 128);
        if (t + 8 <= nsteps) acs4_soft(x[4], x[5], x[6], x[7], soft_lut, pm, pn, pl4, my_rec + static_cast<size_t>((t >> 2) + 1) * 128);
        else if (t + 4 < nsteps) acs_tail_soft(x[4], x[5], x[6], nsteps - t - 4, soft_lut, pm, pn, my_rec + static_cast<size_t>((t >> 2) + 1) * 128);
      }
      t += 8;
      if ((t & (MetricScale<kScale>::kRebaseSteps - 1)) == 0 && t < nsteps) rebase_metrics<kScale>(pm);
    }
  }
#if DABHIP_VIT_TIMES
  if (stamp) g_vit_times[4 * g + 1] = wall_clock64();
#endif
  if (lane < grp.count && !(kBits == 1 && DABHIP_VIT_NOSTORE == 2 && nsteps > 800)) {     // (NOSTORE == 2: ... and without the chain-back)
    const int record = job_ids ? job_ids[grp.first + lane] : grp.first + lane;
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + static_cast<size_t>(record) * record_stride + pl.out_offset);
#if defined(DABHIP_CB_PRIO)
    __builtin_amdgcn_s_setprio(DABHIP_CB_PRIO);          // experiment (tools/gpu/cbprio.sh, round 4): the chain-back's few instructions ahead of the others' forward passes -- priority 1 and 3 measured the same as none (4.86 .. 4.95 ms all three, three rounds on one box)
#endif
    if (kBits == 1) chain_back8(my_rec, nsteps, prbs_words, dst);
    else chain_back(my_rec, nsteps, prbs_words, dst);
  }
#if DABHIP_VIT_TIMES
  if (stamp) g_vit_times[4 * g + 2] = wall_clock64();
#endif
}

#if DABHIP_VIT_TIMES
}  // namespace
}  // namespace dabhip
extern "C" int dabhip_debug_vit_times(unsigned long long* out, int ngroups)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dabhip::g_vit_times), sizeof(unsigned long long) * 4 * static_cast<size_t>(ngroups)) == hipSuccess ? 0 : -1;
}
namespace dabhip {
namespace {
#endif

#include "vit_two_lanes.hpp"
#include "vit_four_lanes.hpp"

// ---------------------------------------------------------------------------------------
// batched device-to-device copy (session carry-over of slots and rows): grid (piece, slice)
__global__ __launch_bounds__(256) void batched_copy_kernel(const CopyDesc* __restrict__ descs)
{
  const CopyDesc d = descs[blockIdx.x];
  const uint32_t* src = reinterpret_cast<const uint32_t*>(d.src);
  uint32_t* dst = reinterpret_cast<uint32_t*>(d.dst);
  const uint32_t nw = d.nbytes >> 2;
  for (uint32_t i = blockIdx.y * 256u + threadIdx.x; i < nw; i += gridDim.y * 256u) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------
// CRC of the 12 FIBs of each TF (misc.c:145-150)
__device__ __forceinline__ uint16_t crc16_step(uint16_t crc, uint8_t byte, const uint16_t* tab)
{
  return static_cast<uint16_t>(tab[(byte ^ (crc >> 8)) & 0xff] ^ (crc << 8));
}

__global__ void fib_crc_kernel(const uint8_t* __restrict__ fibs, int nfib, const uint16_t* __restrict__ crc_tab,
                               uint8_t* __restrict__ ok)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nfib) return;
  const uint8_t* f = fibs + static_cast<size_t>(i) * 32;
  uint16_t crc = 0xffff;
  for (int k = 0; k < 32; ++k) crc = crc16_step(crc, f[k], crc_tab);
  ok[i] = crc == 0x1d0f;
}

// CRC-16/CCITT is linear over GF(2): crc(A || B) = shift(crc(A), |B|) xor crc_0(B), where shift multiplies by
// x^(8 |B|) modulo the polynomial.  shift_cols[i][b] = image of bit b under a shift by 2^i bytes.
__device__ __forceinline__ uint16_t crc_shift(uint16_t crc, int nbytes, const uint16_t* shift_cols)
{
  for (int i = 0; nbytes; ++i, nbytes >>= 1) {
    if (nbytes & 1) {
      uint16_t y = 0;
#pragma unroll
      for (int bit = 0; bit < 16; ++bit)
        if ((crc >> bit) & 1) y ^= shift_cols[i * 16 + bit];
      crc = y;
    }
  }
  return crc;
}

// one WAVE per ETI frame: header bytes (built by the host control plane, init_eti misc.c:153-213), the 96 FIB
// bytes of the oldest CIF (misc.c:239), EOF CRC over FIC+MST, RFU, TIST (misc.c:281-292).  Each lane takes a
// slice of the CRC range; the partial CRCs are combined with crc_shift.  The 0x55 padding behind the trailer is written here too.
// The CRC runs four bytes at a time ("slicing by 4": tab[k][x] = CRC of byte x followed by k zero bytes, so that one 32-bit word
// costs four INDEPENDENT look-ups instead of a chain of four); the range is a multiple of 8 bytes (96 + 8-byte sub-channel units)
// and starts 4-byte aligned, so a lane's slice is whole words.  (Round 3: the words dealt to the lanes one by one instead -- every load of the wave one
// 256-byte run, a lane's part folded by Horner's rule in 256-byte steps with two more look-ups per word -- was SLOWER, 0.25 against 0.22 ms: the kernel
// is not bound by the lane-strided loads.  Nor by the bank conflicts of the random 2-byte look-ups: a copy of the tables per LDS bank (64 KB, 16 frames per
// workgroup, every look-up conflict-free) also came out at 0.25 ms.  What it was bound by: the log-step shift at the end, up to 13 DEPENDENT levels of
// look-ups in a table that sat in global memory -- with the 448 bytes of shift matrices in LDS the kernel takes 0.14 ms.)
__global__ __launch_bounds__(256) void eti_finish_kernel(const EtiFrameMeta* __restrict__ meta, int nframes,
                                                         const uint8_t* __restrict__ headers, int header_stride,
                                                         const uint8_t* __restrict__ fibs, const uint16_t* __restrict__ crc_tab,
                                                         const uint16_t* __restrict__ shift_cols, uint8_t* __restrict__ eti)
{
  __shared__ uint16_t tab[4][256];
  __shared__ uint16_t cols[14 * 16];                     // the shift matrices, next to the lanes (see crc_shift below: up to 13 dependent levels per lane)
  {
    uint16_t v = crc_tab[threadIdx.x];
    tab[0][threadIdx.x] = v;
    if (threadIdx.x < 14 * 16) cols[threadIdx.x] = shift_cols[threadIdx.x];
    __syncthreads();
#pragma unroll
    for (int k = 1; k < 4; ++k) {                          // one more zero byte: v <- step(v, 0)
      v = static_cast<uint16_t>(tab[0][v >> 8] ^ (v << 8));
      tab[k][threadIdx.x] = v;
    }
    __syncthreads();
  }
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (f >= nframes) return;
  const EtiFrameMeta m = meta[f];
  uint8_t* e = eti + static_cast<size_t>(f) * kEtiBytes;
  const uint8_t* h = headers + static_cast<size_t>(f) * header_stride;
  const uint8_t* fb = fibs + static_cast<size_t>(m.fib_block) * 96;
  for (int i = lane; i < m.header_len; i += 64) e[i] = h[i];
  for (int i = lane; i < 96; i += 64) e[m.header_len + i] = fb[i];
  const int n = 96 + m.mst_bytes;                       // CRC range: FIBs then the decoded sub-channel data
  const int nw = n >> 2;                                 // whole words (see above)
  const int chunk = (nw + 63) / 64;
  const int lo = min(nw, lane * chunk), hi = min(nw, lo + chunk);
  unsigned crc = lane == 0 ? 0xffffu : 0u;
  const uint32_t* fbw = reinterpret_cast<const uint32_t*>(fb);
  const uint32_t* mstw = reinterpret_cast<const uint32_t*>(e + m.header_len);   // MST bytes were written by the Viterbi kernel
  for (int i0 = lo; i0 < hi; i0 += 8) {                  // words fetched 8 at a time, ahead of the look-ups
    uint32_t w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + u, hi - 1);
      w[u] = i < 24 ? fbw[i] : mstw[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < hi)
        crc = tab[3][((crc >> 8) ^ w[u]) & 0xff] ^ tab[2][(crc ^ (w[u] >> 8)) & 0xff] ^ tab[1][(w[u] >> 16) & 0xff] ^ tab[0][w[u] >> 24];
  }
  unsigned acc = crc_shift(static_cast<uint16_t>(crc), n - 4 * hi, cols);
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) acc ^= __shfl_xor(acc, s);
  if (lane == 0) {
    int pos = m.header_len + n;
    const uint16_t out = static_cast<uint16_t>(~acc);
    e[pos++] = static_cast<uint8_t>(out >> 8);
    e[pos++] = static_cast<uint8_t>(out & 0xff);
    for (int i = 0; i < 6; ++i) e[pos++] = 0xff;
  }
  // the rest of the frame is padding (misc.c:295): header, FIBs, every sub-channel's bytes (multiples of 8) and the trailer above are
  // all written by someone, so the frame needs no fill beforehand
  uint32_t* ew = reinterpret_cast<uint32_t*>(e);
  for (int i = (m.header_len + n + 8) / 4 + lane; i < kEtiBytes / 4; i += 64) ew[i] = 0x55555555u;
}

}  // namespace

// Segment upload by a kernel that READS page-locked host memory over PCIe (dabhip_stream_prefetch): one launch moves the segments of
// all streams -- a few hundred copy commands of a few MB each cost the copy engines ~8 us apiece on top of the transfer.  A small
// persistent grid (the link, not the CUs, is the limit; the decode of the previous segment runs beside it): workgroup w takes pieces
// w, w + G, ... of 64 KB; 16-byte nontemporal loads and stores, byte tails by the piece's first lanes.
// one piece of a copy, any alignment of source and destination: the destination is brought to a 16-byte boundary by the piece's first lanes, the body
// goes as 16-byte loads from wherever the source then stands (gfx950 serves unaligned global loads; the compiler emits global_load_dwordx4 for them)
// and aligned nontemporal 16-byte stores, the tail bytes by the first lanes again
struct __attribute__((packed, aligned(1))) Unaligned16 { vuint4 v; };
__device__ __forceinline__ void copy_piece(const uint8_t* __restrict__ s, uint8_t* __restrict__ t, uint32_t n)
{
  const uint32_t head = min(n, static_cast<uint32_t>(-reinterpret_cast<uintptr_t>(t)) & 15u);
  if (threadIdx.x < head) t[threadIdx.x] = s[threadIdx.x];
  s += head;
  t += head;
  n -= head;
  const uint32_t nv = n >> 4;
  if ((reinterpret_cast<uintptr_t>(s) & 15u) == 0) {
    for (uint32_t v = threadIdx.x; v < nv; v += 256u)
      __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const vuint4*>(s) + v), reinterpret_cast<vuint4*>(t) + v);
  } else {
    for (uint32_t v = threadIdx.x; v < nv; v += 256u)
      __builtin_nontemporal_store(reinterpret_cast<const Unaligned16*>(s + 16u * v)->v, reinterpret_cast<vuint4*>(t) + v);
  }
  for (uint32_t b = (nv << 4) + threadIdx.x; b < n; b += 256u) t[b] = s[b];
}

__global__ __launch_bounds__(256) void host_gather_kernel(const CopyDesc* __restrict__ descs, int ndesc)
{
  constexpr uint32_t kPiece = 64u << 10;
  uint32_t piece0 = 0;                                     // index of the first piece of descriptor i in the global piece order
  for (int i = 0; i < ndesc; ++i) {
    const CopyDesc d = descs[i];
    const uint32_t npieces = static_cast<uint32_t>((static_cast<uint64_t>(d.nbytes) + kPiece - 1) / kPiece);   // (64-bit sum: sizes close to 2^32)
    // pieces of this descriptor that belong to this workgroup: global index = piece0 + p, taken when (piece0 + p) % gridDim.x == blockIdx.x
    uint32_t p = (blockIdx.x + gridDim.x - piece0 % gridDim.x) % gridDim.x;
    for (; p < npieces; p += gridDim.x) {
      const uint64_t off = static_cast<uint64_t>(p) * kPiece;
      copy_piece(d.src + off, d.dst + off, static_cast<uint32_t>(min(static_cast<uint64_t>(kPiece), d.nbytes - off)));
    }
    piece0 += npieces;
  }
}

// The same for copies inside the device (a session fed from device memory: the segments into their windows, the history in front of them --
// 2 x 256 copy commands of 5 .. 12 us each otherwise, one after the other): grid (piece slot, descriptor), pieces of 64 KB
__global__ __launch_bounds__(256) void device_gather_kernel(const CopyDesc* __restrict__ descs)
{
  constexpr uint32_t kPiece = 64u << 10;
  const CopyDesc d = descs[blockIdx.y];
  // (64-bit offset: with nbytes close to 2^32 a 32-bit one wraps below nbytes again and the loop never ends)
  for (uint64_t off = static_cast<uint64_t>(blockIdx.x) * kPiece; off < d.nbytes; off += static_cast<uint64_t>(gridDim.x) * kPiece)
    copy_piece(d.src + off, d.dst + off, static_cast<uint32_t>(min(static_cast<uint64_t>(kPiece), d.nbytes - off)));
}

hipError_t launch_host_gather(const CopyDesc* descs, int n, int workgroups, hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(host_gather_kernel, dim3(workgroups), dim3(256), 0, stream, descs, n);
  return hipGetLastError();
}

hipError_t launch_device_gather(const CopyDesc* descs, int n, uint32_t max_bytes, hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  const uint32_t pieces = static_cast<uint32_t>((static_cast<uint64_t>(max_bytes) + (64u << 10) - 1) >> 16);   // (no wrap for sizes close to 2^32)
  for (int i = 0; i < n; i += 65535)          // (grid.y holds 65,535 descriptors)
    hipLaunchKernelGGL(device_gather_kernel, dim3(std::max(1u, std::min(pieces, 64u)), std::min(n - i, 65535)), dim3(256), 0, stream, descs + i);
  return hipGetLastError();
}

hipError_t launch_batched_copy(const CopyDesc* descs, int n, hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(batched_copy_kernel, dim3(n, 16), dim3(256), 0, stream, descs);
  return hipGetLastError();
}

hipError_t launch_viterbi(const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans, const uint4* steps,
                          uint2* decisions, const uint32_t* prbs_words, uint8_t* out, int record_stride, hipStream_t stream)
{
  if (ngroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(viterbi_kernel, dim3(ngroups), dim3(64), 0, stream, groups, job_ids, plans, steps, decisions, prbs_words, out,
                     record_stride);
  return hipGetLastError();
}

hipError_t launch_regroup(int soft_bits, const int* job_ids, int ntiles, const DecodeJob* jobs, const int* stream_cif_base,
                          const uint32_t* rows, uint32_t* grouped, hipStream_t stream)
{
  if (ntiles <= 0) return hipSuccess;
  const int bits = soft_bits ? 4 : 1;
  for (int t0 = 0; t0 < ntiles; t0 += 32768) {
    const int nt = min(32768, ntiles - t0);
    const dim3 grid(((27 * bits + 3) / 4) * 8 * ((nt + 7) / 8));   // 108 x bits plane words, 16 per workgroup; tiles in eights (see the kernel)
    uint32_t* dst = grouped + static_cast<size_t>(t0) * 1728 * bits * 64;
    if (soft_bits) hipLaunchKernelGGL(regroup_kernel<4>, grid, dim3(256), 0, stream, job_ids + static_cast<size_t>(t0) * 64, jobs, stream_cif_base, rows, dst, nt);
    else hipLaunchKernelGGL(regroup_kernel<1>, grid, dim3(256), 0, stream, job_ids + static_cast<size_t>(t0) * 64, jobs, stream_cif_base, rows, dst, nt);
  }
  return hipGetLastError();
}

hipError_t launch_fic_group(const uint32_t* fic_rows, int first_block, int nblocks, int block_words, uint32_t* grouped, hipStream_t stream)
{
  if (nblocks <= 0) return hipSuccess;
  const int ntiles = (nblocks + 63) / 64;
  for (int t0 = 0; t0 < ntiles; t0 += 32768) {
    const int nt = min(32768, ntiles - t0);
    hipLaunchKernelGGL(fic_group_kernel, dim3(8, nt), dim3(256), 0, stream, fic_rows, first_block + t0 * 64, nblocks - t0 * 64, block_words,
                       grouped + static_cast<size_t>(t0) * block_words * 64);
  }
  return hipGetLastError();
}

hipError_t launch_viterbi_fused(int soft_bits, const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans,
                                const uint32_t* grouped, int row_words, uint2* decisions, const uint32_t* prbs_words, uint8_t* out,
                                int record_stride, hipStream_t stream)
{
  if (ngroups <= 0) return hipSuccess;
  if (soft_bits)
    hipLaunchKernelGGL(viterbi_fused_kernel<4>, dim3((ngroups + 3) / 4), dim3(256), 0, stream, groups, ngroups, job_ids, plans, grouped, row_words,
                       decisions, prbs_words, out, record_stride);
  else
    hipLaunchKernelGGL(viterbi_fused_kernel<1>, dim3((ngroups + 3) / 4), dim3(256), 0, stream, groups, ngroups, job_ids, plans, grouped, row_words,
                       decisions, prbs_words, out, record_stride);
  return hipGetLastError();
}

// hard decisions, two lanes per code word (vit_two_lanes.hpp): two waves per group
hipError_t launch_viterbi_fused_two(const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans, const uint32_t* grouped, int row_words,
                                    uint2* decisions, const uint32_t* prbs_words, uint8_t* out, int record_stride, hipStream_t stream)
{
  if (ngroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(viterbi_fused_two_kernel, dim3((2 * ngroups + 3) / 4), dim3(256), 0, stream, groups, ngroups, job_ids, plans, grouped, row_words, decisions,
                     prbs_words, out, record_stride);
  return hipGetLastError();
}

// hard decisions, 2^NL lanes per code word without per-lane tables (vit_four_lanes.hpp): lanes = 2 or 4 waves' worth of lanes per group
hipError_t launch_viterbi_fused_lanes(int lanes, const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans, const uint32_t* grouped,
                                      int row_words, uint2* decisions, const uint32_t* prbs_words, uint8_t* out, int record_stride, hipStream_t stream)
{
  if (ngroups <= 0) return hipSuccess;
  if (lanes == 4)
    hipLaunchKernelGGL(viterbi_fused_lanes_kernel<2>, dim3(ngroups), dim3(256), 0, stream, groups, ngroups, job_ids, plans, grouped, row_words, decisions, prbs_words,
                       out, record_stride);
  else
    hipLaunchKernelGGL(viterbi_fused_lanes_kernel<1>, dim3((2 * ngroups + 3) / 4), dim3(256), 0, stream, groups, ngroups, job_ids, plans, grouped, row_words, decisions,
                       prbs_words, out, record_stride);
  return hipGetLastError();
}

hipError_t launch_fib_crc(const uint8_t* fibs, int nfib, const uint16_t* crc_tab, uint8_t* ok, hipStream_t stream)
{
  if (nfib <= 0) return hipSuccess;
  hipLaunchKernelGGL(fib_crc_kernel, dim3((nfib + 255) / 256), dim3(256), 0, stream, fibs, nfib, crc_tab, ok);
  return hipGetLastError();
}

hipError_t launch_eti_finish(const EtiFrameMeta* meta, int nframes, const uint8_t* headers, int header_stride, const uint8_t* fibs,
                             const uint16_t* crc_tab, const uint16_t* shift_cols, uint8_t* eti, hipStream_t stream)
{
  if (nframes <= 0) return hipSuccess;
  hipLaunchKernelGGL(eti_finish_kernel, dim3((nframes + 3) / 4), dim3(256), 0, stream, meta, nframes, headers, header_stride, fibs,
                     crc_tab, shift_cols, eti);
  return hipGetLastError();
}

}  // namespace dabhip
