// k_vitwave.hip — the LOW-LATENCY form of the channel decoder: one WAVE per code word, lane = trellis state.
//
// viterbi_fused_kernel (k_decode.hip) decodes 64 code words per wave, one per lane: the cheapest form per code word (no cross-lane
// traffic, 1.85 wave-instructions per code word and trellis step), but one wave walks its 4614 steps in 1.4 ms whatever the batch.
// A single live ensemble -- the reference's only use (dab2eti.c:60-115, one demod thread) and BASELINE configs[1] -- has 196 ETI
// frames x 12 code words per 64-TF decode: 48 waves on a chip with 1024 SIMDs, and the step is that one wave's latency.  Here the
// 64 states of ONE code word sit in the 64 lanes of a wave ("wavefront-shuffle add-compare-select"): a trellis step is 8 wave
// instructions instead of ~118, so a code word is through in a fraction of the time; per code word it costs four times the lane-ops,
// which is why the engine uses it only below a batch size (Engine::launch_decode_batch / fic_decode_slots_async).
//
// Same decisions as viterbi.c:352-451 (and as the fused kernel): metrics are agreement-minus-disagreement counts (hard) or the sum
// of the signed soft values (soft) -- the common part of a step's branch metrics cancels in every comparison; the high predecessor
// wins only when strictly better (viterbi.c:411), start in state 0, chain back from state 0, MSB first, descrambled (misc.c:41-58).
//
// Lane mapping.  State s sits in lane rotl6(s, r_t) with r_t = -t mod 6.  The predecessors of new state i are (i >> 1) and
// (i >> 1) | 32, and rotl6(p, 1) = (i & ~1) | h: with this rotation both predecessors of the state a lane is about to hold sit in
// that lane itself and in the lane that differs in ONE bit, j_t = 5 - t mod 6.  A step is therefore one exchange with lane ^ 32, 16,
// 8, 4, 2, 1 in turn (v_permlane32_swap / v_permlane16_swap for the first two, DPP for the others), never a general permutation and
// never through the LDS.  The two lanes of such a pair are the two ends of one butterfly and share the branch code word up to its sign.
//
// No per-step scalar work and no branches inside a chunk of 60 steps: a lone wave issues in order, and every VALU -> SALU -> VALU hand-over
// (compare masks edited on the scalar unit, lane selects through M0, a branch per step) cost it tens of cycles -- the first version of this
// kernel, written that way, took 450 cycles per step and was SLOWER than the batch form.
#include <hip/hip_runtime.h>

#include "dab_tables.hpp"
#include "device_types.hpp"
#include "kernels.hpp"

// v_writelane_b32 (this compiler has the LLVM intrinsic but no clang builtin for it): lane `lane` of `old` replaced by the wave-uniform `value`
extern "C" __device__ int dabhip_llvm_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

namespace dabhip {
namespace {

__device__ __forceinline__ uint32_t write_lane(uint32_t old, uint32_t value, int lane)
{
  return static_cast<uint32_t>(dabhip_llvm_writelane(static_cast<int>(value), lane, static_cast<int>(old)));
}

__host__ __device__ constexpr unsigned vw_parity(unsigned x)
{
  x ^= x >> 4;
  x ^= x >> 2;
  x ^= x >> 1;
  return x & 1u;
}
// bits 0..2 of the code word on the branch low predecessor -> state i (bit j = parity(i & poly_j), viterbi.c:35,373-381; bit 3 = bit 0)
__host__ __device__ constexpr unsigned vw_code3(unsigned i)
{
  return vw_parity(i & 0x6d) | (vw_parity(i & 0x4f) << 1) | (vw_parity(i & 0x53) << 2);
}
__host__ __device__ constexpr unsigned vw_rotl6(unsigned s, unsigned j) { return j == 0 ? s & 63u : ((s << j) | (s >> (6 - j))) & 63u; }

constexpr int kInitOther = -(1 << 24);       // "unreachable" start metric of states 1..63 (viterbi.c:387-389): far below what six steps can collect
constexpr int kChunk = 60;                   // steps per chunk: ten rounds of the six exchange distances; 60 of the 64 lanes prepare a chunk's table words

// One trellis step with the exchange over lane bit J.  e: this lane's signed branch metric (see the kernel's phase constants), bsub: lane bit J.
// dreg collects one bit per step: the survivor decision of the state this lane holds afterwards -- as such for J = 5, 4; for J <= 3 complemented
// in the lanes whose bit J is set (the chain-back knows: that bit is bit 0 of the state).
//
// J = 5, 4 (lane ^ 32, lane ^ 16): v_permlane32_swap / v_permlane16_swap (gfx950) exchange the upper half (the odd rows) of their first operand
// with the lower half (the even rows) of the second.  Every lane puts M + e into the first and M - e into the second register, e = the metric
// of the branch LOW predecessor -> its new state: a lane with bit J clear is that low predecessor (its M + e is the low candidate, it needs the
// partner's M - e), a lane with bit J set is the high predecessor (its M - e is the high candidate, it needs the partner's M + e).  After the
// swap every lane holds the LOW candidate in the first and the HIGH candidate in the second register, whichever of the two is its own: one
// compare, strict as in viterbi.c:411, no lane-dependent tie rule.
// J = 3 .. 0: DPP reads the partner's metric inside the subtract; own candidate p = M + e, the other q = M' - e; which of them is the high
// predecessor's depends on lane bit J, so the compare is q > p - bit: the decision where the bit is clear, its complement where it is set.
template <int J>
__device__ __forceinline__ void wave_step(int& m, int e, int bsub, uint32_t& dreg)
{
  if (J >= 4) {
    int a = m + e, b = m - e;
    // (inline asm: the builtins of this compiler miscompile swaps, see k_fused.hip -- so the hazard recogniser does not see them either: a VALU write
    // of an operand needs two wait states before the swap reads it, like a DPP read)
    if (J == 5) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    m = max(a, b);
    asm volatile("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(dreg) : "v"(b), "v"(a) : "vcc");     // dreg = 2 dreg + (high > low)
  } else {
    const int p = m + e;
    int q;
    if (J == 0) q = __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, false) - e;           // quad_perm [1, 0, 3, 2]
    else if (J == 1) q = __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, false) - e;      // quad_perm [2, 3, 0, 1]
    else if (J == 3) q = __builtin_amdgcn_update_dpp(0, m, 0x128, 0xF, 0xF, false) - e;     // row_ror:8 = lane ^ 8 within a row of 16
    else {                                                                                  // lane ^ 4: two shifts by four, each written to its half of the banks
      const int x = __builtin_amdgcn_update_dpp(0, m, 0x114, 0xF, 0xA, false);              // row_shr:4 -> lanes 4..7, 12..15 read lane - 4
      q = __builtin_amdgcn_update_dpp(x, m, 0x104, 0xF, 0x5, false) - e;                    // row_shl:4 -> lanes 0..3, 8..11 read lane + 4
    }
    const int ps = p - bsub;
    m = max(p, q);
    asm volatile("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(dreg) : "v"(q), "v"(ps) : "vcc");
  }
}

struct WaveSegs {                // the five puncturing segments of a code word (depuncture.c:45-132), wave-uniform
  int start[5];                  // first trellis step
  int base[5];                   // received values before the segment
  uint32_t mask[5];              // 32 mother-code bits = 8 steps: bit 4 g + k = value k of step g was transmitted
  int need[5];                   // received values per unit of 8 steps
};

// The table word of a step from its received values (n of them, v): hard: eight signed 4-bit fields d_c = agreements - disagreements of code
// c (3 distinct code bits, bit 3 = bit 0) with the received bits; soft: four signed 8-bit fields d_c = sum_j (c_j ? -s_j : s_j), c = 0..3, the
// complementary code's value being -d_c.
template <int kBits>
__device__ __forceinline__ uint32_t table_word(int n, uint32_t v)
{
  uint32_t w = 0;
  if (kBits == 1) {
    const uint32_t m = (1u << n) - 1u;
#pragma unroll
    for (unsigned c = 0; c < 8; ++c) {
      const unsigned cw = c | ((c & 1u) << 3);
      const int d = 2 * __popc(~(v ^ cw) & m) - n;          // agreements - disagreements over the n received bits
      w |= (static_cast<uint32_t>(d) & 15u) << (4 * c);
    }
  } else {
    int s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = max(static_cast<int>(((v >> (4 * j)) & 15u) ^ 8u) - 8, -7);    // not received: nibble 0 -> 0; a -8 counts as -7 like in SoftLut
    const int a = s[0] + s[3];                              // generator 0 == generator 3: code bits 0 and 3 are the same bit
    const int d[4] = {a + s[1] + s[2], -a + s[1] + s[2], a - s[1] + s[2], -a - s[1] + s[2]};
#pragma unroll
    for (int c = 0; c < 4; ++c) w |= (static_cast<uint32_t>(d[c]) & 255u) << (8 * c);
  }
  return w;
}

// where the received values of step tau start in the code word's value stream, and how many it takes (0..4): the candidate of every
// segment, the one tau lies in selected (values, not addresses: the segment table is wave-uniform and stays in SGPRs)
__device__ __forceinline__ void step_input(const WaveSegs& sg, int tau, int nsteps, int* pos, int* n)
{
  int p = 0, c = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int rel = tau - sg.start[i], unit = rel >> 3, g = rel & 7;
    const int pi = sg.base[i] + unit * sg.need[i] + __popc(sg.mask[i] & ((1u << (4 * g)) - 1u));
    const int ci = __popc((sg.mask[i] >> (4 * g)) & 15u);
    const bool here = i == 0 || tau >= sg.start[i];
    p = here ? pi : p;
    c = here ? ci : c;
  }
  *pos = p;
  *n = tau < nsteps ? c : 0;
}

// One wave = one code word: (group g, lane l of the group) = the job the fused kernel's lane l of wave g would decode.
// decisions: per code word one row of 64 x 8 bytes per chunk of 60 steps (two 30-bit decision words per lane), at row grp.dec_base + l x chunks.
template <int kBits>
__global__ __launch_bounds__(256) void viterbi_wave_kernel(const WaveGroup* __restrict__ groups, int ngroups, const int* __restrict__ job_ids,
                                                           const CodewordPlan* __restrict__ plans, const uint32_t* __restrict__ grouped, int row_words,
                                                           uint2* __restrict__ decisions, const uint32_t* __restrict__ prbs_words, uint8_t* __restrict__ out,
                                                           int record_stride)
{
  const int lane = threadIdx.x & 63;
  const int cw = __builtin_amdgcn_readfirstlane(static_cast<int>(4 * blockIdx.x + (threadIdx.x >> 6)));
  const int g = cw >> 6, l = cw & 63;
  if (g >= ngroups) return;
  const WaveGroup grp = groups[g];
  if (l >= grp.count) return;
  const CodewordPlan pl = plans[grp.plan];
  const int nsteps = grp.nsteps, nchunks = (nsteps + kChunk - 1) / kChunk;
  uint32_t* const dec = reinterpret_cast<uint32_t*>(decisions + (static_cast<size_t>(grp.dec_base) + static_cast<size_t>(l) * nchunks) * 64);

  // per-lane constants of the six phases (phase = step mod 6, exchange over lane bit j = 5 - phase): where this lane's branch metric sits in a
  // step's table word, its sign (soft values), and lane bit j
  int sh[6], sgn[6], bit[6];
#pragma unroll
  for (int ph = 0; ph < 6; ++ph) {
    const unsigned j = 5u - ph;
    const unsigned i = vw_rotl6(static_cast<unsigned>(lane), (6u - j) % 6u);      // the new state this lane will hold
    const unsigned b = (static_cast<unsigned>(lane) >> j) & 1u;                  // = i & 1: own metric is the low (0) / high (1) predecessor's
    // j >= 4: the metric of the branch low predecessor -> i for every lane; j <= 3: that of the branch own predecessor -> i
    const unsigned c = vw_code3(i) ^ ((b && j <= 3) ? 7u : 0u);
    bit[ph] = static_cast<int>(b);
    if (kBits == 1) { sh[ph] = 4 * static_cast<int>(c); sgn[ph] = 1; }
    else { sh[ph] = 8 * static_cast<int>(c < 4 ? c : c ^ 7u); sgn[ph] = c < 4 ? 1 : -1; }
  }

  WaveSegs segs;
  {
    int t = 0, base = 0;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      segs.mask[s] = s < 4 ? pl.mask[s] : (puncture_mask(8) & 0x00ffffffu);
      segs.need[s] = __popc(segs.mask[s]);
      segs.start[s] = t;
      segs.base[s] = base;
      const int units = s < 4 ? 4 * pl.blocks[s] : 1;
      t += 8 * units;
      base += units * segs.need[s];
    }
  }
  // the code word's received values: word w of its row at src[64 w] (rows of 64 records interleaved word by word, regroup_kernel / fic_group_kernel)
  const int word0 = min((pl.start_bit * kBits) >> 5, row_words - 1);      // (clamped like in the batch form: see there)
  const uint32_t* src = grouped + (static_cast<size_t>(grp.first >> 6) * row_words + word0) * 64 + l;
  const int last_word = row_words - 1 - word0;

  int pos, n;
  uint32_t w0, w1;
  auto fetch = [&](int t0) {                   // the two words holding the values of step t0 + lane: issued a chunk ahead of their use
    step_input(segs, t0 + lane, nsteps, &pos, &n);
    const int wi = (pos * kBits) >> 5;
    w0 = src[static_cast<size_t>(min(wi, last_word)) * 64];
    w1 = src[static_cast<size_t>(min(wi + 1, last_word)) * 64];
  };
  fetch(0);

  int m = lane == 0 ? 0 : kInitOther;
  for (int c = 0; c < nchunks; ++c) {
    const uint32_t shft = static_cast<uint32_t>(pos * kBits) & 31u;
    const uint32_t v = static_cast<uint32_t>(((static_cast<uint64_t>(w1) << 32) | w0) >> shft) & ((1u << (n * kBits)) - 1u);   // n kBits <= 16
    const uint32_t tw = table_word<kBits>(n, v);           // lane k: the table word of step 60 c + k (steps past the end: nothing received, all metrics 0)
    if (c + 1 < nchunks) fetch(kChunk * (c + 1));
    uint32_t d0 = 0, d1 = 0;
    auto metric = [&](int k, int ph) -> int {
      const int x = __builtin_amdgcn_sbfe(__builtin_amdgcn_readlane(static_cast<int>(tw), k), sh[ph], kBits == 1 ? 4 : 8);
      return kBits == 1 ? x : x * sgn[ph];
    };
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int k = 6 * r;
      wave_step<5>(m, metric(k, 0), bit[0], d0);
      wave_step<4>(m, metric(k + 1, 1), bit[1], d0);
      wave_step<3>(m, metric(k + 2, 2), bit[2], d0);
      wave_step<2>(m, metric(k + 3, 3), bit[3], d0);
      wave_step<1>(m, metric(k + 4, 4), bit[4], d0);
      wave_step<0>(m, metric(k + 5, 5), bit[5], d0);
    }
#pragma unroll
    for (int r = 5; r < 10; ++r) {
      const int k = 6 * r;
      wave_step<5>(m, metric(k, 0), bit[0], d1);
      wave_step<4>(m, metric(k + 1, 1), bit[1], d1);
      wave_step<3>(m, metric(k + 2, 2), bit[2], d1);
      wave_step<2>(m, metric(k + 3, 3), bit[3], d1);
      wave_step<1>(m, metric(k + 4, 4), bit[4], d1);
      wave_step<0>(m, metric(k + 5, 5), bit[5], d1);
    }
    dec[c * 128 + lane] = d0;                              // step 60 c + k: bit 29 - k of d0 (k < 30), bit 59 - k of d1
    dec[c * 128 + 64 + lane] = d1;
  }
  __threadfence();                                         // the wave reads its own rows back below

  // ---- chain back from state 0 (viterbi.c:438-450), descramble (misc.c:41-58), pack MSB first -------------------------------
  const int record = job_ids ? job_ids[grp.first + l] : grp.first + l;
  uint32_t* const dst = reinterpret_cast<uint32_t*>(out + static_cast<size_t>(record) * record_stride + pl.out_offset);
  const int nwords = (nsteps - 6) >> 5;          // data bits are a multiple of 32 (32 x blocks)
  unsigned state = 0;
  uint32_t bits = 0, outv = 0;
  uint32_t n0 = dec[(nchunks - 1) * 128 + lane], n1 = dec[(nchunks - 1) * 128 + 64 + lane];
  for (int c = nchunks - 1; c >= 0; --c) {
    const uint32_t r0 = n0, r1 = n1;
    if (c > 0) {                                           // the rows of the chunk before, on their way while this one is walked
      n0 = dec[(c - 1) * 128 + lane];
      n1 = dec[(c - 1) * 128 + 64 + lane];
    }
    auto walk = [&](int k, int ph, uint32_t row, int top) {
      const int t = kChunk * c + k;
      if (t >= nsteps || t < 6) return;                    // steps 0..5 only flush the encoder's initial zeros (viterbi.c:361,431)
      const unsigned j = 5u - static_cast<unsigned>(ph);
      const unsigned at = j == 0 ? state : ((state << j) | (state >> (6u - j))) & 63u;      // the lane that held `state` after step t
      const uint32_t w = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(row), static_cast<int>(at)));
      unsigned d = (w >> (top - k)) & 1u;
      if (j <= 3) d ^= state & 1u;                         // recorded complemented where lane bit j (= bit 0 of the state) is set
      state = (state | (d << 6)) >> 1;
      const int i = t - 6;                                 // data bit index
      bits |= d << (8 * ((i >> 3) & 3) + (7 - (i & 7)));
      if ((i & 31) == 0) {
        const int wi = i >> 5;
        outv = write_lane(outv, bits, wi & 63);
        bits = 0;
        if ((wi & 63) == 0 && wi + lane < nwords) dst[wi + lane] = outv ^ prbs_words[wi + lane];     // words wi .. wi + 63, complete since the last flush
      }
    };
#pragma unroll 1
    for (int r = 9; r >= 5; --r) {
      const int k = 6 * r;
      walk(k + 5, 5, r1, 59);
      walk(k + 4, 4, r1, 59);
      walk(k + 3, 3, r1, 59);
      walk(k + 2, 2, r1, 59);
      walk(k + 1, 1, r1, 59);
      walk(k, 0, r1, 59);
    }
#pragma unroll 1
    for (int r = 4; r >= 0; --r) {
      const int k = 6 * r;
      walk(k + 5, 5, r0, 29);
      walk(k + 4, 4, r0, 29);
      walk(k + 3, 3, r0, 29);
      walk(k + 2, 2, r0, 29);
      walk(k + 1, 1, r0, 29);
      walk(k, 0, r0, 29);
    }
  }
}

}  // namespace

// every (group, valid lane) becomes one wave; decisions: rows of 64 x 8 bytes, group g's at groups[g].dec_base, 64 x ceil(nsteps / 60) of them
hipError_t launch_viterbi_wave(int soft_bits, const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans,
                               const uint32_t* grouped, int row_words, uint2* decisions, const uint32_t* prbs_words, uint8_t* out, int record_stride,
                               hipStream_t stream)
{
  if (ngroups <= 0) return hipSuccess;
  const dim3 grid(static_cast<unsigned>(ngroups) * 16u), block(256);
  if (soft_bits)
    hipLaunchKernelGGL(viterbi_wave_kernel<4>, grid, block, 0, stream, groups, ngroups, job_ids, plans, grouped, row_words, decisions, prbs_words, out, record_stride);
  else
    hipLaunchKernelGGL(viterbi_wave_kernel<1>, grid, block, 0, stream, groups, ngroups, job_ids, plans, grouped, row_words, decisions, prbs_words, out, record_stride);
  return hipGetLastError();
}

}  // namespace dabhip
