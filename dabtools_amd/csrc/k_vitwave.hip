// k_vitwave.hip — the LOW-LATENCY form of the channel decoder: one WAVE per code word, lane = trellis state.
//
// viterbi_fused_kernel (k_decode.hip) decodes 64 code words per wave, one per lane: the cheapest form per code word (no cross-lane
// traffic, 1.85 wave-instructions per code word and trellis step), but one wave walks its 4614 steps in 1.4 ms whatever the batch.
// A single live ensemble -- the reference's only use (dab2eti.c:60-115, one demod thread) and BASELINE configs[1] -- has 196 ETI
// frames x 12 code words per 64-TF decode: 48 waves on a chip with 1024 SIMDs, and the step is that one wave's latency.  Here the
// 64 states of ONE code word sit in the 64 lanes of a wave ("wavefront-shuffle add-compare-select"): a trellis step is ~10 wave
// instructions instead of ~118, so a code word is through in a tenth of the time; per code word it costs six times the lane-ops,
// which is why the engine uses it only below a batch size (Engine::msc_launch_async / fic_decode_slots_async).
//
// Same decisions as viterbi.c:352-451 (and as the fused kernel): metrics are agreement-minus-disagreement counts (hard) or the sum
// of the signed soft values (soft) -- the common part of a step's branch metrics cancels in every comparison; the high predecessor
// wins only when strictly better (viterbi.c:411), start in state 0, chain back from state 0, MSB first, descrambled (misc.c:41-58).
//
// Lane mapping.  State s sits in lane rotl6(s, r_t) with r_t = -t mod 6.  The predecessors of new state i are (i >> 1) and
// (i >> 1) | 32, and rotl6(p, 1) = (i & ~1) | h: with this rotation both predecessors of the state a lane is about to hold sit in
// that lane itself and in the lane that differs in ONE bit, j_t = 5 - t mod 6.  A step is therefore one exchange with lane ^ 32, 16,
// 8, 4, 2, 1 in turn (DPP for 1, 2, 4, 8; LDS swizzle / permute for 16 and 32), never a general permutation.  The two lanes of
// such a pair are the two ends of one butterfly and share the branch code word, so each computes P = M + g and Q = M' - g
// (M' the partner's metric, g the code's signed branch metric) and keeps max(P, Q).
#include <hip/hip_runtime.h>

#include <mutex>

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

// the metric of the lane that differs in bit J
template <int J>
__device__ __forceinline__ int partner(int m)
{
  if (J == 0) return __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, false);            // quad_perm [1, 0, 3, 2]
  if (J == 1) return __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, false);            // quad_perm [2, 3, 0, 1]
  if (J == 2) {                                                                           // lane ^ 4: two shifts by four, each written to its half of the banks
    const int a = __builtin_amdgcn_update_dpp(0, m, 0x114, 0xF, 0xA, false);              // row_shr:4 -> lanes 4..7, 12..15 read lane - 4
    return __builtin_amdgcn_update_dpp(a, m, 0x104, 0xF, 0x5, false);                     // row_shl:4 -> lanes 0..3, 8..11 read lane + 4
  }
  if (J == 3) return __builtin_amdgcn_update_dpp(0, m, 0x128, 0xF, 0xF, false);           // row_ror:8 = lane ^ 8 within a row of 16
  if (J == 4) return __builtin_amdgcn_ds_swizzle(m, 0x401F);                              // bit mode: and 0x1f, or 0, xor 0x10
  return __builtin_amdgcn_ds_bpermute(static_cast<int>(((threadIdx.x & 63u) ^ 32u) << 2), m);
}

template <int J>
__device__ __forceinline__ constexpr uint64_t bit_mask()       // lanes whose bit J is set
{
  return J == 0 ? 0xAAAAAAAAAAAAAAAAull : J == 1 ? 0xCCCCCCCCCCCCCCCCull : J == 2 ? 0xF0F0F0F0F0F0F0F0ull : J == 3 ? 0xFF00FF00FF00FF00ull
       : J == 4 ? 0xFFFF0000FFFF0000ull : 0xFFFFFFFF00000000ull;
}

// one trellis step.  w: the step's table word (wave-uniform): hard: eight signed 4-bit fields d_c = agreements - disagreements of code c
// (3 distinct code bits, bit 3 = bit 0) with the received bits; soft: four signed 8-bit fields d_c = sum_j (c_j ? -s_j : s_j), c = 0..3,
// the complementary code's value being -d_c.  sh / sg: this lane's field and sign in the phase of this step.
template <int J, int kBits>
__device__ __forceinline__ void wave_step(int& m, uint32_t w, int sh, int sg, uint32_t& acc_lo, uint32_t& acc_hi, int k)
{
  int g = __builtin_amdgcn_sbfe(static_cast<int>(w), sh, kBits == 1 ? 4 : 8);
  if (kBits != 1) g *= sg;
  const int p = m + g;                                      // this lane's own metric continues along its branch
  const int q = partner<J>(m) - g;                          // the partner's along the complementary branch
  // Which of the two is the HIGH predecessor's?  Lane bit J = bit 0 of the new state = which predecessor this lane's own metric is.
  // bit clear: own = low, partner = high: decision = q > p.  bit set: own = high: decision = p > q = !(q >= p).   (viterbi.c:411: strictly)
  const uint64_t gt = __builtin_amdgcn_ballot_w64(q > p), ge = __builtin_amdgcn_ballot_w64(q >= p);
  const uint64_t dec = (gt & ~bit_mask<J>()) | (~ge & bit_mask<J>());
  m = max(p, q);
  acc_lo = write_lane(acc_lo, static_cast<uint32_t>(dec), k);
  acc_hi = write_lane(acc_hi, static_cast<uint32_t>(dec >> 32), k);
}

struct WaveSegs {                // the five puncturing segments of a code word (depuncture.c:45-132), wave-uniform
  int start[5];                  // first trellis step
  int base[5];                   // received values before the segment
  uint32_t mask[5];              // 32 mother-code bits = 8 steps: bit 4 g + k = value k of step g was transmitted
  int need[5];                   // received values per unit of 8 steps
};

// Table words of the 64 steps t0 .. t0 + 63, one per lane, from the received values (n, v) of the lane's step.
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
// Dynamic LDS: per wave `chunks` x 128 words of decisions (64 steps x 64 states).
template <int kBits>
__global__ __launch_bounds__(256) void viterbi_wave_kernel(const WaveGroup* __restrict__ groups, int ngroups, const int* __restrict__ job_ids,
                                                           const CodewordPlan* __restrict__ plans, const uint32_t* __restrict__ grouped, int row_words,
                                                           const uint32_t* __restrict__ prbs_words, uint8_t* __restrict__ out, int record_stride, int chunks)
{
  extern __shared__ uint32_t dec_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cw = __builtin_amdgcn_readfirstlane(static_cast<int>(blockDim.x >> 6) * blockIdx.x + wave);     // 4, 2 or 1 waves per workgroup (what the LDS allows)
  const int g = cw >> 6, l = cw & 63;
  if (g >= ngroups) return;
  const WaveGroup grp = groups[g];
  if (l >= grp.count) return;
  const CodewordPlan pl = plans[grp.plan];
  const int nsteps = grp.nsteps;
  uint32_t* const dec = dec_lds + static_cast<size_t>(wave) * chunks * 128;

  // per-lane constants of the six phases: the branch code this lane's butterfly uses when the exchange runs over lane bit j = 5 - phase
  int sh[6], sgn[6];
#pragma unroll
  for (int ph = 0; ph < 6; ++ph) {
    const unsigned j = 5u - ph;
    const unsigned i = vw_rotl6(static_cast<unsigned>(lane), (6u - j) % 6u);      // the new state this lane will hold
    const unsigned b = (static_cast<unsigned>(lane) >> j) & 1u;                  // = i & 1: own metric is the low (0) / high (1) predecessor's
    const unsigned c = vw_code3(i) ^ (b ? 7u : 0u);                              // code on the branch own predecessor -> i
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
  const int word0 = (pl.start_bit * kBits) >> 5;
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
  uint32_t acc_lo = 0, acc_hi = 0, tw = 0;
  auto begin_chunk = [&](int t0) {
    const uint32_t shft = static_cast<uint32_t>(pos * kBits) & 31u;
    const uint32_t v = static_cast<uint32_t>(((static_cast<uint64_t>(w1) << 32) | w0) >> shft) & ((1u << (n * kBits)) - 1u);   // n kBits <= 16
    tw = table_word<kBits>(n, v);
    if (t0 + 64 < nsteps) fetch(t0 + 64);
  };
  auto end_chunk = [&](int c) {
    dec[c * 128 + lane] = acc_lo;
    dec[c * 128 + 64 + lane] = acc_hi;
  };
#define DABHIP_WAVE_STEP(J, PH)                                                                                      \
  {                                                                                                                  \
    const int k = t & 63;                                                                                            \
    if (k == 0) begin_chunk(t);                                                                                      \
    wave_step<J, kBits>(m, static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(tw), k)), sh[PH], sgn[PH], acc_lo, acc_hi, k); \
    if (k == 63) end_chunk(t >> 6);                                                                                  \
    if (++t >= nsteps) break;                                                                                        \
  }
  for (int t = 0;;) {
    DABHIP_WAVE_STEP(5, 0)
    DABHIP_WAVE_STEP(4, 1)
    DABHIP_WAVE_STEP(3, 2)
    DABHIP_WAVE_STEP(2, 3)
    DABHIP_WAVE_STEP(1, 4)
    DABHIP_WAVE_STEP(0, 5)
  }
#undef DABHIP_WAVE_STEP
  if (nsteps & 63) end_chunk(nsteps >> 6);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's own LDS stores are done before it reads them back

  // ---- chain back from state 0 (viterbi.c:438-450), descramble (misc.c:41-58), pack MSB first -------------------------------
  const int record = job_ids ? job_ids[grp.first + l] : grp.first + l;
  uint32_t* const dst = reinterpret_cast<uint32_t*>(out + static_cast<size_t>(record) * record_stride + pl.out_offset);
  const int nwords = (nsteps - 6) >> 5;          // data bits are a multiple of 32 (32 x blocks)
  unsigned state = 0;
  uint32_t bits = 0, outv = 0, lo = 0, hi = 0;
  int ph = (nsteps - 1) % 6;
  for (int t = nsteps - 1; t >= 6; --t) {
    const int k = t & 63;
    if (k == 63 || t == nsteps - 1) {
      lo = dec[(t >> 6) * 128 + lane];
      hi = dec[(t >> 6) * 128 + 64 + lane];
    }
    const unsigned j = 5u - static_cast<unsigned>(ph);
    const unsigned at = ((state << j) | (state >> (6u - j))) & 63u;              // the lane that held `state` after step t (j = 0: state >> 6 = 0)
    const uint64_t word = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(hi), k))) << 32) |
                          static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(lo), k));
    const unsigned d = static_cast<unsigned>(word >> at) & 1u;
    state = (state | (d << 6)) >> 1;
    const int i = t - 6;                                                           // data bit index
    bits |= d << (8 * ((i >> 3) & 3) + (7 - (i & 7)));
    if ((i & 31) == 0) {
      const int wi = i >> 5;
      outv = write_lane(outv, bits, wi & 63);
      bits = 0;
      if ((wi & 63) == 0 && wi + lane < nwords) dst[wi + lane] = outv ^ prbs_words[wi + lane];     // words wi .. wi + 63, complete since the last flush
    }
    ph = ph == 0 ? 5 : ph - 1;
  }
}

// dynamic LDS above 64 KB needs the attribute, per DEVICE (several engines of one process may sit on different devices: dabhip_multi)
template <int kBits>
hipError_t wave_attr()
{
  static std::once_flag once[64];
  static hipError_t result[64];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  dev &= 63;
  std::call_once(once[dev], [&]() {
    result[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(&viterbi_wave_kernel<kBits>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  return result[dev];
}

}  // namespace

// groups [0, ngroups) all have at most max_nsteps trellis steps; every (group, valid lane) becomes one wave
hipError_t launch_viterbi_wave(int soft_bits, const WaveGroup* groups, int ngroups, int max_nsteps, const int* job_ids, const CodewordPlan* plans,
                               const uint32_t* grouped, int row_words, const uint32_t* prbs_words, uint8_t* out, int record_stride, hipStream_t stream)
{
  if (ngroups <= 0) return hipSuccess;
  const int chunks = (max_nsteps + 63) / 64;
  // waves per workgroup: as many of 4, 2, 1 as the decisions (512 bytes per 64 steps and wave) leave room for in a CU's 160 KB
  // (the longest code word, 384 kbit/s = 9222 steps, takes 74 KB)
  int waves = 4;
  while (waves > 1 && static_cast<size_t>(waves) * chunks * 512 > 160 * 1024) waves >>= 1;
  const size_t lds = static_cast<size_t>(waves) * chunks * 512;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const hipError_t a = soft_bits ? wave_attr<4>() : wave_attr<1>();
  if (a != hipSuccess) return a;
  const dim3 grid(static_cast<unsigned>(ngroups) * (64u / waves)), block(64u * waves);
  if (soft_bits)
    hipLaunchKernelGGL(viterbi_wave_kernel<4>, grid, block, lds, stream, groups, ngroups, job_ids, plans, grouped, row_words, prbs_words, out, record_stride, chunks);
  else
    hipLaunchKernelGGL(viterbi_wave_kernel<1>, grid, block, lds, stream, groups, ngroups, job_ids, plans, grouped, row_words, prbs_words, out, record_stride, chunks);
  return hipGetLastError();
}

}  // namespace dabhip
