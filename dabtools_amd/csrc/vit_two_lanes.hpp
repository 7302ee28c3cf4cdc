// vit_two_lanes.hpp — the MSC / FIC decoder with TWO lanes per code word (hard decisions), for batches that leave the lane form one wave per SIMD.
//
// Included by k_decode.hip inside its anonymous namespace (it uses that file's helpers: branch codes, metric tables, records, bit packing); not a
// header of its own.  tools/models/twolane_model.py is this file's index algebra run on the CPU against a plain 64-state add-compare-select.
//
// Why: between ~8 and ~32 streams viterbi_fused_kernel<1> has one wave on most SIMDs, a lone wave issues an instruction every ~5.7 clocks whatever the
// instruction costs, and a step of 64 code words is 96 of them (64 adds, 32 packed max) on a chain of 4,614 dependent steps.  Half the code words per wave
// with half the registers per lane is 48 instructions per step -- IF the two lanes of a code word never have to sort registers for each other.  They do not
// when the LANE bit (which lane holds a state) rotates with the trellis the way the PAIR bit (which two states share a register) already does:
//   * a state bit moves up one place per step, so with the lane bit at place L < 5 both predecessors of a butterfly share it: the step is local to the
//     lane (8 butterflies of the usual form on 8 + 8 registers), and the lane bit is at place L + 1 afterwards;
//   * at L = 5 every butterfly straddles the lanes; lane 0 then computes all sixteen EVEN successors and lane 1 all sixteen ODD ones, each from all of its
//     own registers and all of its partner's registers OF THE SAME INDEX -- a DPP quad_perm operand on the add, measured free (profiles/r06_acs_split.txt) --
//     and the lane bit re-enters at place 0;
//   * L(t) = (3 + t) mod 6 never meets the pair bit tau(t) = t mod 4, and sits on place 1, 5 or 3 at the re-pairing points (never on the places 0 and 4
//     whose states the byte permute joins): a schedule of period 12, 24 with the 8-step record blocks = three variants of the block (kV = block mod 3);
//   * a butterfly's branch code depends on its lane through one XOR, code(2 j) ^ lane * code(2 << L), and at L = 5 lane 1's own operand is the HIGH
//     predecessor (untagged): both are permutations of the 8 + 8 metric words of a step, i.e. lane 1 reads them from tables of its own (24, lane 0
//     keeps the 8 of the lane form: 64 KB of LDS), no VALU work.
// Survivor records keep the lane form's volume and byte order per register; where a state's byte lives follows from the lane bit at the block's end
// (5, 1, 3 for block mod 3 = 0, 1, 2): chain_back8_two.  Decisions, ties and outputs are those of viterbi_fused_kernel<1> (the whole GPU suite runs under
// DABHIP_VIT_TWO_LANES=1, tools/gpu/two_lanes.sh).

namespace two {

__host__ __device__ constexpr int remove_bit(int r, int b) { return ((r >> (b + 1)) << b) | (r & ((1 << b) - 1)); }
__host__ __device__ constexpr int insert_bit(int q, int b, int v) { return ((q >> b) << (b + 1)) | (v << b) | (q & ((1 << b) - 1)); }
__host__ __device__ constexpr int lane_bit(int t) { return (3 + t) % 6; }
// register of state k (either member of its pair) inside its lane, in the layout with pair bit tau (0..4; 4 = parked, pairs (k, k ^ 16)) and lane bit L
__host__ __device__ constexpr int phys_of(int k, int tau, int L)
{
  const int side = k >> 5, r = compress_bit(k & 31, tau);
  if (L == 5) return r;
  return side * 8 + remove_bit(r, L < tau ? L : L - 1);
}
// low member (pair bit clear) of the pair in register P of LANE 0; lane 1's differs in the lane bit only
__host__ __device__ constexpr int state_of(int P, int tau, int L)
{
  if (L == 5) return expand_bit(P, tau);
  return 32 * (P >> 3) + expand_bit(insert_bit(P & 7, L < tau ? L : L - 1, 0), tau);
}

// metric tables: [table][part][row] like MetricLut; tables 0..7 = lane 0 at tag bit s (the lane form's words), 8 + 8 v + s = lane 1 in block variant v
typedef uint4 Lut2[4][32];
constexpr int kTables = 32;
__device__ __forceinline__ void build_lut2(Lut2* lut)
{
  for (int e = threadIdx.x; e < kTables * 4 * 32; e += blockDim.x) {
    const int T = e >> 7, part = (e >> 5) & 3, row = e & 31;
    const bool lane1 = T >= 8;
    const int s = lane1 ? (T - 8) & 7 : T, v = lane1 ? (T - 8) >> 3 : 0, tau = s & 3, L = (3 + 8 * v + s) % 6;
    const int n = row < 16 ? 4 : row < 24 ? 3 : row < 28 ? 2 : row < 30 ? 1 : 0;
    const unsigned val = row < 30 ? static_cast<unsigned>(row) - lut_row_base(n) : 0u, m = (1u << n) - 1u;
    int bm[8];
    branch_metrics_hard(val | (m << 4), bm);
    const unsigned gamma = branch_code3(2u << tau);
    const unsigned g = (lane1 && L < 5) ? branch_code3(2u << L) : 0u;      // the lane's share of the branch code
    const bool swap = lane1 && L == 5;                                      // exchange step: lane 1's own operand is the high predecessor
    const bool tagged = (part < 2) != swap;
    uint32_t w[4];
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
      const unsigned c = (4u * (part & 1) + k) ^ g;
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (unsigned q = 0; q < 8; ++q) {                // bm[] is indexed with run-time values: select, do not index
        lo = (q == c) ? static_cast<uint32_t>(bm[q]) : lo;
        hi = (q == (c ^ gamma)) ? static_cast<uint32_t>(bm[q]) : hi;
      }
      w[k] = ((lo | (hi << 16)) << (8 - kMetricShift)) + (tagged ? (0x00010001u << s) : 0u);
    }
    lut[T][part][row] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __syncthreads();
}

// the partner lane's register + b: one instruction (the data-parallel-primitive operand rides on the add)
__device__ __forceinline__ uint32_t add_partner(uint32_t partner_reg, uint32_t b)
{
  return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(partner_reg), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true)) + b;
}
__device__ __forceinline__ pk16 max_pk(uint32_t a, uint32_t b)
{
  return as_pk(__builtin_bit_cast(uint32_t, __builtin_elementwise_max(as_upk(a), as_upk(b))));
}

template <int kTau, int kL, int kQ>
__device__ __forceinline__ void local_butterfly(const pk16 (&p)[16], pk16 (&n)[16], const uint32_t (&A)[8], const uint32_t (&B)[8])
{
  constexpr int j0 = expand_bit(insert_bit(kQ, kL < kTau ? kL : kL - 1, 0), kTau);
  constexpr unsigned c = branch_code3(2 * j0);
  const uint32_t x = as_u32(p[kQ]), y = as_u32(p[8 + kQ]);
  n[phys_of(2 * j0, kTau + 1, kL + 1)] = max_pk(x + A[c], y + B[c ^ 7]);          // viterbi.c:404-414
  n[phys_of(2 * j0 + 1, kTau + 1, kL + 1)] = max_pk(x + A[c ^ 7], y + B[c]);      // viterbi.c:415-421
}
template <int kTau, int kR>
__device__ __forceinline__ void exchange_butterfly(const pk16 (&p)[16], pk16 (&n)[16], const uint32_t (&A)[8], const uint32_t (&B)[8])
{
  constexpr int j = expand_bit(kR, kTau);
  constexpr unsigned c = branch_code3(2 * j);
  const uint32_t own = as_u32(p[kR]);
  // lane 0: max(x + low[c], y + high[c ^ 7]) = successor 2 j; lane 1 (tables swapped): max(y + high[c], x + low[c ^ 7]) = successor 2 j + 1
  n[phys_of(2 * j, kTau + 1, 0)] = max_pk(own + A[c], add_partner(own, B[c ^ 7]));
}
template <int kTau, int kL, int... kI>
__device__ __forceinline__ void all_butterflies(const pk16 (&p)[16], pk16 (&n)[16], const uint32_t (&A)[8], const uint32_t (&B)[8], std::integer_sequence<int, kI...>)
{
  if constexpr (kL == 5) (exchange_butterfly<kTau, kI>(p, n, A, B), ...);
  else (local_butterfly<kTau, kL, kI>(p, n, A, B), ...);
}

// step kS (0..7) of a block of variant kV: pair bit kS & 3, lane bit lane_bit(8 kV + kS), tag bit kS.  In two halves: the step's 16 metric words from the
// lane's table, and the butterflies -- a lone wave on its SIMD has nobody to hide the LDS round trip behind (measured: every step waited for its
// reads in full), so acs8 fetches step s + 1's words before it computes step s.
struct Words { uint4 a0, a1, b0, b1; };
template <int kV, int kS>
__device__ __forceinline__ Words fetch(unsigned row, const Lut2* lut, unsigned lane1)
{
  const unsigned T = lane1 ? 8u + 8u * kV + kS : static_cast<unsigned>(kS);
  return Words{lut[T][0][row], lut[T][1][row], lut[T][2][row], lut[T][3][row]};
}
template <int kV, int kS>
__device__ __forceinline__ void butterflies(const Words& w, const pk16 (&p)[16], pk16 (&n)[16])
{
  constexpr int tau = kS & 3, L = lane_bit(8 * kV + kS);
  static_assert(L != tau, "the lane bit never sits on the pair bit");
  const uint32_t A[8] = {w.a0.x, w.a0.y, w.a0.z, w.a0.w, w.a1.x, w.a1.y, w.a1.z, w.a1.w};
  const uint32_t B[8] = {w.b0.x, w.b0.y, w.b0.z, w.b0.w, w.b1.x, w.b1.y, w.b1.z, w.b1.w};
  if constexpr (L == 5) all_butterflies<tau, L>(p, n, A, B, std::make_integer_sequence<int, 16>{});
  else all_butterflies<tau, L>(p, n, A, B, std::make_integer_sequence<int, 8>{});
}
template <int kV, int kS>
__device__ __forceinline__ void acs_step(unsigned row, const Lut2* lut, unsigned lane1, const pk16 (&p)[16], pk16 (&n)[16])
{
  butterflies<kV, kS>(fetch<kV, kS>(row, lut, lane1), p, n);
}

// parked pairs (k, k ^ 16) -> pairs (k, k ^ 1), inside each lane (lane bit kL is neither 0 nor 4); kClear: the byte tags go
template <bool kClear, int kL, int kP>
__device__ __forceinline__ void repair_one(const pk16 (&n)[16], pk16 (&p)[16])
{
  constexpr int k = state_of(kP, 0, kL), a = phys_of(k, 4, kL), b = phys_of(k + 1, 4, kL);
  constexpr uint32_t sel = ((k >> 4) & 1) ? (kClear ? 0x070c030cu : 0x07060302u) : (kClear ? 0x050c010cu : 0x05040100u);
  p[kP] = as_pk(__builtin_amdgcn_perm(as_u32(n[b]), as_u32(n[a]), sel));
}
template <bool kClear, int kL, int... kP>
__device__ __forceinline__ void repair_all(const pk16 (&n)[16], pk16 (&p)[16], std::integer_sequence<int, kP...>)
{
  static_assert(kL != 0 && kL != 4, "re-pairing joins states that differ in bits 0 and 4");
  (repair_one<kClear, kL, kP>(n, p), ...);
}
template <bool kClear, int kL>
__device__ __forceinline__ void repair(const pk16 (&n)[16], pk16 (&p)[16]) { repair_all<kClear, kL>(n, p, std::make_integer_sequence<int, 16>{}); }

// a lane's half of the survivor record: register P lands in word P >> 1, byte 2 (P & 1) + half (survivor_record8's order)
__device__ __forceinline__ void record(const pk16 (&n)[16], uint4* rec)
{
  uint32_t d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = __builtin_amdgcn_perm(as_u32(n[2 * i + 1]), as_u32(n[2 * i]), 0x06040200u);
  rec_store(rec, d[0], d[1], d[2], d[3]);
  rec_store(rec + 64, d[4], d[5], d[6], d[7]);
}

template <int kV>
__device__ __forceinline__ void acs8(uint32_t rows0, uint32_t rows1, const Lut2* lut, unsigned lane1, pk16 (&pm)[16], pk16 (&pn)[16], pk16 (&pl4)[16], uint4* rec)
{
  Words w0 = fetch<kV, 0>(rows0 & 0xff, lut, lane1);
  Words w1 = fetch<kV, 1>((rows0 >> 8) & 0xff, lut, lane1);
  butterflies<kV, 0>(w0, pm, pn);
  w0 = fetch<kV, 2>((rows0 >> 16) & 0xff, lut, lane1);
  butterflies<kV, 1>(w1, pn, pm);
  w1 = fetch<kV, 3>(rows0 >> 24, lut, lane1);
  butterflies<kV, 2>(w0, pm, pn);
  w0 = fetch<kV, 4>(rows1 & 0xff, lut, lane1);
  butterflies<kV, 3>(w1, pn, pl4);
  w1 = fetch<kV, 5>((rows1 >> 8) & 0xff, lut, lane1);
  repair<false, lane_bit(8 * kV + 4)>(pl4, pm);
  butterflies<kV, 4>(w0, pm, pn);
  w0 = fetch<kV, 6>((rows1 >> 16) & 0xff, lut, lane1);
  butterflies<kV, 5>(w1, pn, pm);
  w1 = fetch<kV, 7>(rows1 >> 24, lut, lane1);
  butterflies<kV, 6>(w0, pm, pn);
  butterflies<kV, 7>(w1, pn, pl4);
  record(pl4, rec);
  repair<true, lane_bit(8 * kV + 8)>(pl4, pm);
}
// the last r = 1..7 steps of a code word: the record is only read for state 0 (lane 0, low half of register 0 in every layout)
template <int kV>
__device__ __forceinline__ void acs8_tail(uint32_t rows0, uint32_t rows1, int r, const Lut2* lut, unsigned lane1, pk16 (&pm)[16], pk16 (&pn)[16], pk16 (&pl4)[16],
                                          uint4* rec)
{
  acs_step<kV, 0>(rows0 & 0xff, lut, lane1, pm, pn);
  if (r == 1) { record(pn, rec); return; }
  acs_step<kV, 1>((rows0 >> 8) & 0xff, lut, lane1, pn, pm);
  if (r == 2) { record(pm, rec); return; }
  acs_step<kV, 2>((rows0 >> 16) & 0xff, lut, lane1, pm, pn);
  if (r == 3) { record(pn, rec); return; }
  acs_step<kV, 3>(rows0 >> 24, lut, lane1, pn, pl4);
  repair<false, lane_bit(8 * kV + 4)>(pl4, pm);
  if (r == 4) { record(pm, rec); return; }
  acs_step<kV, 4>(rows1 & 0xff, lut, lane1, pm, pn);
  if (r == 5) { record(pn, rec); return; }
  acs_step<kV, 5>((rows1 >> 8) & 0xff, lut, lane1, pn, pm);
  if (r == 6) { record(pm, rec); return; }
  acs_step<kV, 6>((rows1 >> 16) & 0xff, lut, lane1, pm, pn);
  record(pn, rec);
}

// re-base every 256 steps (MetricScale<8>): state 0 -- lane 0, register 0, low half -- back to kBase, in both lanes
__device__ __forceinline__ void rebase(pk16 (&pm)[16])
{
  const uint32_t r0 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(as_u32(pm[0])), 0xA0 /* quad_perm [0,0,2,2]: the even lane's */, 0xf, 0xf, true));
  const uint32_t s0 = (r0 & 0xffffu) - MetricScale<8>::kBase;
  const uint32_t base = s0 | (s0 << 16);
#pragma unroll
  for (int r = 0; r < 16; ++r) pm[r] = as_pk(as_u32(pm[r]) - base);
}

// chain back over the two lanes' records: block b of 8 steps at pair_rec[256 b + 64 j + lane], j = 0, 1; pair_rec = the code word's EVEN lane's record base
__device__ __forceinline__ void chain_back8_two(const uint4* pair_rec, int nsteps, const uint32_t* __restrict__ prbs_words, uint32_t* dst)
{
  unsigned state = 0;
  uint32_t acc = 0;
  auto consume = [&](unsigned tags, int t0, int k_hi) {    // steps t0 + k_hi .. t0, newest first (chain_back8's)
#pragma unroll
    for (int k = 7; k >= 0; --k) {
      const int t = t0 + k;
      if (k <= k_hi && t >= 6) {
        const unsigned bit = ((tags >> k) & 1u) ^ 1u;
        state = (state | (bit << 6)) >> 1;
        const int i = t - 6;
        acc |= bit << (8 * ((i >> 3) & 3) + (7 - (i & 7)));
        if ((i & 31) == 0) {
          dst[i >> 5] = acc ^ prbs_words[i >> 5];
          acc = 0;
        }
      }
    }
  };
  const int nfull = nsteps >> 3, r = nsteps & 7;
  if (r) consume(pair_rec[static_cast<size_t>(nfull) * 256].x & 255u, 8 * nfull, r - 1);
  int b3 = (nfull - 1) % 3;                                 // lane bit at the end of block b: 5, 1, 3 for b mod 3 = 0, 1, 2
  // Which byte of a block's record is wanted is known only when the block after it has been walked -- one memory round trip per block if only that
  // byte's word is fetched (chain_back8: right for a full device, where the stage is bound by the records' traffic).  Here the device is far from full and the
  // round trips ARE the chain-back (577 of them for the longest code words, ~0.7 us each): the WHOLE records of four blocks (4 x 64 bytes per code word,
  // both lanes' halves) are fetched at once and the bytes picked from registers -- one round trip per 32 steps, 15 selects per block.
  for (int top = nfull - 1; top >= 0; top -= 4) {
    uint4 q[4][4];                                          // [block top - k][lane * 2 + j]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint4* at = pair_rec + static_cast<size_t>(max(top - k, 0)) * 256;
      q[k][0] = rec_load(at);
      q[k][1] = rec_load(at + 64);
      q[k][2] = rec_load(at + 1);
      q[k][3] = rec_load(at + 65);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int b = top - k;
      if (b < 0) break;
      const unsigned side = state >> 5, r4 = state & 15u, half = (state >> 4) & 1u;
      unsigned lane, P;
      if (b3 == 0) { lane = side; P = r4; }
      else {
        const unsigned Le = b3 == 1 ? 1u : 3u;
        lane = (state >> Le) & 1u;
        P = side * 8u + (((r4 >> (Le + 1u)) << Le) | (r4 & ((1u << Le) - 1u)));
      }
      const unsigned idx = lane * 8u + (P >> 1), byte = 2u * (P & 1u) + half;       // dword idx of the block's 16: [lane][j][word]
      uint32_t d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t lo = (idx & 1u) ? in_vgpr(q[k][u].y) : in_vgpr(q[k][u].x), hi = (idx & 1u) ? in_vgpr(q[k][u].w) : in_vgpr(q[k][u].z);
        d[u] = (idx & 2u) ? hi : lo;
      }
      const uint32_t w = (idx & 8u) ? ((idx & 4u) ? d[3] : d[2]) : ((idx & 4u) ? d[1] : d[0]);
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
        consume(tags, 0, 7);
      }
      b3 = b3 == 0 ? 2 : b3 - 1;
    }
  }
}

}  // namespace two

// the fused decoder (viterbi_fused_kernel<1>'s load, de-puncturing and output) with two lanes per code word: two waves per group of 64 code words
__global__ __launch_bounds__(256, 2) void viterbi_fused_two_kernel(const WaveGroup* __restrict__ groups, int ngroups, const int* __restrict__ job_ids,
                                                                  const CodewordPlan* __restrict__ plans, const uint32_t* __restrict__ grouped, int row_words,
                                                                  uint2* __restrict__ decisions, const uint32_t* __restrict__ prbs_words,
                                                                  uint8_t* __restrict__ out, int record_stride)
{
  __shared__ __attribute__((aligned(16))) two::Lut2 lut[two::kTables];
  two::build_lut2(lut);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(4 * blockIdx.x + (threadIdx.x >> 6));
  const int g = w >> 1, half = w & 1;
  if (g >= ngroups) return;
  const WaveGroup grp = groups[g];
  const CodewordPlan* plan = plans + grp.plan;             // (read field by field where it is used: a local copy indexed by `seg` ends up in LDS)
  const int nsteps = grp.nsteps;
  const unsigned lane1 = lane & 1;
  const int cw = 32 * half + (lane >> 1);                  // this lane's code word within the group
  uint4* my_rec = reinterpret_cast<uint4*>(decisions + grp.dec_base * 64) + 128 * half + lane;

  const int word0 = min(plan->start_bit >> 5, row_words - 1);
  const uint32_t* src = grouped + (static_cast<size_t>(grp.first >> 6) * row_words + word0) * 64 + cw;
  const int last_word = row_words - 1 - word0;
  uint64_t fifo = 0;
  int have = 0;
  uint32_t nextw = src[0];
  int widx = 1;
  auto refill = [&]() {
    fifo |= static_cast<uint64_t>(nextw) << have;
    have += 32;
    nextw = src[static_cast<size_t>(min(widx, last_word)) * 64];
    ++widx;
  };

  pk16 pm[16], pn[16], pl4[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) pm[r] = as_pk(0u);
  pm[0] = as_pk(lane1 ? 0u : MetricScale<8>::kBase);        // state 0: lane 0, register 0, low half (two::phys_of(0, 0, 3) = 0)
  int t = 0, v = 0;                                         // v = block index mod 3: the block's variant
  for (int seg = 0; seg < 5; ++seg) {
    const uint32_t mask = seg < 4 ? plan->mask[seg] : (puncture_mask(8) & 0x00ffffffu);
    const int units = seg < 4 ? 4 * plan->blocks[seg] : 1;
    const int need = __popc(mask);
    uint32_t counts = 0;
    for (int q = 0; q < 8; ++q) counts |= static_cast<uint32_t>(__popc((mask >> (4 * q)) & 15u)) << (3 * q);
    for (int u = 0; u < units; ++u) {
      if (have < need) refill();
      uint32_t ww[2] = {0, 0};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int n = (counts >> (3 * q)) & 7;
        const uint32_t m = (1u << n) - 1u;
        const uint32_t row = (static_cast<uint32_t>(fifo) & m) + lut_row_base(n);
        fifo >>= n;
        ww[q >> 2] |= row << (8 * (q & 3));
      }
      have -= need;
      uint4* rec = my_rec + static_cast<size_t>(t >> 3) * 256;
      if (t + 8 <= nsteps) {
        if (v == 0) two::acs8<0>(ww[0], ww[1], lut, lane1, pm, pn, pl4, rec);
        else if (v == 1) two::acs8<1>(ww[0], ww[1], lut, lane1, pm, pn, pl4, rec);
        else two::acs8<2>(ww[0], ww[1], lut, lane1, pm, pn, pl4, rec);
      } else if (t < nsteps) {
        if (v == 0) two::acs8_tail<0>(ww[0], ww[1], nsteps - t, lut, lane1, pm, pn, pl4, rec);
        else if (v == 1) two::acs8_tail<1>(ww[0], ww[1], nsteps - t, lut, lane1, pm, pn, pl4, rec);
        else two::acs8_tail<2>(ww[0], ww[1], nsteps - t, lut, lane1, pm, pn, pl4, rec);
      }
      t += 8;
      v = v == 2 ? 0 : v + 1;
      if ((t & (MetricScale<8>::kRebaseSteps - 1)) == 0 && t < nsteps) two::rebase(pm);
    }
  }
  if (lane1 == 0 && cw < grp.count) {
    const int record = job_ids ? job_ids[grp.first + cw] : grp.first + cw;
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + static_cast<size_t>(record) * record_stride + plan->out_offset);
    two::chain_back8_two(my_rec, nsteps, prbs_words, dst);
  }
}
