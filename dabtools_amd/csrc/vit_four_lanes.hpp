// vit_four_lanes.hpp — the hard-decision decoder with 2^NL lanes per code word (NL = 2: four lanes; NL = 1: two lanes without tables of their own).
//
// Included by k_decode.hip inside its anonymous namespace, after vit_two_lanes.hpp (whose header explains the rotating lane bit).  What is new here, all of it
// run on the CPU first (tools/models/multilane_model.py, held by the CPU suite):
//   * NL lane bits rotate, at places (3 + t) mod 6 and (5 + t) mod 6: never on the pair bit t mod 4, never on each other, on odd places at the re-pairing
//     points.  A lane holds the 2^(5 - NL) registers of the states whose bits at the lane places equal its lane id (id bit i = lane bit i); registers are
//     numbered by COMPACTION -- the state's bits in place order with the pair bit and the lane bits taken out.  A step is local (half of the registers are
//     low predecessors, half high ones: bit 5 is the top remaining bit) unless one lane bit sits at place 5; then the lanes that differ in THAT bit exchange
//     through quad_perm [1,0,3,2] (id bit 0) or [2,3,0,1] (id bit 1), same-index registers, no selects.
//   * no per-lane metric tables: a lane's branch codes are code(2 j) ^ g, g = XOR of code(2 << L_i) over its set lane bits below place 5, and the metric of
//     code c ^ g for the received nibble v is the metric of code c for v ^ cw(g) (the 4-bit code word is linear in c): the lane reads the lane form's ONE
//     16 KB table at the row of (v ^ cw(g)) & mask -- one v_xor per step with a per-lane constant, and rows of one step are distinct banks or the same
//     address, so the reads stay conflict-free.  At an exchange over bit i the lanes with that bit set own the HIGH predecessor: they read the tagged and the
//     untagged parts swapped (address ^ 1024).
// 24 add / max instructions per step and lane at NL = 2 (two lanes: 48, one: 96); records: one 16-byte store per lane and block of 8 steps.

namespace multi {

__host__ __device__ constexpr int place(int i, int t) { return ((i == 0 ? 3 : 5) + t) % 6; }
template <int NL>
__host__ __device__ constexpr unsigned taken(int tau, int t)          // the places that do not number registers: pair bit and lane bits, as a mask
{
  unsigned m = 1u << tau;
  for (int i = 0; i < NL; ++i) m |= 1u << place(i, t);
  return m;
}
__host__ __device__ constexpr int compact(int k, unsigned removed)
{
  int out = 0, pos = 0;
  for (int b = 0; b < 6; ++b)
    if (!((removed >> b) & 1u)) out |= ((k >> b) & 1) << pos++;
  return out;
}
__host__ __device__ constexpr int expand(int P, unsigned removed)
{
  int out = 0, pos = 0;
  for (int b = 0; b < 6; ++b)
    if (!((removed >> b) & 1u)) out |= ((P >> pos++) & 1) << b;
  return out;
}
template <int NL>
__host__ __device__ constexpr int at_five(int t)                       // which lane bit sits at place 5 before step t (-1: none, the step is local)
{
  for (int i = 0; i < NL; ++i)
    if (place(i, t) == 5) return i;
  return -1;
}
__host__ __device__ constexpr unsigned cw4(unsigned c) { return c | ((c & 1u) << 3); }

template <int kCtrl>
__device__ __forceinline__ uint32_t add_partner(uint32_t partner_reg, uint32_t b)
{
  return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(partner_reg), kCtrl, 0xf, 0xf, true)) + b;
}

template <int NL>
struct Lane {
  static constexpr int kRegs = 1 << (5 - NL);
  uint32_t xm[6];          // cw(g) of this lane at step t, by t mod 6
  uint32_t swap[NL];       // 1024 for a lane whose id bit i is set (exchange over bit i: parts swapped), else 0
};
template <int NL>
__device__ __forceinline__ Lane<NL> make_lane(unsigned id)
{
  Lane<NL> ln;
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int L = place(i, u);
      if (L < 5) x ^= ((id >> i) & 1u) ? cw4(branch_code3(2u << L)) : 0u;
    }
    ln.xm[u] = x;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) ln.swap[i] = ((id >> i) & 1u) ? 1024u : 0u;
  return ln;
}

struct Words { uint4 a0, a1, b0, b1; };
// the step's 16 metric words: row of ((value ^ lane's code word) & mask) in the table of tag bit kS; lut_bytes = LDS address of the table (2048-aligned)
template <int NL, int kV, int kS>
__device__ __forceinline__ Words fetch(uint32_t value, uint32_t mask, uint32_t row_base, const unsigned char* lut_bytes, const Lane<NL>& ln)
{
  constexpr int t = 8 * kV + kS, i5 = at_five<NL>(t);
  const uint32_t row = ((value ^ ln.xm[t % 6]) & mask) + row_base;
  const unsigned char* a = lut_bytes + kS * 2048 + row * 16;
  if constexpr (i5 >= 0) {
    const unsigned char* own = a + ln.swap[i5];                       // parts 0, 1 (tagged) -- or 2, 3 for the lanes that own the high predecessor
    const unsigned char* other = a + (ln.swap[i5] ^ 1024u);
    return Words{*reinterpret_cast<const uint4*>(own), *reinterpret_cast<const uint4*>(own + 512), *reinterpret_cast<const uint4*>(other),
                 *reinterpret_cast<const uint4*>(other + 512)};
  } else {
    return Words{*reinterpret_cast<const uint4*>(a), *reinterpret_cast<const uint4*>(a + 512), *reinterpret_cast<const uint4*>(a + 1024),
                 *reinterpret_cast<const uint4*>(a + 1536)};
  }
}

template <int NL, int kV, int kS, int kQ>
__device__ __forceinline__ void one_butterfly(const pk16 (&p)[Lane<NL>::kRegs], pk16 (&n)[Lane<NL>::kRegs], const uint32_t (&A)[8], const uint32_t (&B)[8])
{
  constexpr int t = 8 * kV + kS, tau = kS & 3, i5 = at_five<NL>(t), half = Lane<NL>::kRegs / 2;
  constexpr unsigned in = taken<NL>(tau, t), out = taken<NL>(tau + 1, t + 1);
  constexpr int j = expand(kQ, in);                                   // the lane-0 state of register kQ (low member of its pair)
  constexpr unsigned c = branch_code3(2 * j);
  if constexpr (i5 >= 0) {
    const uint32_t own = as_u32(p[kQ]);
    n[compact(2 * j, out)] = two::max_pk(own + A[c], add_partner<i5 == 0 ? 0xB1 : 0x4E>(own, B[c ^ 7]));
  } else {
    static_assert(j < 32 && expand(kQ + half, in) == j + 32, "bit 5 is the top remaining bit: the upper half of the registers are the high predecessors");
    const uint32_t x = as_u32(p[kQ]), y = as_u32(p[kQ + half]);
    n[compact(2 * j, out)] = two::max_pk(x + A[c], y + B[c ^ 7]);
    n[compact(2 * j + 1, out)] = two::max_pk(x + A[c ^ 7], y + B[c]);
  }
}
template <int NL, int kV, int kS, int... kQ>
__device__ __forceinline__ void some_butterflies(const Words& w, const pk16 (&p)[Lane<NL>::kRegs], pk16 (&n)[Lane<NL>::kRegs], std::integer_sequence<int, kQ...>)
{
  const uint32_t A[8] = {w.a0.x, w.a0.y, w.a0.z, w.a0.w, w.a1.x, w.a1.y, w.a1.z, w.a1.w};
  const uint32_t B[8] = {w.b0.x, w.b0.y, w.b0.z, w.b0.w, w.b1.x, w.b1.y, w.b1.z, w.b1.w};
  (one_butterfly<NL, kV, kS, kQ>(p, n, A, B), ...);
}
template <int NL, int kV, int kS>
__device__ __forceinline__ void butterflies(const Words& w, const pk16 (&p)[Lane<NL>::kRegs], pk16 (&n)[Lane<NL>::kRegs])
{
  constexpr int count = at_five<NL>(8 * kV + kS) >= 0 ? Lane<NL>::kRegs : Lane<NL>::kRegs / 2;
  some_butterflies<NL, kV, kS>(w, p, n, std::make_integer_sequence<int, count>{});
}

// parked pairs (k, k ^ 16) -> pairs (k, k ^ 1) inside each lane, in the layout before step kT (a multiple of 4: the lane bits sit on odd places)
template <int NL, bool kClear, int kT, int kP>
__device__ __forceinline__ void repair_one(const pk16 (&n)[Lane<NL>::kRegs], pk16 (&p)[Lane<NL>::kRegs])
{
  constexpr unsigned parked = taken<NL>(4, kT), fresh = taken<NL>(0, kT);
  constexpr int k = expand(kP, fresh), a = compact(k, parked), b = compact(k + 1, parked);
  constexpr uint32_t sel = ((k >> 4) & 1) ? (kClear ? 0x070c030cu : 0x07060302u) : (kClear ? 0x050c010cu : 0x05040100u);
  p[kP] = as_pk(__builtin_amdgcn_perm(as_u32(n[b]), as_u32(n[a]), sel));
}
template <int NL, bool kClear, int kT, int... kP>
__device__ __forceinline__ void repair_all(const pk16 (&n)[Lane<NL>::kRegs], pk16 (&p)[Lane<NL>::kRegs], std::integer_sequence<int, kP...>)
{
  (repair_one<NL, kClear, kT, kP>(n, p), ...);
}
template <int NL, bool kClear, int kT>
__device__ __forceinline__ void repair(const pk16 (&n)[Lane<NL>::kRegs], pk16 (&p)[Lane<NL>::kRegs])
{
  repair_all<NL, kClear, kT>(n, p, std::make_integer_sequence<int, Lane<NL>::kRegs>{});
}

// a lane's part of the survivor record: register P -> word P >> 1, byte 2 (P & 1) + half; kRegs / 8 16-byte stores, 64 lanes apart
template <int NL>
__device__ __forceinline__ void record(const pk16 (&n)[Lane<NL>::kRegs], uint4* rec)
{
  constexpr int words = Lane<NL>::kRegs / 2;
  uint32_t d[words];
#pragma unroll
  for (int i = 0; i < words; ++i) d[i] = __builtin_amdgcn_perm(as_u32(n[2 * i + 1]), as_u32(n[2 * i]), 0x06040200u);
#pragma unroll
  for (int j = 0; j < words / 4; ++j) rec_store(rec + 64 * j, d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3]);
}

// values: the received bits of the block's eight steps, one byte per step (the step's n low bits); counts: n per step, 3 bits each (wave-uniform)
template <int NL, int kV, int kS>
__device__ __forceinline__ Words fetch_step(uint32_t values0, uint32_t values1, uint32_t counts, const unsigned char* lut_bytes, const Lane<NL>& ln)
{
  const uint32_t value = ((kS < 4 ? values0 : values1) >> (8 * (kS & 3))) & 0xffu;
  const int n = (counts >> (3 * kS)) & 7;
  return fetch<NL, kV, kS>(value, (1u << n) - 1u, lut_row_base(n), lut_bytes, ln);
}

// eight steps; the block's record and the re-pairing that clears the tags follow in the kernel (with the input top-up between them and the steps)
template <int NL, int kV>
__device__ __forceinline__ void acs8(uint32_t values0, uint32_t values1, uint32_t counts, const unsigned char* lut_bytes, const Lane<NL>& ln,
                                     pk16 (&pm)[Lane<NL>::kRegs], pk16 (&pn)[Lane<NL>::kRegs], pk16 (&pl4)[Lane<NL>::kRegs])
{
  // the table words of a step are fetched two steps ahead (at four lanes a step's butterflies, 24 instructions, are shorter than an LDS round trip;
  // measured the same as one step ahead)
  Words w0 = fetch_step<NL, kV, 0>(values0, values1, counts, lut_bytes, ln);
  Words w1 = fetch_step<NL, kV, 1>(values0, values1, counts, lut_bytes, ln);
  Words w2 = fetch_step<NL, kV, 2>(values0, values1, counts, lut_bytes, ln);
  butterflies<NL, kV, 0>(w0, pm, pn);
  w0 = fetch_step<NL, kV, 3>(values0, values1, counts, lut_bytes, ln);
  butterflies<NL, kV, 1>(w1, pn, pm);
  w1 = fetch_step<NL, kV, 4>(values0, values1, counts, lut_bytes, ln);
  butterflies<NL, kV, 2>(w2, pm, pn);
  w2 = fetch_step<NL, kV, 5>(values0, values1, counts, lut_bytes, ln);
  butterflies<NL, kV, 3>(w0, pn, pl4);
  w0 = fetch_step<NL, kV, 6>(values0, values1, counts, lut_bytes, ln);
  repair<NL, false, 8 * kV + 4>(pl4, pm);
  butterflies<NL, kV, 4>(w1, pm, pn);
  w1 = fetch_step<NL, kV, 7>(values0, values1, counts, lut_bytes, ln);
  butterflies<NL, kV, 5>(w2, pn, pm);
  butterflies<NL, kV, 6>(w0, pm, pn);
  butterflies<NL, kV, 7>(w1, pn, pl4);
}
// the last r = 1..7 steps of a code word: the record is only read for state 0 (lane 0, low half of register 0 in every layout)
template <int NL, int kV>
__device__ __forceinline__ void acs8_tail(uint32_t values0, uint32_t values1, uint32_t counts, int r, const unsigned char* lut_bytes, const Lane<NL>& ln,
                                          pk16 (&pm)[Lane<NL>::kRegs], pk16 (&pn)[Lane<NL>::kRegs], pk16 (&pl4)[Lane<NL>::kRegs], uint4* rec)
{
  butterflies<NL, kV, 0>(fetch_step<NL, kV, 0>(values0, values1, counts, lut_bytes, ln), pm, pn);
  if (r == 1) { record<NL>(pn, rec); return; }
  butterflies<NL, kV, 1>(fetch_step<NL, kV, 1>(values0, values1, counts, lut_bytes, ln), pn, pm);
  if (r == 2) { record<NL>(pm, rec); return; }
  butterflies<NL, kV, 2>(fetch_step<NL, kV, 2>(values0, values1, counts, lut_bytes, ln), pm, pn);
  if (r == 3) { record<NL>(pn, rec); return; }
  butterflies<NL, kV, 3>(fetch_step<NL, kV, 3>(values0, values1, counts, lut_bytes, ln), pn, pl4);
  repair<NL, false, 8 * kV + 4>(pl4, pm);
  if (r == 4) { record<NL>(pm, rec); return; }
  butterflies<NL, kV, 4>(fetch_step<NL, kV, 4>(values0, values1, counts, lut_bytes, ln), pm, pn);
  if (r == 5) { record<NL>(pn, rec); return; }
  butterflies<NL, kV, 5>(fetch_step<NL, kV, 5>(values0, values1, counts, lut_bytes, ln), pn, pm);
  if (r == 6) { record<NL>(pm, rec); return; }
  butterflies<NL, kV, 6>(fetch_step<NL, kV, 6>(values0, values1, counts, lut_bytes, ln), pm, pn);
  record<NL>(pn, rec);
}

// re-base (MetricScale<8>): state 0 -- lane 0 of the code word, register 0, low half -- back to kBase in all of the code word's lanes
template <int NL>
__device__ __forceinline__ void rebase(pk16 (&pm)[Lane<NL>::kRegs])
{
  const uint32_t r0 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(as_u32(pm[0])), NL == 1 ? 0xA0 /* [0,0,2,2] */ : 0x00 /* [0,0,0,0] */, 0xf, 0xf, true));
  const uint32_t s0 = (r0 & 0xffffu) - MetricScale<8>::kBase;
  const uint32_t base = s0 | (s0 << 16);
#pragma unroll
  for (int r = 0; r < Lane<NL>::kRegs; ++r) pm[r] = as_pk(as_u32(pm[r]) - base);
}

// chain back over the lanes' records (two::chain_back8_two's scheme: the whole records of four blocks per memory round trip).  cw_rec = the record base of
// the code word's lane 0; lane l's part of block b at cw_rec[256 b + 64 j + l].
template <int NL>
__device__ __forceinline__ void chain_back(const uint4* cw_rec, int nsteps, const uint32_t* __restrict__ prbs_words, uint32_t* dst)
{
  constexpr int kLanes = 1 << NL, kVec = Lane<NL>::kRegs / 8;          // 16-byte stores per lane and block; kLanes * kVec = 4
  unsigned state = 0;
  uint32_t acc = 0;
  auto consume = [&](unsigned tags, int t0, int k_hi) {
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
  if (r) consume(cw_rec[static_cast<size_t>(nfull) * 256].x & 255u, 8 * nfull, r - 1);
  int b3 = (nfull - 1) % 3;                                 // the lane places at the end of block b = those before step 8 (b + 1): by b mod 3
  for (int top = nfull - 1; top >= 0; top -= 4) {
    uint4 q[4][4];                                          // [block top - k][lane * kVec + j]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint4* at = cw_rec + static_cast<size_t>(max(top - k, 0)) * 256;
#pragma unroll
      for (int l = 0; l < kLanes; ++l)
#pragma unroll
        for (int j = 0; j < kVec; ++j) q[k][l * kVec + j] = rec_load(at + 64 * j + l);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int b = top - k;
      if (b < 0) break;
      // places of the lane bits: (3 + 8 (b + 1)) mod 6 = 5, 1, 3 and (5 + 8 (b + 1)) mod 6 = 1, 3, 5 for b mod 3 = 0, 1, 2
      const unsigned L0 = b3 == 0 ? 5u : b3 == 1 ? 1u : 3u, L1 = b3 == 0 ? 1u : b3 == 1 ? 3u : 5u;
      unsigned removed = (1u << 4) | (1u << L0), lane = (state >> L0) & 1u;
      if (NL == 2) {
        removed |= 1u << L1;
        lane |= ((state >> L1) & 1u) << 1;
      }
      unsigned P = 0, pos = 0;
#pragma unroll
      for (unsigned bit = 0; bit < 6; ++bit) {
        const unsigned keep = ((removed >> bit) & 1u) ^ 1u;
        P |= (((state >> bit) & 1u) & keep) << pos;
        pos += keep;
      }
      const unsigned half = (state >> 4) & 1u, idx = lane * (Lane<NL>::kRegs / 2) + (P >> 1), byte = 2u * (P & 1u) + half;
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

}  // namespace multi

// the fused decoder (viterbi_fused_kernel<1>'s load, de-puncturing and output) with 2^NL lanes per code word: 2^NL waves per group of 64 code words
template <int NL>
__global__ __launch_bounds__(256, NL == 2 ? 4 : 2) void viterbi_fused_lanes_kernel(const WaveGroup* __restrict__ groups, int ngroups, const int* __restrict__ job_ids,
                                                                                   const CodewordPlan* __restrict__ plans, const uint32_t* __restrict__ grouped,
                                                                                   int row_words, uint2* __restrict__ decisions,
                                                                                   const uint32_t* __restrict__ prbs_words, uint8_t* __restrict__ out,
                                                                                   int record_stride)
{
  constexpr int kRegs = multi::Lane<NL>::kRegs, kLanes = 1 << NL, kVec = kRegs / 8;
  __shared__ __attribute__((aligned(2048))) unsigned char lut_raw[8 * sizeof(MetricLut)];
  build_metric_lut(reinterpret_cast<MetricLut*>(lut_raw));
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(4 * blockIdx.x + (threadIdx.x >> 6));
  const int g = w >> NL, part = w & (kLanes - 1);
  if (g >= ngroups) return;
  const WaveGroup grp = groups[g];
  const CodewordPlan* plan = plans + grp.plan;
  const int nsteps = grp.nsteps;
  const unsigned id = lane & (kLanes - 1);
  const multi::Lane<NL> ln = multi::make_lane<NL>(id);
  const int cw = (64 >> NL) * part + (lane >> NL);           // this lane's code word within the group
  uint4* my_rec = reinterpret_cast<uint4*>(decisions + grp.dec_base * 64) + 64 * kVec * part + lane;

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

  refill();                                                  // 32 bits: a block takes at most that; topped up to more than 32 after every block
  pk16 pm[kRegs], pn[kRegs], pl4[kRegs];
#pragma unroll
  for (int r = 0; r < kRegs; ++r) pm[r] = as_pk(0u);
  pm[0] = as_pk(id ? 0u : MetricScale<8>::kBase);            // state 0: lane 0, register 0, low half
  int t = 0, v = 0;                                         // v = block index mod 3: the block's variant
  for (int seg = 0; seg < 5; ++seg) {
    const uint32_t mask = seg < 4 ? plan->mask[seg] : (puncture_mask(8) & 0x00ffffffu);
    const int units = seg < 4 ? 4 * plan->blocks[seg] : 1;
    const int need = __popc(mask);
    uint32_t counts = 0;
    for (int q = 0; q < 8; ++q) counts |= static_cast<uint32_t>(__popc((mask >> (4 * q)) & 15u)) << (3 * q);
    for (int u = 0; u < units; ++u) {
      uint32_t vals[2] = {0, 0};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int n = (counts >> (3 * q)) & 7;
        vals[q >> 2] |= (static_cast<uint32_t>(fifo) & ((1u << n) - 1u)) << (8 * (q & 3));
        fifo >>= n;
      }
      have -= need;
      uint4* rec = my_rec + static_cast<size_t>(t >> 3) * 256;
      if (t + 8 <= nsteps) {
        if (v == 0) multi::acs8<NL, 0>(vals[0], vals[1], counts, lut_raw, ln, pm, pn, pl4);
        else if (v == 1) multi::acs8<NL, 1>(vals[0], vals[1], counts, lut_raw, ln, pm, pn, pl4);
        else multi::acs8<NL, 2>(vals[0], vals[1], counts, lut_raw, ln, pm, pn, pl4);
        // the input is topped up here, before the block's record goes out, so that its wait (loads and stores retire in order on one counter) never
        // includes that store's round trip -- measured against the lane form's place for it, the head of the block: no difference at these sizes
        if (have <= 32) refill();
        multi::record<NL>(pl4, rec);
        if (v == 0) multi::repair<NL, true, 8>(pl4, pm);
        else if (v == 1) multi::repair<NL, true, 16>(pl4, pm);
        else multi::repair<NL, true, 24>(pl4, pm);
      } else if (t < nsteps) {
        if (v == 0) multi::acs8_tail<NL, 0>(vals[0], vals[1], counts, nsteps - t, lut_raw, ln, pm, pn, pl4, rec);
        else if (v == 1) multi::acs8_tail<NL, 1>(vals[0], vals[1], counts, nsteps - t, lut_raw, ln, pm, pn, pl4, rec);
        else multi::acs8_tail<NL, 2>(vals[0], vals[1], counts, nsteps - t, lut_raw, ln, pm, pn, pl4, rec);
      }
      t += 8;
      v = v == 2 ? 0 : v + 1;
      if ((t & (MetricScale<8>::kRebaseSteps - 1)) == 0 && t < nsteps) multi::rebase<NL>(pm);
    }
  }
  if (id == 0 && cw < grp.count) {
    const int record = job_ids ? job_ids[grp.first + cw] : grp.first + cw;
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + static_cast<size_t>(record) * record_stride + plan->out_offset);
    multi::chain_back<NL>(my_rec, nsteps, prbs_words, dst);
  }
}
