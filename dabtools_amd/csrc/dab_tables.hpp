// dab_tables.hpp — ETSI EN 300 401 constants used by the MI355X dab2eti back end.
//
// Replaces the literal arrays of the reference (dab_tables.c:16-127,164-357 and
// sdr_prstab.c:1) with the standard's generating rules where one exists; the UEP profile
// rows are ETSI Table 7 / Table 36 data.  Shared by host code and uploaded to the device
// by Engine (engine.cpp).  Checked against the reference's arrays (tests/golden/tables.npz) through
// dabhip_host_table by tests/test_host.py::test_product_tables_match_reference_arrays.
#pragma once

#include <array>
#include <cstdint>

namespace dabhip {

constexpr int kTfSamples = 196608;           // one Mode-I transmission frame at 2.048 Msps
constexpr int kTfBytes = kTfSamples * 2;     // cu8: I at even byte, Q at odd byte
constexpr int kNullSamples = 2656;
constexpr int kSymSamples = 2552;            // 504 cyclic prefix + 2048
constexpr int kCpSamples = 504;
constexpr int kSymbolsPerTf = 76;            // PRS + 75 data symbols
constexpr int kCarriers = 1536;
constexpr int kBitsPerSym = 3072;
constexpr int kFicBits = 3 * kBitsPerSym;    // 9216
constexpr int kMscBits = 72 * kBitsPerSym;   // 221184
constexpr int kCifBits = 18 * kBitsPerSym;   // 55296 = 864 CU x 64
constexpr int kEtiBytes = 6144;
constexpr int kChunkBytes = 262144;          // librtlsdr buffer size used by dab2eti.c:238

struct UepProfile {            // ETSI Table 7 + Table 36; pi holds PI (1..24), 0 = unused
  int bitrate, size_cu, protlevel;
  int l[4];
  int pi[4];
};

inline const UepProfile* uep_table()
{
  static const UepProfile rows[64] = {
  { 32,  16, 5, { 3,  4,  17, 0}, { 5,  3,  2,  0}},
  { 32,  21, 4, { 3,  3,  18, 0}, {11,  6,  5,  0}},
  { 32,  24, 3, { 3,  4,  14, 3}, {15,  9,  6,  8}},
  { 32,  29, 2, { 3,  4,  14, 3}, {22, 13,  8, 13}},
  { 32,  35, 1, { 3,  5,  13, 3}, {24, 17, 12, 17}},
  { 48,  24, 5, { 4,  3,  26, 3}, { 5,  4,  2,  3}},
  { 48,  29, 4, { 3,  4,  26, 3}, { 9,  6,  4,  6}},
  { 48,  35, 3, { 3,  4,  26, 3}, {15, 10,  6,  9}},
  { 48,  42, 2, { 3,  4,  26, 3}, {24, 14,  8, 15}},
  { 48,  52, 1, { 3,  5,  25, 3}, {24, 18, 13, 18}},
  { 56,  29, 5, { 6, 10,  23, 3}, { 5,  4,  2,  3}},
  { 56,  35, 4, { 6, 10,  23, 3}, { 9,  6,  4,  5}},
  { 56,  42, 3, { 6, 12,  21, 3}, {16,  7,  6,  9}},
  { 56,  52, 2, { 6, 10,  23, 3}, {23, 13,  8, 13}},
  { 64,  32, 5, { 6,  9,  31, 2}, { 5,  3,  2,  3}},
  { 64,  42, 4, { 6,  9,  33, 0}, {11,  6,  5,  0}},
  { 64,  48, 3, { 6, 12,  27, 3}, {16,  8,  6,  9}},
  { 64,  58, 2, { 6, 10,  29, 3}, {23, 13,  8, 13}},
  { 64,  70, 1, { 6, 11,  28, 3}, {24, 18, 12, 18}},
  { 80,  40, 5, { 6, 10,  41, 3}, { 6,  3,  2,  3}},
  { 80,  52, 4, { 6, 10,  41, 3}, {11,  6,  5,  6}},
  { 80,  58, 3, { 6, 11,  40, 3}, {16,  8,  6,  7}},
  { 80,  70, 2, { 6, 10,  41, 3}, {23, 13,  8, 13}},
  { 80,  84, 1, { 6, 10,  41, 3}, {24, 17, 12, 18}},
  { 96,  48, 5, { 7,  9,  53, 3}, { 5,  4,  2,  4}},
  { 96,  58, 4, { 7, 10,  52, 3}, { 9,  6,  4,  6}},
  { 96,  70, 3, { 6, 12,  51, 3}, {16,  9,  6, 10}},
  { 96,  84, 2, { 6, 10,  53, 3}, {22, 12,  9, 12}},
  { 96, 104, 1, { 6, 13,  50, 3}, {24, 18, 13, 19}},
  {112,  58, 5, {14, 17,  50, 3}, { 5,  4,  2,  5}},
  {112,  70, 4, {11, 21,  49, 3}, { 9,  6,  4,  8}},
  {112,  84, 3, {11, 23,  47, 3}, {16,  8,  6,  9}},
  {112, 104, 2, {11, 21,  49, 3}, {23, 12,  9, 14}},
  {128,  64, 5, {12, 19,  62, 3}, { 5,  3,  2,  4}},
  {128,  84, 4, {11, 21,  61, 3}, {11,  6,  5,  7}},
  {128,  96, 3, {11, 22,  60, 3}, {16,  9,  6, 10}},
  {128, 116, 2, {11, 21,  61, 3}, {22, 12,  9, 14}},
  {128, 140, 1, {11, 20,  62, 3}, {24, 17, 13, 19}},
  {160,  80, 5, {11, 19,  87, 3}, { 5,  4,  2,  4}},
  {160, 104, 4, {11, 23,  83, 3}, {11,  6,  5,  9}},
  {160, 116, 3, {11, 24,  82, 3}, {16,  8,  6, 11}},
  {160, 140, 2, {11, 21,  85, 3}, {22, 11,  9, 13}},
  {160, 168, 1, {11, 22,  84, 3}, {24, 18, 12, 19}},
  {192,  96, 5, {11, 20, 110, 3}, { 6,  4,  2,  5}},
  {192, 116, 4, {11, 22, 108, 3}, {10,  6,  4,  9}},
  {192, 140, 3, {11, 24, 106, 3}, {16, 10,  6, 11}},
  {192, 168, 2, {11, 20, 110, 3}, {22, 13,  9, 13}},
  {192, 208, 1, {11, 21, 109, 3}, {24, 20, 13, 24}},
  {224, 116, 5, {12, 22, 131, 3}, { 8,  6,  2,  6}},
  {224, 140, 4, {12, 26, 127, 3}, {12,  8,  4, 11}},
  {224, 168, 3, {11, 20, 134, 3}, {16, 10,  7,  9}},
  {224, 208, 2, {11, 22, 132, 3}, {24, 16, 10, 15}},
  {224, 232, 1, {11, 24, 130, 3}, {24, 20, 12, 20}},
  {256, 128, 5, {11, 24, 154, 3}, { 6,  5,  2,  5}},
  {256, 168, 4, {11, 24, 154, 3}, {12,  9,  5, 10}},
  {256, 192, 3, {11, 27, 151, 3}, {16, 10,  7, 10}},
  {256, 232, 2, {11, 22, 156, 3}, {24, 14, 10, 13}},
  {256, 280, 1, {11, 26, 152, 3}, {24, 19, 14, 18}},
  {320, 160, 5, {11, 26, 200, 3}, { 8,  5,  2,  6}},
  {320, 208, 4, {11, 25, 201, 3}, {13,  9,  5, 10}},
  {320, 280, 2, {11, 26, 200, 3}, {24, 17,  9, 17}},
  {384, 192, 5, {11, 27, 247, 3}, { 8,  6,  2,  7}},
  {384, 280, 3, {11, 24, 250, 3}, {16,  9,  7, 10}},
  {384, 416, 1, {12, 28, 245, 3}, {24, 20, 14, 23}},
  };
  return rows;
}

// Puncturing vector V_PI as a 32-bit mask (bit i = keep mother-code bit i of each group of
// 32), ETSI Table 29: the first bit of every group of four is always kept; PI further bits
// are enabled in the order "second bits of sub-blocks 0,4,2,6,1,5,3,7", then third, then fourth.
constexpr uint32_t puncture_mask(int pi)
{
  constexpr int order[8] = {0, 4, 2, 6, 1, 5, 3, 7};
  uint32_t m = 0x11111111u;
  for (int k = 0; k < pi; ++k) m |= 1u << (4 * order[k & 7] + 1 + (k >> 3));
  return m;
}

// Frequency de-interleaver: carrier index (0..1535, ascending frequency, DC skipped) ->
// QPSK symbol index within the OFDM symbol.  ETSI 14.6.1.
inline const std::array<uint16_t, kCarriers>& carrier_to_qpsk()
{
  static const std::array<uint16_t, kCarriers> tab = [] {
    std::array<uint16_t, kCarriers> t{};
    int pi = 0, n = 0;
    for (int i = 0; i < 2048; ++i) {
      if (i) pi = (13 * pi + 511) % 2048;
      if (pi >= 256 && pi <= 1792 && pi != 1024) {
        int k = pi - 1024;
        k = k < 0 ? 768 + k : 767 + k;
        t[k] = static_cast<uint16_t>(n++);
      }
    }
    return t;
  }();
  return tab;
}

// Phase reference symbol for Mode I as quarter turns (0..3 -> 1, j, -1, -j) per carrier.
// ETSI 14.3.2: phi_k = pi/2 * (h[i][k - k'] + n), Tables 39 and 44.
inline const std::array<uint8_t, kCarriers>& prs_quarter_turns()
{
  static const std::array<uint8_t, kCarriers> tab = [] {
    static const uint8_t h[4][32] = {
        {0, 2, 0, 0, 0, 0, 1, 1, 2, 0, 0, 0, 2, 2, 1, 1, 0, 2, 0, 0, 0, 0, 1, 1, 2, 0, 0, 0, 2, 2, 1, 1},
        {0, 3, 2, 3, 0, 1, 3, 0, 2, 1, 2, 3, 2, 3, 3, 0, 0, 3, 2, 3, 0, 1, 3, 0, 2, 1, 2, 3, 2, 3, 3, 0},
        {0, 0, 0, 2, 0, 2, 1, 3, 2, 2, 0, 2, 2, 0, 1, 3, 0, 0, 0, 2, 0, 2, 1, 3, 2, 2, 0, 2, 2, 0, 1, 3},
        {0, 1, 2, 1, 0, 3, 3, 2, 2, 3, 2, 1, 2, 1, 3, 2, 0, 1, 2, 1, 0, 3, 3, 2, 2, 3, 2, 1, 2, 1, 3, 2}};
    static const uint8_t n[48] = {1, 2, 0, 1, 3, 2, 2, 3, 2, 1, 2, 3, 1, 2, 3, 3, 2, 2, 2, 1, 1, 3, 1, 2,
                                  3, 1, 1, 1, 2, 2, 1, 0, 2, 2, 3, 3, 0, 2, 1, 3, 3, 3, 3, 0, 3, 0, 1, 1};
    std::array<uint8_t, kCarriers> t{};
    for (int k = 0; k < kCarriers; ++k) {
      const int blk = k / 32;
      const int row = blk < 24 ? (blk & 3) : ((4 - (blk & 3)) & 3);
      t[k] = static_cast<uint8_t>((h[row][k & 31] + n[blk]) & 3);
    }
    return t;
  }();
  return tab;
}

// One sub-channel as signalled in FIG 0/1 (reference struct subchannel_info_t, dab.h:35-47).
struct SubChannel {
  int id = -1;        // SubChId 0..63, -1 = slot unused
  int slform = 0;     // 0 = UEP short form, 1 = EEP long form
  int uep_index = 0;
  int start_cu = 0;
  int size_cu = 0;
  int bitrate = 0;    // kbit/s
  int protlev = 0;    // UEP: 1..5; EEP: option<<2 | level
  int ascty = -1;
};

// De-puncturing plan: up to four segments of `blocks` x 128 mother-code bits at puncturing
// index PI, followed by the 24-bit tail at PI 8 (reference depuncture.c:84-132, incl. the
// EEP 2-A @ 8 kbit/s special case of dab_tables.c:98-100).
struct PuncturePlan {
  int blocks[4];
  int pi[4];
  int coded_bits() const   // transmitted bits consumed from the sub-channel
  {
    int n = 12;
    for (int s = 0; s < 4; ++s) n += blocks[s] * 4 * (8 + pi[s]);
    return n;
  }
  int trellis_steps() const { return 32 * (blocks[0] + blocks[1] + blocks[2] + blocks[3]) + 6; }
};

inline int eep_size_multiple(int protlev)
{
  static const int m[8] = {12, 8, 6, 4, 27, 21, 18, 15};
  return m[protlev & 7];
}

inline PuncturePlan puncture_plan(const SubChannel& sc)
{
  PuncturePlan p{};
  if (!sc.slform) {
    const UepProfile& u = uep_table()[sc.uep_index & 63];
    for (int s = 0; s < 4; ++s) { p.blocks[s] = u.l[s]; p.pi[s] = u.pi[s] ? u.pi[s] : 1; }
    return p;
  }
  // {L1 = a n + b, L2 = c n + d, PI1, PI2} per protection level, ETSI Tables 38-41
  static const int eep[8][6] = {{6, -3, 0, 3, 24, 23}, {2, -3, 4, 3, 14, 13}, {6, -3, 0, 3, 8, 7},  {4, -3, 2, 3, 3, 2},
                                {24, -3, 0, 3, 10, 9}, {24, -3, 0, 3, 6, 5},  {24, -3, 0, 3, 4, 3}, {24, -3, 0, 3, 2, 1}};
  const int* e = eep[sc.protlev & 7];
  const int n = sc.size_cu / eep_size_multiple(sc.protlev);
  if (sc.bitrate == 8 && sc.protlev == 1) {
    p.blocks[0] = 5; p.pi[0] = 4; p.blocks[1] = 1; p.pi[1] = 13;
  } else {
    p.blocks[0] = e[0] * n + e[1]; p.pi[0] = e[4];
    p.blocks[1] = e[2] * n + e[3]; p.pi[1] = e[5];
  }
  for (int s = 0; s < 2; ++s) if (p.blocks[s] < 0) p.blocks[s] = 0;
  p.pi[2] = p.pi[3] = 1;
  return p;
}

// FIC: 21 blocks at PI 16, 3 blocks at PI 15, tail (reference fic_depuncture, depuncture.c:45-82)
inline PuncturePlan fic_plan()
{
  PuncturePlan p{};
  p.blocks[0] = 21; p.pi[0] = 16; p.blocks[1] = 3; p.pi[1] = 15; p.pi[2] = p.pi[3] = 1;
  return p;
}

}  // namespace dabhip
