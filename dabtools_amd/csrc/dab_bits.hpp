// dab_bits.hpp — small host-side bit/byte primitives of the DAB channel coder:
// CRC-16/CCITT, energy-dispersal PRBS, the K=7 rate-1/4 mother code.
//
// Host counterparts of reference misc.c:41-58 (dab_descramble_bytes), misc.c:131-150
// (calc_crc / check_fib_crc with the 0x1021 table) and viterbi.c:322-347 (encode).  The
// device versions live in the kernels; these serve the control plane (FIB CRC, ETI header
// CRC) and the synthetic modulator.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace dabhip {

// CRC-16/CCITT, polynomial 0x1021, MSB first, no reflection, caller supplies init value.
inline uint16_t crc16_ccitt(const uint8_t* p, size_t n, uint16_t crc = 0xffff)
{
  struct Table {
    uint16_t t[256];
    Table()
    {
      for (int v = 0; v < 256; ++v) {
        uint16_t c = static_cast<uint16_t>(v << 8);
        for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? static_cast<uint16_t>((c << 1) ^ 0x1021) : static_cast<uint16_t>(c << 1);
        t[v] = c;
      }
    }
  };
  static const Table tab;
  for (size_t i = 0; i < n; ++i) crc = static_cast<uint16_t>(tab.t[(p[i] ^ (crc >> 8)) & 0xff] ^ (crc << 8));
  return crc;
}

// A FIB (30 data bytes + complemented CRC) is good iff the CRC over all 32 bytes leaves 0x1D0F.
inline bool fib_crc_ok(const uint8_t* fib) { return crc16_ccitt(fib, 32) == 0x1d0f; }

// Energy dispersal sequence x^9 + x^5 + 1, register preset to all ones; byte k of the
// sequence, MSB first.  XOR-ing is its own inverse (scramble == descramble).
class Prbs {
 public:
  uint8_t next_byte()
  {
    unsigned q = 0;
    for (int j = 0; j < 8; ++j) {
      const unsigned fb = ((reg_ >> 8) ^ (reg_ >> 4)) & 1u;
      reg_ = ((reg_ << 1) | fb) & 0x1ff;
      q = (q << 1) | fb;
    }
    return static_cast<uint8_t>(q);
  }

 private:
  unsigned reg_ = 0x1ff;
};

inline void energy_dispersal(uint8_t* buf, size_t n)
{
  Prbs g;
  for (size_t i = 0; i < n; ++i) buf[i] ^= g.next_byte();
}

// Mother code generators as tap masks over the 7-bit register whose LSB is the newest bit
// (octal 133,171,145,133 bit-reversed; reference viterbi.c:35).
constexpr unsigned kConvPoly[4] = {0x6d, 0x4f, 0x53, 0x6d};

inline unsigned parity7(unsigned x)
{
  x ^= x >> 4;
  x ^= x >> 2;
  x ^= x >> 1;
  return x & 1u;
}

// Encode `nbits` data bits (MSB first in `data`) plus six zero tail bits into 4*(nbits+6)
// mother-code bits, one bit per output byte.
inline std::vector<uint8_t> conv_encode(const uint8_t* data, int nbits)
{
  std::vector<uint8_t> out(static_cast<size_t>(4) * (nbits + 6));
  unsigned reg = 0;
  size_t o = 0;
  for (int i = 0; i < nbits + 6; ++i) {
    const unsigned bit = i < nbits ? (data[i >> 3] >> (7 - (i & 7))) & 1u : 0u;
    reg = ((reg << 1) | bit) & 0x7f;
    for (int j = 0; j < 4; ++j) out[o++] = static_cast<uint8_t>(parity7(reg & kConvPoly[j]));
  }
  return out;
}

}  // namespace dabhip
