// synth.hpp — pieces of the synthetic Mode-I modulator shared by the host generator (synth.cpp) and the device
// modulator (k_synth.hip): the bit content of a transmission frame and the noise generator's keys.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/dabhip.h"

namespace dabhip {

// The 75 data symbols of successive transmission frames as 0/1 bytes: 3 x 3072 FIC bits (4 punctured FIC blocks of
// 2304 bits), then 4 transmitted CIFs of 55296 bits, time-interleaved over the 16 logical CIFs before.
class SymbolBits {
 public:
  explicit SymbolBits(const dabhip_synth_cfg& cfg);
  void next_tf(uint8_t* symbits /* 75 * 3072 bytes */);

 private:
  dabhip_synth_cfg cfg_;
  std::vector<std::vector<uint8_t>> window_;
  int tf_ = 0;
};

bool synth_validate(const dabhip_synth_cfg& cfg);
bool synth_channel_active(const dabhip_synth_cfg& cfg);             // any stage of dabhip_channel_cfg switched on
double synth_noise_rms(const dabhip_synth_cfg& cfg);                 // per rail, in LSB
uint64_t synth_noise_key(uint64_t seed, uint64_t ctr, uint64_t which);

}  // namespace dabhip
