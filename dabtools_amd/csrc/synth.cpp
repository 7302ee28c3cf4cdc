// synth.cpp — synthetic DAB Mode-I modulator (host only): ensemble configuration ->
// 2.048 Msps cu8 IQ.  The reference contains no transmitter; this is the workload
// generator for tests and bench.py, built as the exact inverse of the receive chain the
// reference implements (fic.c:47-130 FIG parsing, depuncture.c, misc.c:29-58,
// input_sdr.c:132-162 demap, sdr_prstab.c PRS, dab_tables.c tables).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dabhip.h"
#include "dab_bits.hpp"
#include "dab_tables.hpp"
#include "synth.hpp"

namespace dabhip {
void set_error(const std::string& msg);

namespace {

// counter-based generator: value = f(seed, stream of keys); reproducible on any machine
inline uint64_t mix64(uint64_t z)
{
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
inline uint64_t key(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) { return mix64(mix64(mix64(seed ^ mix64(a)) + b) + c); }

enum : uint64_t { kDomPayload = 1, kDomFiller = 2, kDomNoise = 3 };

SubChannel to_subchannel(const dabhip_subch_cfg& c)
{
  SubChannel sc;
  sc.id = c.id;
  sc.start_cu = c.start_cu;
  sc.slform = c.slform;
  if (!c.slform) {
    const UepProfile& u = uep_table()[c.uep_index & 63];
    sc.uep_index = c.uep_index & 63;
    sc.size_cu = u.size_cu;
    sc.bitrate = u.bitrate;
    sc.protlev = u.protlevel;
  } else {
    sc.protlev = c.eep_protlev & 7;
    sc.size_cu = c.size_cu;
    sc.bitrate = (c.size_cu / eep_size_multiple(sc.protlev)) * ((sc.protlev & 4) ? 32 : 8);
  }
  return sc;
}

// the multiplex in force at logical CIF `cif`: in the MSC (lead = 0) or as announced by the FIC (lead = the entry's fic_lead)
struct Multiplex {
  int nsub;
  const dabhip_subch_cfg* sub;
};
Multiplex multiplex_at(const dabhip_synth_cfg& cfg, int cif, bool fic)
{
  Multiplex m{cfg.nsub, cfg.sub};
  for (const dabhip_reconf_cfg& r : cfg.reconf)
    if (r.at_cif > 0 && cif >= r.at_cif - (fic ? r.fic_lead : 0)) m = Multiplex{r.nsub, r.sub};
  return m;
}

bool validate_multiplex(int nsub, const dabhip_subch_cfg* sub)
{
  if (nsub < 0 || nsub > 64) { set_error("synth: nsub out of range"); return false; }
  std::vector<char> used(864, 0);
  for (int k = 0; k < nsub; ++k) {
    const SubChannel sc = to_subchannel(sub[k]);
    if (sc.id < 0 || sc.id > 63 || sc.bitrate <= 0 || sc.start_cu < 0 || sc.start_cu + sc.size_cu > 864) {
      set_error("synth: sub-channel " + std::to_string(k) + " does not fit the CIF");
      return false;
    }
    if (puncture_plan(sc).coded_bits() > sc.size_cu * 64) { set_error("synth: sub-channel over-long"); return false; }
    for (int cu = sc.start_cu; cu < sc.start_cu + sc.size_cu; ++cu) {
      if (used[cu]) { set_error("synth: overlapping sub-channels"); return false; }
      used[cu] = 1;
    }
  }
  return true;
}

bool channel_active(const dabhip_channel_cfg& c)
{
  return c.sro_ppm != 0.0 || c.echo_delay[0] || c.echo_delay[1] || c.fade_depth != 0.0 || c.iq_gain_db != 0.0 || c.iq_phase_deg != 0.0;
}

bool validate(const dabhip_synth_cfg& cfg)
{
  if (!validate_multiplex(cfg.nsub, cfg.sub)) return false;
  int prev = 0;
  for (const dabhip_reconf_cfg& r : cfg.reconf) {
    if (r.at_cif == 0) continue;
    if (r.at_cif <= prev || r.fic_lead < 0) { set_error("synth: reconfigurations must be at ascending CIFs > 0 with fic_lead >= 0"); return false; }
    prev = r.at_cif;
    if (!validate_multiplex(r.nsub, r.sub)) return false;
  }
  if (cfg.skip_samples < 0 || cfg.skip_samples >= kTfSamples) { set_error("synth: skip_samples out of range"); return false; }
  const dabhip_channel_cfg& c = cfg.channel;
  for (int e = 0; e < 2; ++e)
    if (c.echo_delay[e] < 0 || c.echo_delay[e] > 2047) { set_error("synth: echo delay out of range (0 .. 2047 samples)"); return false; }
  if (c.fade_depth < 0.0 || c.fade_depth >= 1.0 || std::fabs(c.sro_ppm) > 1000.0) { set_error("synth: channel parameters out of range"); return false; }
  return true;
}

void payload_bytes(const dabhip_synth_cfg& cfg, int cif, int slot, uint8_t* out, int n)
{
  for (int i = 0; i < n; i += 8) {
    uint64_t v = key(cfg.seed, kDomPayload, (static_cast<uint64_t>(static_cast<uint32_t>(cif)) << 8) | static_cast<uint64_t>(slot), static_cast<uint64_t>(i));
    for (int b = 0; b < 8 && i + b < n; ++b) out[i + b] = static_cast<uint8_t>(v >> (8 * b));
  }
}

// the three FIBs of one CIF: FIG 0/0 (ensemble, CIF counter) and FIG 0/1 (sub-channel
// organisation) entries packed greedily, then end marker / padding and the FIB CRC
void build_fibs(const dabhip_synth_cfg& cfg, int cif, uint8_t* out96)
{
  const int count = ((cfg.cif_count0 + cif) % 5000 + 5000) % 5000;
  std::vector<std::vector<uint8_t>> entries;
  const Multiplex mux = multiplex_at(cfg, cif, true);
  for (int k = 0; k < mux.nsub; ++k) {
    const SubChannel sc = to_subchannel(mux.sub[k]);
    std::vector<uint8_t> e;
    e.push_back(static_cast<uint8_t>((sc.id << 2) | ((sc.start_cu >> 8) & 3)));
    e.push_back(static_cast<uint8_t>(sc.start_cu & 0xff));
    if (!sc.slform) {
      e.push_back(static_cast<uint8_t>(sc.uep_index & 0x3f));
    } else {
      e.push_back(static_cast<uint8_t>(0x80 | ((sc.protlev >> 2) << 4) | ((sc.protlev & 3) << 2) | ((sc.size_cu >> 8) & 3)));
      e.push_back(static_cast<uint8_t>(sc.size_cu & 0xff));
    }
    entries.push_back(e);
  }
  size_t next = 0;
  for (int f = 0; f < 3; ++f) {
    uint8_t* fib = out96 + 32 * f;
    std::memset(fib, 0, 32);
    int pos = 0;
    if (f == 0) {
      const uint8_t fig00[6] = {0x05, 0x00, static_cast<uint8_t>(cfg.eid >> 8), static_cast<uint8_t>(cfg.eid & 0xff),
                                static_cast<uint8_t>(count / 250), static_cast<uint8_t>(count % 250)};
      std::memcpy(fib, fig00, 6);
      pos = 6;
    }
    if (next < entries.size() && pos + 2 + static_cast<int>(entries[next].size()) <= 30) {
      const int hdr = pos;
      pos += 2;
      int len = 1;
      while (next < entries.size() && pos + static_cast<int>(entries[next].size()) <= 30 && len + entries[next].size() <= 31) {
        std::memcpy(fib + pos, entries[next].data(), entries[next].size());
        pos += static_cast<int>(entries[next].size());
        len += static_cast<int>(entries[next].size());
        ++next;
      }
      fib[hdr] = static_cast<uint8_t>(len);   // FIG type 0, length
      fib[hdr + 1] = 0x01;                    // C/N=0 OE=0 P/D=0 extension 1
    }
    if (f == 2 && cfg.fib_patch_len > 0 && cif >= cfg.fib_patch_from_cif && pos == 0) {      // test vector: this FIB's content verbatim (dabhip.h)
      pos = std::min(cfg.fib_patch_len, 30);
      std::memcpy(fib, cfg.fib_patch, static_cast<size_t>(pos));
    }
    if (pos < 30) fib[pos] = 0xff;            // end marker, rest zero padding
    const uint16_t crc = static_cast<uint16_t>(~crc16_ccitt(fib, 30));
    fib[30] = static_cast<uint8_t>(crc >> 8);
    fib[31] = static_cast<uint8_t>(crc & 0xff);
  }
}

// keep the mother-code bits the puncturing plan transmits
void puncture(const std::vector<uint8_t>& mother, const PuncturePlan& plan, std::vector<uint8_t>& out)
{
  size_t x = 0;
  for (int s = 0; s < 4; ++s) {
    const uint32_t m = puncture_mask(plan.pi[s]);
    for (int i = 0; i < 128 * plan.blocks[s]; ++i, ++x)
      if ((m >> (i & 31)) & 1u) out.push_back(mother[x]);
  }
  const uint32_t mt = puncture_mask(8);
  for (int i = 0; i < 24; ++i, ++x)
    if ((mt >> i) & 1u) out.push_back(mother[x]);
}

// logical (pre time-interleaving) CIF r: 55296 bits
void logical_cif(const dabhip_synth_cfg& cfg, int r, uint8_t* bits)
{
  for (int i = 0; i < kCifBits; i += 64) {
    const uint64_t v = key(cfg.seed, kDomFiller, static_cast<uint64_t>(static_cast<uint32_t>(r)), static_cast<uint64_t>(i));
    for (int b = 0; b < 64; ++b) bits[i + b] = static_cast<uint8_t>((v >> b) & 1u);
  }
  if (r < 0) return;   // before the start of the transmission: filler only
  std::vector<uint8_t> data, coded;
  const Multiplex mux = multiplex_at(cfg, r, false);
  for (int k = 0; k < mux.nsub; ++k) {
    const SubChannel sc = to_subchannel(mux.sub[k]);
    const int nbytes = sc.bitrate * 3;
    data.resize(static_cast<size_t>(nbytes));
    payload_bytes(cfg, r, k, data.data(), nbytes);
    energy_dispersal(data.data(), data.size());
    const std::vector<uint8_t> mother = conv_encode(data.data(), nbytes * 8);
    coded.clear();
    puncture(mother, puncture_plan(sc), coded);
    std::memcpy(bits + sc.start_cu * 64, coded.data(), coded.size());
  }
}

void fic_bits_of_cif(const dabhip_synth_cfg& cfg, int cif, uint8_t* bits2304)
{
  uint8_t fibs[96];
  build_fibs(cfg, cif, fibs);
  energy_dispersal(fibs, 96);
  const std::vector<uint8_t> mother = conv_encode(fibs, 768);
  std::vector<uint8_t> coded;
  puncture(mother, fic_plan(), coded);
  std::memcpy(bits2304, coded.data(), 2304);
}

// in-place radix-2 complex DFT, sign = +1 for the synthesis direction
void fft2048(double* re, double* im, int sign)
{
  constexpr int n = 2048;
  struct Twiddles {
    std::vector<double> r, i;
    Twiddles() : r(n / 2), i(n / 2)
    {
      for (int k = 0; k < n / 2; ++k) { r[k] = std::cos(2 * M_PI * k / n); i[k] = std::sin(2 * M_PI * k / n); }
    }
  };
  static const Twiddles tw;   // thread-safe one-time initialisation
  const std::vector<double>&wr = tw.r, &wi = tw.i;
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  for (int len = 2; len <= n; len <<= 1) {
    const int step = n / len;
    for (int i = 0; i < n; i += len)
      for (int k = 0; k < len / 2; ++k) {
        const double cr = wr[k * step], ci = sign * wi[k * step];
        const int a = i + k, b = a + len / 2;
        const double tr = re[b] * cr - im[b] * ci, ti = re[b] * ci + im[b] * cr;
        re[b] = re[a] - tr; im[b] = im[a] - ti;
        re[a] += tr; im[a] += ti;
      }
  }
}

struct Gauss {
  uint64_t seed, ctr = 0;
  bool have = false;
  double spare = 0;
  double next()
  {
    if (have) { have = false; return spare; }
    const uint64_t a = key(seed, kDomNoise, ctr, 0), b = key(seed, kDomNoise, ctr, 1);
    ++ctr;
    const double u1 = (static_cast<double>(a >> 11) + 1.0) / 9007199254740993.0;
    const double u2 = static_cast<double>(b >> 11) / 9007199254740992.0;
    const double r = std::sqrt(-2.0 * std::log(u1));
    spare = r * std::sin(2 * M_PI * u2);
    have = true;
    return r * std::cos(2 * M_PI * u2);
  }
};

// The channel between modulator and receiver (dabhip_channel_cfg): a chain of sample-by-sample stages, each of which is skipped when its parameters
// are zero, so that an ideal channel hands the modulator's samples through untouched (the captures of rounds 1-4, byte for byte).
// Not part of the receive path and of no parity claim: whatever IQ it makes, the reference, the oracle and the GPU get the same bytes.
class Channel {
 public:
  explicit Channel(const dabhip_synth_cfg& cfg)
      : c_(cfg.channel), cfo_turns_(cfg.cfo_hz / 2048000.0), echo_(c_.echo_delay[0] || c_.echo_delay[1]), hist_r_(kHist, 0.0), hist_i_(kHist, 0.0),
        ring_r_(kRing, 0.0), ring_i_(kRing, 0.0)
  {
    rate_ = 1.0 + c_.sro_ppm * 1e-6;
    iq_g_ = std::pow(10.0, c_.iq_gain_db / 20.0);
    iq_c_ = std::cos(c_.iq_phase_deg * M_PI / 180.0);
    iq_s_ = std::sin(c_.iq_phase_deg * M_PI / 180.0);
    iq_ = c_.iq_gain_db != 0.0 || c_.iq_phase_deg != 0.0;
    for (int k = 0; k < 2 * kHalf; ++k) {
      tab_c_[k] = std::cos(M_PI * (k - kHalf + 1) / kHalf);
      tab_s_[k] = std::sin(M_PI * (k - kHalf + 1) / kHalf);
    }
  }
  // modulator sample n (in order); out(re, im) receives the receiver-side samples, in order
  template <class Out>
  void push(double xr, double xi, Out&& out)
  {
    const long long n = n_++;
    if (echo_) {
      hist_r_[n & (kHist - 1)] = xr;
      hist_i_[n & (kHist - 1)] = xi;
      double yr = xr, yi = xi;
      for (int e = 0; e < 2; ++e) {
        const int d = c_.echo_delay[e];
        if (!d || n < d) continue;
        const double ph = 2 * M_PI * std::fmod(c_.echo_phase[e] + c_.echo_doppler_hz[e] / 2048000.0 * static_cast<double>(n), 1.0);
        const double gr = c_.echo_gain[e] * std::cos(ph), gi = c_.echo_gain[e] * std::sin(ph);
        const double pr = hist_r_[(n - d) & (kHist - 1)], pi = hist_i_[(n - d) & (kHist - 1)];
        yr += gr * pr - gi * pi;
        yi += gr * pi + gi * pr;
      }
      xr = yr;
      xi = yi;
    }
    if (c_.fade_depth != 0.0) {
      const double a = 1.0 - c_.fade_depth * 0.5 * (1.0 - std::cos(2 * M_PI * std::fmod(c_.fade_hz / 2048000.0 * static_cast<double>(n), 1.0)));
      xr *= a;
      xi *= a;
    }
    if (cfo_turns_ != 0.0) {
      const double ph = 2 * M_PI * std::fmod(cfo_turns_ * static_cast<double>(n), 1.0);
      const double c = std::cos(ph), s = std::sin(ph), r = xr * c - xi * s;
      xi = xr * s + xi * c;
      xr = r;
    }
    if (c_.sro_ppm == 0.0) { receiver(xr, xi, out); return; }
    ring_r_[n & (kRing - 1)] = xr;
    ring_i_[n & (kRing - 1)] = xi;
    // every output whose 2 kHalf taps end at or before sample n: output m sits at input position m * rate_
    for (;;) {
      const double pos = static_cast<double>(m_) * rate_;
      const long long i0 = static_cast<long long>(std::floor(pos));
      if (i0 + kHalf > n) break;
      const double f = pos - static_cast<double>(i0);
      double yr = 0, yi = 0;
      if (f == 0.0) {
        yr = ring_r_[i0 & (kRing - 1)];
        yi = ring_i_[i0 & (kRing - 1)];
      } else {
        // taps k = -kHalf+1 .. kHalf at distance t = k - f: sinc(t) = -(-1)^k sin(pi f) / (pi t), Hann window 0.5 (1 + cos(pi t / kHalf))
        const double sf = std::sin(M_PI * f), cw = std::cos(M_PI * f / kHalf), sw = std::sin(M_PI * f / kHalf);
        double sum = 0;
        for (int j = 0; j < 2 * kHalf; ++j) {
          const int k = j - kHalf + 1;
          const long long idx = i0 + k;
          const double t = static_cast<double>(k) - f;
          const double w = 0.5 * (1.0 + tab_c_[j] * cw + tab_s_[j] * sw);
          const double h = ((k & 1) ? sf : -sf) / (M_PI * t) * w;
          sum += h;
          if (idx < 0) continue;
          yr += h * ring_r_[idx & (kRing - 1)];
          yi += h * ring_i_[idx & (kRing - 1)];
        }
        yr /= sum;
        yi /= sum;
      }
      ++m_;
      receiver(yr, yi, out);
    }
  }

 private:
  template <class Out>
  void receiver(double xr, double xi, Out&& out)
  {
    if (iq_) xi = iq_g_ * (xi * iq_c_ + xr * iq_s_);
    out(xr, xi);
  }
  static constexpr int kHist = 2048, kRing = 64, kHalf = 8;
  dabhip_channel_cfg c_;
  double cfo_turns_, rate_ = 1.0, iq_g_ = 1.0, iq_c_ = 1.0, iq_s_ = 0.0;
  bool echo_, iq_ = false;
  long long n_ = 0, m_ = 0;
  std::vector<double> hist_r_, hist_i_, ring_r_, ring_i_;
  double tab_c_[2 * kHalf], tab_s_[2 * kHalf];
};

}  // namespace

bool synth_channel_active(const dabhip_synth_cfg& cfg) { return channel_active(cfg.channel); }

uint64_t synth_noise_key(uint64_t seed, uint64_t ctr, uint64_t which) { return key(seed, kDomNoise, ctr, which); }

bool synth_validate(const dabhip_synth_cfg& cfg) { return validate(cfg); }

double synth_noise_rms(const dabhip_synth_cfg& cfg)
{
  return cfg.snr_db >= 100.0 ? 0.0 : cfg.amplitude * std::sqrt(static_cast<double>(kCarriers)) / std::pow(10.0, cfg.snr_db / 20.0) / std::sqrt(2.0);
}

SymbolBits::SymbolBits(const dabhip_synth_cfg& cfg) : cfg_(cfg), window_(16, std::vector<uint8_t>(kCifBits))
{
  for (int r = -15; r < 0; ++r) logical_cif(cfg_, r, window_[(r + 16) & 15].data());
}

void SymbolBits::next_tf(uint8_t* symbits)
{
  static const int tmap[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * tf_ + q;
    fic_bits_of_cif(cfg_, c, symbits + 2304 * q);
    logical_cif(cfg_, c, window_[c & 15].data());
    // time interleaving: inverse of misc.c:29-39 (tx CIF c carries logical CIF c - map[i&15])
    uint8_t* txcif = symbits + 3 * kBitsPerSym + static_cast<size_t>(q) * kCifBits;
    for (int i = 0; i < kCifBits; ++i) txcif[i] = window_[(c - tmap[i & 15]) & 15][i];
  }
  ++tf_;
}

}  // namespace dabhip

using namespace dabhip;

extern "C" int dabhip_synth_preset(int preset, dabhip_synth_cfg* cfg)
{
  if (!cfg) return -1;
  std::memset(cfg, 0, sizeof *cfg);
  cfg->eid = 0xC181;
  cfg->seed = 1;
  cfg->amplitude = 1.0;
  cfg->snr_db = 1000.0;
  auto uep = [&](int id, int cu, int idx) { cfg->sub[cfg->nsub++] = dabhip_subch_cfg{id, cu, 0, idx, 0, 0}; };
  auto eep = [&](int id, int cu, int lev, int size) { cfg->sub[cfg->nsub++] = dabhip_subch_cfg{id, cu, 1, 0, lev, size}; };
  if (preset == 0) {          // 12 sub-channels, 1136 kbit/s, 862 of 864 CU
    uep(1, 0, 35); uep(2, 96, 35); uep(3, 192, 35); uep(4, 288, 35);   // 128 kbit/s PL3
    uep(5, 384, 45); uep(6, 524, 45);                                  // 192 kbit/s PL3
    eep(7, 664, 2, 48); eep(8, 712, 2, 48);                            // 64 kbit/s 3-A
    eep(9, 760, 0, 48);                                                // 32 kbit/s 1-A
    eep(10, 808, 5, 42);                                               // 64 kbit/s 2-B
    eep(11, 850, 1, 8);                                                // 8 kbit/s 2-A (special case)
    eep(12, 858, 3, 4);                                                // 8 kbit/s 4-A
  } else if (preset == 1) {   // 4 light sub-channels
    uep(1, 0, 35); eep(2, 100, 2, 48); eep(5, 200, 5, 21); eep(9, 300, 1, 8);
  } else {
    set_error("synth: unknown preset");
    return -1;
  }
  return 0;
}

extern "C" size_t dabhip_synth_bytes(const dabhip_synth_cfg* cfg, int ntf)
{
  if (!cfg || ntf <= 0) return 0;
  const size_t n = static_cast<size_t>(ntf) * kTfSamples;
  if (cfg->channel.sro_ppm == 0.0) return (n - static_cast<size_t>(cfg->skip_samples)) * 2;
  return (static_cast<size_t>(static_cast<double>(n) / (1.0 + cfg->channel.sro_ppm * 1e-6)) + 16 - static_cast<size_t>(cfg->skip_samples)) * 2;   // upper bound
}

extern "C" int dabhip_synth_payload(const dabhip_synth_cfg* cfg, int cif_index, int slot, uint8_t* out, int cap)
{
  if (!cfg) { set_error("synth_payload: bad slot"); return -1; }
  const Multiplex mux = multiplex_at(*cfg, cif_index, false);
  if (slot < 0 || slot >= mux.nsub) { set_error("synth_payload: bad slot"); return -1; }
  const int n = to_subchannel(mux.sub[slot]).bitrate * 3;
  if (cap < n) { set_error("synth_payload: buffer too small"); return -1; }
  payload_bytes(*cfg, cif_index, slot, out, n);
  return n;
}

extern "C" int dabhip_synth_fibs(const dabhip_synth_cfg* cfg, int cif_index, uint8_t* out96)
{
  if (!cfg || !out96) return -1;
  build_fibs(*cfg, cif_index, out96);
  return 96;
}

extern "C" int64_t dabhip_synth_generate(const dabhip_synth_cfg* cfg, int ntf, uint8_t* iq, size_t cap)
{
  if (!cfg || !iq || ntf <= 0) { set_error("synth_generate: bad arguments"); return -1; }
  if (!validate(*cfg)) return -1;
  const size_t total = dabhip_synth_bytes(cfg, ntf);
  if (cap < total) { set_error("synth_generate: buffer too small"); return -1; }

  const auto& qpsk_of_carrier = carrier_to_qpsk();
  const auto& prs = prs_quarter_turns();
  SymbolBits bits(*cfg);                                                  // logical CIFs in a sliding window of 16
  const double noise_rms_rail = synth_noise_rms(*cfg);
  Gauss gauss{cfg->seed};
  std::vector<uint8_t> symbits(static_cast<size_t>(kBitsPerSym) * 75);   // data symbols 1..75 of one TF
  std::vector<double> re(2048), im(2048);
  std::vector<uint8_t> phase(kCarriers);                                  // in eighth turns
  static const double c8[8] = {1, M_SQRT1_2, 0, -M_SQRT1_2, -1, -M_SQRT1_2, 0, M_SQRT1_2};
  static const double s8[8] = {0, M_SQRT1_2, 1, M_SQRT1_2, 0, -M_SQRT1_2, -1, -M_SQRT1_2};
  size_t outpos = 0;
  long long sample_index = 0;                             // receiver-side sample count
  Channel channel(*cfg);                                  // echoes, fading, carrier offset, sample-rate offset, I/Q imbalance
  auto quantise = [&](double xr, double xi) {
    if (sample_index++ < cfg->skip_samples) return;
    if (outpos + 2 > cap) return;
    if (noise_rms_rail > 0) { xr += noise_rms_rail * gauss.next(); xi += noise_rms_rail * gauss.next(); }
    double a = std::floor(127.0 + xr + 0.5), b = std::floor(127.0 + xi + 0.5);
    a = a < 1 ? 1 : (a > 254 ? 254 : a);
    b = b < 1 ? 1 : (b > 254 ? 254 : b);
    iq[outpos++] = static_cast<uint8_t>(a);
    iq[outpos++] = static_cast<uint8_t>(b);
  };
  auto emit = [&](double xr, double xi) { channel.push(xr, xi, quantise); };

  for (int tf = 0; tf < ntf; ++tf) {
    bits.next_tf(symbits.data());
    for (int n = 0; n < kNullSamples; ++n) emit(0, 0);
    for (int l = 0; l < kSymbolsPerTf; ++l) {
      if (l == 0) {
        for (int k = 0; k < kCarriers; ++k) phase[k] = static_cast<uint8_t>(2 * prs[k]);
      } else {
        const uint8_t* p = symbits.data() + static_cast<size_t>(l - 1) * kBitsPerSym;
        for (int k = 0; k < kCarriers; ++k) {
          const int n = qpsk_of_carrier[k];
          static const uint8_t inc[4] = {1, 3, 7, 5};   // (b0,b1): 00->45deg 10->135 01->315 11->225
          phase[k] = static_cast<uint8_t>((phase[k] + inc[p[n] | (p[1536 + n] << 1)]) & 7);
        }
      }
      std::fill(re.begin(), re.end(), 0.0);
      std::fill(im.begin(), im.end(), 0.0);
      for (int k = 0; k < kCarriers; ++k) {
        const int bin = k < 768 ? 1280 + k : k - 767;
        re[bin] = cfg->amplitude * c8[phase[k]];
        im[bin] = cfg->amplitude * s8[phase[k]];
      }
      fft2048(re.data(), im.data(), +1);
      for (int n = 2048 - kCpSamples; n < 2048; ++n) emit(re[n], im[n]);
      for (int n = 0; n < 2048; ++n) emit(re[n], im[n]);
    }
  }
  return static_cast<int64_t>(outpos);
}
