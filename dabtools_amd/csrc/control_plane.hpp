// control_plane.hpp — the per-ensemble control plane that stays on the host:
// FIG 0/0, 0/1, 0/2 parsing, ensemble-information merge, the 10-TF lock rule, the 16-CIF
// ring and the ETI(NI) header.  384 bytes in, a few hundred bytes out per transmission
// frame; everything data-heavy is on the device.
//
// Reference behaviour reproduced (file:line under src/):
//   fib_parse / fib_decode    fic.c:47-147
//   merge_info                misc.c:14-27
//   dab_process_frame         dab.c:35-98   (lock FSM, ring of 16 CIF pointers over 5 TF buffers)
//   init_eti                  misc.c:153-213
//   CIF counter increment     misc.c:306-313
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dab_bits.hpp"
#include "dab_tables.hpp"
#include "device_types.hpp"

namespace dabhip {

struct EnsembleInfo {
  uint16_t eid = 0;
  uint8_t cif_hi = 0, cif_lo = 0;
  SubChannel sub[64];
  // bookkeeping for speed only (the per-TF pass used to clear and walk all 64 slots, 2 KB, for ensembles that carry a dozen sub-channels):
  // present = slots whose id is set (slot i holds id i), touched = slots written at all (FIG 0/2 may set ascty of a slot without an id)
  uint64_t present = 0, touched = 0;
  // after filling sub[] by hand (dabhip_host_eti_header): derive the masks from the slots
  void rescan()
  {
    present = touched = 0;
    for (int i = 0; i < 64; ++i)
      if (sub[i].id >= 0) { present |= 1ull << i; touched |= 1ull << i; }
  }
};
template <class F>
inline void for_each_slot(uint64_t mask, F&& fn)     // ascending slot order, like the reference's loops over 64 slots
{
  while (mask) {
    const int i = __builtin_ctzll(mask);
    mask &= mask - 1;
    fn(i);
  }
}

// Parse the FIGs of one FIB into `info`.  `limit` bounds reads for FIGs whose length field
// runs past the FIB: the reference reads on into the following FIBs / CRC flags of
// struct tf_fibs_t (dab.h:21-25); the caller passes an image with that layout.
inline void parse_fib(EnsembleInfo& info, const uint8_t* fib, const uint8_t* limit)
{
  auto at = [&](int i) -> int { return fib + i < limit ? fib[i] : 0; };
  int i = 0;
  while (at(i) != 0xff && i < 30) {
    const int type = at(i) >> 5, len = at(i) & 0x1f;
    ++i;
    if (type == 0) {
      const int ext = at(i) & 0x1f, pd = (at(i) >> 5) & 1;
      if (ext == 0) {                     // FIG 0/0: ensemble id and CIF counter
        info.eid = static_cast<uint16_t>((at(i + 1) << 8) | at(i + 2));
        info.cif_hi = static_cast<uint8_t>(at(i + 3) & 0x1f);
        info.cif_lo = static_cast<uint8_t>(at(i + 4));
      } else if (ext == 1) {              // FIG 0/1: sub-channel organisation
        int j = i + 1;
        while (j < i + len) {
          const int id = at(j) >> 2;
          SubChannel& sc = info.sub[id];
          sc.id = id;
          info.present |= 1ull << id;
          info.touched |= 1ull << id;
          sc.start_cu = ((at(j) & 3) << 8) | at(j + 1);
          sc.slform = at(j + 2) >> 7;
          if (!sc.slform) {
            sc.uep_index = at(j + 2) & 0x3f;
            const UepProfile& u = uep_table()[sc.uep_index];
            sc.size_cu = u.size_cu;
            sc.bitrate = u.bitrate;
            sc.protlev = u.protlevel;
            j += 3;
          } else {
            sc.protlev = ((at(j + 2) >> 2) & 3) | (((at(j + 2) >> 4) & 7) << 2);
            sc.size_cu = ((at(j + 2) & 3) << 8) | at(j + 3);
            sc.bitrate = (sc.size_cu / eep_size_multiple(sc.protlev)) * ((sc.protlev & 4) ? 32 : 8);
            j += 4;
          }
        }
      } else if (ext == 2) {              // FIG 0/2: only the audio service component type is kept
        int j = i + 1;
        while (j < i + len) {
          j += pd ? 4 : 2;
          const int ncomp = at(j) & 0x0f;
          ++j;
          for (int k = 0; k < ncomp; ++k) {
            if ((at(j) >> 6) == 0) {
              info.sub[at(j + 1) >> 2].ascty = at(j) & 0x3f;
              info.touched |= 1ull << (at(j + 1) >> 2);
            }
            j += 2;
          }
        }
      }
    }
    i += len;
  }
}

inline void decode_fibs(EnsembleInfo& info, const uint8_t* fibs /*12 x 32*/, const uint8_t* crc_ok /*12*/)
{
  uint8_t image[12 * 32 + 12];
  std::memcpy(image, fibs, 12 * 32);
  std::memcpy(image + 12 * 32, crc_ok, 12);
  for_each_slot(info.touched, [&](int i) { info.sub[i] = SubChannel{}; });     // = info = EnsembleInfo{}: untouched slots are pristine
  info.present = info.touched = 0;
  info.eid = 0;
  info.cif_hi = info.cif_lo = 0;
  for (int f = 0; f < 12; ++f)
    if (crc_ok[f]) parse_fib(info, image + 32 * f, image + sizeof image);
}

// SYNC, FC, STC, EOH of one ETI(NI) frame; returns the byte count (8 + 4 NST + 4).
// keep: bit i set = sub-channel id i is carried (sub-channel filter, TODO.md:28-31; all ones = the reference's frame)
inline int build_eti_header(uint8_t* eti, const EnsembleInfo& info, uint64_t keep = ~0ull)
{
  int n = 0, nst = 0, fl = 0;
  eti[n++] = 0xff;                                              // ERR
  const bool odd = info.cif_lo & 1;
  eti[n++] = odd ? 0xf8 : 0x07;                                 // FSYNC alternates
  eti[n++] = odd ? 0xc5 : 0x3a;
  eti[n++] = odd ? 0x49 : 0xb6;
  eti[n++] = info.cif_lo;                                       // FCT
  for_each_slot(info.present & keep, [&](int i) { ++nst; fl += info.sub[i].bitrate * 3 / 4; });
  fl += nst + 1 + 24;                                           // STC + EOH + FIC (Mode I) in words
  eti[n++] = static_cast<uint8_t>(0x80 | nst);                  // FICF | NST
  const int fp = (info.cif_hi * 250 + info.cif_lo) % 8;
  eti[n++] = static_cast<uint8_t>((fp << 5) | (1 << 3) | ((fl & 0x700) >> 8));   // FP, MID = 1, FL
  eti[n++] = static_cast<uint8_t>(fl & 0xff);
  for_each_slot(info.present & keep, [&](int i) {
    const SubChannel& sc = info.sub[i];
    const int tpl = sc.slform ? (0x20 | sc.protlev) : (0x10 | (sc.protlev - 1));
    const int stl = sc.bitrate * 3 / 8;
    eti[n++] = static_cast<uint8_t>((sc.id << 2) | ((sc.start_cu & 0x300) >> 8));
    eti[n++] = static_cast<uint8_t>(sc.start_cu & 0xff);
    eti[n++] = static_cast<uint8_t>((tpl << 2) | ((stl & 0x300) >> 8));
    eti[n++] = static_cast<uint8_t>(stl & 0xff);
  });
  eti[n++] = 0xff;                                              // MNSC
  eti[n++] = 0xff;
  const uint16_t hcrc = static_cast<uint16_t>(~crc16_ccitt(eti + 4, static_cast<size_t>(n - 4)));
  eti[n++] = static_cast<uint8_t>(hcrc >> 8);
  eti[n++] = static_cast<uint8_t>(hcrc & 0xff);
  return n;
}

// One ETI frame to assemble: the 16 consecutive CIFs starting at `first_cif` (linear CIF
// index within the stream's demodulated TFs) are time de-interleaved; sub-channels of
// `layout` are decoded in SubChId order.
struct EtiJob {
  int32_t first_cif;
  int32_t layout;                  // index into ControlPlane::layouts()
  int32_t header_len;
  int32_t header_off;              // the header's bytes: JobList::header(job)
};
// The jobs of one stream and their header bytes, back to back.  (A fixed 272-byte header inside every record -- room for 64 sub-channels --
// made the control-plane pass write 18 MB per benchmark step for headers of 60 bytes: its time was that memory traffic.)
class JobList {
 public:
  size_t size() const { return jobs_.size(); }
  bool empty() const { return jobs_.empty(); }
  void clear() { jobs_.clear(); bytes_.clear(); }
  void reserve(size_t n) { jobs_.reserve(n); bytes_.reserve(n * 64); }
  std::vector<EtiJob>::const_iterator begin() const { return jobs_.begin(); }
  std::vector<EtiJob>::const_iterator end() const { return jobs_.end(); }
  const EtiJob& operator[](size_t i) const { return jobs_[i]; }
  const uint8_t* header(const EtiJob& j) const { return bytes_.data() + j.header_off; }
  // appends a job; build(dst) writes its header (at most kEtiHeaderMax bytes) and returns the length
  template <class F>
  void emplace(int32_t first_cif, int32_t layout, F&& build)
  {
    const size_t off = bytes_.size();
    bytes_.resize(off + kEtiHeaderMax);
    const int len = build(bytes_.data() + off);
    bytes_.resize(off + static_cast<size_t>(len));
    jobs_.push_back(EtiJob{first_cif, layout, len, static_cast<int32_t>(off)});
  }

 private:
  std::vector<EtiJob> jobs_;
  std::vector<uint8_t> bytes_;
};

// What a signalled multiplex must satisfy for the reference's create_eti (misc.c:218-314) to stay inside its own arrays.  The FIC is protected by a
// 16-bit CRC only: at low SNR a corrupted FIB passes it every few ten thousand FIBs, and fib_parse (fic.c:47-130) validates nothing -- the reference
// then reads past cif_time_deinterleaved[55296] (depuncture.c:84-132 from start_cu * 64), indexes eeptable[] past its 8 rows (fic.c:84) or writes past
// eti[6144] (misc.c:233,246-296): undefined behaviour, no parity target.  A batch engine must not let one such ensemble take the others down: the
// ensemble is flagged and emits no frames while its multiplex is in that state (it never leaves it: sub-channels are only ever added, misc.c:14-21).
enum StreamFault : uint32_t {
  kFaultMuxOverflow = 1,        // header + FIC + sub-channel bytes + trailer exceed 6144 bytes
  kFaultOutsideCif = 2,         // a sub-channel's transmitted bits end beyond CU 863
  kFaultEepOption = 4,          // EEP option > 1 (protection level index >= 8): not in ETSI EN 300 401, past the reference's table
  kFaultSubchSize = 8,          // EEP size below one unit of its protection level (bit rate 0): the reference's frame then carries uninitialised stack
                                // bytes -- viterbi() clears (bits + 7) / 8 bytes, dab_descramble_bytes runs over obytes = 16 of them (misc.c:259-264)
};
inline uint32_t layout_fault(const std::vector<SubChannel>& active, int header_len)
{
  uint32_t f = 0;
  int bytes = header_len + 96;
  for (const SubChannel& sc : active) {
    if (sc.slform && sc.protlev >= 8) { f |= kFaultEepOption; continue; }
    if (sc.slform && sc.bitrate <= 0) { f |= kFaultSubchSize; continue; }
    const PuncturePlan pp = puncture_plan(sc);
    if (sc.start_cu * 64 + pp.coded_bits() > kCifBits) f |= kFaultOutsideCif;
    bytes += (((pp.trellis_steps() - 6) / 8) + 7) & 0xfff8;          // misc.c:259-260: obytes
  }
  if (bytes + 8 > 6144) f |= kFaultMuxOverflow;                       // EOF (4) + TIST (4), misc.c:281-292
  return f;
}

class ControlPlane {
 public:
  ControlPlane()
  {
    ens_.cif_hi = 0xff;            // "CIF counter not latched yet" (dab.c:24-25)
    ens_.cif_lo = 0xff;
  }

  // Feed the decoded FIBs of the next demodulated TF (ordinal = its index among the
  // stream's demodulated TFs).  Appends 0 or 4 jobs.  Mirrors dab_process_frame.
  int on_tf(int ordinal, const uint8_t* fibs, const uint8_t* crc_ok, JobList& jobs)
  {
    int ok_count = 0;
    for (int f = 0; f < 12; ++f) ok_count += crc_ok[f] ? 1 : 0;
    if (ok_count > 0) decode_fibs(tf_info_, fibs, crc_ok);
    if (ok_count == 12) {
      ++okcount_;
      if (okcount_ >= 10 && !locked_) {
        locked_ = true;
        note("Locked\n");                                  // dab.c:51
      }
    } else {
      okcount_ = 0;
      if (locked_) {                 // lock lost: ring is dropped (dab.c:55-61)
        locked_ = false;
        note("Lock lost, resetting ringbuffer\n");       // dab.c:57
        ncifs_ = 0;
        return 0;
      }
    }
    if (!locked_) return 0;

    bool layout_changed = layouts_.empty();
    for_each_slot(tf_info_.present, [&](int i) {
      const SubChannel& s = tf_info_.sub[i];
      SubChannel& d = ens_.sub[i];
      if (d.id != s.id || d.slform != s.slform || d.uep_index != s.uep_index || d.start_cu != s.start_cu ||
          d.size_cu != s.size_cu || d.bitrate != s.bitrate || d.protlev != s.protlev)
        layout_changed = true;
      d = s;
    });
    ens_.present |= tf_info_.present;
    ens_.touched |= tf_info_.present;
    ens_.eid = tf_info_.eid;
    if (ens_.cif_hi == 0xff) { ens_.cif_hi = tf_info_.cif_hi; ens_.cif_lo = tf_info_.cif_lo; }
    if (layout_changed) {
      hdr_valid_ = false;
      std::vector<SubChannel> active;
      for_each_slot(ens_.present & keep_, [&](int i) { active.push_back(ens_.sub[i]); });
      layout_fault_ = layout_fault(active, 8 + 4 * static_cast<int>(active.size()) + 4);
      fault_ |= layout_fault_;
      layouts_.push_back(std::move(active));
    }

    if (ncifs_ < 16) {               // initial fill of the 16-CIF history (dab.c:66-76)
      if (ncifs_ == 0) ring_first_ = 4 * ordinal;
      ncifs_ += 4;
      return 0;
    }
    if (!ens_shown_) {               // the one-time ensemble dump (dab.c:78-82)
      note_ensemble();
      ens_shown_ = true;
    }
    for (int i = 0; i < 4; ++i) {    // emit the oldest CIF, then slide (dab.c:85-95)
      // (a multiplex the reference could not assemble inside its arrays: ring and counter move on, no frame is made -- see StreamFault)
      if (!layout_fault_) jobs.emplace(ring_first_, static_cast<int32_t>(layouts_.size()) - 1, [&](uint8_t* dst) { return frame_header(dst); });
      ++ring_first_;
      if (++ens_.cif_lo == 250) {
        ens_.cif_lo = 0;
        if (++ens_.cif_hi == 20) ens_.cif_hi = 0;
      }
    }
    return layout_fault_ ? 0 : 4;
  }

  bool locked() const { return locked_; }
  uint32_t fault() const { return fault_; }          // StreamFault bits seen so far (sticky)
  const std::vector<std::vector<SubChannel>>& layouts() const { return layouts_; }
  // streaming use: CIF indices are rebased when old TF slots are dropped
  void rebase(int cif_shift) { ring_first_ -= cif_shift; }
  // sub-channel filter: only these SubChIds are listed in the STC, decoded and carried (set before the first frame)
  void set_filter(uint64_t keep) { keep_ = keep; hdr_valid_ = false; }
  // The ETI header of the frame with the current CIF counter (= build_eti_header(dst, ens_, keep_), byte for byte).  Between two changes of the
  // multiplex only FSYNC, FCT and FP move from frame to frame; the rest -- and the header CRC up to the contribution of those two bytes, the
  // CRC being linear over GF(2) -- is kept per layout: a copy, three patches and two table look-ups instead of two walks over the sub-channels and
  // a 56-byte CRC chain per frame (four frames per TF: it was four fifths of the control-plane pass).
  int frame_header(uint8_t* dst)
  {
    if (!hdr_valid_) {
      const uint8_t hi = ens_.cif_hi, lo = ens_.cif_lo;
      ens_.cif_hi = ens_.cif_lo = 0;                       // base: FCT 0, FP 0, even FSYNC
      hdr_len_ = build_eti_header(hdr_base_, ens_, keep_);
      ens_.cif_hi = hi;
      ens_.cif_lo = lo;
      const size_t span = static_cast<size_t>(hdr_len_) - 6;      // the CRC covers bytes 4 .. len - 3
      base_crc_ = crc16_ccitt(hdr_base_ + 4, span);
      // contributions of the 8 FCT bits (CRC byte 0) and the 3 FP bits (CRC byte 2, bits 7..5): 11 chains, the tables are their XOR combinations
      uint8_t delta[kEtiHeaderMax] = {0};
      uint16_t bit_fct[8], bit_fp[3];
      for (int b = 0; b < 8; ++b) {
        delta[0] = static_cast<uint8_t>(1 << b);
        bit_fct[b] = crc16_ccitt(delta, span, 0);
      }
      delta[0] = 0;
      for (int b = 0; b < 3; ++b) {
        delta[2] = static_cast<uint8_t>(0x20 << b);
        bit_fp[b] = crc16_ccitt(delta, span, 0);
      }
      t_fct_[0] = t_fp_[0] = 0;
      for (int v = 1; v < 256; ++v) t_fct_[v] = t_fct_[v & (v - 1)] ^ bit_fct[__builtin_ctz(v)];
      for (int f = 1; f < 8; ++f) t_fp_[f] = t_fp_[f & (f - 1)] ^ bit_fp[__builtin_ctz(f)];
      hdr_valid_ = true;
    }
    std::memcpy(dst, hdr_base_, static_cast<size_t>(hdr_len_));
    const bool odd = ens_.cif_lo & 1;
    dst[1] = odd ? 0xf8 : 0x07;
    dst[2] = odd ? 0xc5 : 0x3a;
    dst[3] = odd ? 0x49 : 0xb6;
    dst[4] = ens_.cif_lo;
    const int fp = (ens_.cif_hi * 250 + ens_.cif_lo) % 8;
    dst[6] = static_cast<uint8_t>(hdr_base_[6] | (fp << 5));
    const uint16_t hcrc = static_cast<uint16_t>(~(base_crc_ ^ t_fct_[ens_.cif_lo] ^ t_fp_[fp]));
    dst[hdr_len_ - 2] = static_cast<uint8_t>(hcrc >> 8);
    dst[hdr_len_ - 1] = static_cast<uint8_t>(hcrc & 0xff);
    return hdr_len_;
  }
  // What the reference prints on stderr for its operator -- "Locked" (dab.c:51), "Lock lost, resetting ringbuffer" (dab.c:57), the one-time ensemble dump
  // (dab.c:78-82 -> dump_ens_info, misc.c:316-328) -- as text, in order, since the last take_log().  dab2eti-hip puts it on stderr; a batch caller may never
  // look (the text of one stream is a few hundred bytes per lock event; capped).
  std::string take_log() { std::string t; t.swap(log_); return t; }
  const std::string& log() const { return log_; }
  // (tests) the current ensemble, to hold frame_header() against build_eti_header()
  const EnsembleInfo& ensemble() const { return ens_; }
  uint64_t filter() const { return keep_; }

 private:
  void note(const char* text)
  {
    if (log_.size() < (size_t(1) << 16)) log_ += text;
  }
  void note_ensemble()                       // dump_ens_info (misc.c:316-328), format strings as there
  {
    char line[160];
    std::snprintf(line, sizeof line, "ENSEMBLE_INFO: EId=0x%04x, CIFCount = %d %d\n", ens_.eid, ens_.cif_hi, ens_.cif_lo);
    note(line);
    for_each_slot(ens_.present, [&](int i) {
      const SubChannel& sc = ens_.sub[i];
      std::snprintf(line, sizeof line, "SubChId=%d, slForm=%d, StartAddress=%d, size=%d, bitrate=%d, ASCTy=0x%02x\n", sc.id, sc.slform, sc.start_cu, sc.size_cu, sc.bitrate,
                    static_cast<unsigned>(sc.ascty));
      note(line);
    });
  }
  std::string log_;
  bool ens_shown_ = false;
  EnsembleInfo tf_info_, ens_;
  bool locked_ = false;
  uint64_t keep_ = ~0ull;
  int okcount_ = 0, ncifs_ = 0, ring_first_ = 0;
  uint32_t fault_ = 0, layout_fault_ = 0;
  std::vector<std::vector<SubChannel>> layouts_;
  bool hdr_valid_ = false;
  int hdr_len_ = 0;
  uint16_t base_crc_ = 0, t_fct_[256], t_fp_[8];
  uint8_t hdr_base_[kEtiHeaderMax];
};

}  // namespace dabhip
