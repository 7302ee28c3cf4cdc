// engine.cpp — host side of the batch engine (see engine.hpp for the pipeline).
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#include "dab_bits.hpp"
#include "dab_tables.hpp"
#include "fifo_view.hpp"
#include "placement.hpp"
#include "../../include/dabhip.h"
#include "kernels.hpp"

namespace dabhip {

namespace {
constexpr int kFftChunkTfs = 4096;                        // spectra buffer: 4096 TF x 1.19 MiB = 4.75 GiB (measured: 1024 -> 4096 shortens K2 by 5 %, launch tails)
constexpr int64_t kMaxDecisionRows = int64_t(48) << 20;   // x 512 B = 24 GiB of survivor decisions per launch
constexpr int kFicWords = kFicBits / 32;                  // 288
constexpr int kMscWords = kMscBits / 32;                  // 6912
constexpr int kCifWords = kCifBits / 32;                  // 1728 words per (logical) CIF row
constexpr int kRowLead = 15;                              // logical rows before a stream's CIF 0 (interleaver depth - 1)

StreamState initial_state()
{
  StreamState st;
  std::memset(&st, 0, sizeof st);
  fifo_reset(st);                                         // empty FIFO, calloc'ed frame buffer (fifo_view.hpp)
  return st;
}

void unpack_bits(const uint32_t* words, int nbits, uint8_t* bytes)
{
  for (int i = 0; i < nbits; ++i) bytes[i] = static_cast<uint8_t>((words[i >> 5] >> (i & 31)) & 1u);
}
void pack_bits(const uint8_t* bytes, int nbits, uint32_t* words)
{
  std::memset(words, 0, static_cast<size_t>(nbits / 32) * 4);
  for (int i = 0; i < nbits; ++i) words[i >> 5] |= static_cast<uint32_t>(bytes[i] & 1u) << (i & 31);
}
}  // namespace

// The single-TF seams and stage entries carry the reference's hard 0/1 bytes (dab.h:27-33): their row strides are those of
// one bit per value.  With soft decisions on, the rows hold four bits per value, so these entry points refuse to run.
bool Engine::hard_only(const char* what)
{
  if (soft_bits_ == 0) return true;
  set_error(std::string(what) + ": not available with soft decisions on (dabhip_engine_set_soft): this entry point carries hard bits");
  return false;
}

bool Engine::check(hipError_t e, const char* what)
{
  if (e == hipSuccess) return true;
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return false;
}
// event records and queries on the decode path: a record that fails would silently corrupt a stage time or -- ev_part0_, ev_chain_ -- an ordering
bool Engine::record(hipEvent_t e, hipStream_t s) { return check(hipEventRecord(e, s), "hipEventRecord"); }
bool Engine::elapsed(float* ms, hipEvent_t a, hipEvent_t b) { return check(hipEventElapsedTime(ms, a, b), "hipEventElapsedTime"); }

Engine::Engine(int device, int host_threads, std::vector<int> cpus) : device_(device), host_cpus_(std::move(cpus))
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device: libdabhip has no CPU fallback"); return; }
  if (device < 0 || device >= ndev) { set_error("device index out of range"); return; }
  if (!check(hipSetDevice(device), "hipSetDevice")) return;
  // The side stream carries the FIC decode (1008 short waves beside the OFDM stage's 16 k workgroups) and the small copies the host waits for:
  // it gets the highest priority, so that the FIBs -- and with them the host control plane -- are not queued behind the bulk of the OFDM stage
  // (DABHIP_SIDE_PRIORITY=0: equal priorities, as before round 3).
  int prio_low = 0, prio_high = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
  static const bool side_prio = !(std::getenv("DABHIP_SIDE_PRIORITY") && std::atoi(std::getenv("DABHIP_SIDE_PRIORITY")) == 0);
  if (!check(hipStreamCreate(&stream_), "hipStreamCreate") ||
      !check(side_prio ? hipStreamCreateWithPriority(&copy_stream_, hipStreamDefault, prio_high) : hipStreamCreate(&copy_stream_), "hipStreamCreate"))
    return;
  for (auto& e : ev_)
    if (!check(hipEventCreate(&e), "hipEventCreate")) return;
  if (!check(hipEventCreate(&ev_upload_), "hipEventCreate") || !check(hipEventCreate(&ev_fic_), "hipEventCreate") || !check(hipEventCreate(&ev_fic_done_), "hipEventCreate") ||
      !check(hipEventCreate(&ev_chain_), "hipEventCreate") || !check(hipEventCreate(&ev_info_), "hipEventCreate") ||
      !check(hipEventCreate(&ev_fibs_), "hipEventCreate") || !check(hipEventCreate(&ev_part0_), "hipEventCreate"))
    return;
  for (auto& e : ev_msc_)
    if (!check(hipEventCreate(&e), "hipEventCreate")) return;
  for (auto& e : ev_h2d_)
    if (!check(hipEventCreate(&e), "hipEventCreate")) return;
  for (auto& e : stage_ev_)
    if (!check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return;
  if (!check(hipStreamCreateWithFlags(&d2h_stream_, hipStreamNonBlocking), "hipStreamCreate") ||
      !check(hipEventCreateWithFlags(&ev_eti_fetch_[0], hipEventDisableTiming), "hipEventCreate") ||
      !check(hipEventCreateWithFlags(&ev_eti_fetch_[1], hipEventDisableTiming), "hipEventCreate"))
    return;

  std::vector<double2> tw2048(2048), tw1536(1536);
  std::vector<float2> twf(2048);
  for (int k = 0; k < 2048; ++k) {
    const double a = 2 * M_PI * k / 2048;
    tw2048[k] = make_double2(std::cos(a), std::sin(a));
    // the quarter points exactly (libm's cos(pi / 2) is 6.1e-17): bins 512 and 1536 of an int8 symbol are Gaussian integers, their differential products can
    // be EXACTLY zero (at low signal levels they are, every few thousand frames), and the fp64 re-decision of such a product must come out as the exact
    // integer arithmetic does -- as the sample-by-sample form happened to, and as any transform with trivial quarter turns (FFTW's codelets) does
    if (k % 512 == 0) tw2048[k] = make_double2(k == 0 ? 1.0 : k == 1024 ? -1.0 : 0.0, k == 512 ? 1.0 : k == 1536 ? -1.0 : 0.0);
    twf[k] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(-std::sin(a)));   // forward kernel
  }
  for (int k = 0; k < 1536; ++k) tw1536[k] = make_double2(std::cos(2 * M_PI * k / 1536), std::sin(2 * M_PI * k / 1536));
  std::vector<uint8_t> prs(prs_quarter_turns().begin(), prs_quarter_turns().end());
  std::vector<uint16_t> qpsk(carrier_to_qpsk().begin(), carrier_to_qpsk().end()), qpsk_inv(kCarriers);
  for (int c = 0; c < kCarriers; ++c) qpsk_inv[qpsk[c]] = static_cast<uint16_t>(c);
  std::vector<uint16_t> crc(256);
  for (int v = 0; v < 256; ++v) {
    const uint8_t b = static_cast<uint8_t>(v);
    crc[v] = crc16_ccitt(&b, 1, 0);
  }
  // CRC shift operators: column b of operator i = CRC register after feeding 2^i zero bytes starting from 1 << b
  std::vector<uint16_t> crc_shift(14 * 16);
  for (int i = 0; i < 14; ++i)
    for (int bit = 0; bit < 16; ++bit) {
      if (i == 0) {
        const uint8_t zero = 0;
        crc_shift[bit] = crc16_ccitt(&zero, 1, static_cast<uint16_t>(1u << bit));
      } else {                               // square the previous operator
        uint16_t v = crc_shift[(i - 1) * 16 + bit], y = 0;
        for (int b = 0; b < 16; ++b)
          if ((v >> b) & 1) y ^= crc_shift[(i - 1) * 16 + b];
        crc_shift[i * 16 + bit] = y;
      }
    }
  std::vector<uint32_t> prbs(1024), zeros(1024, 0u);     // 4096 bytes >= the largest sub-channel (1152 bytes per CIF at 384 kbit/s)
  {
    Prbs g;
    for (auto& w : prbs) {
      uint32_t x = 0;
      for (int b = 0; b < 4; ++b) x |= static_cast<uint32_t>(g.next_byte()) << (8 * b);
      w = x;
    }
  }
  if (!d_tw2048_.upload(tw2048, stream_) || !d_tw1536_.upload(tw1536, stream_) || !d_twf_.upload(twf, stream_) ||
      !d_prs_.upload(prs, stream_) || !d_qpsk_.upload(qpsk, stream_) || !d_qpsk_inv_.upload(qpsk_inv, stream_) || !d_crc_tab_.upload(crc, stream_) || !d_crc_shift_.upload(crc_shift, stream_) ||
      !d_prbs_.upload(prbs, stream_) || !d_zero_words_.upload(zeros, stream_))
    return;
  if (!check(hipStreamSynchronize(stream_), "table upload")) return;
  // host threads for the per-stream control plane: half the CPUs this process may use (its affinity mask capped by the cgroup's CFS quota:
  // placement.hpp usable_cpus(); the machine's thread count says nothing in a container), at most 24, at least 2; DABHIP_HOST_THREADS overrides it
  // (bench.py gives each of N ranks on a node its share, so that 8 ranks do not start 8 full pools)
  const int hw = usable_cpus();
  int nthreads = host_threads > 0 ? std::min(host_threads, 64) : std::max(2, std::min(hw / 2, 24));
  if (const char* env = std::getenv("DABHIP_HOST_THREADS")) nthreads = std::max(1, std::min(64, std::atoi(env)));
  // host placement (placement.hpp): the device's NUMA node; without an explicit CPU list the host threads go to that node's CPUs when the machine
  // has more than one node with CPUs (on a single-socket box there is nothing to choose)
  {
    char bdf[32] = {0};
    if (numa_enabled() && hipDeviceGetPCIBusId(bdf, sizeof bdf, device) == hipSuccess) numa_node_ = numa_node_of_pci(bdf);
    else (void)hipGetLastError();
    if (host_cpus_.empty() && numa_enabled() && numa_node_ >= 0) {
      const std::vector<std::vector<int>> nodes = allowed_node_cpus();       // (a process pinned to one socket sees one populated node: nothing to choose)
      int populated = 0;
      for (const auto& n : nodes) populated += n.empty() ? 0 : 1;
      if (populated > 1 && numa_node_ < static_cast<int>(nodes.size())) host_cpus_ = nodes[static_cast<size_t>(numa_node_)];
    }
  }
  if (const char* env = std::getenv("DABHIP_VIT_WAVE_MAX")) wave_max_codewords_ = wave_max_fic_blocks_ = std::max(0, std::atoi(env));
  if (const char* env = std::getenv("DABHIP_VIT_TWO_LANES")) two_lanes_max_groups_ = std::max(0, std::atoi(env));
  if (const char* env = std::getenv("DABHIP_VIT_FOUR_LANES")) four_lanes_max_groups_ = std::max(0, std::atoi(env));
  if (const char* env = std::getenv("DABHIP_FIC_FOUR_LANES")) fic_four_lanes_max_tiles_ = std::max(0, std::atoi(env));
  if (const char* env = std::getenv("DABHIP_VIT_LANES_PLAIN")) two_lanes_plain_ = std::atoi(env) != 0;
  if (const char* env = std::getenv("DABHIP_FIC_WAVE_MAX")) wave_max_fic_blocks_ = std::max(0, std::atoi(env));
  pool_.reset(new ThreadPool(std::max(0, nthreads - 1), host_cpus_));
  host_lane_.reset(new AsyncLane(host_cpus_));
  ok_ = true;
}

Engine::~Engine()
{
  for (auto& e : ev_)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : chunk_ev_) (void)hipEventDestroy(e);
  if (ev_upload_) (void)hipEventDestroy(ev_upload_);
  if (ev_fic_) (void)hipEventDestroy(ev_fic_);
  if (ev_fic_done_) (void)hipEventDestroy(ev_fic_done_);
  if (ev_chain_) (void)hipEventDestroy(ev_chain_);
  if (ev_info_) (void)hipEventDestroy(ev_info_);
  if (ev_fibs_) (void)hipEventDestroy(ev_fibs_);
  if (ev_part0_) (void)hipEventDestroy(ev_part0_);
  for (auto& e : ev_msc_)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ev_h2d_)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : stage_ev_)
    if (e) (void)hipEventDestroy(e);
  if (d2h_stream_) { (void)hipStreamSynchronize(d2h_stream_); (void)hipStreamDestroy(d2h_stream_); }
  for (hipEvent_t ev : ev_eti_fetch_)
    if (ev) (void)hipEventDestroy(ev);
  if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

// ---------------------------------------------------------------------------------------------
bool Engine::upload_small(const SmallUpload* items, int n, hipStream_t s, SmallStage& stage)
{
  PinnedBuffer<uint32_t>& staging = stage.words;
  size_t total = 0;
  bool words = true;
  for (int i = 0; i < n; ++i) {
    total += items[i].bytes;
    words = words && items[i].bytes % 4 == 0 && reinterpret_cast<uintptr_t>(items[i].src) % 4 == 0;
  }
  if (!words || total > kSmallUploadBytes) {
    for (int i = 0; i < n; ++i)
      if (items[i].bytes && !check(hipMemcpyAsync(items[i].dst, items[i].src, items[i].bytes, hipMemcpyHostToDevice, s), "work list upload")) return false;
    return true;
  }
  if (staging.size() < kSmallUploadBytes / 4 && !staging.resize(kSmallUploadBytes / 4)) return false;   // once: the buffer never moves while a kernel may read it
  // the stage's previous launch reads these words when it runs: still in flight -> wait for it (see engine.hpp; not reached by today's callers)
  if (stage.armed && hipEventQuery(stage.done) == hipErrorNotReady && !check(hipEventSynchronize(stage.done), "work list staging")) return false;
  (void)hipGetLastError();
  if (!stage.done && !check(hipEventCreateWithFlags(&stage.done, hipEventDisableTiming), "hipEventCreate")) return false;
  size_t at = 0;
  HostWordsArgs hw{};
  int k = 0;
  for (int i = 0; i < n; ++i) {
    if (items[i].bytes == 0) continue;
    const uint32_t* src = static_cast<const uint32_t*>(items[i].src);
    if (!items[i].pinned) {
      std::memcpy(staging.data() + at, items[i].src, items[i].bytes);
      src = staging.data() + at;
      at += items[i].bytes / 4;
    }
    hw.src[k] = src;
    hw.dst[k] = static_cast<uint32_t*>(items[i].dst);
    hw.nwords[k] = static_cast<uint32_t>(items[i].bytes / 4);
    if (++k == 4) {
      if (!check(launch_host_words(hw, s), "work list upload")) return false;
      hw = HostWordsArgs{};
      k = 0;
    }
  }
  if (k != 0 && !check(launch_host_words(hw, s), "work list upload")) return false;
  stage.armed = check(hipEventRecord(stage.done, s), "work list staging event");
  return stage.armed;
}

// work lists of a batch to the device (any stream: only the launches below consume them)
bool Engine::upload_decode_batch(const DecodeBatch& b, const HostList<DecodeJob>& jobs, hipStream_t s)
{
  if (b.groups.empty()) return true;
  const int row_words = kCifWords * (soft_bits_ ? 4 : 1);
  const size_t ntiles = b.job_ids.size() / 64;
  return d_plans_.upload(plan_table_.plans(), s) && d_groups_.upload(b.groups, s) && d_job_ids_.upload(b.job_ids, s) && d_jobs_.upload(jobs, s) &&
         d_decisions_.reserve(static_cast<size_t>(b.max_dec_rows) * 64) && d_grouped_.reserve(ntiles * row_words * 64);
}

// regroup + Viterbi over an uploaded batch: queued only; ev_msc_[0..2] bracket the two stages
bool Engine::launch_decode_batch(const DecodeBatch& b, const uint32_t* bits, const int* d_stream_cif_base, const uint32_t* prbs, uint8_t* out,
                                 int record_stride)
{
  if (b.groups.empty()) {                                 // nothing to decode: the three stamps still exist for msc_collect
    if (!record(ev_msc_[0], stream_)) return false;
    if (!record(ev_msc_[1], stream_)) return false;
    if (!record(ev_msc_[2], stream_)) return false;
    return true;
  }
  const int* ids = d_job_ids_.get();
  const int row_words = kCifWords * (soft_bits_ ? 4 : 1);
  const int ntiles = static_cast<int>(b.job_ids.size() / 64);
  if (!record(ev_msc_[0], stream_)) return false;
  if (!check(launch_regroup(soft_bits_, ids, ntiles, d_jobs_.get(), d_stream_cif_base, bits, d_grouped_.get(), stream_), "regroup launch")) return false;
  if (!record(ev_msc_[1], stream_)) return false;
  if (b.wave_form) {
    // small batch: one wave per code word (k_vitwave.hip), all lengths in one launch (longest first); its decisions use the survivor-record buffer
    if (!check(launch_viterbi_wave(soft_bits_, d_groups_.get(), static_cast<int>(b.groups.size()), ids, d_plans_.get(), d_grouped_.get(), row_words,
                                   d_decisions_.get(), prbs, out, record_stride, stream_),
               "viterbi (wave per code word) launch"))
      return false;
    if (!record(ev_msc_[2], stream_)) return false;
    return true;
  }
  // mid-size batches (hard decisions): two lanes per code word (engine.hpp: two_lanes_max_groups_; 1 = always)
  const bool two_lanes = !soft_bits_ && two_lanes_max_groups_ > 0 && (two_lanes_max_groups_ == 1 || static_cast<int>(b.groups.size()) <= two_lanes_max_groups_);
  const bool four_lanes = !soft_bits_ && four_lanes_max_groups_ > 0 && (four_lanes_max_groups_ == 1 || static_cast<int>(b.groups.size()) <= four_lanes_max_groups_);
  for (size_t sl = 0; sl + 1 < b.slice_start.size(); ++sl) {
    const int g0 = b.slice_start[sl], n = b.slice_start[sl + 1] - g0;
    if (four_lanes || (two_lanes && two_lanes_plain_)) {
      if (!check(launch_viterbi_fused_lanes(four_lanes ? 4 : 2, d_groups_.get() + g0, n, ids, d_plans_.get(), d_grouped_.get(), row_words, d_decisions_.get(), prbs, out,
                                            record_stride, stream_),
                 "viterbi (lanes per code word) launch"))
        return false;
      continue;
    }
    if (two_lanes) {
      if (!check(launch_viterbi_fused_two(d_groups_.get() + g0, n, ids, d_plans_.get(), d_grouped_.get(), row_words, d_decisions_.get(), prbs, out, record_stride, stream_),
                 "viterbi (two lanes per code word) launch"))
        return false;
      continue;
    }
    if (!check(launch_viterbi_fused(soft_bits_, d_groups_.get() + g0, n, ids, d_plans_.get(), d_grouped_.get(), row_words, d_decisions_.get(), prbs,
                                    out, record_stride, stream_),
               "viterbi launch"))
      return false;
  }
  if (!record(ev_msc_[2], stream_)) return false;
  return true;
}

// ---------------------------------------------------------------------------------------------
bool Engine::reserve_tf_slots(int nslots, int msc_rows)
{
  if (msc_rows < 0) msc_rows = 4 * nslots + kRowLead + 1;
  // growth discards contents: callers reserve before filling
  const size_t bits = soft_bits_ ? 4 : 1;
  if (nslots > tf_slots_) {
    if (!d_fic_bits_.reserve(static_cast<size_t>(nslots) * kFicWords * bits) || !d_fibs_.reserve(static_cast<size_t>(nslots) * 384) ||
        !d_fib_ok_.reserve(static_cast<size_t>(nslots) * 12))
      return false;
    tf_slots_ = nslots;
  }
  if (msc_rows > msc_rows_) {
    if (!d_msc_bits_.reserve(static_cast<size_t>(msc_rows) * kCifWords * bits)) return false;
    msc_rows_ = msc_rows;
  }
  return true;
}

// S3: host 0/1 bytes of one TF -> FIC row of `slot`, MSC scattered into the planar logical rows (single stream,
// CIF 0 at row kRowLead), the same layout demap_kernel<true> produces.  With soft decisions on the bytes are signed 4-bit
// values (-7 .. 7 as int8; > 0: bit 0) and the rows hold a nibble per value: word u / 8 of a plane, nibble u % 8.
bool Engine::store_tf_bytes(int slot, const uint8_t* fic_bytes, const uint8_t* msc_bytes)
{
  static const int tmap[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  const int bits = soft_bits_ ? 4 : 1, per = 32 / bits;
  const uint32_t vmask = soft_bits_ ? 15u : 1u;
  const size_t fic_words = static_cast<size_t>(kFicWords) * bits, row_words = static_cast<size_t>(kCifWords) * bits, plane_words = 108u * bits;
  std::vector<uint32_t> f(fic_words, 0u), plane(plane_words);
  for (int i = 0; i < kFicBits; ++i) f[i / per] |= (static_cast<uint32_t>(fic_bytes[i]) & vmask) << (bits * (i % per));
  if (!check(blocking_copy(d_fic_bits_.get() + static_cast<size_t>(slot) * fic_words, f.data(), f.size() * 4, hipMemcpyHostToDevice), "fic upload")) return false;
  for (int q = 0; q < 4; ++q) {
    const uint8_t* cif = msc_bytes + static_cast<size_t>(q) * kCifBits;
    for (int r = 0; r < 16; ++r) {
      std::fill(plane.begin(), plane.end(), 0u);
      for (int u = 0; u < kCifBits / 16; ++u) plane[u / per] |= (static_cast<uint32_t>(cif[16 * u + r]) & vmask) << (bits * (u % per));
      const size_t row = static_cast<size_t>(kRowLead + 4 * slot + q - tmap[r]);
      // plane r occupies words [108 r, 108 r + 108) (x 4 with soft values) of the logical row (layout of demap_kernel<true>)
      if (!check(blocking_copy(d_msc_bits_.get() + row * row_words + r * plane_words, plane.data(), plane_words * 4, hipMemcpyHostToDevice), "msc upload")) return false;
    }
  }
  return true;
}

bool Engine::recycle_tf_slots(int used_slots, int keep_slots)
{
  // FIC rows / FIB records: the newest keep_slots; logical CIF rows: everything from 15 rows before the oldest kept CIF
  const size_t bits = soft_bits_ ? 4 : 1;
  const int src_slot = used_slots - keep_slots;
  const size_t row_src = static_cast<size_t>(4 * src_slot), nrows = static_cast<size_t>(4 * keep_slots + kRowLead);
  if (!d_bytes_.reserve(std::max(nrows * kCifWords * 4 * bits, static_cast<size_t>(keep_slots) * kFicWords * 4 * bits))) return false;
  auto mv = [&](void* base, size_t unit, size_t src, size_t n) {
    uint8_t* b = static_cast<uint8_t*>(base);
    return check(blocking_copy(d_bytes_.get(), b + src * unit, n * unit, hipMemcpyDeviceToDevice), "slot move") &&
           check(blocking_copy(b, d_bytes_.get(), n * unit, hipMemcpyDeviceToDevice), "slot move");
  };
  return mv(d_fic_bits_.get(), kFicWords * 4 * bits, src_slot, keep_slots) && mv(d_fibs_.get(), 384, src_slot, keep_slots) &&
         mv(d_fib_ok_.get(), 12, src_slot, keep_slots) && mv(d_msc_bits_.get(), kCifWords * 4 * bits, row_src, nrows);
}

// What the OFDM stage of the LAST decode() / feed() left for transmission frame `tf` (0-based among the stream's TF slots of that
// decode, carried slots of a session first) of `stream`: the content of tf->fic_symbols_demapped / msc_symbols_demapped (dab.h:27-33)
// as the batch path holds it -- FIC row in natural order, MSC values gathered back out of the planar logical rows the demapper
// scattered them into.  Hard decisions: 0 / 1; soft decisions: the signed 4-bit values.
bool Engine::read_demapped_tf(int stream, int tf, int8_t* fic_out, int8_t* msc_out)
{
  static const int tmap[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  if (stream < 0 || stream >= nstreams_ || static_cast<int>(prev_tf_base_.size()) <= stream || tf < 0 || tf >= prev_used_[stream]) {
    set_error("demapped_tf: no such stream / transmission frame in the last decode");
    return false;
  }
  if (!check(hipSetDevice(device_), "hipSetDevice")) return false;
  const int bits = soft_bits_ ? 4 : 1, per = 32 / bits;
  const size_t fic_words = static_cast<size_t>(kFicWords) * bits, row_words = static_cast<size_t>(kCifWords) * bits, plane_words = 108u * bits;
  std::vector<uint32_t> f(fic_words), rows(static_cast<size_t>(kRowLead + 4) * row_words);
  const size_t slot = static_cast<size_t>(prev_tf_base_[stream]) + tf, row0 = static_cast<size_t>(prev_row_base_[stream]) + 4 * tf - kRowLead;
  if (!check(blocking_copy(f.data(), d_fic_bits_.get() + slot * fic_words, f.size() * 4, hipMemcpyDeviceToHost), "fic download") ||
      !check(blocking_copy(rows.data(), d_msc_bits_.get() + row0 * row_words, rows.size() * 4, hipMemcpyDeviceToHost), "msc download"))
    return false;
  auto value = [&](uint32_t w, int k) -> int8_t {
    const uint32_t v = (w >> (bits * k)) & (soft_bits_ ? 15u : 1u);
    return static_cast<int8_t>(soft_bits_ ? static_cast<int>(v ^ 8u) - 8 : static_cast<int>(v));
  };
  for (int i = 0; i < kFicBits; ++i) fic_out[i] = value(f[i / per], i % per);
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < kCifBits; ++i) {
      const int r = i & 15, u = i >> 4;
      const size_t row = static_cast<size_t>(kRowLead + q - tmap[r]);                 // transmitted CIF q of this TF: plane r lives tmap[r] rows earlier
      msc_out[static_cast<size_t>(q) * kCifBits + i] = value(rows[row * row_words + r * plane_words + u / per], u % per);
    }
  return true;
}

// S2 / stage_demap: the TF was demapped in NATURAL order (demap_kernel<false>) into FIC slot `slot`, CIF rows 4*slot..
bool Engine::unpack_tf_slot(int slot, uint8_t* fic_bytes, uint8_t* msc_bytes)
{
  if (!hard_only("unpack_tf_slot")) return false;
  std::vector<uint32_t> f(kFicWords), m(kMscWords);
  if (!check(blocking_copy(f.data(), d_fic_bits_.get() + static_cast<size_t>(slot) * kFicWords, f.size() * 4, hipMemcpyDeviceToHost), "fic download") ||
      !check(blocking_copy(m.data(), d_msc_bits_.get() + static_cast<size_t>(slot) * kMscWords, m.size() * 4, hipMemcpyDeviceToHost), "msc download"))
    return false;
  unpack_bits(f.data(), kFicBits, fic_bytes);
  unpack_bits(m.data(), kMscBits, msc_bytes);
  return true;
}

bool Engine::fic_decode_slots(int first, int n, uint8_t* fibs_host, uint8_t* ok_host)
{
  return fic_decode_slots_async(first, n, fibs_host, ok_host, stream_) && check(hipStreamSynchronize(stream_), "fic decode");
}

// the same without waiting: kernels on the main stream, the FIB / flag download on `copy` (ordered after them by an event)
bool Engine::fic_decode_slots_async(int first, int n, uint8_t* fibs_host, uint8_t* ok_host, hipStream_t copy)
{
  if (n <= 0) return true;
  const int bits = soft_bits_ ? 4 : 1;
  const int pid = plan_table_.id(make_codeword_plan(fic_plan(), 0, 0));
  // record i of this call = FIC block 4 * first + i; 64 blocks per wave, interleaved word by word by fic_group_kernel
  const int nblocks = 4 * n, ntiles = (nblocks + 63) / 64, block_words = 72 * bits;
  std::vector<int> ids(static_cast<size_t>(ntiles) * 64, -1);
  for (int i = 0; i < nblocks; ++i) ids[i] = 4 * first + i;
  std::vector<WaveGroup> groups;
  const bool wave_form = nblocks <= wave_max_fic_blocks_;      // few blocks: one wave per block (k_vitwave.hip), rows per block and chunk of steps
  // more, but not enough to fill the device with one lane per block (774 dependent steps in front of the control plane): four lanes per block
  // (vit_four_lanes.hpp; same records, same arguments), up to 128 tiles = 32 streams x 64 TF (measured: nothing to gain above).
  const bool fic_four_lanes = !wave_form && !soft_bits_ && fic_four_lanes_max_tiles_ > 0 && (fic_four_lanes_max_tiles_ == 1 || ntiles <= fic_four_lanes_max_tiles_);
  const int64_t dr = wave_form ? int64_t(64) * ((plan_table_[pid].nsteps + kWaveChunk - 1) / kWaveChunk) : (plan_table_[pid].nsteps + 7) / 8 * 8;
  for (int g = 0; g < ntiles; ++g) groups.push_back(WaveGroup{pid, 64 * g, std::min(64, nblocks - 64 * g), plan_table_[pid].nsteps, 0, g * dr});
  // The FIC kernels run on the side stream as well, behind what the main stream has queued so far (the FIC bits): 1008 waves of 774
  // steps fill a quarter of the chip's wave slots for 0.3 ms, so the main stream goes straight on with the rest of the OFDM stage
  // and the two share the GPU.  Everything that later touches these buffers on the main stream waits for the side stream
  // (ev_upload_ in decode_impl, the synchronising callers elsewhere).
  hipStream_t ks = copy;
  if (ks != stream_ && (!check(hipEventRecord(ev_fic_, stream_), "fic event") || !check(hipStreamWaitEvent(ks, ev_fic_, 0), "fic event"))) return false;
  {
    const std::vector<CodewordPlan>& plans = plan_table_.plans();
    if (!d_plans_.reserve(plans.size()) || !d_groups_.reserve(groups.size()) || !d_job_ids_.reserve(ids.size()) ||
        !d_grouped_.reserve(static_cast<size_t>(ntiles) * block_words * 64) || !d_decisions_.reserve(static_cast<size_t>(ntiles) * dr * 64))
      return false;
    const SmallUpload items[3] = {{plans.data(), d_plans_.get(), plans.size() * sizeof(CodewordPlan), false},
                                  {groups.data(), d_groups_.get(), groups.size() * sizeof(WaveGroup), false},
                                  {ids.data(), d_job_ids_.get(), ids.size() * sizeof(int), false}};
    if (!upload_small(items, 3, ks, h_small_fic_)) return false;
  }
  if (!check(launch_fic_group(d_fic_bits_.get(), 4 * first, nblocks, block_words, d_grouped_.get(), ks), "fic group launch") ||
      !check(wave_form
                 ? launch_viterbi_wave(soft_bits_, d_groups_.get(), ntiles, d_job_ids_.get(), d_plans_.get(), d_grouped_.get(), block_words,
                                       d_decisions_.get(), d_prbs_.get(), d_fibs_.get(), 96, ks)
             : fic_four_lanes
                 ? launch_viterbi_fused_lanes(4, d_groups_.get(), ntiles, d_job_ids_.get(), d_plans_.get(), d_grouped_.get(), block_words, d_decisions_.get(),
                                              d_prbs_.get(), d_fibs_.get(), 96, ks)
                 : launch_viterbi_fused(soft_bits_, d_groups_.get(), ntiles, d_job_ids_.get(), d_plans_.get(), d_grouped_.get(), block_words,
                                        d_decisions_.get(), d_prbs_.get(), d_fibs_.get(), 96, ks),
             "fic viterbi launch"))
    return false;
  if (!check(launch_fib_crc(d_fibs_.get() + static_cast<size_t>(first) * 384, n * 12, d_crc_tab_.get(), d_fib_ok_.get() + static_cast<size_t>(first) * 12, ks), "fib crc launch")) return false;
  if (!check(hipEventRecord(ev_fic_done_, ks), "fic event")) return false;
  // (few frames, page-locked destinations -- the engine's own: both downloads as one kernel that writes the host arrays itself, see scan_streams' fetch)
  if (n <= 512 && fibs_host == h_fibs_.data() && ok_host == h_fib_ok_.data()) {
    HostWordsArgs hw{};
    hw.src[0] = reinterpret_cast<const uint32_t*>(d_fibs_.get() + static_cast<size_t>(first) * 384);
    hw.dst[0] = reinterpret_cast<uint32_t*>(fibs_host);
    hw.nwords[0] = static_cast<uint32_t>(n) * 96;
    hw.src[1] = reinterpret_cast<const uint32_t*>(d_fib_ok_.get() + static_cast<size_t>(first) * 12);
    hw.dst[1] = reinterpret_cast<uint32_t*>(ok_host);
    hw.nwords[1] = static_cast<uint32_t>(n) * 3;
    return check(launch_host_words(hw, copy), "fib download") && check(hipEventRecord(ev_fibs_, copy), "fib download event");
  }
  return check(hipMemcpyAsync(fibs_host, d_fibs_.get() + static_cast<size_t>(first) * 384, static_cast<size_t>(n) * 384, hipMemcpyDeviceToHost, copy), "fib download") &&
         check(hipMemcpyAsync(ok_host, d_fib_ok_.get() + static_cast<size_t>(first) * 12, static_cast<size_t>(n) * 12, hipMemcpyDeviceToHost, copy), "fib flag download") &&
         check(hipEventRecord(ev_fibs_, copy), "fib download event");
}

bool Engine::msc_prepare(const std::vector<const JobList*>& stream_jobs, const std::vector<const ControlPlane*>& planes,
                         const std::vector<int>& stream_row_base, const std::vector<int>& stream_fib_base, MscWork& out)
{
  static const bool trace_host = std::getenv("DABHIP_TRACE_HOST") != nullptr;
  const auto t_in = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (trace_host) std::fprintf(stderr, "[host]   msc_prepare %-14s %8.3f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_in).count());
  };
  std::string error;
  if (prepare_msc_work(plan_table_, *pool_, stream_jobs, planes, stream_row_base, stream_fib_base, kMaxDecisionRows, out, &error, mark, wave_max_codewords_)) return true;
  set_error(error);
  return false;
}

// work lists, ETI header bytes and frame records of a prepared batch to the device; `s` may be a side stream
bool Engine::msc_upload(const MscWork& w, hipStream_t s)
{
  if (w.nframes == 0) return true;
  if (!d_eti_.reserve(w.nframes * kEtiBytes)) return false;
  const DecodeBatch& b = w.batch;
  if (b.groups.empty())
    return d_meta_.upload(w.meta, s) && d_headers_.upload(w.headers, s) && d_stream_cif_base_.upload(w.stream_row_base, s);
  const std::vector<CodewordPlan>& plans = plan_table_.plans();
  const int row_words = kCifWords * (soft_bits_ ? 4 : 1);
  const size_t ntiles = b.job_ids.size() / 64;
  if (!d_meta_.reserve(w.meta.size()) || !d_headers_.reserve(w.headers.size()) || !d_stream_cif_base_.reserve(w.stream_row_base.size()) ||
      !d_plans_.reserve(plans.size()) || !d_groups_.reserve(b.groups.size()) || !d_job_ids_.reserve(b.job_ids.size()) || !d_jobs_.reserve(w.jobs.size()) ||
      !d_decisions_.reserve(static_cast<size_t>(b.max_dec_rows) * 64) || !d_grouped_.reserve(ntiles * row_words * 64))
    return false;
  // (the work lists are page-locked vectors -- MscWork --, the plan table and the row bases plain ones)
  const SmallUpload items[7] = {{w.meta.data(), d_meta_.get(), w.meta.size() * sizeof(EtiFrameMeta), true},
                                {w.headers.data(), d_headers_.get(), w.headers.size(), true},
                                {w.stream_row_base.data(), d_stream_cif_base_.get(), w.stream_row_base.size() * sizeof(int), false},
                                {plans.data(), d_plans_.get(), plans.size() * sizeof(CodewordPlan), false},
                                {b.groups.data(), d_groups_.get(), b.groups.size() * sizeof(WaveGroup), true},
                                {b.job_ids.data(), d_job_ids_.get(), b.job_ids.size() * sizeof(int), true},
                                {w.jobs.data(), d_jobs_.get(), w.jobs.size() * sizeof(DecodeJob), true}};
  return upload_small(items, 7, s, h_small_msc_);
}

// K4 + K5 queued on the main stream (nothing is awaited: the caller does that once, then msc_collect() reads the events)
bool Engine::msc_launch_async(const MscWork& w)
{
  const size_t nf = w.nframes;
  msc_queued_ = false;
  if (nf == 0) return true;
  // a fetch of the previous decode's frames may still be reading the ETI buffer these launches rewrite (eti_fetch_async)
  // (the newest fetch's event: the copies run in order on one stream.  Only while a fetch is outstanding: one that has been waited for has landed, and
  // a wait packet costs 10 .. 15 us of idle GPU in front of K4)
  if (const uint64_t issued = eti_fetch_issued_.load(); issued != eti_fetch_waited_.load())
    if (!check(hipStreamWaitEvent(stream_, ev_eti_fetch_[(issued - 1) & 1], 0), "eti fetch wait")) return false;
  if (!launch_decode_batch(w.batch, d_msc_bits_.get(), d_stream_cif_base_.get(), d_prbs_.get(), d_eti_.get(), kEtiBytes)) return false;
  if (!check(launch_eti_finish(d_meta_.get(), static_cast<int>(nf), d_headers_.get(), w.header_stride, d_fibs_.get(), d_crc_tab_.get(), d_crc_shift_.get(), d_eti_.get(), stream_), "eti finish launch"))
    return false;
  if (!record(ev_msc_[3], stream_)) return false;
  msc_queued_ = true;
  return true;
}

void Engine::msc_collect()
{
  if (!msc_queued_) return;
  msc_queued_ = false;
  float ms = 0;
  if (elapsed(&ms, ev_msc_[0], ev_msc_[1])) times_.gather += ms;
  if (elapsed(&ms, ev_msc_[1], ev_msc_[2])) times_.viterbi += ms;
  if (elapsed(&ms, ev_msc_[2], ev_msc_[3])) times_.eti += ms;
}

bool Engine::msc_launch(const MscWork& w)
{
  std::unique_lock<std::mutex> heavy;
  if (heavy_mu_) heavy = std::unique_lock<std::mutex>(*heavy_mu_);
  if (!msc_launch_async(w) || !check(hipStreamSynchronize(stream_), "msc decode")) return false;
  msc_collect();
  return true;
}

bool Engine::msc_run(MscWork& w) { return msc_upload(w, stream_) && msc_launch(w); }

bool Engine::read_eti(int64_t first, int64_t n, uint8_t* dst)
{
  if (n <= 0) return true;
  if (!check(hipSetDevice(device_), "hipSetDevice")) return false;   // callers may sit on another device's thread (dabhip_multi)
  // on the download stream and ended by a stream synchronise (engine.hpp: blocking_copy): callers read after every decode or segment, for ever
  return check(hipMemcpyAsync(dst, d_eti_.get() + first * kEtiBytes, static_cast<size_t>(n) * kEtiBytes, hipMemcpyDeviceToHost, d2h_stream_), "eti download") &&
         check(hipStreamSynchronize(d2h_stream_), "eti download");
}

// ---------------------------------------------------------------------------------------------
// Session carry-over: the FIC blocks, FIBs and CRC flags of each stream's last carry_keep_ TF slots and its logical CIF rows
// from 15 before the oldest kept CIF move from the previous segment's layout to the front of the stream's part of the new
// one (through a dense temporary: the buffers may be re-allocated in between).
bool Engine::carry_and_reserve(const std::vector<int>& tf_base, const std::vector<int>& row_base, int nslots, int nrows)
{
  const size_t bits = soft_bits_ ? 4 : 1;
  const size_t unit[4] = {kFicWords * 4 * bits, 384, 12, kCifWords * 4 * bits};
  HostList<CopyDesc>&out = carry_out_descs_, &in = carry_in_descs_;   // (the previous segment's lists were consumed before its feed returned)
  out.clear();
  in.clear();
  size_t tmp_bytes = 0;
  const int n = static_cast<int>(carry_keep_.size());
  uint8_t* base[4] = {reinterpret_cast<uint8_t*>(d_fic_bits_.get()), d_fibs_.get(), d_fib_ok_.get(), reinterpret_cast<uint8_t*>(d_msc_bits_.get())};
  struct Piece { int which; size_t src, dst, bytes, tmp; };
  std::vector<Piece> pieces;
  for (int b = 0; b < n; ++b) {
    const int keep = carry_keep_[b];
    if (keep == 0) continue;
    const size_t src_slot = static_cast<size_t>(prev_tf_base_[b]) + prev_used_[b] - keep, dst_slot = tf_base[b];
    for (int w = 0; w < 3; ++w) {
      pieces.push_back(Piece{w, src_slot * unit[w], dst_slot * unit[w], keep * unit[w], tmp_bytes});
      tmp_bytes += (keep * unit[w] + 15) & ~size_t(15);
    }
    const size_t src_row = static_cast<size_t>(prev_row_base_[b]) - kRowLead + 4 * (prev_used_[b] - keep), dst_row = static_cast<size_t>(row_base[b]) - kRowLead;
    const size_t rows = 4 * static_cast<size_t>(keep) + kRowLead;
    pieces.push_back(Piece{3, src_row * unit[3], dst_row * unit[3], rows * unit[3], tmp_bytes});
    tmp_bytes += rows * unit[3];
  }
  if (!pieces.empty()) {
    if (!d_carry_.reserve(tmp_bytes)) return false;
    for (const Piece& p : pieces) out.push_back(CopyDesc{base[p.which] + p.src, d_carry_.get() + p.tmp, static_cast<uint32_t>(p.bytes)});
    // (the copy out must be over before a growing buffer is given back; when nothing grows -- every segment of a session but the first few --
    // stream order alone keeps the two copies apart, and the host does not wait)
    const bool grows = nslots > tf_slots_ || (nrows < 0 ? 4 * nslots + kRowLead + 1 : nrows) > msc_rows_;
    if (!d_copy_descs_.upload(out, stream_) || !check(launch_batched_copy(d_copy_descs_.get(), static_cast<int>(out.size()), stream_), "carry out") ||
        (grows && !check(hipStreamSynchronize(stream_), "carry out")))
      return false;
  }
  if (!reserve_tf_slots(nslots, nrows)) return false;
  if (!pieces.empty()) {
    uint8_t* nbase[4] = {reinterpret_cast<uint8_t*>(d_fic_bits_.get()), d_fibs_.get(), d_fib_ok_.get(), reinterpret_cast<uint8_t*>(d_msc_bits_.get())};
    for (const Piece& p : pieces) in.push_back(CopyDesc{d_carry_.get() + p.tmp, nbase[p.which] + p.dst, static_cast<uint32_t>(p.bytes)});
    if (!d_copy_descs_.upload(in, stream_) || !check(launch_batched_copy(d_copy_descs_.get(), static_cast<int>(in.size()), stream_), "carry in"))
      return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Parity guard around one demapping launch: guard_begin() hands the kernel its list and its counter (all counters of a decode are cleared
// before the first launch); guard_finish() queues the fp64 re-decision of what was listed; guard_download() (once, behind the last launch)
// copies the entry counts to the host; guard_check() (after the stream has been awaited) adds them up and counts the launches whose
// list overflowed (those were decided again in full).
constexpr int kGuardMinLaunches = 64;
constexpr int kGuardSlotWords = 4;                       // a launch's counter and three spare words
bool Engine::guard_begin(int ntf_in_launch, GuardArgs* out)
{
  // the list: flag rates measured on noisy input are a few decisions per TF (7e-6 of 230,400 at 5 dB); 64 entries per TF, at least
  // 256 K, and a launch that overflows it is decided again in full (exact_decide_all_kernel) instead of failing
  // (the proven level's band is 13 x as wide: 16 x the entries)
  const int level = guard_rule_level();
  uint32_t cap = static_cast<uint32_t>(std::min<int64_t>(int64_t(1) << 30, std::max<int64_t>(int64_t(1) << (level >= 2 ? 20 : 18), static_cast<int64_t>(ntf_in_launch) * (level >= 2 ? 512 : 64))));
  if (guard_cap_override_) cap = guard_cap_override_;
  if (guard_launches_ == 0 && h_guard_counts_.size() < static_cast<size_t>(kGuardMinLaunches) * kGuardSlotWords && !h_guard_counts_.resize(static_cast<size_t>(kGuardMinLaunches) * kGuardSlotWords)) return false;
  if (static_cast<size_t>(guard_launches_ + 1) * kGuardSlotWords > h_guard_counts_.size()) { set_error("parity guard: more guarded launches than planned for in one decode"); return false; }
  if (!d_guard_list_.reserve(cap) || !d_guard_counter_.reserve(h_guard_counts_.size())) return false;
  guard_cap_ = guard_cap_override_ ? guard_cap_override_ : static_cast<uint32_t>(std::min<size_t>(d_guard_list_.capacity(), 0xffffffffu));
  // every launch of a decode has its own counter: ONE clear before the first and ONE download behind the last (guard_download) instead of a
  // clear and a download per launch (small copy-engine operations cost 20 .. 35 us of idle GPU each between two kernels); a decode's layout
  // kernel has normally cleared them already (guard_counters_clear_)
  if (guard_launches_ == 0 && !guard_counters_clear_ &&
      !check(hipMemsetAsync(d_guard_counter_.get(), 0, h_guard_counts_.size() * sizeof(uint32_t), stream_), "guard counters"))
    return false;
  guard_counters_clear_ = false;
  // the capacity THIS launch was given (a later launch of the same decode may find the list re-reserved and larger): guard_check compares with it
  if (guard_caps_.size() <= static_cast<size_t>(guard_launches_)) guard_caps_.resize(static_cast<size_t>(guard_launches_) + 1);
  guard_caps_[static_cast<size_t>(guard_launches_)] = guard_cap_;
  *out = GuardArgs{d_delta_.get(), kSymbolsPerTf, guard_c_of(level), guard_prod_of(level), level >= 2 ? 1 : 0, guard_cap_, d_guard_list_.get(),
                   d_guard_counter_.get() + static_cast<size_t>(guard_launches_) * kGuardSlotWords};
  return true;
}
// the counters' host and device arrays for a decode of ntf frames (the layout kernel clears the device side)
bool Engine::guard_reserve_counters(int ntf)
{
  const size_t words = (static_cast<size_t>(kGuardMinLaunches) + 2 * static_cast<size_t>(ntf / kFftChunkTfs + 1)) * kGuardSlotWords;
  return (h_guard_counts_.size() >= words || h_guard_counts_.resize(words)) && d_guard_counter_.reserve(h_guard_counts_.size());
}
bool Engine::guard_finish(bool planar, int first, int n, int sym_a, int sym_b, bool skip_fic)
{
  uint32_t* const counter = d_guard_counter_.get() + static_cast<size_t>(guard_launches_) * kGuardSlotWords;    // guard_begin's
  ++guard_launches_;
  return check(launch_exact_decide(d_guard_list_.get(), counter, guard_cap_, d_iq_ptrs_.get(), d_descs_.get(), max_calls_, d_frames_.get(),
                                   d_tw2048_.get(), d_qpsk_.get(), d_qpsk_inv_.get(), d_frame_slot_.get(), d_frame_cif_row_.get(), planar, d_fic_bits_.get(), d_msc_bits_.get(), stream_),
               "exact decide launch") &&
         check(launch_exact_decide_all(counter, guard_cap_, d_iq_ptrs_.get(), d_descs_.get(), max_calls_, d_frames_.get(), first, n, sym_a, sym_b,
                                       d_tw2048_.get(), d_qpsk_.get(), d_frame_slot_.get(), d_frame_cif_row_.get(), planar, skip_fic, d_fic_bits_.get(), d_msc_bits_.get(), stream_),
               "exact decide (overflow) launch");
}
// behind the last guarded launch of a decode, before the stream is awaited
bool Engine::guard_download()
{
  return guard_launches_ == 0 ||
         check(hipMemcpyAsync(h_guard_counts_.data(), d_guard_counter_.get(), static_cast<size_t>(guard_launches_) * kGuardSlotWords * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_),
               "guard count download");
}
bool Engine::guard_check()
{
  for (int i = 0; i < guard_launches_; ++i) {
    const uint32_t count = h_guard_counts_[static_cast<size_t>(i) * kGuardSlotWords];
    if (count > guard_caps_[static_cast<size_t>(i)]) ++guard_overflows_;       // that launch was decided again in full: still exact, only slow
    guard_flagged_ += count;
  }
  guard_launches_ = 0;
  return true;
}

int64_t Engine::decode(const uint8_t* const* iq, const size_t* nbytes, int nstreams, bool on_device)
{
  return decode_impl(iq, nbytes, nstreams, on_device, false);
}

// Streaming continuation: iq[b] are DEVICE pointers positioned so that iq[b][x] is byte x of stream b counted from the
// start of the session (only the bytes stream_need_from(b) .. avail[b] have to be backed by memory); avail[b] = bytes
// of the stream received so far.  Decodes the calls that became complete since the previous segment; state carried:
// the front-end state (K1), the lock / CIF-ring state of the control plane, the FIC blocks and FIBs of the last 4 TFs
// and the partly filled logical CIF rows.  The ETI frames of all segments concatenated equal those of one decode().
int64_t Engine::feed(const uint8_t* const* iq, const size_t* avail, int nstreams, bool first_segment)
{
  return decode_impl(iq, avail, nstreams, true, !first_segment);
}

int64_t Engine::stream_need_from(int b) const
{
  if (b < 0 || b >= static_cast<int>(h_states_.size())) return 0;
  const StreamState& st = h_states_[b];
  int64_t need = st.consumed;
  for (int i = 0; i < st.view.nseg; ++i)
    if (st.view.seg_src[i] >= 0) need = std::min(need, st.view.seg_src[i] + (i ? st.view.seg_end[i - 1] : 0));
  return std::max<int64_t>(need, 0) & ~int64_t(1);
}

// session bookkeeping at the start of a decode (fresh) or of a further segment (cont)
bool Engine::begin_decode(int nstreams, bool cont)
{
  if (cont && (nstreams != nstreams_ || static_cast<int>(planes_.size()) != nstreams)) { set_error("feed: the number of streams changed within a session"); return false; }
  nstreams_ = nstreams;
  eti_base_.assign(nstreams, 0);
  eti_count_.assign(nstreams, 0);
  if (!cont || static_cast<int>(stream_status_.size()) != nstreams) stream_status_.assign(nstreams, 0);
  total_eti_ = 0;
  if (!cont) {
    planes_.resize(nstreams);                  // re-initialised by the control-plane pass itself (planes_fresh_): 0.2 ms that would otherwise delay K1
    planes_fresh_ = true;
    carry_keep_.assign(nstreams, 0);
    prev_used_.assign(nstreams, 0);
    calls_done_.assign(nstreams, 0);
    ord_done_.assign(nstreams, 0);
    prev_tf_base_.assign(nstreams + 1, 0);
    prev_row_base_.assign(nstreams, 0);
  }
  return true;
}

// Host-fed decode (the reference's input arrives in host buffers: dab2eti.c:117-130,238).  Streams that live in page-locked memory
// (dabhip_host_alloc, hipHostMalloc / hipHostRegister of the caller's own) go up as plain asynchronous DMA, one copy per stream,
// back to back on the main stream.  Pageable memory cannot be DMA'd from: it is copied into a ring of page-locked staging buffers
// by the engine's host pool (all threads on one piece: a single core's memcpy is slower than the PCIe link) and each piece leaves
// as its own asynchronous copy, so the pool fills one buffer while up to three others drain.  K1 follows in stream order.
bool Engine::upload_iq(const uint8_t* const* iq, const size_t* nbytes, int nstreams, const uint8_t** ptrs)
{
  constexpr size_t kStageBytes = size_t(32) << 20, kPiece = size_t(1) << 20;
  if (!record(ev_h2d_[0], stream_)) return false;
  size_t off = 0;
  int next_buf = 0;
  size_t fill = 0;                             // bytes staged in the current buffer, not yet queued
  size_t fill_dst = 0;                         // device offset the current buffer's bytes go to (streams are laid out back to back)
  auto flush = [&]() -> bool {
    if (fill == 0) return true;
    const bool ok = check(hipMemcpyAsync(d_iq_own_.get() + fill_dst, stage_buf_[next_buf].data(), fill, hipMemcpyHostToDevice, stream_), "IQ upload") &&
                    check(hipEventRecord(stage_ev_[next_buf], stream_), "IQ upload");
    next_buf = (next_buf + 1) % kStageBufs;
    fill = 0;
    return ok;
  };
  for (int b = 0; b < nstreams; ++b) {
    uint8_t* const dst = d_iq_own_.get() + off;
    ptrs[b] = dst;
    const size_t n = nbytes[b], padded = (n + 15) & ~size_t(15);
    // memory the runtime knows (page-locked / registered host memory; also device or managed memory handed in by mistake as "host") is copied
    // by the copy engine directly; everything else is ordinary pageable memory
    hipPointerAttribute_t attr;
    const bool pinned = n && hipPointerGetAttributes(&attr, iq[b]) == hipSuccess &&
                        (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
    if (!pinned) (void)hipGetLastError();      // an unregistered pointer is reported as an error: that is the answer, not a failure
    times_.h2d_bytes += static_cast<double>(n);
    if (pinned) {
      if (!flush()) return false;              // keeps the copies in stream order (cheap: at most one partly filled buffer)
      times_.h2d_pinned_bytes += static_cast<double>(n);
      if (!check(hipMemcpyAsync(dst, iq[b], n, hipMemcpyDefault, stream_), "IQ upload")) return false;
    } else {
      // staged: a stream's bytes continue in the buffer where the previous stream's ended only when they are adjacent on the device
      // (they are, up to the 16-byte padding: a buffer is flushed at a stream boundary when the padding is not zero)
      size_t done = 0;
      while (done < n) {
        if (fill == 0) {
          if (!stage_buf_[next_buf].resize(kStageBytes)) return false;
          if (!check(hipEventSynchronize(stage_ev_[next_buf]), "IQ staging")) return false;   // its previous copy has left (never recorded: returns at once)
          fill_dst = off + done;
        }
        const size_t take = std::min(n - done, kStageBytes - fill);
        const uint8_t* src = iq[b] + done;
        uint8_t* stage = stage_buf_[next_buf].data() + fill;
        const int pieces = static_cast<int>((take + kPiece - 1) / kPiece);
        pool_->parallel_for(pieces, [&](int i) {
          const size_t a = static_cast<size_t>(i) * kPiece;
          std::memcpy(stage + a, src + a, std::min(kPiece, take - a));
        });
        fill += take;
        done += take;
        if (fill == kStageBytes && !flush()) return false;
      }
      if (padded != n && !flush()) return false;
    }
    off += padded;
  }
  if (!flush()) return false;
  if (!record(ev_h2d_[1], stream_)) return false;
  return true;
}

// K1 over the calls that became complete: stages pointers / sizes / states, launches the scan, brings back {status, ordinal}
// per call and the front-end states (main stream, awaited) and the full descriptors (side stream, awaited by the caller's guard)
bool Engine::scan_streams(const uint8_t* const* iq, const size_t* nbytes, int nstreams, bool on_device, bool cont, bool full_scan,
                          const std::function<bool()>& layout)
{
  const auto wall0 = std::chrono::steady_clock::now();
  if (!h_ptrs_.resize(nstreams) || !h_nb_.resize(nstreams)) return false;   // page-locked staging: asynchronous uploads
  const uint8_t** const ptrs = h_ptrs_.data();
  int64_t* const nb = h_nb_.data();
  max_calls_ = 1;
  size_t total = 0;
  for (int b = 0; b < nstreams; ++b) {
    nb[b] = static_cast<int64_t>(nbytes[b]);
    max_calls_ = std::max<int>(max_calls_, static_cast<int>(nbytes[b] / kChunkBytes) - calls_done_[b]);
    total += (nbytes[b] + 15) & ~size_t(15);
  }
  if (on_device) {
    for (int b = 0; b < nstreams; ++b) ptrs[b] = iq[b];
  } else {
    if (!d_iq_own_.reserve(total) || !upload_iq(iq, nbytes, nstreams, ptrs)) return false;
  }
  if (!h_states_.resize(nstreams)) return false;
  StreamState* const states = h_states_.data();
  const size_t ndesc = static_cast<size_t>(nstreams) * max_calls_;
  const bool split_wanted = !(afc_ || full_scan);
  if (!cont) std::fill(states, states + nstreams, initial_state());
  static_assert(sizeof(CallDesc) % 16 == 0, "cleared in 16-byte pieces");
  if (!d_states_.reserve(nstreams) || !d_iq_ptrs_.reserve(nstreams) || !d_nbytes_.reserve(nstreams) || !d_descs_.reserve(ndesc) || !d_info_.reserve(ndesc + 1)) return false;
  if (!d_tail_state_.reserve(static_cast<size_t>(nstreams) * kTailBytes) || !d_tail_images_.reserve(ndesc * kTailBytes) ||
      (split_wanted && !d_tail_prev_.reserve(static_cast<size_t>(nstreams) * kTailBytes)))
    return false;
  const SyncTails tails{d_tail_state_.get(), d_tail_state_.get(), d_tail_images_.get(), kChunkBytes};
  // The look-ahead schedule of the chain (k_sync.hip: sync_ahead_kernel) where the chain would leave most of the device idle: few streams, many calls.
  // set_sync_speculation(0 / 1): never / always (tests run both); default: up to kAheadMaxStreams streams of at least kAheadMinCalls calls.
  // (forced on, the pass is still bounded: beyond kAheadForcedMaxStreams streams -- where it cannot help and its table, nstreams x nspec x 33 x 8 bytes
  // with nspec >= 64, would run to tens of megabytes and 33 x nspec x nstreams workgroups -- the plain chain runs whatever the mode says)
  constexpr int kAheadMinCalls = 16, kAheadForcedMaxStreams = 512;
  // measurement knobs (tools/gpu/k1_ahead_sweep.sh): the window of start positions the pass covers (DABHIP_K1_HYP: odd, 3 .. 63; default 33 = +-16 samples) and
  // the largest batch that takes the pass by default (DABHIP_K1_SPEC_MAX_STREAMS, default 4)
  static const int kAheadHypotheses = [] { const char* e = std::getenv("DABHIP_K1_HYP"); const int v = e ? std::atoi(e) : 33; return std::max(3, std::min(63, v | 1)); }();
  static const int kAheadMaxStreams = [] { const char* e = std::getenv("DABHIP_K1_SPEC_MAX_STREAMS"); const int v = e ? std::atoi(e) : 4; return std::max(0, std::min(512, v)); }();
  const bool use_spec = split_wanted && spec_mode_ != 0 &&
                        (spec_mode_ > 0 ? nstreams <= kAheadForcedMaxStreams : (nstreams <= kAheadMaxStreams && max_calls_ >= kAheadMinCalls));
  // calls of a stream per pass: all it has, within a bound on the table (8 bytes per call, start position and stream)
  const int nspec = std::min(max_calls_, std::min(4096, std::max(64, (1 << 20) / nstreams)));
  if (use_spec && (!d_spec_table_.reserve(static_cast<size_t>(nstreams) * nspec * kAheadHypotheses) || !d_spec_src0_.reserve(static_cast<size_t>(nstreams) * nspec) ||
                   !d_spec_ctl_.reserve(nstreams + 1) || !h_spec_hits_.resize(1)))
    return false;
  if (split_wanted) {
    // d_viol_[0 .. nstreams): first call of a stream that broke the chain's assumption; [nstreams]: calls the fp32 pass of the
    // verification left to the fp64 pass
    if (!h_viol_.resize(nstreams + 1) || !d_viol_.reserve(nstreams + 1) || !d_states_prev_.reserve(nstreams) || !d_calls_before_.reserve(nstreams) ||
        !h_calls_before_.resize(nstreams))
      return false;
    std::copy(calls_done_.begin(), calls_done_.begin() + nstreams, h_calls_before_.data());
  }
  {
    // one launch instead of nine copies and fills (launch_scan_setup): the kernel reads the page-locked host arrays itself
    ScanSetupArgs a{};
    a.h_states = cont ? nullptr : states;
    a.h_ptrs = ptrs;
    a.h_nbytes = nb;
    a.h_calls_before = split_wanted ? h_calls_before_.data() : nullptr;
    a.states = d_states_.get();
    a.states_prev = split_wanted ? d_states_prev_.get() : nullptr;
    a.iq_ptrs = d_iq_ptrs_.get();
    a.nbytes = d_nbytes_.get();
    a.calls_before = split_wanted ? d_calls_before_.get() : nullptr;
    a.viol = split_wanted ? d_viol_.get() : nullptr;
    a.descs = reinterpret_cast<uint4*>(d_descs_.get());
    a.desc_vec = ndesc * (sizeof(CallDesc) / 16);
    a.info = reinterpret_cast<uint4*>(d_info_.get());
    a.info_vec = (ndesc + 1) / 2;
    a.nstreams = nstreams;
    a.tail_state = d_tail_state_.get();
    a.tail_state_prev = split_wanted ? d_tail_prev_.get() : nullptr;
    if (!check(launch_scan_setup(a, stream_), "scan setup launch")) return false;
  }
  scan_setup_ms_ = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();

  if (!h_descs_.resize(ndesc) || !h_info_.resize(ndesc)) return false;
  sync_rescanned_ = 0;
  bool split_scan = false;
  if (!record(ev_[0], stream_)) return false;
  if (afc_ || full_scan) {
    // the reference's order, call after call: with the software AFC every call's NCO depends on the estimates of the call
    // before; and the fallback when the split scan's assumption failed
    if (!check(launch_sync_scan(d_iq_ptrs_.get(), d_nbytes_.get(), d_states_.get(), d_descs_.get(), d_info_.get(), nstreams, max_calls_, -1, -1,
                                d_tw2048_.get(), d_tw1536_.get(), d_prs_.get(), afc_ ? 1 : 0, stream_, false, nullptr, nullptr, tails),
               "sync scan launch"))
      return false;
  } else {
    // Split scan: the per-stream chain carries only what the next call depends on (FIFO, coarse and fine time) and assumes the
    // coarse frequency offset of every frame within +-1 carrier (input_sdr.c:105-109: otherwise the frame is dropped and a
    // resync forced); both frequency estimates are then computed for all frames in parallel (sync_verify_kernel).  A stream that
    // breaks the assumption (a capture more than a carrier off tune, noise) is scanned again from its incoming state in the
    // reference's order, so the result is the same in every case.  (Tried and dropped: the verification on a second stream beside
    // the OFDM stage, and -- after the LDS bank conflicts were gone -- the chain in 2..16 chunks of calls with each chunk's
    // verification beside the next chunk: 1.28 -> 1.30..1.40 ms.  A chain workgroup holds half of a CU's LDS, so the verification
    // beside it runs at half its rate and slows the chain.  Round 3, with the fp32 verification (39.5 KB of LDS, 54 VGPRs): on its own stream beside the FIC
    // symbols' OFDM launch -- step unchanged, 10.4 ms: both are issue-bound, the work only moves.)
    auto chain = [&](const SpecArgs& sp) {
      return check(launch_sync_scan(d_iq_ptrs_.get(), d_nbytes_.get(), d_states_.get(), d_descs_.get(), d_info_.get(), nstreams, max_calls_, -1, -1,
                                    d_tw2048_.get(), d_tw1536_.get(), d_prs_.get(), 0, stream_, true, nullptr, nullptr, tails, sp),
                   "sync chain launch");
    };
    bool chain_ok = true;
    if (use_spec) {
      // a short chain to lock on (a fresh capture: the first frame is dropped, the second finds the null symbol, the third the fine shift -- seven calls;
      // a session's further segment stands where it stands), then passes over all remaining calls at once, each followed by the chain launch that looks
      // its calls up (a long stream: several passes, each predicting from where the chain really got to)
      SpecArgs sp;
      sp.table = d_spec_table_.get();
      sp.src0 = d_spec_src0_.get();
      sp.ctl = d_spec_ctl_.get();
      sp.nspec = nspec;
      sp.nstreams = nstreams;
      sp.nhyp = kAheadHypotheses;
      sp.call_limit = cont ? 0 : 7;
      sp.record_base = 1;
      chain_ok = chain(sp);
      sp.record_base = 0;
      sp.lookup = 1;
      const int passes = std::min(16, (max_calls_ + nspec - 1) / nspec);
      for (int r = 0; r < passes && chain_ok; ++r) {
        sp.call_limit = r + 1 < passes ? nspec : -1;
        chain_ok = check(launch_sync_ahead(d_iq_ptrs_.get(), d_nbytes_.get(), d_states_.get(), nstreams, d_tw2048_.get(), d_tw1536_.get(), d_prs_.get(), stream_, sp),
                         "sync look-ahead launch") &&
                   chain(sp);
      }
    } else {
      chain_ok = chain(SpecArgs{});
    }
    if (!chain_ok ||
        // {status, ordinal} of every call are final once the chain is through (a stream that breaks its assumption is scanned again
        // below): they come back on the side stream while the verification runs, and the caller lays the frames out beside it
        !check(hipEventRecord(ev_chain_, stream_), "chain event") || !check(hipStreamWaitEvent(copy_stream_, ev_chain_, 0), "chain event") ||
        !check(hipMemcpyAsync(h_info_.data(), d_info_.get(), ndesc * sizeof(int2), hipMemcpyDeviceToHost, copy_stream_), "call info download") ||
        !check(hipEventRecord(ev_info_, copy_stream_), "call info event") ||
        !check(launch_sync_verify(d_iq_ptrs_.get(), d_nbytes_.get(), d_calls_before_.get(), d_states_.get(), d_descs_.get(), nstreams, max_calls_,
                                  d_tw2048_.get(), d_prs_.get(), d_viol_.get(), false, stream_),
               "sync verify launch") ||
        // fine_freq_shift carried through the calls that did not demodulate (the kernel skips streams with a violation)
        !check(launch_sync_verify(d_iq_ptrs_.get(), d_nbytes_.get(), d_calls_before_.get(), d_states_.get(), d_descs_.get(), nstreams, max_calls_,
                                  d_tw2048_.get(), d_prs_.get(), d_viol_.get(), true, stream_),
               "sync carry launch"))
      return false;                                        // (the violation marks come back on the side stream, in fetch(): a copy on the main stream sits between K1 and the first OFDM launch)
    split_scan = true;
  }
  if (!record(ev_[1], stream_)) return false;
  // The host only needs {status, ordinal} of every call to lay the frames out: K1 writes those 8 bytes per call to a
  // compact array that comes back first; the full descriptors (trace API) follow on the side stream.
  // (on the side stream, behind the scan's last kernel: what the layout callback may have queued on the main stream meanwhile -- the
  // first OFDM launch -- is not waited for)
  auto fetch = [&]() {
    // Small scans: the four downloads as ONE kernel that writes the page-locked host arrays itself (launch_host_words works in either direction: both
    // sides are addresses the device can reach) instead of four copy-engine commands in a row, each some microseconds of the host waiting.
    const size_t words = (split_scan ? nstreams + 1 : 0) + (split_scan && use_spec ? 1 : 0) + ndesc * 2 + static_cast<size_t>(nstreams) * (sizeof(StreamState) / 4);
    if (words <= (size_t(1) << 18)) {
      HostWordsArgs hw{};
      int k = 0;
      auto add = [&](const void* src, void* dst, size_t n) {
        hw.src[k] = static_cast<const uint32_t*>(src);
        hw.dst[k] = static_cast<uint32_t*>(dst);
        hw.nwords[k++] = static_cast<uint32_t>(n);
      };
      if (split_scan) add(d_viol_.get(), h_viol_.data(), nstreams + 1);
      if (split_scan && use_spec) add(d_spec_ctl_.get() + nstreams, h_spec_hits_.data(), 1);
      add(d_info_.get(), h_info_.data(), ndesc * 2);
      add(d_states_.get(), states, static_cast<size_t>(nstreams) * (sizeof(StreamState) / 4));
      return check(hipStreamWaitEvent(copy_stream_, ev_[1], 0), "scan event") && check(launch_host_words(hw, copy_stream_), "scan results download") &&
             check(hipStreamSynchronize(copy_stream_), "sync scan");
    }
    return check(hipStreamWaitEvent(copy_stream_, ev_[1], 0), "scan event") &&
           (!split_scan || check(hipMemcpyAsync(h_viol_.data(), d_viol_.get(), (nstreams + 1) * sizeof(int), hipMemcpyDeviceToHost, copy_stream_), "violation download")) &&
           (!(split_scan && use_spec) || check(hipMemcpyAsync(h_spec_hits_.data(), d_spec_ctl_.get() + nstreams, sizeof(int), hipMemcpyDeviceToHost, copy_stream_), "look-ahead hits download")) &&
           check(hipMemcpyAsync(h_info_.data(), d_info_.get(), ndesc * sizeof(int2), hipMemcpyDeviceToHost, copy_stream_), "call info download") &&
           check(hipMemcpyAsync(states, d_states_.get(), nstreams * sizeof(StreamState), hipMemcpyDeviceToHost, copy_stream_), "state download") &&
           check(hipStreamSynchronize(copy_stream_), "sync scan");
  };
  if (split_scan && (!check(hipEventSynchronize(ev_info_), "call info") || !layout())) return false;
  if (!fetch()) return false;
  if (!split_scan && !layout()) return false;
  if (split_scan) {
    std::vector<int> redo;
    for (int b = 0; b < nstreams; ++b)
      if (h_viol_[b] != 0x7f7f7f7f) redo.push_back(b);
    sync_rescanned_ = static_cast<int>(redo.size());
    times_.sync_fp64_calls = static_cast<float>(h_viol_[nstreams]);
    times_.sync_spec_calls = use_spec ? static_cast<float>(h_spec_hits_[0]) : 0.0f;
    if (!redo.empty()) {                                   // rare: those streams again, in the reference's order, from their incoming state
      // (what the first layout queued -- its set-up kernel reads the page-locked frame lists when it RUNS -- is through before the lists are rewritten)
      // (the guarded launches of the first layout have run; they are made again for the new frame list: their counters and counts start over --
      // the second layout's set-up kernel clears the device side again)
      guard_launches_ = 0;
      guard_counters_clear_ = false;
      if (!check(hipStreamSynchronize(stream_), "before the rescan") || !d_redo_.upload(redo, stream_) ||
          !check(launch_sync_scan(d_iq_ptrs_.get(), d_nbytes_.get(), d_states_.get(), d_descs_.get(), d_info_.get(), static_cast<int>(redo.size()), max_calls_,
                                  -1, -1, d_tw2048_.get(), d_tw1536_.get(), d_prs_.get(), 0, stream_, false, d_states_prev_.get(), d_redo_.get(),
                                  SyncTails{d_tail_prev_.get(), d_tail_state_.get(), d_tail_images_.get(), kChunkBytes}),
                 "sync rescan launch"))
        return false;
      if (!record(ev_[1], stream_)) return false;
      if (!fetch() || !layout()) return false;             // the frames of those streams may have changed
    }
  }
  if (!check(hipStreamWaitEvent(copy_stream_, ev_[1], 0), "desc download") ||
      !check(hipMemcpyAsync(h_descs_.data(), d_descs_.get(), ndesc * sizeof(CallDesc), hipMemcpyDeviceToHost, copy_stream_), "desc download"))
    return false;
  if (!elapsed(&times_.sync, ev_[0], ev_[1])) return false;
  for (int b = 0; b < nstreams; ++b)
    if (states[b].overflow) { set_error("sync scan: stale-tail bookkeeping overflow (more than kMaxSeg nested short reads)"); return false; }
  return true;
}

int64_t Engine::decode_impl(const uint8_t* const* iq, const size_t* nbytes, int nstreams, bool on_device, bool cont, bool full_scan)
{
  if (!ok_) { set_error("engine not initialised (no GPU?)"); return -1; }
  if (nstreams <= 0) { set_error("decode: no streams"); return -1; }
  if (!check(hipSetDevice(device_), "hipSetDevice")) return -1;
  const auto wall0 = std::chrono::steady_clock::now();
  auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  static const bool trace_host = std::getenv("DABHIP_TRACE_HOST") != nullptr;
  auto mark = [&](const char* what) { if (trace_host) std::fprintf(stderr, "[host] %-18s %8.3f ms\n", what, since(wall0)); };
  times_ = StageTimes{};
  guard_counters_clear_ = false;
  fft_launches_ = fft_tfs_ = 0;
  fft_ms_ = 0;
  guard_flagged_ = guard_decisions_ = 0;
  guard_launches_ = 0;
  guard_overflows_ = 0;
  if (!begin_decode(nstreams, cont)) return -1;
  struct SideStreamGuard {                   // whatever was queued on the side stream is awaited before returning
    hipStream_t s;
    ~SideStreamGuard() { (void)hipStreamSynchronize(s); }
  } side_guard{copy_stream_};
  mark("begin_decode done");
  // frame list: demodulated TFs, stream-major.  Slots and logical CIF rows of a stream: first the ones carried over from
  // the previous segment of a session (the last <= 4 TFs), then this segment's.  Built (and uploaded) by the scan as soon as the
  // calls' {status, ordinal} are known, i.e. while K1's verification kernel still runs.
  std::vector<int> tf_base(nstreams + 1, 0), row_base(nstreams), fib_base(nstreams), nnew(nstreams, 0);
  int next_row = 0, ntf_new = 0;
  float frames_ms = 0;
  auto layout = [&]() -> bool {
    const auto tfr = std::chrono::steady_clock::now();
    const size_t nd = static_cast<size_t>(nstreams) * max_calls_;
    if (!h_frames_.resize(nd) || !h_frame_slot_.resize(nd) || !h_frame_cif_row_.resize(nd)) return false;   // page-locked: uploaded asynchronously
    next_row = 0;
    ntf_new = 0;
    tf_base[0] = 0;
    for (int b = 0; b < nstreams; ++b) {
      const int keep = carry_keep_[b], j0 = ntf_new;
      const int ncalls = static_cast<int>(nbytes[b] / kChunkBytes) - calls_done_[b];
      row_base[b] = next_row + kRowLead;      // each stream gets 15 lead-in rows for the scatter of its first CIFs
      for (int k = 0; k < ncalls; ++k) {
        const int2 d = h_info_[static_cast<size_t>(b) * max_calls_ + k];     // {status, ordinal}
        if (d.x == 2) {
          const int local = keep + (d.y - ord_done_[b]);
          h_frames_[ntf_new] = make_int2(b, k);
          h_frame_slot_[ntf_new] = tf_base[b] + local;
          h_frame_cif_row_[ntf_new] = row_base[b] + 4 * local;
          ++ntf_new;
        }
      }
      nnew[b] = ntf_new - j0;
      tf_base[b + 1] = tf_base[b] + keep + nnew[b];
      fib_base[b] = 4 * tf_base[b];
      next_row += kRowLead + 4 * (keep + nnew[b]);
    }
    // the three lists go up in ONE launch that reads the page-locked arrays itself (three copy-engine copies cost 45 us of idle GPU before the first
    // OFDM launch); with the guard on it also clears the guard's counters, which guard_begin() then leaves alone
    bool up = true;
    if (ntf_new > 0) {
      up = d_frames_.reserve(ntf_new) && d_frame_slot_.reserve(ntf_new) && d_frame_cif_row_.reserve(ntf_new);
      HostWordsArgs hw{};
      hw.src[0] = reinterpret_cast<const uint32_t*>(h_frames_.data());
      hw.dst[0] = reinterpret_cast<uint32_t*>(d_frames_.get());
      hw.nwords[0] = 2u * static_cast<uint32_t>(ntf_new);
      hw.src[1] = reinterpret_cast<const uint32_t*>(h_frame_slot_.data());
      hw.dst[1] = reinterpret_cast<uint32_t*>(d_frame_slot_.get());
      hw.nwords[1] = static_cast<uint32_t>(ntf_new);
      hw.src[2] = reinterpret_cast<const uint32_t*>(h_frame_cif_row_.data());
      hw.dst[2] = reinterpret_cast<uint32_t*>(d_frame_cif_row_.get());
      hw.nwords[2] = static_cast<uint32_t>(ntf_new);
      if (up && guard_active() && guard_launches_ == 0 && guard_reserve_counters(ntf_new)) {
        hw.zero = d_guard_counter_.get();
        hw.nzero = static_cast<uint32_t>(h_guard_counts_.size());
        guard_counters_clear_ = true;
      }
      up = up && check(launch_host_words(hw, stream_), "frame list upload");
    }
    frames_ms += since(tfr);
    return up;
  };
  // Stage A: everything between the layout and the FIC decode -- buffers for the layout, then the FIC symbols (0..3) of every TF
  // through the OFDM stage.  K3 comes first so that the FIC is decoded, and the host control plane can run, while the bulk of the
  // OFDM stage still occupies the GPU: a launch of 4 / 76 of the work, so that the FIBs reach the host 2 ms before the MSC
  // symbols are through (with the first 19 symbols in this launch the host finished 0.4 ms AFTER the OFDM stage).  Two-kernel stage
  // (set_fused(0)): a pre-pass over the same four symbols.
  // A fresh decode runs stage A from the layout callback, i.e. queued right behind K1 while the host still waits for K1's last
  // downloads (0.19 ms of idle GPU otherwise); if a stream is scanned again afterwards (rare), the layout and stage A simply run
  // again for the new frame list.  A session's further segments run it after the scan: their carry-over copies must happen once.
  std::unique_lock<std::mutex> heavy;
  const bool one_kernel = fused_;                        // hard (with or without the guard) and soft decisions alike
  const bool guard = guard_active();
  const bool soft = soft_bits_ != 0;
  const bool energies = guard || soft;                    // the per-symbol sample energies: the guard's error bounds, the soft scale
  int ntf = 0, nslots = 0, chunk = 1;
  uint8_t* fibs = nullptr;
  uint8_t* ok = nullptr;
  GuardArgs soft_args{};                                  // soft decisions, two-kernel stage: K2b reads the energies, lists nothing
  auto fused_parts = [&](int first, int n, int sym_a, int sym_b, int nparts) -> bool {
    if (soft)
      return check(launch_ofdm_demap_fused_soft(afc_, d_iq_ptrs_.get(), d_descs_.get(), max_calls_, d_frames_.get(), first, n, d_twf_.get(), d_frame_slot_.get(),
                                                d_frame_cif_row_.get(), d_qpsk_.get(), d_fic_bits_.get(), d_msc_bits_.get(), stream_, sym_a, sym_b, nparts),
                   "fused fft/demap launch");
    GuardArgs ga{};
    if (guard && !guard_begin(n, &ga)) return false;
    const bool launched =
        guard ? check(launch_ofdm_demap_fused_guarded(d_iq_ptrs_.get(), d_descs_.get(), max_calls_, d_frames_.get(), first, n, d_twf_.get(), d_frame_slot_.get(),
                                                      d_frame_cif_row_.get(), d_qpsk_.get(), d_fic_bits_.get(), d_msc_bits_.get(), ga, stream_, sym_a, sym_b, nparts),
                      "fused fft/demap launch")
              : check(launch_ofdm_demap_fused_plain(afc_, d_iq_ptrs_.get(), d_descs_.get(), max_calls_, d_frames_.get(), first, n, d_twf_.get(), d_frame_slot_.get(),
                                                    d_frame_cif_row_.get(), d_qpsk_.get(), d_fic_bits_.get(), d_msc_bits_.get(), stream_, sym_a, sym_b, nparts),
                      "fused fft/demap launch");
    return launched && (!guard || guard_finish(true, first, n, sym_a, sym_b, false));
  };
  auto stage_a = [&]() -> bool {
    ntf = ntf_new;
    nslots = tf_base[nstreams];
    if (ntf == 0) return true;
    if (!carry_and_reserve(tf_base, row_base, nslots, next_row + 1)) return false;
    if (heavy_mu_ && !heavy.owns_lock()) heavy = std::unique_lock<std::mutex>(*heavy_mu_);
    chunk = one_kernel ? std::max(ntf, 1) : std::min(ntf, kFftChunkTfs);   // only the spectra buffer of the two-kernel stage calls for chunks
    if (!one_kernel && !d_spectra_.reserve(static_cast<size_t>(chunk) * kSymbolsPerTf * 2048)) return false;
    if (!h_fibs_.resize(static_cast<size_t>(nslots) * 384) || !h_fib_ok_.resize(static_cast<size_t>(nslots) * 12)) return false;
    fibs = h_fibs_.data();
    ok = h_fib_ok_.data();
    if (!record(ev_[3], stream_)) return false;
    if (energies && !d_delta_.reserve(static_cast<size_t>(ntf) * kSymbolsPerTf)) return false;
    if (guard && guard_launches_ == 0 && !guard_counters_clear_ && !guard_reserve_counters(ntf)) return false;
    soft_args.delta = d_delta_.get();
    soft_args.delta_stride = kSymbolsPerTf;
    soft_args.c = kSoftNormC;
    if (one_kernel) {
      for (int first = 0; first < ntf; first += chunk)
        if (!fused_parts(first, std::min(chunk, ntf - first), 1, 4, 1)) return false;      // the three FIC symbols (and symbol 0, their reference)
    } else {
      for (int first = 0; first < ntf; first += chunk * 19) {       // 4 of 76 symbols: 19 x as many TFs fit the spectra buffer
        const int n = std::min(chunk * 19, ntf - first);
        GuardArgs ga = soft ? soft_args : GuardArgs{};   // (hard decisions: a non-null delta switches the guard's listing on)
        if (guard && !guard_begin(n, &ga)) return false;
        if (energies && !check(launch_symbol_delta(d_iq_ptrs_.get(), d_descs_.get(), max_calls_, d_frames_.get(), first, n, 4, d_delta_.get(), kSymbolsPerTf, soft ? kSoftNormC : guard_c_of(guard_rule_level()), stream_), "symbol delta launch"))
          return false;
        if (!check(launch_fic_prepass(soft_bits_, d_iq_ptrs_.get(), d_descs_.get(), max_calls_, d_frames_.get(), first, n, d_spectra_.get(), d_twf_.get(),
                                      d_frame_slot_.get(), d_qpsk_.get(), d_fic_bits_.get(), ga, stream_),
                   "fic pre-pass launch"))
          return false;
        if (guard && !guard_finish(true, first, n, 1, 4, false)) return false;
      }
    }
    if (!record(ev_part0_, stream_)) return false;
    return true;
  };
  const bool early_a = !cont;
  auto layout_and_a = [&]() -> bool { return layout() && (!early_a || stage_a()); };
  if (!scan_streams(iq, nbytes, nstreams, on_device, cont, full_scan, layout_and_a)) return -1;
  mark("scan done");
  times_.setup = scan_setup_ms_;
  for (int b = 0; b < nstreams; ++b) calls_done_[b] = std::max(calls_done_[b], static_cast<int>(nbytes[b] / kChunkBytes));
  if (!early_a && !stage_a()) return -1;
  last_ntf_ = ntf;
  if (ntf == 0) return 0;                   // nothing demodulated: layout and carried data stay as they are
  times_.frames = frames_ms;
  if (guard) guard_decisions_ += static_cast<int64_t>(ntf) * (kFicBits + kMscBits);
  // FIC decode kernels and the FIB download on the side stream: the rest of the OFDM stage is queued on the main stream right
  // away and shares the GPU with them, waiting neither for the download nor for the host
  if (!fic_decode_slots_async(0, nslots, fibs, ok, copy_stream_)) return -1;      // carried slots are decoded again: their FIBs are read by K5

  // K2 + K2b in chunks (they share one spectra buffer; stream order keeps them apart), timed with per-chunk events
  bool gpu_ok = true;
  const int nchunks = (ntf + chunk - 1) / chunk;
  while (static_cast<int>(chunk_ev_.size()) < 3 * nchunks) {
    hipEvent_t e = nullptr;
    if (!check(hipEventCreate(&e), "hipEventCreate")) { gpu_ok = false; break; }
    chunk_ev_.push_back(e);
  }
  for (int c = 0; c < nchunks && gpu_ok; ++c) {
    const int first = c * chunk, n = std::min(chunk, ntf - first);
    gpu_ok = record(chunk_ev_[3 * c], stream_);
    if (one_kernel) {
      // the 72 MSC symbols (the FIC symbols ran before the FIC decode was queued); workgroups per frame: measurement knob
      static const int msc_wgs = std::getenv("DABHIP_FUSED_MSC_WGS") ? std::max(1, std::min(8, std::atoi(std::getenv("DABHIP_FUSED_MSC_WGS")))) : 1;
      gpu_ok = gpu_ok && fused_parts(first, n, 4, 76, msc_wgs);
      gpu_ok = gpu_ok && record(chunk_ev_[3 * c + 1], stream_);
    } else {
      GuardArgs ga = soft ? soft_args : GuardArgs{};   // (hard decisions: a non-null delta switches the guard's listing on)
      if (guard && !guard_begin(n, &ga)) { gpu_ok = false; break; }
      // with the guard on (or soft decisions), K2 also leaves the per-symbol sample energies K2b decides with
      gpu_ok = gpu_ok && check(launch_ofdm_fft(d_iq_ptrs_.get(), d_descs_.get(), max_calls_, d_frames_.get(), first, n, d_spectra_.get(), d_twf_.get(), stream_,
                                               energies ? d_delta_.get() : nullptr, soft ? kSoftNormC : guard_c_of(guard_rule_level())),
                               "fft launch");
      gpu_ok = gpu_ok && record(chunk_ev_[3 * c + 1], stream_);
      gpu_ok = gpu_ok && check(launch_demap(true, soft_bits_, d_spectra_.get(), first, n, d_frame_slot_.get(), d_frame_cif_row_.get(), d_qpsk_.get(), d_fic_bits_.get(), d_msc_bits_.get(), ga, stream_), "demap launch");
      if (guard) gpu_ok = gpu_ok && guard_finish(true, first, n, 1, kSymbolsPerTf, true);     // timed with the demapper; the FIC symbols belong to the pre-pass
    }
    gpu_ok = gpu_ok && record(chunk_ev_[3 * c + 2], stream_);
  }
  mark("ofdm queued");
  if (!check(hipEventSynchronize(ev_fibs_), "fic decode")) return -1;
  mark("fibs on host");
  {
    float part0_ms = 0, fic_ms = 0;
    // (the FIC decode runs beside the OFDM stage since round 2: no longer a term of the step.  A failed query: the decode fails below, after the drain)
    if (!elapsed(&part0_ms, ev_[3], ev_part0_) || !elapsed(&fic_ms, ev_part0_, ev_fic_done_)) gpu_ok = false;
    times_.fic = fic_ms + (one_kernel ? 0.0f : part0_ms);   // the pre-pass of the two-kernel stage is FIC work; the FIC symbols' launch of the fused kernel is OFDM work
    if (one_kernel) times_.fft += part0_ms;
  }

  // control plane + work lists on a host thread, hidden behind the MSC symbols' part of the OFDM stage
  std::vector<ControlPlane>& planes = planes_;
  // host work lists live in the engine: ~35 MB per step at the benchmark size, reused instead of re-allocated
  std::vector<JobList>& stream_jobs = stream_jobs_;
  stream_jobs.resize(nstreams);
  for (auto& v : stream_jobs) v.clear();
  MscWork& work = work_;
  bool host_ok = true;
  std::string host_error;
  // (on the engine's persistent lane since round 3; a std::thread created and joined per decode measured the same on an idle host:
  // 4.41 M against 4.40 M ETI frames/s)
  host_lane_->post([&]() {
    (void)hipSetDevice(device_);               // the current device is per thread
    const auto t0 = std::chrono::steady_clock::now();
    const bool fresh = planes_fresh_;
    pool_->parallel_for(nstreams, [&](int b) {
      if (fresh) {
        planes[b] = ControlPlane();
        planes[b].set_filter(subch_keep_);
      }
      stream_jobs[b].reserve(static_cast<size_t>(4) * nnew[b]);
      planes[b].rebase(4 * (prev_used_[b] - carry_keep_[b]));   // CIF numbering of this segment's layout
      for (int s = tf_base[b] + carry_keep_[b]; s < tf_base[b + 1]; ++s)
        planes[b].on_tf(s - tf_base[b], fibs + static_cast<size_t>(s) * 384, ok + static_cast<size_t>(s) * 12, stream_jobs[b]);
    });
    std::vector<const ControlPlane*> plane_ptrs(nstreams);
    std::vector<const JobList*> job_ptrs(nstreams);
    total_eti_ = 0;
    for (int b = 0; b < nstreams; ++b) {
      plane_ptrs[b] = &planes[b];
      job_ptrs[b] = &stream_jobs[b];
      eti_base_[b] = total_eti_;
      eti_count_[b] = static_cast<int64_t>(stream_jobs[b].size());
      stream_status_[b] = planes[b].fault();
      total_eti_ += eti_count_[b];
    }
    planes_fresh_ = false;
    times_.control = since(t0);
    mark("control plane done");
    const auto t1 = std::chrono::steady_clock::now();
    // the work lists go up on the side stream while the OFDM stage still runs on the main one
    host_ok = msc_prepare(job_ptrs, plane_ptrs, row_base, fib_base, work);
    mark("work lists built");
    host_ok = host_ok && msc_upload(work, copy_stream_) && check(hipEventRecord(ev_upload_, copy_stream_), "work list upload");
    mark("work lists queued");
    if (!host_ok) host_error = dabhip_last_error();
    times_.worklist = since(t1);
  });

  // the host thread is done before the OFDM stage (at 4.5 of 5.8 ms into the step with 24 threads): K4 + K5 are queued right
  // behind it, and the whole pipeline is awaited ONCE
  host_lane_->wait();
  if (gpu_ok && host_ok)
    // (the upload is normally through long before this point: then no wait is queued at all -- a wait on an event that has already fired still costs the
    // main stream a barrier packet, 10 .. 15 us of idle GPU before K4)
    gpu_ok = (hipEventQuery(ev_upload_) == hipSuccess || check(hipStreamWaitEvent(stream_, ev_upload_, 0), "work list wait")) && msc_launch_async(work);
  if (guard && gpu_ok) gpu_ok = guard_download();        // the entry counts of all guarded launches, behind everything else
  mark("all queued");
  const bool drained = check(hipStreamSynchronize(stream_), "decode");      // also on the error paths: nothing may stay in flight
  mark("stream drained");
  if (heavy.owns_lock()) heavy.unlock();
  if (!gpu_ok || !drained) return -1;
  if (!host_ok) { set_error(host_error); return -1; }
  for (int c = 0; c < nchunks; ++c) {
    float a = 0, d = 0;
    if (!elapsed(&a, chunk_ev_[3 * c], chunk_ev_[3 * c + 1]) || !elapsed(&d, chunk_ev_[3 * c + 1], chunk_ev_[3 * c + 2])) return -1;

    times_.fft += a;
    times_.demap += d;
    fft_ms_ += a;
    fft_launches_ += 1;
    fft_tfs_ += std::min(chunk, ntf - c * chunk);
  }
  msc_collect();
  if (!on_device && times_.h2d_bytes > 0 && !elapsed(&times_.h2d, ev_h2d_[0], ev_h2d_[1])) return -1;
  if (guard && !guard_check()) return -1;
  // what the next segment of a session starts from
  for (int b = 0; b < nstreams; ++b) {
    prev_used_[b] = carry_keep_[b] + nnew[b];
    carry_keep_[b] = std::min(4, prev_used_[b]);
    ord_done_[b] += nnew[b];
  }
  prev_tf_base_ = tf_base;
  prev_row_base_ = row_base;
  times_.wall = since(wall0);
  mark("return");
  return total_eti_;
}

int64_t Engine::eti_count(int stream) const { return (stream >= 0 && stream < nstreams_) ? eti_count_[stream] : -1; }
uint32_t Engine::stream_status(int stream) const { return (stream >= 0 && stream < static_cast<int>(stream_status_.size())) ? stream_status_[stream] : 0xffffffffu; }

int64_t Engine::eti_read(int stream, uint8_t* dst, int64_t cap_frames)
{
  if (stream < 0 || stream >= nstreams_) { set_error("eti_read: bad stream"); return -1; }
  const int64_t n = std::min(cap_frames, eti_count_[stream]);
  return read_eti(eti_base_[stream], n, dst) ? n : -1;
}

// Up to TWO fetches may be outstanding (the CLI's pipeline: its writer thread still waits for fetch k while the decode thread, through with decode
// k + 1, issues fetch k + 1 into the other output buffer): each has its own event, eti_fetch_wait() waits for the OLDEST one not yet waited for.  A
// third fetch without a wait is refused -- its destination would be a buffer somebody is still reading.  The copies run in order on one stream.
int64_t Engine::eti_fetch_async(uint8_t* dst, int64_t cap_frames)
{
  if (!dst) { set_error("eti_fetch: null destination"); return -1; }
  if (!check(hipSetDevice(device_), "hipSetDevice")) return -1;
  const uint64_t issued = eti_fetch_issued_.load(), waited = eti_fetch_waited_.load();
  if (issued - waited >= 2) { set_error("eti_fetch: two fetches are outstanding -- eti_fetch_wait first"); return -1; }
  const int64_t n = std::min(cap_frames, total_eti_);
  // (decode() has returned: the frames are complete; the copy is ordered before the next decode's K4 by the newest fetch event)
  if (n > 0 && !check(hipMemcpyAsync(dst, d_eti_.get(), static_cast<size_t>(n) * kEtiBytes, hipMemcpyDeviceToHost, d2h_stream_), "eti fetch")) return -1;
  if (!check(hipEventRecord(ev_eti_fetch_[issued & 1], d2h_stream_), "eti fetch event")) return -1;
  eti_fetch_issued_.store(issued + 1);
  return n;
}

bool Engine::eti_fetch_wait()
{
  const uint64_t waited = eti_fetch_waited_.load();
  if (waited == eti_fetch_issued_.load()) return true;
  bool ok = check(hipSetDevice(device_), "hipSetDevice") && check(hipEventSynchronize(ev_eti_fetch_[waited & 1]), "eti fetch");
  // the download stream is only ever waited for through these events: let the runtime drop its records of the finished fetches (engine.hpp:
  // blocking_copy) whenever that costs nothing -- no newer fetch queued -- and on every 32nd fetch of a pipeline that always has one in flight
  if (ok && reap_enabled() && (waited + 1 == eti_fetch_issued_.load() || waited % kReapEvery == kReapEvery - 1)) ok = check(hipStreamSynchronize(d2h_stream_), "eti fetch");
  eti_fetch_waited_.store(waited + 1);
  return ok;
}

const uint8_t* Engine::eti_device(int64_t* nframes) const
{
  if (nframes) *nframes = total_eti_;
  return d_eti_.get();
}

int Engine::trace(int stream, int32_t* ints6, double* ffs, int cap_calls) const
{
  if (stream < 0 || stream >= nstreams_) return -1;
  int n = 0;
  for (int k = 0; k < max_calls_ && n < cap_calls; ++k, ++n) {
    const CallDesc& d = h_descs_[static_cast<size_t>(stream) * max_calls_ + k];
    int32_t* o = ints6 + 6 * k;
    o[0] = d.status == 2; o[1] = d.status >= 1; o[2] = d.coarse_timeshift; o[3] = d.fine_timeshift;
    o[4] = d.coarse_freq_shift; o[5] = d.fifo_count;
    if (ffs) ffs[k] = d.fine_freq_shift;
  }
  return n;
}

int Engine::trace_nco(int stream, int32_t* nco_hz, int cap_calls) const
{
  if (stream < 0 || stream >= nstreams_ || !nco_hz) return -1;
  int n = 0;
  for (int k = 0; k < max_calls_ && n < cap_calls; ++k, ++n) nco_hz[k] = h_descs_[static_cast<size_t>(stream) * max_calls_ + k].nco_hz;
  return n;
}

// K2 (ofdm_fft_kernel) alone over the frames of the last decode: the same IQ, the same frame list and the same launch
// shape (chunks of kFftChunkTfs) as the two-kernel OFDM stage, whatever stage the decode itself used.  This is the
// HBM-roofline measurement of SURVEY.md 8(d): 311,296 B read + 1,245,184 B written per TF.
int Engine::fft_roofline(int reps, int64_t* launches, int64_t* tfs, double* ms)
{
  if (!ok_) { set_error("engine not initialised (no GPU?)"); return -1; }
  if (last_ntf_ <= 0) { set_error("fft_roofline: no decode to measure on"); return -1; }
  if (!check(hipSetDevice(device_), "hipSetDevice")) return -1;
  const int ntf = last_ntf_, chunk = std::min(ntf, kFftChunkTfs);
  if (!d_spectra_.reserve(static_cast<size_t>(chunk) * kSymbolsPerTf * 2048)) return -1;
  reps = std::max(reps, 1);
  int64_t nl = 0, nt = 0;
  double total = 0;
  for (int r = -1; r < reps; ++r) {                      // r = -1: untimed
    for (int first = 0; first < ntf; first += chunk) {
      const int n = std::min(chunk, ntf - first);
      if (!record(ev_[0], stream_)) return -1;
      if (!check(launch_ofdm_fft(d_iq_ptrs_.get(), d_descs_.get(), max_calls_, d_frames_.get(), first, n, d_spectra_.get(), d_twf_.get(), stream_), "fft launch")) return -1;
      if (!record(ev_[1], stream_)) return -1;
      if (!check(hipEventSynchronize(ev_[1]), "fft")) return -1;
      float t = 0;
      if (!elapsed(&t, ev_[0], ev_[1])) return -1;
      if (r >= 0) { total += t; ++nl; nt += n; }
    }
  }
  if (launches) *launches = nl;
  if (tfs) *tfs = nt;
  if (ms) *ms = total;
  return 0;
}

void Engine::fft_stats(int64_t* launches, int64_t* tfs, double* ms) const
{
  if (launches) *launches = fft_launches_;
  if (tfs) *tfs = fft_tfs_;
  if (ms) *ms = fft_ms_;
}

// ---------------------------------------------------------------------------------------------
int Engine::stage_ofdm_fft(const uint8_t* frames, int nframes, float* spectra, bool on_device, int reps, float* kernel_ms)
{
  if (!ok_) { set_error("engine not initialised (no GPU?)"); return -1; }
  if (nframes <= 0) return 0;
  const size_t bytes = static_cast<size_t>(nframes) * kTfBytes;
  const uint8_t* d_in = frames;
  if (!on_device) {
    if (!d_iq_own_.reserve(bytes) || !check(blocking_copy(d_iq_own_.get(), frames, bytes, hipMemcpyHostToDevice), "frame upload")) return -1;
    d_in = d_iq_own_.get();
  }
  std::vector<CallDesc> descs(nframes);
  std::vector<int2> list(nframes);
  for (int j = 0; j < nframes; ++j) {
    std::memset(&descs[j], 0, sizeof(CallDesc));
    descs[j].status = 2;
    descs[j].ordinal = j;
    descs[j].view = initial_state().view;
    descs[j].view.seg_src[0] = static_cast<int64_t>(j) * kTfBytes;
    list[j] = make_int2(0, j);
  }
  std::vector<const uint8_t*> ptrs = {d_in};
  const size_t nspec = static_cast<size_t>(nframes) * kSymbolsPerTf * 2048;
  if (!d_iq_ptrs_.upload(ptrs, stream_) || !d_descs_.upload(descs, stream_) || !d_frames_.upload(list, stream_) || !d_spectra_.reserve(nspec)) return -1;
  reps = std::max(reps, 1);
  // one untimed launch first when timing
  if (reps > 1 && !check(launch_ofdm_fft(d_iq_ptrs_.get(), d_descs_.get(), nframes, d_frames_.get(), 0, nframes, d_spectra_.get(), d_twf_.get(), stream_), "fft launch")) return -1;
  if (!record(ev_[0], stream_)) return -1;
  for (int r = 0; r < reps; ++r)
    if (!check(launch_ofdm_fft(d_iq_ptrs_.get(), d_descs_.get(), nframes, d_frames_.get(), 0, nframes, d_spectra_.get(), d_twf_.get(), stream_), "fft launch")) return -1;
  if (!record(ev_[1], stream_)) return -1;
  if (!check(hipEventSynchronize(ev_[1]), "fft")) return -1;
  float ms = 0;
  if (!elapsed(&ms, ev_[0], ev_[1])) return -1;
  if (kernel_ms) *kernel_ms = ms / reps;
  if (spectra && !check(blocking_copy(spectra, d_spectra_.get(), nspec * sizeof(float2), hipMemcpyDeviceToHost), "spectra download")) return -1;
  return nframes;
}

int Engine::stage_demap(const float* spectra, int nframes, uint8_t* fic, uint8_t* msc)
{
  if (!hard_only("stage_demap")) return -1;
  if (!ok_) { set_error("engine not initialised (no GPU?)"); return -1; }
  if (nframes <= 0) return 0;
  const size_t nspec = static_cast<size_t>(nframes) * kSymbolsPerTf * 2048;
  std::vector<int> slots(nframes), rows(nframes);
  for (int j = 0; j < nframes; ++j) { slots[j] = j; rows[j] = 4 * j; }
  if (!reserve_tf_slots(nframes) || !d_spectra_.reserve(nspec) || !d_frame_slot_.upload(slots, stream_) || !d_frame_cif_row_.upload(rows, stream_)) return -1;
  if (!check(hipMemcpyAsync(d_spectra_.get(), spectra, nspec * sizeof(float2), hipMemcpyHostToDevice, stream_), "spectra upload")) return -1;
  // spectra only: no samples to re-decide from, so this stage entry returns the raw fp32 decisions (no parity guard)
  if (!check(launch_demap(false, 0, d_spectra_.get(), 0, nframes, d_frame_slot_.get(), d_frame_cif_row_.get(), d_qpsk_.get(), d_fic_bits_.get(), d_msc_bits_.get(), GuardArgs{}, stream_), "demap launch") ||
      !check(hipStreamSynchronize(stream_), "demap"))
    return -1;
  for (int j = 0; j < nframes; ++j)
    if (!unpack_tf_slot(j, fic + static_cast<size_t>(j) * kFicBits, msc + static_cast<size_t>(j) * kMscBits)) return -1;
  return nframes;
}

int Engine::stage_fic_decode(const uint8_t* fic, int nframes, uint8_t* fibs, uint8_t* crc_ok)
{
  if (!hard_only("stage_fic_decode")) return -1;
  if (!ok_) { set_error("engine not initialised (no GPU?)"); return -1; }
  if (nframes <= 0) return 0;
  if (!reserve_tf_slots(nframes)) return -1;
  std::vector<uint32_t> words(static_cast<size_t>(nframes) * kFicWords);
  for (int j = 0; j < nframes; ++j) pack_bits(fic + static_cast<size_t>(j) * kFicBits, kFicBits, words.data() + static_cast<size_t>(j) * kFicWords);
  if (!check(blocking_copy(d_fic_bits_.get(), words.data(), words.size() * 4, hipMemcpyHostToDevice), "fic upload")) return -1;
  return fic_decode_slots(0, nframes, fibs, crc_ok) ? nframes : -1;
}

// Decision audit (calibration / test tool of the parity guard): nframes contiguous cu8 frames through K2 + K2b (natural
// layout), optionally with the guard, then decision_audit_kernel's fp64 transforms against the result.
// out8 = {decisions, disagreements with fp64, disagreements on carriers the guard rule does NOT flag, decisions the rule flags,
//         max |X32 - X64| / sqrt(symbol energy), max product error / (|cur|_1 s(l-1) + |prev|_1 s(l)), max residual product
//         error / (|cur|_1 |prev|_1), entries the demapper listed (guard on)}
// fused = true (round 5): the same audit of the kernel the DEFAULT decode runs -- ofdm_demap_kernel's guarded build, through its audit build (the same source
// lines plus stores of its bins and products; k_fused.hip) -- with the frames laid out as a decode lays them out (FIC slot j, logical CIF rows from kRowLead +
// 4 j).  out8[7] = entries that kernel listed.  out_extra (2 values, may be null): {1 when the SHIPPING build (launch_ofdm_demap_fused_guarded) run on the
// same frames leaves exactly the bits the audit build left, before any re-decision; 1 when it lists the same number of decisions}.
int Engine::stage_decision_audit(const uint8_t* frames, int nframes, bool on_device, bool guard_on, double* out8, bool fused, double* out_extra)
{
  if (!ok_) { set_error("engine not initialised (no GPU?)"); return -1; }
  if (!hard_only("stage_decision_audit")) return -1;
  if (nframes <= 0 || !out8) return 0;
  if (fused) return stage_decision_audit_fused(frames, nframes, on_device, guard_on, out8, out_extra);
  struct AuditOut { unsigned long long decisions, disagree, outside, flagged; unsigned bin_bits, dec_bits, prod_bits, pad; };
  DeviceBuffer<uint8_t> d_out;
  if (!d_out.reserve(sizeof(AuditOut)) || !check(hipMemsetAsync(d_out.get(), 0, sizeof(AuditOut), stream_), "audit memset")) return -1;
  const int chunk = 256;
  uint64_t listed = 0;
  const uint8_t* d_in = frames;
  if (!on_device) {
    if (!d_iq_own_.reserve(static_cast<size_t>(nframes) * kTfBytes) ||
        !check(blocking_copy(d_iq_own_.get(), frames, static_cast<size_t>(nframes) * kTfBytes, hipMemcpyHostToDevice), "frame upload"))
      return -1;
    d_in = d_iq_own_.get();
  }
  for (int first = 0; first < nframes; first += chunk) {
    const int n = std::min(chunk, nframes - first);
    std::vector<CallDesc> descs(n);
    std::vector<int2> list(n);
    std::vector<int> slots(n), rows(n);
    for (int j = 0; j < n; ++j) {
      std::memset(&descs[j], 0, sizeof(CallDesc));
      descs[j].status = 2;
      descs[j].ordinal = j;
      descs[j].view = initial_state().view;
      descs[j].view.seg_src[0] = static_cast<int64_t>(first + j) * kTfBytes;
      list[j] = make_int2(0, j);
      slots[j] = j;
      rows[j] = 4 * j;
    }
    std::vector<const uint8_t*> ptrs = {d_in};
    max_calls_ = n;
    if (!reserve_tf_slots(n) || !d_iq_ptrs_.upload(ptrs, stream_) || !d_descs_.upload(descs, stream_) || !d_frames_.upload(list, stream_) ||
        !d_frame_slot_.upload(slots, stream_) || !d_frame_cif_row_.upload(rows, stream_) || !d_spectra_.reserve(static_cast<size_t>(n) * kSymbolsPerTf * 2048))
      return -1;
    GuardArgs ga{};
    guard_launches_ = 0;
    guard_counters_clear_ = false;
    guard_flagged_ = 0;
    if (guard_on && (!d_delta_.reserve(static_cast<size_t>(n) * kSymbolsPerTf) || !guard_begin(n, &ga) ||
                     !check(launch_symbol_delta(d_iq_ptrs_.get(), d_descs_.get(), n, d_frames_.get(), 0, n, kSymbolsPerTf, d_delta_.get(), kSymbolsPerTf, guard_c_of(guard_rule_level()), stream_), "symbol delta launch")))
      return -1;
    if (!check(launch_ofdm_fft(d_iq_ptrs_.get(), d_descs_.get(), n, d_frames_.get(), 0, n, d_spectra_.get(), d_twf_.get(), stream_), "fft launch") ||
        !check(launch_demap(false, 0, d_spectra_.get(), 0, n, d_frame_slot_.get(), d_frame_cif_row_.get(), d_qpsk_.get(), d_fic_bits_.get(), d_msc_bits_.get(), ga, stream_), "demap launch") ||
        (guard_on && !guard_finish(false, 0, n, 1, kSymbolsPerTf, false)) ||
        !check(launch_decision_audit(d_in + static_cast<size_t>(first) * kTfBytes, n, d_spectra_.get(), d_fic_bits_.get(), d_msc_bits_.get(), d_tw2048_.get(), d_qpsk_.get(), d_out.get(), stream_, nullptr, 0, guard_rule_level()), "audit launch") ||
        (guard_on && !guard_download()) || !check(hipStreamSynchronize(stream_), "audit") || (guard_on && !guard_check()))
      return -1;
    listed += static_cast<uint64_t>(guard_flagged_);
  }
  AuditOut h;
  if (!check(blocking_copy(&h, d_out.get(), sizeof h, hipMemcpyDeviceToHost), "audit download")) return -1;
  auto f = [](unsigned bits) { float v; std::memcpy(&v, &bits, 4); return static_cast<double>(v); };
  out8[0] = static_cast<double>(h.decisions); out8[1] = static_cast<double>(h.disagree); out8[2] = static_cast<double>(h.outside);
  out8[3] = static_cast<double>(h.flagged); out8[4] = f(h.bin_bits); out8[5] = f(h.dec_bits); out8[6] = f(h.prod_bits); out8[7] = static_cast<double>(listed);
  return nframes;
}

int Engine::stage_decision_audit_fused(const uint8_t* frames, int nframes, bool on_device, bool guard_on, double* out8, double* out_extra)
{
  struct AuditOut { unsigned long long decisions, disagree, outside, flagged; unsigned bin_bits, dec_bits, prod_bits, pad; };
  DeviceBuffer<uint8_t> d_out;
  DeviceBuffer<float2> d_bins, d_prod;
  if (!d_out.reserve(sizeof(AuditOut)) || !check(hipMemsetAsync(d_out.get(), 0, sizeof(AuditOut), stream_), "audit memset")) return -1;
  const int chunk = 128;
  uint64_t listed = 0;
  bool bits_equal = true, list_equal = true;
  const uint8_t* d_in = frames;
  if (!on_device) {
    if (!d_iq_own_.reserve(static_cast<size_t>(nframes) * kTfBytes) ||
        !check(blocking_copy(d_iq_own_.get(), frames, static_cast<size_t>(nframes) * kTfBytes, hipMemcpyHostToDevice), "frame upload"))
      return -1;
    d_in = d_iq_own_.get();
  }
  const size_t per_frame = static_cast<size_t>(kSymbolsPerTf) * 2048;
  if (!d_bins.reserve(per_frame * chunk) || !d_prod.reserve(per_frame * chunk)) return -1;
  std::vector<uint32_t> bits_a, bits_b;
  for (int first = 0; first < nframes; first += chunk) {
    const int n = std::min(chunk, nframes - first);
    std::vector<CallDesc> descs(n);
    std::vector<int2> list(n);
    std::vector<int> slots(n), rows(n);
    for (int j = 0; j < n; ++j) {
      std::memset(&descs[j], 0, sizeof(CallDesc));
      descs[j].status = 2;
      descs[j].ordinal = j;
      descs[j].view = initial_state().view;
      descs[j].view.seg_src[0] = static_cast<int64_t>(first + j) * kTfBytes;
      list[j] = make_int2(0, j);
      slots[j] = j;
      rows[j] = kRowLead + 4 * j;                         // where a decode puts the frame's first CIF: the scatter reaches kRowLead rows back
    }
    std::vector<const uint8_t*> ptrs = {d_in};
    max_calls_ = n;
    if (!reserve_tf_slots(n) || !d_iq_ptrs_.upload(ptrs, stream_) || !d_descs_.upload(descs, stream_) || !d_frames_.upload(list, stream_) ||
        !d_frame_slot_.upload(slots, stream_) || !d_frame_cif_row_.upload(rows, stream_) || !d_delta_.reserve(static_cast<size_t>(n) * kSymbolsPerTf))
      return -1;
    const size_t fic_words = static_cast<size_t>(n) * kFicWords, msc_words = static_cast<size_t>(4 * n + kRowLead + 1) * kCifWords;
    // two passes: the shipping build first (its raw bits and its list count kept), then the audit build, whose output the audit kernel reads
    uint32_t counts[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {
      GuardArgs ga{};
      guard_launches_ = 0;
      guard_counters_clear_ = false;
      guard_flagged_ = 0;
      if (!check(hipMemsetAsync(d_msc_bits_.get(), 0, msc_words * 4, stream_), "row clear")) return -1;     // (the rows before the first frame's are never written)
      for (int part = 0; part < 2; ++part) {              // the decode's own two launches: FIC symbols, then MSC symbols, one workgroup per frame each
        const int sym_a = part ? 4 : 1, sym_b = part ? kSymbolsPerTf : 4;
        if (!guard_begin(n, &ga)) return -1;
        const hipError_t e = pass == 0
            ? launch_ofdm_demap_fused_guarded(d_iq_ptrs_.get(), d_descs_.get(), n, d_frames_.get(), 0, n, d_twf_.get(), d_frame_slot_.get(), d_frame_cif_row_.get(),
                                              d_qpsk_.get(), d_fic_bits_.get(), d_msc_bits_.get(), ga, stream_, sym_a, sym_b, 1)
            : launch_ofdm_demap_fused_audit(d_iq_ptrs_.get(), d_descs_.get(), n, d_frames_.get(), 0, n, d_twf_.get(), d_frame_slot_.get(), d_frame_cif_row_.get(),
                                            d_qpsk_.get(), d_fic_bits_.get(), d_msc_bits_.get(), ga, stream_, sym_a, sym_b, 1, d_bins.get(), d_prod.get());
        if (!check(e, "fused audit launch")) return -1;
        if (pass == 1 && guard_on) {
          if (!guard_finish(true, 0, n, sym_a, sym_b, false)) return -1;
        } else {
          ++guard_launches_;                              // (guard_finish counts the launch; without it the list is only counted, never acted on)
        }
      }
      std::vector<uint32_t>& keep = pass == 0 ? bits_a : bits_b;
      keep.resize(fic_words + msc_words);
      const bool raw = !(pass == 1 && guard_on);          // bits as the kernel left them
      if (raw && (!check(hipMemcpyAsync(keep.data(), d_fic_bits_.get(), fic_words * 4, hipMemcpyDeviceToHost, stream_), "bits download") ||
                  !check(hipMemcpyAsync(keep.data() + fic_words, d_msc_bits_.get(), msc_words * 4, hipMemcpyDeviceToHost, stream_), "bits download")))
        return -1;
      if (pass == 1 &&
          !check(launch_decision_audit(d_in + static_cast<size_t>(first) * kTfBytes, n, d_bins.get(), d_fic_bits_.get(), d_msc_bits_.get(), d_tw2048_.get(), d_qpsk_.get(),
                                       d_out.get(), stream_, d_prod.get(), kRowLead, guard_rule_level()),
                 "audit launch"))
        return -1;
      if (!guard_download() || !check(hipStreamSynchronize(stream_), "audit") || !guard_check()) return -1;
      counts[pass] = static_cast<uint32_t>(guard_flagged_);
    }
    listed += counts[1];
    list_equal = list_equal && counts[0] == counts[1];
    if (!guard_on) bits_equal = bits_equal && bits_a == bits_b;
  }
  AuditOut h;
  if (!check(blocking_copy(&h, d_out.get(), sizeof h, hipMemcpyDeviceToHost), "audit download")) return -1;
  auto f = [](unsigned bits) { float v; std::memcpy(&v, &bits, 4); return static_cast<double>(v); };
  out8[0] = static_cast<double>(h.decisions); out8[1] = static_cast<double>(h.disagree); out8[2] = static_cast<double>(h.outside);
  out8[3] = static_cast<double>(h.flagged); out8[4] = f(h.bin_bits); out8[5] = f(h.dec_bits); out8[6] = f(h.prod_bits); out8[7] = static_cast<double>(listed);
  if (out_extra) {
    out_extra[0] = guard_on ? -1.0 : (bits_equal ? 1.0 : 0.0);      // (compared on the raw bits only: with the guard on the audit pass's bits are the re-decided ones)
    out_extra[1] = list_equal ? 1.0 : 0.0;
  }
  return nframes;
}

// S1: n code words of `framebits` data bits, symbols 127/129 hard, 128 erased (depuncture.c:36-43)
int Engine::viterbi_batch(const uint8_t* symbols, uint8_t* data, int framebits, int n)
{
  if (!ok_) { set_error("engine not initialised (no GPU?)"); return -1; }
  if (n <= 0) return 0;
  if (framebits <= 0 || framebits % 32 != 0) { set_error("viterbi: framebits must be a positive multiple of 32"); return -1; }
  const int nsteps = framebits + 6, n16 = (nsteps + 15) / 16;
  const int ngroups = (n + 63) / 64;
  CodewordPlan plan;
  std::memset(&plan, 0, sizeof plan);
  plan.nsteps = nsteps;
  plan.out_bytes = framebits / 8;
  const int pid = plan_table_.id(plan);
  std::vector<WaveGroup> groups;
  const int64_t dr = (nsteps + 7) / 8 * 8;
  std::vector<uint4> steps(static_cast<size_t>(ngroups) * n16 * 64, make_uint4(0, 0, 0, 0));
  for (int g = 0; g < ngroups; ++g) {
    groups.push_back(WaveGroup{pid, 64 * g, std::min(64, n - 64 * g), nsteps, static_cast<int64_t>(g) * n16, g * dr});
    for (int l = 0; l < 64; ++l) {
      const int cw = g * 64 + l;
      if (cw >= n) continue;
      const uint8_t* sym = symbols + static_cast<size_t>(cw) * 4 * nsteps;
      for (int t = 0; t < nsteps; ++t) {
        unsigned byte = 0;
        for (int j = 0; j < 4; ++j) {
          const uint8_t sv = sym[4 * t + j];
          if (sv != 128) byte |= (1u << (4 + j)) | ((sv > 128 ? 1u : 0u) << j);
        }
        uint4& u = steps[(static_cast<size_t>(g) * n16 + t / 16) * 64 + l];
        uint32_t* w = &u.x;
        w[(t % 16) / 4] |= byte << (8 * (t % 4));
      }
    }
  }
  // no gather: upload the step rows directly, then run the decoder with an all-zero scrambler
  const size_t out_bytes = static_cast<size_t>(n) * (framebits / 8);
  if (framebits / 32 > 1024) { set_error("viterbi: code word too long"); return -1; }
  if (!d_plans_.upload(plan_table_.plans(), stream_) || !d_groups_.upload(groups, stream_) || !d_steps_.upload(steps, stream_) ||
      !d_decisions_.reserve(static_cast<size_t>(ngroups) * dr * 64) || !d_bytes_.reserve(out_bytes))
    return -1;
  if (!check(launch_viterbi(d_groups_.get(), ngroups, nullptr, d_plans_.get(), d_steps_.get(), d_decisions_.get(), d_zero_words_.get(),
                            d_bytes_.get(), framebits / 8, stream_),
             "viterbi launch") ||
      !check(hipMemcpyAsync(data, d_bytes_.get(), out_bytes, hipMemcpyDeviceToHost, stream_), "decoded download") ||
      !check(hipStreamSynchronize(stream_), "viterbi"))
    return -1;
  return n;
}

// ---------------------------------------------------------------------------------------------
// S2 building blocks: one sdr_demod call on an explicit stream
bool Engine::scan_one_call(const uint8_t* iq_virtual_base, StreamState* d_state, uint8_t* d_tail, int call, int chunk, CallDesc* out)
{
  std::vector<const uint8_t*> ptrs = {iq_virtual_base};
  std::vector<int64_t> nb = {static_cast<int64_t>(call + 1) * kChunkBytes};     // (only bounds the kernel's call loop: this is call number `call`, whatever its length)
  if (!d_iq_ptrs_.upload(ptrs, stream_) || !d_nbytes_.upload(nb, stream_) || !d_descs_.reserve(1) || !d_tail_images_.reserve(kTailBytes)) return false;
  // the kernel indexes descs[stream * max_calls + call]; with max_calls = 0 and the pointer moved back by `call` it hits slot 0 (the tail copy likewise)
  if (!check(launch_sync_scan(d_iq_ptrs_.get(), d_nbytes_.get(), d_state, d_descs_.get() - call, nullptr, 1, 0, call, call + 1, d_tw2048_.get(),
                              d_tw1536_.get(), d_prs_.get(), 0, stream_, false, nullptr, nullptr,
                              SyncTails{d_tail, d_tail, d_tail_images_.get() - static_cast<ptrdiff_t>(call) * kTailBytes, chunk}),
             "sync scan launch"))
    return false;
  return check(hipMemcpyAsync(out, d_descs_.get(), sizeof(CallDesc), hipMemcpyDeviceToHost, stream_), "desc download") &&
         check(hipStreamSynchronize(stream_), "sync scan");
}

bool Engine::demod_one_frame(const uint8_t* iq_virtual_base, const CallDesc& desc, uint8_t* fic_bytes, uint8_t* msc_bytes)
{
  std::vector<const uint8_t*> ptrs = {iq_virtual_base};
  std::vector<int2> list = {make_int2(0, 0)};
  std::vector<int> slots = {0};
  std::vector<CallDesc> d = {desc};
  if (!reserve_tf_slots(1) || !d_iq_ptrs_.upload(ptrs, stream_) || !d_descs_.upload(d, stream_) || !d_frames_.upload(list, stream_) ||
      !d_frame_slot_.upload(slots, stream_) || !d_frame_cif_row_.upload(slots, stream_) ||
      !d_spectra_.reserve(static_cast<size_t>(kSymbolsPerTf) * 2048))
    return false;
  max_calls_ = 1;                                         // the one descriptor uploaded above is frame {0, 0}
  const bool guard = guard_active();
  GuardArgs ga{};
  guard_launches_ = 0;
  guard_counters_clear_ = false;
  if (guard && (!d_delta_.reserve(kSymbolsPerTf) || !guard_begin(1, &ga) ||
                !check(launch_symbol_delta(d_iq_ptrs_.get(), d_descs_.get(), 1, d_frames_.get(), 0, 1, kSymbolsPerTf, d_delta_.get(), kSymbolsPerTf, guard_c_of(guard_rule_level()), stream_), "symbol delta launch")))
    return false;
  if (!check(launch_ofdm_fft(d_iq_ptrs_.get(), d_descs_.get(), 1, d_frames_.get(), 0, 1, d_spectra_.get(), d_twf_.get(), stream_), "fft launch") ||
      !check(launch_demap(false, 0, d_spectra_.get(), 0, 1, d_frame_slot_.get(), d_frame_cif_row_.get(), d_qpsk_.get(), d_fic_bits_.get(), d_msc_bits_.get(), ga, stream_), "demap launch") ||
      (guard && !guard_finish(false, 0, 1, 1, kSymbolsPerTf, false)) || (guard && !guard_download()) ||
      !check(hipStreamSynchronize(stream_), "demod") || (guard && !guard_check()))
    return false;
  return unpack_tf_slot(0, fic_bytes, msc_bytes);
}

}  // namespace dabhip
