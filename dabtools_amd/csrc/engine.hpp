// engine.hpp — batch engine: B independent cu8 IQ streams resident in HBM -> ETI frames.
//
// Pipeline of one decode() (kernel files in brackets):
//   K1  sync scan            [k_sync.hip]   one workgroup per stream, sequential over its TFs
//   K2  OFDM FFT             [k_fft.hip]    all demodulated TFs, chunked to bound the spectra buffer
//   K2b DQPSK/demap          [k_fft.hip]    -> bit-packed FIC / MSC rows, kept for the whole batch
//   K3  FIC decode           [k_decode.hip] gather + Viterbi + CRC -> 12 FIBs per TF  (D2H: 396 B/TF)
//   --  control plane        [control_plane.hpp, host] lock FSM, CIF ring, ETI headers, work lists
//   K4  MSC decode           [k_decode.hip] gather (time de-interleave + de-puncture) + Viterbi
//   K5  ETI finish           [k_decode.hip] header, FIBs, EOF CRC, trailer
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "control_plane.hpp"
#include "device_types.hpp"
#include "thread_pool.hpp"
#include "worklist.hpp"

namespace dabhip {

void set_error(const std::string& msg);

// The HIP runtime keeps its records of a stream's finished commands until somebody synchronises THAT STREAM: an event or a blocking copy that waits
// for the command itself does not let them go (tools/hip_retained_commands.py, ROCm 7.2: 970 bytes of heap per hipMemcpy into pageable memory on the
// null stream, 2.0 KB per asynchronous copy + event record + event synchronise on a stream of its own, for ever; nothing with a stream synchronise
// now and then).  A receiver that runs for weeks makes millions of such calls, so every path that is called once per buffer, frame or segment and
// does not end in a stream synchronise anyway either goes through the helper below or synchronises its stream every kReapEvery uses (the ETI fetches,
// the sessions' prefetch stream): bounded books, and a wait that is free when the stream is idle.
constexpr uint32_t kReapEvery = 32;
inline bool reap_enabled()             // DABHIP_NO_REAP=1: the behaviour before (for tools/soak_cli.py's "before" column only)
{
  static const bool on = std::getenv("DABHIP_NO_REAP") == nullptr;
  return on;
}
inline hipError_t blocking_copy(void* dst, const void* src, size_t nbytes, hipMemcpyKind kind)      // hipMemcpy on the null stream
{
  static std::atomic<uint32_t> calls{0};
  const hipError_t e = hipMemcpy(dst, src, nbytes, kind);
  if (e == hipSuccess && calls.fetch_add(1, std::memory_order_relaxed) % kReapEvery == kReapEvery - 1 && reap_enabled()) (void)hipStreamSynchronize(nullptr);
  return e;
}

// grow-only device allocation; contents are NOT preserved on growth
template <class T>
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { release(); }
  bool reserve(size_t n)
  {
    if (n <= cap_) return true;
    release();
    if (hipMalloc(reinterpret_cast<void**>(&p_), n * sizeof(T)) != hipSuccess) {
      p_ = nullptr;
      set_error("hipMalloc of " + std::to_string(n * sizeof(T)) + " bytes failed");
      return false;
    }
    cap_ = n;
    return true;
  }
  void release()
  {
    if (p_) (void)hipFree(p_);
    p_ = nullptr;
    cap_ = 0;
  }
  bool upload(const T* src, size_t n, hipStream_t s)
  {
    if (!reserve(n ? n : 1)) return false;
    return n == 0 || hipMemcpyAsync(p_, src, n * sizeof(T), hipMemcpyHostToDevice, s) == hipSuccess;
  }
  template <class A>
  bool upload(const std::vector<T, A>& v, hipStream_t s) { return upload(v.data(), v.size(), s); }
  T* get() const { return p_; }
  size_t capacity() const { return cap_; }

 private:
  T* p_ = nullptr;
  size_t cap_ = 0;
};

// std::allocator over page-locked host memory: vectors that are uploaded every decode (the work lists) go to the device
// as plain asynchronous DMA instead of staged copies that block the issuing thread
template <class T>
struct PinnedAllocator {
  using value_type = T;
  PinnedAllocator() = default;
  template <class U>
  PinnedAllocator(const PinnedAllocator<U>&) {}
  T* allocate(size_t n)
  {
    void* p = nullptr;
    if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, size_t) { (void)hipHostFree(p); }
  template <class U>
  bool operator==(const PinnedAllocator<U>&) const { return true; }
  template <class U>
  bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
template <class T>
using HostList = std::vector<T, PinnedAllocator<T>>;

// grow-only page-locked host allocation (device copies to and from it run at full PCIe rate and truly asynchronously)
template <class T>
class PinnedBuffer {
 public:
  PinnedBuffer() = default;
  PinnedBuffer(const PinnedBuffer&) = delete;
  PinnedBuffer& operator=(const PinnedBuffer&) = delete;
  ~PinnedBuffer() { if (p_) (void)hipHostFree(p_); }
  bool resize(size_t n)
  {
    size_ = n;
    if (n <= cap_) return true;
    if (p_) (void)hipHostFree(p_);
    p_ = nullptr;
    cap_ = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&p_), n * sizeof(T), hipHostMallocDefault) != hipSuccess) {
      p_ = nullptr;
      size_ = 0;
      set_error("hipHostMalloc of " + std::to_string(n * sizeof(T)) + " bytes failed");
      return false;
    }
    cap_ = n;
    return true;
  }
  T* data() const { return p_; }
  size_t size() const { return size_; }
  T& operator[](size_t i) const { return p_[i]; }

 private:
  T* p_ = nullptr;
  size_t cap_ = 0, size_ = 0;
};

struct StageTimes {
  float sync = 0, fft = 0, demap = 0, fic = 0, control = 0, gather = 0, viterbi = 0, eti = 0;
  float setup = 0, frames = 0, worklist = 0, wall = 0;   // host-side phases (wall clock)
  float sync_fp64_calls = 0;                             // K1 verification: calls whose fp32 arg-max was not clear-cut (decided in fp64)
  float sync_spec_calls = 0;                             // K1 chain: calls whose estimators came out of the look-ahead pass's table (0: the plain chain ran)
  float h2d = 0;                                         // host-fed decode: upload of the IQ (HIP events; 0 when the IQ was resident)
  double h2d_bytes = 0, h2d_pinned_bytes = 0;            // bytes uploaded, and how many of them came from page-locked memory
};

// the MSC decode's work lists (worklist.hpp) in page-locked memory: they go to the device as plain asynchronous DMA
using DecodeBatch = DecodeBatchT<PinnedAllocator>;
using MscWork = MscWorkT<PinnedAllocator>;

class Engine {
 public:
  // host_threads: threads of the per-stream control-plane pool (0 = half the cores, at most 24; DABHIP_HOST_THREADS overrides)
  // cpus: the CPUs this engine's host threads (control-plane pool, host lane) are bound to; empty = those of the device's NUMA node on a machine
  // with more than one (placement.hpp), or none
  explicit Engine(int device, int host_threads = 0, std::vector<int> cpus = {});
  const std::vector<int>& host_cpus() const { return host_cpus_; }
  int numa_node() const { return numa_node_; }
  ~Engine();
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  bool ok() const { return ok_; }
  // Lanes of one batch engine share this lock around their GPU-saturating phases (FFT/demap/FIC and MSC decode), so
  // those never overlap each other; only the light phases (sync scan, host control plane) run beside them.
  void set_heavy_lock(std::mutex* m) { heavy_mu_ = m; }
  const uint8_t* eti_buffer() const { return d_eti_.get(); }
  // software AFC: an NCO per stream steered by the reference's tuner rule; off = parity mode
  void set_afc(bool on) { afc_ = on; }
  // soft-decision decoding (not in the reference): 4-bit soft values from the demapper through
  // de-interleaving and de-puncturing into the Viterbi branch metrics.  Batch path only; off = parity mode.
  // K2 + K2b as one kernel that never writes the spectra (k_fused.hip; hard decisions only; the default) or as the two
  // kernels K2 (the HBM-roofline stage of SURVEY.md 8(d), always measurable on its own: fft_roofline) and K2b.  The output
  // bits are identical either way.
  void set_fused(bool on) { fused_ = on; }
  // K1's chain: 0 = call after call, 1 = with the look-ahead pass, -1 = the pass for small batches (include/dabhip.h: dabhip_engine_set_sync_speculation)
  void set_sync_speculation(int mode) { spec_mode_ = mode < 0 ? -1 : (mode > 0 ? 1 : 0); }
  // sub-channel filter (TODO.md:28-31): bit i = SubChId i is decoded and carried in the ETI frames; takes effect with the next
  // decode() / first segment of a session.  All ones (default) = the reference's frames.
  void set_subchannel_filter(uint64_t keep) { subch_keep_ = keep; }
  void set_soft(bool on) { soft_bits_ = on ? 4 : 0; tf_slots_ = 0; msc_rows_ = 0; }
  // parity guard (k_parity.hip; default on): hard decisions whose fp32 margin lies inside the error band of the fp32 OFDM
  // transform are re-decided in fp64 from the int8 samples, so the demapped bits are those of exact arithmetic (what the
  // reference's fp64 FFTW path yields).  Inactive with soft decisions and with the software AFC (no reference semantics there).
  // level: 0 = off (raw fp32 decisions), 1 = the measured constants, 2 = the proven ones (device_types.hpp: kGuardC.. / kGuardCProven..)
  void set_parity_guard(int level) { guard_level_ = level < 0 ? kDefaultGuardLevel : (level > 2 ? 2 : level); }   // < 0: the default level
  int parity_guard_level() const { return guard_level_; }
  // decisions flagged and re-decided by the guard in the last decode (FIC pre-pass + OFDM stage), and decisions taken
  // the reference's operator messages of one stream (ControlPlane::take_log): text pending since the last call, cleared by it
  std::string take_stream_log(int stream) { return (stream >= 0 && stream < static_cast<int>(planes_.size())) ? planes_[static_cast<size_t>(stream)].take_log() : std::string(); }
  void guard_stats(int64_t* flagged, int64_t* decisions) const { if (flagged) *flagged = guard_flagged_; if (decisions) *decisions = guard_decisions_; }
  int guard_overflows() const { return guard_overflows_; }
  void set_guard_list_cap(uint32_t cap) { guard_cap_override_ = cap; }   // test knob: a tiny list makes the overflow path run

  // -- batch path ---------------------------------------------------------------------------
  int64_t decode(const uint8_t* const* iq, const size_t* nbytes, int nstreams, bool on_device);
  // -- streaming sessions: decode() of an unbounded stream, one segment at a time ------------
  int64_t feed(const uint8_t* const* iq_virtual, const size_t* avail, int nstreams, bool first_segment);
  int64_t stream_need_from(int stream) const;   // oldest stream byte the next segment may still read
  int64_t eti_count(int stream) const;
  // StreamFault bits (control_plane.hpp) of the stream's multiplex in the last decode / so far in the session: such a stream emits no frames while
  // its signalled multiplex is one the reference could not assemble inside its own arrays; every other stream of the batch is unaffected
  uint32_t stream_status(int stream) const;
  int64_t eti_read(int stream, uint8_t* dst, int64_t cap_frames);
  // All frames of the last decode / segment (stream-major, emission order) to host memory on a stream of their own, without waiting: the copy
  // runs beside the NEXT decode's scan and OFDM stage (only its K4, which rewrites the ETI buffer, waits for it).  dst should be page-locked.
  // eti_fetch_wait() returns when the bytes are there.  The output side of the CLI contract (dab2eti.c:132-135) at the link's rate.
  int64_t eti_fetch_async(uint8_t* dst, int64_t cap_frames);
  bool eti_fetch_wait();
  const uint8_t* eti_device(int64_t* nframes) const;
  int trace(int stream, int32_t* ints6, double* ffs, int cap_calls) const;
  int trace_nco(int stream, int32_t* nco_hz, int cap_calls) const;      // software AFC: the frequency each call's samples were de-rotated by
  const StageTimes& stage_times() const { return times_; }
  void fft_stats(int64_t* launches, int64_t* tfs, double* ms) const;
  int fft_roofline(int reps, int64_t* launches, int64_t* tfs, double* ms);   // K2 alone over the last decode's frames

  // -- stage entries --------------------------------------------------------------------------
  int stage_ofdm_fft(const uint8_t* frames, int nframes, float* spectra, bool on_device, int reps, float* kernel_ms);
  int stage_demap(const float* spectra, int nframes, uint8_t* fic, uint8_t* msc);
  int stage_fic_decode(const uint8_t* fic, int nframes, uint8_t* fibs, uint8_t* crc_ok);
  int stage_decision_audit(const uint8_t* frames, int nframes, bool on_device, bool guard_on, double* out8, bool fused = false, double* out_extra = nullptr);
  int stage_decision_audit_fused(const uint8_t* frames, int nframes, bool on_device, bool guard_on, double* out8, double* out_extra);
  int viterbi_batch(const uint8_t* symbols, uint8_t* data, int framebits, int n);

  // -- building blocks shared with the streaming seams (capi.cpp) ------------------------------
  // FIC rows / FIB records for nslots TFs and msc_rows logical CIF rows (default 4 per slot + 15 lead-in + 1)
  bool reserve_tf_slots(int nslots, int msc_rows = -1);
  // host 0/1 bytes of one demapped TF -> bit rows of TF slot `slot`
  bool store_tf_bytes(int slot, const uint8_t* fic_bytes, const uint8_t* msc_bytes);
  // S3 recycling: keep the newest `keep_slots` TF slots (and the logical CIF rows still being filled) at the front
  bool recycle_tf_slots(int used_slots, int keep_slots);
  // FIC-decode TF slots [first, first+n): FIBs and CRC flags to host
  bool fic_decode_slots(int first, int n, uint8_t* fibs_host, uint8_t* ok_host);
  bool fic_decode_slots_async(int first, int n, uint8_t* fibs_host, uint8_t* ok_host, hipStream_t copy);   // completion: ev_fibs_
  // decode the ETI frames described by the per-stream job lists into the ETI buffer (stream-major order)
  // stream_row_base[b]: logical CIF row of stream b's CIF 0; stream_fib_base[b]: FIB block (4 per TF slot) of its CIF 0
  bool msc_decode(const std::vector<const JobList*>& stream_jobs, const std::vector<const ControlPlane*>& planes,
                  const std::vector<int>& stream_row_base, const std::vector<int>& stream_fib_base)
  {
    return msc_prepare(stream_jobs, planes, stream_row_base, stream_fib_base, work_s3_) && msc_run(work_s3_);   // reused: its lists are page-locked
  }
  // host half (work lists, headers, plans) and GPU half (regroup, Viterbi, ETI finish) of msc_decode
  bool msc_prepare(const std::vector<const JobList*>& stream_jobs, const std::vector<const ControlPlane*>& planes,
                   const std::vector<int>& stream_row_base, const std::vector<int>& stream_fib_base, MscWork& out);
  bool msc_run(MscWork& w);
  bool msc_upload(const MscWork& w, hipStream_t s);
  bool msc_launch(const MscWork& w);
  bool read_eti(int64_t first, int64_t n, uint8_t* dst);
  // the demapped values (hard: 0 / 1, soft: -7 .. 7) of one TF of the last decode, in the reference's hand-off order (dab.h:27-33)
  bool read_demapped_tf(int stream, int tf, int8_t* fic_out, int8_t* msc_out);
  // front end on an explicit single stream (S2 seam): calls [call, call+1)
  // d_tail: the seam's kTailBytes tail bytes (device, zero at sdr_init); chunk: bytes this call appended (input_buffer_len)
  bool scan_one_call(const uint8_t* iq_virtual_base, StreamState* d_state, uint8_t* d_tail, int call, int chunk, CallDesc* out);
  bool demod_one_frame(const uint8_t* iq_virtual_base, const CallDesc& desc, uint8_t* fic_bytes, uint8_t* msc_bytes);
  hipStream_t stream() const { return stream_; }
  int device() const { return device_; }

 private:
  bool check(hipError_t e, const char* what);
  bool record(hipEvent_t e, hipStream_t s);
  bool elapsed(float* ms, hipEvent_t a, hipEvent_t b);
  bool hard_only(const char* what);
  int64_t decode_impl(const uint8_t* const* iq, const size_t* nbytes, int nstreams, bool on_device, bool cont, bool full_scan = false);
  bool begin_decode(int nstreams, bool cont);
  // layout: called when the calls' {status, ordinal} are on the host (h_info_) -- early in the split scan, again after a re-scan
  bool scan_streams(const uint8_t* const* iq, const size_t* nbytes, int nstreams, bool on_device, bool cont, bool full_scan,
                    const std::function<bool()>& layout);
  // host-fed decode: the streams' bytes into d_iq_own_ (ptrs[b] = where stream b landed), queued on the main stream
  bool upload_iq(const uint8_t* const* iq, const size_t* nbytes, int nstreams, const uint8_t** ptrs);
  bool carry_and_reserve(const std::vector<int>& tf_base, const std::vector<int>& row_base, int nslots, int nrows);
  // MSC decode batch: work lists to the device, regroup + fused Viterbi launches
  bool upload_decode_batch(const DecodeBatch& b, const HostList<DecodeJob>& jobs, hipStream_t s);
  // Several small host arrays to the device in one launch per four of them (launch_host_words: the kernel reads page-locked host memory itself) instead
  // of one copy-engine command each -- a small decode is made of those commands and the 5 .. 10 us of idle GPU between two of them.  Arrays that are
  // not page-locked are staged in `staging` first.  The staging words and the page-locked sources are read when the launch RUNS, not when it is queued:
  // every call site has its own SmallStage, whose event marks the end of its last launch -- a call that finds that launch still in flight waits for it
  // before it overwrites the words (no caller does that today: every one drains its stream in between; the wait turns a silent overwrite into a stall).  Large lists
  // (more than kSmallUploadBytes in all) go as plain asynchronous copies, as before.  dst: device memory, reserved by the caller.
  struct SmallUpload {
    const void* src;
    void* dst;
    size_t bytes;
    bool pinned;
  };
  static constexpr size_t kSmallUploadBytes = 256 * 1024;
  struct SmallStage {
    PinnedBuffer<uint32_t> words;
    hipEvent_t done = nullptr;               // recorded behind the stage's last launch
    bool armed = false;
    ~SmallStage() { if (done) (void)hipEventDestroy(done); }
  };
  bool upload_small(const SmallUpload* items, int n, hipStream_t s, SmallStage& staging);
  bool launch_decode_batch(const DecodeBatch& b, const uint32_t* bits, const int* d_stream_cif_base, const uint32_t* prbs, uint8_t* out,
                           int record_stride);
  bool msc_launch_async(const MscWork& w);   // K4 + K5 queued, nothing awaited
  void msc_collect();                        // their stage times, once the stream has been awaited
  bool unpack_tf_slot(int slot, uint8_t* fic_bytes, uint8_t* msc_bytes);

  // parity guard plumbing: list + counter for one launch, fix-up after it, entry count to the host (checked at the end)
  bool guard_active() const { return guard_level_ > 0 && soft_bits_ == 0 && !afc_; }
  int guard_rule_level() const { return guard_level_ > 0 ? guard_level_ : kDefaultGuardLevel; }   // the rule the audits count with when the guard is off
  bool guard_begin(int ntf_in_launch, GuardArgs* out);
  // the launch just queued covered frames [first, first + n) of the frame list, data symbols [sym_a, sym_b); skip_fic: another launch owns symbols 1..3
  bool guard_finish(bool planar, int first, int n, int sym_a, int sym_b, bool skip_fic);
  bool guard_download();
  bool guard_reserve_counters(int ntf);
  bool guard_check();

  bool ok_ = false;
  bool afc_ = false, fused_ = true;
  int guard_level_ = kDefaultGuardLevel;
  uint64_t subch_keep_ = ~0ull;
  int soft_bits_ = 0;
  // Decodes of at most this many code words (MSC: ETI frames x sub-channels; FIC: 4 per TF) run one WAVE per code word (k_vitwave.hip: latency
  // of a code word 0.1 instead of 1.4 ms) instead of one lane per code word (viterbi_fused_kernel: a sixth of the lane-ops).  DABHIP_VIT_WAVE_MAX.
  int wave_max_codewords_ = 12288, wave_max_fic_blocks_ = 3072;     // measured crossovers (tools/gpu/wavesweep.sh): MSC 5..6 streams x 64 TF, FIC 12..16
  // Above that, hard-decision decodes of at most this many groups of 64 code words run TWO LANES per code word (vit_two_lanes.hpp): while the lane form
  // would leave the SIMDs at one or two waves (8 .. 40 streams x 64 TF: decoder stage 1.45 -> 0.96 ms at 16 streams).  Measured crossover between 32 and
  // 64 streams (1,176 and 2,352 groups; profiles/r06_two_lanes_curve.txt).  DABHIP_VIT_TWO_LANES = 0 / 1 / N: never / always / at most N groups.
  int two_lanes_max_groups_ = 1536;
  // ... and of at most this many groups FOUR lanes per code word (vit_four_lanes.hpp): decoder stage 0.96 -> 0.82 ms at 8 and 16 streams, the same as
  // two lanes at 32 (1,176 groups; profiles/r06_lanes_curve.txt).  DABHIP_VIT_FOUR_LANES = 0 / 1 / N likewise;
  // DABHIP_VIT_LANES_PLAIN=1 (measurement) runs the two-lane decodes through that file's table-free two-lane form instead of vit_two_lanes.hpp's.
  int four_lanes_max_groups_ = 800;
  bool two_lanes_plain_ = false;
  // FIC decodes of at most this many tiles of 64 blocks (above the wave form's range: 12 .. 32 streams x 64 TF) run four lanes per block: FIC stage 0.32 -> 0.24 ms
  // at 16 streams, 0.58 -> 0.50 at 32, nothing from 64 streams (256 tiles) on.  DABHIP_FIC_FOUR_LANES = 0 / 1 / N
  int fic_four_lanes_max_tiles_ = 128;
  std::mutex* heavy_mu_ = nullptr;
  std::unique_ptr<ThreadPool> pool_;   // host threads for per-stream control-plane work
  std::unique_ptr<AsyncLane> host_lane_;   // the control-plane pass of a decode, beside its GPU work
  int device_ = 0, numa_node_ = -1;
  std::vector<int> host_cpus_;
  hipStream_t stream_ = nullptr, copy_stream_ = nullptr;   // copy_stream_: work-list uploads from the control-plane thread
  hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_upload_ = nullptr, ev_fic_ = nullptr, ev_fic_done_ = nullptr, ev_fibs_ = nullptr, ev_part0_ = nullptr, ev_chain_ = nullptr, ev_info_ = nullptr;
  hipEvent_t ev_msc_[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_h2d_[2] = {nullptr, nullptr};
  hipStream_t d2h_stream_ = nullptr;           // eti_fetch_async
  hipEvent_t ev_eti_fetch_[2] = {nullptr, nullptr};              // one per outstanding fetch (at most two)
  std::atomic<uint64_t> eti_fetch_issued_{0}, eti_fetch_waited_{0};   // (issued: the decoding thread; waited: whichever thread waits -- the CLI's writer)
  // page-locked staging ring for uploads from pageable memory: the host pool copies piece n + 1 into one buffer while the DMA of
  // piece n drains another
  static constexpr int kStageBufs = 4;
  PinnedBuffer<uint8_t> stage_buf_[kStageBufs];
  hipEvent_t stage_ev_[kStageBufs] = {nullptr, nullptr, nullptr, nullptr};
  bool msc_queued_ = false;
  std::vector<hipEvent_t> chunk_ev_;

  // constant tables
  DeviceBuffer<double2> d_tw2048_, d_tw1536_;
  DeviceBuffer<uint8_t> d_prs_;
  DeviceBuffer<float2> d_twf_;
  DeviceBuffer<uint16_t> d_qpsk_, d_qpsk_inv_, d_crc_tab_, d_crc_shift_;
  DeviceBuffer<uint32_t> d_prbs_, d_zero_words_;

  // batch state
  DeviceBuffer<uint8_t> d_iq_own_;
  DeviceBuffer<const uint8_t*> d_iq_ptrs_;
  DeviceBuffer<int64_t> d_nbytes_;
  DeviceBuffer<StreamState> d_states_, d_states_prev_;
  // the streams' tail bytes (device_types.hpp: kTailBytes): carried like the states, the incoming ones kept for a rescan, and one copy per call
  DeviceBuffer<uint8_t> d_tail_state_, d_tail_prev_, d_tail_images_;
  DeviceBuffer<int> d_viol_, d_redo_, d_calls_before_;
  PinnedBuffer<int> h_viol_, h_calls_before_;
  SmallStage h_small_fic_, h_small_msc_;   // staging of upload_small, one per call site
  // the look-ahead schedule of the K1 chain (small batches; k_sync.hip: sync_ahead_kernel): the estimators' table, the predicted start positions, the descriptor base
  DeviceBuffer<int2> d_spec_table_;
  DeviceBuffer<int64_t> d_spec_src0_;
  DeviceBuffer<int> d_spec_ctl_;
  PinnedBuffer<int> h_spec_hits_;
  int spec_mode_ = std::getenv("DABHIP_K1_SPEC") ? (std::atoi(std::getenv("DABHIP_K1_SPEC")) > 0 ? 1 : 0) : -1;
  int sync_rescanned_ = 0;            // streams the split scan had to scan again in full (last decode)
  DeviceBuffer<CallDesc> d_descs_;
  DeviceBuffer<int2> d_info_;
  DeviceBuffer<int2> d_frames_;
  DeviceBuffer<int> d_frame_slot_, d_frame_cif_row_, d_stream_cif_base_;
  DeviceBuffer<float2> d_spectra_;
  DeviceBuffer<uint32_t> d_fic_bits_, d_msc_bits_;
  DeviceBuffer<uint8_t> d_fibs_, d_fib_ok_;
  DeviceBuffer<DecodeJob> d_jobs_;
  DeviceBuffer<WaveGroup> d_groups_;
  DeviceBuffer<int> d_job_ids_;
  DeviceBuffer<CodewordPlan> d_plans_;
  DeviceBuffer<uint4> d_steps_;
  DeviceBuffer<uint32_t> d_grouped_;
  DeviceBuffer<uint2> d_decisions_;
  DeviceBuffer<EtiFrameMeta> d_meta_;
  DeviceBuffer<uint8_t> d_headers_, d_eti_, d_bytes_;
  int tf_slots_ = 0, msc_rows_ = 0;

  std::vector<JobList> stream_jobs_;
  MscWork work_, work_s3_;
  // session state (decode() resets it, feed() continues it)
  std::vector<ControlPlane> planes_;
  bool planes_fresh_ = false;      // a new session started: the planes are reset where they are first used
  PinnedBuffer<StreamState> h_states_;
  PinnedBuffer<int2> h_frames_;
  PinnedBuffer<const uint8_t*> h_ptrs_;
  PinnedBuffer<int64_t> h_nb_;
  PinnedBuffer<int> h_frame_slot_, h_frame_cif_row_;
  std::vector<int> carry_keep_, prev_used_, calls_done_, ord_done_, prev_tf_base_, prev_row_base_;
  DeviceBuffer<float> d_delta_;
  DeviceBuffer<uint4> d_guard_list_;
  DeviceBuffer<uint32_t> d_guard_counter_;
  PinnedBuffer<uint32_t> h_guard_counts_;
  int guard_launches_ = 0;
  bool guard_counters_clear_ = false;   // the layout kernel of this decode has cleared the device counters
  uint32_t guard_cap_ = 0, guard_cap_override_ = 0;
  std::vector<uint32_t> guard_caps_;   // per guarded launch of the decode: the list capacity it ran with
  int64_t guard_flagged_ = 0, guard_decisions_ = 0;
  int guard_overflows_ = 0;          // launches of the last decode whose list overflowed (decided again in full, fp64)
  DeviceBuffer<uint8_t> d_carry_;
  DeviceBuffer<CopyDesc> d_copy_descs_;
  HostList<CopyDesc> carry_out_descs_, carry_in_descs_;   // page-locked: they go up without the host waiting for the stream

  PlanTable plan_table_;

  PinnedBuffer<CallDesc> h_descs_;
  PinnedBuffer<int2> h_info_;
  PinnedBuffer<uint8_t> h_fibs_, h_fib_ok_;
  int max_calls_ = 0, nstreams_ = 0, last_ntf_ = 0;
  float scan_setup_ms_ = 0;
  std::vector<int64_t> eti_base_, eti_count_;
  std::vector<uint32_t> stream_status_;
  int64_t total_eti_ = 0;
  StageTimes times_;
  int64_t fft_launches_ = 0, fft_tfs_ = 0;
  double fft_ms_ = 0;
};

}  // namespace dabhip
