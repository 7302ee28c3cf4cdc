// capi.cpp — the C ABI of libdabhip.so (include/dabhip.h): the three reference seams
// (S1 viterbi, S2 sdr_demod, S3 dab_process_frame) as single-stream shims over the same
// HIP kernels the batch engine uses, plus the batch and stage entry points.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dabhip.h"
#include "engine.hpp"
#include "fifo_view.hpp"
#include "kernels.hpp"
#include "placement.hpp"

using namespace dabhip;

// The batch engine behind the C ABI: one or two lanes (each a complete Engine with its own HIP stream, buffers and
// host thread).  With two lanes the streams are split in halves; while one lane runs a GPU-saturating phase the other
// does its sync scan or its host-side control plane, so those no longer leave the GPU idle.
struct dabhip_engine {
  std::vector<std::unique_ptr<Engine>> lanes;
  std::mutex heavy;
  std::vector<int> lane_of, local_of;        // per stream of the last decode
  std::vector<int64_t> lane_frames;
  DeviceBuffer<uint8_t> combined;            // all ETI frames, stream-major, when more than one lane produced them
  float wall_ms = 0;
  int device = 0;

  explicit dabhip_engine(int dev, int host_threads = 0, const std::vector<int>& cpus = {}) : device(dev)
  {
    int nl = 1;   // measured on MI355X: splitting the batch costs more (two shorter Viterbi launches, two scans) than the overlap wins
    if (const char* env = std::getenv("DABHIP_LANES")) nl = std::max(1, std::min(4, std::atoi(env)));
    for (int l = 0; l < nl; ++l) {
      lanes.emplace_back(new Engine(dev, host_threads, cpus));
      if (!lanes.back()->ok()) break;
    }
    for (auto& l : lanes) l->set_heavy_lock(lanes.size() > 1 ? &heavy : nullptr);
  }
  bool ok() const { return !lanes.empty() && lanes.back()->ok(); }
  Engine& first() { return *lanes[0]; }

  int64_t decode(const uint8_t* const* iq, const size_t* nbytes, int nstreams, bool on_device)
  {
    const auto t0 = std::chrono::steady_clock::now();
    const int nl = nstreams >= 64 ? static_cast<int>(lanes.size()) : 1;   // small batches: the split does not pay
    lane_of.assign(nstreams, 0);
    local_of.assign(nstreams, 0);
    lane_frames.assign(lanes.size(), 0);
    std::vector<int> begin(nl + 1, 0);
    for (int l = 0; l < nl; ++l) begin[l + 1] = begin[l] + nstreams / nl + (l < nstreams % nl ? 1 : 0);
    for (int l = 0; l < nl; ++l)
      for (int b = begin[l]; b < begin[l + 1]; ++b) { lane_of[b] = l; local_of[b] = b - begin[l]; }
    std::vector<int64_t> result(nl, 0);
    std::vector<std::string> errors(nl);
    auto run = [&](int l) {
      result[l] = lanes[l]->decode(iq + begin[l], nbytes + begin[l], begin[l + 1] - begin[l], on_device);
      if (result[l] < 0) errors[l] = dabhip_last_error();
    };
    std::vector<std::thread> threads;
    for (int l = 1; l < nl; ++l) threads.emplace_back(run, l);
    run(0);
    for (auto& t : threads) t.join();
    int64_t total = 0;
    for (int l = 0; l < nl; ++l) {
      if (result[l] < 0) { set_error(errors[l]); return -1; }
      lane_frames[l] = result[l];
      total += result[l];
    }
    wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return total;
  }
};

namespace {

Engine* default_engine()
{
  static std::mutex mu;
  static std::unique_ptr<Engine> eng;
  std::lock_guard<std::mutex> lock(mu);
  if (!eng) {
    std::unique_ptr<Engine> e(new Engine(0));
    if (!e->ok()) return nullptr;
    eng = std::move(e);
  }
  return eng.get();
}

}  // namespace

extern "C" {

int dabhip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- batch engine -----------------------------------------------------------------------------
dabhip_engine* dabhip_engine_create(int device)
{
  dabhip_engine* e = new (std::nothrow) dabhip_engine(device);
  if (e && !e->ok()) { delete e; return nullptr; }
  return e;
}
dabhip_engine* dabhip_engine_create_ex(int device, int host_threads)
{
  dabhip_engine* e = new (std::nothrow) dabhip_engine(device, host_threads);
  if (e && !e->ok()) { delete e; return nullptr; }
  return e;
}
dabhip_engine* dabhip_engine_create_on_cpus(int device, int host_threads, const int32_t* cpus, int ncpus)
{
  std::vector<int> list;
  for (int i = 0; cpus && i < ncpus; ++i) list.push_back(cpus[i]);
  dabhip_engine* e = new (std::nothrow) dabhip_engine(device, host_threads, list);
  if (!e) return nullptr;
  if (!e->ok()) { delete e; return nullptr; }
  return e;
}
int dabhip_engine_host_cpus(const dabhip_engine* e, int32_t* cpus, int cap, int* numa_node)
{
  if (!e || e->lanes.empty()) return -1;
  const std::vector<int>& c = e->lanes[0]->host_cpus();
  if (numa_node) *numa_node = e->lanes[0]->numa_node();
  for (int i = 0; cpus && i < cap && i < static_cast<int>(c.size()); ++i) cpus[i] = c[static_cast<size_t>(i)];
  return static_cast<int>(c.size());
}
int dabhip_host_cpu_budget(int* affinity_cpus, int* cfs_quota_cpus)
{
  if (affinity_cpus) *affinity_cpus = static_cast<int>(dabhip::allowed_cpus().size());
  if (cfs_quota_cpus) *cfs_quota_cpus = dabhip::cfs_quota_cpus();
  return dabhip::usable_cpus();
}
int dabhip_host_placement_plan(const int32_t* slice_node, int nslices, const char* const* node_cpulist, int nnodes, int32_t* cpu_slice, int ncpu)
{
  if (!slice_node || !node_cpulist || !cpu_slice || nslices <= 0 || nnodes <= 0 || ncpu <= 0) { set_error("placement_plan: bad argument"); return -1; }
  std::vector<int> nodes(slice_node, slice_node + nslices);
  std::vector<std::vector<int>> node_cpus;
  for (int n = 0; n < nnodes; ++n) node_cpus.push_back(dabhip::parse_cpulist(node_cpulist[n] ? node_cpulist[n] : ""));
  const std::vector<std::vector<int>> plan = dabhip::plan_placement(nodes, node_cpus);
  for (int c = 0; c < ncpu; ++c) cpu_slice[c] = -1;
  int bound = 0;
  for (int i = 0; i < nslices; ++i) {
    bound += plan[static_cast<size_t>(i)].empty() ? 0 : 1;
    for (int c : plan[static_cast<size_t>(i)])
      if (c >= 0 && c < ncpu) cpu_slice[c] = i;       // (fewer CPUs than slices on a node: the later slice is the one recorded)
  }
  return bound;
}
void dabhip_engine_destroy(dabhip_engine* e) { delete e; }

int64_t dabhip_engine_decode(dabhip_engine* e, const uint8_t* const* iq, const size_t* nbytes, int nstreams, int on_device)
{
  if (!e || !iq || !nbytes) { set_error("engine_decode: null argument"); return -1; }
  if (nstreams <= 0) { set_error("engine_decode: no streams"); return -1; }
  return e->decode(iq, nbytes, nstreams, on_device != 0);
}
int64_t dabhip_engine_eti_count(const dabhip_engine* e, int stream)
{
  if (!e || stream < 0 || stream >= static_cast<int>(e->lane_of.size())) return -1;
  return e->lanes[e->lane_of[stream]]->eti_count(e->local_of[stream]);
}
uint32_t dabhip_engine_stream_status(const dabhip_engine* e, int stream)
{
  if (!e || stream < 0 || stream >= static_cast<int>(e->lane_of.size())) return 0xffffffffu;
  return e->lanes[e->lane_of[stream]]->stream_status(e->local_of[stream]);
}
int64_t dabhip_engine_eti_read(dabhip_engine* e, int stream, uint8_t* dst, int64_t cap_frames)
{
  if (!e || !dst) { set_error("eti_read: null argument"); return -1; }
  if (stream < 0 || stream >= static_cast<int>(e->lane_of.size())) { set_error("eti_read: bad stream"); return -1; }
  return e->lanes[e->lane_of[stream]]->eti_read(e->local_of[stream], dst, cap_frames);
}
// copy `text` into buf (NUL-terminated, cut at cap - 1 bytes); returns the length of the whole text
static int64_t hand_over_text(const std::string& text, char* buf, int64_t cap)
{
  if (buf && cap > 0) {
    const size_t n = std::min(text.size(), static_cast<size_t>(cap - 1));
    std::memcpy(buf, text.data(), n);
    buf[n] = 0;
  }
  return static_cast<int64_t>(text.size());
}
int64_t dabhip_engine_stream_log(dabhip_engine* e, int stream, char* buf, int64_t cap)
{
  if (!e || stream < 0 || stream >= static_cast<int>(e->lane_of.size())) return -1;
  return hand_over_text(e->lanes[e->lane_of[stream]]->take_stream_log(e->local_of[stream]), buf, cap);
}
int64_t dabhip_engine_eti_fetch(dabhip_engine* e, uint8_t* dst, int64_t cap_frames)
{
  if (!e || !dst) { set_error("eti_fetch: null argument"); return -1; }
  if (e->lanes.size() != 1) { set_error("eti_fetch: one lane only (DABHIP_LANES unset)"); return -1; }
  return e->first().eti_fetch_async(dst, cap_frames);
}
int dabhip_engine_eti_fetch_wait(dabhip_engine* e)
{
  if (!e) { set_error("eti_fetch_wait: null handle"); return -1; }
  return e->first().eti_fetch_wait() ? 0 : -1;
}
int64_t dabhip_engine_eti_drain(dabhip_engine* e, dabhip_eti_sink sink, void* user)
{
  if (!e || !sink) { set_error("eti_drain: null argument"); return -1; }
  int64_t total = 0;
  std::vector<uint8_t> buf;
  for (int b = 0; b < static_cast<int>(e->lane_of.size()); ++b) {
    const int64_t n = dabhip_engine_eti_count(e, b);
    if (n < 0) return -1;
    buf.resize(static_cast<size_t>(n) * DABHIP_ETI_BYTES);
    if (dabhip_engine_eti_read(e, b, buf.data(), n) != n) return -1;
    for (int64_t f = 0; f < n; ++f) sink(buf.data() + f * DABHIP_ETI_BYTES, b, user);
    total += n;
  }
  return total;
}
const void* dabhip_engine_eti_device_ptr(const dabhip_engine* ce, int64_t* nframes)
{
  if (!ce) return nullptr;
  dabhip_engine* e = const_cast<dabhip_engine*>(ce);
  int64_t total = 0, used = 0;
  for (size_t l = 0; l < e->lanes.size(); ++l) { total += e->lane_frames.empty() ? 0 : e->lane_frames[l]; used += (!e->lane_frames.empty() && e->lane_frames[l] > 0) ? 1 : 0; }
  if (nframes) *nframes = total;
  if (used <= 1) {
    for (size_t l = 0; l < e->lanes.size(); ++l)
      if (!e->lane_frames.empty() && e->lane_frames[l] > 0) return e->lanes[l]->eti_buffer();
    return e->lanes[0]->eti_buffer();
  }
  // several lanes produced frames: concatenate them (lane order = stream order)
  if (!e->combined.reserve(static_cast<size_t>(total) * DABHIP_ETI_BYTES)) return nullptr;
  size_t off = 0;
  for (size_t l = 0; l < e->lanes.size(); ++l) {
    const size_t bytes = static_cast<size_t>(e->lane_frames[l]) * DABHIP_ETI_BYTES;
    if (bytes && blocking_copy(e->combined.get() + off, e->lanes[l]->eti_buffer(), bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
      set_error("eti_device_ptr: concatenation failed");
      return nullptr;
    }
    off += bytes;
  }
  return e->combined.get();
}
int dabhip_engine_demapped_tf(dabhip_engine* e, int stream, int tf, int8_t* fic, int8_t* msc)
{
  if (!e || !fic || !msc) { set_error("demapped_tf: null argument"); return -1; }
  if (stream < 0 || stream >= static_cast<int>(e->lane_of.size())) { set_error("demapped_tf: bad stream"); return -1; }
  return e->lanes[e->lane_of[stream]]->read_demapped_tf(e->local_of[stream], tf, fic, msc) ? 0 : -1;
}
int dabhip_engine_trace(const dabhip_engine* e, int stream, int32_t* ints6, double* ffs, int cap_calls)
{
  if (!e || !ints6 || stream < 0 || stream >= static_cast<int>(e->lane_of.size())) return -1;
  return e->lanes[e->lane_of[stream]]->trace(e->local_of[stream], ints6, ffs, cap_calls);
}
int dabhip_engine_trace_nco(const dabhip_engine* e, int stream, int32_t* nco_hz, int cap_calls)
{
  if (!e || stream < 0 || stream >= static_cast<int>(e->lane_of.size())) return -1;
  return e->lanes[e->lane_of[stream]]->trace_nco(e->local_of[stream], nco_hz, cap_calls);
}
int dabhip_engine_stage_ms(const dabhip_engine* e, const char** names, float* ms, int cap)
{
  if (!e) return -1;
  constexpr int kN = 17;
  static const char* kNames[kN] = {"sync", "fft", "demap", "fic", "control", "gather", "viterbi", "eti", "host_setup", "host_frames", "host_worklist", "wall",
                                   "h2d", "h2d_mbytes", "h2d_pinned_mbytes", "sync_fp64_calls", "sync_spec_calls"};
  float v[kN] = {0};
  for (size_t l = 0; l < e->lanes.size(); ++l) {
    if (e->lane_frames.empty() || (l > 0 && e->lane_frames[l] == 0 && e->lane_of.size() < 64)) continue;
    const StageTimes& t = e->lanes[l]->stage_times();
    const float x[11] = {t.sync, t.fft, t.demap, t.fic, t.control, t.gather, t.viterbi, t.eti, t.setup, t.frames, t.worklist};
    for (int i = 0; i < 11; ++i) v[i] += x[i];     // lanes overlap in time: the sum is device/host work, not wall time
    v[12] += t.h2d;
    v[13] += static_cast<float>(t.h2d_bytes * 1e-6);
    v[14] += static_cast<float>(t.h2d_pinned_bytes * 1e-6);
    v[15] += t.sync_fp64_calls;
    v[16] += t.sync_spec_calls;
  }
  v[11] = e->wall_ms;
  int n = 0;
  for (; n < kN && n < cap; ++n) {
    if (names) names[n] = kNames[n];
    if (ms) ms[n] = v[n];
  }
  return n;
}
int dabhip_engine_set_afc(dabhip_engine* e, int enable)
{
  if (!e) return -1;
  for (auto& l : e->lanes) l->set_afc(enable != 0);
  return 0;
}
int dabhip_engine_set_soft(dabhip_engine* e, int enable)
{
  if (!e) return -1;
  for (auto& l : e->lanes) l->set_soft(enable != 0);
  return 0;
}
static uint64_t subchannel_mask(const int32_t* ids, int n)
{
  if (!ids || n <= 0) return ~0ull;
  uint64_t m = 0;
  for (int i = 0; i < n; ++i)
    if (ids[i] >= 0 && ids[i] < 64) m |= 1ull << ids[i];
  return m;
}
int dabhip_engine_set_subchannels(dabhip_engine* e, const int32_t* ids, int n)
{
  if (!e) return -1;
  for (auto& l : e->lanes) l->set_subchannel_filter(subchannel_mask(ids, n));
  return 0;
}
int dabhip_engine_set_parity_guard(dabhip_engine* e, int level)
{
  if (!e) return -1;
  for (auto& l : e->lanes) l->set_parity_guard(level);
  return 0;
}
int dabhip_engine_parity_guard_level(const dabhip_engine* e) { return e ? e->lanes[0]->parity_guard_level() : -1; }
int dabhip_parity_guard_default_level(void) { return dabhip::kDefaultGuardLevel; }
double dabhip_parity_guard_bin_scale(int raw_bin) { return (raw_bin >= 0 && raw_bin < 2048) ? static_cast<double>(dabhip::guard_bin_scale(raw_bin)) : -1.0; }
int dabhip_parity_guard_constants(int level, double* bin_c, double* prod_c)
{
  if (level < 1 || level > 2) return -1;
  if (bin_c) *bin_c = dabhip::guard_c_of(level);
  if (prod_c) *prod_c = dabhip::guard_prod_of(level);
  return 0;
}
int dabhip_engine_guard_stats(const dabhip_engine* e, int64_t* flagged, int64_t* decisions)
{
  if (!e) return -1;
  int64_t a = 0, b = 0;
  const int nl = e->lane_of.size() >= 64 ? static_cast<int>(e->lanes.size()) : 1;
  for (int l = 0; l < nl; ++l) {
    int64_t x = 0, y = 0;
    e->lanes[l]->guard_stats(&x, &y);
    a += x; b += y;
  }
  if (flagged) *flagged = a;
  if (decisions) *decisions = b;
  return 0;
}
int dabhip_engine_guard_overflows(const dabhip_engine* e)
{
  if (!e) return -1;
  int n = 0;
  const int nl = e->lane_of.size() >= 64 ? static_cast<int>(e->lanes.size()) : 1;
  for (int l = 0; l < nl; ++l) n += e->lanes[l]->guard_overflows();
  return n;
}
int dabhip_engine_set_guard_list_cap(dabhip_engine* e, uint32_t cap)
{
  if (!e) return -1;
  for (auto& l : e->lanes) l->set_guard_list_cap(cap);
  return 0;
}
int dabhip_engine_set_fused(dabhip_engine* e, int enable)
{
  if (!e) return -1;
  for (auto& l : e->lanes) l->set_fused(enable != 0);
  return 0;
}
int dabhip_engine_set_sync_speculation(dabhip_engine* e, int mode)
{
  if (!e) return -1;
  for (auto& l : e->lanes) l->set_sync_speculation(mode);
  return 0;
}
int dabhip_engine_fft_stats(const dabhip_engine* e, int64_t* launches, int64_t* tfs, double* ms)
{
  if (!e) return -1;
  int64_t a = 0, b = 0;
  double c = 0;
  const int nl = e->lane_of.size() >= 64 ? static_cast<int>(e->lanes.size()) : 1;
  for (int l = 0; l < nl; ++l) {
    int64_t x = 0, y = 0;
    double z = 0;
    e->lanes[l]->fft_stats(&x, &y, &z);
    a += x; b += y; c += z;
  }
  if (launches) *launches = a;
  if (tfs) *tfs = b;
  if (ms) *ms = c;
  return 0;
}

int dabhip_engine_fft_roofline(dabhip_engine* e, int reps, int64_t* launches, int64_t* tfs, double* ms)
{
  if (!e) { set_error("fft_roofline: null handle"); return -1; }
  int64_t a = 0, b = 0;
  double c = 0;
  const int nl = e->lane_of.size() >= 64 ? static_cast<int>(e->lanes.size()) : 1;
  for (int l = 0; l < nl; ++l) {
    int64_t x = 0, y = 0;
    double z = 0;
    if (e->lanes[l]->fft_roofline(reps, &x, &y, &z) != 0) return -1;
    a += x; b += y; c += z;
  }
  if (launches) *launches = a;
  if (tfs) *tfs = b;
  if (ms) *ms = c;
  return 0;
}

// ---- stage entries ----------------------------------------------------------------------------
int dabhip_stage_ofdm_fft(dabhip_engine* e, const uint8_t* frames, int nframes, float* spectra, int on_device, int reps, float* kernel_ms)
{
  if (!e || !frames) { set_error("stage_ofdm_fft: null argument"); return -1; }
  return e->first().stage_ofdm_fft(frames, nframes, spectra, on_device != 0, reps, kernel_ms);
}
int dabhip_stage_demap(dabhip_engine* e, const float* spectra, int nframes, uint8_t* fic, uint8_t* msc)
{
  if (!e || !spectra || !fic || !msc) { set_error("stage_demap: null argument"); return -1; }
  return e->first().stage_demap(spectra, nframes, fic, msc);
}
int dabhip_stage_decision_audit(dabhip_engine* e, const uint8_t* frames, int nframes, int on_device, int guard_on, double* out8)
{
  if (!e || !frames || !out8) { set_error("stage_decision_audit: null argument"); return -1; }
  return e->first().stage_decision_audit(frames, nframes, on_device != 0, guard_on != 0, out8);
}
int dabhip_stage_decision_audit_fused(dabhip_engine* e, const uint8_t* frames, int nframes, int on_device, int guard_on, double* out10)
{
  if (!e || !frames || !out10) { set_error("stage_decision_audit_fused: null argument"); return -1; }
  return e->first().stage_decision_audit(frames, nframes, on_device != 0, guard_on != 0, out10, true, out10 + 8);
}
int dabhip_stage_fic_decode(dabhip_engine* e, const uint8_t* fic, int nframes, uint8_t* fibs, uint8_t* crc_ok)
{
  if (!e || !fic || !fibs || !crc_ok) { set_error("stage_fic_decode: null argument"); return -1; }
  return e->first().stage_fic_decode(fic, nframes, fibs, crc_ok);
}

// ---- S1: decoder seam ---------------------------------------------------------------------------
void* dabhip_create_viterbi(int /*len*/) { return default_engine(); }
int dabhip_init_viterbi(void) { return default_engine() ? 0 : -1; }

int dabhip_viterbi_batch(void* p, const unsigned char* symbols, unsigned char* data, int framebits, int n)
{
  Engine* eng = p ? static_cast<Engine*>(p) : default_engine();
  if (!eng) return -1;
  if (!symbols || !data) { set_error("viterbi: null argument"); return -1; }
  return eng->viterbi_batch(symbols, data, framebits, n);
}
void dabhip_viterbi(void* p, unsigned char* symbols, unsigned char* data, int framebits)
{
  (void)dabhip_viterbi_batch(p, symbols, data, framebits, 1);   // like the reference: no error return
}

}  // extern "C"

// ---- S2: front-end seam ---------------------------------------------------------------------------
struct dabhip_sdr {
  Engine eng;
  DeviceBuffer<uint8_t> window;      // the most recent IQ bytes, device resident
  DeviceBuffer<StreamState> state;
  DeviceBuffer<uint8_t> tail;        // the last kTailBytes of sdr->buffer (device_types.hpp)
  int64_t base = 0;                  // stream offset of window[0]
  int64_t fed = 0;                   // bytes received so far
  int call = 0;
  CallDesc last{};
  explicit dabhip_sdr(int device) : eng(device) {}
};

namespace {
constexpr int64_t kWindowBytes = int64_t(48) << 20;
constexpr int64_t kKeepBytes = int64_t(16) << 20;
}  // namespace

extern "C" {

dabhip_sdr* dabhip_sdr_init(int device)
{
  dabhip_sdr* s = new (std::nothrow) dabhip_sdr(device);
  if (!s) return nullptr;
  if (!s->eng.ok() || !s->window.reserve(kWindowBytes) || !s->state.reserve(1) || !s->tail.reserve(kTailBytes)) { delete s; return nullptr; }
  StreamState st;
  std::memset(&st, 0, sizeof st);
  fifo_reset(st);
  if (blocking_copy(s->state.get(), &st, sizeof st, hipMemcpyHostToDevice) != hipSuccess || hipMemset(s->tail.get(), 0, kTailBytes) != hipSuccess) {
    set_error("sdr_init: state upload failed");
    delete s;
    return nullptr;
  }
  std::memset(&s->last, 0, sizeof s->last);
  return s;
}
void dabhip_sdr_free(dabhip_sdr* s) { delete s; }

int dabhip_sdr_demod(dabhip_sdr* s, const uint8_t* input_buffer, int input_buffer_len, uint8_t* fic, uint8_t* msc)
{
  if (!s || (!input_buffer && input_buffer_len != 0) || !fic || !msc) { set_error("sdr_demod: null argument"); return -1; }
  // sdr_demod appends whatever the callback left, input_buffer_len bytes (input_sdr.c:36-38; librtlsdr delivers DEFAULT_BUF_LENGTH = 262144,
  // dab2eti.c:125-126,238, a file's last buffer is shorter).  Whole I/Q pairs only: the kernels read the stream two bytes at a time.
  if (input_buffer_len < 0 || input_buffer_len > kChunkBytes || (input_buffer_len & 1)) {
    set_error("sdr_demod: input_buffer_len must be an even number of bytes, 0 .. 262144 (sizeof sdr->input_buffer, input_sdr.h:14)");
    return -1;
  }
  if (s->fed - s->base + input_buffer_len > kWindowBytes) {   // slide the device window
    const int64_t keep_from = s->fed - kKeepBytes;
    DeviceBuffer<uint8_t> tmp;
    if (!tmp.reserve(kKeepBytes)) return -1;
    if (blocking_copy(tmp.get(), s->window.get() + (keep_from - s->base), kKeepBytes, hipMemcpyDeviceToDevice) != hipSuccess ||
        blocking_copy(s->window.get(), tmp.get(), kKeepBytes, hipMemcpyDeviceToDevice) != hipSuccess) {
      set_error("sdr_demod: window slide failed");
      return -1;
    }
    s->base = keep_from;
  }
  if (input_buffer_len && blocking_copy(s->window.get() + (s->fed - s->base), input_buffer, input_buffer_len, hipMemcpyHostToDevice) != hipSuccess) {
    set_error("sdr_demod: IQ upload failed");
    return -1;
  }
  s->fed += input_buffer_len;
  const uint8_t* virtual_base = s->window.get() - s->base;    // stream offset x lives at virtual_base + x
  if (!s->eng.scan_one_call(virtual_base, s->state.get(), s->tail.get(), s->call, input_buffer_len, &s->last)) return -1;
  ++s->call;
  if (s->last.status != 2) return 0;
  for (int i = 0; i < s->last.view.nseg; ++i)
    if (s->last.view.seg_src[i] >= 0 && s->last.view.seg_src[i] < s->base) { set_error("sdr_demod: stale frame tail older than the device window"); return -1; }
  return s->eng.demod_one_frame(virtual_base, s->last, fic, msc) ? 1 : -1;
}

int32_t dabhip_sdr_coarse_timeshift(const dabhip_sdr* s) { return s ? s->last.coarse_timeshift : 0; }
int32_t dabhip_sdr_fine_timeshift(const dabhip_sdr* s) { return s ? s->last.fine_timeshift : 0; }
int32_t dabhip_sdr_coarse_freq_shift(const dabhip_sdr* s) { return s ? s->last.coarse_freq_shift : 0; }
double dabhip_sdr_fine_freq_shift(const dabhip_sdr* s) { return s ? s->last.fine_freq_shift : 0.0; }

}  // extern "C"

// ---- S3: back-end seam ----------------------------------------------------------------------------
struct dabhip_dab {
  Engine eng;
  ControlPlane plane;
  dabhip_eti_callback cb = nullptr;
  std::vector<uint8_t> fic, msc, fibs, ok, eti;
  int slot = 0;                      // TF slot the next frame goes to
  int ordinal = 0;                   // == slot + dropped
  int dropped = 0;                   // TF slots discarded from the front so far
  explicit dabhip_dab(int device) : eng(device), fic(kFicBits), msc(kMscBits), fibs(384), ok(12), eti(4 * kEtiBytes) {}
};

namespace {
constexpr int kDabSlots = 64;
}

extern "C" {

dabhip_dab* dabhip_dab_init(int device, dabhip_eti_callback cb)
{
  dabhip_dab* d = new (std::nothrow) dabhip_dab(device);
  if (!d) return nullptr;
  if (!d->eng.ok() || !d->eng.reserve_tf_slots(kDabSlots)) { delete d; return nullptr; }
  d->cb = cb;
  return d;
}
void dabhip_dab_free(dabhip_dab* d) { delete d; }
uint8_t* dabhip_dab_tf_fic(dabhip_dab* d) { return d ? d->fic.data() : nullptr; }
uint8_t* dabhip_dab_tf_msc(dabhip_dab* d) { return d ? d->msc.data() : nullptr; }
int dabhip_dab_locked(const dabhip_dab* d) { return d && d->plane.locked(); }
int64_t dabhip_dab_take_log(dabhip_dab* d, char* buf, int64_t cap) { return d ? hand_over_text(d->plane.take_log(), buf, cap) : -1; }
uint32_t dabhip_dab_status(const dabhip_dab* d) { return d ? d->plane.fault() : 0xffffffffu; }
int dabhip_dab_set_soft(dabhip_dab* d, int enable)
{
  if (!d) { set_error("dab_set_soft: null handle"); return -1; }
  if (d->slot != 0 || d->dropped != 0) { set_error("dab_set_soft: only before the first frame"); return -1; }
  d->eng.set_soft(enable != 0);
  return d->eng.reserve_tf_slots(kDabSlots) ? 0 : -1;
}
int dabhip_dab_last_fibs(const dabhip_dab* d, uint8_t* fibs, uint8_t* crc_ok)
{
  if (!d || !fibs || !crc_ok) return -1;
  std::memcpy(fibs, d->fibs.data(), 384);
  std::memcpy(crc_ok, d->ok.data(), 12);
  return 0;
}

int dabhip_dab_process_frame(dabhip_dab* d)
{
  if (!d) { set_error("dab_process_frame: null handle"); return -1; }
  if (d->slot == kDabSlots) {        // keep the 4 most recent TFs (16 CIFs of interleaver history)
    if (!d->eng.recycle_tf_slots(kDabSlots, 4)) return -1;
    d->plane.rebase(4 * (kDabSlots - 4));
    d->dropped += kDabSlots - 4;
    d->slot = 4;
  }
  if (!d->eng.store_tf_bytes(d->slot, d->fic.data(), d->msc.data())) return -1;
  if (!d->eng.fic_decode_slots(d->slot, 1, d->fibs.data(), d->ok.data())) return -1;
  JobList jobs;
  d->plane.on_tf(d->slot, d->fibs.data(), d->ok.data(), jobs);
  ++d->slot;
  if (jobs.empty()) return 0;
  std::vector<int> row_base = {15}, fib_base = {0};   // single stream: CIF 0 at logical row 15 (Engine::store_tf_bytes)
  std::vector<const ControlPlane*> planes = {&d->plane};
  std::vector<const JobList*> job_lists = {&jobs};
  if (!d->eng.msc_decode(job_lists, planes, row_base, fib_base)) return -1;
  if (!d->eng.read_eti(0, static_cast<int64_t>(jobs.size()), d->eti.data())) return -1;
  if (d->cb)
    for (size_t f = 0; f < jobs.size(); ++f) d->cb(d->eti.data() + f * kEtiBytes);
  return static_cast<int>(jobs.size());
}

}  // extern "C"

// ---- host-side control plane without a GPU -----------------------------------------------------------
namespace {
void sub_to_row(const SubChannel& s, int32_t* o)
{
  o[0] = s.id; o[1] = s.slform; o[2] = s.uep_index; o[3] = s.start_cu;
  o[4] = s.size_cu; o[5] = s.bitrate; o[6] = s.protlev; o[7] = s.ascty;
}
}  // namespace

extern "C" {

int dabhip_host_parse_fibs(const uint8_t* fibs, const uint8_t* crc_ok, int32_t* hdr3, int32_t* sub)
{
  if (!fibs || !crc_ok || !hdr3 || !sub) { set_error("host_parse_fibs: null argument"); return -1; }
  EnsembleInfo info;
  decode_fibs(info, fibs, crc_ok);
  hdr3[0] = info.eid; hdr3[1] = info.cif_hi; hdr3[2] = info.cif_lo;
  for (int i = 0; i < 64; ++i) sub_to_row(info.sub[i], sub + 8 * i);
  return 0;
}

int dabhip_host_eti_header(const int32_t* hdr3, const int32_t* sub, uint8_t* out, int cap)
{
  if (!hdr3 || !sub || !out || cap < kEtiHeaderMax) { set_error("host_eti_header: bad argument"); return -1; }
  EnsembleInfo info;
  info.eid = static_cast<uint16_t>(hdr3[0]);
  info.cif_hi = static_cast<uint8_t>(hdr3[1]);
  info.cif_lo = static_cast<uint8_t>(hdr3[2]);
  for (int i = 0; i < 64; ++i) {
    const int32_t* r = sub + 8 * i;
    SubChannel& s = info.sub[i];
    s.id = r[0]; s.slform = r[1]; s.uep_index = r[2]; s.start_cu = r[3];
    s.size_cu = r[4]; s.bitrate = r[5]; s.protlev = r[6]; s.ascty = r[7];
  }
  info.rescan();
  return build_eti_header(out, info);
}

// the operator messages (ControlPlane::take_log) of the calling thread's last dabhip_host_control_replay
static thread_local std::string g_replay_log;
int64_t dabhip_host_control_replay_log(char* buf, int64_t cap) { return hand_over_text(g_replay_log, buf, cap); }
int dabhip_host_control_replay(const uint8_t* fibs, const uint8_t* crc_ok, int ntf, int32_t* first_cif, uint8_t* headers,
                               int32_t* header_len, int cap_frames)
{
  if (!fibs || !crc_ok || !first_cif || !headers || !header_len) { set_error("host_control_replay: null argument"); return -1; }
  ControlPlane plane;
  JobList jobs;
  for (int t = 0; t < ntf; ++t) plane.on_tf(t, fibs + static_cast<size_t>(t) * 384, crc_ok + static_cast<size_t>(t) * 12, jobs);
  g_replay_log = plane.take_log();
  const int n = static_cast<int>(jobs.size());
  for (int i = 0; i < n && i < cap_frames; ++i) {
    first_cif[i] = jobs[i].first_cif;
    header_len[i] = jobs[i].header_len;
    std::memset(headers + static_cast<size_t>(i) * kEtiHeaderMax, 0, kEtiHeaderMax);
    std::memcpy(headers + static_cast<size_t>(i) * kEtiHeaderMax, jobs.header(jobs[i]), static_cast<size_t>(jobs[i].header_len));
  }
  return n;
}

}  // extern "C"

// ---- the product's constant tables (dab_tables.hpp), for the CPU test-suite to compare with the reference's arrays ----
extern "C" int dabhip_host_table(int which, int32_t* out, int cap)
{
  if (!out) { set_error("host_table: null argument"); return -1; }
  int n = 0;
  auto put = [&](int v) { if (n < cap) out[n] = v; ++n; };
  switch (which) {
    case 0:                                  // 64 rows {bitrate, size_cu, protlevel, L1..L4, PI1..PI4} (PI as in ETSI: 1..24, 0 = unused)
      for (int i = 0; i < 64; ++i) {
        const UepProfile& u = uep_table()[i];
        put(u.bitrate); put(u.size_cu); put(u.protlevel);
        for (int k = 0; k < 4; ++k) put(u.l[k]);
        for (int k = 0; k < 4; ++k) put(u.pi[k]);
      }
      break;
    case 1:                                  // puncturing vectors PI = 1..24 as 32 flags each
      for (int pi = 1; pi <= 24; ++pi)
        for (int b = 0; b < 32; ++b) put(static_cast<int>((puncture_mask(pi) >> b) & 1u));
      break;
    case 2:                                  // frequency de-interleaver: carrier -> QPSK symbol index
      for (uint16_t v : carrier_to_qpsk()) put(v);
      break;
    case 3:                                  // phase reference symbol, quarter turns per carrier
      for (uint8_t v : prs_quarter_turns()) put(v);
      break;
    default:
      set_error("host_table: unknown table");
      return -1;
  }
  if (n > cap) { set_error("host_table: buffer too small"); return -1; }
  return n;
}

// ---- FIFO / frame-buffer bookkeeping of K1 on the host (fifo_view.hpp), callable without a GPU ---------
struct dabhip_fifo {
  StreamState st;
  uint8_t tail[kTailBytes];          // the last kTailBytes of sdr->buffer, kept as bytes (device_types.hpp)
};
extern "C" dabhip_fifo* dabhip_host_fifo_new(void)
{
  dabhip_fifo* f = new (std::nothrow) dabhip_fifo;
  if (!f) return nullptr;
  std::memset(&f->st, 0, sizeof f->st);
  std::memset(f->tail, 0, sizeof f->tail);
  fifo_reset(f->st);
  return f;
}
extern "C" void dabhip_host_fifo_free(dabhip_fifo* f) { delete f; }
extern "C" int dabhip_host_fifo_call(dabhip_fifo* f, int32_t coarse_timeshift, int32_t fine_timeshift, int32_t chunk_bytes, const uint8_t* stream,
                                     int32_t* nseg, int32_t* seg_end, int64_t* seg_src, int32_t* fifo_count, uint8_t* tail)
{
  if (!f || !nseg || !seg_end || !seg_src) { set_error("host_fifo_call: null argument"); return -1; }
  if (chunk_bytes < 0 || chunk_bytes > kChunkBytes || (chunk_bytes & 1)) { set_error("host_fifo_call: chunk_bytes must be even, 0 .. 262144"); return -1; }
  f->st.coarse_timeshift = coarse_timeshift;
  f->st.fine_timeshift = fine_timeshift;
  const FifoCall c = fifo_call(f->st, chunk_bytes);
  if (f->st.overflow) { set_error("host_fifo_call: more than kMaxSeg nested short reads"); return -1; }
  if (c.status && stream)                                  // the rule K1 applies to its registers (sync_scan_kernel), byte by byte
    for (int p = kTailStart; p < kTfBytes; ++p) {
      const int64_t src = read_source(f->st.view, c.fresh, p);
      if (src >= 0) f->tail[p - kTailStart] = stream[src];
    }
  *nseg = f->st.view.nseg;
  for (int i = 0; i < kMaxSeg; ++i) { seg_end[i] = f->st.view.seg_end[i]; seg_src[i] = f->st.view.seg_src[i]; }
  if (fifo_count) *fifo_count = c.fifo_count;
  if (tail) std::memcpy(tail, f->tail, kTailBytes);
  return c.status ? (c.do_sync ? 2 : 1) : 0;
}

extern "C" int dabhip_host_fifo_skip_unshifted(dabhip_fifo* f, int32_t ncalls, int64_t* fed, int64_t* consumed)
{
  if (!f || ncalls < 0) { set_error("host_fifo_skip_unshifted: bad argument"); return -1; }
  if (f->st.coarse_timeshift + f->st.fine_timeshift != 0 || f->st.startup_delay <= 0) { set_error("host_fifo_skip_unshifted: a shift is pending or the first frame is still to be dropped"); return -1; }
  fifo_skip_unshifted(f->st.fed, f->st.consumed, ncalls);
  if (fed) *fed = f->st.fed;
  if (consumed) *consumed = f->st.consumed;
  return 0;
}

// ---- device-side modulator (k_synth.hip) ------------------------------------------------------
namespace dabhip { int synth_generate_device(const dabhip_synth_cfg* cfgs, int nstreams, int ntf, uint8_t* const* iq, int device); }
extern "C" int dabhip_synth_generate_device(const dabhip_synth_cfg* cfgs, int nstreams, int ntf, uint8_t* const* iq, int device)
{
  return dabhip::synth_generate_device(cfgs, nstreams, ntf, iq, device);
}

// ---- streaming sessions -----------------------------------------------
// B parallel unbounded streams decoded segment by segment.  Per stream the session keeps device windows: segment k lives in
// window k % 3 behind a reserve of kWindowReserve bytes, and the bytes of earlier segments the front end may still read (FIFO
// backlog and stale-tail sources, Engine::stream_need_from) are copied in front of it from window (k - 1) % 3 when segment k is
// fed.  Three windows so that the NEXT segment (k + 1) can be uploading into its window -- which holds segment k - 2, dead since
// feed(k - 1) -- on a stream of its own while segment k decodes (dabhip_stream_prefetch).
namespace {
// room in front of every segment for the bytes of earlier segments K1 may still read: > FIFO capacity (1.5 MiB) + 12 nested stale tails of
// one TF each.  (DABHIP_WINDOW_RESERVE=bytes: test knob -- a small reserve sends every feed through the "more history than the reserve
// holds" path.)
const size_t kWindowReserve = [] {
  const char* env = std::getenv("DABHIP_WINDOW_RESERVE");
  const size_t v = env ? static_cast<size_t>(std::strtoull(env, nullptr, 10)) & ~size_t(255) : 0;
  return v ? v : size_t(8) << 20;
}();
}
struct dabhip_stream {
  Engine eng;
  int n = 0;
  bool first = true;
  std::vector<std::unique_ptr<DeviceBuffer<uint8_t>>> win[3];
  std::vector<int64_t> base, avail;            // per stream: first stream byte still held, bytes received (fed) so far
  std::vector<size_t> org;                     // per stream: offset, in the newest fed window, of stream byte base[b]
  uint64_t fed = 0, queued = 0;                // segments fed / handed over (fed <= queued <= fed + 2)
  bool queued_ever = false;                    // dabhip_stream_prefetch has been used: its stream has work to forget (reap_stream)
  uint32_t up_uses = 0;
  // A feed that fails after it has started to move the session on (windows, offsets, the engine's carried state) leaves a session nobody can
  // re-feed correctly: it is marked and refuses everything but its destruction -- an honest error instead of frames decoded at the wrong offsets.
  bool failed = false;
  bool resident = false;                       // fed through dabhip_stream_feed_resident: the caller's buffers are read in place, no windows
  // prefetch uploads run on a stream of their own.  Measured on the 256-stream workload, 8-TF segments (805 MB each): one gather kernel
  // per segment 56.5 GB/s, 256 copy commands on one stream 54.0, dealt to two / four streams 25 / 36 (they get in each other's way)
  static constexpr int kUpStreams = 1;
  hipStream_t up_stream[kUpStreams] = {nullptr};
  hipEvent_t up_done[3][kUpStreams] = {};
  // upload by a gather kernel that reads the page-locked host segments over PCIe (one launch per segment instead of one copy command
  // per stream): descriptor lists, one per window, page-locked so that they go up asynchronously
  HostList<CopyDesc> gather_descs[3];
  DeviceBuffer<CopyDesc> d_gather_descs[3];
  struct Pending { std::vector<const uint8_t*> iq; std::vector<size_t> nbytes; };
  Pending pending[3];                          // what was prefetched into window i (checked against the feed that consumes it)
  dabhip_stream(int device, int nstreams, int host_threads = 0, std::vector<int> cpus = {})
      : eng(device, host_threads, std::move(cpus)), n(nstreams), base(nstreams, 0), avail(nstreams, 0), org(nstreams, 0)
  {
    for (int s = 0; s < 3; ++s)
      for (int b = 0; b < nstreams; ++b) win[s].emplace_back(new DeviceBuffer<uint8_t>());
    if (eng.ok()) {
      for (auto& st : up_stream) (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
      for (auto& w : up_done)
        for (auto& e : w) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    }
  }
  ~dabhip_stream()
  {
    for (auto& st : up_stream)
      if (st) (void)hipStreamSynchronize(st);
    for (auto& w : up_done)
      for (auto& e : w)
        if (e) (void)hipEventDestroy(e);
    for (auto& st : up_stream)
      if (st) (void)hipStreamDestroy(st);
  }
  bool streams_ok() const
  {
    for (auto st : up_stream)
      if (!st) return false;
    return true;
  }
  // the same by ONE kernel launch on the upload stream, when every non-empty host segment is page-locked (device-visible): a small
  // persistent grid reads the host memory over PCIe (64 workgroups: 56.5 GB/s; 16: 54.5; 256: 43.8 -- and they would take CUs from the
  // decode running beside it).  DABHIP_PREFETCH_KERNEL=0 selects the copy engine instead, = N > 1 another grid size.
  bool upload_by_kernel(int w, const uint8_t* const* iq, const size_t* nbytes)
  {
    static const int mode = std::getenv("DABHIP_PREFETCH_KERNEL") ? std::atoi(std::getenv("DABHIP_PREFETCH_KERNEL")) : 1;
    if (mode <= 0) return false;
    HostList<CopyDesc>& descs = gather_descs[w];
    descs.clear();
    for (int b = 0; b < n; ++b) {
      if (nbytes[b] == 0) continue;
      if (nbytes[b] >= (size_t(1) << 32)) return false;
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, iq[b]) != hipSuccess || attr.type != hipMemoryTypeHost || !attr.devicePointer) { (void)hipGetLastError(); return false; }
      DeviceBuffer<uint8_t>& to = *win[w][b];
      if (!to.reserve(kWindowReserve + std::max<size_t>(nbytes[b], 16))) return false;
      const uint8_t* dev_view = static_cast<const uint8_t*>(attr.devicePointer) + (iq[b] - static_cast<const uint8_t*>(attr.hostPointer));
      descs.push_back(CopyDesc{dev_view, to.get() + kWindowReserve, static_cast<uint32_t>(nbytes[b]), 0});
    }
    if (descs.empty()) return true;
    const int wgs = mode > 1 ? mode : 64;
    return d_gather_descs[w].upload(descs, up_stream[0]) &&
           launch_host_gather(d_gather_descs[w].get(), static_cast<int>(descs.size()), wgs, up_stream[0]) == hipSuccess;
  }
  // segment -> window w of every stream, behind the reserve; on stream `one`, or dealt round-robin to the upload streams
  // segments that already are in device memory: one launch copies them all (256 copy commands cost 2.9 ms back to back and as much on the host)
  bool copy_by_kernel(int w, const uint8_t* const* iq, const size_t* nbytes, hipStream_t st)
  {
    HostList<CopyDesc>& descs = gather_descs[w];
    descs.clear();
    uint32_t longest = 0;
    for (int b = 0; b < n; ++b) {
      DeviceBuffer<uint8_t>& to = *win[w][b];
      if (nbytes[b] >= (size_t(1) << 32)) return false;
      if (!to.reserve(kWindowReserve + std::max<size_t>(nbytes[b], 16))) return false;
      if (nbytes[b] == 0) continue;
      descs.push_back(CopyDesc{iq[b], to.get() + kWindowReserve, static_cast<uint32_t>(nbytes[b]), 0});
      longest = std::max(longest, static_cast<uint32_t>(nbytes[b]));
    }
    if (descs.empty()) return true;
    return d_gather_descs[w].upload(descs, st) && launch_device_gather(d_gather_descs[w].get(), static_cast<int>(descs.size()), longest, st) == hipSuccess;
  }
  HostList<CopyDesc> history_descs;            // the bytes of earlier segments moved in front of the segment being fed (dabhip_stream_feed)
  DeviceBuffer<CopyDesc> d_history_descs;
  bool upload(int w, const uint8_t* const* iq, const size_t* nbytes, bool on_device, hipStream_t one)
  {
    if (!one && !on_device && upload_by_kernel(w, iq, nbytes)) return true;
    if (on_device && copy_by_kernel(w, iq, nbytes, one ? one : up_stream[0])) return true;
    for (int b = 0; b < n; ++b) {
      DeviceBuffer<uint8_t>& to = *win[w][b];
      hipStream_t st = one ? one : up_stream[b % kUpStreams];
      if (!to.reserve(kWindowReserve + std::max<size_t>(nbytes[b], 16))) return false;
      if (nbytes[b] && hipMemcpyAsync(to.get() + kWindowReserve, iq[b], nbytes[b], on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st) != hipSuccess) {
        set_error("stream_feed: segment upload failed");
        return false;
      }
    }
    return true;
  }
};

extern "C" dabhip_stream* dabhip_stream_create(int device, int nstreams)
{
  if (nstreams <= 0) { set_error("stream_create: no streams"); return nullptr; }
  dabhip_stream* s = new dabhip_stream(device, nstreams);
  if (!s->eng.ok() || !s->streams_ok()) { delete s; return nullptr; }
  return s;
}
// the same with the host side chosen by the caller (dabhip_multi_stream_create: one session per device, each on the CPUs of its device's NUMA node)
extern "C" dabhip_stream* dabhip_stream_create_on_cpus(int device, int nstreams, int host_threads, const int32_t* cpus, int ncpus)
{
  if (nstreams <= 0) { set_error("stream_create: no streams"); return nullptr; }
  std::vector<int> list;
  for (int i = 0; cpus && i < ncpus; ++i) list.push_back(cpus[i]);
  dabhip_stream* s = new dabhip_stream(device, nstreams, host_threads, list);
  if (!s->eng.ok() || !s->streams_ok()) { delete s; return nullptr; }
  return s;
}
extern "C" void dabhip_stream_destroy(dabhip_stream* s) { delete s; }
extern "C" int dabhip_stream_set_subchannels(dabhip_stream* s, const int32_t* ids, int n)
{
  if (!s) return -1;
  if (!s->first) { set_error("stream_set_subchannels: only before the first segment"); return -1; }
  s->eng.set_subchannel_filter(subchannel_mask(ids, n));
  return 0;
}
extern "C" int dabhip_stream_set_afc(dabhip_stream* s, int on) { if (!s) return -1; s->eng.set_afc(on != 0); return 0; }
extern "C" int dabhip_stream_set_parity_guard(dabhip_stream* s, int on) { if (!s) return -1; s->eng.set_parity_guard(on); return 0; }
extern "C" int dabhip_stream_set_sync_speculation(dabhip_stream* s, int mode) { if (!s) return -1; s->eng.set_sync_speculation(mode); return 0; }
extern "C" int dabhip_stream_set_soft(dabhip_stream* s, int on)
{
  if (!s) return -1;
  if (!s->first) { set_error("stream_set_soft: only before the first segment"); return -1; }
  s->eng.set_soft(on != 0);
  return 0;
}

// Start uploading a segment that a LATER dabhip_stream_feed will consume, and return at once.  Host segments must live in
// page-locked memory (dabhip_host_alloc) for the copy to be a true asynchronous DMA; they must stay untouched until the feed
// that consumes them has returned.
extern "C" int dabhip_stream_prefetch(dabhip_stream* s, const uint8_t* const* iq, const size_t* nbytes, int on_device)
{
  if (!s || !iq || !nbytes) { set_error("stream_prefetch: null argument"); return -1; }
  if (s->failed) { set_error("stream_prefetch: an earlier feed of this session failed half-way -- destroy the session"); return -1; }
  if (s->resident) { set_error("stream_prefetch: this session is fed through dabhip_stream_feed_resident"); return -1; }
  if (s->queued - s->fed >= 2) { set_error("stream_prefetch: two segments are already waiting to be fed"); return -1; }
  if (hipSetDevice(s->eng.device()) != hipSuccess) { set_error("stream_prefetch: hipSetDevice failed"); return -1; }
  const int w = static_cast<int>(s->queued % 3);
  if (!s->upload(w, iq, nbytes, on_device != 0, nullptr)) return -1;
  for (int i = 0; i < dabhip_stream::kUpStreams; ++i)
    if (hipEventRecord(s->up_done[w][i], s->up_stream[i]) != hipSuccess) { set_error("stream_prefetch: event record failed"); return -1; }
  s->pending[w].iq.assign(iq, iq + s->n);
  s->pending[w].nbytes.assign(nbytes, nbytes + s->n);
  ++s->queued;
  s->queued_ever = true;
  return 0;
}

extern "C" int64_t dabhip_stream_feed(dabhip_stream* s, const uint8_t* const* iq, const size_t* nbytes, int on_device)
{
  if (!s || !iq || !nbytes) { set_error("stream_feed: null argument"); return -1; }
  if (s->failed) { set_error("stream_feed: an earlier feed of this session failed half-way; its state is not trustworthy any more -- destroy the session"); return -1; }
  if (s->resident) { set_error("stream_feed: this session is fed through dabhip_stream_feed_resident"); return -1; }
  if (hipSetDevice(s->eng.device()) != hipSuccess) { set_error("stream_feed: hipSetDevice failed"); return -1; }
  auto broken = [s](const char* msg) -> int64_t {          // from here on an error leaves the session's books half-updated
    s->failed = true;
    if (msg) set_error(msg);
    return -1;
  };
  const int w = static_cast<int>(s->fed % 3), wprev = static_cast<int>((s->fed + 2) % 3);
  hipStream_t st = s->eng.stream();
  if (s->queued > s->fed) {                    // this segment was prefetched: it must be the one handed over first
    const dabhip_stream::Pending& p = s->pending[w];
    for (int b = 0; b < s->n; ++b)
      if (p.iq[b] != iq[b] || p.nbytes[b] != nbytes[b]) { set_error("stream_feed: not the segment that was prefetched first"); return -1; }
    for (int i = 0; i < dabhip_stream::kUpStreams; ++i)
      if (hipStreamWaitEvent(st, s->up_done[w][i], 0) != hipSuccess) { set_error("stream_feed: event wait failed"); return -1; }
  } else {
    if (!s->upload(w, iq, nbytes, on_device != 0, st)) return broken(nullptr);
    ++s->queued;
  }
  std::vector<const uint8_t*> virt(s->n);
  std::vector<size_t> avail(s->n);
  HostList<CopyDesc>& moves = s->history_descs;
  moves.clear();
  uint32_t longest_move = 0;
  for (int b = 0; b < s->n; ++b) {
    const int64_t need = s->first ? 0 : std::min(s->eng.stream_need_from(b), s->avail[b]);
    const size_t kept = static_cast<size_t>(s->avail[b] - need);
    DeviceBuffer<uint8_t>& from = *s->win[wprev][b];
    DeviceBuffer<uint8_t>* to = s->win[w][b].get();
    size_t at = kWindowReserve;                 // where the segment starts in `to`
    if (kept > kWindowReserve) {
      // more history than the reserve holds (not seen in practice): move the segment into a larger window
      std::unique_ptr<DeviceBuffer<uint8_t>> big(new DeviceBuffer<uint8_t>());
      if (!big->reserve(kept + std::max<size_t>(nbytes[b], 16))) return broken(nullptr);
      if (nbytes[b] && hipMemcpyAsync(big->get() + kept, to->get() + kWindowReserve, nbytes[b], hipMemcpyDeviceToDevice, st) != hipSuccess) return broken("stream_feed: window move failed");
      if (hipStreamSynchronize(st) != hipSuccess) return broken("stream_feed: window move failed");
      s->win[w][b] = std::move(big);
      to = s->win[w][b].get();
      at = kept;
    }
    // stream byte x of the bytes still held lives at from + org + (x - base); all streams' moves go in one launch behind the loop
    if (kept >= (size_t(1) << 32)) return broken("stream_feed: more than 4 GiB of a stream's past still referenced");   // (CopyDesc sizes are 32-bit)
    if (kept) {
      moves.push_back(CopyDesc{from.get() + s->org[b] + (need - s->base[b]), to->get() + at - kept, static_cast<uint32_t>(kept), 0});
      longest_move = std::max(longest_move, static_cast<uint32_t>(kept));
    }
    s->org[b] = at - kept;
    s->base[b] = need;
    s->avail[b] += static_cast<int64_t>(nbytes[b]);
    virt[b] = to->get() + at - kept - need;    // byte x of the stream lives at virt[b][x]
    avail[b] = static_cast<size_t>(s->avail[b]);
  }
  if (!moves.empty() && !(s->d_history_descs.upload(moves, st) &&
                          launch_device_gather(s->d_history_descs.get(), static_cast<int>(moves.size()), longest_move, st) == hipSuccess))
    return broken("stream_feed: window move failed");
  ++s->fed;
  const int64_t frames = s->eng.feed(virt.data(), avail.data(), s->n, s->first);
  if (frames < 0) return broken(nullptr);        // (the engine's error text stands)
  // the prefetch stream is only ever waited for through events (engine.hpp: blocking_copy): every 32nd segment, wait for the stream itself -- at
  // most the upload of the next segment, which the next feed needs anyway
  if (s->queued_ever && ++s->up_uses % kReapEvery == 0 && reap_enabled())
    for (auto up : s->up_stream) (void)hipStreamSynchronize(up);
  s->first = false;
  return frames;
}
// A session over streams that LIVE in device memory, without any copy: base[b][x] is byte x of stream b counted from the session's start, of which
// the first avail[b] are there now (avail never shrinks).  The caller keeps the bytes from dabhip_stream_need_from(s, b) on in place -- a linear
// buffer that is appended to -- and may recycle what lies below.  base may change between calls as long as those bytes stay addressable through it.
// Not to be mixed with dabhip_stream_feed / _prefetch on the same session (those own their windows).
extern "C" int64_t dabhip_stream_feed_resident(dabhip_stream* s, const uint8_t* const* base, const size_t* avail)
{
  if (!s || !base || !avail) { set_error("stream_feed_resident: null argument"); return -1; }
  if (s->failed) { set_error("stream_feed_resident: an earlier feed of this session failed half-way -- destroy the session"); return -1; }
  if (s->queued > 0) { set_error("stream_feed_resident: this session is fed through dabhip_stream_feed (windows)"); return -1; }
  if (hipSetDevice(s->eng.device()) != hipSuccess) { set_error("stream_feed_resident: hipSetDevice failed"); return -1; }
  for (int b = 0; b < s->n; ++b)
    if (static_cast<int64_t>(avail[b]) < s->avail[b]) { set_error("stream_feed_resident: a stream's byte count went down"); return -1; }
  if (s->first)                                // the kernels dereference these addresses: a host pointer here would be a fault on the device, not an error code
    for (int b = 0; b < s->n; ++b) {
      if (avail[b] == 0) continue;
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, base[b]) != hipSuccess || (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)) {
        (void)hipGetLastError();
        set_error("stream_feed_resident: base[" + std::to_string(b) + "] is not device memory");
        return -1;
      }
    }
  const int64_t frames = s->eng.feed(base, avail, s->n, s->first);
  if (frames < 0) { s->failed = true; return -1; }     // (the engine's error text stands)
  for (int b = 0; b < s->n; ++b) s->avail[b] = static_cast<int64_t>(avail[b]);
  s->resident = true;
  s->first = false;
  return frames;
}
// oldest byte of stream b a later segment may still read (FIFO backlog and stale-tail sources of the front end): everything below may go
extern "C" int64_t dabhip_stream_need_from(const dabhip_stream* s, int stream)
{
  if (!s || stream < 0 || stream >= s->n) return -1;
  return s->first ? 0 : std::min(s->eng.stream_need_from(stream), s->avail[stream]);
}
extern "C" int64_t dabhip_stream_eti_count(const dabhip_stream* s, int stream) { return s ? s->eng.eti_count(stream) : -1; }
// stage times of the segment fed last (the names of dabhip_engine_stage_ms; "wall" = the engine's part of the feed, without the window moves)
extern "C" int dabhip_stream_stage_ms(const dabhip_stream* s, const char** names, float* ms, int cap)
{
  if (!s) return -1;
  static const char* kNames[13] = {"sync", "fft", "demap", "fic", "control", "gather", "viterbi", "eti", "host_setup", "host_frames", "host_worklist", "wall",
                                   "sync_spec_calls"};
  const StageTimes& t = s->eng.stage_times();
  const float v[13] = {t.sync, t.fft, t.demap, t.fic, t.control, t.gather, t.viterbi, t.eti, t.setup, t.frames, t.worklist, t.wall, t.sync_spec_calls};
  int n = 0;
  for (; n < 13 && n < cap; ++n) {
    if (names) names[n] = kNames[n];
    if (ms) ms[n] = v[n];
  }
  return n;
}
extern "C" uint32_t dabhip_stream_status(const dabhip_stream* s, int stream) { return s ? s->eng.stream_status(stream) : 0xffffffffu; }
extern "C" int64_t dabhip_stream_log(dabhip_stream* s, int stream, char* buf, int64_t cap)
{
  if (!s || stream < 0 || stream >= s->n) return -1;
  return hand_over_text(s->eng.take_stream_log(stream), buf, cap);
}
extern "C" int64_t dabhip_stream_eti_read(dabhip_stream* s, int stream, uint8_t* dst, int64_t cap_frames)
{
  if (!s || !dst) { set_error("stream_eti_read: null argument"); return -1; }
  return s->eng.eti_read(stream, dst, cap_frames);
}
extern "C" int64_t dabhip_stream_eti_fetch(dabhip_stream* s, uint8_t* dst, int64_t cap_frames)
{
  if (!s) { set_error("stream_eti_fetch: null handle"); return -1; }
  return s->eng.eti_fetch_async(dst, cap_frames);
}
extern "C" int dabhip_stream_eti_fetch_wait(dabhip_stream* s)
{
  if (!s) { set_error("stream_eti_fetch_wait: null handle"); return -1; }
  return s->eng.eti_fetch_wait() ? 0 : -1;
}
extern "C" int64_t dabhip_stream_eti_drain(dabhip_stream* s, dabhip_eti_sink sink, void* user)
{
  if (!s || !sink) { set_error("stream_eti_drain: null argument"); return -1; }
  int64_t total = 0;
  std::vector<uint8_t> buf;
  for (int b = 0; b < s->n; ++b) {
    const int64_t n = s->eng.eti_count(b);
    if (n < 0) return -1;
    buf.resize(static_cast<size_t>(n) * DABHIP_ETI_BYTES);
    if (n && s->eng.eti_read(b, buf.data(), n) != n) return -1;
    for (int64_t f = 0; f < n; ++f) sink(buf.data() + f * DABHIP_ETI_BYTES, b, user);
    total += n;
  }
  return total;
}
extern "C" int dabhip_stream_ceiling(int device, size_t bytes, int reps, double* gbs)
{
  if (!gbs) { set_error("stream_ceiling: null argument"); return -1; }
  if (dabhip::stream_ceiling(device, bytes, reps, gbs) != 0) { set_error("stream_ceiling: allocation or launch failed"); return -1; }
  return 0;
}

// which physical device an index is: PCI bus id ("0000:c1:00.0") and marketing name -- what a multi-rank run records per rank, so that N ranks can be
// shown to have sat on N distinct GPUs (bench.py: ranks[].device)
extern "C" int dabhip_device_identity(int device, char* bus_id, int cap_bus, char* name, int cap_name)
{
  if (!bus_id || cap_bus < 16 || !name || cap_name < 2) { set_error("device_identity: buffers too small"); return -1; }
  hipDeviceProp_t prop;
  if (hipDeviceGetPCIBusId(bus_id, cap_bus, device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
    (void)hipGetLastError();
    set_error("device_identity: no such device: " + std::to_string(device));
    return -1;
  }
  std::snprintf(name, static_cast<size_t>(cap_name), "%s", prop.name);
  return 0;
}

// device memory for callers that bring no GPU runtime of their own (the batch entries take device pointers)
extern "C" void* dabhip_device_alloc(size_t nbytes, int device)
{
  void* p = nullptr;
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, nbytes ? nbytes : 1) != hipSuccess) { set_error("device_alloc: hipMalloc of " + std::to_string(nbytes) + " bytes failed"); return nullptr; }
  return p;
}
extern "C" void dabhip_device_free(void* p) { if (p) (void)hipFree(p); }
extern "C" int dabhip_device_copy(void* dst, const void* src, size_t nbytes, int to_device)
{
  if (!dst || !src) { set_error("device_copy: null argument"); return -1; }
  if (blocking_copy(dst, src, nbytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost) != hipSuccess) { set_error("device_copy: hipMemcpy failed"); return -1; }
  return 0;
}

// page-locked host memory for the segments handed to dabhip_stream_feed (read the next one while this one decodes)
extern "C" void* dabhip_host_alloc(size_t nbytes)
{
  void* p = nullptr;
  if (hipHostMalloc(&p, nbytes, hipHostMallocDefault) != hipSuccess) { set_error("host_alloc: hipHostMalloc failed"); return nullptr; }
  return p;
}
extern "C" void dabhip_host_free(void* p) { if (p) (void)hipHostFree(p); }
