// multi.cpp — the batch engine over several MI355X of one node (include/dabhip.h: dabhip_multi_*).
//
// Ensembles are independent end to end, so a batch shards by stream with no data-path exchange: the
// streams are dealt to the listed devices in contiguous slices -- slice i of n takes B / n streams, the first B mod n slices one more
// (dabhip_multi_plan says which, before or after a decode: 2048 streams on 8 devices = 256 each, stream s on device s / 256; 10 on 4 =
// 3, 3, 2, 2) --, every slice is one
// complete batch engine with its own persistent host thread, HIP streams and control-plane pool, and the slices run
// concurrently.  No collective, no peer access, no RCCL.  The ETI frames come back in stream order whatever device made
// them.  This is the single-process form of what bench.py does with one process per GPU; dab2eti.c:237,279-302 (one demod
// thread, one device) is what it stands in for.
//
// A device may be listed more than once: every entry is a slice of its own (tests map all eight slices of a node onto GPU 0).
#include <algorithm>
#include <chrono>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/dabhip.h"
#include "placement.hpp"
#include "thread_pool.hpp"

namespace dabhip {
void set_error(const std::string& msg);
}
using dabhip::AsyncLane;
using dabhip::set_error;

struct dabhip_multi {
  struct Slice {
    int device = 0;
    dabhip_engine* eng = nullptr;
    std::unique_ptr<AsyncLane> lane;     // the slice's host thread: its decode calls run here
    std::vector<int> cpus;               // the CPUs its host threads are bound to (placement.hpp); empty = unbound
    int numa_node = -1;
    int first = 0, count = 0;            // streams [first, first + count) of the last decode
    int64_t frames = 0;
    float wall_ms = 0;
    std::string error;
  };
  std::vector<Slice> slices;
  int nstreams = 0;
  float wall_ms = 0;

  ~dabhip_multi()
  {
    for (Slice& s : slices) {
      s.lane.reset();
      if (s.eng) dabhip_engine_destroy(s.eng);
    }
  }
  const Slice* slice_of(int stream) const
  {
    if (stream < 0 || stream >= nstreams) return nullptr;
    for (const Slice& s : slices)
      if (stream >= s.first && stream < s.first + s.count) return &s;
    return nullptr;
  }
};

extern "C" {

dabhip_multi* dabhip_multi_create(const int* devices, int n)
{
  if (!devices || n <= 0 || n > 64) { set_error("multi_create: need 1..64 devices"); return nullptr; }
  std::unique_ptr<dabhip_multi> m(new (std::nothrow) dabhip_multi);
  if (!m) return nullptr;
  // host threads per slice: the slices share the host, so that eight of them do not start 8 x 24 busy threads
  // (of the CPUs this process may use -- affinity mask and CFS quota, placement.hpp --, not of the machine's)
  const int hw = dabhip::usable_cpus();
  const int host_threads = std::max(2, std::min(24, hw / (2 * n)));
  m->slices.resize(n);
  // host placement (placement.hpp): every slice's threads -- its decode thread here, the engine's control-plane pool and host lane -- on the NUMA
  // node of its device, the node's CPUs dealt to the slices on it in disjoint chunks; page-locked buffers are allocated by those threads, i.e. there
  std::vector<int> nodes(static_cast<size_t>(n), -1);
  if (dabhip::numa_enabled())
    for (int i = 0; i < n; ++i) {
      char bdf[32] = {0};
      if (hipDeviceGetPCIBusId(bdf, sizeof bdf, devices[i]) == hipSuccess) nodes[static_cast<size_t>(i)] = dabhip::numa_node_of_pci(bdf);
      else (void)hipGetLastError();
    }
  const std::vector<std::vector<int>> node_cpus = dabhip::allowed_node_cpus();
  int populated = 0;
  for (const auto& c : node_cpus) populated += c.empty() ? 0 : 1;
  std::vector<std::vector<int>> plan(static_cast<size_t>(n));
  if (populated > 1) plan = dabhip::plan_placement(nodes, node_cpus);          // a single-socket machine: nothing to choose, nothing bound
  for (int i = 0; i < n; ++i) {
    dabhip_multi::Slice& s = m->slices[i];
    s.device = devices[i];
    s.cpus = plan[static_cast<size_t>(i)];
    s.numa_node = nodes[static_cast<size_t>(i)];
    std::vector<int32_t> c32(s.cpus.begin(), s.cpus.end());
    s.eng = dabhip_engine_create_on_cpus(devices[i], host_threads, c32.data(), static_cast<int>(c32.size()));
    if (!s.eng) return nullptr;            // dabhip_last_error() says why (bad index, no GPU: there is no CPU fallback)
    s.lane.reset(new AsyncLane(s.cpus));
  }
  return m.release();
}

void dabhip_multi_destroy(dabhip_multi* m) { delete m; }

int dabhip_multi_slices(const dabhip_multi* m) { return m ? static_cast<int>(m->slices.size()) : -1; }

// The dealing rule as a pure function of (number of slices, batch size): usable BEFORE the first decode, e.g. to put stream b's samples on the
// right device for an on_device decode.  Slice i takes nstreams / n streams, the first nstreams % n slices one more.
int dabhip_multi_plan(const dabhip_multi* m, int nstreams, int stream, int* slice, int* device)
{
  if (!m || nstreams <= 0 || stream < 0 || stream >= nstreams) { set_error("multi_plan: bad argument"); return -1; }
  const int n = static_cast<int>(m->slices.size()), base = nstreams / n, rem = nstreams % n;
  // the first rem slices hold base + 1 streams each
  const int i = stream < rem * (base + 1) ? stream / (base + 1) : rem + (stream - rem * (base + 1)) / std::max(base, 1);
  if (slice) *slice = i;
  if (device) *device = m->slices[static_cast<size_t>(i)].device;
  return 0;
}
int dabhip_multi_slice_cpus(const dabhip_multi* m, int slice, int32_t* cpus, int cap, int* numa_node)
{
  if (!m || slice < 0 || slice >= static_cast<int>(m->slices.size())) return -1;
  const dabhip_multi::Slice& s = m->slices[static_cast<size_t>(slice)];
  if (numa_node) *numa_node = s.numa_node;
  for (int i = 0; cpus && i < cap && i < static_cast<int>(s.cpus.size()); ++i) cpus[i] = s.cpus[static_cast<size_t>(i)];
  return static_cast<int>(s.cpus.size());
}

int dabhip_multi_slice_of(const dabhip_multi* m, int stream, int* device)
{
  if (!m) return -1;
  const dabhip_multi::Slice* s = m->slice_of(stream);
  if (!s) return -1;
  if (device) *device = s->device;
  return static_cast<int>(s - m->slices.data());
}

int64_t dabhip_multi_decode(dabhip_multi* m, const uint8_t* const* iq, const size_t* nbytes, int nstreams, int on_device)
{
  if (!m || !iq || !nbytes) { set_error("multi_decode: null argument"); return -1; }
  if (nstreams <= 0) { set_error("multi_decode: no streams"); return -1; }
  const auto t0 = std::chrono::steady_clock::now();
  const int n = static_cast<int>(m->slices.size());
  const int base = nstreams / n, rem = nstreams % n;
  int next = 0;
  for (int i = 0; i < n; ++i) {
    dabhip_multi::Slice& s = m->slices[i];
    s.first = next;
    s.count = base + (i < rem ? 1 : 0);
    next += s.count;
    s.frames = 0;
    s.wall_ms = 0;
    s.error.clear();
  }
  m->nstreams = nstreams;
  for (dabhip_multi::Slice& s : m->slices) {
    if (s.count == 0) continue;
    dabhip_multi::Slice* sp = &s;
    s.lane->post([sp, iq, nbytes, on_device]() {
      const auto t = std::chrono::steady_clock::now();
      sp->frames = dabhip_engine_decode(sp->eng, iq + sp->first, nbytes + sp->first, sp->count, on_device);
      if (sp->frames < 0) sp->error = dabhip_last_error();      // thread-local text: carried to the caller's thread below
      sp->wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t).count();
    });
  }
  for (dabhip_multi::Slice& s : m->slices)
    if (s.count) s.lane->wait();           // every slice is awaited, also after a failure: nothing may stay in flight
  int64_t total = 0;
  for (size_t i = 0; i < m->slices.size(); ++i) {
    const dabhip_multi::Slice& s = m->slices[i];
    if (s.frames < 0) { set_error("slice " + std::to_string(i) + " (device " + std::to_string(s.device) + "): " + s.error); return -1; }
    total += s.frames;
  }
  m->wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return total;
}

int64_t dabhip_multi_eti_count(const dabhip_multi* m, int stream)
{
  const dabhip_multi::Slice* s = m ? m->slice_of(stream) : nullptr;
  return s ? dabhip_engine_eti_count(s->eng, stream - s->first) : -1;
}

uint32_t dabhip_multi_stream_status(const dabhip_multi* m, int stream)
{
  const dabhip_multi::Slice* s = m ? m->slice_of(stream) : nullptr;
  return s ? dabhip_engine_stream_status(s->eng, stream - s->first) : 0xffffffffu;
}

int64_t dabhip_multi_stream_log(dabhip_multi* m, int stream, char* buf, int64_t cap)
{
  const dabhip_multi::Slice* s = m ? m->slice_of(stream) : nullptr;
  return s ? dabhip_engine_stream_log(s->eng, stream - s->first, buf, cap) : -1;
}

int64_t dabhip_multi_eti_read(dabhip_multi* m, int stream, uint8_t* dst, int64_t cap_frames)
{
  if (!m || !dst) { set_error("multi_eti_read: null argument"); return -1; }
  const dabhip_multi::Slice* s = m->slice_of(stream);
  if (!s) { set_error("multi_eti_read: bad stream"); return -1; }
  return dabhip_engine_eti_read(s->eng, stream - s->first, dst, cap_frames);
}

int64_t dabhip_multi_eti_drain(dabhip_multi* m, dabhip_eti_sink sink, void* user)
{
  if (!m || !sink) { set_error("multi_eti_drain: null argument"); return -1; }
  int64_t total = 0;
  std::vector<uint8_t> buf;
  for (int b = 0; b < m->nstreams; ++b) {   // stream order = slice order: the slices are contiguous
    const int64_t n = dabhip_multi_eti_count(m, b);
    if (n < 0) return -1;
    buf.resize(static_cast<size_t>(n) * DABHIP_ETI_BYTES);
    if (n && dabhip_multi_eti_read(m, b, buf.data(), n) != n) return -1;
    for (int64_t f = 0; f < n; ++f) sink(buf.data() + f * DABHIP_ETI_BYTES, b, user);
    total += n;
  }
  return total;
}

int dabhip_multi_trace(const dabhip_multi* m, int stream, int32_t* ints6, double* ffs, int cap_calls)
{
  const dabhip_multi::Slice* s = m ? m->slice_of(stream) : nullptr;
  return s ? dabhip_engine_trace(s->eng, stream - s->first, ints6, ffs, cap_calls) : -1;
}

dabhip_engine* dabhip_multi_engine(dabhip_multi* m, int slice)
{
  if (!m || slice < 0 || slice >= static_cast<int>(m->slices.size())) return nullptr;
  return m->slices[slice].eng;
}

// wall clock of the last decode: the whole call, and one slice's own decode (slice >= 0)
float dabhip_multi_wall_ms(const dabhip_multi* m, int slice)
{
  if (!m) return -1.0f;
  if (slice < 0) return m->wall_ms;
  return slice < static_cast<int>(m->slices.size()) ? m->slices[slice].wall_ms : -1.0f;
}

#define DABHIP_MULTI_FORWARD(name, call)                       \
  int name                                                     \
  {                                                            \
    if (!m) return -1;                                         \
    for (auto& s : m->slices)                                  \
      if (call != 0) return -1;                                \
    return 0;                                                  \
  }
DABHIP_MULTI_FORWARD(dabhip_multi_set_afc(dabhip_multi* m, int enable), dabhip_engine_set_afc(s.eng, enable))
DABHIP_MULTI_FORWARD(dabhip_multi_set_soft(dabhip_multi* m, int enable), dabhip_engine_set_soft(s.eng, enable))
DABHIP_MULTI_FORWARD(dabhip_multi_set_parity_guard(dabhip_multi* m, int enable), dabhip_engine_set_parity_guard(s.eng, enable))
DABHIP_MULTI_FORWARD(dabhip_multi_set_fused(dabhip_multi* m, int enable), dabhip_engine_set_fused(s.eng, enable))
DABHIP_MULTI_FORWARD(dabhip_multi_set_subchannels(dabhip_multi* m, const int32_t* ids, int n), dabhip_engine_set_subchannels(s.eng, ids, n))
#undef DABHIP_MULTI_FORWARD

}  // extern "C"

// ---- sessions over several devices (include/dabhip.h: dabhip_multi_stream_*) -------------------------------------------------
// The unbounded-input form of the above (dab2eti.c:60-130 is a session: calls arrive for ever): B streams dealt ONCE, at creation, to the listed
// devices in the same contiguous slices; every slice is a complete dabhip_stream session (device windows, carried front-end / lock / ring state) with
// its own host thread, and a feed / prefetch / fetch is the same call made on every slice at once with that slice's part of the pointer arrays.
// Nothing crosses a slice boundary, so the frames of a segment are those of ONE session over all B streams, in global stream order.
struct dabhip_multi_stream {
  struct Slice {
    int device = 0;
    dabhip_stream* s = nullptr;
    std::unique_ptr<AsyncLane> lane;
    std::vector<int> cpus;
    int numa_node = -1;
    int first = 0, count = 0;
    int64_t frames = 0;                  // of the segment fed last
    int64_t rc = 0;                      // result of the call posted last
    std::string error;
  };
  std::vector<Slice> slices;
  int nstreams = 0;
  bool failed = false;                   // a slice failed a feed: the slices are not in step any more
  // fetches issued and not yet waited for, oldest first (up to two: the CLI's writer still waits for segment k's frames while the decode thread, through with
  // segment k + 1, issues the next one): per fetch, which slices took part (a slice without frames issues nothing and must not be waited on)
  std::deque<std::vector<uint8_t>> fetches;
  std::mutex fetch_mu;

  ~dabhip_multi_stream()
  {
    for (Slice& sl : slices) {
      sl.lane.reset();
      if (sl.s) dabhip_stream_destroy(sl.s);
    }
  }
  Slice* slice_of(int stream)
  {
    if (stream < 0 || stream >= nstreams) return nullptr;
    for (Slice& sl : slices)
      if (stream >= sl.first && stream < sl.first + sl.count) return &sl;
    return nullptr;
  }
  const Slice* slice_of(int stream) const { return const_cast<dabhip_multi_stream*>(this)->slice_of(stream); }
  // fn(slice) on every slice's host thread, all at once; false (and the text of the first failure) when one of them returned < 0
  template <class Fn>
  bool on_all(const char* what, Fn fn)
  {
    for (Slice& sl : slices) {
      if (sl.count == 0) continue;
      Slice* sp = &sl;
      sl.lane->post([sp, fn]() {
        sp->error.clear();
        sp->rc = fn(*sp);
        if (sp->rc < 0) sp->error = dabhip_last_error();       // thread-local text: carried to the caller's thread below
      });
    }
    for (Slice& sl : slices)
      if (sl.count) sl.lane->wait();       // every slice is awaited, also after a failure: nothing may stay in flight
    for (size_t i = 0; i < slices.size(); ++i)
      if (slices[i].count && slices[i].rc < 0) {
        set_error(std::string(what) + ": slice " + std::to_string(i) + " (device " + std::to_string(slices[i].device) + "): " + slices[i].error);
        return false;
      }
    return true;
  }
};

extern "C" {

dabhip_multi_stream* dabhip_multi_stream_create(const int* devices, int n, int nstreams)
{
  if (!devices || n <= 0 || n > 64) { set_error("multi_stream_create: need 1..64 devices"); return nullptr; }
  if (nstreams <= 0) { set_error("multi_stream_create: no streams"); return nullptr; }
  std::unique_ptr<dabhip_multi_stream> m(new (std::nothrow) dabhip_multi_stream);
  if (!m) return nullptr;
  const int hw = dabhip::usable_cpus();
  const int host_threads = std::max(2, std::min(24, hw / (2 * n)));
  std::vector<int> nodes(static_cast<size_t>(n), -1);
  if (dabhip::numa_enabled())
    for (int i = 0; i < n; ++i) {
      char bdf[32] = {0};
      if (hipDeviceGetPCIBusId(bdf, sizeof bdf, devices[i]) == hipSuccess) nodes[static_cast<size_t>(i)] = dabhip::numa_node_of_pci(bdf);
      else (void)hipGetLastError();
    }
  const std::vector<std::vector<int>> node_cpus = dabhip::allowed_node_cpus();
  int populated = 0;
  for (const auto& c : node_cpus) populated += c.empty() ? 0 : 1;
  std::vector<std::vector<int>> plan(static_cast<size_t>(n));
  if (populated > 1) plan = dabhip::plan_placement(nodes, node_cpus);
  m->slices.resize(static_cast<size_t>(n));
  m->nstreams = nstreams;
  const int base = nstreams / n, rem = nstreams % n;           // the dealing rule of dabhip_multi_plan
  int next = 0;
  for (int i = 0; i < n; ++i) {
    dabhip_multi_stream::Slice& sl = m->slices[static_cast<size_t>(i)];
    sl.device = devices[i];
    sl.cpus = plan[static_cast<size_t>(i)];
    sl.numa_node = nodes[static_cast<size_t>(i)];
    sl.first = next;
    sl.count = base + (i < rem ? 1 : 0);
    next += sl.count;
    if (sl.count == 0) continue;           // more devices than streams: the slice stays empty
    std::vector<int32_t> c32(sl.cpus.begin(), sl.cpus.end());
    sl.s = dabhip_stream_create_on_cpus(devices[i], sl.count, host_threads, c32.data(), static_cast<int>(c32.size()));
    if (!sl.s) return nullptr;             // dabhip_last_error() says why
    sl.lane.reset(new AsyncLane(sl.cpus));
  }
  return m.release();
}

void dabhip_multi_stream_destroy(dabhip_multi_stream* m) { delete m; }
int dabhip_multi_stream_slices(const dabhip_multi_stream* m) { return m ? static_cast<int>(m->slices.size()) : -1; }
int dabhip_multi_stream_streams(const dabhip_multi_stream* m) { return m ? m->nstreams : -1; }
int dabhip_multi_stream_slice_of(const dabhip_multi_stream* m, int stream, int* device, int* first, int* count)
{
  const dabhip_multi_stream::Slice* sl = m ? m->slice_of(stream) : nullptr;
  if (!sl) return -1;
  if (device) *device = sl->device;
  if (first) *first = sl->first;
  if (count) *count = sl->count;
  return static_cast<int>(sl - m->slices.data());
}
dabhip_stream* dabhip_multi_stream_session(dabhip_multi_stream* m, int slice)
{
  if (!m || slice < 0 || slice >= static_cast<int>(m->slices.size())) return nullptr;
  return m->slices[static_cast<size_t>(slice)].s;
}

int dabhip_multi_stream_prefetch(dabhip_multi_stream* m, const uint8_t* const* iq, const size_t* nbytes, int on_device)
{
  if (!m || !iq || !nbytes) { set_error("multi_stream_prefetch: null argument"); return -1; }
  if (m->failed) { set_error("multi_stream_prefetch: an earlier feed of this session failed -- destroy the session"); return -1; }
  return m->on_all("multi_stream_prefetch", [iq, nbytes, on_device](dabhip_multi_stream::Slice& sl) -> int64_t {
    return dabhip_stream_prefetch(sl.s, iq + sl.first, nbytes + sl.first, on_device);
  }) ? 0 : -1;
}

int64_t dabhip_multi_stream_feed(dabhip_multi_stream* m, const uint8_t* const* iq, const size_t* nbytes, int on_device)
{
  if (!m || !iq || !nbytes) { set_error("multi_stream_feed: null argument"); return -1; }
  if (m->failed) { set_error("multi_stream_feed: an earlier feed of this session failed; the slices are not in step any more -- destroy the session"); return -1; }
  const bool ok = m->on_all("multi_stream_feed", [iq, nbytes, on_device](dabhip_multi_stream::Slice& sl) -> int64_t {
    return sl.frames = dabhip_stream_feed(sl.s, iq + sl.first, nbytes + sl.first, on_device);
  });
  if (!ok) { m->failed = true; return -1; }
  int64_t total = 0;
  for (const auto& sl : m->slices) total += sl.count ? sl.frames : 0;
  return total;
}

int64_t dabhip_multi_stream_feed_resident(dabhip_multi_stream* m, const uint8_t* const* base, const size_t* avail)
{
  if (!m || !base || !avail) { set_error("multi_stream_feed_resident: null argument"); return -1; }
  if (m->failed) { set_error("multi_stream_feed_resident: an earlier feed of this session failed -- destroy the session"); return -1; }
  const bool ok = m->on_all("multi_stream_feed_resident", [base, avail](dabhip_multi_stream::Slice& sl) -> int64_t {
    return sl.frames = dabhip_stream_feed_resident(sl.s, base + sl.first, avail + sl.first);
  });
  if (!ok) { m->failed = true; return -1; }
  int64_t total = 0;
  for (const auto& sl : m->slices) total += sl.count ? sl.frames : 0;
  return total;
}

int64_t dabhip_multi_stream_need_from(const dabhip_multi_stream* m, int stream)
{
  const dabhip_multi_stream::Slice* sl = m ? m->slice_of(stream) : nullptr;
  return sl ? dabhip_stream_need_from(sl->s, stream - sl->first) : -1;
}
int64_t dabhip_multi_stream_eti_count(const dabhip_multi_stream* m, int stream)
{
  const dabhip_multi_stream::Slice* sl = m ? m->slice_of(stream) : nullptr;
  return sl ? dabhip_stream_eti_count(sl->s, stream - sl->first) : -1;
}
uint32_t dabhip_multi_stream_status_of(const dabhip_multi_stream* m, int stream)
{
  const dabhip_multi_stream::Slice* sl = m ? m->slice_of(stream) : nullptr;
  return sl ? dabhip_stream_status(sl->s, stream - sl->first) : 0xffffffffu;
}
int64_t dabhip_multi_stream_log_of(dabhip_multi_stream* m, int stream, char* buf, int64_t cap)
{
  dabhip_multi_stream::Slice* sl = m ? m->slice_of(stream) : nullptr;
  return sl ? dabhip_stream_log(sl->s, stream - sl->first, buf, cap) : -1;
}
int64_t dabhip_multi_stream_eti_read(dabhip_multi_stream* m, int stream, uint8_t* dst, int64_t cap_frames)
{
  if (!m || !dst) { set_error("multi_stream_eti_read: null argument"); return -1; }
  dabhip_multi_stream::Slice* sl = m->slice_of(stream);
  if (!sl) { set_error("multi_stream_eti_read: bad stream"); return -1; }
  return dabhip_stream_eti_read(sl->s, stream - sl->first, dst, cap_frames);
}
int64_t dabhip_multi_stream_eti_drain(dabhip_multi_stream* m, dabhip_eti_sink sink, void* user)
{
  if (!m || !sink) { set_error("multi_stream_eti_drain: null argument"); return -1; }
  int64_t total = 0;
  std::vector<uint8_t> buf;
  for (int b = 0; b < m->nstreams; ++b) {     // global stream order = slice order: the slices are contiguous
    const int64_t n = dabhip_multi_stream_eti_count(m, b);
    if (n < 0) return -1;
    buf.resize(static_cast<size_t>(n) * DABHIP_ETI_BYTES);
    if (n && dabhip_multi_stream_eti_read(m, b, buf.data(), n) != n) return -1;
    for (int64_t f = 0; f < n; ++f) sink(buf.data() + f * DABHIP_ETI_BYTES, b, user);
    total += n;
  }
  return total;
}
// The frames of the segment fed last, all slices, in global stream order, as one asynchronous download per slice into dst (page-locked): slice i's frames
// follow those of slices 0 .. i - 1.  Returns their number; the copies have landed when dabhip_multi_stream_eti_fetch_wait returns.
int64_t dabhip_multi_stream_eti_fetch(dabhip_multi_stream* m, uint8_t* dst, int64_t cap_frames)
{
  if (!m || !dst) { set_error("multi_stream_eti_fetch: null argument"); return -1; }
  int64_t total = 0;
  for (const auto& sl : m->slices) total += sl.count ? sl.frames : 0;
  if (total > cap_frames) { set_error("multi_stream_eti_fetch: destination too small"); return -1; }
  int64_t at = 0;
  std::vector<int64_t> offset(m->slices.size(), 0);
  std::vector<uint8_t> took_part(m->slices.size(), 0);
  for (size_t i = 0; i < m->slices.size(); ++i) {
    offset[i] = at;
    const int64_t n = m->slices[i].count ? m->slices[i].frames : 0;
    took_part[i] = n > 0;
    at += n;
  }
  dabhip_multi_stream::Slice* base = m->slices.data();
  const int64_t* off = offset.data();
  const bool ok = m->on_all("multi_stream_eti_fetch", [dst, base, off](dabhip_multi_stream::Slice& sl) -> int64_t {
    if (sl.frames == 0) return 0;
    const int64_t got = dabhip_stream_eti_fetch(sl.s, dst + static_cast<size_t>(off[&sl - base]) * DABHIP_ETI_BYTES, sl.frames);
    return got == sl.frames ? got : -1;
  });
  {
    std::lock_guard<std::mutex> lk(m->fetch_mu);
    m->fetches.push_back(std::move(took_part));       // (also after a failure: the slices that did issue a copy are still to be waited for)
  }
  return ok ? total : -1;
}
// waits for the OLDEST fetch not yet waited for (every slice's engine keeps its fetches in issue order, so the slices of that fetch wait for the right one)
int dabhip_multi_stream_eti_fetch_wait(dabhip_multi_stream* m)
{
  if (!m) { set_error("multi_stream_eti_fetch_wait: null handle"); return -1; }
  std::vector<uint8_t> took_part;
  {
    std::lock_guard<std::mutex> lk(m->fetch_mu);
    if (m->fetches.empty()) return 0;
    took_part = std::move(m->fetches.front());
    m->fetches.pop_front();
  }
  int rc = 0;
  // (the wait itself touches only the slice's fetch events: made from the caller's thread, so that a writer thread can wait while the lanes decode)
  for (size_t i = 0; i < m->slices.size(); ++i)
    if (took_part[i] && dabhip_stream_eti_fetch_wait(m->slices[i].s) != 0) rc = -1;
  return rc;
}

#define DABHIP_MULTI_STREAM_FORWARD(name, call)                \
  int name                                                     \
  {                                                            \
    if (!m) return -1;                                         \
    for (auto& sl : m->slices)                                 \
      if (sl.s && call != 0) return -1;                        \
    return 0;                                                  \
  }
DABHIP_MULTI_STREAM_FORWARD(dabhip_multi_stream_set_afc(dabhip_multi_stream* m, int enable), dabhip_stream_set_afc(sl.s, enable))
DABHIP_MULTI_STREAM_FORWARD(dabhip_multi_stream_set_soft(dabhip_multi_stream* m, int enable), dabhip_stream_set_soft(sl.s, enable))
DABHIP_MULTI_STREAM_FORWARD(dabhip_multi_stream_set_parity_guard(dabhip_multi_stream* m, int level), dabhip_stream_set_parity_guard(sl.s, level))
DABHIP_MULTI_STREAM_FORWARD(dabhip_multi_stream_set_sync_speculation(dabhip_multi_stream* m, int mode), dabhip_stream_set_sync_speculation(sl.s, mode))
DABHIP_MULTI_STREAM_FORWARD(dabhip_multi_stream_set_subchannels(dabhip_multi_stream* m, const int32_t* ids, int n), dabhip_stream_set_subchannels(sl.s, ids, n))
#undef DABHIP_MULTI_STREAM_FORWARD

}  // extern "C"
