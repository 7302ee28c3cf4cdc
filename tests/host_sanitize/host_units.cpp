// host_units.cpp — the host-only units of the product (no GPU call in any of them), compiled by tests/host_sanitize/Makefile with
// g++ -fsanitize=thread and -fsanitize=address,undefined and run by the CPU suite (tests/test_host_sanitize.py).
//
//   thread_pool.hpp    ThreadPool (per-stream control-plane pool) and AsyncLane (the decode's host lane), used the way the engine
//                      uses them: a lane task that calls parallel_for while the posting thread waits; several engines' worth at once
//   control_plane.hpp  FIG parse, lock rule, CIF ring, ETI headers over the FIBs of synthetic ensembles, streams in parallel
//   worklist.hpp       frame records, header rows, wave-groups and slices of the MSC decode (with plain std::allocator lists)
//   fifo_view.hpp      the closed-form FIFO / stale-tail views under random timing corrections, against a byte-level replay of
//                      cbWrite / sdr_read_fifo's copying rule (sdr_fifo.c:26-61)
//   synth.cpp          the modulator's bit content and sample generation (bounds, UB)
// The reference's own data race (rtlsdr_callback writing sdr->input_buffer while the demod thread reads it, dab2eti.c:117-130)
// has no counterpart here: segments are handed over by value of their pointers and never written while a decode runs.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#include "../../include/dabhip.h"
#include "../../dabtools_amd/csrc/control_plane.hpp"
#include "../../dabtools_amd/csrc/fifo_view.hpp"
#include "../../dabtools_amd/csrc/thread_pool.hpp"
#include "../../dabtools_amd/csrc/worklist.hpp"

using namespace dabhip;

namespace dabhip {
void set_error(const std::string&) {}      // error.cpp's thread-local text is not linked into this binary
}

#define CHECK(cond)                                                                     \
  do {                                                                                  \
    if (!(cond)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
  } while (0)

template <class T>
using StdAlloc = std::allocator<T>;

static void test_pool_and_lane()
{
  ThreadPool pool(5);
  for (int round = 0; round < 400; ++round) {
    const int n = 1 + (round * 37) % 301;
    std::vector<int> hits(static_cast<size_t>(n), 0);
    pool.parallel_for(n, [&](int i) { hits[static_cast<size_t>(i)] += 1; });      // every index exactly once, visible after the call
    for (int h : hits) CHECK(h == 1);
  }
  pool.parallel_for(0, [&](int) { CHECK(false); });
  // the engine's pattern: the decode thread posts the control-plane pass to the lane and waits; the pass fans out over the pool
  AsyncLane lane;
  long total = 0;
  for (int round = 0; round < 200; ++round) {
    std::vector<long> part(64, 0);
    lane.post([&]() {
      pool.parallel_for(64, [&](int i) { part[static_cast<size_t>(i)] = i + round; });
      for (long p : part) total += p;
    });
    lane.wait();
    CHECK(part[63] == 63 + round);
  }
  CHECK(total == 200L * (63 * 64 / 2) + 64L * (199 * 200 / 2));
  // several posts before one wait run in order
  std::vector<int> order;
  for (int i = 0; i < 50; ++i) lane.post([&order, i]() { order.push_back(i); });
  lane.wait();
  CHECK(order.size() == 50);
  for (int i = 0; i < 50; ++i) CHECK(order[static_cast<size_t>(i)] == i);
  // a pool without workers: the caller does everything
  ThreadPool solo(0);
  int sum = 0;
  solo.parallel_for(10, [&](int i) { sum += i; });
  CHECK(sum == 45);
  // host placement (placement.hpp): workers and lanes are bound to their CPU list before they run anything.  (The CPU this process may use: what
  // the container's cpuset leaves; bind to the first one of it.)
  cpu_set_t allowed;
  CHECK(sched_getaffinity(0, sizeof allowed, &allowed) == 0);
  int cpu0 = -1;
  for (int c = 0; c < CPU_SETSIZE && cpu0 < 0; ++c)
    if (CPU_ISSET(c, &allowed)) cpu0 = c;
  CHECK(cpu0 >= 0);
  {
    ThreadPool bound(3, std::vector<int>{cpu0});
    std::atomic<int> elsewhere{0}, by_workers{0};
    const std::thread::id me = std::this_thread::get_id();
    for (int round = 0; round < 20; ++round)
      bound.parallel_for(64, [&](int) {
        if (std::this_thread::get_id() == me) return;                  // the caller takes part and is not bound
        ++by_workers;
        if (sched_getcpu() != cpu0) ++elsewhere;
      });
    CHECK(elsewhere.load() == 0);
    AsyncLane on_cpu(std::vector<int>{cpu0});
    int where = -1;
    on_cpu.post([&]() { where = sched_getcpu(); });
    on_cpu.wait();
    CHECK(where == cpu0);
  }
  // a CPU list that reaches beyond what the process is allowed on (a launcher's taskset): bound to the part inside; a list wholly outside: left alone
  {
    const int outside = CPU_SETSIZE - 1;                               // (no box has 1024 hardware threads in its cpuset)
    CHECK(!CPU_ISSET(outside, &allowed));
    AsyncLane partly(std::vector<int>{outside, cpu0});
    int where = -1;
    partly.post([&]() { where = sched_getcpu(); });
    partly.wait();
    CHECK(where == cpu0);
    AsyncLane none(std::vector<int>{outside});
    cpu_set_t got;
    CPU_ZERO(&got);
    none.post([&]() { CHECK(sched_getaffinity(0, sizeof got, &got) == 0); });
    none.wait();
    CHECK(CPU_EQUAL(&got, &allowed));
    CHECK(intersect_cpus({5, 1, 9}, {1, 2, 5}) == (std::vector<int>{5, 1}));
  }
  // the quota rule: cgroup v2's cpu.max
  CHECK(parse_cpu_max("max 100000") == 0 && parse_cpu_max("1600000 100000") == 16 && parse_cpu_max("150000 100000") == 2 && parse_cpu_max("") == 0);
  CHECK(usable_cpus() >= 1 && usable_cpus() <= static_cast<int>(allowed_cpus().size()));
  // two callers of one pool (the engine's decode thread and its host lane never overlap, but nothing must break if they did)
  {
    ThreadPool shared(4);
    std::vector<int> a(500, 0), b(500, 0);
    std::thread other([&]() { for (int r = 0; r < 50; ++r) shared.parallel_for(500, [&](int i) { a[static_cast<size_t>(i)] += 1; }); });
    for (int r = 0; r < 50; ++r) shared.parallel_for(500, [&](int i) { b[static_cast<size_t>(i)] += 1; });
    other.join();
    for (int i = 0; i < 500; ++i) CHECK(a[static_cast<size_t>(i)] == 50 && b[static_cast<size_t>(i)] == 50);
  }
  // the plan itself: disjoint chunks of a node's CPUs
  {
    const std::vector<std::vector<int>> plan = plan_placement({0, 1, 0, -1, 0}, {parse_cpulist("0-3,8-11"), parse_cpulist("4-7")});
    CHECK(plan.size() == 5 && plan[3].empty() && plan[1] == (std::vector<int>{4, 5, 6, 7}));
    CHECK(plan[0] == (std::vector<int>{0, 1}) && plan[2] == (std::vector<int>{2, 3, 8}) && plan[4] == (std::vector<int>{9, 10, 11}));
  }
}

// the FIBs and CRC flags of `ntf` transmission frames of one synthetic ensemble, as the FIC decode hands them to the control plane
static void make_fibs(int preset, uint64_t seed, int cif0, int ntf, std::vector<uint8_t>& fibs, std::vector<uint8_t>& ok)
{
  dabhip_synth_cfg cfg;
  CHECK(dabhip_synth_preset(preset, &cfg) == 0);
  cfg.seed = seed;
  cfg.cif_count0 = cif0;
  fibs.assign(static_cast<size_t>(ntf) * 384, 0);
  ok.assign(static_cast<size_t>(ntf) * 12, 1);
  for (int c = 0; c < 4 * ntf; ++c) CHECK(dabhip_synth_fibs(&cfg, c, fibs.data() + static_cast<size_t>(c) * 96) == 96);
}

// one engine's host side for a batch: control plane over the pool, then the work lists -- checked for internal consistency
static void control_and_worklist(int nstreams, int ntf, int64_t max_rows, unsigned salt)
{
  ThreadPool pool(3);
  std::vector<std::vector<uint8_t>> fibs(static_cast<size_t>(nstreams)), ok(static_cast<size_t>(nstreams));
  for (int b = 0; b < nstreams; ++b) {
    make_fibs(b % 2, 100 + salt + static_cast<unsigned>(b), (977 * b + static_cast<int>(salt)) % 5000, ntf, fibs[static_cast<size_t>(b)], ok[static_cast<size_t>(b)]);
    if (b % 5 == 3)                       // a stream that loses lock four TFs before its end: one TF with a failed FIB CRC (dab.c:55-61)
      ok[static_cast<size_t>(b)][static_cast<size_t>(12 * (ntf - 4) + 4)] = 0;
  }
  std::vector<ControlPlane> planes(static_cast<size_t>(nstreams));
  std::vector<JobList> jobs(static_cast<size_t>(nstreams));
  pool.parallel_for(nstreams, [&](int b) {
    const size_t sb = static_cast<size_t>(b);
    planes[sb] = ControlPlane();
    if (b % 7 == 6) planes[sb].set_filter(0x6ull);                 // sub-channel filter: SubChIds 1 and 2 only
    for (int t = 0; t < ntf; ++t) {
      planes[sb].on_tf(t, fibs[sb].data() + static_cast<size_t>(t) * 384, ok[sb].data() + static_cast<size_t>(t) * 12, jobs[sb]);
      if (planes[sb].locked() && t >= 10) {                // the cached header against the straightforward builder, at this TF's CIF counter
        uint8_t fast[kEtiHeaderMax], slow[kEtiHeaderMax];
        const int n1 = planes[sb].frame_header(fast), n2 = build_eti_header(slow, planes[sb].ensemble(), planes[sb].filter());
        CHECK(n1 == n2 && std::memcmp(fast, slow, static_cast<size_t>(n1)) == 0);
      }
    }
  });
  std::vector<const ControlPlane*> plane_ptrs;
  std::vector<const JobList*> job_ptrs;
  std::vector<int> row_base, fib_base;
  size_t nf = 0;
  for (int b = 0; b < nstreams; ++b) {
    const size_t sb = static_cast<size_t>(b);
    plane_ptrs.push_back(&planes[sb]);
    job_ptrs.push_back(&jobs[sb]);
    row_base.push_back(15 + b * (4 * ntf + 15));
    fib_base.push_back(4 * ntf * b);
    if (b % 5 != 3) CHECK(static_cast<int>(jobs[sb].size()) == 4 * (ntf - 13));      // 10 TFs to lock (dab.c:50-53), 16 CIFs in the ring before the first frame
    else CHECK(static_cast<int>(jobs[sb].size()) == 4 * (ntf - 4 - 13));
    nf += jobs[sb].size();
  }
  PlanTable plans;
  MscWorkT<StdAlloc> work;
  std::string error;
  CHECK(prepare_msc_work(plans, pool, job_ptrs, plane_ptrs, row_base, fib_base, max_rows, work, &error));
  CHECK(work.nframes == nf && work.jobs.size() == nf && work.meta.size() == nf);
  CHECK(work.header_stride % 16 == 0 && work.headers.size() == nf * static_cast<size_t>(work.header_stride));
  // frame records are stream-major and carry the stream's own jobs
  size_t f = 0;
  for (int b = 0; b < nstreams; ++b)
    for (const EtiJob& j : jobs[static_cast<size_t>(b)]) {
      CHECK(work.jobs[f].stream == b && work.jobs[f].cif == j.first_cif);
      CHECK(work.meta[f].header_len == j.header_len && work.meta[f].fib_block == fib_base[static_cast<size_t>(b)] + j.first_cif);
      CHECK(std::memcmp(work.headers.data() + f * static_cast<size_t>(work.header_stride), jobs[static_cast<size_t>(b)].header(j), static_cast<size_t>(j.header_len)) == 0);
      CHECK(12 + 96 + work.meta[f].mst_bytes + 8 <= 6144 + 12);
      ++f;
    }
  // wave-groups: longest first, lanes name valid frames, every (frame, sub-channel of its layout) exactly once
  const auto& batch = work.batch;
  CHECK(batch.job_ids.size() % 64 == 0 && !batch.groups.empty());
  std::vector<int> decoded_bytes(nf, 0);
  int prev_steps = 1 << 30;
  for (const WaveGroup& g : batch.groups) {
    CHECK(g.nsteps <= prev_steps);
    prev_steps = g.nsteps;
    CHECK(g.count >= 1 && g.count <= 64 && g.first >= 0 && static_cast<size_t>(g.first + g.count) <= batch.job_ids.size());
    const CodewordPlan& p = plans[g.plan];
    CHECK(p.nsteps == g.nsteps && p.out_bytes == (p.nsteps - 6) / 8);
    for (int l = 0; l < g.count; ++l) {
      const int id = batch.job_ids[static_cast<size_t>(g.first + l)];
      CHECK(id >= 0 && static_cast<size_t>(id) < nf);
      CHECK(p.out_offset >= work.meta[static_cast<size_t>(id)].header_len + 96);
      CHECK(p.out_offset + p.out_bytes <= work.meta[static_cast<size_t>(id)].header_len + 96 + work.meta[static_cast<size_t>(id)].mst_bytes);
      decoded_bytes[static_cast<size_t>(id)] += (p.out_bytes + 7) & 0xfff8;
    }
  }
  for (size_t i = 0; i < nf; ++i) CHECK(decoded_bytes[i] == work.meta[i].mst_bytes);   // the sub-channels tile the MST exactly (misc.c:259-260)
  // slices partition the groups; record rows fit the cap (a single group may exceed it on its own)
  CHECK(batch.slice_start.front() == 0 && batch.slice_start.back() == static_cast<int>(batch.groups.size()));
  for (size_t s = 0; s + 1 < batch.slice_start.size(); ++s) {
    CHECK(batch.slice_start[s] < batch.slice_start[s + 1]);
    int64_t rows = 0;
    for (int g = batch.slice_start[s]; g < batch.slice_start[s + 1]; ++g) {
      CHECK(batch.groups[static_cast<size_t>(g)].dec_base == rows);
      rows += (batch.groups[static_cast<size_t>(g)].nsteps + 7) / 8 * 8;
    }
    CHECK(rows <= batch.max_dec_rows);
    CHECK(rows <= max_rows || batch.slice_start[s + 1] - batch.slice_start[s] == 1);
  }
}

static void test_control_and_worklist()
{
  control_and_worklist(24, 30, int64_t(48) << 20, 0);
  control_and_worklist(9, 22, 20000, 1);                 // a small record cap: many slices
  // several engines' host sides at once in one process (dabhip_multi: one lane + one pool per slice)
  std::vector<std::unique_ptr<AsyncLane>> lanes;
  for (int i = 0; i < 4; ++i) lanes.emplace_back(new AsyncLane());
  for (int i = 0; i < 4; ++i) lanes[static_cast<size_t>(i)]->post([i]() { control_and_worklist(10 + i, 24, int64_t(1) << 20, 10u * static_cast<unsigned>(i)); });
  for (auto& l : lanes) l->wait();
}

// byte-level replay of what sdr_read_fifo does to the 393216-byte frame buffer (sdr_fifo.c:43-61), on a stream whose byte at
// offset x is the tag x itself (64-bit), so a view can be compared position by position
static void test_fifo_views()
{
  std::mt19937 rng(7);
  for (int trial = 0; trial < 30; ++trial) {
    StreamState st;
    std::memset(&st, 0, sizeof st);
    fifo_reset(st);
    std::vector<int64_t> buffer(kTfBytes, -1);            // stream offset held at each buffer position (-1: calloc'ed zero)
    std::vector<int64_t> tail(kTailBytes, -1);            // the last kTailBytes as K1 carries them (here: tags instead of bytes), updated by read_source
    int64_t rd = 0, fed = 0;
    int most_segments = 0;
    for (int call = 0; call < 160; ++call) {
      const int r = static_cast<int>(rng() % 10);
      st.coarse_timeshift = r == 0 ? static_cast<int>(rng() % 380000) : 0;               // a coarse resync now and then
      // trial >= 15: a receiver clock that runs fast -- ever larger negative shifts for a long stretch, then ever smaller ones: every read short,
      // each shorter (or longer) than the one before.  As views these nested once per read (more than kMaxSeg: the engine used to give up).
      if (trial >= 15) st.fine_timeshift = -2 * (call < 70 ? call + 1 : 161 - call) - (r < 2 ? 2 * static_cast<int>(rng() % 5) : 0);
      else st.fine_timeshift = r < 7 ? static_cast<int>(rng() % 61) - 30 : -2 * static_cast<int>(rng() % 768);      // >= (768 - 1536) * 2, sdr_sync.c:197-201
      const int chunk = trial % 3 == 2 ? 2 * static_cast<int>(rng() % (kChunkBytes / 2 + 1)) : kChunkBytes;   // input_buffer_len: any even length
      const int shift = st.coarse_timeshift + st.fine_timeshift;
      fed += chunk;
      int64_t count = fed - rd;
      bool read = false;
      if (count >= 3 * kTfSamples) {
        read = true;
        if (shift > 0) {
          for (int p = 0; p < shift && p < kTfBytes && p < count; ++p) buffer[static_cast<size_t>(p)] = rd + p;      // the skipped bytes pass through the buffer
          rd += shift;
          count -= shift;
          const int len = count < kTfBytes ? static_cast<int>(count) : kTfBytes;
          for (int p = 0; p < len; ++p) buffer[static_cast<size_t>(p)] = rd + p;
          rd += len;
        } else {
          const int len = kTfBytes + shift;
          for (int p = 0; p < len; ++p) buffer[static_cast<size_t>(p)] = rd + p;
          rd += len;
        }
      }
      const FifoCall c = fifo_call(st, chunk);
      CHECK(!st.overflow);
      CHECK((c.status != 0) == read);
      CHECK(c.fifo_count == static_cast<int>(fed - rd));
      CHECK(st.consumed == rd && st.fed == fed);
      if (!read) continue;
      CHECK(st.view.nseg >= 1 && st.view.nseg <= kMaxSeg && (c.fresh == 1 || c.fresh == 2));
      most_segments = std::max(most_segments, st.view.nseg);
      for (int p = kTailStart; p < kTfBytes; ++p) {                                       // the tail bytes by the kernel's rule ...
        const int64_t src = read_source(st.view, c.fresh, p);
        if (src >= 0) tail[static_cast<size_t>(p - kTailStart)] = src;
        CHECK(buffer[static_cast<size_t>(p)] == tail[static_cast<size_t>(p - kTailStart)]);
      }
      int lo = 0;                                                                          // ... the views everything below them
      for (int i = 0; i < st.view.nseg && lo < kTailStart; ++i) {
        CHECK(st.view.seg_end[i] > lo || (i == 0 && st.view.seg_end[0] >= 0));
        const int hi = std::min(st.view.seg_end[i], kTailStart);
        for (int p = lo; p < hi; p += 997)                                                // sampled positions plus both ends
          CHECK(buffer[static_cast<size_t>(p)] == (st.view.seg_src[i] < 0 ? -1 : st.view.seg_src[i] + p));
        if (hi > lo) {
          const int p = hi - 1;
          CHECK(buffer[static_cast<size_t>(p)] == (st.view.seg_src[i] < 0 ? -1 : st.view.seg_src[i] + p));
        }
        lo = st.view.seg_end[i];
      }
      CHECK(lo >= kTailStart);
      // whatever the views still refer to is recent: nothing pins the stream's past (Engine::stream_need_from)
      if (shift <= 0 && shift >= -kTailBytes)
        for (int i = 0; i < st.view.nseg; ++i) CHECK(st.view.seg_src[i] < 0 || i == 0);
    }
    CHECK(most_segments <= 6);
  }
}

static void test_synth()
{
  for (int preset = 0; preset < 2; ++preset) {
    dabhip_synth_cfg cfg;
    CHECK(dabhip_synth_preset(preset, &cfg) == 0);
    cfg.seed = 5 + static_cast<uint64_t>(preset);
    cfg.skip_samples = preset ? 4321 : 0;
    cfg.snr_db = preset ? 9.0 : 1000.0;
    cfg.cfo_hz = preset ? 123.0 : 0.0;
    const size_t n = dabhip_synth_bytes(&cfg, 2);
    std::vector<uint8_t> iq(n + 64, 0xAA);
    CHECK(dabhip_synth_generate(&cfg, 2, iq.data(), n) == static_cast<int64_t>(n));
    for (size_t i = n; i < n + 64; ++i) CHECK(iq[i] == 0xAA);
    CHECK(dabhip_synth_generate(&cfg, 2, iq.data(), n - 1) < 0);                           // too small a buffer is refused
    std::vector<uint8_t> pay(4096);
    for (int k = 0; k < cfg.nsub; ++k) CHECK(dabhip_synth_payload(&cfg, 3, k, pay.data(), static_cast<int>(pay.size())) > 0);
  }
}

int main(int argc, char** argv)
{
  const char* only = argc > 1 ? argv[1] : "";
  struct { const char* name; void (*fn)(); } tests[] = {
      {"pool", test_pool_and_lane}, {"worklist", test_control_and_worklist}, {"fifo", test_fifo_views}, {"synth", test_synth}};
  for (const auto& t : tests) {
    if (*only && std::strcmp(only, t.name) != 0) continue;
    t.fn();
    std::printf("ok %s\n", t.name);
  }
  return 0;
}
