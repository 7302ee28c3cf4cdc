// dab2eti_hip.cpp — file-replay front door with the stdout contract of the reference's dab2eti:
// every decoded ETI(NI) frame is written to fd 1 as one 6144-byte record (dab2eti.c:132-135).
//
//   dab2eti-hip capture.cu8 [more.cu8 ...] > ensemble.eti
//   dab2eti-hip --devices 0-7 cap0000.cu8 ... cap2047.cu8 > all.eti     (files dealt to the devices in contiguous slices)
//
//   rtl_sdr -f 220352000 -s 2048000 - | dab2eti-hip - > ensemble.eti         (streaming: "-" = stdin)
//   dab2eti-hip --stream [--segment-calls N] huge.cu8 > ensemble.eti
//   dab2eti-hip --stream --devices 0-7 cap0000.cu8 ... cap2047.cu8 > all.eti  (sessions on several devices: dabhip_multi_stream)
//
// Streaming mode (any input "-", or --stream) decodes unbounded input in segments of N 262,144-byte calls through a dabhip_stream session (on several
// devices: a dabhip_multi_stream, the inputs dealt to the devices in contiguous slices): reader threads fill page-locked buffers while the GPU decodes an
// earlier one and the one in between uploads (dabhip_stream_prefetch), frames leave as soon as their segment is done, memory stays bounded, output bytes are
// those of the one-shot mode.  N defaults to 64 (16 MiB per input and segment: throughput) when every input is a regular file and to 2 (0.5 MiB = 128 ms of
// signal: latency) when one of them is a pipe, a FIFO or a terminal -- a live receiver's frames leave within a fraction of a second of their samples.
//
// stderr carries what the reference's operator sees (dab.c:51,57,78-82): "Locked", "Lock lost, resetting ringbuffer" and the one-time ensemble dump, per
// input (prefixed with the input's name when there are several); --quiet turns them off.
//
// Each file is one 2.048 Msps cu8 IQ capture (I at even bytes, Q at odd bytes), replayed in
// 262,144-byte calls exactly as librtlsdr would deliver it (dab2eti.c:117-130,238), without tuner
// feedback (a file has no tuner).  Several files are decoded as one batch of
// independent ensembles; their frames are emitted file by file.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/dabhip.h"

namespace {
void write_all(const uint8_t* p, size_t n)
{
  while (n) {
    const ssize_t w = write(1, p, n);
    if (w <= 0) { std::perror("dab2eti-hip: write"); std::_Exit(1); }
    p += w;
    n -= static_cast<size_t>(w);
  }
}
// frame sink of the drain calls: the same bytes as one write(1, eti, 6144) per frame (dab2eti.c:132-135), 512 frames to a system call
std::vector<uint8_t> g_out;
void flush_stdout()
{
  write_all(g_out.data(), g_out.size());
  g_out.clear();
}
void to_stdout(const uint8_t* eti, int /*stream*/, void* /*user*/)
{
  g_out.insert(g_out.end(), eti, eti + DABHIP_ETI_BYTES);
  if (g_out.size() >= size_t(512) * DABHIP_ETI_BYTES) flush_stdout();
}
// a capture file mapped read-only: no copy on the way in (the engine's staging pool reads the page cache directly)
struct Mapped {
  const uint8_t* p = nullptr;
  size_t n = 0;
};
bool map_file(const char* name, Mapped* m)
{
  const int fd = open(name, O_RDONLY);
  if (fd < 0) { std::perror(name); return false; }
  struct stat st;
  if (fstat(fd, &st) != 0) { std::perror(name); close(fd); return false; }
  m->n = static_cast<size_t>(st.st_size);
  if (m->n) {
    void* p = mmap(nullptr, m->n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) { std::perror(name); close(fd); return false; }
    (void)madvise(p, m->n, MADV_SEQUENTIAL);
    m->p = static_cast<const uint8_t*>(p);
  }
  close(fd);
  return true;
}

// Streaming mode: three page-locked input buffers (one being decoded, one uploading, one being read into) filled by up to 16 reader threads
// (each owns some of the inputs: one thread's read() of cached files is an order of magnitude below the PCIe rate the decode sustains), and
// two page-locked output buffers: the frames of a segment come back in ONE asynchronous copy (dabhip_stream_eti_fetch) that runs beside the next
// segment's decode, and a writer thread puts them on fd 1 in large writes -- the same bytes in the same order as one 6144-byte write per frame
// (dab2eti.c:132-135), stream by stream within a segment.
bool g_stats = false;                            // --stats: phase times on stderr (one JSON line), for tools/cli_throughput.py
bool g_quiet = false;                            // --quiet: no operator messages
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the reference's operator messages of one input (dabhip_*_log text) on stderr: as they are for a single input, every line prefixed otherwise
void print_log(const char* text, const char* name, bool prefix)
{
  if (!*text) return;
  if (!prefix) { std::fputs(text, stderr); return; }
  for (const char* p = text; *p;) {
    const char* nl = std::strchr(p, '\n');
    const size_t len = nl ? static_cast<size_t>(nl - p) : std::strlen(p);
    std::fprintf(stderr, "%s: %.*s\n", name, static_cast<int>(len), p);
    p += len + (nl ? 1 : 0);
  }
}

// one session on one device, or one per device behind dabhip_multi_stream: the same calls either way
struct Session {
  dabhip_stream* one = nullptr;
  dabhip_multi_stream* many = nullptr;
  bool create(const std::vector<int>& devices, int n)
  {
    if (devices.size() > 1) many = dabhip_multi_stream_create(devices.data(), static_cast<int>(devices.size()), n);
    else one = dabhip_stream_create(devices.empty() ? 0 : devices[0], n);
    return one || many;
  }
  void destroy() { if (one) dabhip_stream_destroy(one); if (many) dabhip_multi_stream_destroy(many); one = nullptr; many = nullptr; }
  void set_afc() { if (one) dabhip_stream_set_afc(one, 1); else dabhip_multi_stream_set_afc(many, 1); }
  void set_soft() { if (one) dabhip_stream_set_soft(one, 1); else dabhip_multi_stream_set_soft(many, 1); }
  void set_subchannels(const std::vector<int32_t>& ids)
  {
    if (one) dabhip_stream_set_subchannels(one, ids.data(), static_cast<int>(ids.size()));
    else dabhip_multi_stream_set_subchannels(many, ids.data(), static_cast<int>(ids.size()));
  }
  int prefetch(const uint8_t* const* iq, const size_t* nbytes) { return one ? dabhip_stream_prefetch(one, iq, nbytes, 0) : dabhip_multi_stream_prefetch(many, iq, nbytes, 0); }
  int64_t feed(const uint8_t* const* iq, const size_t* nbytes) { return one ? dabhip_stream_feed(one, iq, nbytes, 0) : dabhip_multi_stream_feed(many, iq, nbytes, 0); }
  int64_t eti_fetch(uint8_t* dst, int64_t frames) { return one ? dabhip_stream_eti_fetch(one, dst, frames) : dabhip_multi_stream_eti_fetch(many, dst, frames); }
  int eti_fetch_wait() { return one ? dabhip_stream_eti_fetch_wait(one) : dabhip_multi_stream_eti_fetch_wait(many); }
  int64_t eti_count(int i) { return one ? dabhip_stream_eti_count(one, i) : dabhip_multi_stream_eti_count(many, i); }
  int64_t log(int i, char* buf, int64_t cap) { return one ? dabhip_stream_log(one, i, buf, cap) : dabhip_multi_stream_log_of(many, i, buf, cap); }
  int device_of(int i)
  {
    int dev = -1;
    if (many) dabhip_multi_stream_slice_of(many, i, &dev, nullptr, nullptr);
    return dev;
  }
};

int run_streaming(const std::vector<const char*>& names, size_t seg_bytes, bool afc, bool soft, const std::vector<int32_t>& subch, const std::vector<int>& devices)
{
  const double t_start = now_s();
  const int n = static_cast<int>(names.size());
  std::vector<FILE*> in(n);
  for (int i = 0; i < n; ++i) {
    in[i] = std::strcmp(names[i], "-") == 0 ? stdin : std::fopen(names[i], "rb");
    if (!in[i]) { std::perror(names[i]); return 1; }
  }
  Session ses;
  Session* s = &ses;
  if (!ses.create(devices, n)) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  if (afc) ses.set_afc();
  if (soft) ses.set_soft();
  if (!subch.empty()) ses.set_subchannels(subch);
  constexpr int kBufs = 3, kOut = 2;
  uint8_t* buf[kBufs];
  // frames a segment can yield per stream: 4 per transmission frame it completes, plus what the backlog of the segment before adds
  const int64_t out_frames = static_cast<int64_t>(n) * (4 * static_cast<int64_t>(seg_bytes / DABHIP_TF_BYTES + 2));
  uint8_t* out[kOut];
  {
    // page-locking gigabytes takes a noticeable part of a second: the five buffers at once
    std::vector<std::thread> alloc;
    for (auto& b : buf) alloc.emplace_back([&b, seg_bytes, n]() { b = static_cast<uint8_t*>(dabhip_host_alloc(seg_bytes * n)); });
    for (auto& ob : out) alloc.emplace_back([&ob, out_frames]() { ob = static_cast<uint8_t*>(dabhip_host_alloc(static_cast<size_t>(out_frames) * DABHIP_ETI_BYTES)); });
    for (auto& t : alloc) t.join();
    for (auto* b : buf) if (!b) { std::fprintf(stderr, "dab2eti-hip: page-locked input buffer: %s\n", dabhip_last_error()); return 2; }
    for (auto* ob : out) if (!ob) { std::fprintf(stderr, "dab2eti-hip: page-locked output buffer: %s\n", dabhip_last_error()); return 2; }
  }
  std::vector<size_t> got[kBufs];
  for (auto& g : got) g.assign(n, 0);
  std::mutex mu;
  std::condition_variable cv;
  int filled[kBufs] = {0, 0, 0};   // 0 = free, 1 = full, 2 = full and last
  // A buffer is filled once per generation (the main loop opens the next one when it has consumed the buffer) by ALL readers, each reading its own
  // inputs' parts; the reader that finishes a buffer publishes it -- as the last one when every reader found all of its inputs at their end.
  int gen[kBufs] = {0, 0, 0}, readers_done[kBufs] = {0, 0, 0}, readers_at_eof[kBufs] = {0, 0, 0};
  bool stop = false;
  const int nreaders = std::max(1, std::min(n, 16));
  std::vector<std::thread> readers;
  for (int r = 0; r < nreaders; ++r)
    readers.emplace_back([&, r]() {
      std::vector<bool> eof(n, false);
      int my_gen[kBufs] = {0, 0, 0};
      for (int k = 0;; k = (k + 1) % kBufs) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return stop || (filled[k] == 0 && gen[k] == my_gen[k]); });
          if (stop) return;
        }
        bool all_mine_eof = true;
        for (int i = r; i < n; i += nreaders) {
          size_t done = 0;
          while (!eof[i] && done < seg_bytes) {
            const size_t got_now = std::fread(buf[k] + seg_bytes * i + done, 1, seg_bytes - done, in[i]);
            if (got_now == 0) eof[i] = true;
            done += got_now;
          }
          got[k][i] = done;
          all_mine_eof = all_mine_eof && eof[i];
        }
        ++my_gen[k];
        {
          std::lock_guard<std::mutex> lk(mu);
          readers_at_eof[k] += all_mine_eof ? 1 : 0;
          if (++readers_done[k] == nreaders) filled[k] = readers_at_eof[k] == nreaders ? 2 : 1;
        }
        cv.notify_all();
      }
    });
  // writer: the output buffers in order, each once its download has arrived
  struct OutItem { int o; int64_t frames; bool last; };
  std::vector<OutItem> queue;
  bool out_free[kOut] = {true, true};
  std::thread writer([&]() {
    for (;;) {
      OutItem it;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !queue.empty(); }); it = queue.front(); queue.erase(queue.begin()); }
      if (it.frames > 0) {
        if (s->eti_fetch_wait() != 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); std::_Exit(2); }
        write_all(out[it.o], static_cast<size_t>(it.frames) * DABHIP_ETI_BYTES);
      }
      { std::lock_guard<std::mutex> lk(mu); out_free[it.o] = true; }
      cv.notify_all();
      if (it.last) return;
    }
  });
  std::vector<long long> total(n, 0);
  std::vector<std::vector<const uint8_t*>> ptrs(kBufs, std::vector<const uint8_t*>(n));
  for (int k = 0; k < kBufs; ++k)
    for (int i = 0; i < n; ++i) ptrs[k][i] = buf[k] + seg_bytes * i;
  bool prefetched[kBufs] = {false, false, false};
  int rc = 0, o = 0;
  const double t_ready = now_s();
  // running totals for --stats (not a record per segment: a live receiver runs for weeks)
  size_t segments = 0;
  long long all_frames = 0, steady_frames = 0;   // steady state: the frames of the segments AFTER the first one that produced any
  double t_first_frames = -1.0;                  // when that first segment's feed returned
  for (int k = 0;; k = (k + 1) % kBufs) {
    int state, next_state;
    const int kn = (k + 1) % kBufs;
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled[k] != 0; }); state = filled[k]; next_state = filled[kn]; }
    // segments are handed over in order: this one first (unless it went up during the previous decode already) ...
    if (!prefetched[k]) {
      if (s->prefetch(ptrs[k].data(), got[k].data()) != 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); rc = 2; }
      prefetched[k] = true;
    }
    // ... and when the segment after it is already in memory (file replay), its upload runs while this one decodes
    if (!rc && state != 2 && next_state != 0 && !prefetched[kn]) {
      if (s->prefetch(ptrs[kn].data(), got[kn].data()) != 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); rc = 2; }
      prefetched[kn] = true;
    }
    const int64_t frames = rc ? -1 : s->feed(ptrs[k].data(), got[k].data());
    prefetched[k] = false;
    if (frames < 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); rc = 2; }
    if (!rc) {
      if (frames > out_frames) { std::fprintf(stderr, "dab2eti-hip: a segment produced more frames than planned for\n"); rc = 2; }
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return out_free[o]; }); out_free[o] = false; }
      if (!rc && frames > 0 && s->eti_fetch(out[o], frames) != frames) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); rc = 2; }
      for (int i = 0; i < n && rc == 0; ++i) total[i] += s->eti_count(i);
      if (!g_quiet) {                            // the reference's operator messages of this segment (dab.c:51,57,78-82)
        char text[8192];
        for (int i = 0; i < n; ++i)
          if (s->log(i, text, sizeof text) > 0) print_log(text, names[i], n > 1);
      }
      ++segments;
      all_frames += frames;
      if (t_first_frames >= 0) steady_frames += frames;
      else if (frames > 0) t_first_frames = now_s();
    }
    if (rc) std::_Exit(rc);        // readers may be blocked in fread, the writer on its queue
    { std::lock_guard<std::mutex> lk(mu); queue.push_back(OutItem{o, frames, state == 2}); filled[k] = 0; readers_done[k] = 0; readers_at_eof[k] = 0; ++gen[k]; stop = state == 2; }
    cv.notify_all();
    o = (o + 1) % kOut;
    if (state == 2) break;
  }
  writer.join();
  const double t_written = now_s();
  for (auto& t : readers) t.join();
  for (int i = 0; i < n; ++i) {
    if (ses.many) std::fprintf(stderr, "%s: %lld ETI frames (device %d)\n", names[i], total[i], ses.device_of(i));
    else std::fprintf(stderr, "%s: %lld ETI frames\n", names[i], total[i]);
  }
  if (g_stats) {
    // steady state: from the first segment that produced frames to the last, the frames of the segments after that first one
    const double steady_s = t_first_frames >= 0 ? t_written - t_first_frames : 0.0;
    std::fprintf(stderr, "{\"mode\": \"stream\", \"devices\": %zu, \"streams\": %d, \"segments\": %zu, \"segment_bytes_per_stream\": %zu, \"setup_s\": %.4f, \"run_s\": %.4f, \"eti_frames\": %lld, "
                         "\"steady_frames\": %lld, \"steady_s\": %.4f, \"steady_frames_per_s\": %.1f}\n",
                 devices.empty() ? size_t(1) : devices.size(), n, segments, seg_bytes, t_ready - t_start, t_written - t_ready, all_frames, steady_frames, steady_s, steady_s > 0 ? steady_frames / steady_s : 0.0);
  }
  for (auto& b : buf) dabhip_host_free(b);
  for (auto& ob : out) dabhip_host_free(ob);
  ses.destroy();
  return 0;
}
}  // namespace

int main(int argc, char** argv)
{
  bool streaming = false, afc = false, soft = false;
  size_t seg_calls = 0;                        // 0 = default: 64 for regular files, 2 when an input is a pipe / FIFO / terminal
  std::vector<int32_t> subch;                  // --subch 3,7: decode and carry only these SubChIds (TODO.md:28-31)
  std::vector<int> devices;                    // --devices 0-7 | 0,2,3 | 0,0 (an entry per slice; repeats allowed): dabhip_multi
  std::vector<const char*> names;
  for (int i = 1; i < argc; ++i) {
    if (std::strcmp(argv[i], "--stream") == 0) streaming = true;
    else if (std::strcmp(argv[i], "--stats") == 0) g_stats = true;
    else if (std::strcmp(argv[i], "--quiet") == 0) g_quiet = true;
    else if (std::strcmp(argv[i], "--afc") == 0) afc = true;          // software AFC: captures with a carrier offset (no tuner to steer)
    else if (std::strcmp(argv[i], "--soft") == 0) soft = true;        // 4-bit soft decisions (not the reference's hard ones)
    else if (std::strcmp(argv[i], "--subch") == 0 && i + 1 < argc) {
      for (const char* p = argv[++i]; *p;) {
        subch.push_back(static_cast<int32_t>(std::strtol(p, const_cast<char**>(&p), 10)));
        if (*p == ',') ++p;
        else if (*p) { std::fprintf(stderr, "dab2eti-hip: bad --subch list\n"); return 1; }
      }
    }
    else if (std::strcmp(argv[i], "--devices") == 0 && i + 1 < argc) {
      for (const char* p = argv[++i]; *p;) {
        char* end = nullptr;
        const long a = std::strtol(p, &end, 10);
        long b = a;
        if (end == p) { std::fprintf(stderr, "dab2eti-hip: bad --devices list\n"); return 1; }
        p = end;
        if (*p == '-') { b = std::strtol(p + 1, &end, 10); if (end == p + 1 || b < a) { std::fprintf(stderr, "dab2eti-hip: bad --devices range\n"); return 1; } p = end; }
        for (long d = a; d <= b; ++d) devices.push_back(static_cast<int>(d));
        if (*p == ',') ++p;
        else if (*p) { std::fprintf(stderr, "dab2eti-hip: bad --devices list\n"); return 1; }
      }
    }
    else if (std::strcmp(argv[i], "--segment-calls") == 0 && i + 1 < argc) seg_calls = static_cast<size_t>(std::max(1, std::atoi(argv[++i])));
    else { names.push_back(argv[i]); streaming = streaming || std::strcmp(argv[i], "-") == 0; }
  }
  if (names.empty()) {
    std::fprintf(stderr, "Usage: dab2eti-hip [--stream] [--segment-calls N] [--afc] [--soft] [--quiet] [--subch ID[,ID...]] [--devices A-B|A,B,...] capture.cu8|- [more.cu8 ...] > out.eti\n");
    return 1;
  }
  if (streaming) {
    if (seg_calls == 0) {                      // a live source: short segments, so that frames leave as their samples arrive (dab2eti.c:117-130 decodes call by call)
      bool live = false;
      for (const char* name : names) {
        struct stat st;
        if (std::strcmp(name, "-") == 0 ? fstat(0, &st) != 0 || !S_ISREG(st.st_mode) : stat(name, &st) == 0 && !S_ISREG(st.st_mode)) live = true;
      }
      seg_calls = live ? 2 : 64;
    }
    return run_streaming(names, seg_calls * 262144, afc, soft, subch, devices);
  }
  argc = static_cast<int>(names.size()) + 1;
  for (int i = 1; i < argc; ++i) argv[i] = const_cast<char*>(names[i - 1]);
  const double t_start = now_s();
  std::vector<Mapped> files(static_cast<size_t>(argc - 1));
  for (int i = 1; i < argc; ++i)
    if (!map_file(argv[i], &files[static_cast<size_t>(i - 1)])) return 1;
  std::vector<const uint8_t*> ptrs;
  std::vector<size_t> sizes;
  for (const auto& b : files) { ptrs.push_back(b.p); sizes.push_back(b.n); }
  if (!devices.empty()) {
    // several devices (or an explicit one): the files are dealt to them in contiguous slices, all slices decode at once
    dabhip_multi* m = dabhip_multi_create(devices.data(), static_cast<int>(devices.size()));
    if (!m) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
    if (afc) dabhip_multi_set_afc(m, 1);
    if (soft) dabhip_multi_set_soft(m, 1);
    if (!subch.empty()) dabhip_multi_set_subchannels(m, subch.data(), static_cast<int>(subch.size()));
    const int64_t n = dabhip_multi_decode(m, ptrs.data(), sizes.data(), static_cast<int>(ptrs.size()), 0);
    if (n < 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
    for (size_t b = 0; b < files.size(); ++b) {
      int dev = -1;
      dabhip_multi_slice_of(m, static_cast<int>(b), &dev);
      std::fprintf(stderr, "%s: %lld ETI frames (device %d)\n", argv[b + 1], static_cast<long long>(dabhip_multi_eti_count(m, static_cast<int>(b))), dev);
      char text[8192];
      if (!g_quiet && dabhip_multi_stream_log(m, static_cast<int>(b), text, sizeof text) > 0) print_log(text, argv[b + 1], files.size() > 1);
    }
    if (dabhip_multi_eti_drain(m, to_stdout, nullptr) != n) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
    flush_stdout();
    dabhip_multi_destroy(m);
    return 0;
  }
  dabhip_engine* e = dabhip_engine_create(0);
  if (!e) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  if (afc) dabhip_engine_set_afc(e, 1);
  if (soft) dabhip_engine_set_soft(e, 1);
  if (!subch.empty()) dabhip_engine_set_subchannels(e, subch.data(), static_cast<int>(subch.size()));
  const double t_ready = now_s();
  const int64_t n = dabhip_engine_decode(e, ptrs.data(), sizes.data(), static_cast<int>(ptrs.size()), 0);
  if (n < 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  const double t_decoded = now_s();
  for (size_t b = 0; b < files.size(); ++b) {
    std::fprintf(stderr, "%s: %lld ETI frames\n", argv[b + 1], static_cast<long long>(dabhip_engine_eti_count(e, static_cast<int>(b))));
    char text[8192];
    if (!g_quiet && dabhip_engine_stream_log(e, static_cast<int>(b), text, sizeof text) > 0) print_log(text, argv[b + 1], files.size() > 1);
  }
  // all frames, file by file in emission order, in ONE download into page-locked memory and one run of large writes
  if (n > 0) {
    uint8_t* host = static_cast<uint8_t*>(dabhip_host_alloc(static_cast<size_t>(n) * DABHIP_ETI_BYTES));
    if (host && dabhip_engine_eti_fetch(e, host, n) == n && dabhip_engine_eti_fetch_wait(e) == 0) {
      write_all(host, static_cast<size_t>(n) * DABHIP_ETI_BYTES);
    } else {
      // the one-download path serves one decode lane (DABHIP_LANES=1, the default); an engine of several lanes hands its frames over stream by stream
      if (dabhip_engine_eti_drain(e, to_stdout, nullptr) != n) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
      flush_stdout();
    }
    if (host) dabhip_host_free(host);
  }
  if (g_stats)
    std::fprintf(stderr, "{\"mode\": \"batch\", \"streams\": %zu, \"setup_s\": %.4f, \"upload_decode_s\": %.4f, \"download_write_s\": %.4f, \"eti_frames\": %lld}\n", files.size(),
                 t_ready - t_start, t_decoded - t_ready, now_s() - t_decoded, static_cast<long long>(n));
  dabhip_engine_destroy(e);
  return 0;
}
