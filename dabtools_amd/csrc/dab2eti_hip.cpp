// dab2eti_hip.cpp — file-replay front door with the stdout contract of the reference's dab2eti:
// every decoded ETI(NI) frame is written to fd 1 as one 6144-byte record (dab2eti.c:132-135).
//
//   dab2eti-hip capture.cu8 [more.cu8 ...] > ensemble.eti
//   dab2eti-hip --devices 0-7 cap0000.cu8 ... cap2047.cu8 > all.eti     (files dealt to the devices in contiguous slices)
//
//   rtl_sdr -f 220352000 -s 2048000 - | dab2eti-hip - > ensemble.eti         (streaming: "-" = stdin)
//   dab2eti-hip --stream [--segment-calls N] huge.cu8 > ensemble.eti
//
// Streaming mode (any input "-", or --stream) decodes unbounded input in segments of N 262,144-byte calls (default
// 64 = 16 MiB) through a dabhip_stream session: a reader thread fills page-locked buffers while the GPU decodes an earlier one and the
// one in between uploads (dabhip_stream_prefetch), frames leave as soon as their segment is done, memory stays bounded, output bytes are those of the one-shot mode.
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
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int run_streaming(const std::vector<const char*>& names, size_t seg_bytes, bool afc, bool soft, const std::vector<int32_t>& subch, int device)
{
  const double t_start = now_s();
  const int n = static_cast<int>(names.size());
  std::vector<FILE*> in(n);
  for (int i = 0; i < n; ++i) {
    in[i] = std::strcmp(names[i], "-") == 0 ? stdin : std::fopen(names[i], "rb");
    if (!in[i]) { std::perror(names[i]); return 1; }
  }
  dabhip_stream* s = dabhip_stream_create(device, n);
  if (!s) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  if (afc) dabhip_stream_set_afc(s, 1);
  if (soft) dabhip_stream_set_soft(s, 1);
  if (!subch.empty()) dabhip_stream_set_subchannels(s, subch.data(), static_cast<int>(subch.size()));
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
        if (dabhip_stream_eti_fetch_wait(s) != 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); std::_Exit(2); }
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
  std::vector<double> seg_done;                  // when each segment's feed returned
  std::vector<long long> seg_frames;
  for (int k = 0;; k = (k + 1) % kBufs) {
    int state, next_state;
    const int kn = (k + 1) % kBufs;
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled[k] != 0; }); state = filled[k]; next_state = filled[kn]; }
    // segments are handed over in order: this one first (unless it went up during the previous decode already) ...
    if (!prefetched[k]) {
      if (dabhip_stream_prefetch(s, ptrs[k].data(), got[k].data(), 0) != 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); rc = 2; }
      prefetched[k] = true;
    }
    // ... and when the segment after it is already in memory (file replay), its upload runs while this one decodes
    if (!rc && state != 2 && next_state != 0 && !prefetched[kn]) {
      if (dabhip_stream_prefetch(s, ptrs[kn].data(), got[kn].data(), 0) != 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); rc = 2; }
      prefetched[kn] = true;
    }
    const int64_t frames = rc ? -1 : dabhip_stream_feed(s, ptrs[k].data(), got[k].data(), 0);
    prefetched[k] = false;
    if (frames < 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); rc = 2; }
    if (!rc) {
      if (frames > out_frames) { std::fprintf(stderr, "dab2eti-hip: a segment produced more frames than planned for\n"); rc = 2; }
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return out_free[o]; }); out_free[o] = false; }
      if (!rc && frames > 0 && dabhip_stream_eti_fetch(s, out[o], frames) != frames) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); rc = 2; }
      for (int i = 0; i < n && rc == 0; ++i) total[i] += dabhip_stream_eti_count(s, i);
      seg_done.push_back(now_s());
      seg_frames.push_back(frames);
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
  for (int i = 0; i < n; ++i) std::fprintf(stderr, "%s: %lld ETI frames\n", names[i], total[i]);
  if (g_stats) {
    // steady state: from the first segment that produced frames to the last, the frames of the segments after that first one
    size_t first = 0;
    while (first < seg_frames.size() && seg_frames[first] == 0) ++first;
    long long steady_frames = 0, all = 0;
    for (size_t i = 0; i < seg_frames.size(); ++i) { all += seg_frames[i]; if (i > first) steady_frames += seg_frames[i]; }
    const double steady_s = first < seg_done.size() ? t_written - seg_done[first] : 0.0;
    std::fprintf(stderr, "{\"mode\": \"stream\", \"streams\": %d, \"segments\": %zu, \"segment_bytes_per_stream\": %zu, \"setup_s\": %.4f, \"run_s\": %.4f, \"eti_frames\": %lld, "
                         "\"steady_frames\": %lld, \"steady_s\": %.4f, \"steady_frames_per_s\": %.1f}\n",
                 n, seg_frames.size(), seg_bytes, t_ready - t_start, t_written - t_ready, all, steady_frames, steady_s, steady_s > 0 ? steady_frames / steady_s : 0.0);
  }
  for (auto& b : buf) dabhip_host_free(b);
  for (auto& ob : out) dabhip_host_free(ob);
  dabhip_stream_destroy(s);
  return 0;
}
}  // namespace

int main(int argc, char** argv)
{
  bool streaming = false, afc = false, soft = false;
  size_t seg_calls = 64;
  std::vector<int32_t> subch;                  // --subch 3,7: decode and carry only these SubChIds (TODO.md:28-31)
  std::vector<int> devices;                    // --devices 0-7 | 0,2,3 | 0,0 (an entry per slice; repeats allowed): dabhip_multi
  std::vector<const char*> names;
  for (int i = 1; i < argc; ++i) {
    if (std::strcmp(argv[i], "--stream") == 0) streaming = true;
    else if (std::strcmp(argv[i], "--stats") == 0) g_stats = true;
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
    std::fprintf(stderr, "Usage: dab2eti-hip [--stream] [--segment-calls N] [--afc] [--soft] [--subch ID[,ID...]] [--devices A-B|A,B,...] capture.cu8|- [more.cu8 ...] > out.eti\n");
    return 1;
  }
  if (streaming && devices.size() > 1) { std::fprintf(stderr, "dab2eti-hip: --devices with several entries applies to the batch mode (files), not to --stream / stdin\n"); return 1; }
  if (streaming) return run_streaming(names, seg_calls * 262144, afc, soft, subch, devices.empty() ? 0 : devices[0]);
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
  for (size_t b = 0; b < files.size(); ++b)
    std::fprintf(stderr, "%s: %lld ETI frames\n", argv[b + 1], static_cast<long long>(dabhip_engine_eti_count(e, static_cast<int>(b))));
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
