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
#include <unistd.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/dabhip.h"

namespace {
void to_stdout(const uint8_t* eti, int /*stream*/, void* /*user*/)
{
  size_t done = 0;
  while (done < DABHIP_ETI_BYTES) {
    const ssize_t n = write(1, eti + done, DABHIP_ETI_BYTES - done);
    if (n <= 0) { std::perror("dab2eti-hip: write"); std::exit(1); }
    done += static_cast<size_t>(n);
  }
}

// streaming mode: double-buffered page-locked segments, one reader thread
int run_streaming(const std::vector<const char*>& names, size_t seg_bytes, bool afc, bool soft, const std::vector<int32_t>& subch, int device)
{
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
  // three page-locked buffers: one being decoded, one uploading (dabhip_stream_prefetch), one being read into
  constexpr int kBufs = 3;
  uint8_t* buf[kBufs];
  for (auto& b : buf)
    if (!(b = static_cast<uint8_t*>(dabhip_host_alloc(seg_bytes * n)))) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  std::vector<size_t> got[kBufs];
  for (auto& g : got) g.assign(n, 0);
  std::mutex mu;
  std::condition_variable cv;
  int filled[kBufs] = {0, 0, 0};   // 0 = free, 1 = full, 2 = full and last
  std::thread reader([&]() {
    std::vector<bool> eof(n, false);
    for (int k = 0;; k = (k + 1) % kBufs) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled[k] == 0; }); }
      bool all_eof = true;
      for (int i = 0; i < n; ++i) {
        size_t done = 0;
        while (!eof[i] && done < seg_bytes) {
          const size_t r = std::fread(buf[k] + seg_bytes * i + done, 1, seg_bytes - done, in[i]);
          if (r == 0) eof[i] = true;
          done += r;
        }
        got[k][i] = done;
        all_eof = all_eof && eof[i];
      }
      { std::lock_guard<std::mutex> lk(mu); filled[k] = all_eof ? 2 : 1; }
      cv.notify_all();
      if (all_eof) return;
    }
  });
  std::vector<long long> total(n, 0);
  std::vector<std::vector<const uint8_t*>> ptrs(kBufs, std::vector<const uint8_t*>(n));
  for (int k = 0; k < kBufs; ++k)
    for (int i = 0; i < n; ++i) ptrs[k][i] = buf[k] + seg_bytes * i;
  bool prefetched[kBufs] = {false, false, false};
  int rc = 0;
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
    if (frames < 0 || dabhip_stream_eti_drain(s, to_stdout, nullptr) != frames) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); rc = 2; }
    for (int i = 0; i < n && rc == 0; ++i) total[i] += dabhip_stream_eti_count(s, i);
    { std::lock_guard<std::mutex> lk(mu); filled[k] = 0; }
    cv.notify_all();
    if (state == 2 || rc) break;
  }
  if (rc) std::_Exit(rc);          // the reader may be blocked in fread
  reader.join();
  for (int i = 0; i < n; ++i) std::fprintf(stderr, "%s: %lld ETI frames\n", names[i], total[i]);
  for (auto& b : buf) dabhip_host_free(b);
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
  std::vector<std::vector<uint8_t>> files;
  for (int i = 1; i < argc; ++i) {
    FILE* f = std::fopen(argv[i], "rb");
    if (!f) { std::perror(argv[i]); return 1; }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(static_cast<size_t>(n));
    if (n > 0 && std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fprintf(stderr, "%s: short read\n", argv[i]); return 1; }
    std::fclose(f);
    files.push_back(std::move(buf));
  }
  std::vector<const uint8_t*> ptrs;
  std::vector<size_t> sizes;
  for (const auto& b : files) { ptrs.push_back(b.data()); sizes.push_back(b.size()); }
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
    dabhip_multi_destroy(m);
    return 0;
  }
  dabhip_engine* e = dabhip_engine_create(0);
  if (!e) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  if (afc) dabhip_engine_set_afc(e, 1);
  if (soft) dabhip_engine_set_soft(e, 1);
  if (!subch.empty()) dabhip_engine_set_subchannels(e, subch.data(), static_cast<int>(subch.size()));
  const int64_t n = dabhip_engine_decode(e, ptrs.data(), sizes.data(), static_cast<int>(ptrs.size()), 0);
  if (n < 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  for (size_t b = 0; b < files.size(); ++b)
    std::fprintf(stderr, "%s: %lld ETI frames\n", argv[b + 1], static_cast<long long>(dabhip_engine_eti_count(e, static_cast<int>(b))));
  if (dabhip_engine_eti_drain(e, to_stdout, nullptr) != n) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  dabhip_engine_destroy(e);
  return 0;
}
