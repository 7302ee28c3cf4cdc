// dab2eti_hip.cpp — file-replay front door with the stdout contract of the reference's dab2eti:
// every decoded ETI(NI) frame is written to fd 1 as one 6144-byte record (dab2eti.c:132-135).
//
//   dab2eti-hip capture.cu8 [more.cu8 ...] > ensemble.eti
//
// Each file is one 2.048 Msps cu8 IQ capture (I at even bytes, Q at odd bytes), replayed in
// 262,144-byte calls exactly as librtlsdr would deliver it (dab2eti.c:117-130,238), without tuner
// feedback (a file has no tuner).  Several files are decoded as one batch of
// independent ensembles; their frames are emitted file by file.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
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
}  // namespace

int main(int argc, char** argv)
{
  if (argc < 2) {
    std::fprintf(stderr, "Usage: dab2eti-hip capture.cu8 [more.cu8 ...] > out.eti\n");
    return 1;
  }
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
  dabhip_engine* e = dabhip_engine_create(0);
  if (!e) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  std::vector<const uint8_t*> ptrs;
  std::vector<size_t> sizes;
  for (const auto& b : files) { ptrs.push_back(b.data()); sizes.push_back(b.size()); }
  const int64_t n = dabhip_engine_decode(e, ptrs.data(), sizes.data(), static_cast<int>(ptrs.size()), 0);
  if (n < 0) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  for (size_t b = 0; b < files.size(); ++b)
    std::fprintf(stderr, "%s: %lld ETI frames\n", argv[b + 1], static_cast<long long>(dabhip_engine_eti_count(e, static_cast<int>(b))));
  if (dabhip_engine_eti_drain(e, to_stdout, nullptr) != n) { std::fprintf(stderr, "dab2eti-hip: %s\n", dabhip_last_error()); return 2; }
  dabhip_engine_destroy(e);
  return 0;
}
