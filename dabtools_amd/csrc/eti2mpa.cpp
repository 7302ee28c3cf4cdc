// eti2mpa.cpp — host-only counterpart of the reference's eti2mpa (eti2mpa.c:16-68): reads ETI(NI) frames from stdin
// and writes the payload of one sub-channel to stdout, closing the pipe  dab2eti-hip capture.cu8 | eti2mpa N > audio.mp2
// Unlike the reference it re-locates the sub-channel in every frame (the multiplex may be reconfigured) and checks the
// frame's sync bytes.
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace {
bool read_exact(uint8_t* p, size_t n)
{
  size_t got = 0;
  while (got < n) {
    const ssize_t r = read(0, p + got, n - got);
    if (r <= 0) return false;
    got += static_cast<size_t>(r);
  }
  return true;
}
}  // namespace

int main(int argc, char** argv)
{
  if (argc != 2) {
    std::fprintf(stderr, "Usage: eti2mpa N   (N = sub-channel id; ETI on stdin, sub-channel bytes on stdout)\n");
    return 1;
  }
  const int want = std::atoi(argv[1]);
  uint8_t buf[6144];
  long frames = 0, missing = 0;
  while (read_exact(buf, sizeof buf)) {
    ++frames;
    const bool odd = buf[4] & 1;                       // FSYNC alternates with the frame count (misc.c:163-171)
    if (buf[0] != 0xff || buf[1] != (odd ? 0xf8 : 0x07) || buf[2] != (odd ? 0xc5 : 0x3a) || buf[3] != (odd ? 0x49 : 0xb6)) {
      std::fprintf(stderr, "eti2mpa: frame %ld: bad sync\n", frames);
      return 2;
    }
    const int ficf = buf[5] >> 7, nst = buf[5] & 0x7f;
    int offset = 12 + 4 * nst + ficf * 96, length = -1;
    for (int i = 0; i < nst; ++i) {
      const int scid = buf[8 + 4 * i] >> 2;
      const int stl = ((buf[8 + 4 * i + 2] & 3) << 8) | buf[8 + 4 * i + 3];
      if (scid == want) { length = stl * 8; break; }
      offset += stl * 8;
    }
    if (length < 0) { ++missing; continue; }
    size_t done = 0;
    while (done < static_cast<size_t>(length)) {
      const ssize_t w = write(1, buf + offset + done, static_cast<size_t>(length) - done);
      if (w <= 0) return 3;
      done += static_cast<size_t>(w);
    }
  }
  std::fprintf(stderr, "eti2mpa: %ld frames, sub-channel %d absent in %ld\n", frames, want, missing);
  return frames > 0 && missing == frames ? 4 : 0;
}
