// fifo_view.hpp — the byte FIFO of the reference front end in closed form (host + device).
//
// sdr_demod appends every 262144-byte call to a ring (cbWrite, sdr_fifo.c:26-35) and, once 1.5 transmission frames
// are queued, reads ONE frame with sdr_read_fifo(fifo, 393216, shift, buffer) (sdr_fifo.c:43-61, input_sdr.c:36-47),
// shift = coarse_timeshift + fine_timeshift as left by the previous processed frame.  The IQ stream is resident and
// never modified here, so the ring is only a pair of counters and the reference's persistent 393216-byte frame buffer
// is described as a FrameView (device_types.hpp): which stream byte every buffer position holds after the read.
//
//   shift <= 0: the next 393216 + shift stream bytes land in buffer[0 .. 393216 + shift); the tail keeps whatever
//               earlier reads left there (sdr_fifo.c:56-59).  Those last 1536 bytes are carried as BYTES by whoever replays the FIFO
//               (K1's registers, the host replay's array; device_types.hpp: kTailBytes), the views below describe the rest.
//   shift >  0: sdr_read_fifo FIRST copies the `shift` skipped bytes to buffer[0 .. shift) and THEN the `len` frame
//               bytes to buffer[0 .. len) (sdr_fifo.c:49-55).  When the FIFO runs dry (len < shift: a large coarse
//               correction with little backlog) buffer[len .. shift) therefore holds SKIPPED stream bytes, not the
//               older tail.
//
// One function, used by sync_scan_kernel (thread 0 of the stream's workgroup) and by the host-side replay
// dabhip_host_fifo_* that the CPU test-suite compares with the reference's own sdr_fifo.c.
#pragma once

#include <hip/hip_runtime.h>

#include "dab_tables.hpp"
#include "device_types.hpp"

namespace dabhip {

struct FifoCall {
  int status;       // 0 = fewer than 1.5 TF queued: nothing read; 1 = a frame was read
  int do_sync;      // the frame is processed (0 for the first frame read: input_sdr.c:51-55, GAIN_SETTLE_TIME 0)
  int fifo_count;   // sdr->fifo.count after the call
  int fresh;        // leading segments of the new view this read wrote itself: 1, or 2 with the skipped bytes of a dry read
};

// Stream offset of the byte the read described by the first `fresh` segments of `v` leaves at buffer position p, -1 where the read does
// not reach (the position keeps what it held).  The rule K1 updates a stream's tail bytes by (device_types.hpp: kTailBytes).
__host__ __device__ __forceinline__ int64_t read_source(const FrameView& v, int fresh, int p)
{
  if (p < v.seg_end[0]) return v.seg_src[0] + p;
  if (fresh == 2 && p < v.seg_end[1]) return v.seg_src[1] + p;
  return -1;
}

// One sdr_demod call's worth of FIFO work: 262144 bytes appended, at most one frame read with the shift held in `st`.
// chunk: the bytes this call appended (input_buffer_len, input_sdr.c:36-38; 262144 from librtlsdr, dab2eti.c:238).
__host__ __device__ inline FifoCall fifo_call(StreamState& st, int chunk = kChunkBytes)
{
  FifoCall out{0, 0, 0, 0};
  st.fed += chunk;
  int64_t count = st.fed - st.consumed;
  if (count >= 3 * kTfSamples) {
    const int shift = st.coarse_timeshift + st.fine_timeshift;
    const int64_t consumed0 = st.consumed;          // read pointer before this call
    int len, skipped = 0;
    if (shift > 0) {
      st.consumed += shift;
      count -= shift;
      len = count < kTfBytes ? static_cast<int>(count) : kTfBytes;
      skipped = shift;
    } else {
      len = kTfBytes + shift;
    }
    FrameView nv;
    int n = 1;
    nv.seg_end[0] = len;
    nv.seg_src[0] = st.consumed;
    int covered = len;                              // buffer[0 .. covered) was (re)written by this read
    if (skipped > len) {                            // the skipped bytes beyond the short frame stay visible
      nv.seg_end[1] = skipped;
      nv.seg_src[1] = consumed0;                    // buffer[p] = stream[consumed0 + p] for p < shift
      covered = skipped;
      n = 2;
    }
    out.fresh = n;
    // older segments that still show below the tail bytes (which are kept as bytes, not as views: device_types.hpp)
    for (int i = 0; i < st.view.nseg; ++i) {
      if (st.view.seg_end[i] > covered && covered < kTailStart && (i == 0 || st.view.seg_end[i - 1] < kTailStart)) {
        if (n < kMaxSeg) { nv.seg_end[n] = st.view.seg_end[i]; nv.seg_src[n] = st.view.seg_src[i]; ++n; }
        else st.overflow = 1;
      }
    }
    nv.nseg = n;
    for (int i = n; i < kMaxSeg; ++i) { nv.seg_end[i] = kTfBytes; nv.seg_src[i] = -1; }
    nv.tail = st.view.tail;
    st.view = nv;
    st.consumed += len;
    count -= len;
    out.status = 1;
    if (st.startup_delay <= 0) st.startup_delay++;
    else out.do_sync = 1;
  }
  out.fifo_count = static_cast<int>(count);
  return out;
}

// n further calls of fifo_call(st) for a stream that reads without any time shift (the look-ahead pass's prediction, k_sync.hip: sync_ahead_kernel), as far
// as the FIFO's two counters go: fed += n chunks, and a full frame leaves the read pointer whenever 1.5 frames are queued (the branch "shift <= 0" above
// with shift = 0).  Preconditions (the caller's): no shift pending, the first frame already dropped, chunk = 262144.  The counters repeat every three calls
// (3 x 262144 = 2 x 393216, two reads): whole periods are skipped in one step, so the cost does not grow with n.
__host__ __device__ inline void fifo_skip_unshifted(int64_t& fed, int64_t& consumed, int n)
{
  int i = 0;
  auto step = [&]() {
    fed += kChunkBytes;
    if (fed - consumed >= 3 * kTfSamples) consumed += kTfBytes;
    ++i;
  };
  if (n >= 9) {
    const int64_t queued = fed - consumed;
    step(); step(); step();
    if (fed - consumed == queued) {                        // two reads in three calls: the period
      const int q = (n - i) / 3;
      fed += static_cast<int64_t>(q) * 3 * kChunkBytes;
      consumed += static_cast<int64_t>(q) * 2 * kTfBytes;
      i += 3 * q;
    }
  }
  while (i < n) step();
}

// the state sdr_init leaves (input_sdr.c:167-186): empty FIFO, calloc'ed (all-zero) frame buffer
__host__ __device__ inline void fifo_reset(StreamState& st)
{
  st.consumed = st.fed = 0;
  st.coarse_timeshift = st.fine_timeshift = st.startup_delay = st.force_timesync = st.next_ordinal = st.overflow = 0;
  st.fine_freq_shift = 0;
  st.tuner_hz = 0;
  st.rng = 1;
  st.view.nseg = 1;
  st.view.tail = nullptr;
  for (int i = 0; i < kMaxSeg; ++i) { st.view.seg_end[i] = kTfBytes; st.view.seg_src[i] = -1; }
}

}  // namespace dabhip
