// device_types.hpp — plain structs shared by host code and HIP kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace dabhip {

constexpr int kMaxSeg = 12;
// The last kTailBytes of the frame buffer are kept as BYTES, not as views (round 5).  A negative time shift makes sdr_read_fifo read short and
// leave the end of sdr->buffer as it was (sdr_fifo.c:56-59); the fine time shift is at least (768 - 1536) * 2 bytes (sdr_sync.c:197-201) and the
// coarse one is never negative, so a locked receiver's stale data lives in buffer[393216 - 1536 ..).  A receiver whose sample clock runs fast
// against the transmitter's reads short on EVERY call: described as views, the sources of those bytes never age out (a session would have to keep
// the whole stream) and the views nest without bound.  So K1 carries these 1536 bytes along (768 two-byte words in registers of the stream's
// workgroup, sync_scan_kernel) and leaves every frame its own copy (FrameView::tail); views describe the rest of the buffer, where stale data only
// appears after a read that ran the FIFO dry (a large coarse correction) and is overwritten by the next ordinary read.
constexpr int kTailBytes = 1536;
constexpr int kTailStart = 196608 * 2 - kTailBytes;

// How the 393216-byte frame buffer of the reference (sdr->buffer, input_sdr.h:16) looks
// after one sdr_read_fifo() call, expressed as views into the never-modified IQ stream:
// buffer position p in [seg_end[i-1], seg_end[i]) holds stream byte seg_src[i] + p
// (seg_src < 0: the calloc'ed zero byte).  Segment 0 is what this call read; later
// segments are the stale data left by earlier, longer reads (sdr_fifo.c:56-59).
// tail != nullptr: positions p >= kTailStart hold tail[p - kTailStart], whatever the segments say (where segment 0 covers them, the same bytes).
struct FrameView {
  int32_t nseg;
  int32_t seg_end[kMaxSeg];
  int64_t seg_src[kMaxSeg];
  const uint8_t* tail;
};
// buffer position p of a frame -> its byte (the one rule every kernel reads the frame buffer by)
__host__ __device__ __forceinline__ int frame_byte(const uint8_t* stream, const FrameView& v, int p)
{
  if (p >= kTailStart && v.tail) return v.tail[p - kTailStart];
  int i = 0;
  while (i < v.nseg - 1 && p >= v.seg_end[i]) ++i;
  const int64_t s = v.seg_src[i];
  return s < 0 ? 0 : stream[s + p];
}
// two bytes at an even position (segment boundaries and sources are even: both bytes come from the same place), little endian
__host__ __device__ __forceinline__ unsigned frame_u16(const uint8_t* stream, const FrameView& v, int p)
{
  if (p >= kTailStart && v.tail) return *reinterpret_cast<const uint16_t*>(v.tail + (p - kTailStart));
  int i = 0;
  while (i < v.nseg - 1 && p >= v.seg_end[i]) ++i;
  const int64_t s = v.seg_src[i];
  return s < 0 ? 0u : *reinterpret_cast<const uint16_t*>(stream + s + p);
}

// Result of one sdr_demod() call (input_sdr.c:27-165) for one stream.
struct CallDesc {
  int32_t status;        // 0 = no frame read, 1 = frame read but dropped, 2 = demodulated
  int32_t ordinal;       // dense index of this TF among the demodulated TFs of the stream
  int32_t coarse_timeshift, fine_timeshift, coarse_freq_shift, fifo_count;
  double fine_freq_shift;
  int32_t nco_hz;        // software AFC: frequency the samples of this frame were de-rotated by (0 in parity mode)
  int32_t pad;
  FrameView view;
  int64_t pad16;         // (cleared and copied in 16-byte pieces)
};
static_assert(sizeof(CallDesc) % 16 == 0, "CallDesc is cleared in 16-byte pieces");

// Front-end state carried from call to call (struct sdr_state_t, input_sdr.h:12-41)
struct StreamState {
  int64_t consumed;      // stream offset of the FIFO read pointer
  int64_t fed;           // bytes written to the FIFO so far
  int32_t coarse_timeshift, fine_timeshift;
  int32_t startup_delay, force_timesync;
  int32_t next_ordinal;
  int32_t overflow;      // set if the stale-tail bookkeeping exceeded kMaxSeg
  double fine_freq_shift;
  int32_t tuner_hz;      // software AFC: accumulated re-tuning, the NCO's frequency (sdr->frequency - nominal, dab2eti.c:76-103)
  uint32_t rng;          // state of the generator standing in for rand() at dab2eti.c:90-93
  FrameView view;
};

// A rate-compatible punctured code word to decode: where its bits come from, how they
// are punctured and where the decoded bytes go.
struct CodewordPlan {
  int32_t blocks[4];     // segments of 128 mother-code bits
  uint32_t mask[4];      // puncturing masks of the segments
  int32_t nsteps;        // trellis steps = data bits + 6
  int32_t start_bit;     // first bit inside the (time de-interleaved) CIF, or inside the FIC block
  int32_t out_offset;    // byte offset of the decoded data inside the output record
  int32_t out_bytes;     // (nsteps - 6) / 8
};

// One output record to decode into: an ETI frame (MSC) or a 96-byte FIC block.
struct DecodeJob {
  int32_t stream;        // MSC: selects the stream's first CIF row; FIC: unused
  int32_t cif;           // MSC: linear index of the oldest of the 16 CIFs; FIC: 4*tf_slot + block
};

// 64 equal-length code words handled by one wave: lane l decodes job job_ids[first + l]
// (or job first + l when there is no id list) with code word plan `plan`.
struct WaveGroup {
  int32_t plan;
  int32_t first;
  int32_t count;         // valid lanes (<= 64)
  int32_t nsteps;
  int64_t step_base;     // row offset into the trellis-input buffer (rows of 64 x 16 bytes)
  int64_t dec_base;      // row offset into the decision buffer (rows of 64 x 8 bytes)
};

// One piece of a batched device-to-device copy (sizes are multiples of 4 bytes, pointers 4-byte aligned)
struct CopyDesc {
  const uint8_t* src;
  uint8_t* dst;
  uint32_t nbytes;
  uint32_t pad;
};

// Parity guard of the OFDM stage (k_parity.hip): decisions whose fp32 margin is inside the error band are listed for an fp64
// re-decision.  delta[frame * delta_stride + symbol] = c * sqrt(energy of the symbol's 2048 samples), c = GuardArgs::c.
// Two sets of constants (dabhip_engine_set_parity_guard(e, level); DESIGN.md section 3 carries the derivation of the second):
//   level 1, "measured": >= 4.5 x the worst error dabhip_stage_decision_audit_fused has measured over > 10^10 decisions (profiles/r0N_decision_audit.json);
//   level 2, "proven": >= a rigorous forward-error bound of THIS transform (radix 8.8.8.4 as fft_core.hpp writes it, round-to-nearest fp32, fused
//   multiply-adds where the source says so, twiddles = float(cos / sin in double)) and of the differential product (diff_re / diff_im below):
//     |X32[k] - X[k]| <= 2^-24 sqrt(2048) (6.7072 + 7.7072 + 7.7072 + 2) (1 + 1e-5) |x|_2 = 6.5072e-5 |x|_2      (any bin k; tools/fft_error_bound.py)
//     |fl(Re / Im(cur conj(prev))) - (the same of the fp32 bins, exactly)| <= 2 (2^-24) (1 + 2^-24) |cur|_2 |prev|_2 = 1.1921e-7 |cur|_2 |prev|_2
//   both rounded up a little further so that the fp32 evaluation of the threshold itself (sqrtf of the energy, three fused multiply-adds: relative
//   error < 1e-6) stays on the safe side.
constexpr float kGuardC = 5.0e-6f;       // level 1: bound on |X32 - X| of any bin, relative to sqrt(sum_n |x_n|^2)   (measured worst: 7.6e-7)
constexpr float kGuardProd = 5.0e-7f;    // level 1: rounding of the fp32 product Re/Im(cur conj(prev)), relative to |cur|_1 |prev|_1   (measured worst: 1.1e-7)
constexpr float kGuardCProven = 6.6e-5f;      // level 2 (bound: 6.5072e-5)
constexpr float kGuardProdProven = 1.25e-7f;  // level 2 (bound: 1.1921e-7; |.|_2 <= |.|_1 gives the test further room)
constexpr int kDefaultGuardLevel = 2;     // the proven band: exact by construction; price: 13 x the re-decisions on noisy input, none on clean input
__host__ __device__ constexpr float guard_c_of(int level) { return level >= 2 ? kGuardCProven : kGuardC; }
__host__ __device__ constexpr float guard_prod_of(int level) { return level >= 2 ? kGuardProdProven : kGuardProd; }
// the guard's test, the same operations in every kernel that applies it (so that they all list the same decisions):
//   |cur|_1 (dp + prod |prev|_1) + (|prev|_1 + dp) dc
// = |cur|_1 dp + |prev|_1 dc  (the bins' errors dc, dp carried into the product)  +  prod |cur|_1 |prev|_1  (the product's own roundings)  +  dc dp  (error x error).
__host__ __device__ __forceinline__ float guard_threshold(float n1c, float n1p, float dc, float dp, float prod)
{
  return fmaf(n1c, fmaf(n1p, prod, dp), (n1p + dp) * dc);
}
// The proven level lists by the PER-BIN form of the bound (GuardArgs::per_bin): the stage terms depend on the bin's index digits k = a' + 8 b' + 64 c' + 512 d' --
// output q of a radix-8 block takes no turn when q is even and no table factor when q = 0, and stage A's additions of integers are exact --
//     kappa_A(a') = 0 | 2.7071 (even) | 6.7013 (odd),   kappa_B(b'), kappa_C(c') = 3 | 5.7071 | 7.7013,   kappa_D = 2
// so bin k's error is at most guard_bin_scale(k) = (kappa_A + kappa_B + kappa_C + 2) / 24.104 of the worst bin's (0.33 .. 1, 0.79 on average), and its norms
// enter as |.|_2 (what the derivation bounds) instead of |.|_1: 38 % fewer decisions listed on noisy input than by the flat rule, the same guarantee.
// tools/fft_error_bound.py computes the same table (bin_bound(k)); tests/test_bench_launch.py holds this function against it.  The measured level keeps the
// flat rule of rounds 2 - 5 (its constants are measured maxima over all bins, not a per-bin statement).
__host__ __device__ __forceinline__ float guard_bin_scale(int k)
{
  constexpr float kTurn = 1.9942368f, kCmul = 2.7071072f;          // (R2), (R3) of DESIGN.md section 3, rounded up
  constexpr float kMax = (kTurn + 2.0f + kCmul) + 2.0f * (3.0f + kTurn + kCmul) + 2.0f;
  const int a = k & 7, b = (k >> 3) & 7, c = (k >> 6) & 7;
  const float ka = a == 0 ? 0.0f : ((a & 1) ? kTurn + 2.0f + kCmul : kCmul);
  const float kb = b == 0 ? 3.0f : ((b & 1) ? 3.0f + kTurn + kCmul : 3.0f + kCmul);
  const float kc = c == 0 ? 3.0f : ((c & 1) ? 3.0f + kTurn + kCmul : 3.0f + kCmul);
  return (ka + kb + kc + 2.0f) * (1.000004f / kMax);
}
__host__ __device__ __forceinline__ float guard_norm2(float x, float y) { return sqrtf(fmaf(x, x, y * y)) * 1.000001f; }     // >= |(x, y)|_2
__host__ __device__ __forceinline__ float guard_bin_threshold(float cx, float cy, float px, float py, int k, float dc, float dp, float prod)
{
  const float f = guard_bin_scale(k);
  return guard_threshold(guard_norm2(cx, cy), guard_norm2(px, py), f * dc, f * dp, prod);
}
// Soft decisions (extension): value = round(soft_scale x) clamped to +-7, x = Re / Im of cur conj(prev).  The scale is made of
// the two symbols' sample energies -- on a noise-free Mode-I signal mean |x| = (2048 / 1536) s(l) s(l-1) / sqrt(2), s = sqrt(sum_n
// |x_n|^2), which the factor below maps to 7.0 (= the clamp: see below) -- so it is known before a symbol is transformed (the one-kernel OFDM stage cannot
// wait for a mean over its own output) and is the same in every kernel that demaps.  dc, dp = kSoftNormC s of the two symbols (the energies travel
// in the guard's delta array, scaled by GuardArgs::c = kSoftNormC: soft decisions and the guard never run together).
// Round 4: the gain was 4.5 until the soft rule got an oracle (oracle/or_soft.c) that can be run in any quantisation: a sweep of the gain on the CPU
// (tools/soft_quant_loss.py --gain, 14 captures x 24 TF per point) put the optimum of the 4-bit values at 6.5 .. 7 -- a clean value AT the clamp: range is
// worth less than resolution near zero, where the decisions are made.  Payload BER 4-bit / unquantised at 5, 6, 7 dB: 1.17, 1.14, 1.82 at gain 4.5;
// 1.06, 1.06, ~1.3 at 7.0 (profiles/r04_soft_quantisation.json, profiles/r04_soft_gain_sweep.txt).
constexpr float kSoftNormC = 5.0e-6f;
constexpr float kSoftGain = 7.0f / 0.94280904f;
__host__ __device__ inline float soft_scale(float dc, float dp)
{
  const float prod = dc * dp;
  return prod > 0.0f ? kSoftGain * kSoftNormC * kSoftNormC / prod : 0.0f;
}

// The differential product cur conj(prev) of the demapper (input_sdr.c:135-143: re = Re, im = -Im) in fp32, with its roundings FIXED -- one multiply and one
// fused multiply-add each -- so that every build of every kernel that decides on it rounds alike.  Left to the compiler's contraction, "a b + c d" came
// out as different pairings of packed multiplies and fused multiply-adds in the shipping and in the audit build of the fused kernel (round 5: the
// audit's check that both leave the same raw bits failed on 5 .. 8 dB captures; either pairing is within kGuardProd, but an audit has to measure
// the arithmetic that ships).
__host__ __device__ __forceinline__ float diff_re(float cx, float cy, float px, float py) { return fmaf(cx, px, cy * py); }
__host__ __device__ __forceinline__ float diff_im(float cx, float cy, float px, float py) { return fmaf(cx, py, -(cy * px)); }

struct GuardArgs {
  const float* delta;    // nullptr: guard off (the fused OFDM kernel computes its bounds itself and only tests this for null); with soft
                         // decisions: the same array, read for the scale (list == nullptr then)
  int32_t delta_stride;
  float c, prod;         // the level's constants (guard_c_of / guard_prod_of); soft decisions: c = kSoftNormC
  int32_t per_bin;       // 1: list by guard_bin_threshold (the proven level), 0: by the flat rule guard_threshold on |.|_1 (the measured level)
  uint32_t cap;          // capacity of list
  uint4* list;           // {frame index, symbol << 16 | raw bin, address of the first byte of the symbol's window in the IQ stream when it and the previous
                         //  symbol's lie contiguously inside what the call read (low, high word; 0: to be looked up through the frame's view)}
  uint32_t* counter;     // entries appended (may exceed cap: overflow, detected by the host)
};

// Per ETI frame: what eti_finish_kernel needs besides the decoded sub-channel data.
constexpr int kEtiHeaderMax = 272;   // 8 + 4*64 + 4 bytes of SYNC/FC/STC/EOH, rounded up
struct EtiFrameMeta {
  int32_t header_len;    // bytes produced by the header builder (misc.c:153-213)
  int32_t mst_bytes;     // decoded sub-channel bytes following the 96 FIB bytes
  int32_t fib_block;     // index of the 96-byte FIB triple of the oldest CIF (4*tf_slot + cif)
  int32_t pad;
};

}  // namespace dabhip
