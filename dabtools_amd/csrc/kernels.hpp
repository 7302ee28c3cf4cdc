// kernels.hpp — host-callable launchers of the HIP kernels (k_sync.hip, k_fft.hip, k_decode.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <mutex>

#include "device_types.hpp"

namespace dabhip {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: several engines of one process may sit on different devices (dabhip_multi), each with
// its own host thread.  fn() runs once per device (the current one), its result is kept; callers on other threads of the same device wait for it.
template <class F>
inline hipError_t once_per_device(std::once_flag (&once)[64], hipError_t (&result)[64], F&& fn)
{
  int dev = 0;
  const hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  dev &= 63;
  std::call_once(once[dev], [&]() { result[dev] = fn(); });
  return result[dev];
}

// the scan's device-side preparation in one launch (k_sync.hip): h_* are page-locked host arrays the kernel reads directly
struct ScanSetupArgs {
  const StreamState* h_states;        // null: the device's states stay (a session's further segment)
  const uint8_t* const* h_ptrs;
  const int64_t* h_nbytes;
  const int* h_calls_before;
  StreamState* states;
  StreamState* states_prev;           // null unless the split scan runs (with calls_before and viol)
  const uint8_t** iq_ptrs;
  int64_t* nbytes;
  int* calls_before;
  int* viol;                          // nstreams + 1 entries
  uint4* descs;                       // cleared: desc_vec 16-byte pieces
  size_t desc_vec;
  uint4* info;
  size_t info_vec;
  int nstreams;
  uint8_t* tail_state;                // [nstreams][kTailBytes]: cleared with a fresh decode (h_states != null), kept in a session
  uint8_t* tail_state_prev;           // the incoming tail bytes, for a rescan (like states_prev)
};
hipError_t launch_scan_setup(const ScanSetupArgs& a, hipStream_t stream);
// up to four small arrays of 32-bit words from page-locked host memory to the device, and nzero words cleared, in one launch
struct HostWordsArgs {
  const uint32_t* src[4];
  uint32_t* dst[4];
  uint32_t nwords[4];
  uint32_t* zero;
  uint32_t nzero;
};
hipError_t launch_host_words(const HostWordsArgs& a, hipStream_t stream);
// K1: synchronisation scan, one workgroup per stream, calls [call_begin, call_end) (call_end < 0: all).
// chain_only: FIFO bookkeeping, coarse and fine time only, assuming every frame's coarse frequency offset stays within +-1
// carrier; launch_sync_verify then computes both frequency estimates for all frames in parallel and records the first call
// of each stream that breaks the assumption.  states_in (default: states): where the incoming state is read from;
// stream_list (default: all, block b = stream b): the streams to scan.
// tails: the streams' tail bytes (device_types.hpp: kTailBytes) -- carried in from state_in, out to state_out, one copy per call that reads a frame
// into images[(stream * max_calls + call) * kTailBytes] (FrameView::tail of that call's view points there); chunk: bytes appended per call (< 0: 262144)
struct SyncTails {
  const uint8_t* state_in;
  uint8_t* state_out;
  uint8_t* images;
  int chunk;
};
// The look-ahead schedule of a chain-only scan (small batches; k_sync.hip: sync_ahead_kernel): a pass that computes the chain's estimators for every call a
// stream has left, at every start position near the predicted one, all at once; the chain launch behind it looks them up.
struct SpecArgs {
  int2* table = nullptr;                // [nstreams][nspec][nhyp]: {null-symbol energy, fine time shift} of a read beginning at src0 + 2 (h - nhyp / 2); x < 0: none
  int64_t* src0 = nullptr;              // [nstreams][nspec]: predicted start position (stream offset) of the call's read, -1: no demodulating read expected
  int* ctl = nullptr;                   // [nstreams]: descriptor numbering of the scan (its first call), recorded by the launch with record_base; [nstreams]: table hits
  int nstreams = 0;
  int nspec = 0, nhyp = 0;
  int lookup = 0;                       // chain launches: the table stands for the calls from where the stream stands now
  int call_limit = -1;                  // chain launches: at most this many calls per stream (< 0: to the end)
  int record_base = 0;
};
hipError_t launch_sync_scan(const uint8_t* const* iq, const int64_t* nbytes, StreamState* states, CallDesc* descs, int2* info,
                            int nstreams, int max_calls, int call_begin, int call_end, const double2* tw2048,
                            const double2* tw1536, const uint8_t* prs_q, int afc, hipStream_t stream, bool chain_only = false,
                            const StreamState* states_in = nullptr, const int* stream_list = nullptr, SyncTails tails = SyncTails{nullptr, nullptr, nullptr, -1},
                            SpecArgs spec = SpecArgs{});
// the look-ahead pass over the calls every stream has left from where it stands (at most spec.nspec of them)
hipError_t launch_sync_ahead(const uint8_t* const* iq, const int64_t* nbytes, const StreamState* states, int nstreams, const double2* tw2048, const double2* tw1536,
                             const uint8_t* prs_q, hipStream_t stream, const SpecArgs& spec);
// carry_only = false: the verification pass (violation[b] = first offending call, untouched otherwise);
// carry_only = true: fine_freq_shift carried through the calls that did not demodulate, for the streams without a violation
hipError_t launch_sync_verify(const uint8_t* const* iq, const int64_t* nbytes, const int* calls_before, StreamState* states, CallDesc* descs,
                              int nstreams, int max_calls, const double2* tw2048, const uint8_t* prs_q, int* violation, bool carry_only, hipStream_t stream);

// delta != nullptr: the kernel also leaves the guard's per-symbol error bound delta_c sqrt(sum |x|^2) at delta[(first + j) * 76 + symbol]
// (delta_c = GuardArgs::c of the launches that will read it: the guard level's constant, or kSoftNormC for soft decisions)
hipError_t launch_ofdm_fft(const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first,
                           int nframes, float2* spectra, const float2* tw, hipStream_t stream, float* delta = nullptr, float delta_c = kGuardC);

// K2b: DQPSK + demap + frequency de-interleave -> bit-packed rows.  FIC rows at TF slot frame_slot[first + j];
// MSC rows start at CIF row frame_cif_row[first + j]: planar = scattered into time-de-interleaved logical rows
// (see k_fft.hip), else the four transmitted CIFs of the TF in natural bit order.
hipError_t launch_demap(bool planar, int soft_bits, const float2* spectra, int first, int nframes, const int* frame_slot,
                        const int* frame_cif_row, const uint16_t* qpsk_of_carrier, uint32_t* fic_bits, uint32_t* msc_bits,
                        const GuardArgs& guard, hipStream_t stream);

// FIC pre-pass: DFT of symbols 0..3 of every frame + demap of the three FIC symbols (spectra4: [nframes][4][2048])
hipError_t launch_fic_prepass(int soft_bits, const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first,
                              int nframes, float2* spectra4, const float2* tw, const int* frame_slot, const uint16_t* qpsk_of_carrier,
                              uint32_t* fic_bits, const GuardArgs& guard, hipStream_t stream);

// S1 seam: Viterbi on explicit per-step symbol bytes (forward pass + chain-back + pack)
hipError_t launch_viterbi(const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans, const uint4* steps,
                          uint2* decisions, const uint32_t* prbs_words, uint8_t* out, int record_stride, hipStream_t stream);

// K3/K4: lane-interleave the received rows 64 records at a time (soft_bits: 0 = hard bits, 4 = 4-bit soft values),
// then Viterbi with fused de-puncturing
hipError_t launch_regroup(int soft_bits, const int* job_ids, int ntiles, const DecodeJob* jobs, const int* stream_cif_base,
                          const uint32_t* rows, uint32_t* grouped, hipStream_t stream);
hipError_t launch_fic_group(const uint32_t* fic_rows, int first_block, int nblocks, int block_words, uint32_t* grouped, hipStream_t stream);
hipError_t launch_viterbi_fused(int soft_bits, const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans,
                                const uint32_t* grouped, int row_words, uint2* decisions, const uint32_t* prbs_words, uint8_t* out,
                                int record_stride, hipStream_t stream);
hipError_t launch_viterbi_fused_lanes(int lanes, const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans, const uint32_t* grouped,
                                      int row_words, uint2* decisions, const uint32_t* prbs_words, uint8_t* out, int record_stride, hipStream_t stream);
hipError_t launch_viterbi_fused_two(const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans, const uint32_t* grouped, int row_words,
                                    uint2* decisions, const uint32_t* prbs_words, uint8_t* out, int record_stride, hipStream_t stream);

// The low-latency form of the same decoder (k_vitwave.hip): one WAVE per code word, lane = trellis state -- a fraction of the fused kernel's
// time per code word at four times its lane-ops, for small batches.  Same groups / plans / rows / output; decisions: rows of 64 x 8 bytes, one per
// code word and chunk of kWaveChunk steps: group g's 64 x ceil(nsteps / kWaveChunk) rows start at groups[g].dec_base.
constexpr int kWaveChunk = 60;
hipError_t launch_viterbi_wave(int soft_bits, const WaveGroup* groups, int ngroups, const int* job_ids, const CodewordPlan* plans,
                               const uint32_t* grouped, int row_words, uint2* decisions, const uint32_t* prbs_words, uint8_t* out, int record_stride,
                               hipStream_t stream);

// the one-kernel OFDM stage (k_fused.hip, compiled three times): with the parity guard's test in its symbol loop, without it, and
// with 4-bit soft values instead of hard decisions.
// Data symbols [sym_a, sym_b) of every frame (1..3 = FIC, 4..75 = MSC), nparts workgroups per frame; each transforms the symbol before
// its first data symbol as differential reference.
hipError_t launch_ofdm_demap_fused_guarded(const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                                           const float2* tw, const int* frame_slot, const int* frame_cif_row, const uint16_t* qpsk_of_carrier,
                                           uint32_t* fic_bits, uint32_t* msc_bits, const GuardArgs& guard, hipStream_t stream, int sym_a = 1, int sym_b = 76, int nparts = 4);
// the guarded kernel's audit build (k_fused.hip with -DDABHIP_FUSED_AUDIT=1): also leaves dump_bins[frame][76][2048] (by raw bin) and
// dump_prod[frame][76][2048] ((re, im) of cur conj(prev) of the data symbols as the kernel computed them)
hipError_t launch_ofdm_demap_fused_audit(const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                                         const float2* tw, const int* frame_slot, const int* frame_cif_row, const uint16_t* qpsk_of_carrier,
                                         uint32_t* fic_bits, uint32_t* msc_bits, const GuardArgs& guard, hipStream_t stream, int sym_a, int sym_b, int nparts,
                                         float2* dump_bins, float2* dump_prod);
hipError_t launch_ofdm_demap_fused_soft(bool afc, const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                                        const float2* tw, const int* frame_slot, const int* frame_cif_row, const uint16_t* qpsk_of_carrier,
                                        uint32_t* fic_bits, uint32_t* msc_bits, hipStream_t stream, int sym_a = 1, int sym_b = 76, int nparts = 4);
hipError_t launch_ofdm_demap_fused_plain(bool afc, const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                                         const float2* tw, const int* frame_slot, const int* frame_cif_row, const uint16_t* qpsk_of_carrier,
                                         uint32_t* fic_bits, uint32_t* msc_bits, hipStream_t stream, int sym_a = 1, int sym_b = 76, int nparts = 4);
// parity guard (k_parity.hip): per-symbol error bounds, fp64 re-decision of the flagged carriers, and the audit
hipError_t launch_symbol_delta(const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames, int first, int nframes,
                               int nsym, float* delta, int delta_stride, float delta_c, hipStream_t stream);
hipError_t launch_exact_decide(const uint4* list, const unsigned* counter, unsigned cap, const uint8_t* const* iq, const CallDesc* descs,
                               int max_calls, const int2* frames, const double2* tw2048, const uint16_t* qpsk_of_carrier, const uint16_t* carrier_of_qpsk,
                               const int* frame_slot, const int* frame_cif_row, bool planar, uint32_t* fic_bits, uint32_t* msc_bits, hipStream_t stream);
// list overflow of a guarded launch: its frames' symbols [sym_a, sym_b) decided again in full from fp64 transforms (returns at once otherwise)
hipError_t launch_exact_decide_all(const unsigned* counter, unsigned cap, const uint8_t* const* iq, const CallDesc* descs, int max_calls, const int2* frames,
                                   int first, int nframes, int sym_a, int sym_b, const double2* tw2048, const uint16_t* qpsk_of_carrier,
                                   const int* frame_slot, const int* frame_cif_row, bool planar, bool skip_fic, uint32_t* fic_bits, uint32_t* msc_bits,
                                   hipStream_t stream);
hipError_t launch_decision_audit(const uint8_t* frames_iq, int nframes, const float2* spectra, const uint32_t* fic_bits, const uint32_t* msc_bits,
                                 const double2* tw2048, const uint16_t* qpsk_of_carrier, void* out, hipStream_t stream,
                                 const float2* fused_prods = nullptr, int row_lead = 0, int guard_level = 1);   // fused_prods != null: the fused kernel's audit (spectra = its bins by raw bin)
hipError_t launch_batched_copy(const CopyDesc* descs, int n, hipStream_t stream);
// the descriptors' sources may be page-locked HOST memory (read over PCIe by a small persistent grid); nbytes < 4 GiB each
hipError_t launch_host_gather(const CopyDesc* descs, int n, int workgroups, hipStream_t stream);
// n copies inside the device in one launch, any alignment; max_bytes = the longest of them
hipError_t launch_device_gather(const CopyDesc* descs, int n, uint32_t max_bytes, hipStream_t stream);
hipError_t launch_fib_crc(const uint8_t* fibs, int nfib, const uint16_t* crc_tab, uint8_t* ok, hipStream_t stream);

// K5: ETI header/FIB copy, EOF CRC, trailer
hipError_t launch_eti_finish(const EtiFrameMeta* meta, int nframes, const uint8_t* headers, int header_stride, const uint8_t* fibs,
                             const uint16_t* crc_tab, const uint16_t* shift_cols, uint8_t* eti, hipStream_t stream);

// k_probe.hip: streaming rates of this device (fill, copy, K2's read/write mix) in GB/s -- bench.py's yardstick beside the K2 figure
int stream_ceiling(int device, size_t bytes, int reps, double* gbs);

}  // namespace dabhip
