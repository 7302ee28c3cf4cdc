/*
 * dabhip.h — C ABI of libdabhip.so, the MI355X-native DSP back end for dab2eti.
 *
 * Plain C, caller-owned buffers, int error codes (0 = ok, <0 = error, see
 * dabhip_last_error()).  No C++ or torch types cross this boundary.
 *
 * The reference (linuxstb/dabtools) has no plugin system; its seams are C function
 * signatures chosen at link time.  Each entry point below names the reference interface
 * (file:line under src/) it replaces.  INTEGRATION.md shows the reference-side binding.
 *
 *   S1  decoder seam      viterbi.h:6-8, viterbi_spiral.h:22-23   -> dabhip_*viterbi*
 *   S2  front-end seam    input_sdr.h:43-44                        -> dabhip_sdr_*
 *   S3  back-end seam     dab.h:91-92 (+ eti_callback dab.h:88)    -> dabhip_dab_*
 *   batch engine (new: B independent ensembles resident in HBM)    -> dabhip_engine_*
 *   stage entries for parity tests                                 -> dabhip_stage_*
 *   synthetic Mode-I modulator (workload generator; host, or on the GPU) -> dabhip_synth_*
 */
#ifndef DABHIP_H
#define DABHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DABHIP_TF_BYTES 393216        /* one transmission frame of cu8 IQ (input_sdr.h:16) */
#define DABHIP_CHUNK_BYTES 262144     /* DEFAULT_BUF_LENGTH, input_sdr.h:9 */
#define DABHIP_FIC_BITS 9216          /* fic_symbols_demapped[3][3072], dab.h:29 */
#define DABHIP_MSC_BITS 221184        /* msc_symbols_demapped[72][3072], dab.h:32 */
#define DABHIP_ETI_BYTES 6144         /* dab2eti.c:132-135 */

/* ---- library ------------------------------------------------------------------------ */
const char *dabhip_last_error(void);          /* thread-local text of the last failure */
int dabhip_device_count(void);                /* number of visible HIP devices (0 without a GPU) */

/* ---- S1: decoder seam ---------------------------------------------------------------- */
/* Replaces init_viterbi() (viterbi.h:6, viterbi.c:455) / create_viterbi(len)
 * (viterbi_spiral.h:22).  Returns an opaque handle (never NULL on success). */
void *dabhip_create_viterbi(int len);
int dabhip_init_viterbi(void);
/* Replaces viterbi(p, symbols, data, framebits) (viterbi.h:8, viterbi.c:352-451; called
 * from fic.c:186 and misc.c:262).  symbols: 4*(framebits+6) bytes, 127/129 hard values,
 * 128 = erasure (depuncture.c:36-43).  data: (framebits+7)/8 bytes, MSB first.  Decisions
 * are those of the scalar reference decoder (metrics 3/-7/0, ties keep the low
 * predecessor).  p may be NULL (uses a process-wide engine on device 0). */
void dabhip_viterbi(void *p, unsigned char *symbols, unsigned char *data, int framebits);
/* Batch form: n code words of equal length, symbols packed back to back. */
int dabhip_viterbi_batch(void *p, const unsigned char *symbols, unsigned char *data, int framebits, int n);

/* ---- S2: front-end seam --------------------------------------------------------------- */
typedef struct dabhip_sdr dabhip_sdr;
/* Replaces sdr_init(struct sdr_state_t*) (input_sdr.h:44, input_sdr.c:167-186). */
dabhip_sdr *dabhip_sdr_init(int device);
void dabhip_sdr_free(dabhip_sdr *s);
/* Replaces sdr_demod(tf, sdr) (input_sdr.h:43, input_sdr.c:27-165).  input_buffer /
 * input_buffer_len are what rtlsdr_callback stores in sdr->input_buffer (dab2eti.c:125-126).
 * On return 1 the two arrays hold tf->fic_symbols_demapped (9216 bytes of 0/1) and
 * tf->msc_symbols_demapped (221184 bytes of 0/1); on return 0 they are untouched.
 * <0 = error. */
int dabhip_sdr_demod(dabhip_sdr *s, const uint8_t *input_buffer, int input_buffer_len,
                     uint8_t *fic_symbols_demapped, uint8_t *msc_symbols_demapped);
/* The side-channel outputs dab2eti.c:76-103 reads after each call. */
int32_t dabhip_sdr_coarse_timeshift(const dabhip_sdr *s);
int32_t dabhip_sdr_fine_timeshift(const dabhip_sdr *s);
int32_t dabhip_sdr_coarse_freq_shift(const dabhip_sdr *s);
double dabhip_sdr_fine_freq_shift(const dabhip_sdr *s);

/* ---- S3: back-end seam ---------------------------------------------------------------- */
typedef struct dabhip_dab dabhip_dab;
typedef void (*dabhip_eti_callback)(uint8_t *eti);                    /* dab.h:88 */
typedef void (*dabhip_eti_sink)(const uint8_t *eti, int stream, void *user);
/* Replaces init_dab_state(&dab, device_state, eti_callback) (dab.h:91, dab.c:14-33). */
dabhip_dab *dabhip_dab_init(int device, dabhip_eti_callback cb);
void dabhip_dab_free(dabhip_dab *d);
/* The caller fills these before each dabhip_dab_process_frame(), exactly as sdr_demod
 * fills dab->tfs[dab->tfidx] (dab2eti.c:68). */
uint8_t *dabhip_dab_tf_fic(dabhip_dab *d);   /* 9216 bytes */
uint8_t *dabhip_dab_tf_msc(dabhip_dab *d);   /* 221184 bytes */
/* Replaces dab_process_frame(dab) (dab.h:92, dab.c:35-98): invokes the callback 0 or 4
 * times, synchronously, with a pointer to a 6144-byte frame valid during the call. */
int dabhip_dab_process_frame(dabhip_dab *d);
int dabhip_dab_locked(const dabhip_dab *d);
/* What the reference's dab_process_frame prints on stderr for its operator -- "Locked" (dab.c:51), "Lock lost, resetting ringbuffer" (dab.c:57), the one-time
 * ensemble dump (dab.c:78-82, misc.c:316-328) -- as text: what is pending since the last call is copied to buf (NUL-terminated, cut at cap - 1 bytes) and
 * cleared; returns the pending text's length (0: nothing happened), < 0: bad handle.  integration/dab_hip.c writes it to stderr. */
int64_t dabhip_dab_take_log(dabhip_dab *d, char *buf, int64_t cap);
/* as dabhip_engine_stream_status for this seam: with a flagged multiplex dabhip_dab_process_frame invokes the callback 0 times where
 * dab_process_frame (dab.c:85-95 -> misc.c:218-314) would run off its arrays */
uint32_t dabhip_dab_status(const dabhip_dab *d);
/* Soft-decision extension of this seam (not in the reference: dab.h:27-33 carries 0/1 bytes): after dabhip_dab_set_soft(d, 1) --
 * before the first frame only -- the two arrays carry signed 4-bit values as int8 (-7 .. 7; > 0: the hard bit would be 0; the
 * demapper's rule is dabhip_engine_set_soft's) and the FIC / MSC decoders use them as branch metrics.  Lets a test feed the
 * decoders the very values an independent restatement decodes (oracle/or_soft.c). */
int dabhip_dab_set_soft(dabhip_dab *d, int enable);
/* FIBs (12 x 32 bytes) and CRC flags (12) of the TF processed last (struct tf_fibs_t, dab.h:21-25). */
int dabhip_dab_last_fibs(const dabhip_dab *d, uint8_t *fibs, uint8_t *crc_ok);

/* ---- batch engine ---------------------------------------------------------------------- */
typedef struct dabhip_engine dabhip_engine;
dabhip_engine *dabhip_engine_create(int device);
/* The same with an explicit size for the engine's host thread pool (per-stream control plane, work lists, staging copies);
 * 0 = automatic (half the cores, at most 24).  Several engines in one process should share the host between them. */
dabhip_engine *dabhip_engine_create_ex(int device, int host_threads);
/* The same with the engine's host threads bound to the listed CPUs (ncpus = 0: to the CPUs of the device's NUMA node on a machine with several,
 * else unbound).  dabhip_engine_host_cpus reports what was applied. */
dabhip_engine *dabhip_engine_create_on_cpus(int device, int host_threads, const int32_t *cpus, int ncpus);
int dabhip_engine_host_cpus(const dabhip_engine *e, int32_t *cpus, int cap, int *numa_node);
void dabhip_engine_destroy(dabhip_engine *e);

/* Decode B independent cu8 streams (the replay loop of dab2eti.c:60-130 per stream, in
 * 262144-byte chunks, trailing partial chunk dropped).  iq[b] are DEVICE pointers when
 * on_device != 0, host pointers otherwise.  ETI frames stay in device memory; query
 * them with the calls below.  Returns total ETI frames produced, <0 on error. */
int64_t dabhip_engine_decode(dabhip_engine *e, const uint8_t *const *iq, const size_t *nbytes, int nstreams,
                             int on_device);
int64_t dabhip_engine_eti_count(const dabhip_engine *e, int stream);      /* frames of one stream */
/* Per-stream status of the last decode (fault isolation).  The FIC carries a 16-bit CRC and fib_parse (fic.c:47-130) validates nothing: a corrupted
 * FIB that passes the CRC can signal a multiplex the reference cannot assemble inside its own arrays -- create_eti then writes past eti[6144]
 * (misc.c:233,246-296), uep_/eep_depuncture read past cif_time_deinterleaved[55296] (depuncture.c:84-132), eeptable[] is indexed past its 8 rows
 * (fic.c:84): undefined behaviour, no parity target.  Here such a stream is flagged and emits NO frames while its multiplex is in that state (for
 * good: the reference only ever adds sub-channels, misc.c:14-21); lock rule, CIF ring and frame counter move on as they would; every other stream
 * of the batch decodes exactly as if it were alone.  0 = fine; 0xffffffff = no such stream. */
#define DABHIP_STREAM_MUX_OVERFLOW 1u       /* header + FIC + sub-channel bytes + trailer exceed 6144 bytes */
#define DABHIP_STREAM_SUBCH_OUTSIDE_CIF 2u  /* a sub-channel's transmitted bits end beyond capacity unit 863 */
#define DABHIP_STREAM_EEP_OPTION 4u         /* EEP protection option > 1: outside ETSI EN 300 401 and past the reference's table */
#define DABHIP_STREAM_SUBCH_SIZE 8u         /* EEP size below one unit of its level (bit rate 0): the reference's frame carries uninitialised stack bytes there */
uint32_t dabhip_engine_stream_status(const dabhip_engine *e, int stream);
/* The reference's operator feedback for one stream of the last decode: what dab_process_frame prints on stderr -- "Locked" (dab.c:51), "Lock lost,
 * resetting ringbuffer" (dab.c:57), the one-time ensemble dump "ENSEMBLE_INFO: ..." / "SubChId=..." (dab.c:78-82, misc.c:316-328) -- same text, same
 * order.  The text pending since the last call is copied to buf (NUL-terminated, cut at cap - 1 bytes) and cleared; returns its full length (0: nothing
 * to report), < 0: no such stream.  dab2eti-hip prints it on stderr. */
int64_t dabhip_engine_stream_log(dabhip_engine *e, int stream, char *buf, int64_t cap);
/* Copy the ETI frames of one stream (in emission order) to host memory. */
int64_t dabhip_engine_eti_read(dabhip_engine *e, int stream, uint8_t *dst, int64_t cap_frames);
/* Deliver all frames, stream by stream in emission order, to a sink (stdout contract helper). */
int64_t dabhip_engine_eti_drain(dabhip_engine *e, dabhip_eti_sink sink, void *user);
const void *dabhip_engine_eti_device_ptr(const dabhip_engine *e, int64_t *nframes); /* all frames, stream-major */
/* The output leg of the CLI contract (eti_callback -> write(1, eti, 6144), dab2eti.c:132-135) without a stall: ALL frames of the last decode,
 * stream by stream in emission order (the order dabhip_engine_eti_drain delivers them in), copied to dst -- page-locked memory from
 * dabhip_host_alloc for a true asynchronous DMA -- on a stream of their own.  Returns at once with the number of frames on their way (<= cap_frames);
 * the copy runs beside the NEXT decode (only that decode's ETI-writing launches wait for it); dabhip_engine_eti_fetch_wait returns when the
 * bytes have arrived.  dst must stay untouched in between.  At most TWO fetches may be outstanding (two output buffers: one being written out
 * while the next fills); each dabhip_engine_eti_fetch_wait waits for the OLDEST fetch not yet waited for, and a third fetch without a wait is
 * refused (-1).  The waiting thread may be another one than the decoding thread (the CLI's writer thread). */
int64_t dabhip_engine_eti_fetch(dabhip_engine *e, uint8_t *dst, int64_t cap_frames);
int dabhip_engine_eti_fetch_wait(dabhip_engine *e);

/* Software AFC (beyond the reference's file-less operation): when enabled every stream gets an NCO
 * steered by the tuner feedback rule of dab2eti.c:76-103 (coarse offset > 1 carrier: +-1000 Hz; = 1: a random step
 * below 1000 Hz; else fine estimate / 3 when above 50 Hz), so captures with a carrier frequency offset decode without a
 * tuner.  Default off = parity mode (samples untouched, results identical to the reference). */
int dabhip_engine_set_afc(dabhip_engine *e, int enable);

/* Soft-decision decoding (BASELINE config 5; the reference decodes hard decisions only,
 * input_sdr.c:157-158, depuncture.c:36-43): the demapper emits signed 4-bit values that travel through time
 * de-interleaving and de-puncturing into the Viterbi branch metrics of FIC and MSC.  Batch engine only.  Default off =
 * parity mode (bit-exact with the reference).  With soft decisions the output equals the reference's wherever that decodes
 * without errors and is better (lower BER) at low SNR. */
int dabhip_engine_set_soft(dabhip_engine *e, int enable);

/* Parity guard (default ON).  The reference takes its hard decisions as sign tests on fp64 FFTW spectra (input_sdr.c:132-162);
 * the OFDM stage here transforms in fp32.  With the guard on, every decision whose |Re| or |Im| of cur*conj(prev) lies inside
 * the fp32 error band is re-decided in fp64 from the int8 samples, so the demapped bits -- and therefore the ETI bytes -- are
 * those exact arithmetic gives.  `level`:
 *   DABHIP_GUARD_OFF (0)       raw fp32 decisions; their disagreement rate is what dabhip_stage_decision_audit measures;
 *   DABHIP_GUARD_MEASURED (1)  the band is >= 4.5 x the worst error measured over > 10^10 audited decisions ("exact with measured margin");
 *   DABHIP_GUARD_PROVEN (2)    the band is a rigorous forward-error bound of this very transform and product ("exact by construction":
 *                              DESIGN.md section 3 carries the derivation, tools/fft_error_bound.py the arithmetic); 13 x as wide, i.e. 13 x
 *                              the re-decisions on noisy input, none on clean input;
 *   DABHIP_GUARD_DEFAULT (-1)  the library's default level (dabhip_parity_guard_default_level()).
 * Applies to hard decisions without the software AFC (the two configurations that claim reference semantics). */
#define DABHIP_GUARD_DEFAULT (-1)
#define DABHIP_GUARD_OFF 0
#define DABHIP_GUARD_MEASURED 1
#define DABHIP_GUARD_PROVEN 2
int dabhip_engine_set_parity_guard(dabhip_engine *e, int level);
int dabhip_engine_parity_guard_level(const dabhip_engine *e);
int dabhip_parity_guard_default_level(void);
/* the two constants of a level: bound on a bin's error relative to sqrt(sum |x_n|^2), bound on the product's rounding relative to |cur|_1 |prev|_1 */
int dabhip_parity_guard_constants(int level, double *bin_c, double *prod_c);
/* The proven level lists per bin: raw bin k's error bound is this fraction (0.33 .. 1) of the level's bin constant -- the bound's stage terms depend on the
 * bin's index digits; < 0: no such bin. */
double dabhip_parity_guard_bin_scale(int raw_bin);
/* Decisions the guard re-decided in the last decode, and hard decisions taken in all. */
int dabhip_engine_guard_stats(const dabhip_engine *e, int64_t *flagged, int64_t *decisions);
/* The guard lists the decisions to re-decide per kernel launch (64 entries per TF, at least 262,144; the proven level: 1024, 1,048,576).  A launch with more of them
 * -- input that synchronises but has many near-zero products: strongly notched, narrowband or near-DC frames -- does not fail: its
 * frames are decided again in full from fp64 transforms (slow, still the bits of exact arithmetic).  Number of such launches in the
 * last decode; and the list capacity as a test knob (0 = automatic). */
int dabhip_engine_guard_overflows(const dabhip_engine *e);
int dabhip_engine_set_guard_list_cap(dabhip_engine *e, uint32_t cap);

/* Sub-channel filter (the reference's TODO.md:28-31, "save CPU time by not decoding data which will later be discarded"):
 * only the listed SubChIds (0..63) are decoded and carried; the ETI frames then list exactly those in their STC (NST, FL,
 * HCRC and EOF CRC follow; the FIC is passed on unchanged), each one's payload identical to the unfiltered frame's.
 * n <= 0 = all (default, the reference's frames).  Takes effect with the next decode. */
int dabhip_engine_set_subchannels(dabhip_engine *e, const int32_t *ids, int n);

/* OFDM stage variant.  enable != 0 (default): the 2048-point transforms and the DQPSK demap / de-interleave scatter run as
 * ONE kernel that never writes the complex64 spectra (311,296 B read + 28,800 B written per TF).  enable == 0: the two
 * kernels K2 (cu8 -> complex64 spectra, 1,556,480 B per TF: the HBM-roofline stage of SURVEY.md 8(d)) and K2b (spectra ->
 * bits, re-reading the 1.2 MB).  Output bits -- and soft values -- hence ETI bytes, are identical in both.
 * dabhip_engine_fft_stats describes whichever kernel ran; dabhip_engine_fft_roofline measures K2 by itself. */
int dabhip_engine_set_fused(dabhip_engine *e, int enable);

/* Schedule of K1's per-stream chain (sdr_demod's FIFO + time synchronisation, input_sdr.c:36-84, where call n + 1 is positioned by what call n
 * found).  mode 0: the chain, call after call.  mode 1: with a look-ahead pass -- what a call computes from its frame (null-symbol energy,
 * fine time search) depends only on where in the stream its read began, and a locked receiver's reads begin within a few samples of a predictable
 * place: one pass computes both for every remaining call and every start position near the predicted one, all at once, and the chain looks them up
 * (and computes them itself where a read began elsewhere).  Same results in every case, call for call.  mode -1 (default): the pass for small
 * batches (where the chain leaves most of the device idle), the plain chain for large ones; DABHIP_K1_SPEC in the environment sets the default.
 * Mode 1 is a test / measurement knob: it is honoured up to 512 streams (beyond that the pass cannot help and its table would run to tens of megabytes).
 * dabhip_engine_stage_ms reports the calls served from the pass's table as "sync_spec_calls". */
int dabhip_engine_set_sync_speculation(dabhip_engine *e, int mode);

/* ---- several devices of one node (BASELINE configs[3]: 2048 streams = 256 per GPU x 8) --------------
 * dab2eti.c:237,279-302 drives ONE device from one demod thread.  A batch of independent ensembles shards by stream with no
 * data exchange at all: the streams of a decode are dealt to the listed devices in contiguous slices (slice i of n takes B / n
 * streams, the first B mod n slices one more: 2048 on 8 = stream s on device s / 256; 10 on 4 = 3, 3, 2, 2 -- dabhip_multi_plan
 * answers for any batch size, before the decode), every slice is a complete batch engine with its own persistent host thread and HIP streams, and all slices decode at
 * once.  No collective, no peer access.  Frames are read back per GLOBAL stream index, or drained in stream order.
 * A device may be listed more than once (each entry is its own slice).  iq[b]: host pointers, or -- on_device != 0 --
 * device pointers that live on the device of stream b's slice (dabhip_multi_slice_of). */
typedef struct dabhip_multi dabhip_multi;
dabhip_multi *dabhip_multi_create(const int *devices, int n);
void dabhip_multi_destroy(dabhip_multi *m);
int dabhip_multi_slices(const dabhip_multi *m);
/* Slice and device that decode `stream` of a batch of `nstreams` -- the dealing rule itself, valid before the first decode (device pointers of
 * an on_device decode must live on that device).  0, <0 on error. */
int dabhip_multi_plan(const dabhip_multi *m, int nstreams, int stream, int *slice, int *device);
/* The same for the size of the LAST dabhip_multi_decode (-1 before the first one). */
int dabhip_multi_slice_of(const dabhip_multi *m, int stream, int *device);
/* Host placement of a slice (dab2eti.c:237 is the one unplaced demod thread this replaces): its decode thread, control-plane pool and host lane are
 * bound to a chunk of the CPUs of its device's NUMA node (/sys/bus/pci/devices/<bdf>/numa_node; the slices of a node get disjoint chunks), and its
 * page-locked buffers are allocated by those threads.  Returns the number of CPUs (0 = unbound: single-socket machine, node unknown, DABHIP_NUMA=0)
 * and writes up to cap of them. */
int dabhip_multi_slice_cpus(const dabhip_multi *m, int slice, int32_t *cpus, int cap, int *numa_node);
int64_t dabhip_multi_decode(dabhip_multi *m, const uint8_t *const *iq, const size_t *nbytes, int nstreams, int on_device);
int64_t dabhip_multi_eti_count(const dabhip_multi *m, int stream);
uint32_t dabhip_multi_stream_status(const dabhip_multi *m, int stream);   /* as dabhip_engine_stream_status, global stream index */
int64_t dabhip_multi_eti_read(dabhip_multi *m, int stream, uint8_t *dst, int64_t cap_frames);
int64_t dabhip_multi_stream_log(dabhip_multi *m, int stream, char *buf, int64_t cap);   /* as dabhip_engine_stream_log, global stream index */
int64_t dabhip_multi_eti_drain(dabhip_multi *m, dabhip_eti_sink sink, void *user);   /* all frames, global stream order */
int dabhip_multi_trace(const dabhip_multi *m, int stream, int32_t *ints6, double *ffs, int cap_calls);
/* The engine of one slice, for the per-engine queries (stage times, guard statistics); owned by the multi object. */
dabhip_engine *dabhip_multi_engine(dabhip_multi *m, int slice);
/* Wall clock (ms) of the last decode: the whole call (slice < 0) or one slice's own decode on its host thread. */
float dabhip_multi_wall_ms(const dabhip_multi *m, int slice);
int dabhip_multi_set_afc(dabhip_multi *m, int enable);
int dabhip_multi_set_soft(dabhip_multi *m, int enable);
int dabhip_multi_set_parity_guard(dabhip_multi *m, int level);
int dabhip_multi_set_fused(dabhip_multi *m, int enable);
int dabhip_multi_set_subchannels(dabhip_multi *m, const int32_t *ids, int n);

/* ---- streaming sessions ---------------------------------------------
 * The batch engine over UNBOUNDED streams: B parallel captures fed segment by segment (stdin, a socket, a file too
 * large for the device).  The state dab2eti keeps between calls -- FIFO backlog and stale frame tail (sdr_fifo.c),
 * timing corrections (input_sdr.c:36-112), lock counter and the 16-CIF ring (dab.c:35-98) -- is carried on the device and
 * in the host control plane, so the ETI frames of all segments, concatenated, are byte-identical to one
 * dabhip_engine_decode() of the whole capture, whatever the segment sizes (they need not be multiples of 262144).
 * After each feed the frames of THAT segment are read with the eti_* calls below. */
typedef struct dabhip_stream dabhip_stream;
dabhip_stream *dabhip_stream_create(int device, int nstreams);
/* the same with the host side chosen by the caller: host_threads of the control-plane pool (0 = automatic), CPUs to bind them to (ncpus = 0: unbound) */
dabhip_stream *dabhip_stream_create_on_cpus(int device, int nstreams, int host_threads, const int32_t *cpus, int ncpus);
void dabhip_stream_destroy(dabhip_stream *s);
/* Append nbytes[b] bytes to stream b (host pointers, or device pointers when on_device != 0) and decode every
 * 262144-byte call that became complete.  Returns the ETI frames produced by this segment, <0 on error. */
int64_t dabhip_stream_feed(dabhip_stream *s, const uint8_t *const *iq, const size_t *nbytes, int on_device);
/* Overlap of upload and decode: hand over the segment AFTER the one about to be fed.  Its upload starts at once on a stream
 * of its own and runs while dabhip_stream_feed decodes the segment before it; a later dabhip_stream_feed with the SAME
 * pointers and sizes consumes it.  Order of calls: prefetch(0) feed(0)  -or-  feed(0); then prefetch(k+1) feed(k) ...; at most
 * two segments may be waiting.  Host segments must be page-locked (dabhip_host_alloc) for the copy to run asynchronously and must
 * stay untouched until the feed that consumes them has returned.  Returns 0, <0 on error. */
int dabhip_stream_prefetch(dabhip_stream *s, const uint8_t *const *iq, const size_t *nbytes, int on_device);
/* The same session over streams that LIVE in device memory, read in place (no copy into windows): base[b][x] (device memory) is byte x of stream b
 * counted from the session's start, avail[b] the bytes that are there now (it only grows).  The caller keeps the bytes from
 * dabhip_stream_need_from(s, b) on where they are -- a linear buffer that is appended to -- and may recycle everything below.  Not to be mixed
 * with dabhip_stream_feed / _prefetch on one session.  Returns the ETI frames produced by the new bytes, <0 on error. */
int64_t dabhip_stream_feed_resident(dabhip_stream *s, const uint8_t *const *base, const size_t *avail);
int64_t dabhip_stream_need_from(const dabhip_stream *s, int stream);   /* oldest stream byte a later segment may still read; <0: bad argument */
int64_t dabhip_stream_eti_count(const dabhip_stream *s, int stream);
int dabhip_stream_stage_ms(const dabhip_stream *s, const char **names, float *ms, int cap);   /* of the segment fed last; names as dabhip_engine_stage_ms */
uint32_t dabhip_stream_status(const dabhip_stream *s, int stream);        /* as dabhip_engine_stream_status, sticky over the session's segments */
int64_t dabhip_stream_log(dabhip_stream *s, int stream, char *buf, int64_t cap);   /* as dabhip_engine_stream_log: the operator messages since the last call */
int64_t dabhip_stream_eti_read(dabhip_stream *s, int stream, uint8_t *dst, int64_t cap_frames);
int64_t dabhip_stream_eti_drain(dabhip_stream *s, dabhip_eti_sink sink, void *user);
/* as dabhip_engine_eti_fetch / _wait, for the frames of the segment fed last: its download overlaps the next segment's upload and decode */
int64_t dabhip_stream_eti_fetch(dabhip_stream *s, uint8_t *dst, int64_t cap_frames);
int dabhip_stream_eti_fetch_wait(dabhip_stream *s);
int dabhip_stream_set_afc(dabhip_stream *s, int enable);
int dabhip_stream_set_subchannels(dabhip_stream *s, const int32_t *ids, int n);   /* before the first segment only */
int dabhip_stream_set_soft(dabhip_stream *s, int enable);   /* before the first segment only */
int dabhip_stream_set_parity_guard(dabhip_stream *s, int level);   /* see dabhip_engine_set_parity_guard */
int dabhip_stream_set_sync_speculation(dabhip_stream *s, int mode);   /* default -1, see dabhip_engine_set_sync_speculation */
/* ---- sessions over several devices of one node -----------------------------------------------------------------------
 * dab2eti.c:60-130,237 is a session on ONE device: calls arrive for ever from one demod thread.  B independent unbounded streams shard like a batch
 * does: the streams are dealt ONCE, at creation, to the listed devices in contiguous slices (the rule of dabhip_multi_plan: slice i of n takes
 * B / n streams, the first B mod n slices one more), every slice is a complete dabhip_stream session on its device with its own host thread, and
 * every call below is that call made on all slices at once with each slice's part of the pointer arrays.  No collective, no peer access.  The
 * frames of a segment are byte-identical to those of ONE dabhip_stream session over all B streams (and, concatenated over the segments, to one
 * dabhip_engine_decode of the whole captures), in global stream order.  A device may be listed more than once (each entry is its own slice).
 * iq[b] / base[b]: host pointers (page-locked for asynchronous uploads: dabhip_host_alloc memory is visible to every device), or -- on_device /
 * feed_resident -- device pointers that live on the device of stream b's slice (dabhip_multi_stream_slice_of).
 * Call sequence of a host-fed pipeline (what `dab2eti-hip --stream --devices 0-7` does):
 *     prefetch(seg 0); loop k: prefetch(seg k + 1); n = feed(seg k); eti_fetch(out[k & 1], n); ... a writer thread: eti_fetch_wait(); write(out[k & 1]). */
typedef struct dabhip_multi_stream dabhip_multi_stream;
dabhip_multi_stream *dabhip_multi_stream_create(const int *devices, int n, int nstreams);
void dabhip_multi_stream_destroy(dabhip_multi_stream *m);
int dabhip_multi_stream_slices(const dabhip_multi_stream *m);
int dabhip_multi_stream_streams(const dabhip_multi_stream *m);
/* slice index of `stream` (< 0: bad argument); its device, and the slice's first stream and stream count */
int dabhip_multi_stream_slice_of(const dabhip_multi_stream *m, int stream, int *device, int *first, int *count);
/* one slice's session, for the per-session queries (dabhip_stream_stage_ms); owned by the multi object, NULL for an empty slice */
dabhip_stream *dabhip_multi_stream_session(dabhip_multi_stream *m, int slice);
int dabhip_multi_stream_prefetch(dabhip_multi_stream *m, const uint8_t *const *iq, const size_t *nbytes, int on_device);
int64_t dabhip_multi_stream_feed(dabhip_multi_stream *m, const uint8_t *const *iq, const size_t *nbytes, int on_device);
int64_t dabhip_multi_stream_feed_resident(dabhip_multi_stream *m, const uint8_t *const *base, const size_t *avail);
int64_t dabhip_multi_stream_need_from(const dabhip_multi_stream *m, int stream);
int64_t dabhip_multi_stream_eti_count(const dabhip_multi_stream *m, int stream);
uint32_t dabhip_multi_stream_status_of(const dabhip_multi_stream *m, int stream);   /* as dabhip_stream_status, global stream index (dabhip_multi_stream_status is the batch object's) */
int64_t dabhip_multi_stream_eti_read(dabhip_multi_stream *m, int stream, uint8_t *dst, int64_t cap_frames);
int64_t dabhip_multi_stream_log_of(dabhip_multi_stream *m, int stream, char *buf, int64_t cap);   /* as dabhip_stream_log, global stream index */
int64_t dabhip_multi_stream_eti_drain(dabhip_multi_stream *m, dabhip_eti_sink sink, void *user);   /* frames of the segment fed last, global stream order */
/* all frames of the segment fed last in global stream order, one asynchronous download per slice into dst (page-locked); _wait: they have landed */
int64_t dabhip_multi_stream_eti_fetch(dabhip_multi_stream *m, uint8_t *dst, int64_t cap_frames);
int dabhip_multi_stream_eti_fetch_wait(dabhip_multi_stream *m);
int dabhip_multi_stream_set_afc(dabhip_multi_stream *m, int enable);
int dabhip_multi_stream_set_soft(dabhip_multi_stream *m, int enable);                    /* before the first segment only */
int dabhip_multi_stream_set_parity_guard(dabhip_multi_stream *m, int level);
int dabhip_multi_stream_set_sync_speculation(dabhip_multi_stream *m, int mode);
int dabhip_multi_stream_set_subchannels(dabhip_multi_stream *m, const int32_t *ids, int n);   /* before the first segment only */

/* Page-locked host memory for segments: fill the next one while the current one decodes (double buffering). */
void *dabhip_host_alloc(size_t nbytes);
void dabhip_host_free(void *p);

/* Streaming rates of the device, for reading the K2 roofline figure against what a bare kernel reaches (not on the data path):
 * gbs[0] fill, gbs[1] copy, gbs[2] K2's mix (1 byte read per 4 written), GB/s of bytes moved, over buffers of `bytes` bytes. */
int dabhip_stream_ceiling(int device, size_t bytes, int reps, double *gbs);

/* Which physical device an index is: its PCI bus id ("0000:c1:00.0", cap_bus >= 16) and name.  bench.py records both per rank and refuses to report
 * a multi-rank figure whose ranks did not sit on distinct devices.  0, <0: no such device. */
int dabhip_device_identity(int device, char *bus_id, int cap_bus, char *name, int cap_name);

/* Device memory for callers without a GPU runtime of their own: the batch entries (dabhip_engine_decode with on_device,
 * dabhip_synth_generate_device, the stage entries) take plain device pointers.  copy: to_device != 0: host -> device. */
void *dabhip_device_alloc(size_t nbytes, int device);
void dabhip_device_free(void *p);
int dabhip_device_copy(void *dst, const void *src, size_t nbytes, int to_device);

/* What the OFDM stage of the LAST dabhip_engine_decode left for transmission frame `tf` (0-based among the frames `stream`
 * demodulated in that decode) -- the content of tf->fic_symbols_demapped[3][3072] and tf->msc_symbols_demapped[72][3072]
 * (dab.h:27-33, filled at input_sdr.c:146-162) as the batch path holds it: 9216 + 221184 values, 0 / 1 for hard decisions, the
 * signed 4-bit values (-7 .. 7) with soft decisions on.  Host arrays.  0, <0 on error. */
int dabhip_engine_demapped_tf(dabhip_engine *e, int stream, int tf, int8_t *fic, int8_t *msc);

/* Per sdr_demod call trace of one stream, for parity with the reference's state after each
 * call: {ok, frame_read, coarse_timeshift, fine_timeshift, coarse_freq_shift, fifo_count}
 * as int32[6] per call plus fine_freq_shift as double per call. */
int dabhip_engine_trace(const dabhip_engine *e, int stream, int32_t *ints6, double *ffs, int cap_calls);
/* With the software AFC on: the re-tuning (Hz, relative to nominal) in effect during each call -- sdr->frequency of dab2eti.c:76-103 as the NCO
 * applied it to that call's samples; 0 everywhere in parity mode. */
int dabhip_engine_trace_nco(const dabhip_engine *e, int stream, int32_t *nco_hz, int cap_calls);

/* Timing of the stages of the last decode (milliseconds, HIP events on the engine's stream).
 * names: "sync", "fft", "demap", "fic", "control", "gather", "viterbi", "eti", then host wall-clock
 * phases "host_setup", "host_frames", "host_worklist" and the total "wall"; then, for a host-fed decode (on_device == 0),
 * "h2d" (ms of the IQ upload, HIP events), "h2d_mbytes" (10^6 bytes uploaded) and "h2d_pinned_mbytes" (how much of that came
 * straight from page-locked memory; the rest went through the engine's staging ring); and "sync_fp64_calls": calls whose coarse
 * frequency arg-max the single-precision first pass of K1's verification left to the fp64 pass (a count, not a time); "sync_spec_calls": calls of
 * K1's chain whose estimators came out of the look-ahead pass's table (dabhip_engine_set_sync_speculation; 0 when the plain chain ran).
 * Returns number of entries written. */
int dabhip_engine_stage_ms(const dabhip_engine *e, const char **names, float *ms, int cap);
/* Per-launch statistics of the OFDM FFT kernel in the last decode: number of launches,
 * transmission frames transformed, total kernel milliseconds (HIP events). */
int dabhip_engine_fft_stats(const dabhip_engine *e, int64_t *launches, int64_t *tfs, double *ms);
/* K2 (ofdm_fft_kernel) alone, `reps` times over the transmission frames of the LAST decode: same resident IQ, same frame
 * list, same launch shape as the two-kernel stage (one untimed pass first).  HIP events on the engine's stream around
 * every launch: number of timed launches, TFs transformed, total kernel milliseconds.  This is the roofline
 * measurement for input_sdr.c:115-130 whichever OFDM variant the decode itself used. */
int dabhip_engine_fft_roofline(dabhip_engine *e, int reps, int64_t *launches, int64_t *tfs, double *ms);

/* ---- stage entries (parity tests) ------------------------------------------------------ */
/* OFDM FFT stage alone (replaces input_sdr.c:115-130): nframes contiguous cu8 frames of
 * 393216 bytes each (device pointers when on_device) -> fftshifted complex64 spectra
 * [nframes][76][2048][2].  `reps` > 1 repeats the launch for timing; kernel_ms (optional)
 * receives the mean duration of one launch from HIP events. */
int dabhip_stage_ofdm_fft(dabhip_engine *e, const uint8_t *frames, int nframes, float *spectra, int on_device,
                          int reps, float *kernel_ms);
/* DQPSK + demap + frequency de-interleave (input_sdr.c:132-162): spectra of nframes ->
 * fic bytes [nframes][9216] and msc bytes [nframes][221184] (host pointers). */
int dabhip_stage_demap(dabhip_engine *e, const float *spectra, int nframes, uint8_t *fic, uint8_t *msc);
/* Audit of the fp32 OFDM stage against fp64 (test / calibration tool): nframes contiguous cu8 frames through K2 + K2b
 * (guard_on: with the parity guard), then fp64 transforms of every symbol on the GPU and a comparison of all 230,400
 * decisions per frame.  out8 = {decisions, disagreements with fp64, disagreements on carriers the guard rule does not flag,
 * decisions the rule flags, max |X32 - X64| / sqrt(sum |x|^2), max error of Re/Im(cur conj(prev)) / (|cur|_1 s(l-1) + |prev|_1 s(l)),
 * max residual product error / (|cur|_1 |prev|_1), entries the demapper listed}. */
int dabhip_stage_decision_audit(dabhip_engine *e, const uint8_t *frames, int nframes, int on_device, int guard_on, double *out8);
/* The same audit of the kernel the DEFAULT decode runs (round 5): ofdm_demap_kernel's guarded build, through a fourth build of the same source that also
 * stores the bins it holds and the products it decides on; frames laid out as a decode lays them out.  out10[0..8) as above (the errors are those of the
 * fused kernel's own bins and products, [7] = entries it listed); out10[8] = 1 when the shipping build, run on the same frames, left exactly the same bits
 * (guard_on = 0 only; -1 otherwise), out10[9] = 1 when it listed the same number of decisions. */
int dabhip_stage_decision_audit_fused(dabhip_engine *e, const uint8_t *frames, int nframes, int on_device, int guard_on, double *out10);
/* FIC decode of nframes TFs (fic.c:160-208): 9216 demapped bytes each -> 12x32 FIB bytes + 12 flags each. */
int dabhip_stage_fic_decode(dabhip_engine *e, const uint8_t *fic, int nframes, uint8_t *fibs, uint8_t *crc_ok);

/* ---- host-side control plane, callable without a GPU (used by the CPU test-suite) ------ */
/* FIG 0/0, 0/1, 0/2 parse of the 12 FIBs of one TF (fib_decode, fic.c:132-147).
 * hdr3 = {EId, CIFCount_hi, CIFCount_lo}; sub = 64 rows of
 * {id, slForm, uep_index, start_cu, size, bitrate, protlev, ASCTy}. */
int dabhip_host_parse_fibs(const uint8_t *fibs, const uint8_t *crc_ok, int32_t *hdr3, int32_t *sub);
/* ETI(NI) header (init_eti, misc.c:153-213) from the same table layout; returns its length. */
int dabhip_host_eti_header(const int32_t *hdr3, const int32_t *sub, uint8_t *out, int cap);
/* Lock FSM + 16-CIF ring + header sequence (dab_process_frame, dab.c:35-98) over ntf TFs
 * given their decoded FIBs (ntf x 384 bytes) and CRC flags (ntf x 12).  For each ETI frame
 * it would emit, writes first_cif[i] (linear index of the oldest CIF) and the header bytes
 * (272-byte rows) + lengths.  Returns the number of ETI frames. */
int dabhip_host_control_replay(const uint8_t *fibs, const uint8_t *crc_ok, int ntf, int32_t *first_cif,
                               uint8_t *headers, int32_t *header_len, int cap_frames);
/* the operator messages (see dabhip_engine_stream_log) of the calling thread's last dabhip_host_control_replay */
int64_t dabhip_host_control_replay_log(char *buf, int64_t cap);

/* The placement rule by itself (no GPU, no sysfs): slice i sits on NUMA node slice_node[i] (< 0: unknown); node_cpulist[n] is node n's CPU list in
 * the kernel's notation ("0-63,128-191").  cpu_slice[c] = the slice CPU c is given to, -1 = none; the slices of a node get disjoint contiguous
 * chunks of its list.  Returns the number of slices that got CPUs. */
int dabhip_host_placement_plan(const int32_t *slice_node, int nslices, const char *const *node_cpulist, int nnodes, int32_t *cpu_slice, int ncpu);
/* What the host pools are sized from (round 5): returns min(CPUs in the process's affinity mask, CFS quota of its cgroup), at least 1 -- NOT the
 * machine's thread count, which says nothing inside a container (DABHIP_CPUS=n overrides it); the two inputs come back in *affinity_cpus and
 * *cfs_quota_cpus (0 = no quota).  No GPU call.  dab2eti.c:237 has one demod thread and no pool to size. */
int dabhip_host_cpu_budget(int *affinity_cpus, int *cfs_quota_cpus);

/* The constant tables the kernels are built from (dab_tables.hpp generates them from the ETSI rules), so that tests can
 * hold them against the reference's literal arrays: which = 0: the 64 UEP profiles of ueptable (dab_tables.c:16-81) as rows
 * {bitrate, size, protection level, L1..L4, PI1..PI4} (PI as in ETSI, 1..24; the reference stores PI - 1); 1: pvec
 * (dab_tables.c:102-127), 24 x 32 flags; 2: rev_freq_deint_tab (dab_tables.c:164), 1536 entries; 3: the phase reference
 * symbol (sdr_prstab.c) as quarter turns, 1536 entries.  Returns the number of values written, <0 on error. */
int dabhip_host_table(int which, int32_t *out, int cap);

/* The FIFO of sdr_demod (cbWrite / sdr_read_fifo, sdr_fifo.c:26-61; input_sdr.c:36-55) as the sync-scan kernel keeps it:
 * in closed form over the resident stream.  One call = one sdr_demod call that appends chunk_bytes (input_buffer_len; 262144 from librtlsdr),
 * entered with the timing corrections the previous processed frame left in sdr->coarse_timeshift / fine_timeshift.  Returns 0 = nothing read
 * (fewer than 1.5 TF queued), 1 = the first frame, read and discarded (input_sdr.c:51-55), 2 = a frame read for
 * processing; <0 = error.  What sdr->buffer (393216 bytes) holds afterwards comes back in two parts.  Its last 1536 bytes -- the most a
 * negative time shift leaves unread (sdr_sync.c:197-201, sdr_fifo.c:56-59) -- as BYTES in tail[1536] (needs stream: the bytes fed so far;
 * null = not tracked).  Everything below as a view: buffer positions [seg_end[i-1], seg_end[i]) hold stream bytes seg_src[i] + position
 * (seg_src < 0: the calloc'ed zero bytes); 12 entries each. */
typedef struct dabhip_fifo dabhip_fifo;
dabhip_fifo *dabhip_host_fifo_new(void);
void dabhip_host_fifo_free(dabhip_fifo *f);
int dabhip_host_fifo_call(dabhip_fifo *f, int32_t coarse_timeshift, int32_t fine_timeshift, int32_t chunk_bytes, const uint8_t *stream,
                          int32_t *nseg, int32_t *seg_end, int64_t *seg_src, int32_t *fifo_count, uint8_t *tail);

/* ncalls further calls of 262144 bytes for a FIFO that reads without any time shift, counters only: what K1's look-ahead pass (sync_ahead_kernel)
 * predicts the stream's read pointer with, in closed form (the counters repeat every three calls).  Equals ncalls x dabhip_host_fifo_call(f, 0, 0,
 * 262144, ...) in `fed` (bytes appended so far) and `consumed` (stream offset of the read pointer); the frame-buffer view is left as it was.
 * Refused (-1) while a shift is pending or the first frame has not been dropped yet. */
int dabhip_host_fifo_skip_unshifted(dabhip_fifo *f, int32_t ncalls, int64_t *fed, int64_t *consumed);

/* ---- synthetic Mode-I modulator (host only) --------------------------------------------- */
typedef struct dabhip_subch_cfg {
  int32_t id;          /* SubChId 0..63 */
  int32_t start_cu;    /* 0..863 */
  int32_t slform;      /* 0 = UEP (uep_index), 1 = EEP (eep_protlev, size_cu) */
  int32_t uep_index;   /* 0..63, ETSI Table 7 */
  int32_t eep_protlev; /* option<<2 | level: 0..3 = 1-A..4-A, 4..7 = 1-B..4-B */
  int32_t size_cu;     /* EEP only */
} dabhip_subch_cfg;

typedef struct dabhip_reconf_cfg {
  int32_t at_cif;        /* logical CIF index from which the MSC carries this multiplex; 0 = entry unused */
  int32_t fic_lead;      /* the FIC signals it this many CIFs earlier (>= 0) */
  int32_t nsub;
  int32_t pad;
  dabhip_subch_cfg sub[64];
} dabhip_reconf_cfg;

/* Order of application to the modulator's complex samples x[n] (n counts from the capture's first sample, skipped ones included):
 * echoes, fading, carrier offset (cfo_hz above), sample-rate offset, I/Q imbalance; then skip_samples, noise and the cu8 quantisation as before. */
typedef struct dabhip_channel_cfg {
  double sro_ppm;          /* the receiver's sample clock against the transmitter's: output sample m is the signal at input position m (1 + ppm 1e-6)
                              (windowed-sinc interpolation, 16 taps): > 0 = frames arrive SHORTER than 196,608 samples (negative time shifts) */
  int32_t echo_delay[2];   /* two-ray / three-ray multipath: x[n] += gain e^(2 pi i (phase + doppler n / 2.048e6)) x[n - delay]; delay in samples, 0 = no echo, <= 2047 */
  double echo_gain[2];     /* linear, relative to the direct path */
  double echo_phase[2];    /* turns */
  double echo_doppler_hz[2];
  double fade_depth;       /* slow flat fading: amplitude 1 - depth (1 - cos(2 pi fade_hz t)) / 2, 0 <= depth < 1 */
  double fade_hz;
  double iq_gain_db;       /* receiver imbalance: Q rail gain over I rail */
  double iq_phase_deg;     /* quadrature error: Q' = g (Q cos phi + I sin phi) */
} dabhip_channel_cfg;

typedef struct dabhip_synth_cfg {
  uint32_t eid;
  int32_t nsub;
  dabhip_subch_cfg sub[64];
  uint64_t seed;          /* payload / filler / noise seed */
  int32_t cif_count0;     /* CIF counter of the first CIF, 0..4999 */
  int32_t skip_samples;   /* drop this many samples from the start (0..196607): unaligned capture */
  double amplitude;       /* LSB per unit carrier of the unnormalised IDFT (1.0 -> ~28 LSB rms per rail) */
  double snr_db;          /* signal/noise power over the 2.048 MHz band; >= 100 -> no noise */
  double cfo_hz;          /* carrier frequency offset applied to the whole capture (0 = none) */
  /* Test vectors for FIC content the presets do not produce (non-standard or hostile FIGs): when fib_patch_len > 0, the THIRD FIB of every CIF
   * from fib_patch_from_cif on carries these bytes (FIGs, then end marker / padding) under a VALID CRC -- what a corrupted FIB that passes the
   * 16-bit CRC looks like to fib_parse (fic.c:47-130).  The MSC content is unaffected.  0-initialised = off. */
  int32_t fib_patch_len;  /* 0..30 */
  int32_t fib_patch_from_cif;
  uint8_t fib_patch[32];
  /* Multiplex reconfigurations (round 5): from logical CIF reconf[k].at_cif on (counted from the capture's first CIF; 0 = unused; ascending) the MSC
   * carries the sub-channels reconf[k].sub[0..nsub) instead, and the FIG 0/1 entries of the FIC announce them from CIF at_cif - fic_lead on.  The
   * reference knows nothing of ETSI's reconfiguration signalling: merge_info (misc.c:14-27) overwrites the slots of the SubChIds it hears and never
   * removes one, on every locked TF BEFORE the oldest CIFs of the ring are emitted (dab.c:64-97) -- frames already in the ring are laid out by the new
   * FIG 0/1.  Whatever that yields is the parity target.  dabhip_synth_payload's slot then indexes the multiplex in force at that CIF. */
  dabhip_reconf_cfg reconf[2];
  /* Channel between modulator and receiver (round 5; all zero = the ideal channel of rounds 1-4, byte-identical captures).  Host generator only:
   * dabhip_synth_generate_device refuses a configuration with any of these set. */
  dabhip_channel_cfg channel;
} dabhip_synth_cfg;

/* preset 0: 12 sub-channels, 1136 kbit/s, 862 CU (the benchmark mix); 1: 4 light sub-channels. */
int dabhip_synth_preset(int preset, dabhip_synth_cfg *cfg);
/* Bytes dabhip_synth_generate writes for ntf transmission frames -- exactly that with channel.sro_ppm == 0; with a sample-rate offset an upper
 * bound (the capacity to pass), the call's return value being the count. */
size_t dabhip_synth_bytes(const dabhip_synth_cfg *cfg, int ntf);
/* Generate ntf transmission frames of cu8 IQ into iq (capacity cap bytes). Returns bytes written or <0. */
int64_t dabhip_synth_generate(const dabhip_synth_cfg *cfg, int ntf, uint8_t *iq, size_t cap);
/* The payload (before energy dispersal) carried by sub-channel slot k in logical CIF n:
 * bitrate*3 bytes.  Returns the byte count, <0 on error. */
int dabhip_synth_payload(const dabhip_synth_cfg *cfg, int cif_index, int slot, uint8_t *out, int cap);
/* The 96 FIB bytes (3 FIBs with CRC) carried by CIF n. */
int dabhip_synth_fibs(const dabhip_synth_cfg *cfg, int cif_index, uint8_t *out96);
/* Device-side modulator (needs a GPU): the ensembles cfgs[0..nstreams) modulated on `device`
 * straight into DEVICE buffers iq[i] of dabhip_synth_bytes(&cfgs[i], ntf) bytes each.  The bit content comes from the
 * same generator as dabhip_synth_generate; samples may differ from the host generator's by one LSB where fp32 and
 * fp64 round apart, and the AWGN uses the same keyed generator.  Returns 0, <0 on error. */
int dabhip_synth_generate_device(const dabhip_synth_cfg *cfgs, int nstreams, int ntf, uint8_t *const *iq, int device);

#ifdef __cplusplus
}
#endif
#endif
