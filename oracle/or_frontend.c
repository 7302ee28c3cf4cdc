/*
 * or_frontend.c — CPU restatement of the RTL-SDR front end of dab2eti
 * (input_sdr.c, sdr_sync.c, sdr_fifo.c) and of the replay loop of dab2eti.c.
 * TEST INFRASTRUCTURE ONLY (see dab_oracle.h).
 *
 * Pinning: the reference calls libfftw3 (double, version unpinned, Makefile:3) at input_sdr.c:91-93,116-119 and
 * sdr_sync.c:93-95,167-169,223-225; libfftw3 is not in this image.  FFTW's published contract is restated by or_dft():
 * out[k] = sum_j in[j] exp(sign 2 pi i jk/n), unnormalised; only signs and arg-maxima of DFT outputs are consumed
 * downstream.  Since round 4 the reference's own input_sdr.c / sdr_sync.c / sdr_fifo.c, unmodified, are built against the
 * FFTW3-API library the image DOES have (AMD's hipFFTW, oracle/_ref/libdabref_frontend.so) and this file is held against
 * that object code call by call and bit by bit (tests/test_gpu_frontend_ref.py, GPU box only).  The byte FIFO (sdr_fifo.c)
 * also builds by itself and pins or_fifo_* on the CPU.
 */
#include "dab_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* mixed radix 2/3 decimation-in-time DFT, double precision */
static void dft_rec(int n, int stride, const double *in, double *out, int sign, const double *tw, int twstride)
{
  int k;
  if (n == 1) { out[0] = in[0]; out[1] = in[1]; return; }
  if (n % 2 == 0) {
    int h = n / 2;
    dft_rec(h, stride * 2, in, out, sign, tw, twstride * 2);
    dft_rec(h, stride * 2, in + 2 * stride, out + 2 * h, sign, tw, twstride * 2);
    for (k = 0; k < h; k++) {
      double wr = tw[2 * k * twstride], wi = sign * tw[2 * k * twstride + 1];
      double br = out[2 * (k + h)], bi = out[2 * (k + h) + 1];
      double tr = br * wr - bi * wi, ti = br * wi + bi * wr;
      double ar = out[2 * k], ai = out[2 * k + 1];
      out[2 * k] = ar + tr; out[2 * k + 1] = ai + ti;
      out[2 * (k + h)] = ar - tr; out[2 * (k + h) + 1] = ai - ti;
    }
  } else {
    int t = n / 3;
    const double c3 = -0.5, s3 = sign * 0.86602540378443864676;
    dft_rec(t, stride * 3, in, out, sign, tw, twstride * 3);
    dft_rec(t, stride * 3, in + 2 * stride, out + 2 * t, sign, tw, twstride * 3);
    dft_rec(t, stride * 3, in + 4 * stride, out + 4 * t, sign, tw, twstride * 3);
    for (k = 0; k < t; k++) {
      double w1r = tw[2 * k * twstride], w1i = sign * tw[2 * k * twstride + 1];
      double w2r = tw[2 * (2 * k) * twstride], w2i = sign * tw[2 * (2 * k) * twstride + 1];
      double ar = out[2 * k], ai = out[2 * k + 1];
      double xr = out[2 * (k + t)], xi = out[2 * (k + t) + 1];
      double yr = out[2 * (k + 2 * t)], yi = out[2 * (k + 2 * t) + 1];
      double br = xr * w1r - xi * w1i, bi = xr * w1i + xi * w1r;
      double cr = yr * w2r - yi * w2i, ci = yr * w2i + yi * w2r;
      double sr = br + cr, si = bi + ci, dr = br - cr, di = bi - ci;
      out[2 * k] = ar + sr; out[2 * k + 1] = ai + si;
      out[2 * (k + t)] = ar + c3 * sr - s3 * di; out[2 * (k + t) + 1] = ai + c3 * si + s3 * dr;
      out[2 * (k + 2 * t)] = ar + c3 * sr + s3 * di; out[2 * (k + 2 * t) + 1] = ai + c3 * si - s3 * dr;
    }
  }
}

void or_dft(int n, const double *in, double *out, int sign)
{
  static double *tw[3];
  static const int sizes[3] = {2048, 1536, 128};
  int which = (n == 2048) ? 0 : (n == 1536) ? 1 : (n == 128) ? 2 : -1, k;
  double *t;
  if (which < 0) {           /* generic size: throw-away twiddles */
    t = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (k = 0; k < n; k++) { t[2 * k] = cos(2 * M_PI * k / n); t[2 * k + 1] = sin(2 * M_PI * k / n); }
    dft_rec(n, 1, in, out, sign, t, 1);
    free(t);
    return;
  }
  if (!tw[which]) {
    t = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (k = 0; k < n; k++) { t[2 * k] = cos(2 * M_PI * k / n); t[2 * k + 1] = sin(2 * M_PI * k / n); }
    tw[which] = t;
  }
  (void)sizes;
  dft_rec(n, 1, in, out, sign, tw[which], 1);
}

/* ------------------------------------------------------------------------- */
/* sdr_sync.c:34-68 */
uint32_t or_coarse_time_sync(const int8_t *real, int force)
{
  static float filt[(OR_TF_SAMPLES - 2656) / 10 + 1];
  const int tnull = 2656;
  float e = 0, minval = 9999999;
  uint32_t minpos = 0;
  int j, k;
  for (k = 0; k < tnull; k += 10) e += (float)abs(real[k]);
  if (e < 5000 && !force) return 0;
  for (j = 0; j < (OR_TF_SAMPLES - tnull) / 10; j++) filt[j] = 0;
  for (j = 0; j < OR_TF_SAMPLES - tnull; j += 10)
    for (k = 0; k < tnull; k += 10) filt[j / 10] += (float)abs(real[j + k]);
  for (j = 0; j < (OR_TF_SAMPLES - tnull) / 10; j++)
    if (filt[j] < minval) { minval = filt[j]; minpos = (uint32_t)j * 10; }
  return minpos * 2;
}

static void prs_value(int k, double *re, double *im)
{
  static const double c[4] = {1, 0, -1, 0}, s[4] = {0, 1, 0, -1};
  int p = or_prs_phase()[k];
  *re = c[p]; *im = s[p];
}

/* sdr_sync.c:71-202 */
int32_t or_fine_time_sync(const double *frame)
{
  static double spec[2048 * 2], conv[1536 * 2], corr[1536 * 2];
  uint32_t maxpos = 0;
  float maxval = -99999;
  int i;
  or_dft(2048, frame + 2 * (2656 + 504), spec, -1);
  for (i = 0; i < 1536; i++) {
    int bin = (i < 768) ? i + 1280 : i - 765;     /* sdr_sync.c:133-142 (the 2-bin quirk) */
    double pr, pi, xr = spec[2 * bin], xi = spec[2 * bin + 1];
    prs_value(i, &pr, &pi);
    pi = -pi;                                      /* conjugate, sdr_sync.c:110-113 */
    conv[2 * i] = xr * pr - xi * pi;
    conv[2 * i + 1] = xr * pi + xi * pr;
  }
  or_dft(1536, conv, corr, +1);
  for (i = 0; i < 1536; i++) {
    float v = (float)sqrt(corr[2 * i] * corr[2 * i] + corr[2 * i + 1] * corr[2 * i + 1]);
    if (v > maxval) { maxpos = (uint32_t)i; maxval = v; }
  }
  if (maxpos < 768) return (int32_t)(maxpos * 2 + 16);
  return (int32_t)((maxpos - 1536) * 2);           /* unsigned wrap -> negative, sdr_sync.c:199 */
}

/* sdr_sync.c:205-258 */
int32_t or_coarse_freq_sync(const double *sym)
{
  double conv[128 * 2], corr[128 * 2];
  float gmax = -99999;
  int gpos = 0, k, s;
  for (k = -14; k <= 14; k++) {
    float maxval = -99999;
    for (s = 0; s < 128; s++) {
      double pr, pi, xr = sym[2 * (14 + k + 256 + s)], xi = sym[2 * (14 + k + 256 + s) + 1];
      prs_value(14 + s, &pr, &pi);
      conv[2 * s] = pr * xr + pi * xi;
      conv[2 * s + 1] = pr * xi - pi * xr;
    }
    or_dft(128, conv, corr, +1);
    for (s = 0; s < 128; s++) {
      float v = (float)sqrt(corr[2 * s] * corr[2 * s] + corr[2 * s + 1] * corr[2 * s + 1]);
      if (v > maxval) maxval = v;
    }
    if (maxval > gmax) { gmax = maxval; gpos = k; }
  }
  return gpos;
}

/* sdr_sync.c:259-302 (the fine_timeshift argument is overwritten with 0 at :270) */
double or_fine_freq_corr(const double *frame)
{
  double mean = 0;
  int i;
  for (i = 0; i < 504; i++) {
    const double *l = frame + 2 * (2656 + 2048 + i), *r = frame + 2 * (2656 + i);
    double re = l[0] * r[0] + l[1] * r[1];
    double im = -l[0] * r[1] + l[1] * r[0];
    mean += atan2(im, re);
  }
  mean /= 504;
  return mean / (2 * M_PI) * 1000;
}

/* ------------------------------------------------------------------------- */
/* sdr_fifo.c:26-61 in closed form over a flat ring */
#define FIFO_SIZE (OR_TF_BYTES * 4)

struct or_sdr {
  uint8_t *fifo;
  uint32_t start, count;
  uint8_t buffer[OR_TF_BYTES];
  int8_t real[OR_TF_SAMPLES], imag[OR_TF_SAMPLES];
  double *frame;          /* [196608][2] */
  double *symbols;        /* [76][2048][2] */
  int32_t coarse_timeshift, fine_timeshift, coarse_freq_shift;
  double fine_freq_shift;
  int32_t startup_delay, force_timesync;
  int32_t last_ok, last_read;
  /* software AFC (or_replay_afc below): the tuner of dab2eti.c:76-103 replaced by an NCO */
  int32_t afc, tuner_hz, nco_hz_used;
  uint32_t rng;
};

struct or_sdr *or_sdr_new(void)
{
  struct or_sdr *s = (struct or_sdr *)calloc(1, sizeof *s);
  s->fifo = (uint8_t *)calloc(FIFO_SIZE, 1);
  s->frame = (double *)calloc((size_t)OR_TF_SAMPLES * 2, sizeof(double));
  s->symbols = (double *)calloc((size_t)76 * 2048 * 2, sizeof(double));
  return s;
}
void or_sdr_free(struct or_sdr *s) { free(s->fifo); free(s->frame); free(s->symbols); free(s); }

static void fifo_write(struct or_sdr *s, const uint8_t *p, int n)   /* sdr_fifo.c:26-35 */
{
  int i;
  for (i = 0; i < n; i++) {
    s->fifo[(s->start + s->count) % FIFO_SIZE] = p[i];
    if (s->count == FIFO_SIZE) s->start = (s->start + 1) % FIFO_SIZE; else s->count++;
  }
}
static void fifo_read1(struct or_sdr *s, uint8_t *dst)              /* sdr_fifo.c:37-41 */
{
  *dst = s->fifo[s->start];
  s->start = (s->start + 1) % FIFO_SIZE;
  s->count--;
}
static void fifo_read_shifted(struct or_sdr *s, uint32_t bytes, int32_t shift, uint8_t *buffer) /* sdr_fifo.c:43-61 */
{
  int32_t i;
  uint32_t j;
  if (shift > 0) {
    for (i = 0; i < shift; i++) if (s->count) fifo_read1(s, &buffer[i]);
    for (j = 0; j < bytes; j++) if (s->count) fifo_read1(s, &buffer[j]);
  } else {
    for (j = 0; j < bytes + shift; j++) fifo_read1(s, &buffer[j]);
  }
}

/* input_sdr.c:27-165 */
int or_sdr_demod(struct or_sdr *s, const uint8_t *chunk, int len, uint8_t *fic_bits, uint8_t *msc_bits)
{
  const uint16_t *rev = or_rev_freq_deint_tab();
  double tmp[2048 * 2];
  int i, j;
  s->last_ok = 0;
  s->last_read = 0;
  s->coarse_freq_shift = 0;
  s->nco_hz_used = s->afc ? s->tuner_hz : 0;
  fifo_write(s, chunk, len);
  if (s->count < OR_TF_SAMPLES * 3) return 0;
  fifo_read_shifted(s, OR_TF_BYTES, s->coarse_timeshift + s->fine_timeshift, s->buffer);
  s->last_read = 1;
  if (s->startup_delay <= 0) { s->startup_delay++; return 0; }
  for (j = 0; j < OR_TF_BYTES; j += 2) {
    s->real[j / 2] = (int8_t)(s->buffer[j] - 127);
    s->imag[j / 2] = (int8_t)(s->buffer[j + 1] - 127);
  }
  s->coarse_timeshift = (int32_t)or_coarse_time_sync(s->real, s->force_timesync);
  s->force_timesync = 0;
  if (s->coarse_timeshift) return 0;
  for (j = 0; j < OR_TF_SAMPLES; j++) { s->frame[2 * j] = s->real[j]; s->frame[2 * j + 1] = s->imag[j]; }
  if (s->afc && s->tuner_hz != 0) {
    /* NOT in the reference (its tuner moves, dab2eti.c:76-103; a file has none): the frame de-rotated by exp(-2 pi i f n / fs), n = sample index in
     * the frame buffer, f = the re-tuning accumulated so far.  The null-symbol test above ran on the raw samples, as in the product. */
    for (j = 0; j < OR_TF_SAMPLES; j++) {
      const double a = -2.0 * M_PI * (double)s->tuner_hz * (double)j / 2048000.0, c = cos(a), sn = sin(a);
      const double xr = s->frame[2 * j], xi = s->frame[2 * j + 1];
      s->frame[2 * j] = xr * c - xi * sn;
      s->frame[2 * j + 1] = xr * sn + xi * c;
    }
  }
  s->fine_timeshift = or_fine_time_sync(s->frame);
  /* input_sdr.c:86-88 is dead code: coarse_freq_shift was zeroed above */
  or_dft(2048, s->frame + 2 * (2656 + 505 + s->fine_timeshift), tmp, -1);
  for (i = 0; i < 2048; i++) {
    s->symbols[2 * i] = tmp[2 * ((i + 1024) & 2047)];
    s->symbols[2 * i + 1] = tmp[2 * ((i + 1024) & 2047) + 1];
  }
  s->coarse_freq_shift = or_coarse_freq_sync(s->symbols);
  if (abs(s->coarse_freq_shift) > 1) { s->force_timesync = 1; return 0; }
  s->fine_freq_shift = or_fine_freq_corr(s->frame);
  for (i = 0; i < 76; i++) {
    double *sym = s->symbols + (size_t)i * 4096;
    or_dft(2048, s->frame + 2 * (2656 + 2552 * i + 504), tmp, -1);
    for (j = 0; j < 2048; j++) {
      sym[2 * j] = tmp[2 * ((j + 1024) & 2047)];
      sym[2 * j + 1] = tmp[2 * ((j + 1024) & 2047) + 1];
    }
  }
  for (j = 1; j < 76; j++) {
    uint8_t *dst = (j < 4) ? fic_bits + (j - 1) * 3072 : msc_bits + (j - 4) * 3072;
    const double *cur = s->symbols + (size_t)j * 4096, *prev = cur - 4096;
    int k = 0;
    for (i = 0; i < 2048; i++) {
      if (i > 255 && i != 1024 && i < 1793) {
        double cr = cur[2 * i], ci = cur[2 * i + 1], pr = prev[2 * i], pi = prev[2 * i + 1];
        double den = pr * pr + pi * pi;
        double dre = (cr * pr + ci * pi) / den;      /* input_sdr.c:135-138 */
        double dim = (cr * pi - ci * pr) / den;      /* input_sdr.c:139-143 (sign as stored) */
        int kk = rev[k++];
        dst[kk] = (dre > 0) ? 0 : 1;                 /* input_sdr.c:157-158 */
        dst[1536 + kk] = (dim > 0) ? 1 : 0;
      }
    }
  }
  s->last_ok = 1;
  return 1;
}

void or_sdr_get_trace(const struct or_sdr *s, struct or_sdr_trace *t)
{
  t->ok = s->last_ok;
  t->read_frame = s->last_read;
  t->coarse_timeshift = s->coarse_timeshift;
  t->fine_timeshift = s->fine_timeshift;
  t->coarse_freq_shift = s->coarse_freq_shift;
  t->fifo_count = (int32_t)s->count;
  t->fine_freq_shift = s->fine_freq_shift;
}
const double *or_sdr_symbols(const struct or_sdr *s) { return s->symbols; }
const uint8_t *or_sdr_buffer(const struct or_sdr *s) { return s->buffer; }
const double *or_sdr_frame(const struct or_sdr *s) { return s->frame; }

/* ------------------------------------------------------------------------- */
/* The tuner feedback of demod_thread_fn, dab2eti.c:76-103, after EVERY sdr_demod call (also those that produced no frame: coarse_freq_shift is 0
 * then, fine_freq_shift stale), on the accumulated re-tuning instead of sdr->frequency:
 *   |coarse| > 1            -> -+1000 Hz                                   (dab2eti.c:77-85)
 *   |coarse| == 1           -> -+ (rand() % 1000) Hz                       (:87-97; rand() is the C library's: restated with the LCG the product uses,
 *                                                                            x <- 1103515245 x + 12345, step = (x >> 16) % 1000, x0 = 1 = srand's default seed)
 *   coarse == 0 and abs(fine) > 50  -> + fine / 3                          (:98-103; abs() is the INT one: the double is truncated first; the sum is
 *                                                                            stored into the unsigned sdr->frequency: floor) */
void or_sdr_set_afc(struct or_sdr *s, int on) { s->afc = on; s->rng = 1; }
int32_t or_sdr_nco_hz(const struct or_sdr *s) { return s->nco_hz_used; }   /* the frequency the LAST call's samples were de-rotated by */
void or_afc_step(struct or_sdr *s)
{
  const int c = s->coarse_freq_shift;
  if (abs(c) > 1) s->tuner_hz += c < 0 ? -1000 : 1000;
  if (abs(c) == 1) {
    int step;
    s->rng = s->rng * 1103515245u + 12345u;
    step = (int)((s->rng >> 16) % 1000u);
    s->tuner_hz += c < 0 ? -step : step;
  }
  if (c == 0 && abs((int)s->fine_freq_shift) > 50) s->tuner_hz += (int)floor(s->fine_freq_shift / 3);
}

struct sink { uint8_t *out; int cap, n; };
static void sink_cb(const uint8_t *eti, void *user)
{
  struct sink *k = (struct sink *)user;
  if (k->n < k->cap) memcpy(k->out + (size_t)k->n * OR_ETI_BYTES, eti, OR_ETI_BYTES);
  k->n++;
}

/* dab2eti.c:60-130 without USB and threads; afc = 0: without the tuner feedback (a file has no tuner: the reference's behaviour on a recording);
 * afc != 0: with the feedback of dab2eti.c:76-103 steering an NCO (nco_hz[call] = the frequency that call's samples were de-rotated by) */
static int replay(const uint8_t *iq, size_t nbytes, int afc, uint8_t *eti_out, int cap_frames,
                  struct or_sdr_trace *trace, int32_t *nco_hz, int trace_cap, int *ntrace)
{
  struct sink k = {eti_out, cap_frames, 0};
  struct or_sdr *s = or_sdr_new();
  struct or_dab *d = or_dab_new(sink_cb, &k);
  size_t off;
  int nt = 0;
  or_sdr_set_afc(s, afc);
  for (off = 0; off + OR_CHUNK_BYTES <= nbytes; off += OR_CHUNK_BYTES) {
    int ok = or_sdr_demod(s, iq + off, OR_CHUNK_BYTES, or_dab_tf_fic(d), or_dab_tf_msc(d));
    if (trace && nt < trace_cap) or_sdr_get_trace(s, &trace[nt]);
    if (nco_hz && nt < trace_cap) nco_hz[nt] = or_sdr_nco_hz(s);
    nt++;
    if (ok) or_dab_process_frame(d);
    if (afc) or_afc_step(s);
  }
  if (ntrace) *ntrace = nt;
  or_dab_free(d);
  or_sdr_free(s);
  return k.n;
}

int or_replay(const uint8_t *iq, size_t nbytes, uint8_t *eti_out, int cap_frames,
              struct or_sdr_trace *trace, int trace_cap, int *ntrace)
{
  return replay(iq, nbytes, 0, eti_out, cap_frames, trace, NULL, trace_cap, ntrace);
}

int or_replay_afc(const uint8_t *iq, size_t nbytes, uint8_t *eti_out, int cap_frames,
                  struct or_sdr_trace *trace, int32_t *nco_hz, int trace_cap, int *ntrace)
{
  return replay(iq, nbytes, 1, eti_out, cap_frames, trace, nco_hz, trace_cap, ntrace);
}
