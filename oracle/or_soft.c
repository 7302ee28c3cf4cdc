/*
 * or_soft.c — CPU restatement of the SOFT-DECISION extension (BASELINE configs[4]).
 * TEST INFRASTRUCTURE ONLY (see dab_oracle.h): loaded by tests/, and by tools/cpu_baseline.py for the quantisation report.
 *
 * Whose specification this is.  The reference decodes hard decisions only: the demapper stores one bit per carrier
 * component (input_sdr.c:157-158), the de-puncturers turn it into 127 / 129 with 128 for a punctured position
 * (depuncture.c:36-43), and the scalar decoder's metric table is valid for 121..135 only (viterbi.c:126-191,455-462).
 * So there is no reference behaviour to be exact against; the rule restated here is the product's own, written down
 * independently of its kernels (plain loops, fp64, `long` path metrics) so that the kernels can be held against it:
 *
 *   soft value   v = q(g x / (s(l) s(l-1))),   x = Re resp. -Im_stored of cur conj(prev) of a carrier (the two quantities whose
 *                signs the reference tests, input_sdr.c:135-143,157-158; v > 0 <=> the hard bit would be 0),
 *                s(l) = sqrt(sum over the 2048 samples of symbol l's FFT window of re^2 + im^2), g = 7.0 / 0.94280904
 *                (device_types.hpp: soft_scale -- a noise-free Mode-I symbol then has mean |x g / (s s')| = 7.0, the clamp),
 *                q = round to nearest (ties to even) and clamp to +-7            (OR_SOFT_Q4, what the product computes)
 *                q = the same in steps of 1/16, clamp +-127/16                   (OR_SOFT_Q8: the 8 bits SURVEY 8(f) names)
 *                q = identity                                                    (OR_SOFT_FLOAT: no quantisation at all)
 *   punctured    positions carry 0 (depuncture.c:36-43 puts the neutral 128 there)
 *   decoder      the reference's trellis, start / end condition, tie rule and chain-back (viterbi.c:352-451) with the branch
 *                metric  28 + sum_j (c_j ? -v_j : +v_j)  of code word c = (c_0 .. c_3) in place of the table look-up
 *                (viterbi.c:394-399); the 28 keeps integer metrics non-negative and cancels in every comparison.
 *   everything after the decoder (descrambling, CRC, FIB parsing, lock rule, CIF ring, ETI assembly) is the hard path's.
 *
 * Pinning.  (i) With |v| constant (every received value +-a, punctured 0) the metric orders paths exactly like the
 * reference's agreement metric (10 agree - 7 per received symbol), so or_viterbi_soft must return the bytes of the REAL
 * viterbi.c on the same hard input, ties included: tests/test_oracle_soft.py holds it against oracle/_ref/libdabref.so and
 * against the committed known-answer vectors.  (ii) At high SNR the soft replay must reproduce the hard replay's ETI bytes.
 * Beyond that the soft rule has no external authority -- it is the builder's -- and DESIGN.md says so.
 */
#include "dab_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SOFT_GAIN_DEFAULT (7.0 / 0.94280904)
static double soft_gain = SOFT_GAIN_DEFAULT;
#define SOFT_GAIN soft_gain
/* experiments only (tools/soft_quant_loss.py --gain): the product's gain is the default */
void or_soft_set_gain(double mean_abs_value) { soft_gain = mean_abs_value > 0 ? mean_abs_value / 0.94280904 : SOFT_GAIN_DEFAULT; }

double or_soft_quantise(double v, int mode)
{
  double r;
  if (mode == OR_SOFT_FLOAT) return v;
  if (mode == OR_SOFT_Q8) {
    r = nearbyint(v * 16.0);                    /* round-half-even, like v_cvt / __float2int_rn on the device */
    if (r > 127) r = 127;
    if (r < -127) r = -127;
    return r / 16.0;
  }
  r = nearbyint(v);
  if (r > 7) r = 7;
  if (r < -7) r = -7;
  return r;
}

/* energy of the 2048 samples symbol l is transformed from (input_sdr.c:115-130: window at 2656 + 2552 l + 504) */
static double symbol_energy(const double *frame, int l)
{
  const double *x = frame + 2 * (size_t)(2656 + 2552 * l + 504);
  double e = 0;
  int n;
  for (n = 0; n < 2048; n++) e += x[2 * n] * x[2 * n] + x[2 * n + 1] * x[2 * n + 1];
  return e;                                      /* integers below 2^53: exact */
}

/* The demapping loop of input_sdr.c:132-162 with values instead of signs; same carrier walk, same de-interleaver. */
void or_soft_demap(const struct or_sdr *s, int mode, float *fic, float *msc)
{
  const uint16_t *rev = or_rev_freq_deint_tab();
  const double *symbols = or_sdr_symbols(s), *frame = or_sdr_frame(s);
  double sprev = sqrt(symbol_energy(frame, 0));
  int i, j;
  for (j = 1; j < 76; j++) {
    float *dst = (j < 4) ? fic + (j - 1) * 3072 : msc + (j - 4) * 3072;
    const double *cur = symbols + (size_t)j * 4096, *prev = cur - 4096;
    const double scur = sqrt(symbol_energy(frame, j));
    const double scale = (scur * sprev > 0) ? SOFT_GAIN / (scur * sprev) : 0.0;
    int k = 0;
    for (i = 0; i < 2048; i++) {
      if (i > 255 && i != 1024 && i < 1793) {
        double cr = cur[2 * i], ci = cur[2 * i + 1], pr = prev[2 * i], pi = prev[2 * i + 1];
        double re = cr * pr + ci * pi;             /* input_sdr.c:135-138 without the division: Re(cur conj(prev)) */
        double im = cr * pi - ci * pr;             /* input_sdr.c:139-143: the stored imaginary part, sign as stored */
        int kk = rev[k++];
        dst[kk] = (float)or_soft_quantise(re * scale, mode);            /* > 0  <=>  bit 0 (input_sdr.c:157) */
        dst[1536 + kk] = (float)or_soft_quantise(-im * scale, mode);    /* stored im > 0 <=> bit 1 (input_sdr.c:158) */
      }
    }
    sprev = scur;
  }
}

/* ---- decoder ---------------------------------------------------------------------------------------------------- */
static const unsigned soft_poly[4] = {0x6d, 0x4f, 0x53, 0x6d};      /* viterbi.c:35 */
static int soft_parity(unsigned x)
{
  x ^= x >> 4; x ^= x >> 2; x ^= x >> 1;
  return (int)(x & 1);
}

/* Integer form (Q4: units of 1, Q8: units of 1/16): `long` path metrics, nothing re-based, nothing packed. */
static void viterbi_soft_int(const long *v, uint8_t *data, int nbits, long offset)
{
  int nsteps = nbits + 6, t, i, j;
  long cm[64], nm[64];
  uint64_t *paths = (uint64_t *)calloc((size_t)nsteps, sizeof(uint64_t));
  unsigned state;
  int syms[64];
  for (i = 0; i < 64; i++) {                                         /* viterbi.c:373-381 */
    syms[i] = 0;
    for (j = 0; j < 4; j++) syms[i] = (syms[i] << 1) | soft_parity((unsigned)i & soft_poly[j]);
  }
  for (i = 0; i < 64; i++) cm[i] = -999999999L;                      /* viterbi.c:387-389 */
  cm[0] = 0;
  for (t = 0; t < nsteps; t++) {
    long mets[16];
    uint64_t dec = 0;
    int c;
    for (c = 0; c < 16; c++) {                                       /* viterbi.c:394-399 with +-v in place of mettab */
      mets[c] = offset;
      for (j = 0; j < 4; j++) mets[c] += ((c >> (3 - j)) & 1) ? -v[4 * t + j] : v[4 * t + j];
    }
    for (i = 0; i < 64; i++) {                                       /* viterbi.c:402-428 */
      const long m0 = cm[i >> 1] + mets[syms[i]];
      const long m1 = cm[(i >> 1) | 32] + mets[syms[i] ^ 15];
      if (m1 > m0) { nm[i] = m1; dec |= (uint64_t)1 << i; } else nm[i] = m0;
    }
    paths[t] = dec;
    memcpy(cm, nm, sizeof cm);
  }
  memset(data, 0, (size_t)((nbits + 7) / 8));                        /* viterbi.c:438-450 */
  state = 0;
  for (i = nbits - 1; i >= 0; i--) {
    unsigned d = (unsigned)((paths[i + 6] >> state) & 1u);
    if (d) data[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
    state = (state | (d << 6)) >> 1;
  }
  free(paths);
}

/* the same in double precision for unquantised values */
static void viterbi_soft_f64(const float *v, uint8_t *data, int nbits)
{
  int nsteps = nbits + 6, t, i, j;
  double cm[64], nm[64];
  uint64_t *paths = (uint64_t *)calloc((size_t)nsteps, sizeof(uint64_t));
  unsigned state;
  int syms[64];
  for (i = 0; i < 64; i++) {
    syms[i] = 0;
    for (j = 0; j < 4; j++) syms[i] = (syms[i] << 1) | soft_parity((unsigned)i & soft_poly[j]);
  }
  for (i = 0; i < 64; i++) cm[i] = -1e18;
  cm[0] = 0;
  for (t = 0; t < nsteps; t++) {
    double mets[16];
    uint64_t dec = 0;
    int c;
    for (c = 0; c < 16; c++) {
      mets[c] = 0;
      for (j = 0; j < 4; j++) mets[c] += ((c >> (3 - j)) & 1) ? -(double)v[4 * t + j] : (double)v[4 * t + j];
    }
    for (i = 0; i < 64; i++) {
      const double m0 = cm[i >> 1] + mets[syms[i]];
      const double m1 = cm[(i >> 1) | 32] + mets[syms[i] ^ 15];
      if (m1 > m0) { nm[i] = m1; dec |= (uint64_t)1 << i; } else nm[i] = m0;
    }
    paths[t] = dec;
    memcpy(cm, nm, sizeof cm);
  }
  memset(data, 0, (size_t)((nbits + 7) / 8));
  state = 0;
  for (i = nbits - 1; i >= 0; i--) {
    unsigned d = (unsigned)((paths[i + 6] >> state) & 1u);
    if (d) data[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
    state = (state | (d << 6)) >> 1;
  }
  free(paths);
}

/* soft: 4 (nbits + 6) values, 0 at punctured positions.  Q4 values are integers in [-7, 7] (a -8, which the demapper never
 * produces, counts as -7: the product's metric tables do the same); Q8 values multiples of 1/16. */
void or_viterbi_soft(const float *soft, uint8_t *data, int nbits, int mode)
{
  int n = 4 * (nbits + 6), i;
  long *v;
  if (mode == OR_SOFT_FLOAT) { viterbi_soft_f64(soft, data, nbits); return; }
  v = (long *)malloc(sizeof(long) * (size_t)n);
  for (i = 0; i < n; i++) {
    v[i] = lrint((double)soft[i] * (mode == OR_SOFT_Q8 ? 16.0 : 1.0));
    if (mode == OR_SOFT_Q4 && v[i] < -7) v[i] = -7;
  }
  viterbi_soft_int(v, data, nbits, mode == OR_SOFT_Q8 ? 28 * 16 : 28);
  free(v);
}

/* ---- de-puncturing with values (depuncture.c:45-132: the same walks, 0 where the reference puts 128) ------------ */
void or_fic_depuncture_soft(float *out, const float *in)
{
  const uint32_t *pm = or_puncture_masks();
  int i, k = 0, j = 0;
  for (i = 0; i < 21 * 128; i++) out[k++] = ((pm[15] >> (i & 31)) & 1) ? in[j++] : 0.0f;
  for (i = 0; i < 3 * 128; i++) out[k++] = ((pm[14] >> (i & 31)) & 1) ? in[j++] : 0.0f;
  for (i = 0; i < 24; i++) out[k++] = ((pm[7] >> (i & 31)) & 1) ? in[j++] : 0.0f;
}

int or_msc_depuncture_soft(float *out, const float *in, const struct or_subch *sc)
{
  const uint32_t *pm = or_puncture_masks();
  struct or_punct_plan plan;
  int s, i, k = 0, j = 0;
  or_subch_plan(sc, &plan);
  for (s = 0; s < plan.nseg; s++)
    for (i = 0; i < 128 * plan.blocks[s]; i++)
      out[k++] = ((pm[plan.pi[s] - 1] >> (i & 31)) & 1) ? in[j++] : 0.0f;
  for (i = 0; i < 24; i++) out[k++] = ((pm[7] >> (i & 31)) & 1) ? in[j++] : 0.0f;
  return k;
}

/* misc.c:29-39 on values */
void or_time_deinterleave_soft(float *dst, const float *const cifs[16])
{
  static const int map[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  int i;
  for (i = 0; i < OR_CIF_BITS; i++) dst[i] = cifs[map[i & 15]][i];
}

/* fic.c:160-208 on values */
int or_fic_decode_soft(const float *fic_soft, int mode, uint8_t fib[12][32], uint8_t crc_ok[12])
{
  float sym[3096];
  int i, j, ok = 0;
  for (i = 0; i < 4; i++) {
    or_fic_depuncture_soft(sym, fic_soft + i * 2304);
    or_viterbi_soft(sym, fib[3 * i], 768, mode);
    or_descramble(fib[3 * i], 96);
    for (j = 0; j < 3; j++) {
      crc_ok[3 * i + j] = (uint8_t)or_check_fib_crc(fib[3 * i + j]);
      ok += crc_ok[3 * i + j];
    }
  }
  return ok;
}

/* ---- replay: or_replay with the soft demapper and decoders in place of the hard ones ------------------------------ */
struct soft_sink { uint8_t *out; int cap, n; };
static void soft_sink_cb(const uint8_t *eti, void *user)
{
  struct soft_sink *k = (struct soft_sink *)user;
  if (k->n < k->cap) memcpy(k->out + (size_t)k->n * OR_ETI_BYTES, eti, OR_ETI_BYTES);
  k->n++;
}

int or_replay_soft(const uint8_t *iq, size_t nbytes, int mode, uint8_t *eti_out, int cap_frames, float *values_out, int values_cap_tf, int *ntf)
{
  struct soft_sink k = {eti_out, cap_frames, 0};
  struct or_sdr *s = or_sdr_new();
  struct or_dab *d = or_dab_new(soft_sink_cb, &k);
  uint8_t *hard_fic = (uint8_t *)malloc(OR_FIC_BITS), *hard_msc = (uint8_t *)malloc(OR_MSC_BITS);
  size_t off;
  int n = 0;
  or_dab_set_soft(d, mode);
  for (off = 0; off + OR_CHUNK_BYTES <= nbytes; off += OR_CHUNK_BYTES) {
    /* the front end is the hard path's up to the demapper (synchronisation does not look at decisions) */
    if (or_sdr_demod(s, iq + off, OR_CHUNK_BYTES, hard_fic, hard_msc)) {
      or_soft_demap(s, mode, or_dab_tf_sfic(d), or_dab_tf_smsc(d));
      if (values_out && n < values_cap_tf) {
        memcpy(values_out + (size_t)n * (OR_FIC_BITS + OR_MSC_BITS), or_dab_tf_sfic(d), sizeof(float) * OR_FIC_BITS);
        memcpy(values_out + (size_t)n * (OR_FIC_BITS + OR_MSC_BITS) + OR_FIC_BITS, or_dab_tf_smsc(d), sizeof(float) * OR_MSC_BITS);
      }
      n++;
      or_dab_process_frame(d);
    }
  }
  if (ntf) *ntf = n;
  free(hard_fic);
  free(hard_msc);
  or_dab_free(d);
  or_sdr_free(s);
  return k.n;
}
