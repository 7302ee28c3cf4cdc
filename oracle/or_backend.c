/*
 * or_backend.c — CPU restatement of the channel-decoding back end of dab2eti.
 * TEST INFRASTRUCTURE ONLY (see dab_oracle.h).  Pinned against the real reference
 * objects (oracle/_ref/libdabref.so) by tests/test_oracle_vs_ref.py.
 */
#include "dab_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* convolutional code K=7, rate 1/4, generators 0x6d 0x4f 0x53 0x6d (viterbi.c:35) */

static const unsigned conv_poly[4] = {0x6d, 0x4f, 0x53, 0x6d};

static int parity7(unsigned x)
{
  x ^= x >> 4; x ^= x >> 2; x ^= x >> 1;
  return (int)(x & 1);
}

/* viterbi.c:322-347 — shift register shifts left, newest bit in the LSB, 6 zero tail bits */
void or_encode(uint8_t *symbols, const uint8_t *data, unsigned nbytes)
{
  unsigned state = 0, n, j;
  int b;
  for (n = 0; n < nbytes + 1; n++) {
    int nb = (n < nbytes) ? 8 : 6;
    for (b = 0; b < nb; b++) {
      unsigned bit = (n < nbytes) ? ((data[n] >> (7 - b)) & 1u) : 0u;
      state = (state << 1) | bit;
      for (j = 0; j < 4; j++) *symbols++ = (uint8_t)parity7(state & conv_poly[j] & 0x7f);
    }
  }
}

/* viterbi.c:126-191 with init_viterbi's arguments amp=1 noise=1.0 bias=0 scale=4 (viterbi.c:455-462).
 * Only the entries for symbols 127/128/129 are ever used by the reference (depuncture.c:36-43);
 * for the far tails log2(0) is cast to int there (undefined), which this restatement clamps. */
void or_viterbi_mettab(int tab[2][256])
{
  int s, bit;
  for (s = 0; s < 256; s++) {
    double p0, p1, m[2];
    if (s == 0) {
      p1 = 0.5 + 0.5 * erf(((0 - 128 + 0.5) - 1) / M_SQRT2);
      p0 = 0.5 + 0.5 * erf(((0 - 128 + 0.5) + 1) / M_SQRT2);
    } else if (s == 255) {
      p1 = 1 - (0.5 + 0.5 * erf(((255 - 128 - 0.5) - 1) / M_SQRT2));
      p0 = 1 - (0.5 + 0.5 * erf(((255 - 128 - 0.5) + 1) / M_SQRT2));
    } else {
      p1 = (0.5 + 0.5 * erf(((s - 128 + 0.5) - 1) / M_SQRT2)) - (0.5 + 0.5 * erf(((s - 128 - 0.5) - 1) / M_SQRT2));
      p0 = (0.5 + 0.5 * erf(((s - 128 + 0.5) + 1) / M_SQRT2)) - (0.5 + 0.5 * erf(((s - 128 - 0.5) + 1) / M_SQRT2));
    }
    m[0] = log(2 * p0 / (p1 + p0)) * M_LOG2E;
    m[1] = log(2 * p1 / (p1 + p0)) * M_LOG2E;
    for (bit = 0; bit < 2; bit++) {
      double v = floor(m[bit] * 4 + 0.5);
      tab[bit][s] = (v > -1e9 && v < 1e9) ? (int)v : -2147483647 - 1;
    }
  }
}

/* viterbi.c:352-451.  State = last six input bits, newest in bit 0; predecessors of new
 * state i are i>>1 (decision 0) and (i>>1)|32 (decision 1); the high predecessor is taken
 * only if strictly better; start state 0, 6 tail steps, chain back from state 0, MSB first. */
void or_viterbi(const uint8_t *symbols, uint8_t *data, int nbits)
{
  static int mettab[2][256];
  static int syms[64];
  static int ready;
  int nsteps = nbits + 6, t, i, j, c;
  long cm[64], nm[64];
  uint64_t *paths = (uint64_t *)calloc((size_t)nsteps, sizeof(uint64_t));
  unsigned state;

  if (!ready) {
    or_viterbi_mettab(mettab);
    for (i = 0; i < 64; i++) {
      int s = 0;
      for (j = 0; j < 4; j++) s = (s << 1) | parity7((unsigned)i & conv_poly[j]);
      syms[i] = s;          /* code word on the branch low-predecessor -> i (viterbi.c:373-381) */
    }
    ready = 1;
  }
  for (i = 0; i < 64; i++) cm[i] = -999999;   /* viterbi.c:387-389 */
  cm[0] = 0;
  for (t = 0; t < nsteps; t++) {
    int mets[16];
    uint64_t dec = 0;
    for (c = 0; c < 16; c++) {                 /* viterbi.c:394-399 */
      mets[c] = 0;
      for (j = 0; j < 4; j++) mets[c] += mettab[(c >> (3 - j)) & 1][symbols[4 * t + j]];
    }
    for (i = 0; i < 64; i++) {                 /* viterbi.c:402-428 */
      long m0 = cm[i >> 1] + mets[syms[i]];
      long m1 = cm[(i >> 1) | 32] + mets[syms[i] ^ 15];
      if (m1 > m0) { nm[i] = m1; dec |= (uint64_t)1 << i; } else nm[i] = m0;
    }
    paths[t] = dec;
    memcpy(cm, nm, sizeof cm);
  }
  memset(data, 0, (size_t)((nbits + 7) / 8));   /* viterbi.c:438-450 */
  state = 0;
  for (i = nbits - 1; i >= 0; i--) {
    unsigned d = (unsigned)((paths[i + 6] >> state) & 1u);
    if (d) data[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
    state = (state | (d << 6)) >> 1;
  }
  free(paths);
}

/* ------------------------------------------------------------------------- */
/* depuncture.c:36-43: hard bit -> 127/129, erasure 128 */
#define SYM(b) ((uint8_t)(127 + 2 * (b)))
#define ERASED 128

/* depuncture.c:45-82: 21 blocks PI=16, 3 blocks PI=15, 24 tail bits with PI=8 */
void or_fic_depuncture(uint8_t *out, const uint8_t *in)
{
  const uint32_t *pm = or_puncture_masks();
  int i, k = 0, j = 0;
  for (i = 0; i < 21 * 128; i++) out[k++] = ((pm[15] >> (i & 31)) & 1) ? SYM(in[j++]) : ERASED;
  for (i = 0; i < 3 * 128; i++) out[k++] = ((pm[14] >> (i & 31)) & 1) ? SYM(in[j++]) : ERASED;
  for (i = 0; i < 24; i++) out[k++] = ((pm[7] >> (i & 31)) & 1) ? SYM(in[j++]) : ERASED;
}

/* depuncture.c:84-132 */
int or_msc_depuncture(uint8_t *out, const uint8_t *in, const struct or_subch *sc)
{
  const uint32_t *pm = or_puncture_masks();
  struct or_punct_plan plan;
  int s, i, k = 0, j = 0;
  or_subch_plan(sc, &plan);
  for (s = 0; s < plan.nseg; s++)
    for (i = 0; i < 128 * plan.blocks[s]; i++)
      out[k++] = ((pm[plan.pi[s] - 1] >> (i & 31)) & 1) ? SYM(in[j++]) : ERASED;
  for (i = 0; i < 24; i++) out[k++] = ((pm[7] >> (i & 31)) & 1) ? SYM(in[j++]) : ERASED;
  return k;
}

/* misc.c:41-58: PRBS x^9 + x^5 + 1, all-ones start, restarted per block */
void or_descramble(uint8_t *buf, int nbytes)
{
  unsigned reg = 0x1ff;
  int i, j;
  for (i = 0; i < nbytes; i++) {
    unsigned q = 0;
    for (j = 0; j < 8; j++) {
      unsigned fb = ((reg >> 8) ^ (reg >> 4)) & 1u;
      reg = ((reg << 1) | fb) & 0x3ff;
      q = (q << 1) | fb;
    }
    buf[i] ^= (uint8_t)q;
  }
}

/* misc.c:131-143 with the 0x1021 table (misc.c:96-129): MSB-first CRC-16/CCITT, no reflection */
uint16_t or_crc16_ccitt(const uint8_t *data, int len, uint16_t crc)
{
  int i, b;
  for (i = 0; i < len; i++) {
    crc ^= (uint16_t)(data[i] << 8);
    for (b = 0; b < 8; b++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
  }
  return crc;
}

int or_check_fib_crc(const uint8_t *fib) { return or_crc16_ccitt(fib, 32, 0xffff) == 0x1d0f; } /* misc.c:145-150 */

/* misc.c:29-39 */
void or_time_deinterleave(uint8_t *dst, const uint8_t *const cifs[16])
{
  static const int map[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  int i;
  for (i = 0; i < OR_CIF_BITS; i++) dst[i] = cifs[map[i & 15]][i];
}

/* ------------------------------------------------------------------------- */
/* fic.c:160-208 (has_fic == 1 branch) */
int or_fic_decode(const uint8_t *fic_bits, uint8_t fib[12][32], uint8_t crc_ok[12])
{
  uint8_t sym[3096];
  int i, j, ok = 0;
  for (i = 0; i < 4; i++) {
    or_fic_depuncture(sym, fic_bits + i * 2304);
    or_viterbi(sym, fib[3 * i], 768);
    or_descramble(fib[3 * i], 96);
    for (j = 0; j < 3; j++) {
      crc_ok[3 * i + j] = (uint8_t)or_check_fib_crc(fib[3 * i + j]);
      ok += crc_ok[3 * i + j];
    }
  }
  return ok;
}

/* fic.c:47-130: FIG 0/0, 0/1, 0/2 only.  `fib` points into a 12x32 array followed by the
 * 12 CRC flags exactly as struct tf_fibs_t lays them out (dab.h:21-25), so that a FIG
 * whose length runs past the FIB reads the same following bytes as the reference. */
static void fib_parse(struct or_ens_info *info, const uint8_t *fib, const uint8_t *limit)
{
  int i = 0, j, k;
#define RD(ix) (((fib + (ix)) < limit) ? fib[(ix)] : 0)
  while (RD(i) != 0xff && i < 30) {
    int type = (RD(i) & 0xe0) >> 5;
    int len = RD(i) & 0x1f;
    i++;
    if (type == 0) {
      int ext = RD(i) & 0x1f;
      int pd = (RD(i) & 0x20) >> 5;
      if (ext == 0) {
        info->eid = (uint16_t)((RD(i + 1) << 8) | RD(i + 2));
        info->cif_hi = RD(i + 3) & 0x1f;
        info->cif_lo = RD(i + 4);
      } else if (ext == 1) {
        j = i + 1;
        while (j < i + len) {
          int id = (RD(j) & 0xfc) >> 2;
          struct or_subch *sc = &info->sub[id];
          sc->id = id;
          sc->start_cu = ((RD(j) & 0x03) << 8) | RD(j + 1);
          sc->slform = (RD(j + 2) & 0x80) >> 7;
          if (!sc->slform) {
            const struct or_uep_profile *p = &or_uep_table()[RD(j + 2) & 0x3f];
            sc->uep_index = RD(j + 2) & 0x3f;
            sc->size = p->size_cu;
            sc->bitrate = p->bitrate;
            sc->protlev = p->protlevel;
            j += 3;
          } else {
            static const int sizemul[8] = {12, 8, 6, 4, 27, 21, 18, 15};
            int option = (RD(j + 2) & 0x70) >> 4;
            sc->protlev = ((RD(j + 2) & 0x0c) >> 2) | (option << 2);
            sc->size = ((RD(j + 2) & 0x03) << 8) | RD(j + 3);
            /* fic.c:84 indexes eeptable[protlev] unchecked; option > 1 is outside the standard */
            sc->bitrate = (sc->size / sizemul[sc->protlev & 7]) * ((sc->protlev & 4) ? 32 : 8);
            j += 4;
          }
        }
      } else if (ext == 2) {
        j = i + 1;
        while (j < i + len) {
          int n;
          j += pd ? 4 : 2;
          n = RD(j) & 0x0f;
          j++;
          for (k = 0; k < n; k++) {
            if (((RD(j) & 0xc0) >> 6) == 0) info->sub[(RD(j + 1) & 0xfc) >> 2].ascty = RD(j) & 0x3f;
            j += 2;
          }
        }
      }
    }
    i += len;
  }
#undef RD
}

/* fic.c:132-147 */
void or_fib_decode(struct or_ens_info *info, const uint8_t fib[12][32], const uint8_t crc_ok[12])
{
  uint8_t img[12 * 32 + 12];
  int i;
  memset(info, 0, sizeof *info);
  for (i = 0; i < 64; i++) { info->sub[i].id = -1; info->sub[i].ascty = -1; }
  memcpy(img, fib, 12 * 32);
  memcpy(img + 12 * 32, crc_ok, 12);
  for (i = 0; i < 12; i++)
    if (crc_ok[i]) fib_parse(info, img + 32 * i, img + sizeof img);
}

/* misc.c:153-213 */
int or_init_eti(uint8_t *eti, const struct or_ens_info *info)
{
  int i = 0, j, nst = 0, fl = 0, fp;
  unsigned hcrc;
  eti[i++] = 0xff;
  if (info->cif_lo & 1) { eti[i++] = 0xf8; eti[i++] = 0xc5; eti[i++] = 0x49; }
  else { eti[i++] = 0x07; eti[i++] = 0x3a; eti[i++] = 0xb6; }
  eti[i++] = info->cif_lo;
  for (j = 0; j < 64; j++)
    if (info->sub[j].id >= 0) { nst++; fl += (info->sub[j].bitrate * 3) / 4; }
  fl += nst + 1 + 24;
  eti[i++] = (uint8_t)(0x80 | nst);
  fp = (info->cif_hi * 250 + info->cif_lo) % 8;
  eti[i++] = (uint8_t)((fp << 5) | (1 << 3) | ((fl & 0x700) >> 8));
  eti[i++] = (uint8_t)(fl & 0xff);
  for (j = 0; j < 64; j++) {
    const struct or_subch *sc = &info->sub[j];
    if (sc->id >= 0) {
      int tpl = sc->slform ? (0x20 | sc->protlev) : (0x10 | (sc->protlev - 1));
      int stl = (sc->bitrate * 3) / 8;
      eti[i++] = (uint8_t)((sc->id << 2) | ((sc->start_cu & 0x300) >> 8));
      eti[i++] = (uint8_t)(sc->start_cu & 0xff);
      eti[i++] = (uint8_t)((tpl << 2) | ((stl & 0x300) >> 8));
      eti[i++] = (uint8_t)(stl & 0xff);
    }
  }
  eti[i++] = 0xff;
  eti[i++] = 0xff;
  hcrc = (unsigned)~or_crc16_ccitt(eti + 4, i - 4, 0xffff);
  eti[i++] = (uint8_t)((hcrc >> 8) & 0xff);
  eti[i++] = (uint8_t)(hcrc & 0xff);
  return i;
}

/* ------------------------------------------------------------------------- */
/* dab.h:70-89, dab.c:14-98, misc.c:14-27,218-314 */
struct or_tf {
  uint8_t fic[OR_FIC_BITS];
  uint8_t fib[12][32];
  uint8_t crc_ok[12];
  int ok_count;
  uint8_t msc[OR_MSC_BITS];
  float *sfic, *smsc;       /* soft-decision extension (or_soft.c): the same two arrays as values; NULL in the reference's mode */
};

struct or_dab {
  struct or_tf tfs[5];
  struct or_ens_info tf_info, ens;
  const uint8_t *cifs_msc[16];
  const uint8_t *cifs_fibs[16];
  int ncifs, tfidx, locked, okcount;
  or_eti_cb cb;
  void *user;
  int soft;                 /* 0 = the reference (hard decisions); else OR_SOFT_* (or_soft.c) */
  const float *cifs_smsc[16];
};

struct or_dab *or_dab_new(or_eti_cb cb, void *user)
{
  struct or_dab *d = (struct or_dab *)calloc(1, sizeof *d);
  int i;
  for (i = 0; i < 64; i++) { d->ens.sub[i].id = -1; d->ens.sub[i].ascty = -1; }
  d->ens.cif_hi = 0xff;
  d->ens.cif_lo = 0xff;
  d->cb = cb;
  d->user = user;
  return d;
}
void or_dab_free(struct or_dab *d)
{
  int i;
  for (i = 0; i < 5; i++) { free(d->tfs[i].sfic); free(d->tfs[i].smsc); }
  free(d);
}
/* soft-decision extension: the TF hand-off carries values (or_dab_tf_sfic / _smsc) instead of 0/1 bytes; before the first frame */
void or_dab_set_soft(struct or_dab *d, int mode)
{
  int i;
  d->soft = mode;
  for (i = 0; i < 5; i++) {
    d->tfs[i].sfic = (float *)calloc(OR_FIC_BITS, sizeof(float));
    d->tfs[i].smsc = (float *)calloc(OR_MSC_BITS, sizeof(float));
  }
}
float *or_dab_tf_sfic(struct or_dab *d) { return d->tfs[d->tfidx].sfic; }
float *or_dab_tf_smsc(struct or_dab *d) { return d->tfs[d->tfidx].smsc; }
uint8_t *or_dab_tf_fic(struct or_dab *d) { return d->tfs[d->tfidx].fic; }
uint8_t *or_dab_tf_msc(struct or_dab *d) { return d->tfs[d->tfidx].msc; }
int or_dab_locked(const struct or_dab *d) { return d->locked; }
const uint8_t *or_dab_last_fibs(const struct or_dab *d, uint8_t crc_ok[12])
{
  memcpy(crc_ok, d->tfs[d->tfidx].crc_ok, 12);
  return d->tfs[d->tfidx].fib[0];
}

/* misc.c:218-314 */
static void create_eti(struct or_dab *d)
{
  static uint8_t cif[OR_CIF_BITS];
  static uint8_t dp[3072 * 4 * 18];
  static float scif[OR_CIF_BITS], sdp[3072 * 4 * 18];
  uint8_t eti[OR_ETI_BYTES];
  struct or_ens_info *info = &d->ens;
  int e1 = or_init_eti(eti, info), e, i;
  unsigned crc;
  memcpy(eti + e1, d->cifs_fibs[0], 96);
  e = e1 + 96;
  if (d->soft) or_time_deinterleave_soft(scif, d->cifs_smsc);
  else or_time_deinterleave(cif, d->cifs_msc);
  for (i = 0; i < 64; i++) {
    const struct or_subch *sc = &info->sub[i];
    if (sc->id >= 0) {
      int len = d->soft ? or_msc_depuncture_soft(sdp, scif + sc->start_cu * 64, sc) : or_msc_depuncture(dp, cif + sc->start_cu * 64, sc);
      int bits = len / 4 - 6;
      int obytes = ((bits / 8) + 7) & 0xfff8;
      if (d->soft) or_viterbi_soft(sdp, eti + e, bits, d->soft);
      else or_viterbi(dp, eti + e, bits);
      or_descramble(eti + e, obytes);
      e += obytes;
    }
  }
  crc = (unsigned)~or_crc16_ccitt(eti + e1, e - e1, 0xffff);
  eti[e++] = (uint8_t)((crc >> 8) & 0xff);
  eti[e++] = (uint8_t)(crc & 0xff);
  memset(eti + e, 0xff, 6);
  e += 6;
  memset(eti + e, 0x55, (size_t)(OR_ETI_BYTES - e));
  if (d->cb) d->cb(eti, d->user);
  if (++info->cif_lo == 250) {
    info->cif_lo = 0;
    if (++info->cif_hi == 20) info->cif_hi = 0;
  }
}

/* dab.c:35-98 */
void or_dab_process_frame(struct or_dab *d)
{
  struct or_tf *tf = &d->tfs[d->tfidx];
  int i;
  tf->ok_count = d->soft ? or_fic_decode_soft(tf->sfic, d->soft, tf->fib, tf->crc_ok) : or_fic_decode(tf->fic, tf->fib, tf->crc_ok);
  if (tf->ok_count > 0) or_fib_decode(&d->tf_info, (const uint8_t(*)[32])tf->fib, tf->crc_ok);
  if (tf->ok_count == 12) {
    d->okcount++;
    if (d->okcount >= 10 && !d->locked) d->locked = 1;
  } else {
    d->okcount = 0;
    if (d->locked) { d->locked = 0; d->ncifs = 0; d->tfidx = 0; return; }
  }
  if (!d->locked) return;
  /* merge_info, misc.c:14-27 */
  for (i = 0; i < 64; i++)
    if (d->tf_info.sub[i].id >= 0) d->ens.sub[i] = d->tf_info.sub[i];
  d->ens.eid = d->tf_info.eid;
  if (d->ens.cif_hi == 0xff) { d->ens.cif_hi = d->tf_info.cif_hi; d->ens.cif_lo = d->tf_info.cif_lo; }
  if (d->ncifs < 16) {
    for (i = 0; i < 4; i++) {
      d->cifs_fibs[d->ncifs] = tf->fib[3 * i];
      if (d->soft) d->cifs_smsc[d->ncifs] = tf->smsc + i * OR_CIF_BITS;
      d->cifs_msc[d->ncifs++] = tf->msc + i * OR_CIF_BITS;
    }
  } else {
    for (i = 0; i < 4; i++) {
      create_eti(d);
      memmove(d->cifs_fibs, d->cifs_fibs + 1, sizeof d->cifs_fibs[0] * 15);
      memmove(d->cifs_msc, d->cifs_msc + 1, sizeof d->cifs_msc[0] * 15);
      memmove(d->cifs_smsc, d->cifs_smsc + 1, sizeof d->cifs_smsc[0] * 15);
      d->cifs_fibs[15] = tf->fib[3 * i];
      d->cifs_msc[15] = tf->msc + i * OR_CIF_BITS;
      if (d->soft) d->cifs_smsc[15] = tf->smsc + i * OR_CIF_BITS;
    }
  }
  d->tfidx = (d->tfidx + 1) % 5;
}
