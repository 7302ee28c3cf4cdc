/*
 * ref_harness.c — flat C entry points around the REAL reference objects.
 * TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile together with the reference's
 * own, unmodified sources where they lie under /root/reference/src; the result goes to
 * oracle/_ref/ (git-ignored).  No reference source is copied into this repository.
 *
 * Only the reference files that build with gcc alone are used: viterbi.c depuncture.c
 * dab_tables.c fic.c misc.c dab.c sdr_fifo.c (and the viterbi_spiral*.c pair for the SSE
 * variant).  input_sdr.c / sdr_sync.c need an FFTW3 library: see ref_frontend_harness.c.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* REFH_S3_ONLY: only the replay entry points (the back end itself is integration/dab_hip.c over libdabhip);
 * REFH_NO_ENCODE: the reference's viterbi.c (which also holds encode()) is replaced by integration/viterbi_hip.c */
#include "dab.h"
#ifndef REFH_S3_ONLY
#include "dab_tables.h"
#include "depuncture.h"
#include "fic.h"
#include "misc.h"
#include "sdr_fifo.h"
#ifdef ENABLE_SPIRAL_VITERBI
#include "viterbi_spiral.h"
#else
#include "viterbi.h"
int encode(unsigned char *symbols, unsigned char *data, unsigned int nbytes, unsigned int startstate,
           unsigned int endstate);
#endif
int init_eti(uint8_t *eti, struct ens_info_t *info);
void time_deinterleave(uint8_t *dst, uint8_t *cifs[]);
#endif

struct refh {
  struct dab_state_t *dab;
  uint8_t *eti;
  int neti, cap;
};
static struct refh *cur;

static void on_eti(uint8_t *eti)
{
  if (cur->neti == cur->cap) {
    cur->cap = cur->cap ? cur->cap * 2 : 64;
    cur->eti = (uint8_t *)realloc(cur->eti, (size_t)cur->cap * 6144);
  }
  memcpy(cur->eti + (size_t)cur->neti * 6144, eti, 6144);
  cur->neti++;
}

void *refh_new(void)
{
  struct refh *h = (struct refh *)calloc(1, sizeof *h);
  init_dab_state(&h->dab, NULL, on_eti);
  h->dab->device_type = DAB_DEVICE_RTLSDR;
  return h;
}
uint8_t *refh_tf_fic(void *p) { struct refh *h = p; return h->dab->tfs[h->dab->tfidx].fic_symbols_demapped[0]; }
uint8_t *refh_tf_msc(void *p) { struct refh *h = p; return h->dab->tfs[h->dab->tfidx].msc_symbols_demapped[0]; }
void refh_process(void *p)
{
  struct refh *h = p;
  cur = h;
  h->dab->tfs[h->dab->tfidx].has_fic = 1;   /* what sdr_demod does, input_sdr.c:147 */
  dab_process_frame(h->dab);
}
int refh_neti(void *p) { return ((struct refh *)p)->neti; }
const uint8_t *refh_eti(void *p) { return ((struct refh *)p)->eti; }
int refh_locked(void *p) { return ((struct refh *)p)->dab->locked; }
int refh_tfidx(void *p) { return ((struct refh *)p)->dab->tfidx; }
#ifndef REFH_S3_ONLY
/* FIBs + CRC flags of TF buffer `idx` (0..4) */
const uint8_t *refh_fibs(void *p, int idx) { return ((struct refh *)p)->dab->tfs[idx].fibs.FIB[0]; }
const uint8_t *refh_fib_ok(void *p, int idx) { return ((struct refh *)p)->dab->tfs[idx].fibs.FIB_CRC_OK; }

void refh_viterbi(void *p, uint8_t *symbols, uint8_t *data, int nbits)
{
  viterbi(p ? ((struct refh *)p)->dab->v : NULL, symbols, data, nbits);
}
#if !defined(ENABLE_SPIRAL_VITERBI) && !defined(REFH_NO_ENCODE)
void refh_encode(uint8_t *symbols, uint8_t *data, unsigned nbytes) { encode(symbols, data, nbytes, 0, 0); }
#endif
void refh_fic_depuncture(uint8_t *out, uint8_t *in) { fic_depuncture(out, in); }
int refh_uep_depuncture(uint8_t *out, uint8_t *in, int uep_index)
{
  struct subchannel_info_t s;
  int len = 0;
  memset(&s, 0, sizeof s);
  s.uep_index = uep_index;
  uep_depuncture(out, in, &s, &len);
  return len;
}
int refh_eep_depuncture(uint8_t *out, uint8_t *in, int protlev, int size, int bitrate)
{
  struct subchannel_info_t s;
  int len = 0;
  memset(&s, 0, sizeof s);
  s.protlev = protlev; s.size = size; s.bitrate = bitrate;
  eep_depuncture(out, in, &s, &len);
  return len;
}
void refh_descramble(uint8_t *buf, int n) { dab_descramble_bytes(buf, n); }
int refh_check_fib_crc(uint8_t *fib) { return check_fib_crc(fib); }
void refh_time_deinterleave(uint8_t *dst, uint8_t *base, int stride)
{
  uint8_t *cifs[16];
  int i;
  for (i = 0; i < 16; i++) cifs[i] = base + (size_t)i * stride;
  time_deinterleave(dst, cifs);
}
/* sub = 64 rows of {id, slForm, start_cu, bitrate, protlev} */
int refh_init_eti(uint8_t *eti, int eid, int cif_hi, int cif_lo, const int *sub)
{
  struct ens_info_t info;
  int i;
  memset(&info, 0, sizeof info);
  info.EId = (uint16_t)eid; info.CIFCount_hi = (uint8_t)cif_hi; info.CIFCount_lo = (uint8_t)cif_lo;
  for (i = 0; i < 64; i++) {
    info.subchans[i].id = sub[5 * i];
    info.subchans[i].slForm = sub[5 * i + 1];
    info.subchans[i].start_cu = sub[5 * i + 2];
    info.subchans[i].bitrate = sub[5 * i + 3];
    info.subchans[i].protlev = sub[5 * i + 4];
  }
  return init_eti(eti, &info);
}
/* parse 12 FIBs; out = 64 rows of {id, slForm, uep_index, start_cu, size, bitrate, protlev, ASCTy}; hdr = {EId, hi, lo} */
void refh_fib_decode(const uint8_t *fib, const uint8_t *ok, int *hdr, int *out)
{
  struct tf_fibs_t f;
  struct tf_info_t info;
  int i;
  memset(&f, 0, sizeof f);
  memcpy(f.FIB, fib, 12 * 32);
  memcpy(f.FIB_CRC_OK, ok, 12);
  fib_decode(&info, &f, 12);
  hdr[0] = info.EId; hdr[1] = info.CIFCount_hi; hdr[2] = info.CIFCount_lo;
  for (i = 0; i < 64; i++) {
    struct subchannel_info_t *s = &info.subchans[i];
    int *o = out + 8 * i;
    o[0] = s->id; o[1] = s->slForm; o[2] = s->uep_index; o[3] = s->start_cu;
    o[4] = s->size; o[5] = s->bitrate; o[6] = s->protlev; o[7] = s->ASCTy;
  }
}
const uint16_t *refh_rev_freq_deint_tab(void) { return rev_freq_deint_tab; }
const char *refh_pvec(void) { return &pvec[0][0]; }
/* row = {bitrate, subchsz, protlvl, l0..l3, pi0..pi3} */
void refh_uep_row(int idx, int *row)
{
  const struct uepprof *p = &ueptable[idx];
  int i;
  row[0] = (int)p->bitrate; row[1] = (int)p->subchsz; row[2] = (int)p->protlvl;
  for (i = 0; i < 4; i++) { row[3 + i] = p->l[i]; row[7 + i] = p->pi[i]; }
}

/* byte FIFO (sdr_fifo.c) */
void *refh_fifo_new(uint32_t size) { CircularBuffer *cb = calloc(1, sizeof *cb); cbInit(cb, size); return cb; }
void refh_fifo_write(void *cb, uint8_t *p, int n) { int i; for (i = 0; i < n; i++) cbWrite(cb, p + i); }
void refh_fifo_read(void *cb, uint32_t bytes, int32_t shift, uint8_t *buffer) { sdr_read_fifo(cb, bytes, shift, buffer); }
uint32_t refh_fifo_count(void *cb) { return ((CircularBuffer *)cb)->count; }
#endif /* REFH_S3_ONLY */
