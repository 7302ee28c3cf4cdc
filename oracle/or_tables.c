/*
 * or_tables.c — ETSI EN 300 401 constant tables for the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see dab_oracle.h).
 *
 * The reference keeps these as literal arrays (dab_tables.c, sdr_prstab.c); here the
 * regular ones are generated from the rules of the standard and checked against the
 * reference's arrays by tests/test_oracle_vs_ref.py + tests/golden/tables.npz.
 */
#include "dab_oracle.h"

/* UEP profiles: ETSI EN 300 401 Table 7 / Table 36 (reference dab_tables.c:16-81, which stores PI-1). */
static const struct or_uep_profile uep_rows[64] = {
  { 32,  16, 5, { 3,  4,  17, 0}, { 5,  3,  2,  0}},
  { 32,  21, 4, { 3,  3,  18, 0}, {11,  6,  5,  0}},
  { 32,  24, 3, { 3,  4,  14, 3}, {15,  9,  6,  8}},
  { 32,  29, 2, { 3,  4,  14, 3}, {22, 13,  8, 13}},
  { 32,  35, 1, { 3,  5,  13, 3}, {24, 17, 12, 17}},
  { 48,  24, 5, { 4,  3,  26, 3}, { 5,  4,  2,  3}},
  { 48,  29, 4, { 3,  4,  26, 3}, { 9,  6,  4,  6}},
  { 48,  35, 3, { 3,  4,  26, 3}, {15, 10,  6,  9}},
  { 48,  42, 2, { 3,  4,  26, 3}, {24, 14,  8, 15}},
  { 48,  52, 1, { 3,  5,  25, 3}, {24, 18, 13, 18}},
  { 56,  29, 5, { 6, 10,  23, 3}, { 5,  4,  2,  3}},
  { 56,  35, 4, { 6, 10,  23, 3}, { 9,  6,  4,  5}},
  { 56,  42, 3, { 6, 12,  21, 3}, {16,  7,  6,  9}},
  { 56,  52, 2, { 6, 10,  23, 3}, {23, 13,  8, 13}},
  { 64,  32, 5, { 6,  9,  31, 2}, { 5,  3,  2,  3}},
  { 64,  42, 4, { 6,  9,  33, 0}, {11,  6,  5,  0}},
  { 64,  48, 3, { 6, 12,  27, 3}, {16,  8,  6,  9}},
  { 64,  58, 2, { 6, 10,  29, 3}, {23, 13,  8, 13}},
  { 64,  70, 1, { 6, 11,  28, 3}, {24, 18, 12, 18}},
  { 80,  40, 5, { 6, 10,  41, 3}, { 6,  3,  2,  3}},
  { 80,  52, 4, { 6, 10,  41, 3}, {11,  6,  5,  6}},
  { 80,  58, 3, { 6, 11,  40, 3}, {16,  8,  6,  7}},
  { 80,  70, 2, { 6, 10,  41, 3}, {23, 13,  8, 13}},
  { 80,  84, 1, { 6, 10,  41, 3}, {24, 17, 12, 18}},
  { 96,  48, 5, { 7,  9,  53, 3}, { 5,  4,  2,  4}},
  { 96,  58, 4, { 7, 10,  52, 3}, { 9,  6,  4,  6}},
  { 96,  70, 3, { 6, 12,  51, 3}, {16,  9,  6, 10}},
  { 96,  84, 2, { 6, 10,  53, 3}, {22, 12,  9, 12}},
  { 96, 104, 1, { 6, 13,  50, 3}, {24, 18, 13, 19}},
  {112,  58, 5, {14, 17,  50, 3}, { 5,  4,  2,  5}},
  {112,  70, 4, {11, 21,  49, 3}, { 9,  6,  4,  8}},
  {112,  84, 3, {11, 23,  47, 3}, {16,  8,  6,  9}},
  {112, 104, 2, {11, 21,  49, 3}, {23, 12,  9, 14}},
  {128,  64, 5, {12, 19,  62, 3}, { 5,  3,  2,  4}},
  {128,  84, 4, {11, 21,  61, 3}, {11,  6,  5,  7}},
  {128,  96, 3, {11, 22,  60, 3}, {16,  9,  6, 10}},
  {128, 116, 2, {11, 21,  61, 3}, {22, 12,  9, 14}},
  {128, 140, 1, {11, 20,  62, 3}, {24, 17, 13, 19}},
  {160,  80, 5, {11, 19,  87, 3}, { 5,  4,  2,  4}},
  {160, 104, 4, {11, 23,  83, 3}, {11,  6,  5,  9}},
  {160, 116, 3, {11, 24,  82, 3}, {16,  8,  6, 11}},
  {160, 140, 2, {11, 21,  85, 3}, {22, 11,  9, 13}},
  {160, 168, 1, {11, 22,  84, 3}, {24, 18, 12, 19}},
  {192,  96, 5, {11, 20, 110, 3}, { 6,  4,  2,  5}},
  {192, 116, 4, {11, 22, 108, 3}, {10,  6,  4,  9}},
  {192, 140, 3, {11, 24, 106, 3}, {16, 10,  6, 11}},
  {192, 168, 2, {11, 20, 110, 3}, {22, 13,  9, 13}},
  {192, 208, 1, {11, 21, 109, 3}, {24, 20, 13, 24}},
  {224, 116, 5, {12, 22, 131, 3}, { 8,  6,  2,  6}},
  {224, 140, 4, {12, 26, 127, 3}, {12,  8,  4, 11}},
  {224, 168, 3, {11, 20, 134, 3}, {16, 10,  7,  9}},
  {224, 208, 2, {11, 22, 132, 3}, {24, 16, 10, 15}},
  {224, 232, 1, {11, 24, 130, 3}, {24, 20, 12, 20}},
  {256, 128, 5, {11, 24, 154, 3}, { 6,  5,  2,  5}},
  {256, 168, 4, {11, 24, 154, 3}, {12,  9,  5, 10}},
  {256, 192, 3, {11, 27, 151, 3}, {16, 10,  7, 10}},
  {256, 232, 2, {11, 22, 156, 3}, {24, 14, 10, 13}},
  {256, 280, 1, {11, 26, 152, 3}, {24, 19, 14, 18}},
  {320, 160, 5, {11, 26, 200, 3}, { 8,  5,  2,  6}},
  {320, 208, 4, {11, 25, 201, 3}, {13,  9,  5, 10}},
  {320, 280, 2, {11, 26, 200, 3}, {24, 17,  9, 17}},
  {384, 192, 5, {11, 27, 247, 3}, { 8,  6,  2,  7}},
  {384, 280, 3, {11, 24, 250, 3}, {16,  9,  7, 10}},
  {384, 416, 1, {12, 28, 245, 3}, {24, 20, 14, 23}},
};

const struct or_uep_profile *or_uep_table(void) { return uep_rows; }

/* Puncturing vectors V_PI (ETSI Table 29; reference dab_tables.c:102-127).  Every vector
 * keeps the first bit of each group of four; PI further bits are switched on in the
 * order: second bits of groups 0,4,2,6,1,5,3,7, then third bits, then fourth bits. */
static uint32_t pmask[24];
static uint16_t revtab[1536];
static int8_t prs[1536];
static int tables_ready;

/* ETSI Table 44 (h_{i,j}) and Table 39 (Mode I: i and n per block of 32 carriers). */
static const uint8_t prs_h[4][32] = {
  {0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1,0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1},
  {0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0,0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0},
  {0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3,0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3},
  {0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2,0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2}};
static const uint8_t prs_n[48] = {
  1,2,0,1,3,2,2,3,2,1,2,3,1,2,3,3,2,2,2,1,1,3,1,2,
  3,1,1,1,2,2,1,0,2,2,3,3,0,2,1,3,3,3,3,0,3,0,1,1};

static void tables_init(void)
{
  static const int order[8] = {0, 4, 2, 6, 1, 5, 3, 7};
  int pi, k, i, n;
  if (tables_ready) return;
  for (pi = 1; pi <= 24; pi++) {
    uint32_t m = 0x11111111u;
    for (k = 0; k < pi; k++) m |= 1u << (4 * order[k & 7] + 1 + (k >> 3));
    pmask[pi - 1] = m;
  }
  /* frequency interleaver, ETSI 14.6.1: PI(i) = 13 PI(i-1) + 511 mod 2048 (reference
   * generator kept under #if 0 at dab_tables.c:130-162). */
  {
    int ki = 0;
    n = 0;
    for (i = 0; i < 2048; i++) {
      if (i) ki = (13 * ki + 511) % 2048;
      if (ki >= 256 && ki <= 1792 && ki != 1024) {
        k = ki - 1024;
        k = (k < 0) ? 768 + k : 768 + k - 1;   /* carrier index 0..1535, ascending frequency */
        revtab[k] = (uint16_t)n++;
      }
    }
  }
  /* phase reference symbol, ETSI 14.3.2: phi_k = pi/2 (h_{i,k-k'} + n) */
  for (k = 0; k < 1536; k++) {
    int blk = k / 32;
    int row = (blk < 24) ? (blk & 3) : ((4 - (blk & 3)) & 3);
    prs[k] = (int8_t)((prs_h[row][k & 31] + prs_n[blk]) & 3);
  }
  tables_ready = 1;
}

const uint32_t *or_puncture_masks(void) { tables_init(); return pmask; }
const uint16_t *or_rev_freq_deint_tab(void) { tables_init(); return revtab; }
const int8_t *or_prs_phase(void) { tables_init(); return prs; }

/* Depuncture plan of a sub-channel (reference uep_depuncture depuncture.c:84-105 and
 * eep_depuncture :107-132 incl. the 2-A / 8 kbit/s special case; eeptable dab_tables.c:87-100). */
void or_subch_plan(const struct or_subch *sc, struct or_punct_plan *plan)
{
  int i;
  if (!sc->slform) {
    const struct or_uep_profile *p = &uep_rows[sc->uep_index];
    plan->nseg = 4;
    for (i = 0; i < 4; i++) { plan->blocks[i] = p->l[i]; plan->pi[i] = p->pi[i]; }
  } else {
    /* per protection level index (option<<2 | level): size multiple, L1 = a*n + b, L2 = c*n + d, PI1, PI2 */
    static const int eep[8][7] = {
      {12, 6, -3, 0, 3, 24, 23}, {8, 2, -3, 4, 3, 14, 13}, {6, 6, -3, 0, 3, 8, 7}, {4, 4, -3, 2, 3, 3, 2},
      {27, 24, -3, 0, 3, 10, 9}, {21, 24, -3, 0, 3, 6, 5}, {18, 24, -3, 0, 3, 4, 3}, {15, 24, -3, 0, 3, 2, 1}};
    const int *e = eep[sc->protlev & 7];
    int n = sc->size / e[0];
    plan->nseg = 2;
    if (sc->bitrate == 8 && sc->protlev == 1) {   /* depuncture.c:113-114, dab_tables.c:98-100 */
      plan->blocks[0] = 5; plan->pi[0] = 4;
      plan->blocks[1] = 1; plan->pi[1] = 13;
    } else {
      plan->blocks[0] = e[1] * n + e[2]; plan->pi[0] = e[5];
      plan->blocks[1] = e[3] * n + e[4]; plan->pi[1] = e[6];
    }
    plan->blocks[2] = plan->blocks[3] = 0; plan->pi[2] = plan->pi[3] = 0;
  }
}
