/*
 * viterbi_hip.c — seam S1: the decoder of dab2eti on the GPU.
 *
 * The file a dabtools maintainer adds to src/ next to viterbi.c / viterbi_spiral.c (src/Makefile:8-16 picks one of them
 * at link time): VITERBI_OBJS=viterbi_hip.o, LDFLAGS+=-ldabhip.  It defines both spellings of the decoder interface, so it
 * links with or without ENABLE_SPIRAL_VITERBI in dab.c:27-31; symbol conventions are those of the scalar build
 * (127/129 hard, 128 erased: depuncture.c:36-43 without the macro).
 *
 * callers: fic.c:186, misc.c:262 (viterbi), dab.c:28-32 (create_viterbi / init_viterbi).
 * Compiled against the reference's own unmodified callers and tested by oracle/Makefile (_ref/libdabref_hipS1.so) +
 * tests/test_gpu_parity_r2.py::test_reference_callers_over_the_hip_seams.
 */
#include "dabhip.h"

void *create_viterbi(int len) { return dabhip_create_viterbi(len); }          /* viterbi_spiral.h:22 */
int init_viterbi(void) { return dabhip_init_viterbi(); }                      /* viterbi.h:6 */
void viterbi(void *p, unsigned char *symbols, unsigned char *data, int framebits)
{
  dabhip_viterbi(p, symbols, data, framebits);                                /* viterbi.h:8 */
}
