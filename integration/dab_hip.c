/*
 * dab_hip.c — seam S3: the back end of dab2eti (FIC decode, lock, time de-interleave, de-puncture, Viterbi, ETI
 * assembly) on the GPU.  Replaces dab.o fic.o misc.o depuncture.o viterbi*.o; dab2eti.c:70,294-298 are its callers and
 * stay as they are.  Compiled against the reference's dab.h and tested by oracle/Makefile (_ref/libdabref_hipS3.so) +
 * tests/test_gpu_parity_r2.py::test_reference_callers_over_the_hip_seams.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dab.h"
#include "dabhip.h"

void init_dab_state(struct dab_state_t **dab, void *device_state, void (*eti_callback)(uint8_t *eti))   /* dab.h:91 */
{
  *dab = calloc(sizeof(struct dab_state_t), 1);
  (*dab)->device_state = device_state;
  (*dab)->eti_callback = eti_callback;
  /* the handle lives where the reference keeps its decoder's (dab->v, dab.c:29): one GPU back end per dab_state_t */
  (*dab)->v = dabhip_dab_init(0, eti_callback);        /* the callback type is identical: dab.h:88 */
}

void dab_process_frame(struct dab_state_t *dab)                                                         /* dab.h:92 */
{
  dabhip_dab *h = (dabhip_dab *)dab->v;
  struct demapped_transmission_frame_t *tf = &dab->tfs[dab->tfidx];   /* tfidx stays 0: one staging buffer */
  if (!h) return;                                      /* no GPU: no output, like every failure of the reference's back end */
  memcpy(dabhip_dab_tf_fic(h), tf->fic_symbols_demapped, 9216);
  memcpy(dabhip_dab_tf_msc(h), tf->msc_symbols_demapped, 221184);
  dabhip_dab_process_frame(h);                         /* calls eti_callback 0 or 4 times, synchronously */
  dab->locked = dabhip_dab_locked(h);
  {                                                    /* "Locked" / "Lock lost, resetting ringbuffer" / the ensemble dump: dab.c:51,57,78-82 */
    char text[8192];
    if (dabhip_dab_take_log(h, text, sizeof text) > 0) fputs(text, stderr);
  }
}
