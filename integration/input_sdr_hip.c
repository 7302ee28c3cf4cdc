/*
 * input_sdr_hip.c — seam S2: the SDR front end of dab2eti (FIFO, time/frequency synchronisation, 76 x 2048-point OFDM
 * transform, DQPSK demap, frequency de-interleave) on the GPU.  Replaces input_sdr.o sdr_sync.o sdr_fifo.o and with them
 * the libfftw3 dependency; dab2eti.c:68,234 are its callers.
 *
 * Compiled by oracle/Makefile into oracle/_ref/libdabref_hipS2.so against the reference's own dab.h / input_sdr.h (whose <fftw3.h> resolves to the
 * FFTW3-API header the image's ROCm ships, hipfft/hipfftw.h -- for the types of struct sdr_state_t only: nothing here calls an FFT) and run under the
 * same harness as the reference's real front end: tests/test_gpu_frontend_ref.py::test_seam_s2_binding_under_the_reference_harness.
 */
#include <stdio.h>

#include "dab.h"
#include "input_sdr.h"
#include "dabhip.h"

/* One GPU front end per struct sdr_state_t: a small side table keyed by the state's address (the reference's struct has no field to spare,
 * input_sdr.h:12-41).  dab2eti.c has one state; a caller with several gets one handle each.  Not thread-safe across states being initialised at
 * the same time -- neither is the reference's sdr_init (FFTW's planner, sdr_sync.c). */
#define HIP_SDR_MAX 64
static struct { const struct sdr_state_t *key; dabhip_sdr *h; } hip_sdr_tab[HIP_SDR_MAX];

static dabhip_sdr **hip_sdr_slot(const struct sdr_state_t *sdr, int create)
{
  int i;
  for (i = 0; i < HIP_SDR_MAX; i++)
    if (hip_sdr_tab[i].key == sdr) return &hip_sdr_tab[i].h;
  if (create)
    for (i = 0; i < HIP_SDR_MAX; i++)        /* a free slot, or one whose front end could not be made (failed init: no GPU at the time) */
      if (!hip_sdr_tab[i].key || !hip_sdr_tab[i].h) { hip_sdr_tab[i].key = sdr; return &hip_sdr_tab[i].h; }
  return 0;
}

void sdr_init(struct sdr_state_t *sdr)                                                                  /* input_sdr.h:44 */
{
  dabhip_sdr **slot = hip_sdr_slot(sdr, 1);
  if (!slot) {                                         /* the reference's API has no sdr_free: states are never released, and the table is finite */
    fprintf(stderr, "input_sdr_hip: more than %d sdr_state_t initialised: this one gets no GPU front end (sdr_demod will report no frames)\n", HIP_SDR_MAX);
    return;
  }
  if (*slot) dabhip_sdr_free(*slot);                   /* the same state initialised again: a fresh front end, like the reference's memsets */
  *slot = dabhip_sdr_init(0);
  if (!*slot) fprintf(stderr, "input_sdr_hip: %s\n", dabhip_last_error());      /* no GPU: there is no CPU fallback; sdr_demod reports no frames */
}

int sdr_demod(struct demapped_transmission_frame_t *tf, struct sdr_state_t *sdr)                        /* input_sdr.h:43 */
{
  dabhip_sdr **slot = hip_sdr_slot(sdr, 0);
  dabhip_sdr *h = slot ? *slot : 0;
  int ok;
  tf->has_fic = 0;
  if (!h) return 0;                                    /* sdr_init was not called (or failed: no GPU) -- "no frame", the only failure the seam has */
  ok = dabhip_sdr_demod(h, sdr->input_buffer, sdr->input_buffer_len,         /* whatever the callback left (dab2eti.c:125-126), 262144 from librtlsdr */
                        tf->fic_symbols_demapped[0], tf->msc_symbols_demapped[0]);
  sdr->coarse_timeshift = dabhip_sdr_coarse_timeshift(h);
  sdr->fine_timeshift = dabhip_sdr_fine_timeshift(h);
  sdr->coarse_freq_shift = dabhip_sdr_coarse_freq_shift(h);                  /* read by the tuner AFC, dab2eti.c:76-103 */
  sdr->fine_freq_shift = dabhip_sdr_fine_freq_shift(h);
  if (ok == 1) tf->has_fic = 1;
  return ok == 1;
}
