/*
 * input_sdr_hip.c — seam S2: the SDR front end of dab2eti (FIFO, time/frequency synchronisation, 76 x 2048-point OFDM
 * transform, DQPSK demap, frequency de-interleave) on the GPU.  Replaces input_sdr.o sdr_sync.o sdr_fifo.o and with them
 * the libfftw3 dependency; dab2eti.c:68,234 are its callers.
 *
 * Compiled by oracle/Makefile into oracle/_ref/libdabref_hipS2.so against the reference's own dab.h / input_sdr.h (whose <fftw3.h> resolves to the
 * FFTW3-API header the image's ROCm ships, hipfft/hipfftw.h -- for the types of struct sdr_state_t only: nothing here calls an FFT) and run under the
 * same harness as the reference's real front end: tests/test_gpu_frontend_ref.py::test_seam_s2_binding_under_the_reference_harness.
 */
#include "dab.h"
#include "input_sdr.h"
#include "dabhip.h"

static dabhip_sdr *hip_sdr;

void sdr_init(struct sdr_state_t *sdr) { (void)sdr; hip_sdr = dabhip_sdr_init(0); }                     /* input_sdr.h:44 */

int sdr_demod(struct demapped_transmission_frame_t *tf, struct sdr_state_t *sdr)                        /* input_sdr.h:43 */
{
  int ok;
  tf->has_fic = 0;
  ok = dabhip_sdr_demod(hip_sdr, sdr->input_buffer, sdr->input_buffer_len,   /* 262144 = DEFAULT_BUF_LENGTH, dab2eti.c:238 */
                        tf->fic_symbols_demapped[0], tf->msc_symbols_demapped[0]);
  sdr->coarse_timeshift = dabhip_sdr_coarse_timeshift(hip_sdr);
  sdr->fine_timeshift = dabhip_sdr_fine_timeshift(hip_sdr);
  sdr->coarse_freq_shift = dabhip_sdr_coarse_freq_shift(hip_sdr);            /* read by the tuner AFC, dab2eti.c:76-103 */
  sdr->fine_freq_shift = dabhip_sdr_fine_freq_shift(hip_sdr);
  if (ok == 1) tf->has_fic = 1;
  return ok == 1;
}
