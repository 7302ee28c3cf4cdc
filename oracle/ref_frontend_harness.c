/*
 * ref_frontend_harness.c — flat entry points over the REAL reference front end (input_sdr.c, sdr_sync.c, sdr_fifo.c: compiled
 * unmodified from /root/reference/src by oracle/Makefile into oracle/_ref/libdabref_frontend.so).  TEST INFRASTRUCTURE ONLY.
 *
 * What the FFT calls bind to.  The reference includes <fftw3.h> and links -lfftw3 (Makefile:3, version unpinned); libfftw3 is not in
 * this image.  The ROCm installation of the image ships AMD's own implementation of the FFTW3 API -- /opt/rocm/include/hipfft/hipfftw.h
 * and libhipfftw.so, a front for hipFFT / rocFFT, double precision, host pointers in and out -- and THAT is what this build uses: the
 * Makefile gives the vendor header the name the reference asks for (a symbolic link under oracle/_ref/include, no text of ours) and
 * links the vendor library.  So sdr_demod, the four estimators, the FIFO and the demapping loop below are the reference's own object
 * code; the DFT behind fftw_execute is a third party's, as it is with libfftw3 -- a different third party, whose results agree with
 * any correct double-precision DFT to ~1e-13 relative, far inside what the sign tests and arg-maxima downstream resolve.  It needs a
 * GPU at run time (plans execute on the device), so only -m gpu tests load it.
 *
 * This harness plays demod_thread_fn's part for one stream (dab2eti.c:60-75 without the tuner): copy a 262,144-byte buffer into
 * sdr->input_buffer, call sdr_demod, hand back what it left.
 */
#include <stdlib.h>
#include <string.h>

#include "dab.h"
#include "input_sdr.h"

struct reff {
  struct sdr_state_t sdr;
  struct demapped_transmission_frame_t tf;
};

#ifdef __cplusplus          /* (built by g++: the vendor's FFTW3 header is a C++ header; the entry points keep C names) */
extern "C" {
#endif

void *reff_new(void)
{
  struct reff *h = (struct reff *)calloc(1, sizeof *h);       /* dab2eti.c:39,146: the state is a zero-initialised static */
  if (h) sdr_init(&h->sdr);                                   /* input_sdr.c:167-186 */
  return h;
}

void reff_free(void *p)
{
  struct reff *h = (struct reff *)p;
  if (!h) return;
  cbFree(&h->sdr.fifo);
  free(h);
}

/* one sdr_demod call (input_sdr.c:27-165).  ints6 = {ok, frame read (unknown to the caller: -1), coarse_timeshift, fine_timeshift,
 * coarse_freq_shift, fifo.count}; fic / msc filled when it returns 1 */
int reff_demod(void *p, const uint8_t *chunk, int len, uint8_t *fic, uint8_t *msc, int32_t *ints6, double *fine_freq_shift)
{
  struct reff *h = (struct reff *)p;
  int ok;
  if (len > DEFAULT_BUF_LENGTH) return -1;
  memcpy(h->sdr.input_buffer, chunk, (size_t)len);            /* rtlsdr_callback, dab2eti.c:125-126 */
  h->sdr.input_buffer_len = len;
  ok = sdr_demod(&h->tf, &h->sdr);
  if (ok) {
    memcpy(fic, h->tf.fic_symbols_demapped, sizeof h->tf.fic_symbols_demapped);
    memcpy(msc, h->tf.msc_symbols_demapped, sizeof h->tf.msc_symbols_demapped);
  }
  if (ints6) {
    ints6[0] = ok;
    ints6[1] = -1;
    ints6[2] = h->sdr.coarse_timeshift;
    ints6[3] = h->sdr.fine_timeshift;
    ints6[4] = h->sdr.coarse_freq_shift;
    ints6[5] = (int32_t)h->sdr.fifo.count;
  }
  if (fine_freq_shift) *fine_freq_shift = h->sdr.fine_freq_shift;
  return ok;
}

/* the fftshifted spectra of the last demodulated frame: sdr->symbols[76][2048] (input_sdr.c:115-130) */
const double *reff_symbols(void *p) { return (const double *)((struct reff *)p)->sdr.symbols; }
/* the frame buffer after the last read (the stale-tail semantics of sdr_read_fifo, sdr_fifo.c:43-61) */
const uint8_t *reff_buffer(void *p) { return ((struct reff *)p)->sdr.buffer; }

#ifdef __cplusplus
}
#endif
