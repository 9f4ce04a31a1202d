/*
 * CPU definition of the "extension" onset analysis: Hann-windowed STFT (1024 or
 * 2048 points, hop = half the length), low-band energy, spectral flux and
 * adaptive-threshold peak picking.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as bd_oracle.h).
 *
 * NO REFERENCE COUNTERPART: beat-detector computes none of this (it feeds the
 * raw ring to an FFT, keeps three bins and compares with a constant:
 * src/strategies/spectrum.rs:82-92).  This file is therefore the definition of
 * the analysis, not a restatement, and results checked against it are
 * "extension mode", never "parity with the reference".  The arithmetic is
 * specified operation by operation (f32, one rounding per operation, fused
 * multiply-add exactly where written) so that an implementation that follows
 * the same data flow reproduces it bit for bit; bdo_ext_rfft_mag_f64 gives the
 * float64 value of the same spectrum for accuracy checks.
 */
#ifndef BD_ORACLE_EXT_H
#define BD_ORACLE_EXT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bdo_onset_params {
    uint32_t fft_len;      /* 1024 or 2048 */
    uint32_t hop;          /* fft_len / 2 */
    uint32_t window;       /* 0 rectangular, 1 periodic Hann */
    float    band_max_hz;  /* band energy over bins with k * fs / N <= band_max_hz */
    uint32_t mean_frames;  /* W: the threshold follows the mean of flux[j-W .. j+W] */
    uint32_t max_frames;   /* w: an onset is the maximum of flux[j-w .. j+w] */
    float    delta;        /* threshold = delta + lambda * mean */
    float    lambda;
    uint32_t min_gap_ms;   /* onsets closer than this to the previous one are dropped */
} bdo_onset_params;

void   bdo_onset_default_params(bdo_onset_params *p);
/* full frames only: 0 if n < fft_len, else (n - fft_len) / hop + 1 */
size_t bdo_onset_frames(size_t n_samples, uint32_t fft_len, uint32_t hop);
/* centre of frame j in ms: ((j * hop + fft_len / 2) * 1000) / fs, integer arithmetic */
uint32_t bdo_onset_frame_ms(size_t j, uint32_t fft_len, uint32_t hop, uint32_t sampling_rate);

/* |X[k]|, k = 0 .. n/2, of one frame (n = 1024 or 2048), f32 data flow as specified in the .c file */
int bdo_ext_rfft_mag(const int16_t *frame, uint32_t n, uint32_t window, float *mag);
/* the same spectrum in float64 (direct evaluation) */
int bdo_ext_rfft_mag_f64(const int16_t *frame, uint32_t n, uint32_t window, double *mag);

/* flux[j], band_energy[j] for every frame of one stream (either may be NULL) */
int bdo_onset_stft(const int16_t *samples, size_t n_samples, uint32_t sampling_rate,
                   const bdo_onset_params *p, float *flux, float *band_energy);
/* peak picking over a flux series; returns the number of onsets (even beyond max_onsets) */
size_t bdo_onset_pick(const float *flux, size_t n_frames, uint32_t sampling_rate,
                      const bdo_onset_params *p, uint8_t *candidates /*[n_frames] or NULL*/,
                      uint32_t *onset_frame, uint32_t *onset_ms, size_t max_onsets);

#ifdef __cplusplus
}
#endif
#endif
