/*
 * CPU definition of the "extension" onset analysis: Hann-windowed STFT (1024 or
 * 2048 points, hop = half or a quarter of the length), low-band energy, spectral flux and
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
    uint32_t use_median;   /* 1: the threshold follows the median of flux[j-W .. j+W] (element c/2 of the
                              sorted window of c values) instead of the mean; W <= 32 */
    uint32_t flux_power;   /* 1: flux (and band energy) from the squared magnitudes, no square roots */
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

/* ---- carried-state biquad envelope (extension, no reference counterpart) ----
 * The reference's low-pass is first order and restarts in every window
 * (src/strategies/lpf.rs:66-71).  This is the north-star variant: a second-order
 * low-pass whose state is carried through the whole stream, evaluated in chunks of
 * 64 samples so that the chunk-to-chunk recurrence is a scan.  See the .c file for
 * the arithmetic specification. */
#define BDO_BQ_CHUNK 64
typedef struct bdo_biquad_design {
    float b0, b1, b2, na1, na2;   /* y[n] = na1 y[n-1] + na2 y[n-2] + u[n], u = FIR part */
    float c1b, c2b;               /* two-step form: y[n+1] = c1b y[n-1] + c2b y[n-2] + na1 u[n] + u[n+1] */
    float pw[6][4];               /* P^(2^k), k = 0..5, P = state transition over one chunk; row-major 2x2 */
    float q[64][4];               /* P^(l+1), l = 0..63 */
} bdo_biquad_design;

typedef struct bdo_biquad_params {
    uint32_t cutoff_hz;      /* 120 */
    float    q_factor;       /* 0.70710678 */
    uint32_t window_len;     /* decision window, power of two, 64 .. 4096 */
    float    ratio;          /* 0.77: threshold = trunc(ratio * max|x|) */
    uint32_t min_amplitude;  /* 9830 */
} bdo_biquad_params;

void bdo_biquad_default_params(bdo_biquad_params *p);
/* RBJ low-pass, float64 design rounded once to f32; returns 0, or -1 for unusable parameters */
int  bdo_biquad_design_lowpass(uint32_t sampling_rate, uint32_t cutoff_hz, float q_factor, bdo_biquad_design *d);
/* Per complete window of one stream: value = max |y| over the window (f32), flag = gate && value >= threshold.
 * y_out (n_samples floats, may be NULL) receives the filtered signal of the complete chunks. */
int  bdo_biquad_envelope(const int16_t *samples, size_t n_samples, const bdo_biquad_design *d,
                         const bdo_biquad_params *p, float *values, uint8_t *flags, float *y_out);
/* the same filter evaluated sample by sample in float64 (accuracy reference) */
void bdo_biquad_filter_f64(const int16_t *samples, size_t n_samples, uint32_t sampling_rate, uint32_t cutoff_hz,
                           double q_factor, double *y);

#ifdef __cplusplus
}
#endif
#endif
