/*
 * CPU restatement of beat-detector's per-window analysis path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing that ships (the HIP library, its host
 * wrappers, the C-ABI) may include, link or call this file.  Its users are the
 * test-suite, the smoke check and the CPU-baseline leg of the benchmark, and
 * there only as the checker.
 *
 * Pinning status.  Everything the reference itself holds is restated literally
 * and is pinned by the reference's one known-answer test (the millisecond
 * clock, src/strategies/mod.rs:122-164) and by its source constants.  The FFT /
 * spectrum scaling and the low-pass recurrence live in third-party crates
 * (spectrum-analyzer ^1.1.0, lowpass-filter ^0.2.4; Cargo.toml:19-20) whose
 * source is not part of the reference tree and for which no lock-file exists:
 * for those two pieces this file restates the crates' published algorithms and
 * is PARITY UNPINNED.  Each unpinned choice is a field of bdo_params.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference repository root).
 */
#ifndef BD_ORACLE_H
#define BD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDO_RING_LEN 1024 /* src/strategies/spectrum.rs:43,49 */

enum { BDO_KIND_LPF = 0, BDO_KIND_SPECTRUM = 1 }; /* src/lib.rs:144-149 */

/* P1: what `divide_by_N` divides by. */
enum { BDO_DIV_BINS = 0, BDO_DIV_FFT_LEN = 1, BDO_DIV_NONE = 2 };

typedef struct bdo_params {
    /* shared gates */
    uint32_t min_amplitude;    /* src/lib.rs:134  (32767*0.3) as i16 = 9830 */
    uint32_t min_beat_gap_ms;  /* src/strategies/spectrum.rs:130, lpf.rs:113 = 400 */
    /* Spectrum strategy */
    float    spectrum_max_hz;      /* spectrum.rs:85  = 90.0 */
    float    spectrum_threshold;   /* spectrum.rs:92  = 2_100_000.0 */
    uint32_t spectrum_divisor;     /* P1, default BDO_DIV_BINS */
    uint32_t spectrum_include_dc;  /* P2, default 1 */
    uint32_t spectrum_inclusive;   /* P3, default 1: keep bins with f <= max_hz */
    uint32_t spectrum_dc_packs_nyquist; /* P6, default 0: |X0| = hypot(DC, Nyquist) */
    /* LPF strategy */
    uint32_t lpf_cutoff_hz;        /* lpf.rs:65 = 120 */
    float    lpf_ratio;            /* lpf.rs:74 = 0.77 */
} bdo_params;

void bdo_default_params(bdo_params *p);

/* ---- AnalysisState, src/strategies/mod.rs:39-116 ---- */
typedef struct bdo_state {
    uint32_t sampling_rate;
    float    ms_per_sample;
    uint32_t beat_time_ms;
    uint32_t time_ms;
    uint32_t last_beat_timestamp;
} bdo_state;

void bdo_state_new(bdo_state *s, uint32_t sampling_rate);       /* mod.rs:60-69 */
void bdo_state_update_time(bdo_state *s, size_t frame_len);      /* mod.rs:76-84 */
void bdo_state_mark_beat(bdo_state *s);                          /* mod.rs:89-91 */

/* ---- shared gates, src/lib.rs:115-136 ---- */
int bdo_last_beat_beyond_threshold(const bdo_state *s, uint32_t min_gap_ms);
int bdo_amplitude_high_enough(uint16_t window_max, uint32_t min_amplitude);

/* ---- WindowStats, src/strategies/window_stats.rs:41-55 ---- */
uint16_t bdo_window_max(const int16_t *samples, size_t n);

/* ---- third-party arithmetic, restated (PARITY UNPINNED) ---- */
/* lowpass_filter::simple::sp::apply_lpf_i16_sp, called at lpf.rs:67-71 */
void bdo_apply_lpf_i16_sp(int16_t *data, size_t n, uint16_t sample_rate_hz,
                          uint16_t cutoff_hz);
/* spectrum_analyzer::samples_fft_to_spectrum(.., Max(90), divide_by_N).max(),
 * called at spectrum.rs:82-92.  Returns the scaled maximum; *n_bins gets the
 * number of retained bins (0 means the reference would panic). */
float bdo_spectrum_band_max(const float *ring /*[1024] oldest first*/,
                            uint32_t sampling_rate, const bdo_params *p,
                            int *n_bins);
/* number of retained bins for a sampling rate (same rule as above) */
int bdo_spectrum_bin_range(uint32_t sampling_rate, const bdo_params *p,
                           int *first_bin, int *last_bin);

/* ---- detectors, spectrum.rs:37-99 and lpf.rs:34-82 ---- */
typedef struct bdo_detector {
    int        kind;
    bdo_params params;
    bdo_state  state;
    float      ring[BDO_RING_LEN]; /* Spectrum only; chronological ring */
    uint32_t   ring_head;          /* index of the oldest element */
    /* diagnostics of the last call (not part of the reference's API) */
    float      last_value;      /* Spectrum: scaled band max; LPF: max|y| */
    int        last_candidate;  /* decision ignoring the refractory gate */
    int        last_computed;   /* 1 if the expensive part ran */
} bdo_detector;

int  bdo_detector_new(bdo_detector *d, int kind, uint32_t sampling_rate,
                      const bdo_params *p /* NULL = defaults */);
/* Strategy::is_beat, src/lib.rs:78.  Returns 1 and sets *relative_ms on a beat,
 * 0 otherwise, <0 if the reference would panic (empty slice). */
int  bdo_is_beat(bdo_detector *d, const int16_t *samples, size_t n,
                 uint32_t *relative_ms);
/* Same call, but evaluates the amplitude gate and the strategy decision even
 * when the refractory gate would have returned early; fills the diagnostics.
 * Beat results are identical to bdo_is_beat. */
int  bdo_is_beat_full(bdo_detector *d, const int16_t *samples, size_t n,
                      uint32_t *relative_ms);

/* ---- ingest converters, src/record.rs:151-166 ---- */
void bdo_u16_data_to_i16(const uint16_t *in, int16_t *out, size_t n);
void bdo_f32_data_to_i16(const float *in, int16_t *out, size_t n);

/* ---- offline harness, src/lib.rs:263-337 ---- */
/* stereo interleaved -> mono, lib.rs:319-324 */
void   bdo_stereo_to_mono(const int16_t *interleaved, size_t n_frames, int16_t *mono);
/* length after the harness's padding rule, lib.rs:270-274 */
size_t bdo_harness_padded_len(size_t n, size_t window_len);

/* Feed one stream callback by callback (equal callback length; a trailing
 * partial callback is dropped as at lib.rs:276).  Optional outputs may be NULL.
 * full != 0 uses bdo_is_beat_full so value/candidate are filled for every
 * callback.  Returns the number of beats found (even beyond max_beats). */
size_t bdo_run_stream(int kind, uint32_t sampling_rate, const bdo_params *p,
                      const int16_t *samples, size_t n_samples, size_t callback_len,
                      int full,
                      float *values, uint8_t *candidates,
                      uint32_t *beat_idx, uint32_t *beat_ms, size_t max_beats);

/* Many streams, stream-major, on n_threads host threads (for the CPU baseline
 * timing).  Returns total beats. */
uint64_t bdo_run_batch(int kind, uint32_t sampling_rate, const bdo_params *p,
                       const int16_t *samples, size_t n_streams, size_t n_samples,
                       size_t stream_stride, size_t callback_len, int n_threads,
                       uint32_t *beat_count /*[n_streams] or NULL*/);
/* The same, and the beat lists themselves: callback index and relative_ms of the first max_beats
 * beats of every stream ([n_streams][max_beats], either may be NULL), as the harness loop of
 * lib.rs:286-292 would print them. */
uint64_t bdo_run_batch_lists(int kind, uint32_t sampling_rate, const bdo_params *p,
                             const int16_t *samples, size_t n_streams, size_t n_samples,
                             size_t stream_stride, size_t callback_len, int n_threads,
                             uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms, size_t max_beats);

/* float64 direct DFT of the retained bins, for error analysis of the f32 FFT
 * (no reference counterpart). */
double bdo_spectrum_band_max_f64(const float *ring, uint32_t sampling_rate,
                                 const bdo_params *p);

#ifdef __cplusplus
}
#endif
#endif
