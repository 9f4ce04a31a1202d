/*
 * beat_detector_hip.h -- C ABI of the MI355X (gfx950) implementation of
 * beat-detector's per-window analysis path.
 *
 * The reference (a Rust crate) has no FFI of its own; its boundary for this
 * path is the `Strategy` trait, the `StrategyKind` enum and
 * `record::start_listening`.  Each entry point below names the reference
 * interface it stands in for (paths relative to the reference repository).
 *
 * Conventions
 *   - every function returns an int status (BD_OK or a negative BD_ERR_*),
 *     never throws, never takes ownership of caller memory;
 *   - `bd_last_error()` returns a thread-local message for the last failure;
 *   - a handle (bd_detector, bd_live) is used by one thread at a time, like the
 *     reference's detectors (`Send`, not `Sync`: src/lib.rs:155);
 *   - "device pointer" means memory of the GPU that is current for the call
 *     (hipMalloc, or a torch CUDA tensor's data_ptr); `hip_stream` is a
 *     hipStream_t passed as void* (NULL = the default stream);
 *   - there is no CPU fallback: without a usable GPU every compute call fails
 *     with BD_ERR_NO_DEVICE / BD_ERR_HIP.
 */
#ifndef BEAT_DETECTOR_HIP_H
#define BEAT_DETECTOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BD_OK                 0
#define BD_ERR_INVALID_ARG   -1
#define BD_ERR_NO_DEVICE     -2
#define BD_ERR_HIP           -3
#define BD_ERR_UNSUPPORTED   -4
#define BD_ERR_EMPTY_WINDOW  -5  /* the reference panics: window_stats.rs:51 */
#define BD_ERR_NO_BINS       -6  /* the reference panics: spectrum.rs:88 unwrap */
#define BD_ERR_STOPPED       -7  /* record.rs:45-47 keep_recording false */

/* StrategyKind, src/lib.rs:144-149; numbering follows StrategyKind::values()
 * (src/lib.rs:182-184) and the examples' menu
 * (examples/audio_input_beat_detection.rs:95-99). */
#define BD_KIND_LPF       0
#define BD_KIND_SPECTRUM  1

/* What `divide_by_N` divides by (third-party behaviour, not pinned by the
 * reference): number of retained bins, FFT length, or nothing. */
#define BD_DIV_BINS     0
#define BD_DIV_FFT_LEN  1
#define BD_DIV_NONE     2

/* Constants of the path.  Defaults are the reference's hard-coded values; the
 * spectrum_divisor / include_dc / inclusive / dc_packs_nyquist fields select
 * among behaviours of the un-vendored spectrum-analyzer crate. */
typedef struct bd_params {
    uint32_t min_amplitude;      /* src/lib.rs:134 -> 9830 */
    uint32_t min_beat_gap_ms;    /* spectrum.rs:126-131, lpf.rs:109-114 -> 400 */
    float    spectrum_max_hz;    /* spectrum.rs:85 -> 90.0 */
    float    spectrum_threshold; /* spectrum.rs:92 -> 2_100_000.0 (strict >) */
    uint32_t spectrum_divisor;   /* BD_DIV_BINS */
    uint32_t spectrum_include_dc;/* 1 */
    uint32_t spectrum_inclusive; /* 1: keep bins with f <= max_hz */
    uint32_t spectrum_dc_packs_nyquist; /* 0 */
    uint32_t lpf_cutoff_hz;      /* lpf.rs:65 -> 120 */
    float    lpf_ratio;          /* lpf.rs:74 -> 0.77 */
} bd_params;

void bd_default_params(bd_params *p);

const char *bd_strerror(int status);
const char *bd_last_error(void);

/* Number of visible GPUs (0 and BD_OK when there is none). */
int bd_device_count(int *count);

/* The GPU that is current for the calling thread (the one every "device pointer" argument refers to),
 * and a way to choose it, for host languages without a HIP binding of their own: a program that shards
 * streams over several GPUs calls the device-pointer entries from one thread or process per GPU, each
 * after bd_set_device.  BD_ERR_NO_DEVICE for an index outside [0, bd_device_count).  No counterpart in
 * the reference (it runs on one CPU thread, src/record.rs:87). */
int bd_current_device(int *device);
int bd_set_device(int device);

/* Name and memory of GPU `device`: the analysis-side counterpart of the reference's device listing
 * helpers (record::audio_input_device_list / print_audio_input_device_configs / get_backends,
 * src/record.rs:168-198, which enumerate cpal audio devices; audio capture itself stays with the caller).
 * `name` receives a NUL-terminated string of at most `capacity` bytes (truncated if longer);
 * `total_memory_bytes` may be NULL.  BD_ERR_NO_DEVICE for an index outside [0, bd_device_count). */
int bd_device_name(int device, char *name, size_t capacity, uint64_t *total_memory_bytes);

/* ---- StrategyKind metadata ------------------------------------------- */
/* StrategyKind::values().len(), src/lib.rs:182-184 */
int bd_kind_count(void);
/* StrategyKind::name / Strategy::name, src/lib.rs:164-170, spectrum.rs:106-111,
 * lpf.rs:89-94.  NULL for an unknown kind. */
const char *bd_kind_name(int kind);
/* StrategyKind::description, src/lib.rs:173-179, spectrum.rs:113-120,
 * lpf.rs:96-103 */
const char *bd_kind_description(int kind);
/* Strategy::min_duration_between_beats_ms, spectrum.rs:126-131, lpf.rs:109-114 */
uint32_t bd_kind_min_duration_between_beats_ms(int kind);

/* ---- per-window detector --------------------------------------------- */
typedef struct bd_detector bd_detector;

/* StrategyKind::detector(sampling_rate), src/lib.rs:155-161.
 * params == NULL selects the defaults.  device < 0 selects the current device. */
int bd_detector_create(int kind, uint32_t sampling_rate, const bd_params *params,
                       int device, bd_detector **out);
/* Strategy::is_beat(&self, samples: &[i16]) -> Option<BeatInfo>, src/lib.rs:78.
 * `samples` is host memory, any n >= 1.  Returns 1 and stores
 * BeatInfo::relative_ms (src/lib.rs:63-65) on a beat, 0 when there is none,
 * BD_ERR_EMPTY_WINDOW for n == 0. */
int bd_detector_is_beat(bd_detector *d, const int16_t *samples, size_t n,
                        uint32_t *relative_ms);
/* Strategy::kind, src/lib.rs:82 */
int bd_detector_kind(const bd_detector *d);
/* Diagnostics of the last call (no reference counterpart): the analysed value
 * (Spectrum: scaled low-band maximum; LPF: max |y| after the filter) and the
 * decision before the refractory rule. */
int bd_detector_last(const bd_detector *d, float *value, int *candidate);
void bd_detector_destroy(bd_detector *d);

/* ---- batched offline analysis (new; thousands of streams per launch) --- */
/* Equivalent to running, for every stream s independently, the harness loop
 * of src/lib.rs:276-292: a fresh detector, then is_beat on consecutive
 * callbacks of `callback_len` samples; a trailing partial callback is dropped. */

/* callbacks per stream = n_samples / callback_len (src/lib.rs:276) */
uint64_t bd_batch_callbacks(uint64_t n_samples, uint32_t callback_len);
/* an upper bound on beats per stream for these arguments */
uint32_t bd_batch_max_beats(uint32_t kind, uint32_t sampling_rate, const bd_params *params,
                            uint64_t n_samples, uint32_t callback_len);

/* All pointers are device pointers on the current device.
 *   samples   [n_streams][stream_stride] i16, the first n_samples of each row used
 *   values    [n_streams][out_stride] f32 or NULL    analysed value per callback
 *   flags     [n_streams][out_stride] u8             decision before the refractory rule
 *             (the first n_callbacks entries of each row are written; out_stride = 0
 *             means n_callbacks; a multiple of 128 makes every store a full line)
 *   beat_count[n_streams] u32                        beats found (may exceed max_beats)
 *   beat_idx  [n_streams][max_beats] u32 or NULL     callback index of each beat
 *   beat_ms   [n_streams][max_beats] u32 or NULL     BeatInfo::relative_ms of each beat
 * The call only enqueues work on `hip_stream`; it allocates nothing and does
 * not synchronise. */
int bd_analyze_batch(int kind, uint32_t sampling_rate, const bd_params *params,
                     const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                     uint64_t stream_stride, uint32_t callback_len,
                     float *values, uint8_t *flags, uint64_t out_stride,
                     uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                     uint32_t max_beats, void *hip_stream);

/* Streams of different lengths in one batch (tracks of a library): row s holds
 * lengths[s] <= max_samples samples (`lengths` is a device array of n_streams
 * u64); the rest of the row may hold anything.  Stream s gets the beats of its
 * own lengths[s] / callback_len callbacks, exactly as if analysed alone
 * (src/lib.rs:276-292 drops a trailing partial window per file); values / flags
 * entries past a stream's own callbacks are written but meaningless. */
int bd_analyze_batch_ragged(int kind, uint32_t sampling_rate, const bd_params *params,
                            const int16_t *samples, uint64_t n_streams, const uint64_t *lengths,
                            uint64_t max_samples, uint64_t stream_stride, uint32_t callback_len,
                            float *values, uint8_t *flags, uint64_t out_stride,
                            uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                            uint32_t max_beats, void *hip_stream);

/* First stage of bd_analyze_batch alone: per-callback value and decision
 * (amplitude gate + strategy decision, src/strategies/spectrum.rs:77-92 /
 * lpf.rs:60-75), without the refractory rule.  Same pointer rules. */
int bd_batch_candidates(int kind, uint32_t sampling_rate, const bd_params *params,
                        const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                        uint64_t stream_stride, uint32_t callback_len,
                        float *values, uint8_t *flags, uint64_t out_stride, void *hip_stream);

/* Same analysis with every pointer in host memory; copies to `device`
 * (< 0: current), runs, copies back and synchronises.  values/flags/beat_idx/
 * beat_ms may be NULL. */
int bd_analyze_batch_host(int kind, uint32_t sampling_rate, const bd_params *params,
                          const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                          uint64_t stream_stride, uint32_t callback_len,
                          float *values, uint8_t *flags,
                          uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                          uint32_t max_beats, int device);

/* Stream-sharded over several GPUs of one node: contiguous stream ranges, one
 * host thread per device, no inter-GPU traffic.  Host pointers. */
int bd_analyze_batch_multi(int kind, uint32_t sampling_rate, const bd_params *params,
                           const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                           uint64_t stream_stride, uint32_t callback_len,
                           float *values, uint8_t *flags,
                           uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                           uint32_t max_beats, const int *devices, int n_devices);

/* The same sharding for samples that already live on the GPUs: shard g is a
 * batch of n_streams[g] streams resident on devices[g], and samples[g] /
 * values[g] / flags[g] / beat_count[g] / beat_idx[g] / beat_ms[g] are device
 * pointers there, with the meaning and the NULL rules of bd_analyze_batch
 * (the arrays themselves are host memory; `values`, `beat_idx`, `beat_ms`
 * may be NULL as a whole).  One host thread per shard selects its device,
 * enqueues the batch on a stream of its own and waits for it; the call
 * returns when every shard has finished.  No inter-GPU traffic: streams are
 * independent (src/strategies/mod.rs:39-56, spectrum.rs:38-44).  A device may
 * be listed more than once (its shards then run one after the other or
 * concurrently, as the device schedules them). */
int bd_analyze_batch_devices(int kind, uint32_t sampling_rate, const bd_params *params,
                             const int16_t **samples, const uint64_t *n_streams, uint64_t n_samples,
                             uint64_t stream_stride, uint32_t callback_len,
                             float **values, uint8_t **flags, uint64_t out_stride,
                             uint32_t **beat_count, uint32_t **beat_idx, uint32_t **beat_ms,
                             uint32_t max_beats, const int *devices, int n_devices);

/* The refractory rule alone (Strategy::last_beat_beyond_threshold,
 * src/lib.rs:115-127, with the clock of src/strategies/mod.rs:76-84) applied
 * to per-callback decisions already on the device (rows of flags_stride
 * bytes, 0 = n_callbacks). */
int bd_refractory(uint32_t sampling_rate, const bd_params *params, uint32_t callback_len,
                  const uint8_t *flags, uint64_t flags_stride, uint64_t n_streams, uint64_t n_callbacks,
                  uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                  uint32_t max_beats, void *hip_stream);

/* bd_analyze_batch without the `flags` / `values` arguments, for callers that
 * cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.
 * The per-callback decisions that bd_analyze_batch keeps in the caller's
 * `flags` live in a fixed device array inside the library (one copy per GPU,
 * BD_NOALLOC_SCRATCH_BYTES, held until the process ends); the call allocates
 * and frees nothing and only enqueues on `hip_stream`.  Limit: n_streams *
 * (n_samples / callback_len) <= BD_NOALLOC_SCRATCH_BYTES (one byte per
 * callback), else BD_ERR_UNSUPPORTED.  The first call on a device looks up that
 * device's copy of the array (make it before a graph capture starts).  All calls on one GPU share that array: calls that overlap
 * in time on the same GPU race and must be serialised by the caller (calls on
 * one HIP stream are).  Beat lists equal bd_analyze_batch's. */
#define BD_NOALLOC_SCRATCH_BYTES (96ull << 20)
int bd_analyze_batch_noalloc(int kind, uint32_t sampling_rate, const bd_params *params,
                             const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                             uint64_t stream_stride, uint32_t callback_len,
                             uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                             uint32_t max_beats, void *hip_stream);

/* ---- live path ----------------------------------------------------------- */
/* Sample formats of an input device (cpal::SampleFormat, src/record.rs:90-133). */
#define BD_FMT_I16 0
#define BD_FMT_U16 1   /* converted by record::u16_data_to_i16, src/record.rs:151-158 */
#define BD_FMT_F32 2   /* converted by record::f32_data_to_i16, src/record.rs:160-166 */

/* Multi-stream live engine (new): n_streams independent detectors advanced
 * together, one callback per stream and tick, through a ring of `depth` slots
 * (0 = 4) of pinned staging buffers; each tick replays a captured HIP graph
 * (copy in, convert, analyse, copy out) on one stream.  Callbacks of at most
 * max_callback_len samples. */
typedef struct bd_live bd_live;
typedef void (*bd_live_on_beat_fn)(void *user, uint32_t stream, uint32_t relative_ms);

int bd_live_create(int kind, uint32_t sampling_rate, const bd_params *params,
                   uint32_t n_streams, uint32_t max_callback_len, uint32_t depth,
                   int device, bd_live **out);
/* Optional, before the first submit: beats are then delivered by a dispatcher
 * thread, in tick order then stream order (the on_beat contract of
 * src/record.rs:96 for many streams), and slots are recycled automatically. */
int bd_live_set_on_beat(bd_live *live, bd_live_on_beat_fn on_beat, void *user);
/* One tick: `samples` is host memory [n_streams][n] in `format`.  Returns once the
 * tick is staged and enqueued; *tick_id numbers ticks from 0. */
int bd_live_submit(bd_live *live, const void *samples, uint32_t n, int format, uint64_t *tick_id);
/* One tick in which every stream brings its own number of samples.  In the reference every stream is
 * a device of its own whose closure receives its own data.len() (src/record.rs:93-95, :107-109,
 * :121-123; "mostly 500-540" samples on Linux, src/strategies/spectrum.rs:40-42), so concurrent streams
 * do not deliver equal lengths on one tick.  `samples` is host memory [n_streams][row_pitch] in `format`
 * (row_pitch <= max_callback_len; the whole block is copied); stream m's callback is the first
 * n_per_stream[m] <= row_pitch samples of its row.  A stream with n_per_stream[m] == 0 skips the tick:
 * no is_beat call happens for it, its ring and clock do not advance, its result is "no beat". */
int bd_live_submit_ragged(bd_live *live, const void *samples, const uint32_t *n_per_stream, uint32_t row_pitch,
                          int format, uint64_t *tick_id);
/* Result of a tick: 1 = done (outputs filled, each [n_streams] or NULL), 0 = not
 * finished yet (wait == 0), < 0 error.  Without an on_beat callback every tick must
 * be polled before its slot is reused. */
int bd_live_poll(bd_live *live, uint64_t tick_id, int wait, uint8_t *beats, uint32_t *relative_ms,
                 float *values);
void bd_live_destroy(bd_live *live);

/* record::start_listening(on_beat_cb, input_dev, strategy, keep_recording)
 * -> Result<JoinHandle<()>, String>, src/record.rs:39-44, with the cpal device
 * replaced by a source of callbacks.  `read` blocks until the next callback is
 * available, writes at most `capacity` samples of `sample_format` and returns
 * their number (0 ends the stream).  Spawns a thread (record.rs:87) that feeds
 * every callback to a detector and calls on_beat on that thread for every beat
 * (record.rs:95-97) until *keep_recording becomes 0 (record.rs:141-145).
 * Errors as the reference: BD_ERR_STOPPED if keep_recording is already false
 * (record.rs:45-47), BD_ERR_NO_DEVICE without a source (record.rs:51-55). */
typedef void (*bd_on_beat_fn)(void *user, uint32_t relative_ms);
typedef struct bd_input_source {
    uint32_t sampling_rate;      /* default_input_config().sample_rate(), record.rs:58 */
    int      sample_format;      /* BD_FMT_*, record.rs:59 */
    uint32_t max_callback_len;   /* 0 = 4096 */
    size_t (*read)(void *user, void *buffer, size_t capacity);
    void    *user;
} bd_input_source;
typedef struct bd_listener bd_listener;
int bd_start_listening(bd_on_beat_fn on_beat, void *on_beat_user, const bd_input_source *source,
                       int kind, const volatile int *keep_recording, bd_listener **out);
/* JoinHandle::join: waits for the thread, returns its status and frees the handle. */
int bd_listener_join(bd_listener *listener);

/* ---- ingest converters, src/record.rs:151-166 (device pointers) -------- */
int bd_u16_data_to_i16(const uint16_t *in, int16_t *out, uint64_t n, void *hip_stream);
int bd_f32_data_to_i16(const float *in, int16_t *out, uint64_t n, void *hip_stream);
/* Down-mix of interleaved stereo i16 frames, src/lib.rs:310-337 (the reference's
 * test harness: (l + r) / 2 truncated toward zero).  Device pointers. */
int bd_stereo_to_mono(const int16_t *interleaved, int16_t *mono, uint64_t n_frames, void *hip_stream);

/* ---- deterministic synthetic click tracks (integer arithmetic only) ---- */
/* Stream `first_stream + s` gets the same samples whatever batch it is
 * generated in.  `samples` is a device pointer. */
int bd_synth_fill(int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                  uint64_t stream_stride, uint64_t first_stream, uint64_t seed,
                  uint32_t sampling_rate, void *hip_stream);
/* The same generator on the host (no GPU needed); bit-identical output. */
int bd_synth_fill_host(int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                       uint64_t stream_stride, uint64_t first_stream, uint64_t seed,
                       uint32_t sampling_rate);

/* ---- extension analysis: STFT onset detection --------------------------- */
/* NO COUNTERPART IN THE REFERENCE, which keeps the three raw FFT bins up to
 * 90 Hz of an unwindowed 1024-sample ring and compares their maximum with a
 * constant (src/strategies/spectrum.rs:82-92).  These entry points are the
 * windowed, hop-based version of that idea: Hann STFT (1024 or 2048 points,
 * hop = half or a quarter of the length), energy of the band up to band_max_hz, spectral flux, onsets by
 * adaptive-threshold peak picking.  Results are defined by the CPU definition
 * that the tests hold (same f32 operation order: bit-identical), not by the
 * reference.  All pointers are device pointers; calls only enqueue. */
typedef struct bd_onset_params {
    uint32_t fft_len;      /* 1024 or 2048 */
    uint32_t hop;          /* fft_len / 2 (default) or fft_len / 4 */
    uint32_t window;       /* 0 rectangular, 1 periodic Hann */
    float    band_max_hz;  /* band energy over bins with k * fs / fft_len <= band_max_hz */
    uint32_t mean_frames;  /* W: threshold follows the mean of flux[j-W .. j+W] */
    uint32_t max_frames;   /* w: an onset is the maximum of flux[j-w .. j+w] */
    float    delta;        /* threshold = delta + lambda * mean */
    float    lambda;
    uint32_t min_gap_ms;   /* onsets closer than this to the previous one are dropped */
    uint32_t use_median;   /* 1: the threshold follows the median of flux[j-W .. j+W] (element c/2 of the
                              sorted window of c values) instead of the mean; W <= 32 */
    uint32_t flux_power;   /* 0 (default): flux of the magnitude spectrum, sum_k max(0, |X_j[k]| - |X_j-1[k]|);
                              1: flux of the power spectrum, sum_k max(0, |X_j[k]|^2 - |X_j-1[k]|^2), which
                              needs no square roots (about a fifth faster); band energy is the same quantity
                              in both modes */
} bd_onset_params;
void bd_onset_default_params(bd_onset_params *p);
/* complete frames of a stream: 0 if n_samples < fft_len, else (n_samples - fft_len) / hop + 1 */
uint64_t bd_onset_frames(uint64_t n_samples, uint32_t fft_len, uint32_t hop);
/* flux[s * out_stride + j], band_energy[...] for every stream s and frame j
 * (either output may be NULL; out_stride 0 = frames per stream).  Rows of
 * `samples` may start anywhere; 128-byte aligned rows read fastest. */
int bd_onset_stft(uint32_t sampling_rate, const bd_onset_params *p, const int16_t *samples,
                  uint64_t n_streams, uint64_t n_samples, uint64_t stream_stride,
                  float *flux, float *band_energy, uint64_t out_stride, void *hip_stream);
/* Peak picking over flux rows.  candidates ([n_streams][flux_stride], may be NULL)
 * gets the per-frame decision before the minimum-gap rule; onset_count[s] counts
 * every onset, onset_frame/onset_ms ([n_streams][max_onsets], may be NULL) keep
 * the first max_onsets.  Time of frame j: ((j * hop + fft_len / 2) * 1000) / fs. */
int bd_onset_pick(uint32_t sampling_rate, const bd_onset_params *p, const float *flux,
                  uint64_t n_streams, uint64_t n_frames, uint64_t flux_stride,
                  uint8_t *candidates, uint32_t *onset_count, uint32_t *onset_frame,
                  uint32_t *onset_ms, uint32_t max_onsets, void *hip_stream);
/* bd_onset_stft followed by bd_onset_pick on the same stream. */
int bd_onset_batch(uint32_t sampling_rate, const bd_onset_params *p, const int16_t *samples,
                   uint64_t n_streams, uint64_t n_samples, uint64_t stream_stride,
                   float *flux, float *band_energy, uint64_t out_stride,
                   uint32_t *onset_count, uint32_t *onset_frame, uint32_t *onset_ms,
                   uint32_t max_onsets, void *hip_stream);

/* ---- extension analysis: carried-state biquad envelope ------------------ */
/* NO COUNTERPART IN THE REFERENCE: its low-pass is first order and restarts in
 * every window (src/strategies/lpf.rs:66-71).  Here a second-order low-pass runs
 * through the whole stream (state carried across windows), evaluated in chunks
 * of 64 samples joined by a scan; per decision window the envelope max |y| is
 * compared with trunc(ratio * max |x|), gated by min_amplitude as in
 * src/lib.rs:133-136.  Results are defined by the CPU definition the tests hold
 * (same f32 operation order: bit-identical), not by the reference. */
typedef struct bd_biquad_design {
    float b0, b1, b2, na1, na2;   /* y[n] = na1 y[n-1] + na2 y[n-2] + u[n]; u = FIR part */
    float c1b, c2b;               /* two-step form: y[n+1] = c1b y[n-1] + c2b y[n-2] + na1 u[n] + u[n+1] */
    float pw[6][4];               /* P^(2^k), k = 0..5; P = state transition over 64 samples (row-major 2x2) */
    float q[64][4];               /* P^(l+1), l = 0..63 */
} bd_biquad_design;
typedef struct bd_biquad_params {
    uint32_t cutoff_hz;      /* 120 */
    float    q_factor;       /* 0.70710678 (Butterworth) */
    uint32_t window_len;     /* decision window: power of two, 64 .. 4096 (1024) */
    float    ratio;          /* 0.77, src/strategies/lpf.rs:75 */
    uint32_t min_amplitude;  /* 9830, src/lib.rs:134 */
} bd_biquad_params;
void bd_biquad_default_params(bd_biquad_params *p);
/* RBJ low-pass designed in float64 and rounded once (host only, no GPU needed). */
int bd_biquad_design_lowpass(uint32_t sampling_rate, uint32_t cutoff_hz, float q_factor, bd_biquad_design *d);
/* values[s * out_stride + w] = max |y| over window w (may be NULL), flags[...] = decision.
 * params NULL = defaults; design NULL = designed from params and sampling_rate.
 * Device pointers, enqueue only; rows of `samples` may start anywhere (16-byte aligned rows read fastest).
 * One wavefront per stream: the launch fills the GPU from about 2,000 streams up. */
int bd_biquad_envelope(uint32_t sampling_rate, const bd_biquad_params *params, const bd_biquad_design *design,
                       const int16_t *samples, uint64_t n_streams, uint64_t n_samples, uint64_t stream_stride,
                       float *values, uint8_t *flags, uint64_t out_stride, void *hip_stream);

/* ---- chunked executor: shards larger than device memory (new) ----------- */
/* One GPU's share of a stream-sharded job may exceed its memory (131,072
 * five-minute tracks over 8 GPUs: 433 GB of samples per GPU).  The executor
 * cuts the shard into chunks of `chunk_streams` streams, keeps two sample
 * buffers of one chunk each on the device, and has a *source* fill chunk c + 1
 * on its own side stream while chunk c is analysed on the caller's stream.
 * Streams are independent (one AnalysisState and one ring per detector:
 * src/strategies/mod.rs:39-56, src/strategies/spectrum.rs:38-44), so chunking
 * changes no result: stream s gets exactly the beats / onsets of
 * bd_analyze_batch / bd_onset_batch over the whole shard.
 *
 * A source enqueues, on `hip_stream`, whatever puts the samples of streams
 * [first_stream, first_stream + n_streams) of the shard into rows of `dst`
 * (device, row pitch `stream_stride` samples): a generator kernel, or copies
 * from pinned host memory after a decode.  It may block the host; it must not
 * synchronise the device.  Returns BD_OK or a negative status (ends the pass). */
typedef int (*bd_chunk_source_fn)(void *user, int16_t *dst, uint64_t first_stream, uint64_t n_streams,
                                  uint64_t n_samples, uint64_t stream_stride, void *hip_stream);
/* The synthetic click-track generator as a source (user = const bd_synth_source *):
 * stream s of the shard is stream first_stream + s of bd_synth_fill. */
typedef struct bd_synth_source {
    uint64_t seed;
    uint64_t first_stream;
    uint32_t sampling_rate;
} bd_synth_source;
int bd_chunk_source_synth(void *user, int16_t *dst, uint64_t first_stream, uint64_t n_streams,
                          uint64_t n_samples, uint64_t stream_stride, void *hip_stream);

typedef struct bd_chunked bd_chunked;
/* Allocates the two sample buffers (chunk_streams rows of n_samples, rows on
 * 128-byte lines), the side stream and the events on `device` (< 0: current). */
int bd_chunked_create(uint64_t chunk_streams, uint64_t n_samples, int device, bd_chunked **out);
/* Device memory the executor holds at the moment. */
uint64_t bd_chunked_device_bytes(const bd_chunked *ex);
/* bd_analyze_batch over n_streams streams delivered by `source`; beat_count /
 * beat_idx / beat_ms are device arrays over all n_streams ([n_streams],
 * [n_streams][max_beats]).  Enqueues the whole pass and returns; the results are
 * complete when `hip_stream` has drained.  The first pass of a given analysis
 * allocates the chunk's intermediate buffer, later passes allocate nothing. */
int bd_chunked_analyze(bd_chunked *ex, int kind, uint32_t sampling_rate, const bd_params *params,
                       uint32_t callback_len, bd_chunk_source_fn source, void *user, uint64_t n_streams,
                       uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms, uint32_t max_beats,
                       void *hip_stream);
/* bd_onset_batch (extension analysis) the same way. */
int bd_chunked_onset(bd_chunked *ex, uint32_t sampling_rate, const bd_onset_params *p,
                     bd_chunk_source_fn source, void *user, uint64_t n_streams,
                     uint32_t *onset_count, uint32_t *onset_frame, uint32_t *onset_ms, uint32_t max_onsets,
                     void *hip_stream);
/* Per chunk of the last pass: milliseconds of its dominant kernel (strategy / STFT)
 * and of the pass after it (refractory rule / peak picking), from events on the
 * caller's stream.  Waits for those chunks; returns the number of chunks of the
 * pass (the first `capacity` are written; either array may be NULL) or < 0. */
int bd_chunked_last_timing(bd_chunked *ex, float *main_ms, float *post_ms, uint32_t capacity);
void bd_chunked_destroy(bd_chunked *ex);

/* ---- instrumentation --------------------------------------------------- */
/* Name and grid of the dominant kernel of the last bd_analyze_batch call on
 * this thread (for profilers). */
const char *bd_last_kernel_name(void);
/* The library's experiment switches (environment variables BD_SPEC_*, BD_SPLIT_*, BD_LPF_VARIANT,
 * BD_HOST_*: kernel routing and staging variants used by tests and tuning scripts; every default is the
 * measured best) are read from the environment once, on first use, never on a launch path.  This call
 * reads them again; launches that run on other threads at that moment use either snapshot.
 * No counterpart in the reference. */
void bd_tuning_reload(void);
/* Read-bandwidth probe: streams `bytes` of device memory through the same kind of
 * loads the analysis kernels use and does nothing else; timing it gives the
 * ceiling those kernels are compared with.  `sink` (device, 256 * waves_per_cu
 * dwords, may be NULL) receives one word per wavefront. */
int bd_probe_read(const void *src, uint64_t bytes, uint32_t *sink, uint32_t waves_per_cu, void *hip_stream);
/* The same with `depth` (2, 4, 8 or 16) loads per lane in flight and, with `contiguous`
 * set, one contiguous share of the buffer per wavefront instead of side-by-side sweeps. */
int bd_probe_read_ex(const void *src, uint64_t bytes, uint32_t *sink, uint32_t waves_per_cu,
                     uint32_t depth, uint32_t contiguous, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* BEAT_DETECTOR_HIP_H */
