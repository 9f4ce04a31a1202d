/*
 * CPU restatement of beat-detector's per-window analysis path.
 * TEST INFRASTRUCTURE ONLY -- see bd_oracle.h for who may use it and for the
 * pinning status (the FFT/scaling and the low-pass recurrence are restated
 * from third-party crates and are PARITY UNPINNED).
 *
 * Build with -ffp-contract=off and without -ffast-math: the reference is plain
 * Rust f32 arithmetic, which never fuses a*b+c and never reassociates.
 */
#include "bd_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* Rust `as` casts (saturating, NaN -> 0, truncating toward zero).     */
/* ------------------------------------------------------------------ */
static int16_t f32_as_i16(float x)
{
    if (x != x) return 0;
    if (x >= 32767.0f) return 32767;
    if (x <= -32768.0f) return -32768;
    return (int16_t)x; /* C truncates toward zero, in range here */
}

static uint32_t f32_as_u32(float x)
{
    if (x != x) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 4294967296.0f) return 4294967295u;
    return (uint32_t)x;
}

/* ------------------------------------------------------------------ */
/* Parameters                                                          */
/* ------------------------------------------------------------------ */
void bdo_default_params(bdo_params *p)
{
    /* src/lib.rs:134: (i16::MAX as f32 * 0.3) as i16 */
    p->min_amplitude = (uint32_t)f32_as_i16(32767.0f * 0.3f);
    p->min_beat_gap_ms = 400;          /* spectrum.rs:130, lpf.rs:113 */
    p->spectrum_max_hz = 90.0f;        /* spectrum.rs:85 */
    p->spectrum_threshold = 2100000.0f;/* spectrum.rs:92 */
    p->spectrum_divisor = BDO_DIV_BINS;
    p->spectrum_include_dc = 1;
    p->spectrum_inclusive = 1;
    p->spectrum_dc_packs_nyquist = 0;
    p->lpf_cutoff_hz = 120;            /* lpf.rs:65 */
    p->lpf_ratio = 0.77f;              /* lpf.rs:74 */
}

/* ------------------------------------------------------------------ */
/* AnalysisState -- src/strategies/mod.rs                              */
/* ------------------------------------------------------------------ */
void bdo_state_new(bdo_state *s, uint32_t sampling_rate)
{
    s->sampling_rate = sampling_rate;
    /* mod.rs:64: 1.0 / sampling_rate as f32 * 1000.0 */
    s->ms_per_sample = 1.0f / (float)sampling_rate * 1000.0f;
    s->beat_time_ms = 0;
    s->time_ms = 0;
    s->last_beat_timestamp = 0;
}

void bdo_state_update_time(bdo_state *s, size_t frame_len)
{
    /* mod.rs:78-83 */
    float ms_of_frame = s->ms_per_sample * (float)frame_len;
    s->beat_time_ms = s->time_ms + f32_as_u32(ms_of_frame / 2.0f);
    s->time_ms = s->time_ms + f32_as_u32(ms_of_frame);
}

void bdo_state_mark_beat(bdo_state *s)
{
    s->last_beat_timestamp = s->beat_time_ms; /* mod.rs:89-91 */
}

/* ------------------------------------------------------------------ */
/* Shared gates -- src/lib.rs:115-136                                  */
/* ------------------------------------------------------------------ */
int bdo_last_beat_beyond_threshold(const bdo_state *s, uint32_t min_gap_ms)
{
    if (s->beat_time_ms > 0) {                                 /* lib.rs:120 */
        uint32_t threshold = s->last_beat_timestamp + min_gap_ms; /* :121 */
        if (s->beat_time_ms < threshold) return 0;             /* :122-124 */
    }
    return 1;
}

int bdo_amplitude_high_enough(uint16_t window_max, uint32_t min_amplitude)
{
    return (uint32_t)window_max >= min_amplitude; /* lib.rs:135 */
}

/* ------------------------------------------------------------------ */
/* WindowStats -- src/strategies/window_stats.rs:41-55                 */
/* The reference collects |x| into a fresh Vec, sorts it and takes the */
/* last element; the structure (allocation + sort) is kept so that the */
/* CPU baseline pays what the reference pays.                          */
/* ------------------------------------------------------------------ */
static void sort_i16(int16_t *a, long n)
{
    /* non-recursive quicksort with an insertion-sort finish */
    long stack[64];
    int sp = 0;
    long lo = 0, hi = n - 1;
    for (;;) {
        while (hi - lo > 16) {
            int16_t pivot = a[lo + (hi - lo) / 2];
            long i = lo, j = hi;
            while (i <= j) {
                while (a[i] < pivot) i++;
                while (a[j] > pivot) j--;
                if (i <= j) { int16_t t = a[i]; a[i] = a[j]; a[j] = t; i++; j--; }
            }
            if (j - lo < hi - i) {
                if (i < hi) { stack[sp++] = i; stack[sp++] = hi; }
                hi = j;
            } else {
                if (lo < j) { stack[sp++] = lo; stack[sp++] = j; }
                lo = i;
            }
        }
        if (sp == 0) break;
        hi = stack[--sp];
        lo = stack[--sp];
    }
    for (long i = 1; i < n; i++) {
        int16_t v = a[i];
        long j = i - 1;
        while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; j--; }
        a[j + 1] = v;
    }
}

uint16_t bdo_window_max(const int16_t *samples, size_t n)
{
    if (n == 0) return 0; /* the reference panics (window_stats.rs:51); callers check */
    int16_t *abs_samples = (int16_t *)malloc(n * sizeof(int16_t));
    for (size_t i = 0; i < n; i++) {
        int16_t x = samples[i];
        if (x == INT16_MIN) x = (int16_t)(x + 1);      /* window_stats.rs:47 */
        abs_samples[i] = (int16_t)(x < 0 ? -x : x);    /* :48 */
    }
    sort_i16(abs_samples, (long)n);                    /* :50 */
    uint16_t max = (uint16_t)abs_samples[n - 1];       /* :51 */
    free(abs_samples);
    return max;
}

/* ------------------------------------------------------------------ */
/* lowpass_filter::simple::sp::apply_lpf_i16_sp (crate lowpass-filter  */
/* 0.2.x, src/simple/sp.rs), called at src/strategies/lpf.rs:67-71.    */
/* Published algorithm: single-pole RC low-pass,                        */
/*   rc = 1/(2*pi*fc), dt = 1/fs, alpha = dt/(rc+dt)                    */
/*   y[0] = alpha*x[0];  y[i] = y[i-1] + alpha*(x[i]-y[i-1])            */
/* evaluated in f32, in place on the i16 buffer, so every y is          */
/* truncated to i16 before it feeds the next step.  PARITY UNPINNED.    */
/* ------------------------------------------------------------------ */
static float lpf_alpha(uint16_t sample_rate_hz, uint16_t cutoff_hz)
{
    const float pi = 3.14159265358979323846f;
    float rc = 1.0f / ((float)cutoff_hz * 2.0f * pi);
    float dt = 1.0f / (float)sample_rate_hz;
    return dt / (rc + dt);
}

void bdo_apply_lpf_i16_sp(int16_t *data, size_t n, uint16_t sample_rate_hz,
                          uint16_t cutoff_hz)
{
    if (n == 0) return;
    float alpha = lpf_alpha(sample_rate_hz, cutoff_hz);
    data[0] = f32_as_i16(alpha * (float)data[0]);
    for (size_t i = 1; i < n; i++) {
        float prev = (float)data[i - 1];
        float d = (float)data[i] - prev;
        float m = alpha * d;
        data[i] = f32_as_i16(prev + m);
    }
}

/* ------------------------------------------------------------------ */
/* spectrum_analyzer::samples_fft_to_spectrum (crate spectrum-analyzer */
/* 1.1.x), called at src/strategies/spectrum.rs:82-88.                  */
/* Published algorithm: forward, unnormalised FFT of the real samples;  */
/* bins 0..N/2 mapped to k*fs/N; bins outside the FrequencyLimit        */
/* dropped; magnitude sqrt(re^2+im^2); scaling function applied to each */
/* magnitude (`divide_by_N`: value / n).  PARITY UNPINNED: what n is    */
/* (P1), whether bin 0 is kept (P2), limit inclusivity (P3) and whether */
/* bin 0 carries the Nyquist term of a packed real FFT (P6).            */
/* ------------------------------------------------------------------ */
#define FFT_N BDO_RING_LEN
#define FFT_LOG2N 10

static float g_tw_re[FFT_N / 2], g_tw_im[FFT_N / 2];
static uint16_t g_bitrev[FFT_N];
static pthread_once_t g_fft_once = PTHREAD_ONCE_INIT;

static void fft_init(void)
{
    for (int k = 0; k < FFT_N / 2; k++) {
        double a = -2.0 * 3.14159265358979323846 * (double)k / (double)FFT_N;
        g_tw_re[k] = (float)cos(a);
        g_tw_im[k] = (float)sin(a);
    }
    for (int i = 0; i < FFT_N; i++) {
        int r = 0;
        for (int b = 0; b < FFT_LOG2N; b++)
            if (i & (1 << b)) r |= 1 << (FFT_LOG2N - 1 - b);
        g_bitrev[i] = (uint16_t)r;
    }
}

/* in-order radix-2 decimation-in-time complex FFT, f32 */
static void fft1024(const float *in, float *re, float *im)
{
    pthread_once(&g_fft_once, fft_init);
    for (int i = 0; i < FFT_N; i++) {
        re[g_bitrev[i]] = in[i];
        im[g_bitrev[i]] = 0.0f;
    }
    for (int len = 2; len <= FFT_N; len <<= 1) {
        int half = len >> 1;
        int step = FFT_N / len;
        for (int base = 0; base < FFT_N; base += len) {
            for (int j = 0; j < half; j++) {
                float wr = g_tw_re[j * step], wi = g_tw_im[j * step];
                int a = base + j, b = a + half;
                float tr = re[b] * wr - im[b] * wi;
                float ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] = re[a] + tr;
                im[a] = im[a] + ti;
            }
        }
    }
}

int bdo_spectrum_bin_range(uint32_t sampling_rate, const bdo_params *p,
                           int *first_bin, int *last_bin)
{
    /* frequency of bin k = k as f32 * (sampling_rate as f32 / N as f32) */
    float res = (float)sampling_rate / (float)FFT_N;
    int first = p->spectrum_include_dc ? 0 : 1;
    int last = -1;
    for (int k = 0; k <= FFT_N / 2; k++) {
        float fr = (float)k * res;
        int keep = p->spectrum_inclusive ? (fr <= p->spectrum_max_hz)
                                         : (fr < p->spectrum_max_hz);
        if (keep) last = k; else break;
    }
    if (first_bin) *first_bin = first;
    if (last_bin) *last_bin = last;
    return last >= first ? last - first + 1 : 0;
}

static float spectrum_divisor(const bdo_params *p, int n_bins)
{
    switch (p->spectrum_divisor) {
    case BDO_DIV_BINS: return (float)n_bins;
    case BDO_DIV_FFT_LEN: return (float)FFT_N;
    default: return 1.0f;
    }
}

float bdo_spectrum_band_max(const float *ring, uint32_t sampling_rate,
                            const bdo_params *p, int *n_bins_out)
{
    float re[FFT_N], im[FFT_N];
    int first, last;
    int n_bins = bdo_spectrum_bin_range(sampling_rate, p, &first, &last);
    if (n_bins_out) *n_bins_out = n_bins;
    if (n_bins <= 0) return 0.0f;
    fft1024(ring, re, im);
    float n = spectrum_divisor(p, n_bins);
    float best = 0.0f;
    for (int k = first; k <= last; k++) {
        float r = re[k], i = im[k];
        if (k == 0 && p->spectrum_dc_packs_nyquist) i = re[FFT_N / 2];
        float mag = sqrtf(r * r + i * i);
        float scaled = (n == 0.0f) ? mag : mag / n;
        if (k == first || scaled > best) best = scaled;
    }
    return best;
}

double bdo_spectrum_band_max_f64(const float *ring, uint32_t sampling_rate,
                                 const bdo_params *p)
{
    int first, last;
    int n_bins = bdo_spectrum_bin_range(sampling_rate, p, &first, &last);
    if (n_bins <= 0) return 0.0;
    double n = (double)spectrum_divisor(p, n_bins);
    double best = 0.0;
    for (int k = first; k <= last; k++) {
        double r = 0.0, i = 0.0;
        for (int t = 0; t < FFT_N; t++) {
            int m = (int)(((long)k * t) % FFT_N);
            double a = -2.0 * 3.14159265358979323846 * (double)m / (double)FFT_N;
            r += (double)ring[t] * cos(a);
            i += (double)ring[t] * sin(a);
        }
        if (k == 0 && p->spectrum_dc_packs_nyquist) {
            double ny = 0.0;
            for (int t = 0; t < FFT_N; t++) ny += (t & 1) ? -(double)ring[t] : (double)ring[t];
            i = ny;
        }
        double mag = sqrt(r * r + i * i);
        double scaled = (n == 0.0) ? mag : mag / n;
        if (k == first || scaled > best) best = scaled;
    }
    return best;
}

/* ------------------------------------------------------------------ */
/* Detectors                                                           */
/* ------------------------------------------------------------------ */
int bdo_detector_new(bdo_detector *d, int kind, uint32_t sampling_rate,
                     const bdo_params *p)
{
    if (kind != BDO_KIND_LPF && kind != BDO_KIND_SPECTRUM) return -1;
    memset(d, 0, sizeof(*d));
    d->kind = kind;
    if (p) d->params = *p; else bdo_default_params(&d->params);
    bdo_state_new(&d->state, sampling_rate);
    /* spectrum.rs:50-51: ring pre-filled with 1024 zeros */
    for (int i = 0; i < BDO_RING_LEN; i++) d->ring[i] = 0.0f;
    d->ring_head = 0;
    return 0;
}

static void ring_to_vec(const bdo_detector *d, float *out)
{
    /* oldest first, as ConstGenericRingBuffer::to_vec (spectrum.rs:83) */
    uint32_t h = d->ring_head;
    memcpy(out, d->ring + h, (BDO_RING_LEN - h) * sizeof(float));
    memcpy(out + (BDO_RING_LEN - h), d->ring, h * sizeof(float));
}

static int spectrum_is_beat(bdo_detector *d, const int16_t *samples, size_t n,
                            uint32_t *relative_ms, int full)
{
    /* spectrum.rs:65-68: push every callback sample, overwriting the oldest */
    for (size_t i = 0; i < n; i++) {
        d->ring[d->ring_head] = (float)samples[i];
        d->ring_head = (d->ring_head + 1) % BDO_RING_LEN;
    }
    bdo_state_update_time(&d->state, n);                              /* :71 */
    int allowed = bdo_last_beat_beyond_threshold(&d->state, d->params.min_beat_gap_ms);
    d->last_computed = 0; d->last_candidate = 0; d->last_value = 0.0f;
    if (!allowed && !full) return 0;                                  /* :73-75 */
    uint16_t wmax = bdo_window_max(samples, n);                       /* :77 */
    int gate = bdo_amplitude_high_enough(wmax, d->params.min_amplitude);
    if (!gate && !full) return 0;                                     /* :78-80 */
    float vec[BDO_RING_LEN];
    ring_to_vec(d, vec);
    int n_bins;
    float v = bdo_spectrum_band_max(vec, d->state.sampling_rate, &d->params, &n_bins); /* :82-88 */
    d->last_computed = 1;
    d->last_value = v;
    d->last_candidate = gate && (v > d->params.spectrum_threshold);   /* :92 strict > */
    if (allowed && d->last_candidate) {
        bdo_state_mark_beat(&d->state);                               /* :94 */
        if (relative_ms) *relative_ms = d->state.beat_time_ms;        /* :95 */
        return 1;
    }
    return 0;
}

static int lpf_is_beat(bdo_detector *d, const int16_t *samples, size_t n,
                       uint32_t *relative_ms, int full)
{
    bdo_state_update_time(&d->state, n);                              /* lpf.rs:54 */
    int allowed = bdo_last_beat_beyond_threshold(&d->state, d->params.min_beat_gap_ms);
    d->last_computed = 0; d->last_candidate = 0; d->last_value = 0.0f;
    if (!allowed && !full) return 0;                                  /* :56-58 */
    uint16_t wmax = bdo_window_max(samples, n);                       /* :60 */
    if (!bdo_amplitude_high_enough(wmax, d->params.min_amplitude))    /* :61-63 */
        return 0;
    int16_t *copy = (int16_t *)malloc(n * sizeof(int16_t));           /* :66 */
    memcpy(copy, samples, n * sizeof(int16_t));
    bdo_apply_lpf_i16_sp(copy, n, (uint16_t)d->state.sampling_rate,   /* :67-71, `as u16` wraps */
                         (uint16_t)d->params.lpf_cutoff_hz);
    int16_t threshold = f32_as_i16(d->params.lpf_ratio * (float)wmax);/* :74 */
    int is_beat = 0;
    int ymax = 0;
    for (size_t i = 0; i < n; i++) {                                  /* :75 */
        int a = copy[i] < 0 ? -(int)copy[i] : (int)copy[i];
        if (a > ymax) ymax = a;
        if (a >= (int)threshold) is_beat = 1;
    }
    free(copy);
    d->last_computed = 1;
    d->last_value = (float)ymax;
    d->last_candidate = is_beat;
    if (allowed && is_beat) {
        bdo_state_mark_beat(&d->state);                               /* :79 */
        if (relative_ms) *relative_ms = d->state.beat_time_ms;        /* :80 */
        return 1;
    }
    return 0;
}

int bdo_is_beat(bdo_detector *d, const int16_t *samples, size_t n, uint32_t *relative_ms)
{
    if (n == 0) return -1; /* window_stats.rs:51 unwrap on an empty Vec panics */
    return d->kind == BDO_KIND_SPECTRUM ? spectrum_is_beat(d, samples, n, relative_ms, 0)
                                        : lpf_is_beat(d, samples, n, relative_ms, 0);
}

int bdo_is_beat_full(bdo_detector *d, const int16_t *samples, size_t n, uint32_t *relative_ms)
{
    if (n == 0) return -1;
    return d->kind == BDO_KIND_SPECTRUM ? spectrum_is_beat(d, samples, n, relative_ms, 1)
                                        : lpf_is_beat(d, samples, n, relative_ms, 1);
}

/* ------------------------------------------------------------------ */
/* Ingest converters -- src/record.rs:151-166                          */
/* ------------------------------------------------------------------ */
void bdo_u16_data_to_i16(const uint16_t *in, int16_t *out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        int32_t x = (int32_t)in[i];            /* record.rs:154 */
        x = x - (int32_t)INT16_MAX / 2;        /* :155  (32767/2 = 16383) */
        out[i] = (int16_t)(uint16_t)(uint32_t)x; /* :156 wrapping `as i16` */
    }
}

void bdo_f32_data_to_i16(const float *in, int16_t *out, size_t n)
{
    for (size_t i = 0; i < n; i++)
        out[i] = f32_as_i16(in[i] * 32767.0f); /* record.rs:163-164 */
}

/* ------------------------------------------------------------------ */
/* Offline harness -- src/lib.rs:263-337                               */
/* ------------------------------------------------------------------ */
void bdo_stereo_to_mono(const int16_t *interleaved, size_t n_frames, int16_t *mono)
{
    for (size_t i = 0; i < n_frames; i++) {
        int32_t s = (int32_t)interleaved[2 * i];
        int32_t t = (int32_t)interleaved[2 * i + 1];
        mono[i] = f32_as_i16((float)(s + t) / 2.0f); /* lib.rs:323 */
    }
}

size_t bdo_harness_padded_len(size_t n, size_t window_len)
{
    /* lib.rs:271-274: appends `remainder` zeros, not window_len - remainder */
    size_t remainder = n % window_len;
    return n + remainder;
}

size_t bdo_run_stream(int kind, uint32_t sampling_rate, const bdo_params *p,
                      const int16_t *samples, size_t n_samples, size_t callback_len,
                      int full, float *values, uint8_t *candidates,
                      uint32_t *beat_idx, uint32_t *beat_ms, size_t max_beats)
{
    bdo_detector *d = (bdo_detector *)malloc(sizeof(bdo_detector));
    if (bdo_detector_new(d, kind, sampling_rate, p) != 0 || callback_len == 0) { free(d); return 0; }
    size_t n_cb = n_samples / callback_len; /* lib.rs:276 */
    size_t beats = 0;
    for (size_t j = 0; j < n_cb; j++) {     /* lib.rs:286-292 */
        uint32_t ms = 0;
        const int16_t *w = samples + j * callback_len;
        int r = full ? bdo_is_beat_full(d, w, callback_len, &ms)
                     : bdo_is_beat(d, w, callback_len, &ms);
        if (values) values[j] = d->last_value;
        if (candidates) candidates[j] = (uint8_t)d->last_candidate;
        if (r == 1) {
            if (beats < max_beats) {
                if (beat_idx) beat_idx[beats] = (uint32_t)j;
                if (beat_ms) beat_ms[beats] = ms;
            }
            beats++;
        }
    }
    free(d);
    return beats;
}

typedef struct batch_job {
    int kind; uint32_t sr; const bdo_params *p; const int16_t *samples;
    size_t s0, s1, n_samples, stride, cb; uint32_t *beat_count; uint64_t total;
    uint32_t *beat_idx, *beat_ms; size_t max_beats;   /* optional: the lists themselves, [n_streams][max_beats] */
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *j = (batch_job *)arg;
    j->total = 0;
    for (size_t s = j->s0; s < j->s1; s++) {
        size_t b = bdo_run_stream(j->kind, j->sr, j->p, j->samples + s * j->stride,
                                  j->n_samples, j->cb, 0, NULL, NULL,
                                  j->beat_idx ? j->beat_idx + s * j->max_beats : NULL,
                                  j->beat_ms ? j->beat_ms + s * j->max_beats : NULL, j->max_beats);
        if (j->beat_count) j->beat_count[s] = (uint32_t)b;
        j->total += b;
    }
    return NULL;
}

uint64_t bdo_run_batch(int kind, uint32_t sampling_rate, const bdo_params *p,
                       const int16_t *samples, size_t n_streams, size_t n_samples,
                       size_t stream_stride, size_t callback_len, int n_threads,
                       uint32_t *beat_count)
{
    return bdo_run_batch_lists(kind, sampling_rate, p, samples, n_streams, n_samples, stream_stride,
                               callback_len, n_threads, beat_count, NULL, NULL, 0);
}

uint64_t bdo_run_batch_lists(int kind, uint32_t sampling_rate, const bdo_params *p,
                             const int16_t *samples, size_t n_streams, size_t n_samples,
                             size_t stream_stride, size_t callback_len, int n_threads,
                             uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms, size_t max_beats)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_streams) n_threads = (int)(n_streams ? n_streams : 1);
    pthread_once(&g_fft_once, fft_init);
    batch_job *jobs = (batch_job *)calloc((size_t)n_threads, sizeof(batch_job));
    pthread_t *tids = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].kind = kind; jobs[t].sr = sampling_rate; jobs[t].p = p;
        jobs[t].samples = samples; jobs[t].n_samples = n_samples;
        jobs[t].stride = stream_stride; jobs[t].cb = callback_len;
        jobs[t].beat_count = beat_count;
        jobs[t].beat_idx = beat_idx; jobs[t].beat_ms = beat_ms; jobs[t].max_beats = max_beats;
        jobs[t].s0 = n_streams * (size_t)t / (size_t)n_threads;
        jobs[t].s1 = n_streams * (size_t)(t + 1) / (size_t)n_threads;
        if (n_threads == 1) batch_worker(&jobs[t]);
        else pthread_create(&tids[t], NULL, batch_worker, &jobs[t]);
    }
    uint64_t total = 0;
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(tids[t], NULL);
        total += jobs[t].total;
    }
    free(jobs); free(tids);
    return total;
}
