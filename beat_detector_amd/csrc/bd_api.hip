// C-ABI glue: parameters, strategy metadata, error reporting and the batched
// entry points (device pointers, host pointers, stream-sharded multi-GPU).
#include "bd_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---- errors -------------------------------------------------------------------
static thread_local char bd_err_buf[512] = "";
static thread_local const char *bd_last_kernel = "";

void bd_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(bd_err_buf, sizeof(bd_err_buf), fmt, ap);
    va_end(ap);
}

void bd_set_last_kernel(const char *name) { bd_last_kernel = name; }

int bd_check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return BD_OK;
    bd_set_error("%s: %s", what, hipGetErrorString(e));
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return BD_ERR_NO_DEVICE;
    return BD_ERR_HIP;
}

extern "C" const char *bd_last_error(void) { return bd_err_buf; }
extern "C" const char *bd_last_kernel_name(void) { return bd_last_kernel; }

extern "C" const char *bd_strerror(int status)
{
    switch (status) {
    case BD_OK: return "ok";
    case BD_ERR_INVALID_ARG: return "invalid argument";
    case BD_ERR_NO_DEVICE: return "no usable GPU";
    case BD_ERR_HIP: return "HIP runtime error";
    case BD_ERR_UNSUPPORTED: return "unsupported size or configuration";
    case BD_ERR_EMPTY_WINDOW: return "empty sample window";
    case BD_ERR_NO_BINS: return "no spectrum bin inside the frequency limit";
    case BD_ERR_STOPPED: return "keep_recording is false";
    default: return "unknown status";
    }
}

extern "C" int bd_device_count(int *count)
{
    if (!count) return BD_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return BD_OK;
}

extern "C" int bd_current_device(int *device)
{
    if (!device) { bd_set_error("bd_current_device: device is NULL"); return BD_ERR_INVALID_ARG; }
    return bd_check_hip(hipGetDevice(device), "hipGetDevice");
}

extern "C" int bd_set_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    if (device < 0 || device >= n) { bd_set_error("bd_set_device: no GPU with index %d", device); return BD_ERR_NO_DEVICE; }
    return bd_check_hip(hipSetDevice(device), "hipSetDevice");
}

extern "C" int bd_device_name(int device, char *name, size_t capacity, uint64_t *total_memory_bytes)
{
    if (!name || capacity == 0) { bd_set_error("bd_device_name: name buffer missing"); return BD_ERR_INVALID_ARG; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    if (device < 0 || device >= n) { bd_set_error("bd_device_name: no GPU with index %d", device); return BD_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    const int st = bd_check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (st != BD_OK) return st;
    snprintf(name, capacity, "%s (%s)", prop.name, prop.gcnArchName);
    if (total_memory_bytes) *total_memory_bytes = (uint64_t)prop.totalGlobalMem;
    return BD_OK;
}

// ---- experiment switches -----------------------------------------------------------
// One immutable snapshot of the BD_* environment variables; launch paths only dereference a pointer.
// bd_tuning_reload() publishes a new snapshot (the old ones stay allocated: launches running on
// other threads may still read them; a reload happens a handful of times in a test process).
static std::atomic<const bd_tuning *> bd_tuning_now{nullptr};
static std::mutex bd_tuning_mutex;

static const bd_tuning *bd_tuning_read()
{
    auto num = [](const char *name) -> long long { const char *v = getenv(name); return v ? atoll(v) : 0; };
    auto set = [](const char *name) -> int { const char *v = getenv(name); return v && *v && strcmp(v, "0") != 0; };   // "", "0": off
    bd_tuning *t = new bd_tuning();
    t->spec_force_granule = set("BD_SPEC_FORCE_GRANULE");
    t->spec_run = (int)num("BD_SPEC_RUN");
    const long long pad = getenv("BD_SPEC_LDS_PAD") ? num("BD_SPEC_LDS_PAD") : -1;
    t->spec_lds_pad = pad < 0 ? -1 : (pad > 140000 ? 140000 : (int)pad);
    t->spec_no_split = set("BD_SPEC_NO_SPLIT");
    t->spec_no_quarter = set("BD_SPEC_NO_QUARTER");
    t->refractory_wave = set("BD_REFRACTORY_WAVE");
    t->spec_no_granule = set("BD_SPEC_NO_GRANULE");
    t->spec_no_small_granule = set("BD_SPEC_NO_SMALL_GRANULE");
    t->spec_depth = (int)num("BD_SPEC_DEPTH");
    t->spec_store = (int)num("BD_SPEC_STORE");
    t->split_no_rem = set("BD_SPLIT_NO_REM");
    t->split_stage = (int)num("BD_SPLIT_STAGE");
    t->split_subruns = (int)num("BD_SPLIT_SUBRUNS");
    t->lpf_variant = (int)num("BD_LPF_VARIANT");
    t->lpf_groups = (int)num("BD_LPF_GROUPS");
    const long long thr = num("BD_HOST_THREADS");
    t->host_threads = thr > 0 ? (unsigned)thr : 0u;
    const long long chunk = num("BD_HOST_CHUNK_BYTES");
    t->host_chunk_bytes = chunk > 0 ? (uint64_t)chunk : 0ull;
    return t;
}

const bd_tuning &bd_tuning_get()
{
    const bd_tuning *t = bd_tuning_now.load(std::memory_order_acquire);
    if (t) return *t;
    std::lock_guard<std::mutex> lock(bd_tuning_mutex);
    t = bd_tuning_now.load(std::memory_order_acquire);
    if (!t) { t = bd_tuning_read(); bd_tuning_now.store(t, std::memory_order_release); }
    return *t;
}

extern "C" void bd_tuning_reload(void)
{
    std::lock_guard<std::mutex> lock(bd_tuning_mutex);
    bd_tuning_now.store(bd_tuning_read(), std::memory_order_release);
}

// ---- parameters and plan ----------------------------------------------------------
extern "C" void bd_default_params(bd_params *p)
{
    if (!p) return;
    p->min_amplitude = 9830;           // (i16::MAX as f32 * 0.3) as i16, src/lib.rs:134
    p->min_beat_gap_ms = 400;          // spectrum.rs:130, lpf.rs:113
    p->spectrum_max_hz = 90.0f;        // spectrum.rs:85
    p->spectrum_threshold = 2100000.0f;// spectrum.rs:92
    p->spectrum_divisor = BD_DIV_BINS;
    p->spectrum_include_dc = 1;
    p->spectrum_inclusive = 1;
    p->spectrum_dc_packs_nyquist = 0;
    p->lpf_cutoff_hz = 120;            // lpf.rs:65
    p->lpf_ratio = 0.77f;              // lpf.rs:74
}

int bd_make_plan(int kind, uint32_t sampling_rate, const bd_params *params, bd_plan *plan)
{
    if (kind != BD_KIND_LPF && kind != BD_KIND_SPECTRUM) {
        bd_set_error("unknown strategy kind %d", kind);
        return BD_ERR_INVALID_ARG;
    }
    if (sampling_rate == 0) { bd_set_error("sampling rate must be positive"); return BD_ERR_INVALID_ARG; }
    bd_params p;
    if (params) p = *params; else bd_default_params(&p);
    memset(plan, 0, sizeof(*plan));
    plan->kind = kind;
    plan->sampling_rate = sampling_rate;
    plan->ms_per_sample = 1.0f / (float)sampling_rate * 1000.0f; // mod.rs:64
    plan->min_amp = p.min_amplitude;
    plan->gap_ms = p.min_beat_gap_ms;
    plan->threshold = p.spectrum_threshold;
    plan->dc_packs_nyquist = p.spectrum_dc_packs_nyquist ? 1u : 0u;
    plan->lpf_ratio = p.lpf_ratio;
    if (kind == BD_KIND_SPECTRUM) {
        // bins 0..N/2 at k * (fs as f32 / N as f32), kept while inside the limit
        const float res = (float)sampling_rate / (float)BD_RING_LEN;
        int last = -1;
        for (int k = 0; k <= BD_RING_LEN / 2; k++) {
            const float fr = (float)k * res;
            const bool keep = p.spectrum_inclusive ? (fr <= p.spectrum_max_hz) : (fr < p.spectrum_max_hz);
            if (keep) last = k; else break;
        }
        plan->first_bin = p.spectrum_include_dc ? 0 : 1;
        plan->last_bin = last;
        plan->n_bins = last - plan->first_bin + 1;
        if (plan->n_bins <= 0 || !(p.spectrum_max_hz <= (float)sampling_rate / 2.0f)) {
            bd_set_error("no spectrum bin within %.1f Hz at %u Hz sampling rate", p.spectrum_max_hz, sampling_rate);
            return BD_ERR_NO_BINS;
        }
        switch (p.spectrum_divisor) {
        case BD_DIV_BINS: plan->divisor = (float)plan->n_bins; break;
        case BD_DIV_FFT_LEN: plan->divisor = (float)BD_RING_LEN; break;
        case BD_DIV_NONE: plan->divisor = 0.0f; break;
        default: bd_set_error("unknown spectrum_divisor %u", p.spectrum_divisor); return BD_ERR_INVALID_ARG;
        }
    } else {
        // lowpass-filter: rc = 1/(fc*2*pi), dt = 1/fs, alpha = dt/(rc+dt); the
        // reference passes the sampling rate `as u16` (lpf.rs:69)
        const uint16_t sr16 = (uint16_t)sampling_rate;
        const uint16_t fc16 = (uint16_t)p.lpf_cutoff_hz;
        const float pi = 3.14159265358979323846f;
        const float rc = 1.0f / ((float)fc16 * 2.0f * pi);
        const float dt = 1.0f / (float)sr16;
        const float alpha = dt / (rc + dt);
        plan->alpha = alpha;
        plan->alpha_valid = (alpha == alpha) && alpha >= 0.0f && alpha <= 1.0f;
    }
    return BD_OK;
}

void bd_clock_step(const bd_plan &plan, uint32_t callback_len, uint32_t *step_ms, uint32_t *half_ms)
{
    const float ms_of_frame = plan.ms_per_sample * (float)callback_len; // mod.rs:78
    *half_ms = bd_f32_as_u32(ms_of_frame / 2.0f);                        // mod.rs:81
    *step_ms = bd_f32_as_u32(ms_of_frame);                               // mod.rs:83
}

// ---- StrategyKind metadata ------------------------------------------------------------
extern "C" int bd_kind_count(void) { return 2; }

extern "C" const char *bd_kind_name(int kind)
{
    switch (kind) {
    case BD_KIND_LPF: return "Simple Lowpass Filter";         // lpf.rs:93
    case BD_KIND_SPECTRUM: return "Simple Spectrum Analysis"; // spectrum.rs:110
    default: return NULL;
    }
}

extern "C" const char *bd_kind_description(int kind)
{
    switch (kind) {
    case BD_KIND_LPF: // lpf.rs:100-102
        return "A simple beat detection using a lowpass filter. It's not smart enough "
               "to cope with 'complex' music, like most of today's pop. But it will give "
               "pretty good results in 'easy' music, like most of 90s pop hits.";
    case BD_KIND_SPECTRUM: // spectrum.rs:117-119
        return "A simple beat detection using a spectrum analysis. It's not smart enough "
               "to cope with 'complex' music, like most of today's pop. But it will give "
               "pretty good results in 'easy' music, like most of 90s pop hits.";
    default: return NULL;
    }
}

extern "C" uint32_t bd_kind_min_duration_between_beats_ms(int kind)
{
    return (kind == BD_KIND_LPF || kind == BD_KIND_SPECTRUM) ? 400u : 0u; // spectrum.rs:130, lpf.rs:113
}

// ---- batched analysis ---------------------------------------------------------------------
extern "C" uint64_t bd_batch_callbacks(uint64_t n_samples, uint32_t callback_len)
{
    return callback_len ? n_samples / callback_len : 0; // src/lib.rs:276
}

extern "C" uint32_t bd_batch_max_beats(uint32_t kind, uint32_t sampling_rate, const bd_params *params,
                                       uint64_t n_samples, uint32_t callback_len)
{
    bd_plan plan;
    if (callback_len == 0 || bd_make_plan((int)kind, sampling_rate, params, &plan) != BD_OK) return 0;
    const uint64_t n_cb = n_samples / callback_len;
    uint32_t step, half;
    bd_clock_step(plan, callback_len, &step, &half);
    if (step == 0 || plan.gap_ms == 0) return (uint32_t)(n_cb > 0xffffffffull ? 0xffffffffull : n_cb);
    // two beats are at least ceil(gap/step) callbacks apart
    const uint64_t spacing = (plan.gap_ms + step - 1) / step;
    const uint64_t bound = n_cb / spacing + 2;
    return (uint32_t)(bound > n_cb ? n_cb : bound);
}

static int bd_check_batch_args(const int16_t *samples, uint64_t n_samples, uint64_t stride,
                               uint32_t callback_len, const uint8_t *flags)
{
    if (callback_len == 0) { bd_set_error("callback_len must be >= 1 (the reference panics on an empty window)"); return BD_ERR_EMPTY_WINDOW; }
    // with no complete callback there is nothing to read or write
    if ((!samples || !flags) && n_samples / callback_len > 0) { bd_set_error("samples and flags must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (stride < n_samples) { bd_set_error("stream_stride (%llu) < n_samples (%llu)", (unsigned long long)stride, (unsigned long long)n_samples); return BD_ERR_INVALID_ARG; }
    return BD_OK;
}

static int bd_candidates_impl(const bd_plan &plan, const int16_t *samples, uint64_t n_streams,
                              uint64_t n_samples, uint64_t stride, uint32_t callback_len,
                              float *values, uint8_t *flags, uint64_t out_stride, hipStream_t st)
{
    if (plan.kind == BD_KIND_SPECTRUM)
        return bd_launch_spectrum(plan, samples, n_streams, n_samples, stride, callback_len, values, flags, out_stride, st);
    return bd_launch_lpf(plan, samples, n_streams, n_samples, stride, callback_len, values, flags, out_stride, st);
}

// out_stride == 0 means densely packed rows
static int bd_resolve_out_stride(uint64_t *out_stride, uint64_t n_cb)
{
    if (*out_stride == 0) *out_stride = n_cb;
    if (*out_stride < n_cb) { bd_set_error("out_stride (%llu) < callbacks per stream (%llu)", (unsigned long long)*out_stride, (unsigned long long)n_cb); return BD_ERR_INVALID_ARG; }
    return BD_OK;
}

extern "C" int bd_batch_candidates(int kind, uint32_t sampling_rate, const bd_params *params,
                                   const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                                   uint64_t stream_stride, uint32_t callback_len,
                                   float *values, uint8_t *flags, uint64_t out_stride, void *hip_stream)
{
    bd_plan plan;
    int s = bd_make_plan(kind, sampling_rate, params, &plan);
    if (s != BD_OK) return s;
    s = bd_check_batch_args(samples, n_samples, stream_stride, callback_len, flags);
    if (s != BD_OK) return s;
    s = bd_resolve_out_stride(&out_stride, n_samples / callback_len);
    if (s != BD_OK) return s;
    return bd_candidates_impl(plan, samples, n_streams, n_samples, stream_stride, callback_len,
                              values, flags, out_stride, (hipStream_t)hip_stream);
}

extern "C" int bd_refractory(uint32_t sampling_rate, const bd_params *params, uint32_t callback_len,
                             const uint8_t *flags, uint64_t flags_stride, uint64_t n_streams, uint64_t n_callbacks,
                             uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                             uint32_t max_beats, void *hip_stream)
{
    bd_plan plan;
    int s = bd_make_plan(BD_KIND_LPF, sampling_rate, params, &plan);
    if (s != BD_OK) return s;
    if (!flags || !beat_count || callback_len == 0) { bd_set_error("bd_refractory: bad arguments"); return BD_ERR_INVALID_ARG; }
    s = bd_resolve_out_stride(&flags_stride, n_callbacks);
    if (s != BD_OK) return s;
    return bd_launch_refractory(plan, callback_len, flags, flags_stride, n_streams, n_callbacks, beat_count,
                                beat_idx, beat_ms, max_beats, (hipStream_t)hip_stream);
}

extern "C" int bd_analyze_batch(int kind, uint32_t sampling_rate, const bd_params *params,
                                const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                                uint64_t stream_stride, uint32_t callback_len,
                                float *values, uint8_t *flags, uint64_t out_stride,
                                uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                                uint32_t max_beats, void *hip_stream)
{
    bd_plan plan;
    int s = bd_make_plan(kind, sampling_rate, params, &plan);
    if (s != BD_OK) return s;
    s = bd_check_batch_args(samples, n_samples, stream_stride, callback_len, flags);
    if (s != BD_OK) return s;
    if (!beat_count) { bd_set_error("beat_count must not be NULL"); return BD_ERR_INVALID_ARG; }
    s = bd_resolve_out_stride(&out_stride, n_samples / callback_len);
    if (s != BD_OK) return s;
    hipStream_t st = (hipStream_t)hip_stream;
    s = bd_candidates_impl(plan, samples, n_streams, n_samples, stream_stride, callback_len, values, flags, out_stride, st);
    if (s != BD_OK) return s;
    const char *k = bd_last_kernel; // keep the dominant kernel's name
    s = bd_launch_refractory(plan, callback_len, flags, out_stride, n_streams, n_samples / callback_len,
                             beat_count, beat_idx, beat_ms, max_beats, st);
    bd_last_kernel = k;
    return s;
}

extern "C" int bd_analyze_batch_ragged(int kind, uint32_t sampling_rate, const bd_params *params,
                                       const int16_t *samples, uint64_t n_streams, const uint64_t *lengths,
                                       uint64_t max_samples, uint64_t stream_stride, uint32_t callback_len,
                                       float *values, uint8_t *flags, uint64_t out_stride,
                                       uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                                       uint32_t max_beats, void *hip_stream)
{
    if (!lengths) { bd_set_error("bd_analyze_batch_ragged: lengths must not be NULL"); return BD_ERR_INVALID_ARG; }
    bd_plan plan;
    int s = bd_make_plan(kind, sampling_rate, params, &plan);
    if (s != BD_OK) return s;
    s = bd_check_batch_args(samples, max_samples, stream_stride, callback_len, flags);
    if (s != BD_OK) return s;
    if (!beat_count) { bd_set_error("beat_count must not be NULL"); return BD_ERR_INVALID_ARG; }
    s = bd_resolve_out_stride(&out_stride, max_samples / callback_len);
    if (s != BD_OK) return s;
    hipStream_t st = (hipStream_t)hip_stream;
    // every row is analysed to the common length: a callback depends on samples before its end
    // only, so whatever follows a stream's own samples cannot reach its own callbacks
    s = bd_candidates_impl(plan, samples, n_streams, max_samples, stream_stride, callback_len, values, flags, out_stride, st);
    if (s != BD_OK) return s;
    const char *k = bd_last_kernel;
    s = bd_launch_refractory(plan, callback_len, flags, out_stride, n_streams, max_samples / callback_len,
                             beat_count, beat_idx, beat_ms, max_beats, st, lengths);
    bd_last_kernel = k;
    return s;
}

// Host-pointer batches are processed in chunks of streams through two slots: while the GPU
// copies and analyses one chunk, host threads pack the next one into pinned memory (rows
// repacked to start on 128-byte boundaries).  Device and pinned memory stay bounded whatever
// the batch size, and the copy runs at pinned-memory PCIe speed instead of pageable speed.
struct bd_host_slot {
    int16_t *h_in = nullptr;      // pinned staging
    int16_t *d_samples = nullptr; float *d_values = nullptr; uint8_t *d_flags = nullptr;
    uint32_t *d_count = nullptr, *d_idx = nullptr, *d_ms = nullptr;
    hipStream_t st = nullptr;
    void release()
    {
        if (st) { (void)hipStreamSynchronize(st); }
        if (h_in) (void)hipHostFree(h_in);
        if (d_samples) (void)hipFree(d_samples);
        if (d_values) (void)hipFree(d_values);
        if (d_flags) (void)hipFree(d_flags);
        if (d_count) (void)hipFree(d_count);
        if (d_idx) (void)hipFree(d_idx);
        if (d_ms) (void)hipFree(d_ms);
        if (st) (void)hipStreamDestroy(st);
    }
};
struct bd_host_job {
    bd_host_slot slot[2];
    void release() { slot[0].release(); slot[1].release(); }
};

#define BD_HOST_CHUNK_BYTES (256ull << 20)
#define BD_HOST_PACK_THREADS 8

// device workers of one bd_analyze_batch_multi call share the host's packing threads
static thread_local unsigned bd_host_pack_share = 1;

static void bd_host_pack(int16_t *dst, uint64_t d_stride, const int16_t *src, uint64_t stride, uint64_t n_samples,
                         uint64_t rows)
{
    const unsigned knob = bd_tuning_get().host_threads;
    unsigned want = knob ? knob : BD_HOST_PACK_THREADS;
    want = want / bd_host_pack_share ? want / bd_host_pack_share : 1;
    const unsigned n_thr = rows >= 4 * want ? want : 1;
    auto work = [=](unsigned t) {
        for (uint64_t r = rows * t / n_thr; r < rows * (t + 1) / n_thr; r++)
            memcpy(dst + r * d_stride, src + r * stride, n_samples * sizeof(int16_t));
    };
    if (n_thr == 1) { work(0); return; }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < n_thr; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
}

static int bd_host_run(bd_host_job &j, int kind, uint32_t sampling_rate, const bd_params *params,
                       const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                       uint64_t stride, uint32_t callback_len, float *values, uint8_t *flags,
                       uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms, uint32_t max_beats)
{
    const uint64_t n_cb = n_samples / callback_len;
    if (n_cb == 0) { // no complete callback anywhere: nothing to copy or launch
        memset(beat_count, 0, n_streams * sizeof(uint32_t));
        return BD_OK;
    }
    // input and output rows are repacked so that each starts on a 128-byte
    // boundary: the fast kernels apply and every load / store covers whole lines
    const uint64_t d_stride = (n_samples + 63) & ~63ull;
    const uint64_t o_stride = (n_cb + 127) & ~127ull;
    const uint64_t chunk_knob = bd_tuning_get().host_chunk_bytes; // tests use small chunks to exercise the pipeline
    const uint64_t chunk_bytes = chunk_knob ? chunk_knob : BD_HOST_CHUNK_BYTES;
    uint64_t chunk = chunk_bytes / (d_stride * sizeof(int16_t));
    if (chunk < 1) chunk = 1;
    if (chunk > n_streams) chunk = n_streams;
    const uint64_t n_chunks = (n_streams + chunk - 1) / chunk;
    const int n_slots = n_chunks > 1 ? 2 : 1;
    for (int i = 0; i < n_slots; i++) {
        bd_host_slot &sl = j.slot[i];
        BD_HIP(hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking));
        BD_HIP(hipHostMalloc((void **)&sl.h_in, chunk * d_stride * sizeof(int16_t), hipHostMallocDefault));
        BD_HIP(hipMalloc((void **)&sl.d_samples, (chunk * d_stride + 8) * sizeof(int16_t)));
        BD_HIP(hipMalloc((void **)&sl.d_flags, chunk * o_stride));
        if (values) BD_HIP(hipMalloc((void **)&sl.d_values, chunk * o_stride * sizeof(float)));
        BD_HIP(hipMalloc((void **)&sl.d_count, (chunk + 1) * sizeof(uint32_t)));
        if (max_beats && beat_idx) BD_HIP(hipMalloc((void **)&sl.d_idx, chunk * max_beats * sizeof(uint32_t)));
        if (max_beats && beat_ms) BD_HIP(hipMalloc((void **)&sl.d_ms, chunk * max_beats * sizeof(uint32_t)));
    }
    // results of chunk c to the caller's (pageable) memory; such copies may hold the host until
    // the chunk's kernels have finished, so they are issued after the next chunk has been
    // packed and enqueued
    auto drain = [&](uint64_t c) -> int {
        bd_host_slot &sl = j.slot[c & 1];
        const uint64_t s0 = c * chunk, rows = (s0 + chunk <= n_streams) ? chunk : n_streams - s0;
        if (values) BD_HIP(hipMemcpy2DAsync(values + s0 * n_cb, n_cb * sizeof(float), sl.d_values, o_stride * sizeof(float),
                                            n_cb * sizeof(float), rows, hipMemcpyDeviceToHost, sl.st));
        if (flags) BD_HIP(hipMemcpy2DAsync(flags + s0 * n_cb, n_cb, sl.d_flags, o_stride, n_cb, rows,
                                           hipMemcpyDeviceToHost, sl.st));
        BD_HIP(hipMemcpyAsync(beat_count + s0, sl.d_count, rows * sizeof(uint32_t), hipMemcpyDeviceToHost, sl.st));
        if (sl.d_idx) BD_HIP(hipMemcpyAsync(beat_idx + s0 * max_beats, sl.d_idx, rows * max_beats * sizeof(uint32_t), hipMemcpyDeviceToHost, sl.st));
        if (sl.d_ms) BD_HIP(hipMemcpyAsync(beat_ms + s0 * max_beats, sl.d_ms, rows * max_beats * sizeof(uint32_t), hipMemcpyDeviceToHost, sl.st));
        return BD_OK;
    };
    for (uint64_t c = 0; c < n_chunks; c++) {
        bd_host_slot &sl = j.slot[c & 1];
        const uint64_t s0 = c * chunk, rows = (s0 + chunk <= n_streams) ? chunk : n_streams - s0;
        BD_HIP(hipStreamSynchronize(sl.st)); // the slot's previous chunk is complete, results included
        bd_host_pack(sl.h_in, d_stride, samples + s0 * stride, stride, n_samples, rows);
        BD_HIP(hipMemcpyAsync(sl.d_samples, sl.h_in, rows * d_stride * sizeof(int16_t), hipMemcpyHostToDevice, sl.st));
        int s = bd_analyze_batch(kind, sampling_rate, params, sl.d_samples, rows, n_samples, d_stride,
                                 callback_len, sl.d_values, sl.d_flags, o_stride, sl.d_count, sl.d_idx, sl.d_ms,
                                 max_beats, sl.st);
        if (s != BD_OK) return s;
        if (c > 0) { s = drain(c - 1); if (s != BD_OK) return s; }
    }
    { int s = drain(n_chunks - 1); if (s != BD_OK) return s; }
    for (int i = 0; i < n_slots; i++) BD_HIP(hipStreamSynchronize(j.slot[i].st));
    return BD_OK;
}

extern "C" int bd_analyze_batch_host(int kind, uint32_t sampling_rate, const bd_params *params,
                                     const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                                     uint64_t stream_stride, uint32_t callback_len,
                                     float *values, uint8_t *flags,
                                     uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                                     uint32_t max_beats, int device)
{
    bd_plan plan;
    int s = bd_make_plan(kind, sampling_rate, params, &plan);
    if (s != BD_OK) return s;
    if (!samples || !beat_count) { bd_set_error("samples and beat_count must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (callback_len == 0) { bd_set_error("callback_len must be >= 1"); return BD_ERR_EMPTY_WINDOW; }
    if (stream_stride < n_samples) { bd_set_error("stream_stride < n_samples"); return BD_ERR_INVALID_ARG; }
    if (n_streams == 0) return BD_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        bd_set_error("no HIP device available (this library has no CPU path)");
        return BD_ERR_NO_DEVICE;
    }
    int prev = 0;
    BD_HIP(hipGetDevice(&prev));
    if (device < 0) device = prev;
    if (device >= count) { bd_set_error("device %d out of range (%d devices)", device, count); return BD_ERR_INVALID_ARG; }
    if (device != prev) BD_HIP(hipSetDevice(device));
    bd_host_job j;
    s = bd_host_run(j, kind, sampling_rate, params, samples, n_streams, n_samples, stream_stride,
                    callback_len, values, flags, beat_count, beat_idx, beat_ms, max_beats);
    j.release();
    if (device != prev) (void)hipSetDevice(prev);
    return s;
}

extern "C" int bd_analyze_batch_multi(int kind, uint32_t sampling_rate, const bd_params *params,
                                      const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                                      uint64_t stream_stride, uint32_t callback_len,
                                      float *values, uint8_t *flags,
                                      uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                                      uint32_t max_beats, const int *devices, int n_devices)
{
    if (!devices || n_devices <= 0) { bd_set_error("bd_analyze_batch_multi: empty device list"); return BD_ERR_INVALID_ARG; }
    // argument errors are reported once, here, before any worker starts or any pointer is offset
    {
        bd_plan plan;
        const int s = bd_make_plan(kind, sampling_rate, params, &plan);
        if (s != BD_OK) return s;
    }
    if (!samples || !beat_count) { bd_set_error("samples and beat_count must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (callback_len == 0) { bd_set_error("callback_len must be >= 1"); return BD_ERR_EMPTY_WINDOW; }
    if (stream_stride < n_samples) { bd_set_error("stream_stride < n_samples"); return BD_ERR_INVALID_ARG; }
    if (n_streams == 0) return BD_OK;
    {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            bd_set_error("no HIP device available (this library has no CPU path)");
            return BD_ERR_NO_DEVICE;
        }
        for (int g = 0; g < n_devices; g++)
            if (devices[g] < 0 || devices[g] >= count) {
                bd_set_error("device %d out of range (%d devices)", devices[g], count);
                return BD_ERR_INVALID_ARG;
            }
    }
    const uint64_t n_cb = n_samples / callback_len;
    std::vector<int> status((size_t)n_devices, BD_OK);
    std::vector<std::string> messages((size_t)n_devices);
    std::vector<std::thread> workers;
    for (int g = 0; g < n_devices; g++) {
        // contiguous stream ranges, no data shared between devices
        const uint64_t s0 = n_streams * (uint64_t)g / (uint64_t)n_devices;
        const uint64_t s1 = n_streams * (uint64_t)(g + 1) / (uint64_t)n_devices;
        workers.emplace_back([=, &status, &messages] {
            if (s1 == s0) return;
            bd_host_pack_share = (unsigned)n_devices;
            status[(size_t)g] = bd_analyze_batch_host(
                kind, sampling_rate, params, samples + s0 * stream_stride, s1 - s0, n_samples,
                stream_stride, callback_len,
                values ? values + s0 * n_cb : nullptr, flags ? flags + s0 * n_cb : nullptr,
                beat_count + s0, beat_idx ? beat_idx + s0 * max_beats : nullptr,
                beat_ms ? beat_ms + s0 * max_beats : nullptr, max_beats, devices[g]);
            if (status[(size_t)g] != BD_OK) messages[(size_t)g] = bd_last_error();
        });
    }
    for (auto &w : workers) w.join();
    for (int g = 0; g < n_devices; g++)
        if (status[(size_t)g] != BD_OK) {
            bd_set_error("device %d: %s", devices[g], messages[(size_t)g].c_str());
            return status[(size_t)g];
        }
    return BD_OK;
}

extern "C" int bd_analyze_batch_devices(int kind, uint32_t sampling_rate, const bd_params *params,
                                        const int16_t **samples, const uint64_t *n_streams, uint64_t n_samples,
                                        uint64_t stream_stride, uint32_t callback_len,
                                        float **values, uint8_t **flags, uint64_t out_stride,
                                        uint32_t **beat_count, uint32_t **beat_idx, uint32_t **beat_ms,
                                        uint32_t max_beats, const int *devices, int n_devices)
{
    if (!devices || n_devices <= 0) { bd_set_error("bd_analyze_batch_devices: empty device list"); return BD_ERR_INVALID_ARG; }
    if (!samples || !n_streams || !flags || !beat_count) {
        bd_set_error("bd_analyze_batch_devices: samples, n_streams, flags and beat_count must not be NULL");
        return BD_ERR_INVALID_ARG;
    }
    // argument errors are reported once, here, before any worker starts
    {
        bd_plan plan;
        int s = bd_make_plan(kind, sampling_rate, params, &plan);
        if (s != BD_OK) return s;
        if (callback_len == 0) { bd_set_error("callback_len must be >= 1"); return BD_ERR_EMPTY_WINDOW; }
        if (stream_stride < n_samples) { bd_set_error("stream_stride < n_samples"); return BD_ERR_INVALID_ARG; }
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            bd_set_error("no HIP device available (this library has no CPU path)");
            return BD_ERR_NO_DEVICE;
        }
        for (int g = 0; g < n_devices; g++) {
            if (devices[g] < 0 || devices[g] >= count) {
                bd_set_error("device %d out of range (%d devices)", devices[g], count);
                return BD_ERR_INVALID_ARG;
            }
            if (n_streams[g] > 0 && (!samples[g] || !flags[g] || !beat_count[g])) {
                bd_set_error("shard %d: samples, flags and beat_count must not be NULL", g);
                return BD_ERR_INVALID_ARG;
            }
        }
    }
    std::vector<int> status((size_t)n_devices, BD_OK);
    std::vector<std::string> messages((size_t)n_devices);
    std::vector<std::thread> workers;
    for (int g = 0; g < n_devices; g++) {
        if (n_streams[g] == 0) continue;
        workers.emplace_back([=, &status, &messages] {
            auto fail = [&](int st) { status[(size_t)g] = st; messages[(size_t)g] = bd_last_error(); };
            int st = bd_check_hip(hipSetDevice(devices[g]), "hipSetDevice");
            if (st != BD_OK) { fail(st); return; }
            hipStream_t stream = nullptr;
            st = bd_check_hip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
            if (st != BD_OK) { fail(st); return; }
            st = bd_analyze_batch(kind, sampling_rate, params, samples[g], n_streams[g], n_samples, stream_stride,
                                  callback_len, values ? values[g] : nullptr, flags[g], out_stride, beat_count[g],
                                  beat_idx ? beat_idx[g] : nullptr, beat_ms ? beat_ms[g] : nullptr, max_beats, stream);
            if (st != BD_OK) fail(st);
            const int sync = bd_check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
            if (st == BD_OK && sync != BD_OK) fail(sync);
            (void)hipStreamDestroy(stream);
        });
    }
    for (auto &w : workers) w.join();
    for (int g = 0; g < n_devices; g++)
        if (status[(size_t)g] != BD_OK) {
            bd_set_error("shard %d on device %d: %s", g, devices[g], messages[(size_t)g].c_str());
            return status[(size_t)g];
        }
    return BD_OK;
}
