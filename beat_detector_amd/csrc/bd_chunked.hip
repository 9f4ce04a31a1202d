// Chunked executor: one GPU analyses a shard of streams that does not fit its
// memory (131,072 five-minute tracks over 8 GPUs are 433 GB of samples per GPU
// against 288 GB of HBM).
//
// The shard is cut into chunks of `chunk_streams` streams.  Two sample buffers
// of one chunk each live on the device.  A *source* (the synthetic generator, or
// a caller's decode + copy stage) fills the buffer of chunk c + 1 on the
// executor's side stream while the analysis kernels of chunk c run on the
// caller's stream; events order "filled -> analysed -> refilled" per buffer, the
// host never waits inside a pass.  Streams are independent (each has its own
// detector state in the reference: src/strategies/mod.rs:39-56,
// src/strategies/spectrum.rs:38-44), so a chunk's results are simply the rows
// [first, first + rows) of the caller's per-stream outputs.
//
// Everything is allocated by bd_chunked_create, except the one intermediate
// buffer (per-callback decisions or per-frame flux of one chunk) whose size
// depends on the analysis: it grows on the first pass that needs more.
#include "bd_internal.h"

#include <vector>

struct bd_chunked {
    int device = 0;
    uint64_t chunk_streams = 0, n_samples = 0, stride = 0;
    int16_t *buf[2] = {nullptr, nullptr};
    hipStream_t fill = nullptr;
    hipEvent_t filled[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    bool consumed_valid[2] = {false, false};
    void *mid = nullptr;       // intermediates of the chunk being analysed
    uint64_t mid_bytes = 0;
    // three timing events per chunk of the last pass: before the analysis, after its dominant
    // kernel, after its last kernel (created on demand, reused by later passes)
    std::vector<hipEvent_t> t_ev;
    uint64_t timed_chunks = 0;
};

extern "C" int bd_chunk_source_synth(void *user, int16_t *dst, uint64_t first_stream, uint64_t n_streams,
                                     uint64_t n_samples, uint64_t stream_stride, void *hip_stream)
{
    if (!user) { bd_set_error("bd_chunk_source_synth: user must point to a bd_synth_source"); return BD_ERR_INVALID_ARG; }
    const bd_synth_source *src = static_cast<const bd_synth_source *>(user);
    return bd_synth_fill(dst, n_streams, n_samples, stream_stride, src->first_stream + first_stream, src->seed,
                         src->sampling_rate, hip_stream);
}

extern "C" void bd_chunked_destroy(bd_chunked *ex)
{
    if (!ex) return;
    int prev = 0;
    const bool switched = hipGetDevice(&prev) == hipSuccess && prev != ex->device && hipSetDevice(ex->device) == hipSuccess;
    if (ex->fill) (void)hipStreamSynchronize(ex->fill);
    for (int i = 0; i < 2; i++) {
        if (ex->consumed_valid[i]) (void)hipEventSynchronize(ex->consumed[i]);
        if (ex->buf[i]) (void)hipFree(ex->buf[i]);
        if (ex->filled[i]) (void)hipEventDestroy(ex->filled[i]);
        if (ex->consumed[i]) (void)hipEventDestroy(ex->consumed[i]);
    }
    if (ex->mid) (void)hipFree(ex->mid);
    for (hipEvent_t e : ex->t_ev) (void)hipEventDestroy(e);
    if (ex->fill) (void)hipStreamDestroy(ex->fill);
    if (switched) (void)hipSetDevice(prev);
    delete ex;
}

static int bd_chunked_init(bd_chunked *ex)
{
    const uint64_t bytes = ex->chunk_streams * ex->stride * sizeof(int16_t) + 128; // kernels read whole 16-byte units
    for (int i = 0; i < 2; i++) {
        BD_HIP(hipMalloc((void **)&ex->buf[i], bytes));
        BD_HIP(hipEventCreateWithFlags(&ex->filled[i], hipEventDisableTiming));
        BD_HIP(hipEventCreateWithFlags(&ex->consumed[i], hipEventDisableTiming));
    }
    BD_HIP(hipStreamCreateWithFlags(&ex->fill, hipStreamNonBlocking));
    return BD_OK;
}

extern "C" int bd_chunked_create(uint64_t chunk_streams, uint64_t n_samples, int device, bd_chunked **out)
{
    if (!out) { bd_set_error("bd_chunked_create: out must not be NULL"); return BD_ERR_INVALID_ARG; }
    *out = nullptr;
    if (chunk_streams == 0 || n_samples == 0) { bd_set_error("bd_chunked_create: chunk_streams and n_samples must be positive"); return BD_ERR_INVALID_ARG; }
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
    bd_chunked *ex = new bd_chunked;
    ex->device = device;
    ex->chunk_streams = chunk_streams;
    ex->n_samples = n_samples;
    ex->stride = (n_samples + 63) & ~63ull; // rows start on 128-byte lines
    const int s = bd_chunked_init(ex);
    if (device != prev) (void)hipSetDevice(prev);
    if (s != BD_OK) { bd_chunked_destroy(ex); return s; }
    *out = ex;
    return BD_OK;
}

extern "C" uint64_t bd_chunked_device_bytes(const bd_chunked *ex)
{
    if (!ex) return 0;
    return 2 * (ex->chunk_streams * ex->stride * sizeof(int16_t) + 128) + ex->mid_bytes;
}

static int bd_chunked_ensure_mid(bd_chunked *ex, uint64_t bytes, hipStream_t run)
{
    if (bytes <= ex->mid_bytes) return BD_OK;
    if (ex->mid) { // a previous pass may still be reading it
        BD_HIP(hipStreamSynchronize(run));
        for (int i = 0; i < 2; i++) if (ex->consumed_valid[i]) BD_HIP(hipEventSynchronize(ex->consumed[i]));
        BD_HIP(hipFree(ex->mid));
        ex->mid = nullptr; ex->mid_bytes = 0;
    }
    BD_HIP(hipMalloc(&ex->mid, bytes));
    ex->mid_bytes = bytes;
    return BD_OK;
}

// One pass over `n_streams` streams; `analyse(samples, first, rows, mid_event)` enqueues the analysis of a
// filled chunk on `run` and records `mid_event` after its dominant kernel.
template <class F>
static int bd_chunked_pass(bd_chunked *ex, bd_chunk_source_fn source, void *user, uint64_t n_streams, hipStream_t run,
                           F analyse)
{
    const uint64_t chunk = ex->chunk_streams;
    const uint64_t n_chunks = (n_streams + chunk - 1) / chunk;
    ex->timed_chunks = 0;
    while (ex->t_ev.size() < 3 * n_chunks) {
        hipEvent_t e = nullptr;
        BD_HIP(hipEventCreate(&e));
        ex->t_ev.push_back(e);
    }
    auto fill = [&](uint64_t c) -> int {
        const int slot = (int)(c & 1);
        const uint64_t first = c * chunk, rows = first + chunk <= n_streams ? chunk : n_streams - first;
        // the buffer is free once the analysis that read it last has finished
        if (ex->consumed_valid[slot]) BD_HIP(hipStreamWaitEvent(ex->fill, ex->consumed[slot], 0));
        const int s = source(user, ex->buf[slot], first, rows, ex->n_samples, ex->stride, (void *)ex->fill);
        if (s != BD_OK) return s < 0 ? s : BD_ERR_INVALID_ARG;
        BD_HIP(hipEventRecord(ex->filled[slot], ex->fill));
        return BD_OK;
    };
    // The intermediate buffer (decisions / flux) is shared by every pass: a pass on another stream than the
    // one before it waits until that pass's last analyses, which may still use the buffer, have finished.
    for (int slot = 0; slot < 2; slot++)
        if (ex->consumed_valid[slot]) BD_HIP(hipStreamWaitEvent(run, ex->consumed[slot], 0));
    if (n_chunks) { const int s = fill(0); if (s != BD_OK) return s; }
    for (uint64_t c = 0; c < n_chunks; c++) {
        const int slot = (int)(c & 1);
        const uint64_t first = c * chunk, rows = first + chunk <= n_streams ? chunk : n_streams - first;
        if (c + 1 < n_chunks) { const int s = fill(c + 1); if (s != BD_OK) return s; } // runs beside the analysis of chunk c
        BD_HIP(hipStreamWaitEvent(run, ex->filled[slot], 0));
        BD_HIP(hipEventRecord(ex->t_ev[3 * c], run));
        const int s = analyse(ex->buf[slot], first, rows, ex->t_ev[3 * c + 1]);
        if (s != BD_OK) return s;
        BD_HIP(hipEventRecord(ex->t_ev[3 * c + 2], run));
        BD_HIP(hipEventRecord(ex->consumed[slot], run));
        ex->consumed_valid[slot] = true;
        ex->timed_chunks = c + 1;
    }
    return BD_OK;
}

struct bd_device_guard {
    int prev = 0; bool switched = false;
    int enter(int device)
    {
        BD_HIP(hipGetDevice(&prev));
        if (prev != device) { BD_HIP(hipSetDevice(device)); switched = true; }
        return BD_OK;
    }
    ~bd_device_guard() { if (switched) (void)hipSetDevice(prev); }
};

extern "C" int bd_chunked_analyze(bd_chunked *ex, int kind, uint32_t sampling_rate, const bd_params *params,
                                  uint32_t callback_len, bd_chunk_source_fn source, void *user, uint64_t n_streams,
                                  uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms, uint32_t max_beats,
                                  void *hip_stream)
{
    if (!ex || !source) { bd_set_error("bd_chunked_analyze: executor and source must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (!beat_count && n_streams) { bd_set_error("beat_count must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (callback_len == 0) { bd_set_error("callback_len must be >= 1 (the reference panics on an empty window)"); return BD_ERR_EMPTY_WINDOW; }
    bd_plan plan;
    int s = bd_make_plan(kind, sampling_rate, params, &plan); // argument errors before anything is enqueued
    if (s != BD_OK) return s;
    bd_device_guard guard;
    s = guard.enter(ex->device);
    if (s != BD_OK) return s;
    hipStream_t run = (hipStream_t)hip_stream;
    const uint64_t n_cb = ex->n_samples / callback_len;
    const uint64_t row = (n_cb + 127) & ~127ull;
    s = bd_chunked_ensure_mid(ex, ex->chunk_streams * (row ? row : 128), run);
    if (s != BD_OK) return s;
    uint8_t *flags = static_cast<uint8_t *>(ex->mid);
    return bd_chunked_pass(ex, source, user, n_streams, run,
                           [&](const int16_t *samples, uint64_t first, uint64_t rows, hipEvent_t mid_event) {
        int st = bd_batch_candidates(kind, sampling_rate, params, samples, rows, ex->n_samples, ex->stride, callback_len,
                                     nullptr, flags, row, (void *)run);
        if (st != BD_OK) return st;
        BD_HIP(hipEventRecord(mid_event, run));
        return bd_refractory(sampling_rate, params, callback_len, flags, row, rows, n_cb, beat_count + first,
                             beat_idx ? beat_idx + first * max_beats : nullptr,
                             beat_ms ? beat_ms + first * max_beats : nullptr, max_beats, (void *)run);
    });
}

extern "C" int bd_chunked_onset(bd_chunked *ex, uint32_t sampling_rate, const bd_onset_params *p,
                                bd_chunk_source_fn source, void *user, uint64_t n_streams,
                                uint32_t *onset_count, uint32_t *onset_frame, uint32_t *onset_ms, uint32_t max_onsets,
                                void *hip_stream)
{
    if (!ex || !source || !p) { bd_set_error("bd_chunked_onset: executor, source and parameters must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (!onset_count && n_streams) { bd_set_error("onset_count must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (p->hop == 0) { bd_set_error("hop must be positive"); return BD_ERR_INVALID_ARG; }
    bd_device_guard guard;
    int s = guard.enter(ex->device);
    if (s != BD_OK) return s;
    hipStream_t run = (hipStream_t)hip_stream;
    const uint64_t n_frames = bd_onset_frames(ex->n_samples, p->fft_len, p->hop);
    const uint64_t row = (n_frames + 31) & ~31ull; // rows of flux start on 128-byte lines
    s = bd_chunked_ensure_mid(ex, ex->chunk_streams * (row ? row : 32) * sizeof(float), run);
    if (s != BD_OK) return s;
    float *flux = static_cast<float *>(ex->mid);
    return bd_chunked_pass(ex, source, user, n_streams, run,
                           [&](const int16_t *samples, uint64_t first, uint64_t rows, hipEvent_t mid_event) {
        int st = bd_onset_stft(sampling_rate, p, samples, rows, ex->n_samples, ex->stride, flux, nullptr, row, (void *)run);
        if (st != BD_OK) return st;
        BD_HIP(hipEventRecord(mid_event, run));
        return bd_onset_pick(sampling_rate, p, flux, rows, n_frames, row, nullptr, onset_count + first,
                             onset_frame ? onset_frame + first * max_onsets : nullptr,
                             onset_ms ? onset_ms + first * max_onsets : nullptr, max_onsets, (void *)run);
    });
}

extern "C" int bd_chunked_last_timing(bd_chunked *ex, float *main_ms, float *post_ms, uint32_t capacity)
{
    if (!ex) { bd_set_error("bd_chunked_last_timing: executor must not be NULL"); return BD_ERR_INVALID_ARG; }
    const uint64_t n = ex->timed_chunks;
    for (uint64_t c = 0; c < n && c < capacity; c++) {
        BD_HIP(hipEventSynchronize(ex->t_ev[3 * c + 2]));
        float a = 0.0f, b = 0.0f;
        BD_HIP(hipEventElapsedTime(&a, ex->t_ev[3 * c], ex->t_ev[3 * c + 1]));
        BD_HIP(hipEventElapsedTime(&b, ex->t_ev[3 * c + 1], ex->t_ev[3 * c + 2]));
        if (main_ms) main_ms[c] = a;
        if (post_ms) post_ms[c] = b;
    }
    return (int)(n > 0x7fffffffull ? 0x7fffffff : n);
}
