// Per-callback ("live") path: the stateful detectors of the reference
// (SABeatDetector, src/strategies/spectrum.rs:37-99; LpfBeatDetector,
// src/strategies/lpf.rs:34-82) with their state kept on the GPU.  One kernel
// launch analyses one callback for each of M independent streams, one
// workgroup per stream: ring push, clock update, refractory gate, amplitude
// gate, strategy decision, exactly in the reference's order.  Any callback
// length >= 1, any sampling rate (any number of retained bins).
//
// bd_detector (the Strategy::is_beat stand-in) is this engine with M = 1 and
// host staging through pinned memory.
#include "bd_internal.h"

#include <new>
#include <string>
#include <vector>
#include <string.h>
#include <stdlib.h>

static __device__ const float bd_cos1024[1024] = {
#include "bd_cos1024.inc"
};
static __device__ const float bd_sin1024[1024] = {
#include "bd_sin1024.inc"
};

struct bd_live_state {      // AnalysisState, src/strategies/mod.rs:39-56
    uint32_t beat_time_ms;
    uint32_t time_ms;
    uint32_t last_beat;
    uint32_t head;          // ring slot of the oldest sample
};

struct bd_live_result {
    uint32_t beat;          // 1 = beat
    uint32_t relative_ms;   // BeatInfo::relative_ms
    float    value;         // analysed value (0 when the reference returns early)
    uint32_t candidate;     // decision before the refractory rule (0 when not evaluated)
};

// What changes from tick to tick, in pinned host memory that the kernels read through its
// device address: a captured graph never has to be rebuilt or updated for a new callback length.
struct bd_live_ctl {
    uint32_t n;             // samples per stream in this tick (equal callbacks)
    uint32_t ragged;        // 1: stream m brings len[m] samples, rows are `pitch` samples apart
    uint32_t pitch;
    uint32_t reserved;
    uint32_t len[1];        // [n_streams] (the block is allocated with room for them)
};
static size_t bd_live_ctl_bytes(size_t n_streams) { return sizeof(bd_live_ctl) + (n_streams ? n_streams - 1 : 0) * sizeof(uint32_t); }

struct bd_live_args {
    const int16_t  *in;     // [M][in_stride]
    uint64_t        in_stride;
    uint32_t        n;      // samples per stream in this callback (ctl == NULL)
    const bd_live_ctl *ctl; // if set: n = in_stride = ctl->n
    int16_t        *ring;   // [M][1024]
    bd_live_state  *state;  // [M]
    bd_live_result *result; // [M]
    bd_plan         plan;
};

__device__ __forceinline__ uint32_t bd_dev_f32_as_u32(float x)
{
    if (x != x || x <= 0.f) return 0;
    if (x >= 4294967296.0f) return 4294967295u;
    return (uint32_t)x;
}

__device__ float bd_block_sum(float v, float *red)
{
    v = bd_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ float bd_block_max(float v, float *red)
{
    v = bd_wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

#define BD_LIVE_STAGE 2048

// Copies samples [base, base + cnt) of a callback into LDS: 16-byte loads (the input may sit in
// pinned host memory, where every request crosses PCIe) at 2-byte-aligned addresses, the last
// cnt % 8 samples one by one.  Callers synchronise.
__device__ __forceinline__ void bd_live_stage_chunk(const int16_t *in, uint32_t base, uint32_t cnt, uint4 *stage)
{
    typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
    const uint32_t units = cnt / 8;
    for (uint32_t u = threadIdx.x; u < units; u += 256) {
        const u32x4_a2 v = *reinterpret_cast<const u32x4_a2 *>(in + base + 8 * u);
        stage[u] = make_uint4(v.x, v.y, v.z, v.w);
    }
    int16_t *s16 = reinterpret_cast<int16_t *>(stage);
    for (uint32_t i = 8 * units + threadIdx.x; i < cnt; i += 256) s16[i] = in[base + i];
}

__global__ __launch_bounds__(256) void bd_live_tick_kernel(bd_live_args a)
{
    __shared__ float red[4];
    __shared__ uint4 stage_u4[BD_LIVE_STAGE / 8];
    __shared__ uint32_t sh_allowed;
    const int16_t *stage = reinterpret_cast<const int16_t *>(stage_u4);
    const uint32_t m = blockIdx.x, tid = threadIdx.x;
    // every stream is its own device in the reference and its closure gets its own data.len()
    // (src/record.rs:93-95, :107-109, :121-123): a tick may bring a different length per stream
    const bool ragged = a.ctl && a.ctl->ragged;
    const uint32_t n = a.ctl ? (ragged ? a.ctl->len[m] : a.ctl->n) : a.n;
    const uint64_t in_stride = a.ctl ? (uint64_t)(ragged ? a.ctl->pitch : n) : a.in_stride;
    if (n == 0) {
        // the stream's device delivered nothing in this tick: no is_beat call, nothing advances
        if (tid == 0) {
            bd_live_result r;
            r.beat = 0u; r.relative_ms = 0u; r.value = 0.f; r.candidate = 0u;
            a.result[m] = r;
        }
        return;
    }
    const int16_t *in = a.in + (uint64_t)m * in_stride;
    int16_t *ring = a.ring + (uint64_t)m * BD_RING_LEN;
    bd_live_state st = a.state[m];
    const bd_plan &pl = a.plan;

    // one pass over the callback through LDS: window_stats.rs:41-55 over the callback's own
    // samples, and (Spectrum) spectrum.rs:65-68: push every sample, overwriting the oldest
    const uint32_t skip = n > BD_RING_LEN ? n - BD_RING_LEN : 0;
    float amax = 0.f;
    for (uint32_t base = 0; base < n; base += BD_LIVE_STAGE) {
        const uint32_t cnt = (n - base < BD_LIVE_STAGE) ? n - base : BD_LIVE_STAGE;
        if (base) __syncthreads();
        bd_live_stage_chunk(in, base, cnt, stage_u4);
        __syncthreads();
        for (uint32_t i = tid; i < cnt; i += 256) {
            const int16_t v = stage[i];
            amax = fmaxf(amax, fabsf((float)v));
            if (pl.kind == BD_KIND_SPECTRUM && base + i >= skip) ring[(st.head + base + i) & (BD_RING_LEN - 1)] = v;
        }
    }
    const uint32_t new_head = pl.kind == BD_KIND_SPECTRUM ? (st.head + n) & (BD_RING_LEN - 1) : st.head;
    // spectrum.rs:71 / lpf.rs:54: update_time (mod.rs:76-84)
    const float ms_of_frame = pl.ms_per_sample * (float)n;
    const uint32_t beat_time = st.time_ms + bd_dev_f32_as_u32(ms_of_frame / 2.0f);
    const uint32_t time_ms = st.time_ms + bd_dev_f32_as_u32(ms_of_frame);
    // lib.rs:115-127
    bool allowed = true;
    if (beat_time > 0 && beat_time < st.last_beat + pl.gap_ms) allowed = false;

    amax = fminf(bd_block_max(amax, red), 32767.f);
    const bool gate = amax >= (float)pl.min_amp;       // lib.rs:133-136

    float value = 0.f;
    bool candidate = false;
    if (allowed && gate) { // block-uniform
        if (pl.kind == BD_KIND_SPECTRUM) {
            __syncthreads(); // ring writes visible
            float x[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                x[q] = (float)ring[(new_head + tid + 256 * q) & (BD_RING_LEN - 1)];
            float best = 0.f;
            for (int k = pl.first_bin; k <= pl.last_bin; k++) {
                float re = 0.f, im = 0.f;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int tw = (k * (int)(tid + 256 * q)) & (BD_RING_LEN - 1);
                    re = __builtin_fmaf(x[q], bd_cos1024[tw], re);
                    im = __builtin_fmaf(x[q], bd_sin1024[tw], im);
                }
                re = bd_block_sum(re, red);
                im = bd_block_sum(im, red);
                if (k == 0 && pl.dc_packs_nyquist) {
                    float ny = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; q++) ny += (tid & 1) ? -x[q] : x[q];
                    im = bd_block_sum(ny, red);
                }
                const float mag = sqrtf(re * re + im * im);
                best = (k == pl.first_bin) ? mag : fmaxf(best, mag);
            }
            value = (pl.divisor == 0.f) ? best : best / pl.divisor;   // divide_by_N
            candidate = value > pl.threshold;                          // spectrum.rs:92
        } else {
            // lpf.rs:66-75; the recurrence is serial, thread 0 walks it over LDS.  A NaN alpha
            // (sampling rate 0 as u16) turns every filtered sample into 0 in the reference
            // (`NaN as i16`); here y becomes NaN and the maximum, which skips NaNs, stays 0.
            float y = 0.f, ymax = 0.f;
            for (uint32_t base = 0; base < n; base += BD_LIVE_STAGE) {
                const uint32_t cnt = (n - base < BD_LIVE_STAGE) ? n - base : BD_LIVE_STAGE;
                if (n > BD_LIVE_STAGE) { // longer than the staging buffer: it holds the last chunk, fetch again
                    __syncthreads();
                    bd_live_stage_chunk(in, base, cnt, stage_u4);
                    __syncthreads();
                }
                if (tid == 0) {
                    for (uint32_t i = 0; i < cnt; i++) {
                        const float xx = (float)stage[i];
                        const float d = xx - y;
                        const float mm = pl.alpha * d;
                        y = truncf(y + mm);
                        ymax = fmaxf(ymax, fabsf(y));
                    }
                }
            }
            if (tid == 0) {
                float thr = truncf(pl.lpf_ratio * amax);               // lpf.rs:74
                thr = (thr != thr) ? 0.f : fminf(fmaxf(thr, -32768.f), 32767.f);
                sh_allowed = (ymax >= thr) ? 1u : 0u;                  // lpf.rs:75
                red[0] = ymax;
            }
            __syncthreads();
            candidate = sh_allowed != 0;
            value = red[0];
        }
    }
    if (tid == 0) {
        bd_live_result r;
        r.beat = candidate ? 1u : 0u;
        r.relative_ms = candidate ? beat_time : 0u;
        r.value = value;
        r.candidate = candidate ? 1u : 0u;
        a.result[m] = r;
        st.beat_time_ms = beat_time;
        st.time_ms = time_ms;
        if (candidate) st.last_beat = beat_time;   // mod.rs:89-91
        st.head = new_head;
        a.state[m] = st;
    }
}

// ---- host side ------------------------------------------------------------------
struct bd_detector {
    bd_plan         plan;
    int             device;
    hipStream_t     stream;
    int16_t        *d_in;      size_t in_cap;
    int16_t        *h_in;      // pinned staging, same capacity
    int16_t        *d_ring;
    bd_live_state  *d_state;
    bd_live_result *d_result;
    bd_live_result *h_result;  // pinned
    bd_live_result  last;
};

static int bd_detector_reserve(bd_detector *d, size_t n)
{
    if (n <= d->in_cap) return BD_OK;
    size_t cap = d->in_cap ? d->in_cap : 4096;
    while (cap < n) cap *= 2;
    if (d->d_in) { BD_HIP(hipFree(d->d_in)); d->d_in = nullptr; }
    if (d->h_in) { BD_HIP(hipHostFree(d->h_in)); d->h_in = nullptr; }
    d->in_cap = 0;
    BD_HIP(hipMalloc((void **)&d->d_in, cap * sizeof(int16_t)));
    BD_HIP(hipHostMalloc((void **)&d->h_in, cap * sizeof(int16_t), hipHostMallocDefault));
    d->in_cap = cap;
    return BD_OK;
}

extern "C" void bd_detector_destroy(bd_detector *d)
{
    if (!d) return;
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && prev != d->device) (void)hipSetDevice(d->device);
    if (d->stream) { (void)hipStreamSynchronize(d->stream); (void)hipStreamDestroy(d->stream); }
    if (d->d_in) (void)hipFree(d->d_in);
    if (d->h_in) (void)hipHostFree(d->h_in);
    if (d->d_ring) (void)hipFree(d->d_ring);
    if (d->d_state) (void)hipFree(d->d_state);
    if (d->d_result) (void)hipFree(d->d_result);
    if (d->h_result) (void)hipHostFree(d->h_result);
    if (prev >= 0 && prev != d->device) (void)hipSetDevice(prev);
    delete d;
}

static int bd_detector_init(bd_detector *d)
{
    BD_HIP(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    BD_HIP(hipMalloc((void **)&d->d_ring, BD_RING_LEN * sizeof(int16_t)));
    BD_HIP(hipMalloc((void **)&d->d_state, sizeof(bd_live_state)));
    BD_HIP(hipMalloc((void **)&d->d_result, sizeof(bd_live_result)));
    BD_HIP(hipHostMalloc((void **)&d->h_result, sizeof(bd_live_result), hipHostMallocDefault));
    // spectrum.rs:50-51 ring of zeros; mod.rs:65-67 clock at zero
    BD_HIP(hipMemsetAsync(d->d_ring, 0, BD_RING_LEN * sizeof(int16_t), d->stream));
    BD_HIP(hipMemsetAsync(d->d_state, 0, sizeof(bd_live_state), d->stream));
    BD_HIP(hipStreamSynchronize(d->stream));
    return bd_detector_reserve(d, 4096);
}

extern "C" int bd_detector_create(int kind, uint32_t sampling_rate, const bd_params *params,
                                  int device, bd_detector **out)
{
    if (!out) { bd_set_error("bd_detector_create: out is NULL"); return BD_ERR_INVALID_ARG; }
    *out = nullptr;
    bd_plan plan;
    int s = bd_make_plan(kind, sampling_rate, params, &plan);
    if (s != BD_OK) return s;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        bd_set_error("no HIP device available (this library has no CPU path)");
        return BD_ERR_NO_DEVICE;
    }
    int prev = 0;
    BD_HIP(hipGetDevice(&prev));
    if (device < 0) device = prev;
    if (device >= count) { bd_set_error("device %d out of range (%d devices)", device, count); return BD_ERR_INVALID_ARG; }
    bd_detector *d = new (std::nothrow) bd_detector();
    if (!d) { bd_set_error("out of host memory"); return BD_ERR_HIP; }
    d->plan = plan;
    d->device = device;
    if (device != prev) BD_HIP(hipSetDevice(device));
    s = bd_detector_init(d);
    if (device != prev) (void)hipSetDevice(prev);
    if (s != BD_OK) { bd_detector_destroy(d); return s; }
    *out = d;
    return BD_OK;
}

extern "C" int bd_detector_is_beat(bd_detector *d, const int16_t *samples, size_t n,
                                   uint32_t *relative_ms)
{
    if (!d || (!samples && n)) { bd_set_error("bd_detector_is_beat: null argument"); return BD_ERR_INVALID_ARG; }
    if (n == 0) { bd_set_error("empty window (the reference panics at window_stats.rs:51)"); return BD_ERR_EMPTY_WINDOW; }
    if (n > 0xffffffffull) { bd_set_error("callback too long"); return BD_ERR_UNSUPPORTED; }
    int prev = 0;
    BD_HIP(hipGetDevice(&prev));
    if (prev != d->device) BD_HIP(hipSetDevice(d->device));
    int s = bd_detector_reserve(d, n);
    if (s == BD_OK) {
        memcpy(d->h_in, samples, n * sizeof(int16_t));
        bd_live_args a;
        a.in = d->d_in; a.in_stride = d->in_cap; a.n = (uint32_t)n; a.ctl = nullptr;
        a.ring = d->d_ring; a.state = d->d_state; a.result = d->d_result; a.plan = d->plan;
        s = bd_check_hip(hipMemcpyAsync(d->d_in, d->h_in, n * sizeof(int16_t), hipMemcpyHostToDevice, d->stream), "copy callback to device");
        if (s == BD_OK) {
            hipLaunchKernelGGL(bd_live_tick_kernel, dim3(1), dim3(256), 0, d->stream, a);
            s = bd_check_hip(hipGetLastError(), "launch bd_live_tick_kernel");
        }
        if (s == BD_OK) s = bd_check_hip(hipMemcpyAsync(d->h_result, d->d_result, sizeof(bd_live_result), hipMemcpyDeviceToHost, d->stream), "copy result to host");
        if (s == BD_OK) s = bd_check_hip(hipStreamSynchronize(d->stream), "synchronise detector stream");
    }
    if (prev != d->device) (void)hipSetDevice(prev);
    if (s != BD_OK) return s;
    d->last = *d->h_result;
    if (d->last.beat) {
        if (relative_ms) *relative_ms = d->last.relative_ms;
        return 1;
    }
    return 0;
}

extern "C" int bd_detector_kind(const bd_detector *d) { return d ? d->plan.kind : BD_ERR_INVALID_ARG; }

extern "C" int bd_detector_last(const bd_detector *d, float *value, int *candidate)
{
    if (!d) return BD_ERR_INVALID_ARG;
    if (value) *value = d->last.value;
    if (candidate) *candidate = (int)d->last.candidate;
    return BD_OK;
}

// =============================================================================
// Multi-stream live engine: the successor of record::start_listening's callback
// loop (src/record.rs:87-146) for M concurrent streams.  Every tick carries one
// callback of n samples for each stream.  Ticks go through a ring of K slots
// (pinned host staging, pinned control block, pinned result buffer); a tick is
// one replay of a HIP graph captured when the engine was created: optional
// ingest conversion (record.rs:151-166) and the live kernel, both working on the
// pinned buffers in place.  All ticks run on one stream, in order, so detector state
// advances exactly as in the reference.  If an on_beat callback is installed a
// dispatcher thread calls it in tick order, stream order, off the submitting
// thread (the reference runs on_beat_cb on the audio thread: record.rs:96).
// =============================================================================
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

// Ingest conversion of one tick (record.rs:151-166); the grid is sized for the longest callback,
// the tick's own length comes from the control block.
__global__ void bd_live_convert_kernel(const void *in, int16_t *out, const bd_live_ctl *ctl, uint32_t n_streams, int format)
{
    // rows of a ragged tick are converted whole (what lies past a stream's own samples is not looked at)
    const uint64_t n = (uint64_t)n_streams * (ctl->ragged ? ctl->pitch : ctl->n);
    const uint64_t i = bd_block_index() * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (format == BD_FMT_U16) {
        const int32_t x = (int32_t)((const uint16_t *)in)[i] - (int32_t)(INT16_MAX / 2); // record.rs:154-155
        out[i] = (int16_t)(uint16_t)(uint32_t)x;                                       // :156
    } else {
        float t = truncf(((const float *)in)[i] * 32767.0f);                           // record.rs:163-164
        t = (t != t) ? 0.f : fminf(fmaxf(t, -32768.f), 32767.f);
        out[i] = (int16_t)(int)t;
    }
}

struct bd_live_slot {
    void           *h_in = nullptr;     // pinned, M * max_len * 4 bytes; the kernels read it in place
    bd_live_ctl    *h_ctl = nullptr;    // pinned: this tick's callback length
    int16_t        *d_i16 = nullptr;    // converted samples (unused for i16 input)
    bd_live_result *h_res = nullptr;    // pinned [M]; the kernel writes it in place
    hipEvent_t      done = nullptr;
    uint64_t        tick = 0;           // tick currently in the slot (valid if busy)
    bool            busy = false;
    hipGraphExec_t  graph[3] = {nullptr, nullptr, nullptr}; // one per sample format, any callback length
};

struct bd_live {
    bd_plan        plan;
    int            device = 0;
    uint32_t       n_streams = 0, max_len = 0, depth = 0;
    hipStream_t    stream = nullptr;
    hipStream_t    cap_stream = nullptr; // graph capture only
    int16_t       *d_ring = nullptr;
    bd_live_state *d_state = nullptr;
    std::vector<bd_live_slot> slots;
    uint64_t       next_tick = 0;       // next tick id to hand out
    uint64_t       next_free = 0;       // oldest tick not yet released
    std::mutex     mu;
    std::condition_variable cv;
    // dispatcher
    bd_live_on_beat_fn on_beat = nullptr;
    void          *on_beat_user = nullptr;
    std::thread    dispatcher;
    std::atomic<bool> stop{false};
    std::atomic<int>  dispatch_error{BD_OK};
};

static void bd_live_release_locked(bd_live *L, bd_live_slot &s)
{
    s.busy = false;
    L->next_free = s.tick + 1;
    L->cv.notify_all();
}

static void bd_live_dispatch_loop(bd_live *L)
{
    (void)hipSetDevice(L->device);
    uint64_t tick = 0;
    for (;;) {
        bd_live_slot *s = nullptr;
        {
            std::unique_lock<std::mutex> lk(L->mu);
            L->cv.wait(lk, [&] { return L->stop.load() || L->next_tick > tick; });
            if (L->next_tick <= tick) return; // stopping and nothing left
            s = &L->slots[tick % L->depth];
        }
        if (hipEventSynchronize(s->done) != hipSuccess) {
            // wake everyone who waits for a slot or a tick: they return the error
            std::lock_guard<std::mutex> lk(L->mu);
            L->dispatch_error.store(BD_ERR_HIP);
            L->cv.notify_all();
            return;
        }
        for (uint32_t m = 0; m < L->n_streams; m++)
            if (s->h_res[m].beat) L->on_beat(L->on_beat_user, m, s->h_res[m].relative_ms);
        {
            std::lock_guard<std::mutex> lk(L->mu);
            bd_live_release_locked(L, *s);
        }
        tick++;
    }
}

extern "C" void bd_live_destroy(bd_live *L)
{
    if (!L) return;
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && prev != L->device) (void)hipSetDevice(L->device);
    if (L->dispatcher.joinable()) {
        { std::lock_guard<std::mutex> lk(L->mu); L->stop = true; }
        L->cv.notify_all();
        L->dispatcher.join();
    }
    if (L->stream) (void)hipStreamSynchronize(L->stream);
    for (auto &s : L->slots) {
        for (auto &g : s.graph) if (g) (void)hipGraphExecDestroy(g);
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.h_ctl) (void)hipHostFree(s.h_ctl);
        if (s.d_i16) (void)hipFree(s.d_i16);
        if (s.h_res) (void)hipHostFree(s.h_res);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (L->d_ring) (void)hipFree(L->d_ring);
    if (L->d_state) (void)hipFree(L->d_state);
    if (L->stream) (void)hipStreamDestroy(L->stream);
    if (L->cap_stream) (void)hipStreamDestroy(L->cap_stream);
    if (prev >= 0 && prev != L->device) (void)hipSetDevice(prev);
    delete L;
}

// Captures the graph of one tick for this slot and sample format: [ingest conversion ->] tick
// kernel.  The kernels read the pinned staging buffer and write the pinned result buffer through
// their device addresses (no copy nodes) and take the callback length from the slot's control
// block, so one graph serves every callback length and no tick ever pays for a capture.
// Captured on a stream of its own: a thread may be waiting on an event of the working stream.
static int bd_live_build_graph(bd_live *L, bd_live_slot &s, int format, hipGraphExec_t *out)
{
    const size_t M = L->n_streams;
    hipGraph_t graph = nullptr;
    void *dp_in = nullptr, *dp_res = nullptr, *dp_ctl = nullptr;
    BD_HIP(hipHostGetDevicePointer(&dp_in, s.h_in, 0));
    BD_HIP(hipHostGetDevicePointer(&dp_res, s.h_res, 0));
    BD_HIP(hipHostGetDevicePointer(&dp_ctl, s.h_ctl, 0));
    BD_HIP(hipStreamBeginCapture(L->cap_stream, hipStreamCaptureModeRelaxed));
    const int16_t *in16 = (const int16_t *)dp_in;
    if (format != BD_FMT_I16) {
        const uint64_t most = (uint64_t)M * L->max_len;
        dim3 grid;
        (void)bd_grid_dims((most + 255) / 256, &grid);
        hipLaunchKernelGGL(bd_live_convert_kernel, grid, dim3(256), 0, L->cap_stream,
                           (const void *)dp_in, s.d_i16, (const bd_live_ctl *)dp_ctl, (uint32_t)M, format);
        in16 = s.d_i16;
    }
    bd_live_args a;
    a.in = in16; a.in_stride = 0; a.n = 0; a.ctl = (const bd_live_ctl *)dp_ctl;
    a.ring = L->d_ring; a.state = L->d_state; a.result = (bd_live_result *)dp_res; a.plan = L->plan;
    hipLaunchKernelGGL(bd_live_tick_kernel, dim3((unsigned)M), dim3(256), 0, L->cap_stream, a);
    const hipError_t e = hipGetLastError();
    const hipError_t e2 = hipStreamEndCapture(L->cap_stream, &graph);
    if (e != hipSuccess || e2 != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return bd_check_hip(e != hipSuccess ? e : e2, "capture live tick graph");
    }
    const hipError_t e3 = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    return bd_check_hip(e3, "instantiate live tick graph");
}

static int bd_live_init(bd_live *L)
{
    const size_t M = L->n_streams, cap = (size_t)L->max_len;
    if ((uint64_t)M * cap > 0x7fffffffull) { bd_set_error("n_streams x max_callback_len too large"); return BD_ERR_UNSUPPORTED; }
    BD_HIP(hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking));
    BD_HIP(hipStreamCreateWithFlags(&L->cap_stream, hipStreamNonBlocking));
    BD_HIP(hipMalloc((void **)&L->d_ring, M * BD_RING_LEN * sizeof(int16_t)));
    BD_HIP(hipMalloc((void **)&L->d_state, M * sizeof(bd_live_state)));
    BD_HIP(hipMemsetAsync(L->d_ring, 0, M * BD_RING_LEN * sizeof(int16_t), L->stream)); // spectrum.rs:50-51
    BD_HIP(hipMemsetAsync(L->d_state, 0, M * sizeof(bd_live_state), L->stream));        // mod.rs:65-67
    L->slots.resize(L->depth);
    for (auto &s : L->slots) {
        BD_HIP(hipHostMalloc(&s.h_in, M * cap * 4, hipHostMallocDefault));
        BD_HIP(hipHostMalloc((void **)&s.h_ctl, bd_live_ctl_bytes(M), hipHostMallocDefault));
        BD_HIP(hipMalloc((void **)&s.d_i16, M * cap * sizeof(int16_t)));
        BD_HIP(hipHostMalloc((void **)&s.h_res, M * sizeof(bd_live_result), hipHostMallocDefault));
        BD_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        memset(s.h_ctl, 0, bd_live_ctl_bytes(M));
    }
    BD_HIP(hipStreamSynchronize(L->stream));
    // every graph a tick can need exists before the first tick
    for (auto &s : L->slots)
        for (int format = 0; format < 3; format++) {
            const int st = bd_live_build_graph(L, s, format, &s.graph[format]);
            if (st != BD_OK) return st;
        }
    return BD_OK;
}

extern "C" int bd_live_create(int kind, uint32_t sampling_rate, const bd_params *params,
                              uint32_t n_streams, uint32_t max_callback_len, uint32_t depth,
                              int device, bd_live **out)
{
    if (!out) { bd_set_error("bd_live_create: out is NULL"); return BD_ERR_INVALID_ARG; }
    *out = nullptr;
    if (n_streams == 0 || max_callback_len == 0) { bd_set_error("bd_live_create: n_streams and max_callback_len must be positive"); return BD_ERR_INVALID_ARG; }
    bd_plan plan;
    int s = bd_make_plan(kind, sampling_rate, params, &plan);
    if (s != BD_OK) return s;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        bd_set_error("no HIP device available (this library has no CPU path)");
        return BD_ERR_NO_DEVICE;
    }
    int prev = 0;
    BD_HIP(hipGetDevice(&prev));
    if (device < 0) device = prev;
    if (device >= count) { bd_set_error("device %d out of range (%d devices)", device, count); return BD_ERR_INVALID_ARG; }
    bd_live *L = new (std::nothrow) bd_live();
    if (!L) { bd_set_error("out of host memory"); return BD_ERR_HIP; }
    L->plan = plan; L->device = device; L->n_streams = n_streams; L->max_len = max_callback_len;
    L->depth = depth ? (depth > 64 ? 64 : depth) : 4;
    if (device != prev) BD_HIP(hipSetDevice(device));
    s = bd_live_init(L);
    if (device != prev) (void)hipSetDevice(prev);
    if (s != BD_OK) { bd_live_destroy(L); return s; }
    *out = L;
    return BD_OK;
}

extern "C" int bd_live_set_on_beat(bd_live *L, bd_live_on_beat_fn cb, void *user)
{
    if (!L || !cb) { bd_set_error("bd_live_set_on_beat: null argument"); return BD_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(L->mu);
    if (L->dispatcher.joinable() || L->next_tick != 0) {
        bd_set_error("bd_live_set_on_beat: must be called once, before the first submit");
        return BD_ERR_INVALID_ARG;
    }
    L->on_beat = cb; L->on_beat_user = user;
    L->dispatcher = std::thread(bd_live_dispatch_loop, L);
    return BD_OK;
}

// One tick: equal callbacks (lengths == NULL, `n` samples per stream, rows n apart) or one length per
// stream (rows `n` = pitch samples apart).
static int bd_live_submit_impl(bd_live *L, const void *samples, uint32_t n, const uint32_t *lengths, int format,
                               uint64_t *tick_id)
{
    uint64_t tick;
    bd_live_slot *s;
    {
        std::unique_lock<std::mutex> lk(L->mu);
        tick = L->next_tick;
        s = &L->slots[tick % L->depth];
        if (s->busy && !L->on_beat) {
            bd_set_error("live ring full: poll tick %llu first", (unsigned long long)s->tick);
            return BD_ERR_UNSUPPORTED;
        }
        // the dispatcher frees slots; if it has failed nobody will
        L->cv.wait(lk, [&] { return !s->busy || L->dispatch_error.load() != BD_OK; });
        if (L->dispatch_error.load() != BD_OK) { bd_set_error("live dispatcher failed"); return L->dispatch_error.load(); }
    }
    int prev = 0;
    BD_HIP(hipGetDevice(&prev));
    if (prev != L->device) BD_HIP(hipSetDevice(L->device));
    const size_t elem = format == BD_FMT_F32 ? 4 : 2;
    memcpy(s->h_in, samples, (size_t)L->n_streams * n * elem);
    if (lengths) {
        s->h_ctl->n = 0; s->h_ctl->ragged = 1; s->h_ctl->pitch = n;
        memcpy(s->h_ctl->len, lengths, (size_t)L->n_streams * sizeof(uint32_t));
    } else {
        s->h_ctl->n = n; s->h_ctl->ragged = 0; s->h_ctl->pitch = n;
    }
    int st = bd_check_hip(hipGraphLaunch(s->graph[format], L->stream), "launch live tick graph");
    if (st == BD_OK) st = bd_check_hip(hipEventRecord(s->done, L->stream), "record tick event");
    if (prev != L->device) (void)hipSetDevice(prev);
    if (st != BD_OK) return st;
    {
        std::lock_guard<std::mutex> lk(L->mu);
        s->busy = true; s->tick = tick;
        L->next_tick = tick + 1;
    }
    L->cv.notify_all();
    if (tick_id) *tick_id = tick;
    return BD_OK;
}

extern "C" int bd_live_submit(bd_live *L, const void *samples, uint32_t n, int format, uint64_t *tick_id)
{
    if (!L || !samples) { bd_set_error("bd_live_submit: null argument"); return BD_ERR_INVALID_ARG; }
    if (n == 0) { bd_set_error("empty window (the reference panics at window_stats.rs:51)"); return BD_ERR_EMPTY_WINDOW; }
    if (n > L->max_len) { bd_set_error("callback of %u samples exceeds max_callback_len %u", n, L->max_len); return BD_ERR_INVALID_ARG; }
    if (format != BD_FMT_I16 && format != BD_FMT_U16 && format != BD_FMT_F32) { bd_set_error("unknown sample format %d", format); return BD_ERR_INVALID_ARG; }
    return bd_live_submit_impl(L, samples, n, nullptr, format, tick_id);
}

extern "C" int bd_live_submit_ragged(bd_live *L, const void *samples, const uint32_t *n_per_stream, uint32_t row_pitch,
                                     int format, uint64_t *tick_id)
{
    if (!L || !samples || !n_per_stream) { bd_set_error("bd_live_submit_ragged: null argument"); return BD_ERR_INVALID_ARG; }
    if (row_pitch == 0) { bd_set_error("bd_live_submit_ragged: row_pitch must be positive"); return BD_ERR_INVALID_ARG; }
    if (row_pitch > L->max_len) { bd_set_error("row pitch of %u samples exceeds max_callback_len %u", row_pitch, L->max_len); return BD_ERR_INVALID_ARG; }
    if (format != BD_FMT_I16 && format != BD_FMT_U16 && format != BD_FMT_F32) { bd_set_error("unknown sample format %d", format); return BD_ERR_INVALID_ARG; }
    for (uint32_t m = 0; m < L->n_streams; m++)
        if (n_per_stream[m] > row_pitch) {
            bd_set_error("stream %u: callback of %u samples exceeds the row pitch %u", m, n_per_stream[m], row_pitch);
            return BD_ERR_INVALID_ARG;
        }
    return bd_live_submit_impl(L, samples, row_pitch, n_per_stream, format, tick_id);
}

extern "C" int bd_live_poll(bd_live *L, uint64_t tick_id, int wait, uint8_t *beats, uint32_t *relative_ms,
                            float *values)
{
    if (!L) { bd_set_error("bd_live_poll: null handle"); return BD_ERR_INVALID_ARG; }
    bd_live_slot *s;
    {
        std::lock_guard<std::mutex> lk(L->mu);
        if (tick_id >= L->next_tick) { bd_set_error("tick %llu has not been submitted", (unsigned long long)tick_id); return BD_ERR_INVALID_ARG; }
        s = &L->slots[tick_id % L->depth];
        if (!L->on_beat && (!s->busy || s->tick != tick_id)) { bd_set_error("tick %llu is no longer in the ring", (unsigned long long)tick_id); return BD_ERR_INVALID_ARG; }
    }
    if (L->on_beat) {
        // results are delivered through the callback; poll only waits for the tick
        std::unique_lock<std::mutex> lk(L->mu);
        if (!wait) return L->next_free > tick_id ? 1 : 0;
        L->cv.wait(lk, [&] { return L->next_free > tick_id || L->dispatch_error.load() != BD_OK; });
        return L->dispatch_error.load() != BD_OK ? L->dispatch_error.load() : 1;
    }
    if (wait) BD_HIP(hipEventSynchronize(s->done));
    else {
        hipError_t e = hipEventQuery(s->done);
        if (e == hipErrorNotReady) return 0;
        BD_HIP(e);
    }
    for (uint32_t m = 0; m < L->n_streams; m++) {
        if (beats) beats[m] = (uint8_t)s->h_res[m].beat;
        if (relative_ms) relative_ms[m] = s->h_res[m].relative_ms;
        if (values) values[m] = s->h_res[m].value;
    }
    {
        std::lock_guard<std::mutex> lk(L->mu);
        bd_live_release_locked(L, *s);
    }
    return 1;
}

// =============================================================================
// record::start_listening (src/record.rs:39-149) with the cpal device replaced
// by a caller-supplied source of callbacks.
// =============================================================================
struct bd_listener {
    std::thread thread;
    int         status = BD_OK;
    std::string message;
};

extern "C" int bd_start_listening(bd_on_beat_fn on_beat, void *on_beat_user, const bd_input_source *source,
                                  int kind, const volatile int *keep_recording, bd_listener **out)
{
    if (!out || !on_beat || !keep_recording) { bd_set_error("bd_start_listening: null argument"); return BD_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!__atomic_load_n(keep_recording, __ATOMIC_SEQ_CST)) {               // record.rs:45-47
        bd_set_error("Variable keep_recording is false from the beginning!?");
        return BD_ERR_STOPPED;
    }
    if (!source || !source->read) {                                         // record.rs:51-55
        bd_set_error("Must have input device!");
        return BD_ERR_NO_DEVICE;
    }
    if (source->sample_format != BD_FMT_I16 && source->sample_format != BD_FMT_U16 && source->sample_format != BD_FMT_F32) {
        bd_set_error("unknown sample format %d", source->sample_format);
        return BD_ERR_INVALID_ARG;
    }
    bd_live *L = nullptr;
    const uint32_t cap = source->max_callback_len ? source->max_callback_len : 4096;
    int s = bd_live_create(kind, source->sampling_rate, nullptr, 1, cap, 4, -1, &L); // record.rs:85 detector(sampling_rate)
    if (s != BD_OK) return s;
    bd_listener *h = new (std::nothrow) bd_listener();
    if (!h) { bd_live_destroy(L); bd_set_error("out of host memory"); return BD_ERR_HIP; }
    const bd_input_source src = *source;
    h->thread = std::thread([=] {                                           // record.rs:87 spawn
        std::vector<unsigned char> buf((size_t)cap * 4);
        while (__atomic_load_n(keep_recording, __ATOMIC_SEQ_CST)) {         // record.rs:141-145
            const size_t n = src.read(src.user, buf.data(), cap);
            if (n == 0) break;                                              // source exhausted
            uint64_t tick = 0;
            int rc = bd_live_submit(L, buf.data(), (uint32_t)n, src.sample_format, &tick);
            uint8_t beat = 0; uint32_t ms = 0;
            if (rc == BD_OK) rc = bd_live_poll(L, tick, 1, &beat, &ms, nullptr);
            if (rc < 0) { h->status = rc; h->message = bd_last_error(); break; }
            if (beat) on_beat(on_beat_user, ms);                            // record.rs:95-97, on this thread
        }
        bd_live_destroy(L);
    });
    *out = h;
    return BD_OK;
}

extern "C" int bd_listener_join(bd_listener *h)
{
    if (!h) return BD_ERR_INVALID_ARG;
    if (h->thread.joinable()) h->thread.join();                             // JoinHandle::join
    const int s = h->status;
    if (s != BD_OK) bd_set_error("%s", h->message.c_str());
    delete h;
    return s;
}
