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
#include <string.h>

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

struct bd_live_args {
    const int16_t  *in;     // [M][in_stride]
    uint64_t        in_stride;
    uint32_t        n;      // samples per stream in this callback
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

#define BD_LIVE_LPF_STAGE 2048

__global__ __launch_bounds__(256) void bd_live_tick_kernel(bd_live_args a)
{
    __shared__ float red[4];
    __shared__ int16_t stage[BD_LIVE_LPF_STAGE];
    __shared__ uint32_t sh_allowed;
    const uint32_t m = blockIdx.x, tid = threadIdx.x;
    const int16_t *in = a.in + (uint64_t)m * a.in_stride;
    int16_t *ring = a.ring + (uint64_t)m * BD_RING_LEN;
    bd_live_state st = a.state[m];
    const uint32_t n = a.n;
    const bd_plan &pl = a.plan;

    uint32_t new_head = st.head;
    if (pl.kind == BD_KIND_SPECTRUM) {
        // spectrum.rs:65-68: push every sample, overwriting the oldest
        const uint32_t skip = n > BD_RING_LEN ? n - BD_RING_LEN : 0;
        for (uint32_t i = skip + tid; i < n; i += 256)
            ring[(st.head + i) & (BD_RING_LEN - 1)] = in[i];
        new_head = (st.head + n) & (BD_RING_LEN - 1);
    }
    // spectrum.rs:71 / lpf.rs:54: update_time (mod.rs:76-84)
    const float ms_of_frame = pl.ms_per_sample * (float)n;
    const uint32_t beat_time = st.time_ms + bd_dev_f32_as_u32(ms_of_frame / 2.0f);
    const uint32_t time_ms = st.time_ms + bd_dev_f32_as_u32(ms_of_frame);
    // lib.rs:115-127
    bool allowed = true;
    if (beat_time > 0 && beat_time < st.last_beat + pl.gap_ms) allowed = false;

    // window_stats.rs:41-55 over the callback's own samples
    float amax = 0.f;
    for (uint32_t i = tid; i < n; i += 256) amax = fmaxf(amax, fabsf((float)in[i]));
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
        } else if (pl.alpha_valid) {
            // lpf.rs:66-75; the recurrence is serial, thread 0 walks it over LDS
            float y = 0.f, ymax = 0.f;
            for (uint32_t base = 0; base < n; base += BD_LIVE_LPF_STAGE) {
                const uint32_t cnt = (n - base < BD_LIVE_LPF_STAGE) ? n - base : BD_LIVE_LPF_STAGE;
                __syncthreads();
                for (uint32_t i = tid; i < cnt; i += 256) stage[i] = in[base + i];
                __syncthreads();
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
        a.in = d->d_in; a.in_stride = d->in_cap; a.n = (uint32_t)n;
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
