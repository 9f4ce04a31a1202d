// bd_analyze_batch for callers that must not allocate (code under HIP graph
// capture, tight serving loops, harnesses that forbid hipMalloc).
//
// bd_analyze_batch needs one piece of device memory beyond its inputs and its
// beat lists: the per-callback decisions that the first kernel writes and the
// refractory pass reads.  There the caller brings it in (`flags`); here it is a
// file-scope device array of a fixed size, so the entry point below calls no
// hipMalloc / hipFree and no allocating library routine.
//
// Each GPU holds its own copy of the array at its own address.  The array's
// name is only a host-side handle: the device address is looked up once per
// device with hipGetSymbolAddress (with that device current) and kept in
// bd_scratch_ptr[]; kernels, and anything else that needs the memory, get that
// pointer.
//
// The array keeps whatever the previous call left in it.  Nothing has to start
// at zero: the first kernel writes the decision of every complete callback of
// every stream, and the refractory pass reads exactly those entries, later on
// the same HIP stream.  Two calls that overlap in time on one GPU (different
// streams or threads) would share the array and race: callers serialise them.
#include "bd_internal.h"

#include <atomic>
#include <mutex>

#define BD_NOALLOC_MAX_DEVICES 64

static __device__ __align__(256) uint8_t bd_scratch[BD_NOALLOC_SCRATCH_BYTES];

static std::mutex bd_scratch_mu;
static std::atomic<uint8_t *> bd_scratch_ptr[BD_NOALLOC_MAX_DEVICES];

// Device address of the current device's copy of bd_scratch (looked up on the first call per device).
static int bd_scratch_for_current_device(uint8_t **out)
{
    int dev = -1;
    BD_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= BD_NOALLOC_MAX_DEVICES) {
        bd_set_error("device index %d outside 0..%d", dev, BD_NOALLOC_MAX_DEVICES - 1);
        return BD_ERR_UNSUPPORTED;
    }
    uint8_t *p = bd_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        std::lock_guard<std::mutex> lock(bd_scratch_mu);
        p = bd_scratch_ptr[dev].load(std::memory_order_relaxed);
        if (!p) {
            void *addr = nullptr;
            BD_HIP(hipGetSymbolAddress(&addr, HIP_SYMBOL(bd_scratch)));
            if (!addr) { bd_set_error("hipGetSymbolAddress returned a null address for device %d", dev); return BD_ERR_HIP; }
            p = static_cast<uint8_t *>(addr);
            bd_scratch_ptr[dev].store(p, std::memory_order_release);
        }
    }
    *out = p;
    return BD_OK;
}

extern "C" int bd_analyze_batch_noalloc(int kind, uint32_t sampling_rate, const bd_params *params,
                                        const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                                        uint64_t stream_stride, uint32_t callback_len,
                                        uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms,
                                        uint32_t max_beats, void *hip_stream)
{
    bd_plan plan;
    int s = bd_make_plan(kind, sampling_rate, params, &plan);
    if (s != BD_OK) return s;
    if (callback_len == 0) { bd_set_error("callback_len must be >= 1 (the reference panics on an empty window)"); return BD_ERR_EMPTY_WINDOW; }
    if (!beat_count) { bd_set_error("beat_count must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (stream_stride < n_samples) {
        bd_set_error("stream_stride (%llu) < n_samples (%llu)", (unsigned long long)stream_stride, (unsigned long long)n_samples);
        return BD_ERR_INVALID_ARG;
    }
    const uint64_t n_cb = n_samples / callback_len;
    if (!samples && n_cb > 0 && n_streams > 0) { bd_set_error("samples must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (n_streams == 0) return BD_OK;
    // rows of decisions start on 128-byte lines while that fits, densely packed otherwise
    uint64_t row = (n_cb + 127) & ~127ull;
    if (row > BD_NOALLOC_SCRATCH_BYTES / n_streams) row = n_cb;
    if (n_cb > BD_NOALLOC_SCRATCH_BYTES / n_streams) {
        bd_set_error("bd_analyze_batch_noalloc: %llu streams x %llu callbacks exceed the fixed capacity of %llu "
                     "callbacks per call; split the batch or use bd_analyze_batch with a caller-owned flags buffer",
                     (unsigned long long)n_streams, (unsigned long long)n_cb, (unsigned long long)BD_NOALLOC_SCRATCH_BYTES);
        return BD_ERR_UNSUPPORTED;
    }
    uint8_t *flags = nullptr;
    s = bd_scratch_for_current_device(&flags);
    if (s != BD_OK) return s;
    hipStream_t st = (hipStream_t)hip_stream;
    if (plan.kind == BD_KIND_SPECTRUM)
        s = bd_launch_spectrum(plan, samples, n_streams, n_samples, stream_stride, callback_len, nullptr, flags, row, st);
    else
        s = bd_launch_lpf(plan, samples, n_streams, n_samples, stream_stride, callback_len, nullptr, flags, row, st);
    if (s != BD_OK) return s;
    return bd_launch_refractory(plan, callback_len, flags, row, n_streams, n_cb, beat_count, beat_idx, beat_ms,
                                max_beats, st);
}
