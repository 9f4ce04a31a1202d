// The only cross-callback dependency of the path: the 400 ms refractory rule
// (Strategy::last_beat_beyond_threshold, src/lib.rs:115-127) driven by the
// integer millisecond clock (AnalysisState::update_time,
// src/strategies/mod.rs:76-84).
//
// With equal callback lengths the clock is closed-form: every callback adds
// step = (ms_per_sample * len) as u32 to time_ms, and the beat time of callback
// j is j * step + half, half = (ms_per_sample * len / 2) as u32.  Both gates of
// the reference only suppress output, so the per-callback decisions can be
// computed for all callbacks first and this pass replays the rule over them:
// a decision at callback j becomes a beat unless beat_time > 0 and
// beat_time < last_beat + gap (last_beat starts at 0, so nothing fires before
// `gap` ms).  One wavefront per stream reads 64 decisions per step (coalesced),
// turns them into a 64-bit mask and walks the set bits.
#include "bd_internal.h"

__global__ __launch_bounds__(256) void bd_refractory_kernel(
    const uint8_t *flags, uint64_t flags_stride, uint64_t n_streams, uint64_t n_cb_max,
    const uint64_t *lengths, uint32_t callback_len, uint32_t step_ms, uint32_t half_ms, uint32_t gap_ms,
    uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms, uint32_t max_beats)
{
    const int lane = threadIdx.x & 63;
    const uint64_t s = bd_block_index() * 4 + (threadIdx.x >> 6);
    if (s >= n_streams) return; // wave-uniform
    const uint8_t *f = flags + s * flags_stride;
    // a stream shorter than the batch stops at its own last complete callback
    uint64_t n_cb = n_cb_max;
    if (lengths) {
        const uint64_t own = lengths[s] / callback_len;
        n_cb = own < n_cb_max ? own : n_cb_max;
    }
    uint32_t last_beat = 0, count = 0;
    for (uint64_t base = 0; base < n_cb; base += 64) {
        const uint64_t j = base + lane;
        const bool set = (j < n_cb) && (f[j] != 0);
        unsigned long long mask = __ballot(set);
        while (mask) { // wave-uniform
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint64_t jj = base + (uint64_t)b;
            const uint32_t beat_time = (uint32_t)jj * step_ms + half_ms; // wraps like u32 adds
            bool allowed = true;
            if (beat_time > 0) {                                // lib.rs:120
                const uint32_t threshold = last_beat + gap_ms;  // lib.rs:121
                if (beat_time < threshold) allowed = false;     // lib.rs:122-124
            }
            if (allowed) {
                if (lane == 0 && count < max_beats) {
                    if (beat_idx) beat_idx[s * max_beats + count] = (uint32_t)jj;
                    if (beat_ms) beat_ms[s * max_beats + count] = beat_time;
                }
                count++;
                last_beat = beat_time;                          // mod.rs:89-91
            }
        }
    }
    if (lane == 0) beat_count[s] = count;
}

int bd_launch_refractory(const bd_plan &plan, uint32_t callback_len, const uint8_t *flags,
                         uint64_t flags_stride, uint64_t n_streams, uint64_t n_cb, uint32_t *beat_count,
                         uint32_t *beat_idx, uint32_t *beat_ms, uint32_t max_beats,
                         hipStream_t st, const uint64_t *lengths)
{
    if (n_streams == 0) return BD_OK;
    uint32_t step_ms, half_ms;
    bd_clock_step(plan, callback_len, &step_ms, &half_ms);
    const uint64_t grid64 = (n_streams + 3) / 4;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("too many streams for one launch"); return BD_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(bd_refractory_kernel, grid_dims, dim3(256), 0, st,
                       flags, flags_stride, n_streams, n_cb, lengths, callback_len, step_ms, half_ms, plan.gap_ms,
                       beat_count, beat_idx, beat_ms, max_beats);
    return bd_check_hip(hipGetLastError(), "launch bd_refractory_kernel");
}
