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
// `gap` ms).
//
// Two kernels.  bd_refractory_kernel: one wavefront per stream reads 64 decisions
// per step (1,024 on long rows), turns them into bit masks and walks the set bits.
// bd_refractory_lanes_kernel: a wavefront packs the decisions of 8 streams into
// bits in LDS and then walks them with one lane per stream.  The launcher picks
// by row alignment and length.
#include "bd_internal.h"

// Four bits, one per byte of x: set where the byte is not zero.  The high bit of each byte of t says so; the
// multiplication gathers bits 0, 8, 16, 24 of t >> 7 into bits 21 .. 24 (the partial products land on distinct bits).
__device__ __forceinline__ uint32_t bd_nonzero_bytes(uint32_t x)
{
    const uint32_t t = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
    return (((t >> 7) * 0x00204081u) >> 21) & 0xfu;
}

// IDX is the type callback numbers are counted in: uint32_t when the batch has fewer than 2^31 callbacks per
// stream and the launcher found the shortcut below valid (skip > 0), uint64_t with skip == 0 otherwise.
//
// The shortcut: with a clock that advances (step > 0) and cannot wrap within the batch, a beat at callback j
// refuses exactly the decisions of callbacks j + n with n * step < gap, that is n < skip = ceil(gap / step).
// Decisions before j_min = j + skip are dropped unseen; a stream with a decision in every callback then costs
// one walk per beat, not one per decision.
template <typename IDX>
__global__ __launch_bounds__(256) void bd_refractory_kernel(
    const uint8_t *flags, uint64_t flags_stride, uint64_t n_streams, uint64_t n_cb_max,
    const uint64_t *lengths, uint32_t callback_len, uint32_t step_ms, uint32_t half_ms, uint32_t gap_ms, uint32_t skip,
    uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms, uint32_t max_beats)
{
    constexpr bool SKIP = sizeof(IDX) == 4;
    const int lane = threadIdx.x & 63;
    const uint64_t s = bd_block_index() * 4 + (threadIdx.x >> 6);
    if (s >= n_streams) return; // wave-uniform
    const uint8_t *f = flags + s * flags_stride;
    // a stream shorter than the batch stops at its own last complete callback
    uint64_t n_cb64 = n_cb_max;
    if (lengths) {
        const uint64_t own = lengths[s] / callback_len;
        n_cb64 = own < n_cb_max ? own : n_cb_max;
    }
    const IDX n_cb = (IDX)n_cb64;
    uint32_t last_beat = 0, count = 0;
    IDX j_min = 0;                             // SKIP: decisions before this callback are refused
    auto decide = [&](IDX jj) -> bool {        // the decision at callback jj (wave-uniform)
        const uint32_t beat_time = (uint32_t)jj * step_ms + half_ms; // wraps like u32 adds
        if (beat_time > 0) {                                // lib.rs:120
            const uint32_t threshold = last_beat + gap_ms;  // lib.rs:121
            if (beat_time < threshold) return false;        // lib.rs:122-124
        }
        if (lane == 0 && count < max_beats) {
            if (beat_idx) beat_idx[s * max_beats + count] = (uint32_t)jj;
            if (beat_ms) beat_ms[s * max_beats + count] = beat_time;
        }
        count++;
        last_beat = beat_time;                              // mod.rs:89-91
        if (SKIP) j_min = jj + (IDX)skip;
        return true;
    };
    // bits d and up of a 64-bit mask
    auto from_bit = [](unsigned long long bits, IDX d) -> unsigned long long { return d >= 64 ? 0ull : bits & (~0ull << d); };
    // Long rows (short callbacks: tens of thousands of decisions per stream, nearly all of them zero) on
    // 16-byte boundaries: 1,024 decisions per step, 16 per lane; the lanes that hold a non-zero byte are
    // visited in order and the set bits of their 16-bit masks walked.  The rest of the row, and rows that do
    // not start on a 16-byte boundary, go 64 decisions per step.
    IDX base = 0;
    if ((reinterpret_cast<uintptr_t>(f) & 15) == 0 && n_cb >= 4096) {
        const bd_u32x4 *row = reinterpret_cast<const bd_u32x4 *>(f) + lane;
        bd_u32x4 next = row[0];
        for (; base + 1024 <= n_cb; base += 1024) {
            const bd_u32x4 v = next;
            if (base + 2048 <= n_cb) next = row[(base + 1024) / 16]; // in flight while this step's decisions are walked
            // bit k of m16: byte k of the lane's 16 is a decision
            const uint32_t m16 = bd_nonzero_bytes(v.x) | (bd_nonzero_bytes(v.y) << 4) | (bd_nonzero_bytes(v.z) << 8) | (bd_nonzero_bytes(v.w) << 12);
            unsigned long long lanes = __ballot(m16 != 0);
            if (SKIP && j_min > base) lanes = from_bit(lanes, (j_min - base) >> 4);
            while (lanes) { // wave-uniform
                const int b = __builtin_ctzll(lanes);
                lanes &= lanes - 1;
                const IDX first = base + 16 * (IDX)b;
                uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)m16, b);
                while (m) {
                    const int k = __builtin_ctz(m);
                    m &= m - 1;
                    const IDX jj = first + (IDX)k;
                    if (SKIP && jj < j_min) continue;    // inside the gap of the last beat
                    if (decide(jj) && SKIP && lanes) lanes = from_bit(lanes, (j_min - base) >> 4);
                }
            }
        }
    }
    for (; base < n_cb; base += 64) {
        const IDX j = base + (IDX)lane;
        const bool set = (j < n_cb) && (f[j] != 0);
        unsigned long long mask = __ballot(set);
        if (SKIP && j_min > base) mask = from_bit(mask, j_min - base);
        while (mask) { // wave-uniform
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            if (decide(base + (IDX)b) && SKIP && mask) mask = from_bit(mask, j_min - base);
        }
    }
    if (lane == 0) beat_count[s] = count;
}

// The same pass for rows on 16-byte boundaries with a LANE per stream in the walk.  A wavefront takes S streams
// and goes through them in segments of 4,096 callbacks:
//   pack  a lane loads 16 decisions at a time (4 S independent loads per wavefront and segment), turns them into
//         16 bits and stores those to LDS; a stream's segment is 64 words of 64 bits, bit i of word w = callback
//         64 w + i;
//   walk  lane q < S goes through the words of stream q and applies the rule to the set bits: the walks of the
//         S streams run side by side (in the kernel above a wavefront walks one stream and 63 lanes watch).
// The wavefronts of a workgroup share nothing; each has its own part of the LDS array.  IDX and skip as above.
// The launcher sends rows of fewer than 16 callbacks to the kernel above.
#define BD_RF_SEG_WORDS 64
#define BD_RF_PITCH (BD_RF_SEG_WORDS + 2) // words: rows stay on 16-byte boundaries and start on different banks

template <typename IDX, int S>
__global__ __launch_bounds__(256) void bd_refractory_lanes_kernel(
    const uint8_t *flags, uint64_t flags_stride, uint64_t n_streams, uint64_t n_cb_max,
    const uint64_t *lengths, uint32_t callback_len, uint32_t step_ms, uint32_t half_ms, uint32_t gap_ms, uint32_t skip,
    uint32_t *beat_count, uint32_t *beat_idx, uint32_t *beat_ms, uint32_t max_beats)
{
    constexpr bool SKIP = sizeof(IDX) == 4;
    constexpr uint64_t SEG = 64ull * BD_RF_SEG_WORDS;
    __shared__ __attribute__((aligned(16))) unsigned long long packed[4][S * BD_RF_PITCH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t s0 = (bd_block_index() * 4 + (uint64_t)wave) * S;
    if (s0 >= n_streams) return; // wave-uniform; the kernel has no workgroup barrier
    unsigned long long *mine = packed[wave];
    uint16_t *dst = reinterpret_cast<uint16_t *>(mine);
    const uint64_t n_seg = (n_cb_max + SEG - 1) / SEG;
    const uint64_t full16 = n_cb_max & ~15ull;  // callbacks in complete groups of 16

    // walk state: lane q < S = stream s0 + q
    const uint64_t s = s0 + (uint64_t)lane;
    const bool has = lane < S && s < n_streams;
    uint64_t n_cb64 = 0;
    if (has) {
        n_cb64 = n_cb_max;
        if (lengths) { // a stream shorter than the batch stops at its own last complete callback
            const uint64_t own = lengths[s] / callback_len;
            n_cb64 = own < n_cb_max ? own : n_cb_max;
        }
    }
    const IDX n_cb = (IDX)n_cb64;
    uint32_t last_beat = 0, count = 0;
    IDX j_min = 0;
    auto from_bit = [](unsigned long long bits, IDX d) -> unsigned long long { return d >= 64 ? 0ull : bits & (~0ull << d); };
    auto walk = [&](unsigned long long word, uint64_t first64) { // per lane
        if (first64 >= n_cb64) return;      // the lane's stream has ended
        const IDX first = (IDX)first64;
        if (n_cb - first < 64) word &= (1ull << (n_cb - first)) - 1; // bytes past the stream's decisions
        if (SKIP && j_min > first) word = from_bit(word, j_min - first);
        while (word) {
            const int b = __builtin_ctzll(word);
            word &= word - 1;
            const IDX jj = first + (IDX)b;
            const uint32_t beat_time = (uint32_t)jj * step_ms + half_ms; // wraps like u32 adds
            if (beat_time > 0) {                                // lib.rs:120
                const uint32_t threshold = last_beat + gap_ms;  // lib.rs:121
                if (beat_time < threshold) continue;            // lib.rs:122-124
            }
            if (count < max_beats) {
                if (beat_idx) beat_idx[s * max_beats + count] = (uint32_t)jj;
                if (beat_ms) beat_ms[s * max_beats + count] = beat_time;
            }
            count++;
            last_beat = beat_time;                              // mod.rs:89-91
            if (SKIP) {
                j_min = jj + (IDX)skip;
                word = from_bit(word, j_min - first);
            }
        }
    };

    for (uint64_t seg = 0; seg < n_seg; seg++) {
        const uint64_t seg0 = seg * SEG;
        // ---- pack ----  (4 streams = 16 loads at a time; a descriptor per stream and segment ends at the row's
        // last complete group of 16, so a load past it returns zeros and nothing past the row is read)
        const unsigned seg_bytes = seg0 >= full16 ? 0u : (unsigned)(full16 - seg0 < SEG ? full16 - seg0 : SEG);
#pragma unroll
        for (int q0 = 0; q0 < S; q0 += 4) {
            bd_u32x4 v[4][SEG / 1024];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // a stream past the batch's last reads the last stream's row instead; its bits are never walked
                const uint64_t sa = s0 + q0 + q < n_streams ? s0 + q0 + q : n_streams - 1;
                const __amdgpu_buffer_rsrc_t rs = bd_uniform_rsrc(flags + sa * flags_stride + seg0, seg_bytes);
#pragma unroll
                for (int step = 0; step < (int)(SEG / 1024); step++)
                    v[q][step] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16u * (unsigned)lane, 1024 * step, 0);
            }
            __builtin_amdgcn_sched_barrier(0); // all 16 loads are in flight before the first is waited for
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int step = 0; step < (int)(SEG / 1024); step++) {
                    const bd_u32x4 x = v[q][step];
                    const uint32_t bits = bd_nonzero_bytes(x.x) | (bd_nonzero_bytes(x.y) << 4) | (bd_nonzero_bytes(x.z) << 8) | (bd_nonzero_bytes(x.w) << 12);
                    dst[(q0 + q) * (BD_RF_PITCH * 4) + 64 * step + lane] = (uint16_t)bits;
                }
        }
        if ((n_cb_max & 15) && full16 >= seg0 && full16 < seg0 + SEG) { // the rows' last, cut group: byte by byte
            if (has) {
                const uint8_t *row = flags + s * flags_stride + full16;
                uint32_t bits = 0;
                for (int k = 0; k < (int)(n_cb_max & 15); k++)
                    if (row[k] != 0) bits |= 1u << k;
                dst[lane * (BD_RF_PITCH * 4) + (int)((full16 - seg0) >> 4)] = (uint16_t)bits;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- walk ----
        const uint64_t left = n_cb_max - seg0;
        const int n_words = (int)(left >= SEG ? BD_RF_SEG_WORDS : (left + 63) / 64); // wave-uniform
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(mine + (has ? lane : 0) * BD_RF_PITCH);
        for (int w = 0; w < n_words; w += 2) { // the second word of the last pair may lie past n_words: then it is past every stream's end
            const ulonglong2 pair = src[w >> 1];
            if (__ballot(has && (pair.x | pair.y) != 0) == 0) continue; // wave-uniform
            if (has) {
                walk(pair.x, seg0 + 64ull * w);
                walk(pair.y, seg0 + 64ull * w + 64);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (has) beat_count[s] = count;
}

int bd_launch_refractory(const bd_plan &plan, uint32_t callback_len, const uint8_t *flags,
                         uint64_t flags_stride, uint64_t n_streams, uint64_t n_cb, uint32_t *beat_count,
                         uint32_t *beat_idx, uint32_t *beat_ms, uint32_t max_beats,
                         hipStream_t st, const uint64_t *lengths)
{
    if (n_streams == 0) return BD_OK;
    uint32_t step_ms, half_ms;
    bd_clock_step(plan, callback_len, &step_ms, &half_ms);
    uint32_t skip = 0;
    if (step_ms > 0 && n_cb < (1ull << 31) && n_cb * (uint64_t)step_ms + half_ms + plan.gap_ms < (1ull << 32))
        skip = (uint32_t)(((uint64_t)plan.gap_ms + step_ms - 1) / step_ms);
    // rows on 16-byte boundaries: a lane per stream in the walk; other rows (and rows of a few callbacks): a wavefront per stream
    // Rows on 16-byte boundaries of up to six segments: a lane per stream in the walk.  Its time goes with the
    // number of segments, the other kernel's with the decisions of the densest streams: from about 25,000
    // callbacks per stream on (callbacks of under 100 samples in a minute of audio) the wavefront per stream is faster.
    // Rows off the boundaries or of a few callbacks go that way too.
    constexpr int S = 8; // streams per wavefront, a multiple of 4
    const bool lanes = ((reinterpret_cast<uintptr_t>(flags) | flags_stride) & 15) == 0 && n_cb >= 16 &&
                       n_cb <= 6 * 64 * BD_RF_SEG_WORDS && !bd_tuning_get().refractory_wave;
    const uint64_t grid64 = lanes ? ((n_streams + S - 1) / S + 3) / 4 : (n_streams + 3) / 4;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("too many streams for one launch"); return BD_ERR_UNSUPPORTED; }
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid_dims, dim3(256), 0, st,
                           flags, flags_stride, n_streams, n_cb, lengths, callback_len, step_ms, half_ms, plan.gap_ms, skip,
                           beat_count, beat_idx, beat_ms, max_beats);
    };
    if (lanes) {
        if (skip > 0) launch(bd_refractory_lanes_kernel<uint32_t, S>);
        else launch(bd_refractory_lanes_kernel<uint64_t, S>);
    } else {
        if (skip > 0) launch(bd_refractory_kernel<uint32_t>);
        else launch(bd_refractory_kernel<uint64_t>);
    }
    return bd_check_hip(hipGetLastError(), "launch bd_refractory_kernel");
}
