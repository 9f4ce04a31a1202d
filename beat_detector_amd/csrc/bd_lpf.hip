// LPF strategy, batched: for every callback (window) of every stream, the
// decision of src/strategies/lpf.rs:60-75 -- amplitude gate, single-pole
// low-pass over a fresh copy of the window (the filter restarts every window,
// lpf.rs:66), threshold (0.77 * max) as i16, any(|y| >= threshold).
//
// The recurrence truncates to i16 at every step (the filter runs in place on
// an i16 buffer), so it is not linear and is evaluated exactly in order:
//     y = trunc( y + alpha * (x - y) ),  y(-1) = 0
// with separate f32 multiply and add (no fused multiply-add; this file must be
// compiled with -ffp-contract=off).  Windows are independent, so one lane owns
// one window and 64 consecutive windows of a stream form a wavefront's task.
// To keep HBM reads coalesced, the wavefront loads 64 samples of each of its 64
// windows at a time (8 windows x 128 B per load instruction), transposes them
// through a private LDS tile (144-byte rows: conflict-free ds_read_b128), and
// prefetches the next chunk into registers while it filters the current one.
// The kernel is bound by HBM bandwidth: 2 bytes read per sample, ~6 VALU
// operations per sample.
#include "bd_internal.h"

#include <stdlib.h>

struct bd_lpf_args {
    const int16_t *samples;
    float         *values;
    uint8_t       *flags;
    uint64_t       stride;      // samples between streams
    uint64_t       n_cb;        // callbacks per stream
    uint64_t       out_stride;  // callbacks between the rows of values / flags
    uint64_t       n_tasks;     // n_streams * tasks_per_stream
    uint32_t       groups_per_stream; // ceil(n_cb / 64)
    uint32_t       groups_per_task;   // consecutive groups of 64 windows one wavefront walks (aligned staging)
    uint32_t       tasks_per_stream;  // ceil(groups_per_stream / groups_per_task)
    uint32_t       callback_len;
    uint32_t       min_amp;
    float          alpha;
    float          ratio;
};

// CHUNK = samples of each window staged per step (64 or 128).  An LDS row holds
// CHUNK samples plus 16 B of padding: 144 B or 272 B, i.e. a row stride of 36 or
// 68 dwords, both = 4 (mod 32), which makes the per-lane ds_read_b128 of
// "my row, piece p" conflict-free.

__device__ __forceinline__ void bd_lpf_step(float x, float alpha, float &y, float &ymax, float &xmax)
{
    const float d = x - y;
    const float m = alpha * d;
    const float s = y + m;
    y = truncf(s);
    ymax = fmaxf(ymax, fabsf(y));
    xmax = fmaxf(xmax, fabsf(x));
}

// The same step for code under a per-lane predicate.  After a divergent branch the compiler no longer
// knows that the running maxima are canonical numbers and puts a v_max_f32 v, v, v in front of every
// fmaxf (two more instructions per sample); written as the instruction itself, which skips a NaN
// operand exactly as fmaxf does, none is needed.
__device__ __forceinline__ void bd_lpf_step_masked(float x, float alpha, float &y, float &ymax, float &xmax)
{
    const float d = x - y;
    const float m = alpha * d;
    const float s = y + m;
    y = truncf(s);
    asm("v_max_f32 %0, %1, |%2|" : "=v"(ymax) : "v"(ymax), "v"(y));
    asm("v_max_f32 %0, %1, |%2|" : "=v"(xmax) : "v"(xmax), "v"(x));
}

// `j0` is the first window of the wavefront's group, `rows` the number of valid lanes
__device__ __forceinline__ void bd_lpf_finish(const bd_lpf_args &a, uint64_t s, uint64_t j0, unsigned rows, int lane,
                                              float xmax, float ymax)
{
    const float wmax = fminf(xmax, 32767.f);                 // window_stats.rs:47
    const bool gate = wmax >= (float)a.min_amp;              // lib.rs:135
    // (ratio * max as f32) as i16: truncate, saturate, NaN -> 0  (lpf.rs:74)
    float thr = truncf(a.ratio * wmax);
    thr = (thr != thr) ? 0.f : fminf(fmaxf(thr, -32768.f), 32767.f);
    const bool flag = gate && (ymax >= thr);                 // lpf.rs:75
    const uint64_t o0 = s * a.out_stride + j0;
    // lanes past `rows` fall outside the descriptors and are dropped
    bd_store_stream_u8(bd_out_rsrc(a.flags + o0, rows), (unsigned)lane, flag ? 1 : 0);
    if (a.values) bd_store_stream_f32(bd_out_rsrc(a.values + o0, rows * 4u), (unsigned)lane * 4u, gate ? ymax : 0.f);
}

// ALIGNED: callback_len % 8 == 0, stride % 8 == 0, base 16-byte aligned; otherwise rows at any
// 2-byte-aligned address, cut at the lines of memory.
// WPB: wavefronts per workgroup (each has a private LDS tile).
template <bool ALIGNED, int CHUNK, int WPB>
__global__ __launch_bounds__(64 * WPB) void bd_lpf_window_kernel(bd_lpf_args a)
{
    constexpr int BD_LPF_CHUNK = CHUNK;
    constexpr int PIECES = CHUNK / 8;          // 16-byte pieces per row and chunk
    constexpr int BD_LPF_ROW_U4 = PIECES + 1;
    constexpr int ROWS_PER_LOAD = 64 / PIECES; // rows covered by one load instruction
    __shared__ uint4 tile_all[WPB][64 * BD_LPF_ROW_U4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4 *tile = tile_all[wave];
    const uint64_t task = bd_block_index() * WPB + wave;
    if (task >= a.n_tasks) return; // wave-uniform
    const uint64_t s = task / a.tasks_per_stream;
    const uint32_t g0 = (uint32_t)(task - s * a.tasks_per_stream) * a.groups_per_task;   // first group of the task
    const uint32_t n_groups = a.groups_per_stream - g0 < a.groups_per_task ? a.groups_per_stream - g0 : a.groups_per_task;
    const uint64_t j0 = (uint64_t)g0 * 64;
    const uint32_t L = a.callback_len;
    const int16_t *gp = a.samples + s * a.stride + j0 * L; // first window of the first group
    const uint64_t rows = (a.n_cb - j0 < 64) ? (a.n_cb - j0) : 64;
    // Chunk boundaries are placed on 128-byte lines of memory: when the rows do
    // not start on a line (possible when callback_len % 64 == 0 makes all windows
    // of the group equally misaligned) the first chunk is shortened to `head`
    // samples, so that no line is fetched by two different chunks.
    uint32_t shift = 0;
    if (ALIGNED && (L % 64) == 0) {
        const uint32_t head = (uint32_t)((128 - ((uintptr_t)gp & 127)) & 127) / 2;
        if (head > 0 && head < L) shift = BD_LPF_CHUNK - head;
    }
    const uint32_t n_chunks = (L + shift + BD_LPF_CHUNK - 1) / BD_LPF_CHUNK;
    auto chunk_start = [&](uint32_t ch) { const uint32_t v = ch * BD_LPF_CHUNK; return (v > shift ? v : shift) - shift; };
    auto chunk_end = [&](uint32_t ch) { const uint32_t v = (ch + 1) * BD_LPF_CHUNK; return (v < L + shift ? v : L + shift) - shift; };

    float y = 0.f, ymax = 0.f, xmax = 0.f;
    const float alpha = a.alpha;

    if (ALIGNED) {
        // load instruction i covers windows ROWS_PER_LOAD * i + lane / PIECES,
        // 16-byte piece lane % PIECES: CHUNK * 2 contiguous bytes per window
        const int lrow = lane / PIECES, piece = lane % PIECES;
        uint4 stage[PIECES];
        // A task is a run of groups (64 windows each) of one stream, walked as one sequence of steps
        // (group, chunk): the loads of the next step -- the next chunk, or the first chunk of the next
        // group -- are in flight while this one is filtered, so short windows (one or two chunks per
        // group) keep the pipeline of long ones.
        auto load_chunk = [&](uint32_t g, uint32_t ch) {
            const uint32_t first = chunk_start(ch) + piece * 8; // sample offset inside the window
            const uint32_t end = chunk_end(ch);
            const uint64_t jg = j0 + (uint64_t)g * 64;
            const uint64_t rows_g = (a.n_cb - jg < 64) ? (a.n_cb - jg) : 64;
            const int16_t *gpg = gp + (uint64_t)g * 64 * L;
#pragma unroll
            for (int i = 0; i < PIECES; i++) {
                const uint32_t row = ROWS_PER_LOAD * i + lrow;
                if (row < rows_g && first < end)
                    stage[i] = bd_load_stream16(gpg + (uint64_t)row * L + first);
                else
                    stage[i] = make_uint4(0, 0, 0, 0);
            }
        };
        load_chunk(0, 0);
        for (uint32_t g = 0; g < n_groups; g++) {
            for (uint32_t ch = 0; ch < n_chunks; ch++) {
                // all lanes have finished reading the previous chunk (LDS operations of
                // one wavefront complete in order); keep the compiler from reordering
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < PIECES; i++)
                    tile[(ROWS_PER_LOAD * i + lrow) * BD_LPF_ROW_U4 + piece] = stage[i];
                if (ch + 1 < n_chunks) load_chunk(g, ch + 1);
                else if (g + 1 < n_groups) load_chunk(g + 1, 0);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int pieces = (int)((chunk_end(ch) - chunk_start(ch)) / 8);
                for (int p = 0; p < pieces; p++) {
                    const uint4 v = tile[lane * BD_LPF_ROW_U4 + p];
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int d = (int)w[q];
                        bd_lpf_step((float)(short)(d & 0xffff), alpha, y, ymax, xmax);
                        bd_lpf_step((float)(short)(d >> 16), alpha, y, ymax, xmax);
                    }
                }
            }
            const uint64_t jg = j0 + (uint64_t)g * 64;
            bd_lpf_finish(a, s, jg, (unsigned)((a.n_cb - jg < 64) ? (a.n_cb - jg) : 64), lane, xmax, ymax);
            y = 0.f; ymax = 0.f; xmax = 0.f;      // the filter restarts in every window (lpf.rs:66)
        }
        return;
    } else {
        // Rows at any 2-byte-aligned address (callback length not a multiple of 64, odd strides):
        // every row is cut at the SEG-byte lines of memory it crosses, so that a line is fetched
        // by one chunk of the row only, with 16-byte loads that are aligned again.  Chunk `ch` of
        // row r is line `ch` of that row: its samples [lo, hi) belong to the window, where lo > 0
        // only in the first line and hi < CHUNK only in the last.  A piece that reaches outside the
        // analysed samples of the stream -- possibly outside the buffer -- is gathered sample by
        // sample; pieces that merely reach into the neighbouring window are loaded whole.
        // Positions are bytes from the stream's first sample plus that sample's address modulo SEG
        // (equal to the address modulo SEG), and every load address is the kernel's argument plus an
        // offset: global loads, which return in order, so that the next chunk's loads stay in flight.
        constexpr uint32_t SEG = CHUNK * 2;
        const char *s_ptr = reinterpret_cast<const char *>(a.samples + s * a.stride);
        const int64_t s_begin = (int64_t)(reinterpret_cast<uintptr_t>(s_ptr) & (SEG - 1));
        const int64_t g_addr = s_begin + 2 * (int64_t)(j0 * L);
        const int64_t s_end = s_begin + 2 * (int64_t)(a.n_cb * (uint64_t)L);
        const int lrow = lane / PIECES, piece = lane % PIECES;
        const uint32_t n_lines = (L + 2 * BD_LPF_CHUNK - 2) / BD_LPF_CHUNK; // lines a row can touch
        uint4 stage[PIECES];
        auto load_chunk = [&](uint32_t ch) {
#pragma unroll
            for (int i = 0; i < PIECES; i++) {
                const uint32_t row = ROWS_PER_LOAD * i + lrow;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < rows) {
                    const int64_t row_a = g_addr + 2 * (int64_t)row * L, row_e = row_a + 2 * (int64_t)L;
                    const int64_t pa = (row_a & ~(int64_t)(SEG - 1)) + (int64_t)ch * SEG + 16 * piece;
                    if (pa + 16 > row_a && pa < row_e) {
                        if (pa >= s_begin && pa + 16 <= s_end) {
                            v = bd_load_stream16(s_ptr + (pa - s_begin));
                        } else {
                            unsigned w[4] = {0, 0, 0, 0};
                            for (uint32_t e = 0; e < 8; e++) {
                                const int64_t ea = pa + 2 * e;
                                if (ea >= row_a && ea < row_e)
                                    w[e >> 1] |= (unsigned)*reinterpret_cast<const uint16_t *>(s_ptr + (ea - s_begin)) << (16 * (e & 1));
                            }
                            v = make_uint4(w[0], w[1], w[2], w[3]);
                        }
                    }
                }
                stage[i] = v;
            }
        };
        // this lane's window inside its lines
        const uint32_t off = (uint32_t)((g_addr + 2 * (uint64_t)lane * L) & (SEG - 1)) / 2;
        const uint32_t len = (uint64_t)lane < rows ? L : 0;
        load_chunk(0);
        for (uint32_t ch = 0; ch < n_lines; ch++) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < PIECES; i++)
                tile[(ROWS_PER_LOAD * i + lrow) * BD_LPF_ROW_U4 + piece] = stage[i];
            if (ch + 1 < n_lines) load_chunk(ch + 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t c0 = ch * BD_LPF_CHUNK;                       // first line position of the chunk
            const uint32_t lo = off > c0 ? off - c0 : 0;
            const uint32_t end = off + len;                               // line position after the window
            const uint32_t hi = end > c0 ? (end - c0 < (uint32_t)BD_LPF_CHUNK ? end - c0 : (uint32_t)BD_LPF_CHUNK) : 0;
            const bool whole = lo == 0 && hi == (uint32_t)BD_LPF_CHUNK;
            if (__builtin_amdgcn_ballot_w64(!whole) == 0) {
                for (int p = 0; p < PIECES; p++) {
                    const uint4 v = tile[lane * BD_LPF_ROW_U4 + p];
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int d = (int)w[q];
                        bd_lpf_step((float)(short)(d & 0xffff), alpha, y, ymax, xmax);
                        bd_lpf_step((float)(short)(d >> 16), alpha, y, ymax, xmax);
                    }
                }
            } else {
                const uint32_t span = hi > lo ? hi - lo : 0;
                for (int p = 0; p < PIECES; p++) {
                    const uint4 v = tile[lane * BD_LPF_ROW_U4 + p];
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
                    const uint32_t rel = (uint32_t)(8 * p) - lo;                // position - lo (wraps below lo)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int d = (int)w[q];
                        if (rel + 2 * q < span) bd_lpf_step((float)(short)(d & 0xffff), alpha, y, ymax, xmax);
                        if (rel + 2 * q + 1 < span) bd_lpf_step((float)(short)(d >> 16), alpha, y, ymax, xmax);
                    }
                }
            }
        }
    }
    bd_lpf_finish(a, s, j0, (unsigned)rows, lane, xmax, ymax);
}


// Line staging for rows at any 2-byte-aligned address, callback lengths from 64 samples to 2^22:
// the scheme of the ALIGNED = false branch above -- chunk `ch` of a row is the ch-th 128-byte line of
// memory the row touches, read with aligned 16-byte loads, a lane filters its window's samples
// [lo, hi) of the line -- with two changes that matter where callbacks are a few hundred samples long
// (441: the kernel above issued 12.2 VALU instructions per sample, 7.3 of them the filter, and fetched
// 14 % more bytes than the samples):
//  * addresses are 32-bit offsets from the line that holds the group's first sample, kept per lane and
//    load slot; loads go through a buffer descriptor over the group's bytes, so a row past the end of
//    the stream is an out-of-range offset (dropped, zeros) instead of a branch, and only the first and
//    the last two lines of a row evaluate per-piece conditions;
//  * KEEP: the line in which row r ends is the line in which row r + 1 begins (unless r + 1 starts on a
//    line); fetched for both, several lines apart in time, it travels from memory twice.  The first line of
//    every row stays in registers and is handed to the row before it when that row reaches its last
//    line; only the last row of a group reads its last line from memory.
// A 16-byte piece that holds at least one sample of the analysed stream is loaded whole: it lies inside
// one 128-byte line with that sample, so the bytes of it before the stream's first or after its last
// analysed sample are readable, and the filter's [lo, hi) never looks at them.
template <bool KEEP, int WPB>
__global__ __launch_bounds__(64 * WPB) void bd_lpf_lines_kernel(bd_lpf_args a)
{
    constexpr uint32_t CH = 64, SEG = 128, PIECES = 8, ROW_U4 = PIECES + 1, ROWS_PER_LOAD = 8;
    constexpr uint32_t DROPPED = 0x80000000u; // an offset beyond every descriptor
    __shared__ uint4 tile_all[WPB][64 * ROW_U4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4 *tile = tile_all[wave];
    const uint64_t task = bd_block_index() * WPB + wave;
    if (task >= a.n_tasks) return; // wave-uniform
    const uint64_t s = task / a.groups_per_stream;
    const uint64_t j0 = (task - s * a.groups_per_stream) * 64;
    const uint32_t L = a.callback_len;
    const uint32_t rows = (a.n_cb - j0 < 64) ? (uint32_t)(a.n_cb - j0) : 64u;
    const char *g_ptr = reinterpret_cast<const char *>(a.samples + s * a.stride + j0 * L); // the group's first sample
    const uint32_t c0 = __builtin_amdgcn_readfirstlane((uint32_t)(reinterpret_cast<uintptr_t>(g_ptr) & (SEG - 1)));
    const __amdgpu_buffer_rsrc_t rs = bd_uniform_rsrc(g_ptr - c0, (c0 + 2u * rows * L + 15u) & ~15u); // from its line on
    const uint32_t lrow = (uint32_t)lane / PIECES, piece = (uint32_t)lane % PIECES;
    const uint32_t n_lines = (L + 2 * CH - 2) / CH; // lines a row can touch; it touches n_lines - 1 or n_lines

    // load slot i of this lane: piece `piece` of the lines of row 8 i + lrow; positions are bytes from `base`
    uint32_t o[PIECES];
#pragma unroll
    for (uint32_t i = 0; i < PIECES; i++) {
        const uint32_t row = ROWS_PER_LOAD * i + lrow;
        o[i] = row < rows ? ((c0 + 2u * row * L) & ~(SEG - 1)) + 16u * piece : DROPPED;
    }
    uint4 stage[PIECES], keep[KEEP ? PIECES : 1];
    auto load_chunk = [&](uint32_t ch) {
        if (ch > 0 && ch + 2 < n_lines) { // lines that lie inside their rows
#pragma unroll
            for (uint32_t i = 0; i < PIECES; i++) {
                const bd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o[i] + SEG * ch, 0, 2 /* nt */);
                stage[i] = make_uint4(v.x, v.y, v.z, v.w);
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < PIECES; i++) {
                const uint32_t row = ROWS_PER_LOAD * i + lrow;
                const uint32_t ra = c0 + 2u * row * L, re = ra + 2u * L;
                const uint32_t la = (ra & ~(SEG - 1)) + SEG * ch, pa = la + 16u * piece;
                // pieces that hold samples of the row; the first line of a row that hands it on is loaded whole
                bool want = row < rows && pa < re && (pa + 16u > ra || (KEEP && ch == 0 && row >= 1));
                // the last line comes from the row after this one
                if (KEEP && ch > 0 && row + 1 < rows && la + SEG > re) want = false;
                const bd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, want ? pa : DROPPED, 0, 2 /* nt */);
                stage[i] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
    };
    // this lane's window inside its lines
    const uint32_t off = ((c0 + 2u * (uint32_t)lane * L) & (SEG - 1)) / 2;
    const uint32_t len = (uint32_t)lane < rows ? L : 0;
    float y = 0.f, ymax = 0.f, xmax = 0.f;
    const float alpha = a.alpha;
    load_chunk(0);
    if (KEEP) {
#pragma unroll
        for (uint32_t i = 0; i < PIECES; i++) keep[i] = stage[i];
    }
    for (uint32_t ch = 0; ch < n_lines; ch++) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t i = 0; i < PIECES; i++)
            tile[(ROWS_PER_LOAD * i + lrow) * ROW_U4 + piece] = stage[i];
        if (KEEP && ch + 2 >= n_lines) {
            // a row's last line is its line n_lines - 2 or n_lines - 1 (these stores follow the ones above
            // in program order and win)
#pragma unroll
            for (uint32_t i = 0; i < PIECES; i++) {
                const uint32_t row = ROWS_PER_LOAD * i + lrow;
                const uint32_t ra = c0 + 2u * row * L;
                if (row >= 1 && row < rows && (ra & (SEG - 1)) != 0 && ra / SEG - (ra - 2u * L) / SEG == ch)
                    tile[(row - 1) * ROW_U4 + piece] = keep[i];
            }
        }
        if (ch + 1 < n_lines) load_chunk(ch + 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t p0 = ch * CH;                                  // first line position of the chunk
        const uint32_t lo = off > p0 ? off - p0 : 0;
        const uint32_t end = off + len;                               // line position after the window
        const uint32_t hi = end > p0 ? (end - p0 < CH ? end - p0 : CH) : 0;
        const bool whole = lo == 0 && hi == CH;
        if (__builtin_amdgcn_ballot_w64(!whole) == 0) {
            for (uint32_t p = 0; p < PIECES; p++) {
                const uint4 v = tile[lane * ROW_U4 + p];
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int d = (int)w[q];
                    bd_lpf_step((float)(short)(d & 0xffff), alpha, y, ymax, xmax);
                    bd_lpf_step((float)(short)(d >> 16), alpha, y, ymax, xmax);
                }
            }
        } else if (ch == 0 && n_lines > 2) {
            // the first line of rows of more than a line: a lane's samples are those from `lo` on
#pragma unroll
            for (uint32_t p = 0; p < PIECES; p++) {
                const uint4 v = tile[lane * ROW_U4 + p];
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const int d = (int)w[q];
                    if (lo <= 8 * p + 2 * q) bd_lpf_step_masked((float)(short)(d & 0xffff), alpha, y, ymax, xmax);
                    if (lo <= 8 * p + 2 * q + 1) bd_lpf_step_masked((float)(short)(d >> 16), alpha, y, ymax, xmax);
                }
            }
        } else if (ch > 0 && n_lines > 2) {
            // a row's last line (or, for lanes whose row ends in the next line, a whole one): the samples before `hi`
#pragma unroll
            for (uint32_t p = 0; p < PIECES; p++) {
                const uint4 v = tile[lane * ROW_U4 + p];
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const int d = (int)w[q];
                    if (8 * p + 2 * q < hi) bd_lpf_step_masked((float)(short)(d & 0xffff), alpha, y, ymax, xmax);
                    if (8 * p + 2 * q + 1 < hi) bd_lpf_step_masked((float)(short)(d >> 16), alpha, y, ymax, xmax);
                }
            }
        } else {
            // rows of 64 or 65 samples: a line can hold a row's first and last sample
            const uint32_t span = hi > lo ? hi - lo : 0;
            for (uint32_t p = 0; p < PIECES; p++) {
                const uint4 v = tile[lane * ROW_U4 + p];
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
                const uint32_t rel = 8 * p - lo;                       // position - lo (wraps below lo)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int d = (int)w[q];
                    if (rel + 2 * q < span) bd_lpf_step_masked((float)(short)(d & 0xffff), alpha, y, ymax, xmax);
                    if (rel + 2 * q + 1 < span) bd_lpf_step_masked((float)(short)(d >> 16), alpha, y, ymax, xmax);
                }
            }
        }
    }
    bd_lpf_finish(a, s, j0, rows, lane, xmax, ymax);
}

int bd_launch_lpf(const bd_plan &plan, const int16_t *samples, uint64_t n_streams,
                  uint64_t n_samples, uint64_t stride, uint32_t callback_len,
                  float *values, uint8_t *flags, uint64_t out_stride, hipStream_t st)
{
    const uint64_t n_cb = n_samples / callback_len;
    if (n_cb == 0 || n_streams == 0) return BD_OK;
    // A NaN alpha (the sampling rate wraps to 0 as u16, lpf.rs:69) needs no path of its own: the
    // reference turns every filtered sample into 0 (`NaN as i16`) and compares 0 with the window's
    // threshold; in the kernel y becomes NaN, the running maximum (which skips NaNs) stays 0, and the
    // same comparison follows.
    bd_lpf_args a;
    a.samples = samples; a.values = values; a.flags = flags;
    a.stride = stride; a.n_cb = n_cb; a.out_stride = out_stride;
    const uint64_t groups = (n_cb + 63) / 64;
    if (groups > 0xffffffffull) { bd_set_error("stream too long"); return BD_ERR_UNSUPPORTED; }
    a.groups_per_stream = (uint32_t)groups;
    a.groups_per_task = 1; a.tasks_per_stream = (uint32_t)groups;
    a.n_tasks = n_streams * groups;
    a.callback_len = callback_len; a.min_amp = plan.min_amp;
    a.alpha = plan.alpha; a.ratio = plan.lpf_ratio;
    // Rows that keep their place inside a 128-byte line (callback length a multiple of 64) use the
    // aligned staging; so do short callbacks, where the partial first and last lines of the line
    // staging weigh more than the lines it saves (8,192 x 60 s: 120 samples 5.9 vs 4.2 TB/s, 136: 4.5 vs 4.45,
    // 144: 4.65 vs 4.6, 152: 4.45 vs 4.7, 160: 4.65 vs 5.0; from 168 samples on the two are level).
    const bool aligned = ((uintptr_t)samples % 16 == 0) && (stride % 8 == 0) && (callback_len % 8 == 0) &&
                         (callback_len % 64 == 0 || callback_len <= 144) &&
                         !(bd_tuning_get().lpf_variant == 7 && callback_len >= 64);   // 7: line staging wherever it applies
    // tuning knob for experiments: 0 = 128-sample chunks, 2 wavefronts per workgroup (default);
    // 1 = 128 / 1; 3 = 64-sample chunks, 4 wavefronts
    const int variant_env = bd_tuning_get().lpf_variant;
    // (64-sample chunks where the window is no longer than one, 4,096 x 60 s: 16 samples 2.25 -> 4.15 TB/s,
    // 32: 3.8 -> 5.2, 48: 4.5 -> 5.3, 64: 5.1 -> 5.6; and at 192 and 256 samples: 6.1 -> 6.2, 6.3 -> 6.4)
    const bool chunk64 = callback_len <= 64 || callback_len == 192 || callback_len == 256;
    const int variant = variant_env == 0 ? (chunk64 ? 0 : 2) : (variant_env == 3 ? 0 : variant_env);
    const int wpb = (aligned && variant == 1) ? 1 : (aligned && variant == 2) ? 2 : 4;
    if (aligned) {
        // Windows of up to 128 samples: a wavefront walks 256 / L groups of 64 windows (8,192 streams x 60 s:
        // 64 samples 4.25 -> 4.9 TB/s with four groups, 128 samples 5.4 -> 5.85 with two; from 192 samples on one
        // group per wavefront is as fast or faster), fewer where the batch would leave CUs without work.
        uint32_t gpt = callback_len <= 128 ? 256 / callback_len : 1;
        if (gpt > 8) gpt = 8;
        // (windows of 2,048 samples -- rows 4 KB apart -- are the one longer length that gains from two groups:
        // 6.0 -> 6.35 TB/s at 4,096 streams, 6.2 -> 6.5 at 8,192, 6.35 -> 6.6 at 16,384; 1,024 and 4,096 lose 2 %)
        if (callback_len == 2048) gpt = 2;
        while (gpt > 1 && n_streams * ((groups + gpt - 1) / gpt) < 32768) gpt >>= 1;
        const int knob = bd_tuning_get().lpf_groups;
        if (knob >= 1 && knob <= 64) gpt = (uint32_t)knob;
        a.groups_per_task = gpt;
        a.tasks_per_stream = (uint32_t)((groups + gpt - 1) / gpt);
        a.n_tasks = n_streams * a.tasks_per_stream;
    }
    const uint64_t grid64 = (a.n_tasks + wpb - 1) / wpb;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("batch too large for one launch"); return BD_ERR_UNSUPPORTED; }
    const char *kernel_name = "bd_lpf_window_kernel";
    if (!aligned && callback_len >= 64 && callback_len < (1u << 22)) {
        if (variant_env != 6) kernel_name = "bd_lpf_lines_kernel";
        // BD_LPF_VARIANT 5: every line from memory; 6: the kernel with per-piece 64-bit addressing
        if (variant_env == 5)
            hipLaunchKernelGGL((bd_lpf_lines_kernel<false, 4>), grid_dims, dim3(256), 0, st, a);
        else if (variant_env == 6)
            hipLaunchKernelGGL((bd_lpf_window_kernel<false, 64, 4>), grid_dims, dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((bd_lpf_lines_kernel<true, 4>), grid_dims, dim3(256), 0, st, a);
    } else if (!aligned)
        hipLaunchKernelGGL((bd_lpf_window_kernel<false, 64, 4>), grid_dims, dim3(256), 0, st, a);
    else if (variant == 1)
        hipLaunchKernelGGL((bd_lpf_window_kernel<true, 128, 1>), grid_dims, dim3(64), 0, st, a);
    else if (variant == 2)
        hipLaunchKernelGGL((bd_lpf_window_kernel<true, 128, 2>), grid_dims, dim3(128), 0, st, a);
    else
        hipLaunchKernelGGL((bd_lpf_window_kernel<true, 64, 4>), grid_dims, dim3(256), 0, st, a);
    bd_set_last_kernel(kernel_name);
    return bd_check_hip(hipGetLastError(), "launch LPF kernel");
}
