// Spectrum strategy, batched: for every callback of every stream, the value the
// reference computes at src/strategies/spectrum.rs:82-92 -- the largest
// magnitude among the bins of a 1024-point DFT of the latest 1024 samples whose
// frequency is <= 90 Hz, scaled by `divide_by_N` -- and the decision
// "amplitude gate passed (spectrum.rs:77-80) and value > threshold (:92)".
//
// Only bins 0..2 are ever kept at audio sampling rates (k * fs/1024 <= 90 Hz),
// so no full FFT is evaluated: the kept bins are computed as a pruned DFT, one
// HBM read per sample.  Nothing here is a dense contraction; the kernel is
// bound by HBM bandwidth.
//
// Fast kernel (callback length 512, 1024, or a longer multiple of 512 up to 8192 -- then only the
// callback's last two half-blocks form the ring; 16-byte aligned rows)
//   Samples are consumed in half-blocks of 512.  With
//       P_k(h) = sum_{n<512} x[512 h + n] W^{k n},   W = exp(-2 pi i / 1024)
//   the DFT of the ring [half-block h-1 | half-block h] is
//       X_k = P_k(h-1) + (-1)^k P_k(h)               (W^{512 k} = (-1)^k)
//   so overlapping callbacks (hop 512) share partial sums and every sample is
//   read once.  A wavefront walks a run of consecutive half-blocks of one
//   stream, four per iteration: row g (16 lanes) owns half-block 4*it+g, lane l
//   of the row loads 4 x 16 B (samples 128 c + 8 l + m, c < 4, m < 8), so each
//   row reads 256 contiguous bytes per load instruction.  Per lane the twiddle
//   is split as W^{k(128c+m)} (compile-time constants shared by all lanes)
//   times W^{8 k l} (one complex multiply per lane per half-block).  Partial
//   sums are reduced inside the row with DPP row rotations; the previous
//   half-block's sums come from the row above (row_bcast:15) or, for row 0,
//   from the previous iteration (v_readlane 63).
//
// Window kernel (any callback length, bins 0..2): see the comment above it.
//
// Generic kernel (any callback length, alignment or bin count; stream starts)
//   One wavefront per callback, twiddles from a 1024-entry table.
#include "bd_internal.h"

#include <type_traits>

#include "bd_twiddles.inc"
#include "bd_granule_tw.inc"

typedef short bd_short2 __attribute__((ext_vector_type(2)));

struct bd_spec_args {
    const int16_t *samples;
    float         *values;
    uint8_t       *flags;
    uint64_t       stride;          // samples between streams
    uint64_t       n_tasks;         // n_streams * runs_per_stream
    uint64_t       n_cb;            // callbacks per stream
    uint64_t       out_stride;      // callbacks between the rows of values / flags
    uint32_t       n_hb;            // half-blocks per stream that belong to callbacks
    uint32_t       run_hbs;         // half-blocks per task (multiple of 4)
    uint32_t       runs_per_stream;
    uint32_t       min_amp;
    uint32_t       include_dc;
    float          divisor;
    float          threshold;
    uint32_t       store_mode;      // result-store variant; only builds with BD_STORE_EXPERIMENT look at it
    uint32_t       hb_per_cb;       // half-blocks per callback (callbacks of 1536 .. 8192 samples: 3 .. 16)
};

static __device__ const float bd_cos1024[1024] = {
#include "bd_cos1024.inc"
};
static __device__ const float bd_sin1024[1024] = {
#include "bd_sin1024.inc"
};

// The four 16-byte pieces of one lane (bytes 256 c + 16 l of its half-block) at byte offset `off`
// of the run, through a buffer descriptor that spans exactly the half-blocks the run may read: a
// piece outside it -- the rows before the first half-block of a hop-512 run, the rows and the
// look-ahead iterations past its last one -- is dropped by the range check (it returns zeros and
// touches no memory), so every load of the loop is unconditional and the compiler can count them
// (s_waitcnt vmcnt(N) instead of vmcnt(0)) whatever the depth of the pipeline.
template <unsigned PIECE_STRIDE = 256>
__device__ __forceinline__ void bd_load_hb(uint4 (&d)[4], __amdgpu_buffer_rsrc_t rs, unsigned off)
{
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const bd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + PIECE_STRIDE * c, 0, 2 /* nt */);
        d[c] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

#ifndef BD_SPEC_MAX_RUN
#define BD_SPEC_MAX_RUN 128 // half-blocks per task, upper bound (LDS slots of a run)
#endif

// Sums of one half-block, parked in LDS until the end of the run: two planes of 16-byte entries.
// At the end of the run lane i reads the entries of callback i with 16-byte LDS reads; entries 16 bytes
// apart put the 16 lanes that such a read serves together on 64 different banks (one 32-byte record
// per half-block put two of them on every bank).  Half-block h of the run sits at entry
// bd_spec_entry(h): as it comes for 512-sample callbacks (lane i reads h = i, i + 1), even and odd
// half-blocks in two halves of the plane for 1024-sample callbacks (lane i reads h = 2 i, 2 i + 1); callbacks
// of m > 2 half-blocks keep the plain order (lane i reads h = m i .. m i + m - 1, once per run).
struct __attribute__((aligned(16))) bd_spec_sums { float re1, im1, re2, im2; };
struct __attribute__((aligned(16))) bd_spec_misc { int dc, ny; float amax, pad; };
#define BD_SPEC_SLOTS (BD_SPEC_MAX_RUN + 4)
template <int N> struct bd_spec_slots_t {
    bd_spec_sums sums[N];
    bd_spec_misc misc[N];
};
typedef bd_spec_slots_t<BD_SPEC_SLOTS> bd_spec_slots;
// 128-sample callbacks: a run of 64 half-blocks and the two before it, four eighth-blocks each
#define BD_SPEC_SLOTS_E (4 * (64 + 2))
typedef bd_spec_slots_t<BD_SPEC_SLOTS_E> bd_spec_slots_e;
template <int MODE> __device__ __forceinline__ int bd_spec_entry(int h)
{
    return MODE != 1 ? h : ((h & 1) * (BD_SPEC_SLOTS / 2) + (h >> 1));
}

// One iteration: four half-blocks (one per row of 16 lanes) held in `d`; the row's sums go to
// LDS slot `slot` (a negative slot belongs to a stand-in row and is dropped).
template <int MODE, int NK, bool NYQ>
__device__ __forceinline__ void bd_spec_iter(const uint4 (&d)[4], int slot, int l,
                                             float lc1, float ls1, float lc2, float ls2, bd_spec_slots *slots)
{
    float a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f, amax = 0.f;
    int dc = 0, ny = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const unsigned w[4] = {d[c].x, d[c].y, d[c].z, d[c].w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int v = (int)w[q];
            const float x0 = (float)(short)(v & 0xffff);
            const float x1 = (float)(short)(v >> 16);
            const int i0 = c * 8 + 2 * q, i1 = i0 + 1;
            dc = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_short2, v), bd_short2{1, 1}, dc, false);
            if (NYQ) ny = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_short2, v), bd_short2{1, -1}, ny, false);
            amax = fmaxf(fmaxf(amax, fabsf(x0)), fabsf(x1));
            if (NK >= 1) {
                a1 = __builtin_fmaf(x0, BD_TW_C[0][i0], a1); b1 = __builtin_fmaf(x0, BD_TW_S[0][i0], b1);
                a1 = __builtin_fmaf(x1, BD_TW_C[0][i1], a1); b1 = __builtin_fmaf(x1, BD_TW_S[0][i1], b1);
            }
            if (NK >= 2) {
                a2 = __builtin_fmaf(x0, BD_TW_C[1][i0], a2); b2 = __builtin_fmaf(x0, BD_TW_S[1][i0], b2);
                a2 = __builtin_fmaf(x1, BD_TW_C[1][i1], a2); b2 = __builtin_fmaf(x1, BD_TW_S[1][i1], b2);
            }
        }
    }
    // (a - i b) * (cos - i sin): apply the per-lane offset 8 l, then reduce the row
    bd_spec_sums r;
    bd_spec_misc rm;
    r.re1 = r.im1 = r.re2 = r.im2 = 0.f;
    if (NK >= 1) {
        r.re1 = bd_row_sum(a1 * lc1 - b1 * ls1);
        r.im1 = bd_row_sum(a1 * ls1 + b1 * lc1);
    }
    if (NK >= 2) {
        r.re2 = bd_row_sum(a2 * lc2 - b2 * ls2);
        r.im2 = bd_row_sum(a2 * ls2 + b2 * lc2);
    }
    rm.dc = bd_row_sum(dc);
    rm.ny = NYQ ? bd_row_sum(ny) : 0;
    rm.amax = bd_row_max(amax);
    rm.pad = 0.f;
    // Lane 0 of each row parks the row's sums.  Combining neighbouring half-blocks into
    // callbacks, magnitudes, scaling and the decision happen once per run with all lanes busy
    // (bd_spec_finish) instead of on four lanes in every iteration.
    // (The empty statement keeps the sums computed in every lane: moved into the branch of the one
    // lane that stores them, the last step of each reduction is a lane move plus an add instead of
    // one add that reads its operand through the lane rotation.)
    if (NK >= 1) asm volatile("" :: "v"(r.re1), "v"(r.im1));
    if (NK >= 2) asm volatile("" :: "v"(r.re2), "v"(r.im2));
    if (NYQ) asm volatile("" :: "v"(rm.ny));
    asm volatile("" :: "v"(rm.dc), "v"(rm.amax));
    if (l == 0 && slot >= 0) {
        const int e = bd_spec_entry<MODE>(slot);
        slots->sums[e] = r;
        slots->misc[e] = rm;
    }
}

// all-reduce inside each half of a row (8 lanes): mirror the half, then the two quad permutations
template <int LANES> __device__ __forceinline__ float bd_sub_row_sum(float v)
{
    static_assert(LANES == 8, "half rows");
    v += bd_dpp_f<0x141>(v); v += bd_dpp_f<0xB1>(v); v += bd_dpp_f<0x4E>(v);
    return v;
}
template <int LANES> __device__ __forceinline__ int bd_sub_row_sum(int v)
{
    static_assert(LANES == 8, "half rows");
    v += bd_dpp_i<0x141>(v); v += bd_dpp_i<0xB1>(v); v += bd_dpp_i<0x4E>(v);
    return v;
}
template <int LANES> __device__ __forceinline__ float bd_sub_row_max(float v)
{
    static_assert(LANES == 8, "half rows");
    v = fmaxf(v, bd_dpp_f<0x141>(v)); v = fmaxf(v, bd_dpp_f<0xB1>(v)); v = fmaxf(v, bd_dpp_f<0x4E>(v));
    return v;
}

// Callbacks of 256 and 128 samples (MODE 3, 4): SUB = 2 or 4 callbacks per half-block.  The ring is
// 2 SUB sub-blocks of 512 / SUB samples, and with P_k(q) = sum_n x[(512 / SUB) q + n] W^{k n}
//     X_k = sum_{j < 2 SUB} W^{(512 / SUB) k j} P_k(q - 2 SUB + 1 + j),
// the factors being powers of -i (256 samples) or of exp(-i pi / 4) (128 samples).
// One iteration is still four half-blocks, one per row of 16 lanes, but each half of a row (8 lanes) owns a
// quarter of 256 samples: lane lt of it loads 4 x 16 B (samples 64 c + 8 lt + m), 128 contiguous bytes -- a
// line -- per half-row and load instruction, and the sums are reduced over those 8 lanes.  With 128-sample
// callbacks the half-row takes the two eighth-blocks of its quarter one after the other (pieces c = 0, 1,
// then 2, 3).  (Four lanes per eighth-block, 32 samples per lane and one reduction each, read 64 contiguous
// bytes per load and were slower than the granule kernel.)  Half-block slot `slot` parks its sub-blocks in
// entries SUB slot .. SUB slot + SUB - 1.
template <int SUB, int NK, bool NYQ, typename SLOTS>
__device__ __forceinline__ void bd_spec_iter_sub(const uint4 (&d)[4], int slot, int l,
                                                 float lc1, float ls1, float lc2, float ls2, SLOTS *slots)
{
    constexpr int SEQ = SUB / 2;          // sub-blocks a half-row takes one after the other
    constexpr int PIECES = 4 / SEQ;       // 16-byte pieces of a lane per sub-block
#pragma unroll
    for (int h = 0; h < SEQ; h++) {
        float a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f, amax = 0.f;
        int dc = 0, ny = 0;
#pragma unroll
        for (int c = 0; c < PIECES; c++) {
            const uint4 dv = d[h * PIECES + c];
            const unsigned w[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int v = (int)w[q];
                const float x0 = (float)(short)(v & 0xffff);
                const float x1 = (float)(short)(v >> 16);
                const int i0 = c * 8 + 2 * q, i1 = i0 + 1;
                dc = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_short2, v), bd_short2{1, 1}, dc, false);
                if (NYQ) ny = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_short2, v), bd_short2{1, -1}, ny, false);
                amax = fmaxf(fmaxf(amax, fabsf(x0)), fabsf(x1));
                if (NK >= 1) {
                    a1 = __builtin_fmaf(x0, BD_TWQ_C[0][i0], a1); b1 = __builtin_fmaf(x0, BD_TWQ_S[0][i0], b1);
                    a1 = __builtin_fmaf(x1, BD_TWQ_C[0][i1], a1); b1 = __builtin_fmaf(x1, BD_TWQ_S[0][i1], b1);
                }
                if (NK >= 2) {
                    a2 = __builtin_fmaf(x0, BD_TWQ_C[1][i0], a2); b2 = __builtin_fmaf(x0, BD_TWQ_S[1][i0], b2);
                    a2 = __builtin_fmaf(x1, BD_TWQ_C[1][i1], a2); b2 = __builtin_fmaf(x1, BD_TWQ_S[1][i1], b2);
                }
            }
        }
        bd_spec_sums r;
        bd_spec_misc rm;
        r.re1 = r.im1 = r.re2 = r.im2 = 0.f;
        if (NK >= 1) {
            r.re1 = bd_sub_row_sum<8>(a1 * lc1 - b1 * ls1);
            r.im1 = bd_sub_row_sum<8>(a1 * ls1 + b1 * lc1);
        }
        if (NK >= 2) {
            r.re2 = bd_sub_row_sum<8>(a2 * lc2 - b2 * ls2);
            r.im2 = bd_sub_row_sum<8>(a2 * ls2 + b2 * lc2);
        }
        rm.dc = bd_sub_row_sum<8>(dc);
        rm.ny = NYQ ? bd_sub_row_sum<8>(ny) : 0;
        rm.amax = bd_sub_row_max<8>(amax);
        rm.pad = 0.f;
        if (NK >= 1) asm volatile("" :: "v"(r.re1), "v"(r.im1));
        if (NK >= 2) asm volatile("" :: "v"(r.re2), "v"(r.im2));
        if (NYQ) asm volatile("" :: "v"(rm.ny));
        asm volatile("" :: "v"(rm.dc), "v"(rm.amax));
        if ((l & 7) == 0 && slot >= 0) {
            const int e = SUB * slot + SEQ * (l >> 3) + h;
            slots->sums[e] = r;
            slots->misc[e] = rm;
        }
    }
}

// Callback i of the run: entries i + 1 .. i + 2 SUB are the sub-blocks of its ring, oldest first (the run's
// entries start at 2 SUB, after the two half-blocks before the run).  The stored pair (re, im) stands for the
// complex number re - i im.
template <int SUB, int NK, bool NYQ, typename SLOTS>
__device__ __forceinline__ void bd_spec_finish_sub(const bd_spec_args &a, const SLOTS *slots, unsigned i,
                                                   float *value_out, uint8_t *flag_out)
{
    constexpr int R = 2 * SUB;
    bd_spec_sums p[R];
    int dc = 0, ny = 0;
#pragma unroll
    for (int j = 0; j < R; j++) {
        p[j] = slots->sums[i + 1 + j];
        const bd_spec_misc m = slots->misc[i + 1 + j];
        dc += m.dc; ny += m.ny;
    }
    const float x_dc = (float)dc;
    float mag0 = fabsf(x_dc);
    if (NYQ) {
        const float x_ny = (float)ny;
        mag0 = sqrtf(x_dc * x_dc + x_ny * x_ny);
    }
    float best = a.include_dc ? mag0 : 0.f;
    // z = zr + i zi with zr = re, zi = -im;  (-i) z = zi - i zr
    if (NK >= 1) {
        float xr, xi;
        if (SUB == 2) {       // Z0 - i Z1 - Z2 + i Z3 = (Z0 - Z2) - i (Z1 - Z3)
            const float ar = p[0].re1 - p[2].re1, ai = -(p[0].im1 - p[2].im1);
            const float br = p[1].re1 - p[3].re1, bi = -(p[1].im1 - p[3].im1);
            xr = ar + bi; xi = ai - br;
        } else {              // with D_j = Z_j - Z_{j+4} and w = exp(-i pi / 4): (D0 - i D2) + w (D1 - i D3)
            const float d0r = p[0].re1 - p[4].re1, d0i = -(p[0].im1 - p[4].im1);
            const float d1r = p[1].re1 - p[5].re1, d1i = -(p[1].im1 - p[5].im1);
            const float d2r = p[2].re1 - p[6].re1, d2i = -(p[2].im1 - p[6].im1);
            const float d3r = p[3].re1 - p[7].re1, d3i = -(p[3].im1 - p[7].im1);
            const float ar = d0r + d2i, ai = d0i - d2r;
            const float br = d1r + d3i, bi = d1i - d3r;
            const float h = 0x1.6a09e6p-1f;   // sqrt(1/2): w b = ((br + bi) + i (bi - br)) sqrt(1/2)
            xr = ar + h * (br + bi); xi = ai + h * (bi - br);
        }
        best = fmaxf(best, sqrtf(xr * xr + xi * xi));
    }
    if (NK >= 2) {
        float xr, xi;
        if (SUB == 2) {       // Z0 - Z1 + Z2 - Z3
            xr = (p[0].re2 + p[2].re2) - (p[1].re2 + p[3].re2); xi = (p[0].im2 + p[2].im2) - (p[1].im2 + p[3].im2);
        } else {              // sum (-i)^j Z_j = (Z0 - Z2 + Z4 - Z6) - i (Z1 - Z3 + Z5 - Z7)
            const float ar = (p[0].re2 - p[2].re2) + (p[4].re2 - p[6].re2), ai = -((p[0].im2 - p[2].im2) + (p[4].im2 - p[6].im2));
            const float br = (p[1].re2 - p[3].re2) + (p[5].re2 - p[7].re2), bi = -((p[1].im2 - p[3].im2) + (p[5].im2 - p[7].im2));
            xr = ar + bi; xi = ai - br;
        }
        best = fmaxf(best, sqrtf(xr * xr + xi * xi));
    }
    const float value = (a.divisor == 0.f) ? best : best / a.divisor;
    const float wmax = fminf(slots->misc[i + R].amax, 32767.f);   // the callback's own samples (spectrum.rs:77)
    const bool gate = wmax >= (float)a.min_amp;
    *value_out = value;
    *flag_out = (gate && (value > a.threshold)) ? 1 : 0;
}

// Callback i of the run from the sums of the two half-blocks its ring consists of: X_k = P_k(prev) +
// (-1)^k P_k(cur).  A callback of m > 2 half-blocks ends on them; the ones before only feed the gate.
template <int MODE, int NK, bool NYQ>
__device__ __forceinline__ void bd_spec_finish(const bd_spec_args &a, const bd_spec_slots *slots, unsigned i,
                                               float *value_out, uint8_t *flag_out)
{
    const int m = MODE == 2 ? (int)a.hb_per_cb : 2;
    const int ep = bd_spec_entry<MODE>(MODE == 0 ? (int)i : m * (int)i + m - 2);
    const int ec = bd_spec_entry<MODE>(MODE == 0 ? (int)i + 1 : m * (int)i + m - 1);
    const bd_spec_sums p = slots->sums[ep], c = slots->sums[ec];
    const bd_spec_misc pm = slots->misc[ep], cm = slots->misc[ec];
    const float x_dc = (float)(pm.dc + cm.dc);
    float m0 = fabsf(x_dc);
    if (NYQ) {
        const float x_ny = (float)(pm.ny + cm.ny);
        m0 = sqrtf(x_dc * x_dc + x_ny * x_ny);
    }
    float best = a.include_dc ? m0 : 0.f;
    if (NK >= 1) {
        const float xr = p.re1 - c.re1, xi = p.im1 - c.im1;
        best = fmaxf(best, sqrtf(xr * xr + xi * xi));
    }
    if (NK >= 2) {
        const float xr = p.re2 + c.re2, xi = p.im2 + c.im2;
        best = fmaxf(best, sqrtf(xr * xr + xi * xi));
    }
    const float value = (a.divisor == 0.f) ? best : best / a.divisor;
    // the amplitude gate looks at the callback's own samples (spectrum.rs:77): one half-block
    // for 512-sample callbacks, both for 1024, all m of a longer callback
    float amax = MODE == 0 ? cm.amax : fmaxf(cm.amax, pm.amax);
    if (MODE == 2)
        for (int h = 0; h < m - 2; h++) amax = fmaxf(amax, slots->misc[m * (int)i + h].amax);
    const float wmax = fminf(amax, 32767.f);
    const bool gate = wmax >= (float)a.min_amp;
    *value_out = value;
    *flag_out = (gate && (value > a.threshold)) ? 1 : 0;
}

// MODE 0: callback length 512 (hop 512); MODE 1: callback length 1024; MODE 2: 512 m for m = 3 .. 16
// (runs are whole callbacks: a.run_hbs is a multiple of m and of 4); MODE 3 / 4: callback length 256 / 128 (runs
// of at most 64 half-blocks: their callbacks and the sub-blocks of the two half-blocks before them fill the LDS slots).
// NK: number of non-DC bins evaluated (0..2).  NYQ: bin 0 carries the Nyquist sum.
// DEPTH: register buffers of one iteration (4 KB per wavefront) each; loads run DEPTH - 1 iterations ahead.
template <int MODE, int NK, bool NYQ, int DEPTH>
__global__ __launch_bounds__(256) void bd_spectrum_band_kernel(bd_spec_args a)
{
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, l = lane & 15;
    // everything that selects the task is the same in all lanes: keep it scalar
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t task = bd_block_index() * 4 + wave;
    if (task >= a.n_tasks) {
        if (MODE == 0 || MODE >= 3) __syncthreads();    // the block's other wavefronts meet at one barrier (below)
        return;
    }
    const uint64_t s = task / a.runs_per_stream;
    const uint32_t r = (uint32_t)(task - s * a.runs_per_stream);
    const int64_t hb0 = (int64_t)r * a.run_hbs;
    const int16_t *sp = a.samples + s * a.stride;
    const int64_t n_hb = a.n_hb;
    // iterations of this run that contain at least one half-block of the stream
    const int64_t left = n_hb - hb0;
    const int n_it = (int)((left < (int64_t)a.run_hbs ? left + 3 : (int64_t)a.run_hbs) / 4);
    // hop 512: the first callback of a run needs the half-block before the run.  The four wavefronts of a
    // block walk four consecutive runs, so three of them find that half-block's sums in their
    // neighbour's LDS slots once the block has met at a barrier; only a block's first wavefront (and a
    // stream's first run after another stream's last) reads the half-block again.
    // 256- and 128-sample callbacks: the first callbacks of a run need the three quarter-blocks (seven eighth-
    // blocks) before it, i.e. the two half-blocks before the run: from the neighbour's slots like the one
    // half-block of hop 512, or read again.
    constexpr int SUB = MODE == 3 ? 2 : MODE == 4 ? 4 : 1;     // callbacks per half-block
    constexpr int SUB_LANES = SUB > 1 ? 8 : 16;               // lanes that share a reduction
    const bool from_neighbour = (MODE == 0 || SUB > 1) && wave > 0 && r > 0;
    const bool reread = (MODE == 0 || SUB > 1) && hb0 > 0 && !from_neighbour;
    const int it_begin = reread ? -1 : 0;
    const int64_t hb_lo = reread ? hb0 - (SUB > 1 ? 2 : 1) : hb0;   // first half-block the run reads

    const int lt = l & (SUB_LANES - 1);                      // position of the lane's samples: 8 lt + m
    const float lc1 = BD_LANE_C[0][lt], ls1 = BD_LANE_S[0][lt];
    const float lc2 = BD_LANE_C[1][lt], ls2 = BD_LANE_S[1][lt];

    // slot 0 of a hop-512 run is the half-block before the run; the others follow
    typedef typename std::conditional<MODE == 4, bd_spec_slots_e, bd_spec_slots>::type slots_t;
    __shared__ slots_t lds_slots[4];
    slots_t *slots = &lds_slots[wave];
    constexpr int SLOT0 = MODE == 0 ? 1 : (SUB > 1 ? 2 : 0);
    if (SUB > 1 && hb0 == 0 && lane < 2 * SUB) {
        bd_spec_sums z;
        bd_spec_misc zm;
        z.re1 = z.im1 = z.re2 = z.im2 = zm.amax = zm.pad = 0.f;
        zm.dc = zm.ny = 0;
        slots->sums[lane] = z;
        slots->misc[lane] = zm;
    }
    if (MODE == 0 && hb0 == 0 && lane == 0) {
        // before the stream the ring holds zeros (spectrum.rs:50-51)
        bd_spec_sums z;
        bd_spec_misc zm;
        z.re1 = z.im1 = z.re2 = z.im2 = zm.amax = zm.pad = 0.f;
        zm.dc = zm.ny = 0;
        slots->sums[0] = z;
        slots->misc[0] = zm;
    }

    // DEPTH register buffers in a ring, each filled DEPTH - 1 iterations ahead of its use.  Rows whose
    // loads were dropped produce sums of zeros that no stored result uses.
    const int64_t hb_end = n_hb < hb0 + (int64_t)a.run_hbs ? n_hb : hb0 + (int64_t)a.run_hbs;
    // (256- and 128-sample callbacks: the callbacks may end inside the stream's last half-block; the descriptor ends with them)
    int64_t s_end = hb_end * BD_HALF_BLOCK;
    if (SUB > 1 && (int64_t)a.n_cb * (BD_HALF_BLOCK / SUB) < s_end) s_end = (int64_t)a.n_cb * (BD_HALF_BLOCK / SUB);
    const __amdgpu_buffer_rsrc_t rs = bd_uniform_rsrc(sp + hb_lo * BD_HALF_BLOCK, (unsigned)((s_end - hb_lo * BD_HALF_BLOCK) * 2));
    // byte offset of this lane's first piece in iteration 0, counted from half-block hb_lo (negative,
    // i.e. out of range as an unsigned offset, for the rows of iteration -1 before it)
    const int lane_off = (int)(hb0 - hb_lo + g) * (BD_HALF_BLOCK * 2) + (SUB > 1 ? 512 * (l >> 3) : 0) + 16 * lt;
    uint4 R[DEPTH][4];
    auto load = [&](uint4 (&d)[4], int it) {
        bd_load_hb<16u * SUB_LANES>(d, rs, (unsigned)(lane_off + it * (4 * BD_HALF_BLOCK * 2)));
    };
    auto run = [&](const uint4 (&d)[4], int it) {
        if constexpr (SUB > 1) bd_spec_iter_sub<SUB, NK, NYQ>(d, 4 * it + g + SLOT0, l, lc1, ls1, lc2, ls2, slots);
        else bd_spec_iter<MODE, NK, NYQ>(d, 4 * it + g + SLOT0, l, lc1, ls1, lc2, ls2, slots);
    };
    {
        int it = it_begin;
#pragma unroll
        for (int d = 0; d < DEPTH - 1; d++) load(R[d], it + d);
        // whole trips of DEPTH iterations (one entry, one back edge, no exit inside: the waits between
        // a load and its use stay counted), then the up to DEPTH - 1 iterations left, without loads
        const int total = n_it - it_begin;
        for (int trip = total / DEPTH; trip > 0; trip--) {
#pragma unroll
            for (int b = 0; b < DEPTH; b++) {
                load(R[(b + DEPTH - 1) % DEPTH], it + DEPTH - 1);
                run(R[b], it);
                ++it;
            }
        }
#pragma unroll
        for (int b = 0; b < DEPTH - 1; b++)
            if (it < n_it) { run(R[b], it); ++it; }
    }
    // results of the run: consecutive lanes finish and store consecutive callbacks
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (MODE == 0 || SUB > 1) {
        __syncthreads();                       // every wavefront of the block has parked its sums
        if (MODE == 0 && from_neighbour && lane == 0) {
            // the last half-block of the run before this one (a whole run: this is not its stream's first)
            const int e = bd_spec_entry<MODE>((int)a.run_hbs);
            slots->sums[0] = lds_slots[wave - 1].sums[e];
            slots->misc[0] = lds_slots[wave - 1].misc[e];
        }
        if (SUB > 1 && from_neighbour && lane < 2 * SUB) {
            // the sub-blocks of the last two half-blocks of the run before this one: its last 2 SUB entries
            const int e = SUB * (int)a.run_hbs + lane;
            slots->sums[lane] = lds_slots[wave - 1].sums[e];
            slots->misc[lane] = lds_slots[wave - 1].misc[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint64_t cb0 = MODE == 2 ? (uint64_t)hb0 / a.hb_per_cb : SUB > 1 ? (uint64_t)hb0 * SUB : (uint64_t)(hb0 >> MODE);
    const uint64_t cb_left = a.n_cb - cb0;
    const unsigned n_res_run = MODE == 2 ? (unsigned)(n_it * 4) / a.hb_per_cb : (unsigned)(n_it * (SUB > 1 ? 4 * SUB : MODE == 0 ? 4 : 2));
    const unsigned n_res = cb_left < n_res_run ? (unsigned)cb_left : n_res_run; // buffer stores past it are dropped
    const uint64_t o0 = s * a.out_stride + cb0;
    __amdgpu_buffer_rsrc_t rf = bd_out_rsrc(a.flags + o0, n_res);
    __amdgpu_buffer_rsrc_t rv = bd_out_rsrc(a.values ? a.values + o0 : nullptr, a.values ? n_res * 4u : 0u);
#ifdef BD_STORE_EXPERIMENT
    // Experiment build (tools/exp_store.py): the same results through other store policies, or with
    // one of the two outputs dropped or redirected, to find what the result stores cost.
    if (a.store_mode == 7) rf = bd_out_rsrc(nullptr, 0);                       // no decisions either
    if (a.store_mode == 8 && a.values)                                          // values into one 1 MiB window
        rv = bd_out_rsrc(a.values + (o0 & 0x3ffffull & ~127ull), n_res * 4u);
    for (unsigned i = lane; i < n_res_run; i += 64) {
        float value; uint8_t flag;
        if constexpr (SUB > 1) bd_spec_finish_sub<SUB, NK, NYQ>(a, slots, i, &value, &flag);
        else bd_spec_finish<MODE, NK, NYQ>(a, slots, i, &value, &flag);
        bd_store_stream_u8(rf, i, flag);
        const unsigned vb = __builtin_bit_cast(unsigned, value);
        switch (a.store_mode) {
        case 1: __builtin_amdgcn_raw_buffer_store_b32(vb, rv, i * 4u, 0, 0); break;
        case 2: __builtin_amdgcn_raw_buffer_store_b32(vb, rv, i * 4u, 0, 2); break;
        case 3: __builtin_amdgcn_raw_buffer_store_b32(vb, rv, i * 4u, 0, 16); break;
        case 4: __builtin_amdgcn_raw_buffer_store_b32(vb, rv, i * 4u, 0, 19); break;
        case 5: __builtin_amdgcn_raw_buffer_store_b32(vb, rv, i * 4u, 0, 17); break;
        case 6: __builtin_amdgcn_raw_buffer_store_b32(vb, rv, i * 4u, 0, 1); break;
        default: bd_store_stream_f32(rv, i * 4u, value); break;
        }
    }
#else
    for (unsigned i = lane; i < n_res_run; i += 64) {
        float value; uint8_t flag;
        if constexpr (SUB > 1) bd_spec_finish_sub<SUB, NK, NYQ>(a, slots, i, &value, &flag);
        else bd_spec_finish<MODE, NK, NYQ>(a, slots, i, &value, &flag);
        bd_store_stream_u8(rf, i, flag);
        bd_store_stream_f32(rv, i * 4u, value);   // a zero-sized descriptor drops the store
    }
#endif
}

// ---- generic path ---------------------------------------------------------
struct bd_spec_generic_args {
    const int16_t *samples;
    float         *values;
    uint8_t       *flags;
    uint64_t       stride;
    uint64_t       n_cb;     // callbacks per stream evaluated by this launch: j_first .. j_first + n_cb - 1
    uint64_t       j_first;
    uint64_t       n_total;  // n_streams * n_cb
    uint64_t       out_stride;
    uint32_t       callback_len;
    uint32_t       min_amp;
    int            first_bin, last_bin;
    uint32_t       nyq;
    float          divisor;
    float          threshold;
};

__global__ __launch_bounds__(256) void bd_spectrum_generic_kernel(bd_spec_generic_args a)
{
    const int lane = threadIdx.x & 63;
    const uint64_t t = bd_block_index() * 4 + (threadIdx.x >> 6);
    if (t >= a.n_total) return;
    const uint64_t s = t / a.n_cb, j = a.j_first + (t - s * a.n_cb);
    const int16_t *sp = a.samples + s * a.stride;
    const int64_t end = (int64_t)(j + 1) * a.callback_len;
    // ring = samples [end-1024, end), zeros before the stream starts (spectrum.rs:50-51);
    // lane l holds the 16 consecutive ring positions 16 l + i: two 16-byte loads at an address
    // that is only 2-byte aligned (global memory allows it), 2 KB contiguous per wavefront
    float x[16];
    const int64_t start = end - BD_RING_LEN + 16 * lane;
    if (end >= BD_RING_LEN) {
        typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
        const u32x4_a2 *p = reinterpret_cast<const u32x4_a2 *>(sp + start);
        const bd_u32x4 lo = p[0], hi = p[1];
        const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            x[2 * q] = (float)(short)(w[q] & 0xffff);
            x[2 * q + 1] = (float)(short)(w[q] >> 16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = start + i >= 0 ? (float)sp[start + i] : 0.f;
    }
    // amplitude gate over the callback's own samples (spectrum.rs:77): the tail of the ring,
    // plus whatever of a longer callback lies before the ring
    float amax = 0.f;
    const int own_from = BD_RING_LEN - (int)(a.callback_len < BD_RING_LEN ? a.callback_len : BD_RING_LEN);
#pragma unroll
    for (int i = 0; i < 16; i++) amax = fmaxf(amax, (16 * lane + i >= own_from) ? fabsf(x[i]) : 0.f);
    for (int64_t idx = end - a.callback_len + lane; idx < end - BD_RING_LEN; idx += 64)
        amax = fmaxf(amax, fabsf((float)sp[idx]));
    amax = fminf(bd_wave_max(amax), 32767.f);
    float best = 0.f;
    bool first = true;
    for (int k = a.first_bin; k <= a.last_bin; k++) {
        // W^{k (16 l + i)} = W^{16 k l} W^{k i}: sixteen table values shared by the wavefront
        // (scalar loads), one complex multiply per lane
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = (k * i) & (BD_RING_LEN - 1);
            re = __builtin_fmaf(x[i], bd_cos1024[m], re);
            im = __builtin_fmaf(x[i], bd_sin1024[m], im);
        }
        const int ml = (k * 16 * lane) & (BD_RING_LEN - 1);
        const float lc = bd_cos1024[ml], ls = bd_sin1024[ml];
        float pr = re * lc - im * ls, pi = re * ls + im * lc;
        pr = bd_wave_sum(pr);
        pi = bd_wave_sum(pi);
        if (k == 0 && a.nyq) {
            float ny = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) ny += (i & 1) ? -x[i] : x[i];
            pi = bd_wave_sum(ny);
        }
        const float mag = sqrtf(pr * pr + pi * pi);
        best = first ? mag : fmaxf(best, mag);
        first = false;
    }
    const float value = (a.divisor == 0.f) ? best : best / a.divisor;
    const bool flag = (amax >= (float)a.min_amp) && (value > a.threshold);
    if (lane == 0) {
        const uint64_t o = s * a.out_stride + j;
        if (a.values) a.values[o] = value;
        a.flags[o] = flag ? 1 : 0;
    }
}

// ---- window path ------------------------------------------------------------
// Any callback length, kept bins 0 .. 2.  One wavefront walks up to 64 consecutive
// callbacks of one stream whose ring lies inside the stream (the few callbacks
// before that go to the generic kernel).  Per callback the wavefront reads the
// ring with two 16-byte loads per lane at 2-byte-aligned addresses (2 KB
// contiguous; overlapping rings of neighbouring callbacks come from the caches),
// lane l owns ring positions 16 l .. 16 l + 15: pruned DFT with compile-time
// twiddles w^{k i}, one rotation by the lane's w^{16 k l}, then one reduction
// network for all four sums: two lane-half swaps and one row swap leave
// (re1, re2, im1, im2) spread over the four rows of 16 lanes, and four DPP steps
// finish them.  The amplitude gate is a ballot (some lane holds a sample of the
// callback with |x| >= min_amplitude).  Sums are parked in LDS; at the end of the
// run lane i turns callback i's sums into value and decision.
struct bd_win_args {
    const int16_t *samples;
    float         *values;
    uint8_t       *flags;
    uint64_t       stride, out_stride, n_cb, n_tasks;
    uint64_t       j_min;           // first callback whose ring starts inside the stream
    uint32_t       runs_per_stream, callback_len;
    uint32_t       min_amp, include_dc;
    uint32_t       nk_keep;         // non-DC bins that count (<= the NK the kernel evaluates)
    float          divisor, threshold;
};

// The two lane-swap instructions of gfx950, as assembly: with the builtins the compiler adds the
// first result to itself when the operands are floats.  Two wait states separate a VALU write from
// the swap that reads it.
// v_permlane32_swap: lanes 32-63 of a <-> lanes 0-31 of b.
// v_permlane16_swap: rows 1, 3 of a <-> rows 0, 2 of b (rows of 16 lanes).
template <typename T>
__device__ __forceinline__ void bd_swap_halves(T &a, T &b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <typename T>
__device__ __forceinline__ void bd_swap_rows(T &a, T &b)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// lanes 0-31: a[l] + a[l + 32]; lanes 32-63: b[l - 32] + b[l]
template <typename T>
__device__ __forceinline__ T bd_halves_sum(T a, T b)
{
    bd_swap_halves(a, b);
    return a + b;
}
// rows 0, 2: a[row] + a[row + 1]; rows 1, 3: b[row - 1] + b[row]   (lane by lane)
template <typename T>
__device__ __forceinline__ T bd_rows_sum(T a, T b)
{
    bd_swap_rows(a, b);
    return a + b;
}

__device__ __forceinline__ void bd_win_absmax(float &m, float a, float b)
{
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
}

// NK: non-DC bins evaluated: 0 .. 2 at the usual sampling rates; 3, 4, 5, 8 or 11 where the band
// up to 90 Hz holds more bins (down to 8 kHz), of which the first nk_keep count.  Bins go through
// the reduction network in pairs.  NYQ: bin 0 carries the Nyquist sum.  WHOLE: the callback's own
// samples start on a lane boundary of the ring (callback length a multiple of 16, or >= 1024).
template <int NK, bool NYQ, bool WHOLE>
__global__ __launch_bounds__(256) void bd_spectrum_window_kernel(bd_win_args a)
{
    constexpr int PAIRS = (NK + 1) / 2;
    __shared__ __attribute__((aligned(16))) float st_f[4][64][PAIRS > 0 ? PAIRS : 1][4];
    __shared__ __attribute__((aligned(8))) int st_i[4][64][2];

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t task = bd_block_index() * 4 + wave;
    if (task >= a.n_tasks) return;
    const uint64_t s = task / a.runs_per_stream;
    const uint64_t j0 = a.j_min + (task - s * a.runs_per_stream) * 64;
    const uint64_t left = a.n_cb - j0;
    const int n_win = left < 64 ? (int)left : 64;
    const int16_t *sp = a.samples + s * a.stride;
    const uint64_t L = a.callback_len;

    // w^{16 k l}, w = exp(2 pi i / 1024) (conjugate DFT kernel: same magnitudes)
    float lc[NK > 0 ? NK : 1], ls[NK > 0 ? NK : 1];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int ml = ((k + 1) * 16 * lane) & (BD_RING_LEN - 1);
        lc[k] = bd_cos1024[ml]; ls[k] = bd_sin1024[ml];
    }
    // the callback's own samples are the ring positions >= own_from (spectrum.rs:77)
    const int own_from = L < BD_RING_LEN ? BD_RING_LEN - (int)L : 0;
    const int t_own = own_from - 16 * lane;   // this lane's positions i >= t_own count

    typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
    bd_u32x4 n_lo, n_hi;
    auto load = [&](uint64_t j) {
        const u32x4_a2 *p = reinterpret_cast<const u32x4_a2 *>(sp + ((j + 1) * L - BD_RING_LEN) + 16 * lane);
        n_lo = p[0]; n_hi = p[1];
    };
    load(j0);
    uint64_t gate_mask = 0;
    for (int i = 0; i < n_win; i++) {
        const bd_u32x4 lo = n_lo, hi = n_hi;
        if (i + 1 < n_win) load(j0 + i + 1);
        const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        float x[16];
        int dc = 0, ny = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int v = (int)w[q];
            x[2 * q] = (float)(short)(v & 0xffff);
            x[2 * q + 1] = (float)(short)(v >> 16);
            dc = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_short2, v), bd_short2{1, 1}, dc, false);
            if (NYQ) ny = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_short2, v), bd_short2{1, -1}, ny, false);
        }
        // amplitude gate
        float amax = 0.f;
        if (WHOLE) {
#pragma unroll
            for (int q = 0; q < 8; q++) bd_win_absmax(amax, x[2 * q], x[2 * q + 1]);
            amax = t_own <= 0 ? amax : 0.f;
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++)
                bd_win_absmax(amax, 2 * q >= t_own ? x[2 * q] : 0.f, 2 * q + 1 >= t_own ? x[2 * q + 1] : 0.f);
        }
        if (L > BD_RING_LEN) {
            // The part of a longer callback that lies before the ring, 512 samples per step (16 bytes per
            // lane).  A step that starts before the ring may reach up to 511 samples into it: those belong
            // to the callback as well, so nothing needs masking.
            const int64_t end = (int64_t)((j0 + i + 1) * L);
            const int64_t head_end = end - BD_RING_LEN;
            for (int64_t c = end - (int64_t)L; c < head_end; c += 512) {
                const bd_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a2 *>(sp + c + 8 * lane));
                const unsigned hw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    bd_win_absmax(amax, (float)(short)(hw[q] & 0xffff), (float)(short)((int)hw[q] >> 16));
            }
        }
        const bool gate = __builtin_amdgcn_ballot_w64(fminf(amax, 32767.f) >= (float)a.min_amp) != 0;
        gate_mask |= gate ? (1ull << i) : 0ull;

        // pruned DFT of the lane's 16 positions, rotated to the lane's place in the ring
        if (NK > 0) {
            float pr[NK > 0 ? NK : 1], pi[NK > 0 ? NK : 1];
#pragma unroll
            for (int k = 0; k < NK; k++) {
                float re = x[0], im = 0.f;
#pragma unroll
                for (int n = 1; n < 16; n++) {
                    re = __builtin_fmaf(x[n], BD_GR_C[k][n], re);
                    im = __builtin_fmaf(x[n], BD_GR_S[k][n], im);
                }
                pr[k] = __builtin_fmaf(re, lc[k], -(im * ls[k]));
                pi[k] = __builtin_fmaf(re, ls[k], im * lc[k]);
            }
            // bins k, k + 1 of a pair: the rows of 16 lanes end up with (re_k, re_k+1, im_k, im_k+1);
            // a last single bin with (re, re, im, im)
#pragma unroll
            for (int pp = 0; pp < PAIRS; pp++) {
                float v = bd_halves_sum(pr[2 * pp], pi[2 * pp]);
                if (2 * pp + 1 < NK) v = bd_rows_sum(v, bd_halves_sum(pr[2 * pp + 1 < NK ? 2 * pp + 1 : 0], pi[2 * pp + 1 < NK ? 2 * pp + 1 : 0]));
                else v = bd_rows_sum(v, v);
                v = bd_row_sum(v);
                if ((lane & 15) == 0) st_f[wave][i][pp][lane >> 4] = v;
            }
        }
        {   // halves of 32 lanes end up with (dc, ny)
            int v = bd_halves_sum(dc, NYQ ? ny : dc);
            v = bd_rows_sum(v, v);
            v = bd_row_sum(v);
            if ((lane & 31) == 0) st_i[wave][i][lane >> 5] = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane i finishes callback j0 + i
    float value = 0.f;
    bool flag = false;
    if (lane < n_win) {
        const float x_dc = (float)st_i[wave][lane][0];
        float m0 = fabsf(x_dc);
        if (NYQ) {
            const float x_ny = (float)st_i[wave][lane][1];
            m0 = sqrtf(x_dc * x_dc + x_ny * x_ny);
        }
        float best = a.include_dc ? m0 : 0.f;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            if (NK <= 2 || k < (int)a.nk_keep) {
                const float xr = st_f[wave][lane][k / 2][k & 1], xi = st_f[wave][lane][k / 2][2 + (k & 1)];
                best = fmaxf(best, sqrtf(xr * xr + xi * xi));
            }
        }
        value = (a.divisor == 0.f) ? best : best / a.divisor;
        flag = ((gate_mask >> lane) & 1ull) && value > a.threshold;
    }
    const uint64_t o0 = s * a.out_stride + j0;
    __amdgpu_buffer_rsrc_t rf = bd_out_rsrc(a.flags + o0, (unsigned)n_win);
    bd_store_stream_u8(rf, (unsigned)lane, flag ? 1 : 0);          // stores past n_win are dropped
    if (a.values) {
        __amdgpu_buffer_rsrc_t rv = bd_out_rsrc(a.values + o0, (unsigned)n_win * 4u);
        bd_store_stream_f32(rv, (unsigned)lane * 4u, value);
    }
}

template <int NK, bool NYQ>
static void bd_win_launch2(bool whole, const bd_win_args &a, dim3 grid, hipStream_t st)
{
    if (whole) hipLaunchKernelGGL((bd_spectrum_window_kernel<NK, NYQ, true>), dim3(grid), dim3(256), 0, st, a);
    else       hipLaunchKernelGGL((bd_spectrum_window_kernel<NK, NYQ, false>), dim3(grid), dim3(256), 0, st, a);
}
template <int NK>
static void bd_win_launch(bool nyq, bool whole, const bd_win_args &a, dim3 grid, hipStream_t st)
{
    if (nyq) bd_win_launch2<NK, true>(whole, a, grid, st);
    else     bd_win_launch2<NK, false>(whole, a, grid, st);
}

// ---- launcher ---------------------------------------------------------------
template <int MODE, int NK, int DEPTH>
static void launch_fast2(bool nyq, const bd_spec_args &a, dim3 grid, unsigned lds_pad, hipStream_t st)
{
    if (nyq) hipLaunchKernelGGL((bd_spectrum_band_kernel<MODE, NK, true, DEPTH>), dim3(grid), dim3(256), lds_pad, st, a);
    else     hipLaunchKernelGGL((bd_spectrum_band_kernel<MODE, NK, false, DEPTH>), dim3(grid), dim3(256), lds_pad, st, a);
}
template <int MODE, int NK>
static void launch_fast(bool nyq, int depth, const bd_spec_args &a, dim3 grid, unsigned lds_pad, hipStream_t st)
{
    if (depth == 2) launch_fast2<MODE, NK, 2>(nyq, a, grid, lds_pad, st);
    else if (depth == 4) launch_fast2<MODE, NK, 4>(nyq, a, grid, lds_pad, st);
    else launch_fast2<MODE, NK, 3>(nyq, a, grid, lds_pad, st);
}

int bd_launch_spectrum(const bd_plan &plan, const int16_t *samples, uint64_t n_streams,
                       uint64_t n_samples, uint64_t stride, uint32_t callback_len,
                       float *values, uint8_t *flags, uint64_t out_stride, hipStream_t st)
{
    const uint64_t n_cb = n_samples / callback_len;
    if (n_cb == 0 || n_streams == 0) return BD_OK;
    const bool aligned = ((uintptr_t)samples % 16 == 0) && (stride % 8 == 0);
    // 512 and 1024 samples, and the longer multiples of 512 up to 8192 (2048, 4096, ...: the ring is the
    // callback's last two half-blocks)
    const uint32_t hb_per_cb = callback_len / BD_HALF_BLOCK;
    // (and 256 / 128 samples: two / four callbacks per half-block; BD_SPEC_NO_QUARTER=1 leaves them to the granule kernel)
    const bool quarter = callback_len == 256 || callback_len == 128;
    const uint32_t cb_per_hb = quarter ? BD_HALF_BLOCK / callback_len : 1;
    const bool fast_len = (callback_len % BD_HALF_BLOCK == 0 && hb_per_cb >= 1 && hb_per_cb <= 16) ||
                          (quarter && !bd_tuning_get().spec_no_quarter);
    const bool fast_bins = plan.last_bin <= 2;
    const bd_tuning &tune = bd_tuning_get();   // experiment switches, read from the environment once
    // spec_force_granule routes 512 / 1024 to the granule kernel as well (A/B experiments)
    if (aligned && fast_len && fast_bins && !tune.spec_force_granule) {
        bd_spec_args a;
        a.samples = samples; a.values = values; a.flags = flags;
        a.stride = stride; a.n_cb = n_cb; a.out_stride = out_stride;
        const uint64_t n_hb = quarter ? (n_cb + cb_per_hb - 1) / cb_per_hb : n_cb * (callback_len / BD_HALF_BLOCK);
        if (n_hb > 0xffffffffull) { bd_set_error("stream too long for the fast Spectrum kernel"); return BD_ERR_UNSUPPORTED; }
        a.n_hb = (uint32_t)n_hb;
        // run length: long runs amortise the one re-read half-block at each run
        // start; short streams still need enough wavefronts to fill 256 CUs.
        // (Measured on 16,384 streams x 60 s with two workgroups per CU: runs of 64 half-blocks 12.4 ms,
        // of 128 12.65 ms, of 32 12.9 ms.)
        // (callbacks of m > 2 half-blocks: runs are multiples of lcm(4, m) half-blocks, 56 .. 64 at most)
        uint32_t unit = 8;
        if (hb_per_cb > 2) for (unit = hb_per_cb; unit % 4; unit += hb_per_cb) {}
        uint32_t run = unit * (64 / unit);
        while (run > unit && n_streams * ((n_hb + run - 1) / run) < 32768) run = unit * ((run / unit) / 2);
        const int run_override = tune.spec_run;
        if (run_override >= 8 && run_override % (int)unit == 0 && run_override <= (quarter ? 64 : BD_SPEC_MAX_RUN)) run = (uint32_t)run_override;
        a.hb_per_cb = hb_per_cb;
        a.run_hbs = run;
        a.runs_per_stream = (uint32_t)((n_hb + run - 1) / run);
        a.n_tasks = n_streams * a.runs_per_stream;
        a.min_amp = plan.min_amp; a.include_dc = plan.first_bin == 0;
        a.divisor = plan.divisor; a.threshold = plan.threshold;
        a.store_mode = (uint32_t)tune.spec_store;
        const uint64_t grid64 = (a.n_tasks + 3) / 4;
        dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("batch too large for one launch"); return BD_ERR_UNSUPPORTED; }
        const dim3 grid = grid_dims;
        // Unused dynamic LDS caps the workgroups resident per CU at two (eight wavefronts): the read
        // stream then comes from a quarter as many places at once, and the launch is faster than with
        // the 32 wavefronts per CU that registers and LDS would allow (16,384 streams x 60 s: 12.4 ms
        // against 12.7 - 12.9 ms, and the result stores cost less on top).  Eight wavefronts with two
        // iterations (8 KB) in flight each cover the memory latency; four do not (13.5 ms and more).
        const unsigned lds_pad = tune.spec_lds_pad >= 0 ? (unsigned)tune.spec_lds_pad : 64000u;
        const int depth = tune.spec_depth ? tune.spec_depth : 2;
        const int nk = plan.last_bin; // bins 1..last_bin
        const bool nyq = plan.dc_packs_nyquist && plan.first_bin == 0;
        if (callback_len == 256) {
            if (nk == 0) launch_fast<3, 0>(nyq, depth, a, grid, lds_pad, st);
            else if (nk == 1) launch_fast<3, 1>(nyq, depth, a, grid, lds_pad, st);
            else launch_fast<3, 2>(nyq, depth, a, grid, lds_pad, st);
        } else if (callback_len == 128) {
            // (the slots of 128-sample callbacks take 4 x 4 KB more static LDS per workgroup: the same two workgroups per CU)
            const unsigned pad_e = lds_pad > 17000u && tune.spec_lds_pad < 0 ? lds_pad - 17000u : lds_pad;
            if (nk == 0) launch_fast<4, 0>(nyq, depth, a, grid, pad_e, st);
            else if (nk == 1) launch_fast<4, 1>(nyq, depth, a, grid, pad_e, st);
            else launch_fast<4, 2>(nyq, depth, a, grid, pad_e, st);
        } else if (callback_len == 512) {
            if (nk == 0) launch_fast<0, 0>(nyq, depth, a, grid, lds_pad, st);
            else if (nk == 1) launch_fast<0, 1>(nyq, depth, a, grid, lds_pad, st);
            else launch_fast<0, 2>(nyq, depth, a, grid, lds_pad, st);
        } else if (callback_len == 1024) {
            if (nk == 0) launch_fast<1, 0>(nyq, depth, a, grid, lds_pad, st);
            else if (nk == 1) launch_fast<1, 1>(nyq, depth, a, grid, lds_pad, st);
            else launch_fast<1, 2>(nyq, depth, a, grid, lds_pad, st);
        } else {
            if (nk == 0) launch_fast<2, 0>(nyq, depth, a, grid, lds_pad, st);
            else if (nk == 1) launch_fast<2, 1>(nyq, depth, a, grid, lds_pad, st);
            else launch_fast<2, 2>(nyq, depth, a, grid, lds_pad, st);
        }
        bd_set_last_kernel("bd_spectrum_band_kernel");
        return bd_check_hip(hipGetLastError(), "launch bd_spectrum_band_kernel");
    }
    // other multiples of 64 up to 4096 (those of 512 as well when the rows are not 16-byte aligned or the band
    // holds more than three bins), of 32 up to 512, of 16 up to 256: the granule kernel
    // (BD_SPEC_NO_GRANULE=1 forces the generic one, BD_SPEC_NO_SMALL_GRANULE=1 the window kernel
    // for the lengths that need granules below 64 samples)
    // (also 512 / 1024 when the rows are not 16-byte aligned: it reads such rows with unaligned loads)
    // (more bins than three -- sampling rates below 32 kHz, down to 8 kHz -- as far as the granule
    // size allows; 512 and 1024 included then)
    // (granules below 64 samples and of 63 samples only where the split kernel does not apply -- fewer than
    // 129 samples, 768 and more, more than three bins: on 129 .. 767 it is faster than either: 144 3.8 against
    // 3.2 TB/s, 240 5.2 / 3.2, 480 6.0 / 5.0, 441 5.1 / 5.0, 504 6.1 / 5.05, 756 6.2 / 5.05 at 16,384 streams; of the multiples of 64 it takes 576, 640 and 704: 6.3 .. 6.6
    // against 6.05 .. 6.15; 192 .. 448 are faster or as fast on granules of 64; below 129 samples it takes what is
    // off the 32- and 63-sample grids: 100 3.65 TB/s (window kernel 1.0), 112 4.05 (granules of 16: 3.2), while 96
    // stays on granules of 32 at 4.6)
    const uint32_t gran = bd_spectrum_granule_size(callback_len);
    const bool split_first = (callback_len % 64 != 0 || callback_len > 512) && !(callback_len < 129 && gran >= 32) &&
                             bd_spectrum_split_applies(plan, callback_len) && !tune.spec_no_split;
    if (gran != 0 && plan.last_bin <= bd_spectrum_granule_max_bins(gran) && !split_first &&
        !tune.spec_no_granule && !(callback_len % 64 != 0 && tune.spec_no_small_granule))
        // (granules of 63 samples for 63 q: 441, 882 below the window kernel's HBM-bound range)
        return bd_launch_spectrum_granule(plan, samples, n_streams, n_samples, stride, callback_len, values, flags,
                                          out_stride, st);
    // everything else with at most 12 kept bins: the window kernel; the callbacks whose ring starts
    // before the stream does (zeros, spectrum.rs:50-51) take the generic kernel
    const bool window_path = plan.last_bin <= BD_GR_MAX_BINS && !tune.spec_no_granule;
    uint64_t n_generic = n_cb;
    if (window_path) {
        const uint64_t j_min = (BD_RING_LEN + callback_len - 1) / callback_len - 1; // callbacks with (j + 1) L < 1024
        n_generic = j_min < n_cb ? j_min : n_cb;
    }
    auto launch_generic = [&](uint64_t j_first, uint64_t count) -> int {
        bd_spec_generic_args a;
        a.samples = samples; a.values = values; a.flags = flags;
        a.stride = stride; a.n_cb = count; a.j_first = j_first; a.n_total = n_streams * count; a.out_stride = out_stride;
        a.callback_len = callback_len; a.min_amp = plan.min_amp;
        a.first_bin = plan.first_bin; a.last_bin = plan.last_bin;
        a.nyq = plan.dc_packs_nyquist; a.divisor = plan.divisor; a.threshold = plan.threshold;
        const uint64_t grid64 = (a.n_total + 3) / 4;
        dim3 grid_dims;
        if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("batch too large for one launch"); return BD_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL(bd_spectrum_generic_kernel, grid_dims, dim3(256), 0, st, a);
        bd_set_last_kernel("bd_spectrum_generic_kernel");
        return bd_check_hip(hipGetLastError(), "launch bd_spectrum_generic_kernel");
    };
    // 129 .. 767 samples off the 64-sample grid (the reference's Linux callbacks, 500 .. 540, among them): the split kernel, which
    // multiplies every sample by its twiddles once; the last callback of a stream goes to the generic kernel
    // (the split kernel's 16-byte loads may reach a few samples past a callback's end)
    if (window_path && bd_spectrum_split_applies(plan, callback_len) && !tune.spec_no_split) {
        int rc = launch_generic(n_cb - 1, 1);
        if (rc != BD_OK || n_cb == 1) return rc;
        return bd_launch_spectrum_split(plan, samples, n_streams, n_cb - 1, stride, callback_len, values, flags, out_stride, st);
    }
    if (n_generic > 0) {
        int rc = launch_generic(0, n_generic);
        if (rc != BD_OK || n_generic == n_cb) return rc;
    }
    bd_win_args a;
    a.samples = samples; a.values = values; a.flags = flags;
    a.stride = stride; a.out_stride = out_stride; a.n_cb = n_cb; a.j_min = n_generic;
    const uint64_t runs = (n_cb - n_generic + 63) / 64;
    if (runs > 0xffffffffull) { bd_set_error("stream too long"); return BD_ERR_UNSUPPORTED; }
    a.runs_per_stream = (uint32_t)runs;
    a.n_tasks = n_streams * runs;
    a.callback_len = callback_len; a.min_amp = plan.min_amp; a.include_dc = plan.first_bin == 0;
    a.divisor = plan.divisor; a.threshold = plan.threshold;
    const uint64_t grid64 = (a.n_tasks + 3) / 4;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("batch too large for one launch"); return BD_ERR_UNSUPPORTED; }
    const bool nyq = plan.dc_packs_nyquist && plan.first_bin == 0;
    const bool whole = callback_len % 16 == 0 || callback_len >= BD_RING_LEN;
    a.nk_keep = (uint32_t)plan.last_bin;
    if (plan.last_bin == 0) bd_win_launch<0>(nyq, whole, a, grid_dims, st);
    else if (plan.last_bin == 1) bd_win_launch<1>(nyq, whole, a, grid_dims, st);
    else if (plan.last_bin == 2) bd_win_launch<2>(nyq, whole, a, grid_dims, st);
    else if (plan.last_bin == 3) bd_win_launch<3>(nyq, whole, a, grid_dims, st);
    else if (plan.last_bin == 4) bd_win_launch<4>(nyq, whole, a, grid_dims, st);
    else if (plan.last_bin == 5) bd_win_launch<5>(nyq, whole, a, grid_dims, st);
    else if (plan.last_bin <= 8) bd_win_launch<8>(nyq, whole, a, grid_dims, st);
    else bd_win_launch<11>(nyq, whole, a, grid_dims, st);
    bd_set_last_kernel("bd_spectrum_window_kernel");
    return bd_check_hip(hipGetLastError(), "launch bd_spectrum_window_kernel");
}
