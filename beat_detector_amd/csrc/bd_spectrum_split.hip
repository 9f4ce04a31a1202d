// Spectrum strategy, batched, for callback lengths of 65 .. 767 samples that sit on no
// grid (the reference's author reports "mostly 500-540" samples per callback on Linux,
// src/strategies/spectrum.rs:40-42; 1024 elsewhere).
//
// The ring of callback j is the latest 1024 samples (spectrum.rs:65-68, :82-83).  With equal
// callbacks of L samples, q = 1024 / L whole callbacks fit into it and t = 1024 - q L samples remain:
//
//     ring(j) = tail_t(callback j - q)  |  callback j - q + 1  |  ...  |  callback j
//
// so, with every partial sum anchored at the first sample of its piece,
//
//     X_k(j) = T_k(j - q) + sum_{m = 1..q} w^{k (t + (m - 1) L)} S_k(j - q + m)      w = exp(2 pi i / 1024)
//     S_k(j) = sum over the whole callback,   T_k(j) = w^{-k c} (S_k(j) - H_k(j)),   c = L - t
//     H_k(j) = sum over the first c samples of the callback (its "head")
//
// (the conjugate DFT kernel: same magnitudes).  Every sample is therefore multiplied by its
// twiddles once, instead of once per ring that contains it (about 1024 / L times in a kernel that
// evaluates each ring on its own).  Callbacks before the stream contribute zeros (spectrum.rs:50-51).
//
// A row of 16 lanes owns one callback per iteration (four callbacks per wavefront and
// iteration); lane l holds the samples 128 p + 8 l + m of it (p < 6, m < 8), so every load
// instruction covers 256 contiguous bytes per row (16-byte loads at 2-byte-aligned addresses),
// 768 samples in all.  Twiddles are compile-time constants w^{k (128 p + m)} times one rotation by the
// lane's w^{8 k l}; DPP row reductions.  Samples of a lane that lie past the callback are removed
// by per-lane masks computed once.  The head is a prefix of the callback: the accumulators as they
// stand after the pieces that lie entirely inside it (a compile-time piece index, a lane
// threshold) give a second, head-only row sum; the one 16-byte piece that straddles the cut is
// parked in LDS as it is, and finished at the end of the run, when lane i turns callback i's sums
// into value and decision and adds that piece's partial sum itself.  (Reading the piece again from
// memory at that point costs a whole cache line per callback: the streaming loads leave nothing in
// L2.  Measured on 529-sample callbacks: 31 % more HBM traffic than the samples themselves.)
// The amplitude gate (spectrum.rs:77-80) is the row maximum: the callback's own samples.
#include "bd_internal.h"

#include "bd_twiddles.inc"  // BD_LANE_C / BD_LANE_S [k - 1][l] = cos / sin (2 pi k 8 l / 1024)
#include "bd_split_tw.inc" // BD_SP_C / BD_SP_S [k - 1][8 p + m] = cos / sin (2 pi k (128 p + m) / 1024)

typedef short bd_short2 __attribute__((ext_vector_type(2)));

static __device__ const float bd_cos1024[1024] = {
#include "bd_cos1024.inc"
};
static __device__ const float bd_sin1024[1024] = {
#include "bd_sin1024.inc"
};

#ifndef BD_SPLIT_RUN
#define BD_SPLIT_RUN 64   // callbacks per wavefront task; four more LDS slots hold the callbacks before it
#endif
#ifndef BD_SPLIT_REM_MAX
#define BD_SPLIT_REM_MAX 24 // remainders up to this many samples are summed at the end of the run (multiple of 8)
#endif
// Wavefronts (tasks) per block.  The staged kernels that also park remainders hold 13.8 KB of LDS per
// wavefront: in blocks of four, two blocks fit a CU (eight wavefronts); one by one, eleven do
// (measured on 529-sample callbacks: 4.9 -> 5.3 TB/s).  The others lose a little in small blocks.
__host__ __device__ constexpr int bd_split_waves(bool staged, bool rem) { return staged && rem ? 1 : 4; }

struct bd_split_args {
    const int16_t *samples;
    float         *values;
    uint8_t       *flags;
    uint64_t       stride, out_stride;
    uint64_t       n_cb;            // callbacks per stream that this kernel evaluates: 0 .. n_cb - 1
    uint64_t       n_tasks;
    uint32_t       runs_per_stream, callback_len;
    uint32_t       subruns;         // runs of BD_SPLIT_RUN callbacks that one wavefront walks, one after the other
    uint32_t       q, head_len, tail_len;
    uint32_t       min_amp, include_dc;
    float          divisor, threshold;
};

struct __attribute__((aligned(16))) bd_split_slot {
    float re1, im1, re2, im2;       // S_k: the whole callback
    float hre1, him1, hre2, him2;   // the pieces entirely inside the head
    int   dc, hdc;
    float amax, pad;
};

struct bd_split_acc {
    float a1, b1, a2, b2;
    int   dc;
};

__device__ __forceinline__ void bd_rot(float &re, float &im, int idx)
{
    const float c = bd_cos1024[idx & (BD_RING_LEN - 1)], s = bd_sin1024[idx & (BD_RING_LEN - 1)];
    const float r = re * c - im * s, i = re * s + im * c;
    re = r; im = i;
}

typedef unsigned int bd_u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));

// One dword (two samples) into the running sums; TW = index of the first sample's twiddle in BD_SP_*.
template <int NK, int TW>
__device__ __forceinline__ void bd_split_dword(int v, bd_split_acc &x, float &amax)
{
    const float x0 = (float)(short)(v & 0xffff);
    const float x1 = (float)(short)(v >> 16);
    x.dc = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_short2, v), bd_short2{1, 1}, x.dc, false);
    amax = fmaxf(fmaxf(amax, fabsf(x0)), fabsf(x1));
    if (NK >= 1) {
        x.a1 = __builtin_fmaf(x0, BD_SP_C[0][TW], x.a1); x.b1 = __builtin_fmaf(x0, BD_SP_S[0][TW], x.b1);
        x.a1 = __builtin_fmaf(x1, BD_SP_C[0][TW + 1], x.a1); x.b1 = __builtin_fmaf(x1, BD_SP_S[0][TW + 1], x.b1);
    }
    if (NK >= 2) {
        x.a2 = __builtin_fmaf(x0, BD_SP_C[1][TW], x.a2); x.b2 = __builtin_fmaf(x0, BD_SP_S[1][TW], x.b2);
        x.a2 = __builtin_fmaf(x1, BD_SP_C[1][TW + 1], x.a2); x.b2 = __builtin_fmaf(x1, BD_SP_S[1][TW + 1], x.b2);
    }
}

// The first `m` (< 8 * N8) samples of the 16-byte units at `units` as a sum anchored at the first
// sample; twiddles w^{k i}.
template <int NK, int N8>
__device__ __forceinline__ void bd_split_partial(const uint4 *units, int m, bd_split_acc &x, float &amax)
{
#pragma unroll
    for (int u = 0; u < N8; u++) {
        if (8 * u >= m) break;          // wave-uniform
        const uint4 v4 = units[u];
        const unsigned w[4] = {v4.x, v4.y, v4.z, v4.w};
        bd_split_acc y;                 // this unit, anchored at its own first sample
        y.a1 = y.b1 = y.a2 = y.b2 = 0.f; y.dc = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int i = 8 * u + 2 * d;
            const unsigned keep = (i + 1 < m) ? 0xffffffffu : ((i < m) ? 0x0000ffffu : 0u);
            const int v = (int)(w[d] & keep);
            if (d == 0) bd_split_dword<NK, 0>(v, y, amax);
            else if (d == 1) bd_split_dword<NK, 2>(v, y, amax);
            else if (d == 2) bd_split_dword<NK, 4>(v, y, amax);
            else bd_split_dword<NK, 6>(v, y, amax);
        }
        bd_rot(y.a1, y.b1, 1 * 8 * u);
        bd_rot(y.a2, y.b2, 2 * 8 * u);
        x.a1 += y.a1; x.b1 += y.b1; x.a2 += y.a2; x.b2 += y.b2; x.dc += y.dc;
    }
}

// Twiddles of sample GROUP p + m of a callback (without the lane's offset 8 l), entry 8 p + m.
template <int LPC> struct bd_split_tw;
template <> struct bd_split_tw<16> {
    static __device__ __forceinline__ constexpr float C(int k, int i) { return BD_SP_C[k][i]; }
    static __device__ __forceinline__ constexpr float S(int k, int i) { return BD_SP_S[k][i]; }
};
template <> struct bd_split_tw<8> {
    static __device__ __forceinline__ constexpr float C(int k, int i) { return BD_SP8_C[k][i]; }
    static __device__ __forceinline__ constexpr float S(int k, int i) { return BD_SP8_S[k][i]; }
};
template <> struct bd_split_tw<4> {
    static __device__ __forceinline__ constexpr float C(int k, int i) { return BD_SP4_C[k][i]; }
    static __device__ __forceinline__ constexpr float S(int k, int i) { return BD_SP4_S[k][i]; }
};

// Sum / maximum over the LPC lanes of a callback, valid in all of them.  Groups of 4 lanes: partners
// l ^ 1, l ^ 2; of 8 lanes: then the mirror inside the half-row (the other quad's sum).
template <int LPC> __device__ __forceinline__ float bd_group_sum(float v)
{
    if constexpr (LPC == 16) return bd_row_sum(v);
    v += bd_dpp_f<0xB1>(v); v += bd_dpp_f<0x4E>(v);
    if constexpr (LPC == 8) v += bd_dpp_f<0x141>(v);
    return v;
}
template <int LPC> __device__ __forceinline__ int bd_group_sum(int v)
{
    if constexpr (LPC == 16) return bd_row_sum(v);
    v += bd_dpp_i<0xB1>(v); v += bd_dpp_i<0x4E>(v);
    if constexpr (LPC == 8) v += bd_dpp_i<0x141>(v);
    return v;
}
template <int LPC> __device__ __forceinline__ float bd_group_max(float v)
{
    if constexpr (LPC == 16) {
        return bd_row_max(v);
    } else {
        v = fmaxf(v, bd_dpp_f<0xB1>(v)); v = fmaxf(v, bd_dpp_f<0x4E>(v));
        if constexpr (LPC == 8) v = fmaxf(v, bd_dpp_f<0x141>(v));
        return v;
    }
}

// Row reductions of the sixteen-lane kernel, all at once.  A DPP instruction writes only the quads of a row that its
// bank mask names, so one register can carry the partial sums of several quantities side by side:
//   level 1 (lane l + lane l ^ 8): quantity A in the lower half of the row, B in the upper half     2 instructions per pair
//   level 2 (lane l + lane 7 - l of its half): pairs (A, B) and (C, D) into one register, A C B D by quad     2 per four
//   levels 3, 4 (inside the quad): 2 per register, whatever it carries
// Eight float sums cost 16 instructions instead of 32, the two integer sums 5 instead of 8, the maximum 4.
// Results: s1 = sums of (a, c, b, d) in quads (0, 1, 2, 3) of the row, s2 the same of (e, g, f, h); t = sum of ia in the
// row's lower half, of ib in its upper half; mx = maximum of m, in every lane.
// (One assembly statement: the compiler does not give DPP instructions a bank mask.  A DPP operand must have been
// written at least two instructions earlier: the order below keeps that distance, the s_nop covers the inputs.)
__device__ __forceinline__ void bd_split_reduce16(float a, float b, float c, float d, float e, float f, float g, float h,
                                                  int ia, int ib, float m, float &s1, float &s2, int &t, float &mx)
{
    float r1, r2, r3, r4;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %[r1], %[a], %[a] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[r1], %[b], %[b] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r2], %[c], %[c] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[r2], %[d], %[d] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r3], %[e], %[e] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[r3], %[f], %[f] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r4], %[g], %[g] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[r4], %[h], %[h] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_u32_dpp %[t], %[ia], %[ia] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_u32_dpp %[t], %[ib], %[ib] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_max_f32_dpp %[mx], %[m], %[m] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s1], %[r1], %[r1] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[s2], %[r3], %[r3] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[s1], %[r2], %[r2] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[s2], %[r4], %[r4] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_u32_dpp %[t], %[t], %[t] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %[mx], %[mx], %[mx] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s1], %[s1], %[s1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s2], %[s2], %[s2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %[t], %[t], %[t] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %[mx], %[mx], %[mx] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s1], %[s1], %[s1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s2], %[s2], %[s2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %[mx], %[mx], %[mx] row_ror:1 row_mask:0xf bank_mask:0xf"
        : [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [s1] "=&v"(s1), [s2] "=&v"(s2), [t] "=&v"(t),
          [mx] "=&v"(mx)
        : [a] "v"(a), [b] "v"(b), [c] "v"(c), [d] "v"(d), [e] "v"(e), [f] "v"(f), [g] "v"(g), [h] "v"(h), [ia] "v"(ia),
          [ib] "v"(ib), [m] "v"(m));
}

// PSTAR: index of the group of GROUP samples that holds the end of the head (head_len / GROUP).
// PL: index of the group that holds the end of the callback (callback_len / GROUP): the groups
// before it are whole in every lane, the groups after it are not read at all.
// REM: the samples of group PL (the callback's remainder, at most BD_SPLIT_REM_MAX) stay out of
// the main loop, which then has no masks and one piece less per lane: the first lanes of a row park
// the remainder's 16-byte units in LDS, and lane i adds the remainder of callback i from there at the
// end of the run.  Otherwise group PL is one more, masked piece.
// STAGED: callback lengths that are not multiples of four samples put most rows at addresses that
// are not 8-byte aligned, where 16-byte global loads run at 80 % of their aligned rate (measured:
// 4.2 TB/s for odd lengths, 4.7 for 4 k + 2, 5.2 for multiples of four).  Such lengths copy the
// iteration's four callbacks to LDS as they lie in memory (aligned, contiguous 16-byte units),
// and the lanes pick their pieces from there: a 16-byte LDS read at the dword below the piece,
// one more dword, and a funnel shift by 16 bits in the rows that start on an odd sample.
// LPC: lanes per callback.  16 (groups of 128 samples, four callbacks per iteration) for callbacks of 385
// samples and more; 8 (groups of 64 samples, eight callbacks per iteration) for 129 .. 383 samples, whose
// ring spans up to seven whole callbacks; 4 (groups of 32 samples, sixteen callbacks per iteration) for
// 65 .. 127 samples: up to fifteen whole callbacks.
// SH: rows are read on the grid of 16-byte lines (see the comment at `lane_set` below).
// MULTI: a wavefront walks a.subruns runs (else one: the loop over runs costs registers even when it runs once).
template <int NK, int PSTAR, int PL, bool STAGED, bool REM, int LPC = 16, bool SH = !STAGED, bool MULTI = STAGED>
__global__ __launch_bounds__(64 * bd_split_waves(STAGED, REM)) void bd_spectrum_split_kernel(bd_split_args a)
{
    constexpr int WAVES = bd_split_waves(STAGED, REM);
    constexpr int GROUP = 8 * LPC;                    // samples of a callback that one load instruction covers
    constexpr int ROWS = 64 / LPC;                    // callbacks per iteration
    constexpr int PIECES = REM ? PL : PL + 1;
    // 64 x LOADS units of 16 bytes hold ROWS callbacks of up to GROUP (PL + 1) - 1 samples from any 2-byte offset
    constexpr int LOADS = ((ROWS * (GROUP * (PL + 1) - 1) * 2 + 15 + 15) / 16 + 1 + 63) / 64;
    constexpr int REM_UNITS = BD_SPLIT_REM_MAX / 8;
    constexpr int NBUF = STAGED ? LOADS : (REM ? PIECES + 1 : PIECES);
    constexpr int KEEP = REM ? 1 + REM_UNITS : 1;     // parked per callback: the head's last piece, the remainder
    static_assert(PSTAR < PIECES, "the head ends inside a group of the main loop");
    __shared__ bd_split_slot lds_slots[WAVES][BD_SPLIT_RUN + ROWS];
    __shared__ __attribute__((aligned(16))) uint4 lds_keep[WAVES][BD_SPLIT_RUN + ROWS][KEEP];
    __shared__ __attribute__((aligned(16))) uint4 lds_stage[STAGED ? WAVES : 1][STAGED ? 64 * LOADS + 2 : 1];

    const int lane_id = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t task = bd_block_index() * WAVES + wave;
    if (task >= a.n_tasks) return;
    const uint64_t s = task / a.runs_per_stream;
    // A wavefront walks `subruns` consecutive runs of BD_SPLIT_RUN callbacks.  Only the first of them reads the
    // q callbacks before it again (one extra iteration); the others find those callbacks' sums where
    // the run before them left them, moved to the front of the LDS slots.
    const int64_t j_task = (int64_t)(task - s * a.runs_per_stream) * BD_SPLIT_RUN * (int64_t)a.subruns;
    const int16_t *sp = a.samples + s * a.stride;
    const int64_t L = a.callback_len;
    bd_split_slot *slots = lds_slots[wave];
    uint4 (*keep)[KEEP] = lds_keep[wave];
    const int q_cb = (int)a.q;

    // MULTI: the wavefront walks several runs.  Everything a lane derives from its number is then derived anew
    // in every run (the lane number is made opaque at the top of the loop): kept across the loop, those values
    // stay in registers through every run's finish pass and cost the kernel a wavefront per SIMD.
    for (int sub = 0; sub < (MULTI ? (int)a.subruns : 1); sub++) {
    int lane = lane_id;
    if (MULTI) asm volatile("" : "+v"(lane));
    const int g = lane / LPC, l = lane % LPC;

    // ---- what a lane's place inside its callback fixes, once ----
    // SH: a row is read from the 16-byte line that holds its first sample -- of memory (the kernels that do
    // not stage: all global loads aligned, whatever the callback length) or of the staged copy (aligned
    // 16-byte LDS reads, no funnel shift) -- so the row starts `s` samples (0 .. 7) into the lane grid: lane
    // l, piece p holds the samples 128 p + 8 l - s + m.  The s samples before the row (the previous
    // callback's last ones, in lane 0's first piece) are masked out, the shift goes into the lane's
    // rotation w^{k (8 l - s)}, and the cut at the end of the head and the end of the callback move by s
    // lane positions.  Rows four callbacks apart are 8 L bytes apart: s is the same in every iteration for
    // even L and alternates between s and s ^ 4 for odd L, so a lane needs two sets of constants, one per
    // register buffer (each buffer serves every other iteration).
    constexpr bool ALD = !STAGED;
    static_assert(!ALD || (LPC == 16 && SH), "the narrow variants stage; unstaged rows are read on the line grid");
    static_assert(!SH || LPC == 16, "the line grid is for sixteen lanes per callback");
    const int c = (int)a.head_len;
    struct lane_set {
        float    lc1, ls1, lc2, ls2;   // w^{k (8 l - s)}
        unsigned first[4];             // keep-masks of the first piece's dwords (SH: samples before the row)
        unsigned mask[4];              // keep-masks of the masked last piece's dwords
        int      off_last;             // where that piece is read (a piece past the callback: the lane's first piece)
        unsigned below_mask;           // all ones if this lane's piece of group PSTAR lies entirely inside the head
        bool     at_cut;               // this lane's piece of group PSTAR holds the cut
    };
    auto make_set = [&](int sh) {
        lane_set k;
        if (SH) {
            const int i1 = (8 * l - sh) & (BD_RING_LEN - 1), i2 = (2 * (8 * l - sh)) & (BD_RING_LEN - 1);
            k.lc1 = bd_cos1024[i1]; k.ls1 = bd_sin1024[i1];
            k.lc2 = bd_cos1024[i2]; k.ls2 = bd_sin1024[i2];
        } else {
            k.lc1 = BD_LANE_C[0][l]; k.ls1 = BD_LANE_S[0][l];
            k.lc2 = BD_LANE_C[1][l]; k.ls2 = BD_LANE_S[1][l];
        }
        // dword d of the masked piece holds grid positions GROUP PL + 8 l + 2 d, + 1: samples at those minus sh
        const int o_last = GROUP * PL + 8 * l;
        k.off_last = o_last - sh < (int)L ? o_last : 8 * l;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            k.mask[d] = (o_last + 2 * d + 1 - sh < (int)L) ? 0xffffffffu : ((o_last + 2 * d - sh < (int)L) ? 0x0000ffffu : 0u);
            k.first[d] = (8 * l + 2 * d >= sh) ? 0xffffffffu : ((8 * l + 2 * d + 1 >= sh) ? 0xffff0000u : 0u);
        }
        // the head ends inside group PSTAR: there the lanes below `lstar` lie entirely inside it
        const int lstar = (c + sh - GROUP * PSTAR) >> 3;
        k.below_mask = l < lstar ? 0xffffffffu : 0u;
        k.at_cut = l == lstar;
        // (the masks are made opaque: the compiler would otherwise recompute them from their
        // comparisons inside the loop, as v_cndmask_b32)
#pragma unroll
        for (int d = 0; d < 4; d++) asm volatile("" : "+v"(k.mask[d]), "+v"(k.first[d]));
        asm volatile("" : "+v"(k.below_mask));
        return k;
    };
    // shift of callback j's row (samples): where its first sample sits in its 16-byte line
    auto shift_of = [&](int64_t j) -> int {
        return SH ? (int)((((uintptr_t)sp + (uintptr_t)(j * L * 2)) & 15) >> 1) : 0;
    };
    // K0: this lane's rows in the even iterations (callbacks j + g, j + 2 ROWS + g, ... of a run that starts at
    // callback j), K1: in the odd ones and in the iteration before the run (-1).  A run is 128 L bytes long, a
    // multiple of 16: the sets serve every run of the task.
    const lane_set K0 = make_set(shift_of(j_task + g)), K1 = make_set(shift_of(j_task + ROWS + g));
    const int o_last = GROUP * PL + 8 * l;   // (staged kernels: grid position of the lane's last piece)
    const int off_last = K0.off_last;

    const int64_t j0 = j_task + (int64_t)sub * BD_SPLIT_RUN;
    const int64_t left = (int64_t)a.n_cb - j0;
    if (left <= 0) break;
    const int n_win = left < BD_SPLIT_RUN ? (int)left : BD_SPLIT_RUN;
    const int n_it = (n_win + ROWS - 1) / ROWS;
    const bool warm = sub == 0 && j0 > 0;   // the q callbacks before the run (q <= ROWS) are read again; none before the stream
    const int it_begin = warm ? -1 : 0;
    const int64_t j_last = j0 + n_win - 1;
    // The staged copy counts bytes from `run_ptr`, the 16-byte line that holds the first sample this
    // run reads (callback `cb0`): 32-bit arithmetic inside the loop, and every load address stays an
    // offset from the kernel's argument (global loads, which return in order and leave the loads of
    // the next iteration in flight while this one's are awaited).  It never reaches past the stream's
    // last callback (index n_cb, not evaluated here): `limit_rel`.
    const int64_t cb0 = it_begin < 0 ? j0 - q_cb : j0;
    const unsigned L2 = 2u * (unsigned)L;
    const unsigned m0 = (unsigned)(((int64_t)((uintptr_t)sp & 15) + cb0 * L * 2) & 15);
    const char *run_ptr = reinterpret_cast<const char *>(sp) + (cb0 * L * 2 - (int64_t)m0);
    const int64_t limit64 = (int64_t)m0 + ((int64_t)a.n_cb + 1 - cb0) * L * 2;
    const unsigned limit_rel = limit64 < (1 << 24) ? (unsigned)limit64 : (1u << 24);
    const int first_rel = (int)(j0 - cb0), last_rel = (int)(j_last - cb0);   // the run's callbacks, counted from cb0
    uint4 A[NBUF];
    uint4 B[STAGED ? 1 : NBUF];
    const int64_t run_base = (int64_t)(s * a.stride) + j0 * L;   // the run's first sample, counted from a.samples
    const int run_lo = (int)(((((uintptr_t)a.samples & 15) >> 1) + (uint64_t)run_base) & 7);   // its place in its 16-byte line
    auto load = [&](uint4 (&d)[NBUF], int it) {
        if (STAGED) {
            // callbacks jb .. jb + ROWS - 1 (those up to j_last) as 16-byte units from the unit that holds
            // the first sample of callback jb
            // (before the run: only the q callbacks that its first rings reach into)
            const int jb = it < 0 ? 0 : first_rel + ROWS * it;
            const int je = it < 0 ? first_rel - 1 : (jb + ROWS - 1 < last_rel ? jb + ROWS - 1 : last_rel);
            const unsigned base = (m0 + (unsigned)jb * L2) & ~15u;
            unsigned end = m0 + (unsigned)(je + 1) * L2;
            end = end < limit_rel ? end : limit_rel;
            const unsigned n_units = (end - base + 15) >> 4;
            const unsigned whole = (limit_rel - base) >> 4;         // units that end at or before the limit
            // Through a buffer descriptor over exactly these units: the loads past the last one are dropped by the
            // range check (no memory access, zeros that nobody reads), every load is unconditional -- a load
            // under a lane predicate would turn the wait for this buffer into a wait for every load in flight,
            // the next iteration's included -- and no per-lane address arithmetic is left: the descriptor moves
            // (scalar registers: everything it is made of is the same in all lanes), the lane's offset 16 * lane
            // never changes.
            const uint64_t ubase = (uint64_t)(uintptr_t)run_ptr + base;
            const uint32_t b_lo = __builtin_amdgcn_readfirstlane((uint32_t)ubase);
            const uint32_t b_hi = __builtin_amdgcn_readfirstlane((uint32_t)(ubase >> 32));
            const uint32_t n_rec = __builtin_amdgcn_readfirstlane(16u * (n_units < whole ? n_units : whole));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void *>((uintptr_t)(((uint64_t)b_hi << 32) | b_lo)), 0, (int)n_rec, 0x00020000);
#pragma unroll
            for (int i = 0; i < LOADS; i++) {
                const bd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, 16u * (unsigned)lane, 1024 * i, 2 /* nt */);
                d[i] = make_uint4(v.x, v.y, v.z, v.w);
            }
        } else {
            // the row, counted from the run's first callback (32-bit arithmetic per lane)
            int jr = ROWS * it + g;
            jr = jr < n_win - 1 ? jr : n_win - 1;               // stand-in rows of the last iteration
            // before the run: only the q callbacks that its first rings reach into; the rows before them
            // repeat the first of those (no load sits under a lane predicate: the wait for this buffer
            // would otherwise have to be a wait for every load in flight)
            jr = jr < -q_cb ? -q_cb : jr;
            // the 16-byte line that holds the row's first sample; lane positions count from there
            // (the address stays an offset from the kernel's argument: global loads, not flat ones)
            const int rel = jr * (int)L;                                        // samples from the run's first callback
            const int back = (run_lo + rel) & 7;                                // the row's shift, in samples
            const int16_t *cp = a.samples + run_base + (rel - back);
            const int ol = it & 1 ? K1.off_last : K0.off_last;
#pragma unroll
            for (int p = 0; p < PIECES; p++)
                d[p] = bd_load_stream16(cp + (p < PL ? GROUP * p + 8 * l : ol));
            if (REM) {
                // the remainder's units in the row's first lanes (they may reach into the next callback,
                // which exists); the other lanes repeat the last of them
                const int lr = l < REM_UNITS ? l : REM_UNITS - 1;
                d[NBUF - 1] = bd_load_stream16(cp + GROUP * PL + 8 * lr);
            }
        }
    };
    using TW = bd_split_tw<LPC>;
    auto compute = [&](const auto &d, int it, const lane_set &K) {
        const float lc1 = K.lc1, ls1 = K.ls1, lc2 = K.lc2, ls2 = K.ls2;
        bd_split_acc x;
        x.a1 = x.b1 = x.a2 = x.b2 = 0.f; x.dc = 0;
        bd_split_acc h_before = x, h_after = x;   // accumulators before / after the pieces of group PSTAR
        float amax = 0.f;
#pragma unroll
        for (int p = 0; p < PIECES; p++) {
            if (p == PSTAR) h_before = x;
            const unsigned w[4] = {d[p].x, d[p].y, d[p].z, d[p].w};
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const int v = (int)(p < PL ? ((SH && p == 0) ? (w[qq] & K.first[qq]) : w[qq]) : (w[qq] & K.mask[qq]));
                const float x0 = (float)(short)(v & 0xffff);
                const float x1 = (float)(short)(v >> 16);
                const int i0 = 8 * p + 2 * qq, i1 = i0 + 1;   // table entry of sample GROUP p + 2 qq (+ 1)
                x.dc = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_short2, v), bd_short2{1, 1}, x.dc, false);
                amax = fmaxf(fmaxf(amax, fabsf(x0)), fabsf(x1));
                if (NK >= 1) {
                    x.a1 = __builtin_fmaf(x0, TW::C(0, i0), x.a1); x.b1 = __builtin_fmaf(x0, TW::S(0, i0), x.b1);
                    x.a1 = __builtin_fmaf(x1, TW::C(0, i1), x.a1); x.b1 = __builtin_fmaf(x1, TW::S(0, i1), x.b1);
                }
                if (NK >= 2) {
                    x.a2 = __builtin_fmaf(x0, TW::C(1, i0), x.a2); x.b2 = __builtin_fmaf(x0, TW::S(1, i0), x.b2);
                    x.a2 = __builtin_fmaf(x1, TW::C(1, i1), x.a2); x.b2 = __builtin_fmaf(x1, TW::S(1, i1), x.b2);
                }
            }
            if (p == PSTAR) h_after = x;
        }
        // the pieces of this lane that lie entirely inside the head
        // (selected bit by bit with the lane's mask: one v_bfi_b32 each; a v_cndmask_b32 occupies the SIMD
        // several times as long)
        bd_split_acc h;
        const unsigned bm = K.below_mask;
        auto pick = [bm](float x, float y) {   // x where the lane is below the cut, else y
            return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, x) & bm) | (__builtin_bit_cast(unsigned, y) & ~bm));
        };
        h.a1 = pick(h_after.a1, h_before.a1); h.b1 = pick(h_after.b1, h_before.b1);
        h.a2 = pick(h_after.a2, h_before.a2); h.b2 = pick(h_after.b2, h_before.b2);
        h.dc = (int)(((unsigned)h_after.dc & bm) | ((unsigned)h_before.dc & ~bm));
        // (a + i b) w^{8 k l}: sums anchored at the callback's first sample
        const int slot = ROWS * it + g + ROWS;   // callback j0 - ROWS + slot
        if constexpr (LPC == 16 && NK == 2) {
            // All eleven row reductions in one network (bd_split_reduce16): 25 lane-crossing instructions
            // instead of 44.  The sums come out spread over the row's quads; the first lane of each quad
            // stores its share of the slot.
            float s1, s2, mx;
            int t;
            bd_split_reduce16(x.a1 * lc1 - x.b1 * ls1, x.a1 * ls1 + x.b1 * lc1, x.a2 * lc2 - x.b2 * ls2, x.a2 * ls2 + x.b2 * lc2,
                              h.a1 * lc1 - h.b1 * ls1, h.a1 * ls1 + h.b1 * lc1, h.a2 * lc2 - h.b2 * ls2, h.a2 * ls2 + h.b2 * lc2,
                              x.dc, h.dc, amax, s1, s2, t, mx);
            if ((l & 3) == 0) {
                const int qd = l >> 2;                       // quad 0: re1 / hre1, 1: re2 / hre2, 2: im1 / him1, 3: im2 / him2
                const int f = (qd & 1) * 2 + (qd >> 1);
                float *sf = reinterpret_cast<float *>(&slots[slot]);
                sf[f] = s1;
                sf[4 + f] = s2;
                reinterpret_cast<int *>(sf)[8 + (qd >> 1)] = t;   // lower half of the row: dc, upper half: hdc
                sf[10] = mx;
            }
        } else {
        bd_split_slot r;
        r.re1 = r.im1 = r.re2 = r.im2 = r.hre1 = r.him1 = r.hre2 = r.him2 = 0.f;
        if (NK >= 1) {
            r.re1 = bd_group_sum<LPC>(x.a1 * lc1 - x.b1 * ls1); r.im1 = bd_group_sum<LPC>(x.a1 * ls1 + x.b1 * lc1);
            r.hre1 = bd_group_sum<LPC>(h.a1 * lc1 - h.b1 * ls1); r.him1 = bd_group_sum<LPC>(h.a1 * ls1 + h.b1 * lc1);
        }
        if (NK >= 2) {
            r.re2 = bd_group_sum<LPC>(x.a2 * lc2 - x.b2 * ls2); r.im2 = bd_group_sum<LPC>(x.a2 * ls2 + x.b2 * lc2);
            r.hre2 = bd_group_sum<LPC>(h.a2 * lc2 - h.b2 * ls2); r.him2 = bd_group_sum<LPC>(h.a2 * ls2 + h.b2 * lc2);
        }
        r.dc = bd_group_sum<LPC>(x.dc);
        r.hdc = bd_group_sum<LPC>(h.dc);
        r.amax = bd_group_max<LPC>(amax);
        r.pad = 0.f;
        // (The empty statements keep the sums computed in every lane: moved into the branch of the one
        // lane that stores them, the last step of each reduction is a lane move plus an add instead
        // of one add that reads its operand through the lane rotation.)
        if (NK >= 1) asm volatile("" :: "v"(r.re1), "v"(r.im1), "v"(r.hre1), "v"(r.him1));
        if (NK >= 2) asm volatile("" :: "v"(r.re2), "v"(r.im2), "v"(r.hre2), "v"(r.him2));
        asm volatile("" :: "v"(r.dc), "v"(r.hdc), "v"(r.amax));
        if (l == 0) slots[slot] = r;
        }
        if (K.at_cut) keep[slot][0] = d[PSTAR];       // the piece that straddles the end of the head
    };
    // (the scheduling barriers keep the next iteration's loads ahead of this iteration's arithmetic)
    if constexpr (STAGED) {
        // One iteration ahead in LDS, two in memory: while the sums of iteration `it` are formed from
        // pieces in registers, the pieces of it + 1 are on their way from the staged copy and the units
        // of it + 2 from memory.
        constexpr int NPC = REM ? PIECES + 1 : PIECES;
        uint4 *stage = lds_stage[wave];
        auto put = [&](const uint4 (&d)[NBUF]) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < LOADS; i++) stage[lane + 64 * i] = d[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        struct raw_piece { uint4 v; unsigned e, sh; };    // five dwords from the one below the piece; the shift
        auto pick = [&](raw_piece (&pc)[NPC], int it) {
            const int jb = it < 0 ? 0 : first_rel + ROWS * it;
            int jj = first_rel + ROWS * it + g;
            jj = jj < last_rel ? jj : last_rel;
            jj = jj < jb ? jb : jj;                 // rows before the run that no ring reaches: stand-ins
            // byte offset of the row's first sample inside the staged copy
            const unsigned row_off = ((m0 + (unsigned)jb * L2) & 15) + (unsigned)(jj - jb) * L2;
            const unsigned sh = (row_off & 2) ? 16u : 0u;
            // SH: whole 16-byte units of the copy from the one that holds the row's first sample (the copy
            // starts on a 16-byte line of memory); otherwise five dwords from the one below the piece
            const char *sb = reinterpret_cast<const char *>(stage) + (row_off & (SH ? ~15u : ~3u));
            const int ol = it & 1 ? K1.off_last : K0.off_last;
#pragma unroll
            for (int p = 0; p < NPC; p++) {
                if (p == PIECES && l >= REM_UNITS) break;   // the remainder's units: the row's first lanes
                const unsigned po = (unsigned)(2 * (p < PL ? GROUP * p + 8 * l : (REM ? o_last : (SH ? ol : off_last))));
                if (SH) {
                    pc[p].v = *reinterpret_cast<const uint4 *>(sb + po);
                } else {
                    typedef unsigned int u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
                    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(sb + po);
                    pc[p].v = make_uint4(v.x, v.y, v.z, v.w);
                    pc[p].e = *reinterpret_cast<const unsigned *>(sb + po + 16);
                }
            }
            pc[0].sh = sh;
        };
        auto sums = [&](const raw_piece (&raw)[NPC], int it, const lane_set &K) {
            const unsigned sh = raw[0].sh;
            uint4 pc[NPC];
#pragma unroll
            for (int p = 0; p < NPC; p++) {
                if (SH) { pc[p] = raw[p].v; continue; }
                pc[p].x = __builtin_amdgcn_alignbit(raw[p].v.y, raw[p].v.x, sh);
                pc[p].y = __builtin_amdgcn_alignbit(raw[p].v.z, raw[p].v.y, sh);
                pc[p].z = __builtin_amdgcn_alignbit(raw[p].v.w, raw[p].v.z, sh);
                pc[p].w = __builtin_amdgcn_alignbit(raw[p].e, raw[p].v.w, sh);
            }
            compute(pc, it, K);
            if (REM && l < REM_UNITS) keep[ROWS * it + g + ROWS][1 + l] = pc[NPC - 1];
        };
        // P, Q: pieces of the iteration at hand and of the next one, as read from the staged copy
        raw_piece P[NPC], Q[NPC];
        auto advance = [&](raw_piece (&next)[NPC], int it) {   // the staged copy moves on to iteration it + 1
            if (it + 1 < n_it) {
                put(A);
                if (it + 2 < n_it) load(A, it + 2);
                pick(next, it + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (warm) {   // the iteration before the run, on its own
            load(A, -1);
            put(A);
            pick(P, -1);
            sums(P, -1, K1);
        }
        int it = 0;
        load(A, it);
        put(A);
        if (it + 1 < n_it) load(A, it + 1);
        pick(P, it);
        for (;;) {
            advance(Q, it);
            sums(P, it, K0);
            if (++it >= n_it) break;
            advance(P, it);
            sums(Q, it, K1);
            if (++it >= n_it) break;
        }
    } else {
        auto run = [&](const uint4 (&d)[NBUF], int it, const lane_set &K) {
            compute(d, it, K);
            if (REM && l < REM_UNITS) keep[ROWS * it + g + ROWS][1 + l] = d[NBUF - 1];
        };
        if (warm) {   // the iteration before the run, on its own
            load(B, -1);
            load(A, 0);
            __builtin_amdgcn_sched_barrier(0);
            run(B, -1, K1);
        } else {
            load(A, 0);
        }
        int it = 0;
        for (; it + 2 < n_it; it += 2) {
            load(B, it + 1);
            __builtin_amdgcn_sched_barrier(0);
            run(A, it, K0);
            load(A, it + 2);
            __builtin_amdgcn_sched_barrier(0);
            run(B, it + 1, K1);
        }
        if (it + 1 < n_it) {
            load(B, it + 1);
            __builtin_amdgcn_sched_barrier(0);
            run(A, it, K0);
            run(B, it + 1, K1);
        } else {
            run(A, it, K0);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- end of the run: slot i holds callback j0 - ROWS + i ----
    if (REM) {
        // the callbacks' remainders, samples GROUP PL .. L - 1, from the parked units
        for (int sidx = lane + (it_begin < 0 ? 0 : ROWS); sidx < n_win + ROWS; sidx += 64) {
            bd_split_acc y;
            y.a1 = y.b1 = y.a2 = y.b2 = 0.f; y.dc = 0;
            float ramax = 0.f;
            // parked from grid position GROUP PL of the row, that is from sample GROUP PL - sh
            const int sh = shift_of(j0 - ROWS + sidx);
            bd_split_partial<NK, REM_UNITS>(keep[sidx] + 1, (int)L - GROUP * PL + sh, y, ramax);
            bd_rot(y.a1, y.b1, 1 * (GROUP * PL - sh));
            bd_rot(y.a2, y.b2, 2 * (GROUP * PL - sh));
            bd_split_slot S = slots[sidx];
            S.re1 += y.a1; S.im1 += y.b1; S.re2 += y.a2; S.im2 += y.b2; S.dc += y.dc;
            S.amax = fmaxf(S.amax, ramax);
            slots[sidx] = S;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint64_t o0 = s * a.out_stride + (uint64_t)j0;
    __amdgpu_buffer_rsrc_t rf = bd_out_rsrc(a.flags + o0, (unsigned)n_win);
    __amdgpu_buffer_rsrc_t rv = bd_out_rsrc(a.values ? a.values + o0 : nullptr, a.values ? (unsigned)n_win * 4u : 0u);
    // lane i finishes callback j0 + i (slot i + ROWS)
    for (int i = lane; i < n_win; i += 64) {
        const int64_t jj = j0 + i;
        const int q = (int)a.q, t = (int)a.tail_len;
        float xr1 = 0.f, xi1 = 0.f, xr2 = 0.f, xi2 = 0.f;
        int xdc = 0;
        const int64_t jt = jj - q;                          // the callback whose tail opens the ring
        if (jt >= 0 && t > 0) {
            const bd_split_slot T = slots[i + ROWS - q];
            // the piece that straddles the cut (grid position c + sh of that row): its first m samples
            // belong to the head
            const int sh = shift_of(jt);
            const int m = (c + sh) & 7;
            bd_split_acc hh;
            hh.a1 = hh.b1 = hh.a2 = hh.b2 = 0.f; hh.dc = 0;
            float unused = 0.f;
            bd_split_partial<NK, 1>(keep[i + ROWS - q], m, hh, unused);
            // anchored at the piece's first sample, c - m: move to the callback's first sample
            bd_rot(hh.a1, hh.b1, 1 * (c - m));
            bd_rot(hh.a2, hh.b2, 2 * (c - m));
            // T = w^{-k c} (S - H)
            xr1 = T.re1 - (T.hre1 + hh.a1); xi1 = T.im1 - (T.him1 + hh.b1);
            xr2 = T.re2 - (T.hre2 + hh.a2); xi2 = T.im2 - (T.him2 + hh.b2);
            bd_rot(xr1, xi1, -1 * c);
            bd_rot(xr2, xi2, -2 * c);
            xdc = T.dc - (T.hdc + hh.dc);
        }
        for (int mm = 1; mm <= q; mm++) {
            if (jt + mm < 0) continue;                      // before the stream: zeros
            const bd_split_slot S = slots[i + ROWS - q + mm];
            float r1 = S.re1, i1 = S.im1, r2 = S.re2, i2 = S.im2;
            const int off = t + (mm - 1) * (int)L;
            bd_rot(r1, i1, 1 * off);
            bd_rot(r2, i2, 2 * off);
            xr1 += r1; xi1 += i1; xr2 += r2; xi2 += i2;
            xdc += S.dc;
        }
        const float x_dc = (float)xdc;
        float best = a.include_dc ? fabsf(x_dc) : 0.f;
        if (NK >= 1) best = fmaxf(best, sqrtf(xr1 * xr1 + xi1 * xi1));
        if (NK >= 2) best = fmaxf(best, sqrtf(xr2 * xr2 + xi2 * xi2));
        const float value = (a.divisor == 0.f) ? best : best / a.divisor;
        const float wmax = fminf(slots[i + ROWS].amax, 32767.f);   // window_stats.rs:47
        const bool flag = (wmax >= (float)a.min_amp) && (value > a.threshold);
        bd_store_stream_u8(rf, (unsigned)i, flag ? 1 : 0);
        bd_store_stream_f32(rv, (unsigned)i * 4u, value);       // a zero-sized descriptor drops the store
    }
    // the next run of this wavefront finds its ROWS predecessors at the front of the slots
    if (MULTI && sub + 1 < (int)a.subruns && n_win == BD_SPLIT_RUN) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < ROWS) {
            slots[lane] = slots[BD_SPLIT_RUN + lane];
#pragma unroll
            for (int u = 0; u < KEEP; u++) keep[lane][u] = keep[BD_SPLIT_RUN + lane][u];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    }   // runs of this wavefront
}

// Callback lengths that run faster with rows loaded straight from memory than through the staged copy.
// Since the staged kernels walk several runs per wavefront and load through a descriptor, they are level
// with the direct kernels or ahead of them almost everywhere; the direct kernels keep 525 .. 529 samples
// (remainder of 13 .. 17 past four groups of 128, parked).  Measured at 16,384 streams x 60 s, whole call,
// staged / direct in TB/s: 385 5.44 / 5.38, 398 5.40 / 5.45, 401 5.46 / 5.44, 441 5.81 / 5.31, 500 6.10 / 5.79,
// 514 5.88 / 5.88, 517 5.98 / 5.71, 520 6.07 / 5.75, 522 5.85 / 5.70, 524 5.95 / 5.92, 525 5.78 / 5.90,
// 527 5.79 / 5.91, 528 5.94 / 6.02, 529 5.78 / 5.93, 532 5.97 / 5.62, 600 6.25 / 5.86, 641 6.17 / 5.89,
// 654 6.03 / 5.95, 657 6.14 / 6.04, 700 6.34 / 5.81, 767 6.38 / 6.37.
static bool bd_split_prefers_direct(uint32_t callback_len)
{
    return callback_len >= 525 && callback_len <= 529;
}

bool bd_spectrum_split_applies(const bd_plan &plan, uint32_t callback_len)
{
    // bins 0 .. 2, no Nyquist packing; lengths that 16 lanes x 24 .. 48 samples (8 lanes x 24 .. 32 below 256)
    // cover with most lanes busy and whose ring spans at most four (eight) callbacks
    if (callback_len <= 64 || callback_len == 128 || callback_len == 256 || callback_len >= 768) return false;
    return plan.last_bin <= 2 && !(plan.dc_packs_nyquist && plan.first_bin == 0);
}

// Callbacks 0 .. n_cb_eval - 1 of every stream (the caller keeps the last callback of a stream out:
// the 16-byte loads of a row may reach a few samples past the callback they belong to).
int bd_launch_spectrum_split(const bd_plan &plan, const int16_t *samples, uint64_t n_streams, uint64_t n_cb_eval,
                             uint64_t stride, uint32_t callback_len, float *values, uint8_t *flags,
                             uint64_t out_stride, hipStream_t st)
{
    if (n_cb_eval == 0 || n_streams == 0) return BD_OK;
    bd_split_args a;
    a.samples = samples; a.values = values; a.flags = flags;
    a.stride = stride; a.out_stride = out_stride; a.n_cb = n_cb_eval;
    a.callback_len = callback_len;
    a.q = BD_RING_LEN / callback_len;
    a.tail_len = BD_RING_LEN - a.q * callback_len;
    a.head_len = callback_len - a.tail_len;
    a.min_amp = plan.min_amp; a.include_dc = plan.first_bin == 0;
    a.divisor = plan.divisor; a.threshold = plan.threshold;
    // 385 .. 511 samples: two whole callbacks in the ring; 512 .. 767: one.  The callback ends in group 3
    // (385 .. 511), 4 (513 .. 639) or 5.  A remainder (samples past the last whole group) of at most
    // BD_SPLIT_REM_MAX samples stays out of the main loop.
    // 65 .. 127 samples: four lanes per callback (groups of 32 samples), eight to fifteen whole callbacks.
    // 129 .. 383 samples: eight lanes per callback (groups of 64 samples), two to seven whole callbacks in
    // the ring; the last group is always a masked piece of the main loop.  (Sixteen lanes per callback were
    // measured on 257 .. 383 as well: 4.0 .. 5.3 TB/s against 4.8 .. 5.8 with eight, which reduce over three
    // lane steps instead of four and twice as many callbacks per iteration.)
    const bool narrow = callback_len < 384, narrow4 = callback_len < 128;
    const int group = narrow4 ? 32 : (narrow ? 64 : 128);
    const int pstar = (int)(a.head_len / group), pl = (int)(callback_len / group);
    const int rem_len = (int)callback_len - group * pl;      // samples past the last whole group
    // Sixteen lanes per callback: rows are read on the grid of 16-byte lines (the row starts up to 7
    // samples into the lane grid) unless a shift by 7 samples would move the end of the head or of the
    // callback into another group of 128, or the head is shorter than one piece: those lengths, and the
    // narrow variants, read the staged copy at 4-byte-aligned addresses and funnel-shift odd rows.
    const bool sh = !narrow && a.head_len >= 8 && a.head_len % 128 <= 120 && rem_len <= 120;
    // Staged copy (global -> LDS in whole contiguous lines, lanes pick their pieces there) or rows loaded
    // straight from memory at 16-byte-aligned addresses (a row of 16 lanes then covers 256 contiguous bytes
    // that straddle three 128-byte lines).  The narrow variants always stage (a row of 4 or 8 lanes would
    // load only 64 or 128 contiguous bytes per instruction: 100 samples 3.0 -> 3.5 TB/s, 300 4.6 -> 5.2).
    const int mode = bd_tuning_get().split_stage;          // experiment switch: 1 = always staged, 2 = direct where possible
    const bool direct = sh && mode != 1 && (mode == 2 || bd_split_prefers_direct(callback_len));
    const bool staged = !direct;
    const bool rem = !narrow && rem_len + (sh ? 7 : 0) <= BD_SPLIT_REM_MAX && !bd_tuning_get().split_no_rem;
    // A wavefront of the staged kernels walks several runs of BD_SPLIT_RUN callbacks in a row (the runs after the
    // first need no warm-up iteration: 441 samples 5.53 -> 5.78 TB/s with four, 300 5.26 -> 5.65 and 100 3.4 -> 4.0
    // with eight), fewer where the batch would otherwise leave CUs without work.  The kernels that load rows
    // straight from memory keep one run per wavefront (529 samples: 6.05 TB/s with one, 5.75 - 5.85 with more).
    const uint64_t runs64 = (n_cb_eval + BD_SPLIT_RUN - 1) / BD_SPLIT_RUN;
    uint32_t subruns = staged ? (narrow ? 8 : 4) : 1;
    const int knob = bd_tuning_get().split_subruns;
    if (!staged) subruns = 1;                      // (those kernels are built for one run per wavefront)
    else if (knob >= 1 && knob <= 64) subruns = (uint32_t)knob;
    else while (subruns > 1 && n_streams * ((runs64 + subruns - 1) / subruns) < 16384) subruns >>= 1;
    const uint64_t runs = (runs64 + subruns - 1) / subruns;
    if (runs > 0xffffffffull) { bd_set_error("stream too long"); return BD_ERR_UNSUPPORTED; }
    a.subruns = subruns;
    a.runs_per_stream = (uint32_t)runs;
    a.n_tasks = n_streams * runs;
    const unsigned waves = (unsigned)bd_split_waves(staged, rem);
    dim3 grid;
    if (!bd_grid_dims((a.n_tasks + waves - 1) / waves, &grid)) { bd_set_error("batch too large for one launch"); return BD_ERR_UNSUPPORTED; }
#define BD_SPLIT_GO(NK, PS, PLAST, RM, LANES) do { \
        if (staged && sh) hipLaunchKernelGGL((bd_spectrum_split_kernel<NK, PS, PLAST, true, RM, LANES, true>), grid, dim3(64 * waves), 0, st, a); \
        else if (staged) hipLaunchKernelGGL((bd_spectrum_split_kernel<NK, PS, PLAST, true, RM, LANES, false>), grid, dim3(64 * waves), 0, st, a); \
        else hipLaunchKernelGGL((bd_spectrum_split_kernel<NK, PS, PLAST, false, RM, LANES, true>), grid, dim3(64 * waves), 0, st, a); \
        launched = true; } while (0)
#define BD_SPLIT_GO_ST(NK, PS, PLAST, LANES) do { \
        hipLaunchKernelGGL((bd_spectrum_split_kernel<NK, PS, PLAST, true, false, LANES, false>), grid, dim3(64 * waves), 0, st, a); \
        launched = true; } while (0)
#define BD_SPLIT_PICK(NK) do { \
        if (narrow4) { \
            if (pl == 2 && pstar == 0) BD_SPLIT_GO_ST(NK, 0, 2, 4); \
            else if (pl == 2 && pstar == 1) BD_SPLIT_GO_ST(NK, 1, 2, 4); \
            else if (pl == 2 && pstar == 2) BD_SPLIT_GO_ST(NK, 2, 2, 4); \
            else if (pl == 3 && pstar == 0) BD_SPLIT_GO_ST(NK, 0, 3, 4); \
            else if (pl == 3 && pstar == 1) BD_SPLIT_GO_ST(NK, 1, 3, 4); \
            else if (pl == 3 && pstar == 2) BD_SPLIT_GO_ST(NK, 2, 3, 4); \
            else if (pl == 3 && pstar == 3) BD_SPLIT_GO_ST(NK, 3, 3, 4); \
        } \
        else if (narrow) { \
            if (pl == 2 && pstar == 0) BD_SPLIT_GO_ST(NK, 0, 2, 8); \
            else if (pl == 2 && pstar == 1) BD_SPLIT_GO_ST(NK, 1, 2, 8); \
            else if (pl == 2 && pstar == 2) BD_SPLIT_GO_ST(NK, 2, 2, 8); \
            else if (pl == 3 && pstar == 0) BD_SPLIT_GO_ST(NK, 0, 3, 8); \
            else if (pl == 3 && pstar == 1) BD_SPLIT_GO_ST(NK, 1, 3, 8); \
            else if (pl == 3 && pstar == 2) BD_SPLIT_GO_ST(NK, 2, 3, 8); \
            else if (pl == 3 && pstar == 3) BD_SPLIT_GO_ST(NK, 3, 3, 8); \
            else if (pl == 4 && pstar == 0) BD_SPLIT_GO_ST(NK, 0, 4, 8); \
            else if (pl == 4 && pstar == 1) BD_SPLIT_GO_ST(NK, 1, 4, 8); \
            else if (pl == 4 && pstar == 2) BD_SPLIT_GO_ST(NK, 2, 4, 8); \
            else if (pl == 4 && pstar == 3) BD_SPLIT_GO_ST(NK, 3, 4, 8); \
            else if (pl == 4 && pstar == 4) BD_SPLIT_GO_ST(NK, 4, 4, 8); \
            else if (pl == 5 && pstar == 0) BD_SPLIT_GO_ST(NK, 0, 5, 8); \
            else if (pl == 5 && pstar == 1) BD_SPLIT_GO_ST(NK, 1, 5, 8); \
            else if (pl == 5 && pstar == 2) BD_SPLIT_GO_ST(NK, 2, 5, 8); \
            else if (pl == 5 && pstar == 3) BD_SPLIT_GO_ST(NK, 3, 5, 8); \
            else if (pl == 5 && pstar == 4) BD_SPLIT_GO_ST(NK, 4, 5, 8); \
            else if (pl == 5 && pstar == 5) BD_SPLIT_GO_ST(NK, 5, 5, 8); \
        } \
        else if (pl == 4 && pstar == 0) { if (rem) BD_SPLIT_GO(NK, 0, 4, true, 16); else BD_SPLIT_GO(NK, 0, 4, false, 16); } \
        else if (pl == 4 && pstar == 1) { if (rem) BD_SPLIT_GO(NK, 1, 4, true, 16); else BD_SPLIT_GO(NK, 1, 4, false, 16); } \
        else if (pl == 3 && pstar == 1) { if (rem) BD_SPLIT_GO(NK, 1, 3, true, 16); else BD_SPLIT_GO(NK, 1, 3, false, 16); } \
        else if (pl == 3 && pstar == 2) { if (rem) BD_SPLIT_GO(NK, 2, 3, true, 16); else BD_SPLIT_GO(NK, 2, 3, false, 16); } \
        else if (pl == 3 && pstar == 3) { BD_SPLIT_GO(NK, 3, 3, false, 16); } \
        else if (pl == 5 && pstar == 2) { if (rem) BD_SPLIT_GO(NK, 2, 5, true, 16); else BD_SPLIT_GO(NK, 2, 5, false, 16); } \
        else if (pl == 5 && pstar == 3) { BD_SPLIT_GO(NK, 3, 5, false, 16); } } while (0)
    bool launched = false;
    if (plan.last_bin == 0) BD_SPLIT_PICK(0);
    else if (plan.last_bin == 1) BD_SPLIT_PICK(1);
    else BD_SPLIT_PICK(2);
#undef BD_SPLIT_PICK
#undef BD_SPLIT_GO_ST
#undef BD_SPLIT_GO
    if (!launched) { bd_set_error("callback length %u outside the split kernel's range", callback_len); return BD_ERR_UNSUPPORTED; }
    bd_set_last_kernel("bd_spectrum_split_kernel");
    return bd_check_hip(hipGetLastError(), "launch bd_spectrum_split_kernel");
}
