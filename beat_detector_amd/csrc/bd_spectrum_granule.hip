// Spectrum strategy, batched, for callback lengths that are multiples of 64, 32
// or 16 samples (up to 4096, 512, 256) other than the two the half-block kernel covers: the value
// of src/strategies/spectrum.rs:82-92 -- largest magnitude among the bins <= 90 Hz
// of the 1024-point DFT of the latest 1024 samples -- and the decision of
// spectrum.rs:77-92, one HBM read per sample.
//
// The stream is cut into granules of G = 64 samples (32 or 16 when the callback length
// asks for it); the ring of a callback is the R = 1024 / G granules that end with it.
// With P_k(g) = sum_{n<G} x[G g + n] w^{k n}, w = exp(2 pi i / 1024) (the conjugate
// of the DFT kernel: magnitudes are the same),
//     X_k(e) = sum_{i<R} w^{G k i} P_k(e - R + 1 + i)
// is built by log2 R doubling levels
//     Q^{j+1}(e) = Q^{j}(e - 2^j) + w^{G k 2^j} Q^{j}(e),     Q^0 = P,  X = Q^{log2 R},
// so every granule is transformed once however the callbacks overlap the ring.
//
// One wavefront walks consecutive steps of 64 granules (8 KB for G = 64) of one stream: coalesced
// 16-byte loads, transposed through a private LDS tile (rows of 2 G + 16 bytes, conflict-free)
// so that lane l owns granule l; per-lane pruned DFT with compile-time twiddles; the
// doubling levels fetch lane l - 2^j from this step or, for the first lanes, from the
// values the previous step left in registers.  The amplitude gate (max |x| over the
// callback's own granules) is a sliding maximum built the same way.  Lanes whose
// granule ends a callback finish it; results are compacted, staged in LDS and stored
// contiguously.  A run that does not start the stream first replays the step before it.
#include "bd_internal.h"
#include "bd_granule_tw.inc"


typedef short bd_gr_short2 __attribute__((ext_vector_type(2)));

struct bd_gr_args {
    const int16_t *samples;
    float         *values;
    uint8_t       *flags;
    uint64_t       stride, out_stride, n_cb, n_tasks;
    uint64_t       n_gran;          // granules that belong to callbacks = n_cb * m
    uint32_t       m;               // granules per callback (1 .. 64)
    uint32_t       misaligned;      // rows are not 16-byte aligned: loads at 2-byte-aligned addresses
    uint32_t       steps_per_run, runs_per_stream;
    uint32_t       min_amp, include_dc;
    uint32_t       nk_keep;         // non-DC bins that count (<= the NK the kernel evaluates)
    float          divisor, threshold;
};

__device__ __forceinline__ uint4 bd_gr_load16(const int16_t *p, bool misaligned)
{
    if (!misaligned) return bd_load_stream16(p);
    typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
    const bd_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a2 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ void bd_gr_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// running maximum of magnitudes, kept in program order (the optimizer would re-associate
// a chain of fmaxf into a tree after the loop and keep every operand alive)
__device__ __forceinline__ void bd_gr_absmax(float &m, float a, float b)
{
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
}

// value of lane (l - d) of this step, or of lane (l - d + 64) of the previous step.  The lanes whose
// value wraps around (the last d) hand over the previous step's value, so that one rotation of the
// wavefront by d delivers both: a DPP wave rotation for d = 1, 2 (no LDS traffic; `d` is a constant
// there after unrolling), ds_bpermute otherwise.
__device__ __forceinline__ int bd_gr_rot1(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int bd_gr_back(int cur, int prev, int lane, int d)
{
    const int send = lane >= 64 - d ? prev : cur;
    if (__builtin_constant_p(d) && d == 1) return bd_gr_rot1(send);
    if (__builtin_constant_p(d) && d == 2) return bd_gr_rot1(bd_gr_rot1(send));
    return __shfl(send, (lane - d) & 63, 64);
}
__device__ __forceinline__ float bd_gr_back(float cur, float prev, int lane, int d)
{
    return __builtin_bit_cast(float, bd_gr_back(__builtin_bit_cast(int, cur), __builtin_bit_cast(int, prev), lane, d));
}

// G: samples per granule (64, 32 or 16).  NK: non-DC bins evaluated: 0 .. 2 at the usual sampling
// rates; 3, 4, 5, 8 or 11 where the band up to 90 Hz holds more bins (sampling rates down to 8 kHz;
// at most 8 with G = 32, 5 with G = 16), of which the first nk_keep count.
template <int NK, bool NYQ, int G>
__device__ __forceinline__ void bd_spectrum_granule_body(const bd_gr_args &a)
{
    constexpr int DW = G / 2;                 // dwords of samples per granule
    constexpr int U4 = G / 8;                 // 16-byte units per granule
    constexpr int BD_GR_ROW = DW + 4;         // dwords per LDS row: the samples + 16 B pad
    constexpr int LV = G == 64 ? 4 : G == 32 ? 5 : 6;   // doubling levels: log2(1024 / G)
    constexpr int LV0 = 6 - LV;               // first entry of the level table
    __shared__ __attribute__((aligned(16))) uint32_t tile_all[4][64 * BD_GR_ROW];
    __shared__ float out_v[4][128];
    __shared__ uint8_t out_f[4][128];

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t task = bd_block_index() * 4 + wave;
    if (task >= a.n_tasks) return;
    const uint64_t s = task / a.runs_per_stream;
    const uint64_t r = task - s * a.runs_per_stream;
    uint32_t *tile = tile_all[wave];
    const int16_t *sp = a.samples + s * a.stride;
    const uint64_t n_steps = (a.n_gran + 63) / 64;
    const uint64_t st0 = r * a.steps_per_run;
    const uint64_t st1 = st0 + a.steps_per_run < n_steps ? st0 + a.steps_per_run : n_steps;
    const uint64_t st_first = st0 > 0 ? st0 - 1 : 0;   // replayed step: fills the "previous step" values
    const uint64_t last_unit = a.n_gran * U4 - 1;       // last 16-byte unit that belongs to a callback
    const uint32_t m = a.m;
    // sliding maximum over m granules = two overlapping windows of 2^lg granules
    const int lg = 31 - __builtin_clz(m);
    const int tail = (int)m - (1 << lg);

    const bool mis = a.misaligned != 0;
    uint4 ld[U4];
    auto load_step = [&](uint64_t st) {
        if ((st + 1) * 64 <= a.n_gran) {
            const int16_t *base = sp + st * (64 * G) + lane * 8;
#pragma unroll
            for (int j = 0; j < U4; j++) ld[j] = bd_gr_load16(base + j * 512, mis);
        } else { // units past the analysed range are replaced by the last valid one (never reported)
#pragma unroll
            for (int j = 0; j < U4; j++) {
                uint64_t unit = st * (64 * U4) + (uint64_t)(j * 64 + lane);
                unit = unit < last_unit ? unit : last_unit;
                ld[j] = bd_gr_load16(sp + unit * 8, mis);
            }
        }
    };
    load_step(st_first);
    // unit j * 64 + lane of the step belongs to granule (j * 64 + lane) / U4, piece (j * 64 + lane) % U4
    uint4 *wr = reinterpret_cast<uint4 *>(tile + (lane / U4) * BD_GR_ROW + (lane % U4) * 4); // + j * (64 / U4) rows
    const uint4 *rd = reinterpret_cast<const uint4 *>(tile + lane * BD_GR_ROW);

    // values of the previous step, per doubling level (zero = silence before the stream)
    float p_re[LV][NK > 0 ? NK : 1], p_im[LV][NK > 0 ? NK : 1], p_amax[6];
    int p_dc[LV], p_ny[LV];
#pragma unroll
    for (int j = 0; j < LV; j++) {
        p_dc[j] = 0; p_ny[j] = 0;
#pragma unroll
        for (int k = 0; k < (NK > 0 ? NK : 1); k++) { p_re[j][k] = 0.f; p_im[j][k] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) p_amax[j] = 0.f;
    float p_amax_top = 0.f;                              // level lg of the previous step

    uint32_t pend = 0;                                   // results staged in LDS
    uint64_t cb_base = (st0 * 64) / m;                   // callbacks that end before this run

    for (uint64_t st = st_first; st < st1; st++) {
        bd_gr_sync();
#pragma unroll
        for (int j = 0; j < U4; j++) wr[j * (64 / U4) * (BD_GR_ROW / 4)] = ld[j];
        bd_gr_sync();
        uint32_t raw[DW];
#pragma unroll
        for (int c = 0; c < U4; c++) {
            const uint4 q = rd[c];
            raw[4 * c] = q.x; raw[4 * c + 1] = q.y; raw[4 * c + 2] = q.z; raw[4 * c + 3] = q.w;
        }
        if (st + 1 < st1) load_step(st + 1);

        // pruned DFT of the lane's granule
        float re[NK > 0 ? NK : 1], im[NK > 0 ? NK : 1], amax = 0.f;
        int dc = 0, ny = 0;
#pragma unroll
        for (int k = 0; k < (NK > 0 ? NK : 1); k++) { re[k] = 0.f; im[k] = 0.f; }
#pragma unroll
        for (int i = 0; i < DW; i++) {
            const int v = (int)raw[i];
            const float x0 = (float)(short)(v & 0xffff), x1 = (float)(short)(v >> 16);
            dc = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_gr_short2, v), bd_gr_short2{1, 1}, dc, false);
            if (NYQ) ny = __builtin_amdgcn_sdot2(__builtin_bit_cast(bd_gr_short2, v), bd_gr_short2{1, -1}, ny, false);
            bd_gr_absmax(amax, x0, x1);
#pragma unroll
            for (int k = 0; k < NK; k++) {
                re[k] = __builtin_fmaf(x0, BD_GR_C[k][2 * i], re[k]);     im[k] = __builtin_fmaf(x0, BD_GR_S[k][2 * i], im[k]);
                re[k] = __builtin_fmaf(x1, BD_GR_C[k][2 * i + 1], re[k]); im[k] = __builtin_fmaf(x1, BD_GR_S[k][2 * i + 1], im[k]);
            }
        }
        // doubling levels: 1 -> 1024 / G granules
#pragma unroll
        for (int j = 0; j < LV; j++) {
            const int d = 1 << j;
            const int b_dc = bd_gr_back(dc, p_dc[j], lane, d);
            p_dc[j] = dc;
            dc = b_dc + dc;
            if (NYQ) {
                const int b_ny = bd_gr_back(ny, p_ny[j], lane, d);
                p_ny[j] = ny;
                ny = b_ny + ny;
            }
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const float b_re = bd_gr_back(re[k], p_re[j][k], lane, d), b_im = bd_gr_back(im[k], p_im[j][k], lane, d);
                p_re[j][k] = re[k]; p_im[j][k] = im[k];
                const float c = BD_GRL_C[k][LV0 + j], sn = BD_GRL_S[k][LV0 + j];
                const float nr = b_re + (c * re[k] - sn * im[k]);
                const float ni = b_im + (c * im[k] + sn * re[k]);
                re[k] = nr; im[k] = ni;
            }
        }
        // amplitude gate: maximum over the callback's own m granules (lg doubling levels, then
        // two overlapping windows of 2^lg granules)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            if (j < lg) {
                const float b = bd_gr_back(amax, p_amax[j], lane, 1 << j);
                p_amax[j] = amax;
                amax = fmaxf(amax, b);
            }
        }
        float gate_max = amax;
        if (tail > 0) gate_max = fmaxf(amax, bd_gr_back(amax, p_amax_top, lane, tail));
        p_amax_top = amax;

        // lanes whose granule ends a callback finish it
        const uint64_t g = st * 64 + (uint64_t)lane;
        const bool ends = st >= st0 && g < a.n_gran && (g + 1) % m == 0;
        float value = 0.f;
        bool flag = false;
        {
            const float x_dc = (float)dc;
            float m0 = fabsf(x_dc);
            if (NYQ) {
                const float x_ny = (float)ny;
                m0 = sqrtf(x_dc * x_dc + x_ny * x_ny);
            }
            float best = a.include_dc ? m0 : 0.f;
#pragma unroll
            for (int k = 0; k < NK; k++)
                if (NK <= 2 || k < (int)a.nk_keep) best = fmaxf(best, sqrtf(re[k] * re[k] + im[k] * im[k]));
            value = (a.divisor == 0.f) ? best : best / a.divisor;
            flag = fminf(gate_max, 32767.f) >= (float)a.min_amp && value > a.threshold;
        }
        const unsigned long long mask = __ballot(ends);
        if (ends) {
            const uint32_t slot = pend + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            out_v[wave][slot] = value;
            out_f[wave][slot] = flag ? 1 : 0;
        }
        pend += (uint32_t)__builtin_popcountll(mask);
        if (pend >= 64 || st + 1 == st1) {
            bd_gr_sync();
            const uint64_t o0 = s * a.out_stride + cb_base;
            if (a.values) {
                __amdgpu_buffer_rsrc_t rv = bd_out_rsrc(a.values + o0, pend * 4u);
                for (unsigned i = lane; i < pend; i += 64) bd_store_stream_f32(rv, i * 4u, out_v[wave][i]);
            }
            __amdgpu_buffer_rsrc_t rf = bd_out_rsrc(a.flags + o0, pend);
            for (unsigned i = lane; i < pend; i += 64) bd_store_stream_u8(rf, i, out_f[wave][i]);
            cb_base += pend;
            pend = 0;
        }
    }
}

template <int NK, bool NYQ, int G>
__global__ __launch_bounds__(256) void bd_spectrum_granule_kernel(bd_gr_args a)
{
    bd_spectrum_granule_body<NK, NYQ, G>(a);
}
// eight and more bins: the doubling levels of every bin fill the register file; asking for two waves
// per SIMD keeps the compiler from settling just above that limit
template <int NK, bool NYQ, int G>
__global__ __launch_bounds__(256, 2) void bd_spectrum_granule_kernel_w2(bd_gr_args a)
{
    bd_spectrum_granule_body<NK, NYQ, G>(a);
}

// ---- granules of 63 samples -------------------------------------------------
// Callback lengths 63 q (441 = 10 ms at 44.1 kHz, 882, ...) end on a grid of 63-sample granules.
// The ring of a callback is the last 16 granules (1008 samples) plus the last 16 samples of the
// granule before them.  Every lane sums its granule in two parts, A over samples 0 .. 46 and B over
// samples 47 .. 62 (P = A + B), and
//     X_k(e) = B_k(e - 16) + w^{63 k} Q_k(e),   Q = the four doubling levels over P with offsets 63 * 2^j
// up to a rotation of the whole ring, which no magnitude depends on.  The callback's own samples are
// its last q granules, so the amplitude gate is the same sliding maximum as for the other granule
// sizes.  A step is 64 granules = 4032 samples = 504 16-byte units, copied to LDS as they lie in
// memory; lane l reads its 63 samples from byte 126 l on.
struct bd_g63_part { float re[2], im[2]; int dc, ny; };

// Samples FROM .. TO - 1 of the lane's granule.  (Written sample by sample: the compiler turns the
// reads into 16-byte LDS reads at the row's 2-byte-aligned address; a version on explicit dwords
// with packed integer sums was slower.)
template <int NK, bool NYQ, int FROM, int TO>
__device__ __forceinline__ void bd_g63_sum(const int16_t *row, bd_g63_part &p, float &amax)
{
#pragma unroll
    for (int k = 0; k < 2; k++) { p.re[k] = 0.f; p.im[k] = 0.f; }
    p.dc = 0; p.ny = 0;
#pragma unroll
    for (int n = FROM; n < TO; n++) {
        const int v = row[n];
        const float x = (float)v;
        p.dc += v;
        if (NYQ) p.ny += (n & 1) ? -v : v;
        amax = fmaxf(amax, fabsf(x));
#pragma unroll
        for (int k = 0; k < NK; k++) {
            p.re[k] = __builtin_fmaf(x, BD_GR_C[k][n], p.re[k]);
            p.im[k] = __builtin_fmaf(x, BD_GR_S[k][n], p.im[k]);
        }
    }
}

template <int NK, bool NYQ>
__global__ __launch_bounds__(256, 3) void bd_spectrum_g63_kernel(bd_gr_args a)
{
    constexpr int G = 63, UNITS = 504;          // samples per granule, 16-byte units per step
    __shared__ __attribute__((aligned(16))) uint4 tile_all[4][UNITS];
    __shared__ float out_v[4][128];
    __shared__ uint8_t out_f[4][128];

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t task = bd_block_index() * 4 + wave;
    if (task >= a.n_tasks) return;
    const uint64_t s = task / a.runs_per_stream;
    const uint64_t r = task - s * a.runs_per_stream;
    uint4 *tile = tile_all[wave];
    const int16_t *sp = a.samples + s * a.stride;
    const uint64_t n_steps = (a.n_gran + 63) / 64;
    const uint64_t st0 = r * a.steps_per_run;
    const uint64_t st1 = st0 + a.steps_per_run < n_steps ? st0 + a.steps_per_run : n_steps;
    const uint64_t st_first = st0 > 0 ? st0 - 1 : 0;   // replayed step: fills the "previous step" values
    const uint64_t n_valid = a.n_gran * G;              // samples that belong to callbacks
    const uint32_t m = a.m;
    const int lg = 31 - __builtin_clz(m);
    const int tail = (int)m - (1 << lg);

    const bool mis = a.misaligned != 0;
    uint4 ld[8];
    auto load_step = [&](uint64_t st) {
        const uint64_t first = st * (64 * G);           // first sample of the step
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int u = j * 64 + lane;
            const uint64_t at = first + (uint64_t)u * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (u < UNITS && at + 8 <= n_valid) {
                v = bd_gr_load16(sp + at, mis);
            } else if (u < UNITS && at < n_valid) {      // the unit that holds the last samples: one by one
                unsigned w[4] = {0, 0, 0, 0};
                for (uint64_t e = 0; e < n_valid - at; e++) w[e >> 1] |= (unsigned)(uint16_t)sp[at + e] << (16 * (e & 1));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            ld[j] = v;
        }
    };
    load_step(st_first);
    const int16_t *row = reinterpret_cast<const int16_t *>(tile) + G * lane;

    // values of the previous step, per doubling level (zero = silence before the stream)
    float p_re[4][NK > 0 ? NK : 1], p_im[4][NK > 0 ? NK : 1], p_amax[6];
    int p_dc[4], p_ny[4];
    float pb_re[NK > 0 ? NK : 1], pb_im[NK > 0 ? NK : 1];   // B of the previous step
    int pb_dc = 0, pb_ny = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        p_dc[j] = 0; p_ny[j] = 0;
#pragma unroll
        for (int k = 0; k < (NK > 0 ? NK : 1); k++) { p_re[j][k] = 0.f; p_im[j][k] = 0.f; }
    }
#pragma unroll
    for (int k = 0; k < (NK > 0 ? NK : 1); k++) { pb_re[k] = 0.f; pb_im[k] = 0.f; }
#pragma unroll
    for (int j = 0; j < 6; j++) p_amax[j] = 0.f;
    float p_amax_top = 0.f;

    uint32_t pend = 0;
    uint64_t cb_base = (st0 * 64) / m;

    for (uint64_t st = st_first; st < st1; st++) {
        bd_gr_sync();
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j * 64 + lane < UNITS) tile[j * 64 + lane] = ld[j];
        bd_gr_sync();
        // the lane's granule in two parts; the alternating sum takes the sign of the sample's place
        // in the stream (granule g starts at an odd sample when g is odd)
        bd_g63_part A, B;
        float amax = 0.f;
        bd_g63_sum<NK, NYQ, 0, 47>(row, A, amax);
        bd_g63_sum<NK, NYQ, 47, 63>(row, B, amax);
        if (st + 1 < st1) load_step(st + 1);
        if (NYQ && (lane & 1)) { A.ny = -A.ny; B.ny = -B.ny; }
        float re[NK > 0 ? NK : 1], im[NK > 0 ? NK : 1];
        int dc = A.dc + B.dc, ny = A.ny + B.ny;
#pragma unroll
        for (int k = 0; k < NK; k++) { re[k] = A.re[k] + B.re[k]; im[k] = A.im[k] + B.im[k]; }
        // B of the granule 16 before this one
        float b_re[NK > 0 ? NK : 1], b_im[NK > 0 ? NK : 1];
        const int b_dc = bd_gr_back(B.dc, pb_dc, lane, 16);
        const int b_ny = NYQ ? bd_gr_back(B.ny, pb_ny, lane, 16) : 0;
        pb_dc = B.dc; pb_ny = B.ny;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            b_re[k] = bd_gr_back(B.re[k], pb_re[k], lane, 16);
            b_im[k] = bd_gr_back(B.im[k], pb_im[k], lane, 16);
            pb_re[k] = B.re[k]; pb_im[k] = B.im[k];
        }
        // doubling levels: 1 -> 16 granules
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int d = 1 << j;
            const int q_dc = bd_gr_back(dc, p_dc[j], lane, d);
            p_dc[j] = dc;
            dc = q_dc + dc;
            if (NYQ) {
                const int q_ny = bd_gr_back(ny, p_ny[j], lane, d);
                p_ny[j] = ny;
                ny = q_ny + ny;
            }
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const float q_re = bd_gr_back(re[k], p_re[j][k], lane, d), q_im = bd_gr_back(im[k], p_im[j][k], lane, d);
                p_re[j][k] = re[k]; p_im[j][k] = im[k];
                const float c = BD_G63L_C[k][j], sn = BD_G63L_S[k][j];
                const float nr = q_re + (c * re[k] - sn * im[k]);
                const float ni = q_im + (c * im[k] + sn * re[k]);
                re[k] = nr; im[k] = ni;
            }
        }
        // the 16 samples before the 16 granules
        dc += b_dc; ny += b_ny;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const float c = BD_G63L_C[k][0], sn = BD_G63L_S[k][0];
            const float nr = b_re[k] + (c * re[k] - sn * im[k]);
            const float ni = b_im[k] + (c * im[k] + sn * re[k]);
            re[k] = nr; im[k] = ni;
        }
        // amplitude gate: maximum over the callback's own m granules
#pragma unroll
        for (int j = 0; j < 6; j++) {
            if (j < lg) {
                const float b = bd_gr_back(amax, p_amax[j], lane, 1 << j);
                p_amax[j] = amax;
                amax = fmaxf(amax, b);
            }
        }
        float gate_max = amax;
        if (tail > 0) gate_max = fmaxf(amax, bd_gr_back(amax, p_amax_top, lane, tail));
        p_amax_top = amax;

        const uint64_t g = st * 64 + (uint64_t)lane;
        const bool ends = st >= st0 && g < a.n_gran && (g + 1) % m == 0;
        float value = 0.f;
        bool flag = false;
        {
            const float x_dc = (float)dc;
            float m0 = fabsf(x_dc);
            if (NYQ) {
                const float x_ny = (float)ny;
                m0 = sqrtf(x_dc * x_dc + x_ny * x_ny);
            }
            float best = a.include_dc ? m0 : 0.f;
#pragma unroll
            for (int k = 0; k < NK; k++) best = fmaxf(best, sqrtf(re[k] * re[k] + im[k] * im[k]));
            value = (a.divisor == 0.f) ? best : best / a.divisor;
            flag = fminf(gate_max, 32767.f) >= (float)a.min_amp && value > a.threshold;
        }
        const unsigned long long mask = __ballot(ends);
        if (ends) {
            const uint32_t slot = pend + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            out_v[wave][slot] = value;
            out_f[wave][slot] = flag ? 1 : 0;
        }
        pend += (uint32_t)__builtin_popcountll(mask);
        if (pend >= 64 || st + 1 == st1) {
            bd_gr_sync();
            const uint64_t o0 = s * a.out_stride + cb_base;
            if (a.values) {
                __amdgpu_buffer_rsrc_t rv = bd_out_rsrc(a.values + o0, pend * 4u);
                for (unsigned i = lane; i < pend; i += 64) bd_store_stream_f32(rv, i * 4u, out_v[wave][i]);
            }
            __amdgpu_buffer_rsrc_t rf = bd_out_rsrc(a.flags + o0, pend);
            for (unsigned i = lane; i < pend; i += 64) bd_store_stream_u8(rf, i, out_f[wave][i]);
            cb_base += pend;
            pend = 0;
        }
    }
}

template <int NK>
static void bd_g63_launch(bool nyq, const bd_gr_args &a, dim3 grid, hipStream_t st)
{
    if (nyq) hipLaunchKernelGGL((bd_spectrum_g63_kernel<NK, true>), dim3(grid), dim3(256), 0, st, a);
    else     hipLaunchKernelGGL((bd_spectrum_g63_kernel<NK, false>), dim3(grid), dim3(256), 0, st, a);
}

template <int NK, int G>
static void bd_gr_launch2(bool nyq, const bd_gr_args &a, dim3 grid, hipStream_t st)
{
    if (NK >= 8) {
        if (nyq) hipLaunchKernelGGL((bd_spectrum_granule_kernel_w2<NK, true, G>), dim3(grid), dim3(256), 0, st, a);
        else     hipLaunchKernelGGL((bd_spectrum_granule_kernel_w2<NK, false, G>), dim3(grid), dim3(256), 0, st, a);
        return;
    }
    if (nyq) hipLaunchKernelGGL((bd_spectrum_granule_kernel<NK, true, G>), dim3(grid), dim3(256), 0, st, a);
    else     hipLaunchKernelGGL((bd_spectrum_granule_kernel<NK, false, G>), dim3(grid), dim3(256), 0, st, a);
}
template <int NK>
static void bd_gr_launch(uint32_t gran, bool nyq, const bd_gr_args &a, dim3 grid, hipStream_t st)
{
    if (gran == 64) bd_gr_launch2<NK, 64>(nyq, a, grid, st);
    else if (gran == 32) bd_gr_launch2<NK, 32>(nyq, a, grid, st);
    else bd_gr_launch2<NK, 16>(nyq, a, grid, st);
}

// Granule size for a callback length, 0 if the granule kernel does not take it: 64 for multiples
// of 64 up to 4096; 32 for multiples of 32 up to 512 and 16 for multiples of 16 up to 256, where
// the granule kernel beats the window kernel (measured, 2,048 streams x 60 s: 480: 4.3 vs 4.0 TB/s,
// 160: 4.4 vs 1.6, 240: 2.5 vs 2.3; but 720 with granules of 16: 2.4 vs 5.3).
uint32_t bd_spectrum_granule_size(uint32_t callback_len)
{
    if (callback_len % 64 == 0) return callback_len <= 4096 ? 64 : 0;
    if (callback_len % 32 == 0) return callback_len <= 512 ? 32 : 0;
    if (callback_len % 16 == 0) return callback_len <= 256 ? 16 : 0;
    // 63 q: 441 (10 ms at 44.1 kHz) and its kin, while the window kernel is not yet bound by HBM
    if (callback_len % 63 == 0) return callback_len <= 819 ? 63 : 0;
    return 0;
}

// Most non-DC bins the kernel evaluates with granules of `gran` samples (the doubling levels of
// every bin live in registers: fewer bins where there are more levels).
int bd_spectrum_granule_max_bins(uint32_t gran)
{
    return gran == 64 ? BD_GR_MAX_BINS : gran == 32 ? 8 : gran == 16 ? 5 : gran == 63 ? 2 : -1;
}

// Returns BD_OK after enqueueing, or a negative status; the caller has checked that
// bd_spectrum_granule_size(callback_len) != 0 and the kept bins are 0 .. last_bin with
// last_bin <= bd_spectrum_granule_max_bins(granule size).
int bd_launch_spectrum_granule(const bd_plan &plan, const int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                               uint64_t stride, uint32_t callback_len, float *values, uint8_t *flags,
                               uint64_t out_stride, hipStream_t st)
{
    bd_gr_args a;
    a.samples = samples; a.values = values; a.flags = flags;
    a.stride = stride; a.out_stride = out_stride;
    a.n_cb = n_samples / callback_len;
    const uint32_t gran = bd_spectrum_granule_size(callback_len);
    if (gran == 0) { bd_set_error("callback length has no granule size"); return BD_ERR_INVALID_ARG; }
    a.m = callback_len / gran;
    a.misaligned = (((uintptr_t)samples % 16) || (stride % 8)) ? 1u : 0u;
    a.n_gran = a.n_cb * a.m;
    const uint64_t n_steps = (a.n_gran + 63) / 64;
    // long runs amortise the replayed step; short batches still need wavefronts for 256 CUs
    uint64_t spr = n_steps;
    while (spr > 16 && n_streams * ((n_steps + spr - 1) / spr) < 8192) spr = (spr + 1) / 2;
    a.steps_per_run = (uint32_t)spr;
    a.runs_per_stream = (uint32_t)((n_steps + spr - 1) / spr);
    a.n_tasks = n_streams * a.runs_per_stream;
    a.min_amp = plan.min_amp; a.include_dc = plan.first_bin == 0;
    a.divisor = plan.divisor; a.threshold = plan.threshold;
    const uint64_t grid64 = (a.n_tasks + 3) / 4;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims) || n_steps > 0xffffffffull) { bd_set_error("batch too large for one launch"); return BD_ERR_UNSUPPORTED; }
    const bool nyq = plan.dc_packs_nyquist && plan.first_bin == 0;
    const dim3 grid = grid_dims;
    a.nk_keep = (uint32_t)plan.last_bin;
    if (plan.last_bin > bd_spectrum_granule_max_bins(gran)) { bd_set_error("too many bins for this granule size"); return BD_ERR_UNSUPPORTED; }
    if (gran == 63) {
        if (plan.last_bin == 0) bd_g63_launch<0>(nyq, a, grid, st);
        else if (plan.last_bin == 1) bd_g63_launch<1>(nyq, a, grid, st);
        else bd_g63_launch<2>(nyq, a, grid, st);
        bd_set_last_kernel("bd_spectrum_g63_kernel");
        return bd_check_hip(hipGetLastError(), "launch bd_spectrum_g63_kernel");
    }
    if (plan.last_bin == 0) bd_gr_launch<0>(gran, nyq, a, grid, st);
    else if (plan.last_bin == 1) bd_gr_launch<1>(gran, nyq, a, grid, st);
    else if (plan.last_bin == 2) bd_gr_launch<2>(gran, nyq, a, grid, st);
    else if (plan.last_bin == 3) bd_gr_launch<3>(gran, nyq, a, grid, st);
    else if (plan.last_bin == 4) bd_gr_launch<4>(gran, nyq, a, grid, st);
    else if (plan.last_bin == 5) bd_gr_launch<5>(gran, nyq, a, grid, st);
    else if (plan.last_bin <= 8) { if (gran == 64) bd_gr_launch2<8, 64>(nyq, a, grid, st); else bd_gr_launch2<8, 32>(nyq, a, grid, st); }
    else bd_gr_launch2<11, 64>(nyq, a, grid, st);
    bd_set_last_kernel("bd_spectrum_granule_kernel");
    return bd_check_hip(hipGetLastError(), "launch bd_spectrum_granule_kernel");
}
