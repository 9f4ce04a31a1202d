// Extension analysis (no counterpart in the reference, whose low-pass is first
// order and restarts in every window: src/strategies/lpf.rs:66-71): envelope of a
// second-order low-pass whose state is carried through the whole stream.
//
// The recurrence is evaluated in chunks of 64 samples so that the chunk-to-chunk
// dependence becomes a scan over 2x2 affine maps:
//   * one wavefront per stream, one step = 64 chunks = 4096 samples = 8 KB of
//     contiguous input: coalesced 16-byte loads, transposed through a private LDS
//     tile (144-byte rows: conflict-free ds_write_b128 / ds_read_b128) so that lane
//     l owns chunk l; the next step's loads are in flight during the arithmetic;
//   * pass 1: FIR part u[n] and the zero-state response of the chunk, two samples per
//     step of the recurrence (packed fmas; u stays in registers);
//   * Hillis-Steele scan over the 64 chunk responses with the precomputed powers
//     P^(2^k) of the chunk transition, then e_l = P^(l+1) s_in + v_l gives the state
//     at every chunk boundary; s_in is carried from step to step in the wavefront;
//   * pass 2: the recurrence again from the true start state, max |y| per lane;
//   * per decision window (64 .. 4096 samples = 1 .. 64 lanes): max |x|, max |y| by
//     a butterfly, threshold, gate; results staged in LDS and stored contiguously.
// Every f32 operation follows the CPU definition of the analysis, so values and
// decisions are bit-identical to it.  HBM-bound by design: each sample is read
// once, about 10 VALU instructions per sample.
#include "bd_internal.h"

#define BD_BQ_ROW 36 /* dwords per LDS row: 128 B of samples + 16 B pad */
#ifndef BD_BQ_WAVES
#define BD_BQ_WAVES 4 /* wavefronts per SIMD the register budget is sized for */
#endif

struct bd_bq_args {
    const int16_t *samples;
    float         *values;
    uint8_t       *flags;
    uint64_t       stride, out_stride, n_streams, n_windows, n_used;
    uint32_t       misaligned;       // rows are not 16-byte aligned: loads at 2-byte-aligned addresses
    uint32_t       lanes_per_window; // window_len / 64
    uint32_t       min_amp;
    float          ratio;
    bd_biquad_design d;
};

// 16 bytes of samples; rows that are not 16-byte aligned are read at their 2-byte-aligned
// addresses (global memory allows it, at a lower rate)
__device__ __forceinline__ uint4 bd_bq_load16(const int16_t *p, bool misaligned)
{
    if (!misaligned) return bd_load_stream16(p);
    typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
    const bd_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a2 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ void bd_bq_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// (M v + w) with the fma pattern of the definition
__device__ __forceinline__ void bd_bq_affine(const float m0, const float m1, const float m2, const float m3,
                                             float va, float vb, float wa, float wb, float &oa, float &ob)
{
    oa = __builtin_fmaf(m1, vb, __builtin_fmaf(m0, va, wa));
    ob = __builtin_fmaf(m3, vb, __builtin_fmaf(m2, va, wb));
}

typedef float v2f __attribute__((ext_vector_type(2)));

// Two samples of the recurrence from prev = (y[2i-2], y[2i-1]):
//   y[2i]   = fma(na1, p1, fma(na2, p2, u[2i]))
//   y[2i+1] = fma(c1b, p1, fma(c2b, p2, w_i)),  w_i = fma(na1, u[2i], u[2i+1])
// as two packed fmas; c2 = (na2, c2b), c1 = (na1, c1b) (uniform: scalar register pairs),
// uw = (u[2i], w_i).  The op_sel modifiers broadcast one half of `prev` (the compiler does
// not form them).
// m = max(m, |p.x|, |p.y|).  Assembly, because the optimizer otherwise re-associates the
// chain of maxima into a tree evaluated after the loop and keeps every operand alive.
__device__ __forceinline__ void bd_bq_absmax(float &m, v2f p)
{
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(p.x), "v"(p.y));
}

__device__ __forceinline__ v2f bd_bq_pair(v2f c1, v2f c2, v2f prev, v2f uw)
{
    v2f y;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %4, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "=&v"(y) : "s"(c2), "v"(prev), "v"(uw), "s"(c1));
    return y;
}

__global__ __launch_bounds__(256, BD_BQ_WAVES) void bd_biquad_env_kernel(bd_bq_args a)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile_all[4][64 * BD_BQ_ROW];
    __shared__ float out_v[4][128];
    __shared__ uint8_t out_f[4][128];

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t s = bd_block_index() * 4 + wave;
    if (s >= a.n_streams) return;
    uint32_t *tile = tile_all[wave];
    const int16_t *sp = a.samples + s * a.stride;
    const uint64_t n_steps = (a.n_used + 4095) / 4096;
    const uint64_t last_unit = a.n_used / 8 - 1; // last 16-byte unit that belongs to a window

    const float b0 = a.d.b0, b1 = a.d.b1, b2 = a.d.b2, na1 = a.d.na1, na2 = a.d.na2;
    const v2f c1v = v2f{na1, a.d.c1b}, c2v = v2f{na2, a.d.c2b};
    const v2f b0v = v2f{b0, b0}, b2v = v2f{b2, b2};

    // loads of one step: 16-byte unit j * 64 + lane of the step's 512 units.  A step that
    // lies inside the analysed range loads from one base address; in the last, partial step
    // units past the range are replaced by the last valid one (their chunks are never
    // reported and cannot influence earlier ones).
    const bool mis = a.misaligned != 0;
    uint4 ld[8];
    auto load_step = [&](uint64_t st) {
        if ((st + 1) * 4096 <= a.n_used) {
            const int16_t *base = sp + st * 4096 + lane * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) ld[j] = bd_bq_load16(base + j * 512, mis);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint64_t unit = st * 512 + (uint64_t)(j * 64 + lane);
                unit = unit < last_unit ? unit : last_unit;
                ld[j] = bd_bq_load16(sp + unit * 8, mis);
            }
        }
    };
    load_step(0);
    uint4 *wr = reinterpret_cast<uint4 *>(tile + (lane >> 3) * BD_BQ_ROW + (lane & 7) * 4); // + j * 8 rows
    const uint4 *rd = reinterpret_cast<const uint4 *>(tile + lane * BD_BQ_ROW);

    float sa = 0.f, sb = 0.f;  // state at the start of the step (uniform)
    uint32_t carry_x = 0;      // last two samples of the previous step
    uint32_t pend = 0;         // results staged in LDS
    uint64_t win_base = 0;     // index of the first staged window
    const uint32_t lpw = a.lanes_per_window, wins_per_step = 64 / lpw;

    for (uint64_t st = 0; st < n_steps; st++) {
        bd_bq_sync();
#pragma unroll
        for (int j = 0; j < 8; j++) wr[j * 8 * (BD_BQ_ROW / 4)] = ld[j];
        bd_bq_sync();
        uint32_t raw[32];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint4 r = rd[c];
            raw[4 * c] = r.x; raw[4 * c + 1] = r.y; raw[4 * c + 2] = r.z; raw[4 * c + 3] = r.w;
        }
        // the two samples before the chunk
        uint32_t prevd = (uint32_t)__shfl_up((int)raw[31], 1, 64);
        if (lane == 0) prevd = carry_x;
        carry_x = (uint32_t)__builtin_amdgcn_readlane((int)raw[31], 63);
        v2f xp = v2f{(float)(short)(prevd & 0xffff), (float)(short)(prevd >> 16)}; // (x[n-2], x[n-1])

        // pass 1: FIR part and zero-state response, two samples per step of the recurrence
        v2f uw[32];                 // (u[2i], w_i)
        v2f zp = v2f{0.f, 0.f};     // (z[2i-2], z[2i-1])
        float mxf = 0.f;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int d = (int)raw[i];
            const v2f xc = v2f{(float)(short)(d & 0xffff), (float)(short)(d >> 16)};
            // u[n] = fma(b1, x[n-1], fma(b2, x[n-2], b0 * x[n])): the b0 and b2 terms of two samples packed
            const v2f t = __builtin_elementwise_fma(b2v, xp, b0v * xc);
            const float u0 = __builtin_fmaf(b1, xp.y, t.x), u1 = __builtin_fmaf(b1, xc.x, t.y);
            uw[i] = v2f{u0, __builtin_fmaf(na1, u0, u1)};
            bd_bq_absmax(mxf, xc);
            zp = bd_bq_pair(c1v, c2v, zp, uw[i]);
            xp = xc;
        }
        // the raw samples are dead: their registers take the next step's loads, which stay in
        // flight during the scan and the second pass
        if (st + 1 < n_steps) load_step(st + 1);
        // scan over the chunk responses
        float va = zp.y, vb = zp.x;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const float pa = __shfl_up(va, 1 << k, 64), pb = __shfl_up(vb, 1 << k, 64);
            float ta, tb;
            bd_bq_affine(a.d.pw[k][0], a.d.pw[k][1], a.d.pw[k][2], a.d.pw[k][3], pa, pb, va, vb, ta, tb);
            if (lane >= (1 << k)) { va = ta; vb = tb; }
        }
        float ea, eb;
        bd_bq_affine(a.d.q[lane][0], a.d.q[lane][1], a.d.q[lane][2], a.d.q[lane][3], sa, sb, va, vb, ea, eb);
        float y1 = __shfl_up(ea, 1, 64), y2 = __shfl_up(eb, 1, 64);
        if (lane == 0) { y1 = sa; y2 = sb; }
        sa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ea), 63));
        sb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eb), 63));
        // pass 2: the chunk from its true start state
        float myf = 0.f;
        v2f yp = v2f{y2, y1};
#pragma unroll
        for (int i = 0; i < 32; i++) {
            yp = bd_bq_pair(c1v, c2v, yp, uw[i]);
            bd_bq_absmax(myf, yp);
        }
        // decision windows: maxima over the lanes of a window
        for (uint32_t o = 1; o < lpw; o <<= 1) {
            mxf = fmaxf(mxf, __shfl_xor(mxf, (int)o, 64));
            myf = fmaxf(myf, __shfl_xor(myf, (int)o, 64));
        }
        const int mx = (int)fminf(mxf, 32767.f);       // -32768 counts as 32767
        const int thr = (int)(a.ratio * (float)mx);    // truncation
        const bool flag = (uint32_t)mx >= a.min_amp && myf >= (float)thr;
        if ((lane & (lpw - 1)) == 0) {
            const uint32_t slot = pend + (uint32_t)lane / lpw;
            out_v[wave][slot] = myf;
            out_f[wave][slot] = flag ? 1 : 0;
        }
        pend += wins_per_step;
        if (pend >= 64 || st + 1 == n_steps) {
            bd_bq_sync();
            uint64_t n_out = a.n_windows - win_base;
            if (n_out > pend) n_out = pend;
            const uint64_t o0 = s * a.out_stride + win_base;
            if (a.values) {
                __amdgpu_buffer_rsrc_t r = bd_out_rsrc(a.values + o0, (unsigned)n_out * 4u);
                for (unsigned i = lane; i < n_out; i += 64) bd_store_stream_f32(r, i * 4u, out_v[wave][i]);
            }
            {
                __amdgpu_buffer_rsrc_t r = bd_out_rsrc(a.flags + o0, (unsigned)n_out);
                for (unsigned i = lane; i < n_out; i += 64) bd_store_stream_u8(r, i, out_f[wave][i]);
            }
            win_base += pend;
            pend = 0;
        }
    }
}

// ---- host side --------------------------------------------------------------
static void bd_mat2_mul(const double a[4], const double b[4], double c[4])
{
    const double r0 = a[0] * b[0] + a[1] * b[2], r1 = a[0] * b[1] + a[1] * b[3];
    const double r2 = a[2] * b[0] + a[3] * b[2], r3 = a[2] * b[1] + a[3] * b[3];
    c[0] = r0; c[1] = r1; c[2] = r2; c[3] = r3;
}

extern "C" void bd_biquad_default_params(bd_biquad_params *p)
{
    if (!p) return;
    p->cutoff_hz = 120; p->q_factor = 0.70710678f; p->window_len = 1024; p->ratio = 0.77f; p->min_amplitude = 9830;
}

extern "C" int bd_biquad_design_lowpass(uint32_t sampling_rate, uint32_t cutoff_hz, float q_factor, bd_biquad_design *d)
{
    if (!d || sampling_rate == 0 || cutoff_hz == 0 || 2.0 * cutoff_hz >= (double)sampling_rate || !(q_factor > 0.0f)) {
        bd_set_error("biquad design: need 0 < cutoff < sampling_rate / 2 and q_factor > 0");
        return BD_ERR_INVALID_ARG;
    }
    // RBJ low-pass in float64, rounded once; chunk transition powers from the rounded coefficients
    const double pi = 3.14159265358979323846;
    const double w0 = 2.0 * pi * (double)cutoff_hz / (double)sampling_rate;
    const double cw = cos(w0), sw = sin(w0), alpha = sw / (2.0 * (double)q_factor);
    const double a0 = 1.0 + alpha;
    d->b0 = (float)(((1.0 - cw) / 2.0) / a0);
    d->b1 = (float)((1.0 - cw) / a0);
    d->b2 = d->b0;
    d->na1 = (float)((2.0 * cw) / a0);
    d->na2 = (float)(-(1.0 - alpha) / a0);
    d->c1b = (float)((double)d->na1 * (double)d->na1 + (double)d->na2);
    d->c2b = (float)((double)d->na1 * (double)d->na2);
    const double am[4] = {(double)d->na1, (double)d->na2, 1.0, 0.0};
    double p[4] = {1.0, 0.0, 0.0, 1.0};
    for (int i = 0; i < 64; i++) bd_mat2_mul(am, p, p);
    double m[4] = {p[0], p[1], p[2], p[3]};
    for (int k = 0; k < 6; k++) {
        for (int j = 0; j < 4; j++) d->pw[k][j] = (float)m[j];
        bd_mat2_mul(m, m, m);
    }
    double ql[4] = {p[0], p[1], p[2], p[3]};
    for (int l = 0; l < 64; l++) {
        for (int j = 0; j < 4; j++) d->q[l][j] = (float)ql[j];
        bd_mat2_mul(p, ql, ql);
    }
    return BD_OK;
}

extern "C" int bd_biquad_envelope(uint32_t sampling_rate, const bd_biquad_params *params, const bd_biquad_design *design,
                                  const int16_t *samples, uint64_t n_streams, uint64_t n_samples, uint64_t stream_stride,
                                  float *values, uint8_t *flags, uint64_t out_stride, void *hip_stream)
{
    bd_biquad_params p;
    if (params) p = *params; else bd_biquad_default_params(&p);
    const uint32_t w = p.window_len;
    if (w < 64 || w > 4096 || (w & (w - 1))) {
        bd_set_error("biquad envelope: window_len must be a power of two between 64 and 4096");
        return BD_ERR_UNSUPPORTED;
    }
    bd_bq_args a;
    if (design) a.d = *design;
    else {
        int st = bd_biquad_design_lowpass(sampling_rate, p.cutoff_hz, p.q_factor, &a.d);
        if (st != BD_OK) return st;
    }
    const uint64_t n_win = n_samples / w;
    if (n_win == 0 || n_streams == 0) return BD_OK;
    if (!samples || !flags || stream_stride < n_samples) { bd_set_error("biquad envelope: bad buffers"); return BD_ERR_INVALID_ARG; }
    if ((uintptr_t)samples % 2) { bd_set_error("biquad envelope: samples must be 2-byte aligned"); return BD_ERR_INVALID_ARG; }
    if (out_stride == 0) out_stride = n_win;
    if (out_stride < n_win) { bd_set_error("out_stride < windows per stream"); return BD_ERR_INVALID_ARG; }
    a.samples = samples; a.values = values; a.flags = flags;
    a.stride = stream_stride; a.out_stride = out_stride; a.n_streams = n_streams;
    a.n_windows = n_win; a.n_used = n_win * w;
    a.misaligned = (((uintptr_t)samples % 16) || (stream_stride % 8)) ? 1u : 0u;
    a.lanes_per_window = w / 64; a.min_amp = p.min_amplitude; a.ratio = p.ratio;
    const uint64_t grid64 = (n_streams + 3) / 4;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("too many streams for one launch"); return BD_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(bd_biquad_env_kernel, grid_dims, dim3(256), 0, (hipStream_t)hip_stream, a);
    bd_set_last_kernel("bd_biquad_env_kernel");
    return bd_check_hip(hipGetLastError(), "launch bd_biquad_env_kernel");
}
