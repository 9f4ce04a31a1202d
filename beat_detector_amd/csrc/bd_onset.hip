// Extension analysis (no counterpart in the reference, which keeps three raw
// FFT bins and a constant threshold: src/strategies/spectrum.rs:82-92):
// Hann-windowed STFT of 1024 or 2048 points with hop = half or a quarter of the length,
// low-band energy, spectral flux, adaptive-threshold peak picking.
//
// STFT kernel.  One wavefront owns a run of consecutive frames of one stream and
// computes each frame's real FFT entirely on chip:
//   * N real samples are treated as M = N/2 complex points, 64 lanes x E points
//     (E = 8 for N = 1024, E = 16 for N = 2048);
//   * radix-2 decimation-in-time data flow in three register phases
//     (4+4+2 stages for E = 16, 3+3+3 for E = 8) with two exchanges through a
//     padded LDS tile; the bit-reversal is folded into the global loads (lane L,
//     slot q loads complex point rev(q) * 64 + rev6(L): every load instruction
//     still covers 256 contiguous bytes), and the raw points of the quarters a
//     frame shares with the next one (two for hop = N/2, three for hop = N/4) are
//     kept in registers, so each sample is read from memory once;
//   * twiddles of the first phase are compile-time constants, those of the later
//     phases and of the post-processing are prepared per lane in registers; the
//     window comes from a table in LDS;
//   * complex arithmetic on packed f32 instructions (a general butterfly is three);
//   * real-input post-processing pairs Z[k] with Z[M-k] through LDS, magnitudes
//     stay in registers for the next frame's flux; sums over bins are per-lane
//     partial sums + a butterfly on the DPP network.
// Every f32 operation is ordered exactly as in the CPU definition of the
// analysis, so spectra, flux and band energy are reproduced bit for bit.
// This kernel is bound by VALU issue, not by HBM (about 740 VALU instructions per
// 2048-point frame and wavefront).
#include "bd_internal.h"
#include "bd_ext_tables.inc"

struct bd_onset_args {
    const int16_t *samples;
    float         *flux;        // [n_streams][out_stride] or NULL
    float         *band;        // [n_streams][out_stride] or NULL
    uint64_t       stride;      // samples between streams
    uint64_t       out_stride;  // frames between output rows
    uint64_t       n_frames;    // frames per stream
    uint64_t       n_tasks;     // n_streams * runs_per_stream
    uint32_t       run_frames;  // frames per task (<= BD_ONSET_MAX_RUN)
    uint32_t       runs_per_stream;
    uint32_t       kmax;        // band energy over bins 0 .. kmax
    uint32_t       window;      // 1 = Hann
};

#define BD_ONSET_MAX_RUN 64

typedef float v2f __attribute__((ext_vector_type(2)));

template <int E> struct bd_fft_cfg {
    static constexpr int LGE = (E == 16) ? 4 : 3;
    static constexpr int M = 64 * E;           // complex points
    static constexpr int N = 2 * M;            // real samples per frame
    static constexpr int LGN = LGE + 7;
    static constexpr int NS1 = LGE;            // stages per phase
    static constexpr int NS2 = (E == 16) ? 4 : 3;
    static constexpr int NS3 = (LGE + 6) - NS1 - NS2;
    // Complex tile of one wavefront.  Point p of the transform lives at p + (p >> PAD1) in
    // the exchange between phases 1 and 2 and at p + (p >> PAD2) in the exchange between
    // phases 2 and 3.  ds_write_b64 is banked on 32 dwords over groups of 16 lanes,
    // ds_read_b64 on 64 dwords over groups of 32 lanes: for N = 2048 one pad slot per 16
    // points makes the first exchange conflict-free (lanes 16 points apart write, lanes read
    // points 16 apart), one per 32 points the second (lanes write points 16 apart, 64
    // consecutive lanes read).  N = 1024 pads every 8 points in both (a few two-way
    // conflicts remain in phase 3).  The exchange with the mirror bin M - k uses the plain
    // index and one more slot (Z[M] := Z[0]).
    static constexpr int PAD1 = (E == 16) ? 4 : 3;
    static constexpr int PAD2 = (E == 16) ? 5 : 3;
    static constexpr int DATA_LEN = M + (M >> 3) + 1;
    static constexpr int ROW3 = 64 + (64 >> PAD2);  // padded distance of points 64 apart
};

// (cos, -sin)(2 pi idx / N): the butterfly twiddle
template <int E> __device__ __forceinline__ v2f bd_tw_fwd(int idx)
{
    if (E == 16) return v2f{BD_EXT_TW2048[idx][0], -BD_EXT_TW2048[idx][1]};
    return v2f{BD_EXT_TW1024[idx][0], -BD_EXT_TW1024[idx][1]};
}
// (cos, sin)(2 pi idx / N)
template <int E> __device__ __forceinline__ v2f bd_tw_cs(int idx)
{
    if (E == 16) return v2f{BD_EXT_TW2048[idx][0], BD_EXT_TW2048[idx][1]};
    return v2f{BD_EXT_TW1024[idx][0], BD_EXT_TW1024[idx][1]};
}

__device__ __forceinline__ int bd_rev_bits(int v, int bits)
{
    return (int)(__builtin_bitreverse32((unsigned)v) >> (32 - bits));
}

// Packed-f32 helpers.  The VOP3P source modifiers (op_sel picks the half of a
// register pair, neg_lo / neg_hi negate one half) make a general butterfly three
// instructions; the compiler does not form them from vector shuffles, hence the
// inline assembly.  All are single IEEE operations per half, in the order the
// arithmetic specification gives.
//
// General butterfly with twiddle w = (wr, wi):
//     u = (fma(b.re, wr, a.re), fma(b.re, wi, a.im))
//     X = (fma(-b.im, wi, u.re), fma(b.im, wr, u.im))          = a + w b
//     Y = (fma(2, a.re, -X.re), fma(2, a.im, -X.im))           = a - w b
__device__ __forceinline__ void bd_bfly_w(v2f &a, v2f &b, v2f w)
{
    v2f x, y;
    asm("v_pk_fma_f32 %0, %2, %3, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(x) : "v"(a), "v"(b), "v"(w));
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(y) : "v"(a), "v"(x));
    a = x; b = y;
}
// butterfly with twiddle (1, 0): X = a + b, Y = a - b
__device__ __forceinline__ void bd_bfly_one(v2f &a, v2f &b)
{
    const v2f x = a, t = b;
    a = x + t;
    b = x - t;
}
// butterfly with t = (b.im, -b.re)  (twiddle (0, -1))
__device__ __forceinline__ void bd_bfly_rot(v2f &a, v2f &b)
{
    v2f ra, rb;
    asm("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        : "=&v"(ra), "=&v"(rb) : "v"(a), "v"(b));
    a = ra; b = rb;
}
// Real-input post-processing of one bin (spectrum at twice its value):
// z = Z[k], q = Z[M - k], cs = (cos, sin)(2 pi k / N); returns (Y.re, Y.im) with
//     e = (z.re + q.re, z.im - q.im),  d = (z.re - q.re, z.im + q.im)
//     Y.re = fma(-s, d.re, fma(c, d.im, e.re)),  Y.im = fma(-s, d.im, fma(-c, d.re, e.im))
__device__ __forceinline__ v2f bd_post_bin(v2f z, v2f q, v2f cs)
{
    v2f e, d, y;
    asm("v_pk_add_f32 %0, %3, %4 neg_hi:[0,1]\n\t"                                              // e
        "v_pk_add_f32 %1, %3, %4 neg_lo:[0,1]\n\t"                                              // d
        "v_pk_fma_f32 %2, %5, %1, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]\n\t"        // (c di + er, -c dr + ei)
        "v_pk_fma_f32 %2, %5, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" // (-s dr + ., -s di + .)
        : "=&v"(e), "=&v"(d), "=&v"(y) : "v"(z), "v"(q), "v"(cs));
    return y;
}

// Stages of the first phase (twiddle indices known at compile time).  Point q of
// the lane, global stage s = t + 1; twiddles (1,0) and (0,-1) take the two-instruction forms
// (in these stages only: the later ones use the general butterfly for every twiddle).
template <int E>
__device__ __forceinline__ void bd_fft_phase1(v2f *z)
{
    using C = bd_fft_cfg<E>;
#pragma unroll
    for (int t = 0; t < C::NS1; t++) {
#pragma unroll
        for (int i = 0; i < E; i++) {
            if (i & (1 << t)) continue;
            const int ib = i + (1 << t);
            const int idx = (i & ((1 << t) - 1)) << (C::LGN - (t + 1));
            if (idx == 0) bd_bfly_one(z[i], z[ib]);
            else if (idx == C::N / 4) bd_bfly_rot(z[i], z[ib]);
            else bd_bfly_w(z[i], z[ib], bd_tw_fwd<E>(idx));
        }
    }
}

// NS stages on 2^NS points in registers (stride STRIDE in the array), twiddles
// tw[(1 << t) - 1 + jl] prepared per lane.
template <int NS, int STRIDE>
__device__ __forceinline__ void bd_fft_phase(v2f *z, const v2f *tw)
{
    constexpr int H = 1 << (NS - 1);
#pragma unroll
    for (int t = 0; t < NS; t++) {
        // pair n: a = ((n >> t) << (t + 1)) + (n & ((1 << t) - 1)), b = a + (1 << t), twiddle jl = n & ((1 << t) - 1)
#pragma unroll
        for (int n = 0; n < H; n++) {
            const int jl = n & ((1 << t) - 1), ia = (((n >> t) << (t + 1)) + jl) * STRIDE, ib = ia + (1 << t) * STRIDE;
            bd_bfly_w(z[ia], z[ib], tw[(1 << t) - 1 + jl]);
        }
    }
}

// Correctly rounded square root of x >= 2^-96.  v_sqrt_f32 is within one ulp
// for normal arguments: choose among the result and its two neighbours by the
// sign of the exact residuals.
__device__ __forceinline__ float bd_sqrt_rn_normal(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const int si = __builtin_bit_cast(int, s);
    const float sd = __builtin_bit_cast(float, si - 1), su = __builtin_bit_cast(float, si + 1);
    // sd * s - x and su * s - x: the exact residuals with their signs turned, so that "residual > 0" is the
    // sign bit (an exact zero comes out as +0) and the choice is two shifts and adds, no compare / select
    // (assembly keeps the negation a source modifier of a scalar fma)
    float nd, nu;
    asm("v_fma_f32 %0, %2, %4, -%5\n\tv_fma_f32 %1, %3, %4, -%5" : "=&v"(nd), "=&v"(nu) : "v"(sd), "v"(su), "v"(s), "v"(x));
    const unsigned up = (__builtin_bit_cast(unsigned, nd) >> 31) + (__builtin_bit_cast(unsigned, nu) >> 31);
    return __builtin_bit_cast(float, si - 1 + (int)up);
}

// Sum over the wavefront as the butterfly p[l] += p[l ^ o], o = 1, 2, 4, 8, 16, 32,
// on the DPP network; the total is valid in lane 63.  After the steps o = 1, 2 all
// lanes of a quad agree, so the mirror patterns (l ^ 7, l ^ 15) deliver the same
// values as l ^ 4, l ^ 8; the row broadcasts add row r - 1 into rows 1, 3 and
// rows 0 - 1 into row 3.
__device__ __forceinline__ float bd_butterfly_sum_lane63(float p)
{
    p = p + bd_dpp_f<0xB1>(p);  // quad_perm [1,0,3,2]
    p = p + bd_dpp_f<0x4E>(p);  // quad_perm [2,3,0,1]
    p = p + bd_dpp_f<0x141>(p); // row_half_mirror
    p = p + bd_dpp_f<0x140>(p); // row_mirror
    p = p + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x142, 0xa, 0xf, true)); // row_bcast:15 -> rows 1, 3
    p = p + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x143, 0xc, 0xf, true)); // row_bcast:31 -> rows 2, 3
    return p;
}

// One ds_read_b64 per point.  The volatile access keeps the compiler from pairing
// two of them into a ds_read2_b64, which moves the same bytes in twice the LDS
// cycles and is banked on 32 instead of 64 dwords.
__device__ __forceinline__ v2f bd_lds_read(const v2f *p)
{
    typedef const volatile __attribute__((address_space(3))) v2f *lds_ptr;
    return *(lds_ptr)p;
}

__device__ __forceinline__ void bd_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// HQ: hop in quarters of the frame (2: hop = N / 2, 1: hop = N / 4).
// PWR: flux and band energy from the squared magnitudes (no square roots: a sixth of the
// frame's instructions are the correctly rounded roots).
template <int E, int HQ, bool PWR>
__device__ __forceinline__ void bd_stft_flux_body(const bd_onset_args &a)
{
    using C = bd_fft_cfg<E>;
    constexpr int M = C::M, LGE = C::LGE, N = C::N, LGN = C::LGN;
    constexpr int HOP = M * HQ / 2;            // samples between frames
    __shared__ v2f tile_all[4][C::DATA_LEN];
    __shared__ v2f win_lds[E][64];   // window at samples 2 n, 2 n + 1 of point n = slot q of a lane
    // (cos, sin)(2 pi (lane + 64 t) / N) of the post-processing: in LDS for N = 2048, where
    // registers decide the occupancy (3 waves per SIMD), per lane in registers for N = 1024
    constexpr bool TWP_LDS = (E == 16);
    __shared__ v2f twp_lds[TWP_LDS ? E : 1][64];
    // results of a run: N = 2048 keeps frame f0 + i in lane i (its LDS is what limits the blocks
    // per CU), N = 1024 stages them in LDS (its registers are what limits the waves per SIMD)
    __shared__ float out_lds[TWP_LDS ? 1 : 4][2][BD_ONSET_MAX_RUN];

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r6 = bd_rev_bits(lane, 6);
    for (int q = wave; q < E; q += 4) {
        const int nq = bd_rev_bits(q, LGE) * 64 + r6;
        const int i0 = 2 * nq, i1 = i0 + 1;
        v2f w;
        w.x = __builtin_fmaf(-0.5f, bd_tw_cs<E>(i0 <= M ? i0 : N - i0).x, 0.5f);
        w.y = __builtin_fmaf(-0.5f, bd_tw_cs<E>(i1 <= M ? i1 : N - i1).x, 0.5f);
        win_lds[q][lane] = a.window ? w : v2f{1.f, 1.f};
        if constexpr (TWP_LDS) twp_lds[q][lane] = bd_tw_cs<E>(lane + 64 * q);
    }
    __syncthreads();
    v2f *tile = tile_all[wave];
    const uint64_t task = bd_block_index() * 4 + wave;
    if (task >= a.n_tasks) return;
    const uint64_t s = task / a.runs_per_stream;
    const uint64_t f0 = (task - s * a.runs_per_stream) * a.run_frames;
    const uint64_t f_end = f0 + a.run_frames < a.n_frames ? f0 + a.run_frames : a.n_frames;
    const int16_t *sp = a.samples + s * a.stride;

    // per-lane twiddles of the second and third phase and of the post-processing
    constexpr int S2 = C::NS1, S3 = C::NS1 + C::NS2, CNT2 = 1 << C::NS2, CNT3 = 1 << C::NS3, GROUPS = E / CNT3;
    const int low = lane & ((1 << S2) - 1);
    const int high = lane >> S2;
    v2f tw2[CNT2 - 1], tw3[GROUPS][CNT3 - 1], twp[TWP_LDS ? 1 : E];
#pragma unroll
    for (int t = 0; t < C::NS2; t++)
#pragma unroll
        for (int jl = 0; jl < (1 << t); jl++)
            tw2[(1 << t) - 1 + jl] = bd_tw_fwd<E>((low + (jl << S2)) << (LGN - (S2 + t + 1)));
#pragma unroll
    for (int g = 0; g < GROUPS; g++)
#pragma unroll
        for (int t = 0; t < C::NS3; t++)
#pragma unroll
            for (int jl = 0; jl < (1 << t); jl++)
                tw3[g][(1 << t) - 1 + jl] = bd_tw_fwd<E>((lane + 64 * g + (jl << S3)) << (LGN - (S3 + t + 1)));
    if constexpr (!TWP_LDS) {
#pragma unroll
        for (int t = 0; t < E; t++) twp[t] = bd_tw_cs<E>(lane + 64 * t);
    }

    // LDS addresses (in points): every access below is one of these plus a constant
    constexpr int PAD1 = C::PAD1, PAD2 = C::PAD2;
    v2f *t1 = tile + E * lane + ((E * lane) >> PAD1);                         // phase 1 out: point E * lane + q
    const int base2 = (high << (S2 + C::NS2)) + low;                          // phase 2: point base2 + i * E
    const v2f *t2r = tile + base2 + (base2 >> PAD1);
    v2f *t2w = tile + base2 + (base2 >> PAD2);
    const v2f *t3 = tile + lane + (lane >> PAD2);                             // phase 3 in: point lane + 64 t
    v2f *t4 = tile + lane;                                                    // mirror exchange: Z[lane + 64 t]
    const v2f *t5 = tile + M - lane;                                          //                  Z[M - lane - 64 t]

    float keep_flux = 0.f, keep_band = 0.f;
    float prev[E], prev_m = 0.f; // doubled magnitudes of the previous frame: bins lane + 64 t, and bin M
#pragma unroll
    for (int t = 0; t < E; t++) prev[t] = 0.f;

    // Raw points (two i16 per dword) of the current frame, one register set per quarter of the
    // frame: slot q of the lane is point rev(q) * 64 + rev6(lane), so its quarter is given by the
    // two low bits of q.  The sets change roles from frame to frame instead of being copied.
    constexpr int EQ = E / 4;
    unsigned raw_a[EQ], raw_b[EQ], raw_c[EQ], raw_d[EQ];
    const uint64_t f_first = f0 > 0 ? f0 - 1 : 0; // one frame of warm-up gives the flux its predecessor
    // rows need no alignment beyond that of a sample: dword loads at 2-byte-aligned addresses
    typedef unsigned bd_u32_a2 __attribute__((aligned(2)));
    const bd_u32_a2 *fp = reinterpret_cast<const bd_u32_a2 *>(sp + f_first * (uint64_t)HOP) + r6;
#pragma unroll
    for (int m = 0; m < EQ; m++) {   // slots 4 m + 0, 2, 1, 3 = quarters 0, 1, 2, 3
        raw_a[m] = __builtin_nontemporal_load(fp + bd_rev_bits(4 * m, LGE) * 64);
        raw_b[m] = __builtin_nontemporal_load(fp + bd_rev_bits(4 * m + 2, LGE) * 64);
        raw_c[m] = __builtin_nontemporal_load(fp + bd_rev_bits(4 * m + 1, LGE) * 64);
        raw_d[m] = __builtin_nontemporal_load(fp + bd_rev_bits(4 * m + 3, LGE) * 64);
    }
    float prev_b[E];

    // One frame.  r0 .. r3: raw points of its four quarters; afterwards the sets that the next frame
    // no longer needs (r0, and r1 for hop = N / 2) hold the next frame's last quarter(s).  pin:
    // magnitudes of the previous frame, pout: those of this one.
    auto frame = [&](const uint64_t f, unsigned (&r0)[EQ], unsigned (&r1)[EQ], unsigned (&r2)[EQ], unsigned (&r3)[EQ],
                     float (&pin)[E], float (&pout)[E]) __attribute__((always_inline)) {
        v2f z[E];
#pragma unroll
        for (int q = 0; q < E; q++) {
            const int c = ((q & 1) << 1) | ((q >> 1) & 1);
            const int d = (int)(c == 0 ? r0[q >> 2] : c == 1 ? r1[q >> 2] : c == 2 ? r2[q >> 2] : r3[q >> 2]);
            const v2f x = v2f{(float)(short)(d & 0xffff), (float)(short)(d >> 16)};
            z[q] = bd_lds_read(&win_lds[q][lane]) * x;   // (exact for the rectangular window: 1 * x)
        }
        // the quarters the next frame adds
        if (f + 1 < f_end) {
            const bd_u32_a2 *np = reinterpret_cast<const bd_u32_a2 *>(sp + (f + 1) * (uint64_t)HOP) + r6;
#pragma unroll
            for (int m = 0; m < EQ; m++) {
                if (HQ == 2) {
                    r0[m] = __builtin_nontemporal_load(np + bd_rev_bits(4 * m + 1, LGE) * 64);   // its quarter 2
                    r1[m] = __builtin_nontemporal_load(np + bd_rev_bits(4 * m + 3, LGE) * 64);   // its quarter 3
                } else {
                    r0[m] = __builtin_nontemporal_load(np + bd_rev_bits(4 * m + 3, LGE) * 64);   // its quarter 3
                }
            }
        }
        // phase 1: stages 1 .. NS1 on points E*lane .. E*lane + E - 1 (bit-reversed order)
        bd_fft_phase1<E>(z);
        bd_wave_lds_sync();
#pragma unroll
        for (int q = 0; q < E; q++) t1[q] = z[q];
        bd_wave_lds_sync();
        // phase 2: stages NS1+1 .. NS1+NS2 on points base + i * E
        {
            v2f y[CNT2];
#pragma unroll
            for (int i = 0; i < CNT2; i++) y[i] = bd_lds_read(t2r + i * E + ((i * E) >> PAD1));
            bd_fft_phase<C::NS2, 1>(y, tw2);
            bd_wave_lds_sync();
#pragma unroll
            for (int i = 0; i < CNT2; i++) t2w[i * E + ((i * E) >> PAD2)] = y[i];
            bd_wave_lds_sync();
        }
        // phase 3: the last NS3 stages; afterwards slot t holds Z[lane + 64 t]
#pragma unroll
        for (int t = 0; t < E; t++) z[t] = bd_lds_read(t3 + t * C::ROW3);
        // group g: points (lane + 64 g) + i * 2^S3  ->  slots g + GROUPS * i
#pragma unroll
        for (int g = 0; g < GROUPS; g++) bd_fft_phase<C::NS3, GROUPS>(z + g, tw3[g]);
        bd_wave_lds_sync();
#pragma unroll
        for (int t = 0; t < E; t++) t4[64 * t] = z[t];
        if (lane == 0) tile[M] = z[0]; // Z[M] := Z[0], the partner of bin 0
        bd_wave_lds_sync();
        // real-input post-processing (spectrum carried at twice its value): squared magnitudes
        float xs[E], xs_m, x_min;
#pragma unroll
        for (int t = 0; t < E; t++) {
            const v2f y = bd_post_bin(z[t], bd_lds_read(t5 - 64 * t), TWP_LDS ? bd_lds_read(&twp_lds[t][lane]) : twp[TWP_LDS ? 0 : t]);
            const float yi2 = y.y * y.y;
            xs[t] = __builtin_fmaf(y.x, y.x, yi2);
            x_min = t == 0 ? xs[0] : fminf(x_min, xs[t]);
        }
        {   // bin M pairs Z[0] with itself, twiddle (cos pi, sin pi) = (-1, 0)
            const v2f z0 = tile[0];
            const v2f y = bd_post_bin(z0, z0, bd_tw_cs<E>(M));
            const float yi2 = y.y * y.y;
            xs_m = __builtin_fmaf(y.x, y.x, yi2);
            x_min = fminf(x_min, xs_m);
        }
        // magnitudes; a wavefront with a tiny or zero argument takes the library square root
        if (PWR) {
            // flux of the power spectrum: xs stays 4 |X|^2
        } else if (__builtin_amdgcn_ballot_w64(x_min < 0x1p-96f) != 0) {
#pragma unroll
            for (int t = 0; t < E; t++) xs[t] = __builtin_sqrtf(xs[t]);
            xs_m = __builtin_sqrtf(xs_m);
        } else {
#pragma unroll
            for (int t = 0; t < E; t++) xs[t] = bd_sqrt_rn_normal(xs[t]);
            xs_m = bd_sqrt_rn_normal(xs_m);
        }
        // lane l sums its bins l, l + 64, ... in ascending order (zero terms of the band sum skipped)
        float flux_p = 0.f, band_p = (uint32_t)lane <= a.kmax ? (PWR ? xs[0] : xs[0] * xs[0]) : 0.f;
        if (__builtin_expect(a.kmax >= 64, 0)) {
#pragma unroll
            for (int t = 1; t < E; t++) band_p += (uint32_t)(lane + 64 * t) <= a.kmax ? (PWR ? xs[t] : xs[t] * xs[t]) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < E; t++) {
            const float d = xs[t] - pin[t];
            flux_p += fmaxf(d, 0.f);
            pout[t] = xs[t];
        }
        if (lane == 0) {
            const float d = xs_m - prev_m;
            flux_p += d > 0.f ? d : 0.f;
            band_p += (uint32_t)M <= a.kmax ? (PWR ? xs_m : xs_m * xs_m) : 0.f;
        }
        prev_m = xs_m;
        const float flux = (PWR ? 0.25f : 0.5f) * bd_butterfly_sum_lane63(flux_p);
        const float band = 0.25f * bd_butterfly_sum_lane63(band_p);
        // lane i keeps the results of frame f0 + i (run_frames <= 64)
        const float flux_u = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(flux), 63));
        const float band_u = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(band), 63));
        if constexpr (TWP_LDS) {
            if (f >= f0 && (uint64_t)lane == f - f0) { keep_flux = flux_u; keep_band = band_u; }
        } else if (f >= f0 && lane == 63) {
            out_lds[wave][0][f - f0] = flux;
            out_lds[wave][1][f - f0] = band;
        }
    };
    for (uint64_t f = f_first; f < f_end; f += 2) {
        frame(f, raw_a, raw_b, raw_c, raw_d, prev, prev_b);
        if (f + 1 >= f_end) break;
        if (HQ == 2) {
            frame(f + 1, raw_c, raw_d, raw_a, raw_b, prev_b, prev);
        } else {
            frame(f + 1, raw_b, raw_c, raw_d, raw_a, prev_b, prev);
            // two frames on, the quarters are (c, d, a, b): rename by exchanging the sets
#pragma unroll
            for (int m = 0; m < EQ; m++) {
                const unsigned ta = raw_a[m], tb = raw_b[m];
                raw_a[m] = raw_c[m]; raw_b[m] = raw_d[m]; raw_c[m] = ta; raw_d[m] = tb;
            }
        }
    }
    // results of the run, contiguous
    if constexpr (!TWP_LDS) {
        bd_wave_lds_sync();
        if ((unsigned)lane < (unsigned)(f_end - f0)) { keep_flux = out_lds[wave][0][lane]; keep_band = out_lds[wave][1][lane]; }
    }
    const unsigned n_res = (unsigned)(f_end - f0);
    const uint64_t o0 = s * a.out_stride + f0;
    if (a.flux) {
        __amdgpu_buffer_rsrc_t r = bd_out_rsrc(a.flux + o0, n_res * 4u);
        if ((unsigned)lane < n_res) bd_store_stream_f32(r, lane * 4u, keep_flux);
    }
    if (a.band) {
        __amdgpu_buffer_rsrc_t r = bd_out_rsrc(a.band + o0, n_res * 4u);
        if ((unsigned)lane < n_res) bd_store_stream_f32(r, lane * 4u, keep_band);
    }
}

// N = 2048 asks for three waves per SIMD (168 registers); N = 1024 leaves the
// register count to the compiler (about 110, four waves per SIMD).
template <int E, int HQ, bool PWR> __global__ void bd_stft_flux_kernel(bd_onset_args a);
template <> __global__ __launch_bounds__(256, 3) void bd_stft_flux_kernel<16, 2, false>(bd_onset_args a) { bd_stft_flux_body<16, 2, false>(a); }
template <> __global__ __launch_bounds__(256, 3) void bd_stft_flux_kernel<16, 1, false>(bd_onset_args a) { bd_stft_flux_body<16, 1, false>(a); }
template <> __global__ __launch_bounds__(256) void bd_stft_flux_kernel<8, 2, false>(bd_onset_args a) { bd_stft_flux_body<8, 2, false>(a); }
template <> __global__ __launch_bounds__(256) void bd_stft_flux_kernel<8, 1, false>(bd_onset_args a) { bd_stft_flux_body<8, 1, false>(a); }
template <> __global__ __launch_bounds__(256, 3) void bd_stft_flux_kernel<16, 2, true>(bd_onset_args a) { bd_stft_flux_body<16, 2, true>(a); }
template <> __global__ __launch_bounds__(256, 3) void bd_stft_flux_kernel<16, 1, true>(bd_onset_args a) { bd_stft_flux_body<16, 1, true>(a); }
template <> __global__ __launch_bounds__(256) void bd_stft_flux_kernel<8, 2, true>(bd_onset_args a) { bd_stft_flux_body<8, 2, true>(a); }
template <> __global__ __launch_bounds__(256) void bd_stft_flux_kernel<8, 1, true>(bd_onset_args a) { bd_stft_flux_body<8, 1, true>(a); }

// ---- peak picking -----------------------------------------------------------
// One wavefront per stream.  Lane = frame: moving mean of flux[j-W .. j+W]
// (summed in index order), local maximum over [j-w, j+w] (ties: the earliest
// frame wins), threshold delta + lambda * mean; then the minimum-gap rule is
// replayed over the candidates in time order, as in the refractory pass.
struct bd_pick_args {
    const float *flux;
    uint8_t     *cand;        // [n_streams][out_stride] or NULL
    uint32_t    *count, *frame, *ms;
    uint64_t     out_stride, n_frames, n_streams;
    uint32_t     mean_frames, max_frames, min_gap_ms, max_onsets;
    uint32_t     fft_len, hop, sampling_rate, use_median;
    float        delta, lambda;
};

__global__ __launch_bounds__(256) void bd_peak_pick_kernel(bd_pick_args a)
{
    const int lane = threadIdx.x & 63;
    const uint64_t s = bd_block_index() * 4 + (threadIdx.x >> 6);
    if (s >= a.n_streams) return;
    const float *fl = a.flux + s * a.out_stride;
    const int64_t n = (int64_t)a.n_frames;
    uint32_t count = 0, last_ms = 0;
    bool have_last = false;
    for (int64_t base = 0; base < n; base += 64) {
        const int64_t j = base + lane;
        bool cand = false;
        if (j < n) {
            const int64_t lo = j > a.mean_frames ? j - a.mean_frames : 0;
            const int64_t hi = j + a.mean_frames < n - 1 ? j + a.mean_frames : n - 1;
            float sum = 0.f;
            for (int64_t i = lo; i <= hi; i++) sum += fl[i];
            float mean = sum / (float)(hi - lo + 1);
            if (a.use_median) { // element c/2 of the sorted window, found by counting ranks
                const int64_t c = hi - lo + 1;
                for (int64_t i = lo; i <= hi; i++) {
                    const float vi = fl[i];
                    int64_t rank = 0;
                    for (int64_t k = lo; k <= hi; k++) {
                        const float vk = fl[k];
                        rank += (vk < vi || (vk == vi && k < i)) ? 1 : 0;
                    }
                    if (rank == c / 2) mean = vi;
                }
            }
            const float thr = __builtin_fmaf(a.lambda, mean, a.delta);
            const int64_t mlo = j > a.max_frames ? j - a.max_frames : 0;
            const int64_t mhi = j + a.max_frames < n - 1 ? j + a.max_frames : n - 1;
            const float fj = fl[j];
            bool is_max = true;
            for (int64_t i = mlo; i <= mhi; i++) {
                const float fi = fl[i];
                if (i < j && fi >= fj) is_max = false;
                if (i > j && fi > fj) is_max = false;
            }
            cand = is_max && fj > thr;
            if (a.cand) a.cand[s * a.out_stride + j] = cand ? 1 : 0;
        }
        unsigned long long mask = __ballot(cand);
        while (mask) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint64_t jj = (uint64_t)base + b;
            const uint32_t t_ms = (uint32_t)(((jj * a.hop + a.fft_len / 2) * 1000ull) / a.sampling_rate);
            if (have_last && t_ms < last_ms + a.min_gap_ms) continue;
            have_last = true; last_ms = t_ms;
            if (lane == 0 && count < a.max_onsets) {
                if (a.frame) a.frame[s * a.max_onsets + count] = (uint32_t)jj;
                if (a.ms) a.ms[s * a.max_onsets + count] = t_ms;
            }
            count++;
        }
    }
    if (lane == 0) a.count[s] = count;
}

// ---- C ABI ------------------------------------------------------------------
extern "C" void bd_onset_default_params(bd_onset_params *p)
{
    if (!p) return;
    p->fft_len = 2048; p->hop = 1024; p->window = 1; p->band_max_hz = 90.0f;
    p->mean_frames = 8; p->max_frames = 3; p->delta = 0.0f; p->lambda = 1.5f; p->min_gap_ms = 100;
    p->use_median = 0;
    p->flux_power = 0;
}

extern "C" uint64_t bd_onset_frames(uint64_t n_samples, uint32_t fft_len, uint32_t hop)
{
    if (hop == 0 || n_samples < fft_len) return 0;
    return (n_samples - fft_len) / hop + 1;
}

static int bd_onset_check(const bd_onset_params *p, uint32_t sampling_rate)
{
    if (!p) { bd_set_error("onset parameters must not be NULL"); return BD_ERR_INVALID_ARG; }
    if ((p->fft_len != 1024 && p->fft_len != 2048) || (p->hop != p->fft_len / 2 && p->hop != p->fft_len / 4)) {
        bd_set_error("fft_len must be 1024 or 2048 and hop = fft_len / 2 or fft_len / 4");
        return BD_ERR_UNSUPPORTED;
    }
    if (sampling_rate == 0) { bd_set_error("sampling rate must be positive"); return BD_ERR_INVALID_ARG; }
    return BD_OK;
}

extern "C" int bd_onset_stft(uint32_t sampling_rate, const bd_onset_params *p, const int16_t *samples,
                             uint64_t n_streams, uint64_t n_samples, uint64_t stream_stride,
                             float *flux, float *band_energy, uint64_t out_stride, void *hip_stream)
{
    int st = bd_onset_check(p, sampling_rate);
    if (st != BD_OK) return st;
    const uint64_t n_frames = bd_onset_frames(n_samples, p->fft_len, p->hop);
    if (n_frames == 0 || n_streams == 0) return BD_OK;
    if (!samples || stream_stride < n_samples) { bd_set_error("bd_onset_stft: bad sample buffer"); return BD_ERR_INVALID_ARG; }
    if ((uintptr_t)samples % 2) { bd_set_error("bd_onset_stft: samples must be 2-byte aligned"); return BD_ERR_INVALID_ARG; }
    if (out_stride == 0) out_stride = n_frames;
    if (out_stride < n_frames) { bd_set_error("out_stride < frames per stream"); return BD_ERR_INVALID_ARG; }
    bd_onset_args a;
    a.samples = samples; a.flux = flux; a.band = band_energy;
    a.stride = stream_stride; a.out_stride = out_stride; a.n_frames = n_frames;
    uint32_t run = 64;
    while (run > 8 && n_streams * ((n_frames + run - 1) / run) < 16384) run >>= 1;
    a.run_frames = run;
    const uint64_t runs = (n_frames + run - 1) / run;
    if (runs > 0xffffffffull) { bd_set_error("stream too long"); return BD_ERR_UNSUPPORTED; }
    a.runs_per_stream = (uint32_t)runs;
    a.n_tasks = n_streams * runs;
    const float res = (float)sampling_rate / (float)p->fft_len;
    uint32_t kmax = 0;
    for (uint32_t k = 0; k <= p->fft_len / 2; k++) { if ((float)k * res <= p->band_max_hz) kmax = k; else break; }
    a.kmax = kmax; a.window = p->window ? 1u : 0u;
    const uint64_t grid64 = (a.n_tasks + 3) / 4;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("batch too large for one launch"); return BD_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)hip_stream;
    const bool half = p->hop == p->fft_len / 2;
#define BD_STFT_LAUNCH(EE, HH) do { \
        if (p->flux_power) hipLaunchKernelGGL((bd_stft_flux_kernel<EE, HH, true>), grid_dims, dim3(256), 0, s, a); \
        else hipLaunchKernelGGL((bd_stft_flux_kernel<EE, HH, false>), grid_dims, dim3(256), 0, s, a); } while (0)
    if (p->fft_len == 2048) { if (half) BD_STFT_LAUNCH(16, 2); else BD_STFT_LAUNCH(16, 1); }
    else { if (half) BD_STFT_LAUNCH(8, 2); else BD_STFT_LAUNCH(8, 1); }
#undef BD_STFT_LAUNCH
    bd_set_last_kernel("bd_stft_flux_kernel");
    return bd_check_hip(hipGetLastError(), "launch bd_stft_flux_kernel");
}

extern "C" int bd_onset_pick(uint32_t sampling_rate, const bd_onset_params *p, const float *flux,
                             uint64_t n_streams, uint64_t n_frames, uint64_t flux_stride,
                             uint8_t *candidates, uint32_t *onset_count, uint32_t *onset_frame,
                             uint32_t *onset_ms, uint32_t max_onsets, void *hip_stream)
{
    int st = bd_onset_check(p, sampling_rate);
    if (st != BD_OK) return st;
    if ((!flux && n_frames) || !onset_count) { bd_set_error("bd_onset_pick: flux and onset_count must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (n_streams == 0) return BD_OK;
    if (flux_stride == 0) flux_stride = n_frames;
    if (flux_stride < n_frames) { bd_set_error("bd_onset_pick: flux_stride < n_frames"); return BD_ERR_INVALID_ARG; }
    if (max_onsets == 0 && (onset_frame || onset_ms)) { bd_set_error("bd_onset_pick: max_onsets is 0"); return BD_ERR_INVALID_ARG; }
    bd_pick_args a;
    a.flux = flux; a.cand = candidates; a.count = onset_count; a.frame = onset_frame; a.ms = onset_ms;
    a.out_stride = flux_stride; a.n_frames = n_frames; a.n_streams = n_streams;
    a.mean_frames = p->mean_frames; a.max_frames = p->max_frames; a.min_gap_ms = p->min_gap_ms;
    a.max_onsets = max_onsets; a.fft_len = p->fft_len; a.hop = p->hop; a.sampling_rate = sampling_rate;
    a.delta = p->delta; a.lambda = p->lambda; a.use_median = p->use_median ? 1u : 0u;
    if (a.use_median && p->mean_frames > 32) { bd_set_error("bd_onset_pick: the median threshold supports mean_frames <= 32"); return BD_ERR_UNSUPPORTED; }
    const uint64_t grid64 = (n_streams + 3) / 4;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("too many streams for one launch"); return BD_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(bd_peak_pick_kernel, grid_dims, dim3(256), 0, (hipStream_t)hip_stream, a);
    return bd_check_hip(hipGetLastError(), "launch bd_peak_pick_kernel");
}

extern "C" int bd_onset_batch(uint32_t sampling_rate, const bd_onset_params *p, const int16_t *samples,
                              uint64_t n_streams, uint64_t n_samples, uint64_t stream_stride,
                              float *flux, float *band_energy, uint64_t out_stride,
                              uint32_t *onset_count, uint32_t *onset_frame, uint32_t *onset_ms,
                              uint32_t max_onsets, void *hip_stream)
{
    if (!p) { bd_set_error("onset parameters must not be NULL"); return BD_ERR_INVALID_ARG; }
    if (!flux && bd_onset_frames(n_samples, p->fft_len, p->hop)) {
        bd_set_error("bd_onset_batch: flux must not be NULL (peak picking reads it)");
        return BD_ERR_INVALID_ARG;
    }
    int st = bd_onset_stft(sampling_rate, p, samples, n_streams, n_samples, stream_stride, flux, band_energy,
                           out_stride, hip_stream);
    if (st != BD_OK) return st;
    const uint64_t n_frames = bd_onset_frames(n_samples, p->fft_len, p->hop);
    const char *k = bd_last_kernel_name();
    st = bd_onset_pick(sampling_rate, p, flux, n_streams, n_frames, out_stride ? out_stride : n_frames, nullptr,
                       onset_count, onset_frame, onset_ms, max_onsets, hip_stream);
    bd_set_last_kernel(k);
    return st;
}
