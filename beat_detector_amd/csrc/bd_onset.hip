// Extension analysis (no counterpart in the reference, which keeps three raw
// FFT bins and a constant threshold: src/strategies/spectrum.rs:82-92):
// Hann-windowed STFT of 1024 or 2048 points with hop = half the length,
// low-band energy, spectral flux, adaptive-threshold peak picking.
//
// STFT kernel.  One wavefront owns a run of consecutive frames of one stream and
// computes each frame's real FFT entirely on chip:
//   * N real samples are treated as M = N/2 complex points, 64 lanes x E points
//     (E = 8 for N = 1024, E = 16 for N = 2048);
//   * radix-2 decimation-in-time data flow in three register phases
//     (4+4+2 stages for E = 16, 3+3+3 for E = 8) with two exchanges through a
//     padded, bank-conflict-free LDS tile; the bit-reversal is folded into the
//     global loads (lane L, slot q loads complex point rev(q) * 64 + rev6(L): every
//     load instruction still covers 256 contiguous bytes), and because hop = N/2
//     the raw points of a frame's second half are kept in registers for the next
//     frame, so each sample is read from memory once;
//   * twiddles of the first phase are compile-time constants, the others come
//     from a padded table in LDS;
//   * real-input post-processing pairs Z[k] with Z[M-k] through LDS, magnitudes
//     stay in registers for the next frame's flux; sums over bins are per-lane
//     partial sums + a xor butterfly.
// Every f32 operation is ordered exactly as in the CPU definition of the
// analysis, so spectra, flux and band energy are reproduced bit for bit.
// This kernel is bound by VALU / LDS throughput, not by HBM (about 60 VALU
// operations per sample).
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

#define BD_ONSET_MAX_RUN 128

template <int E> struct bd_fft_cfg {
    static constexpr int LGE = (E == 16) ? 4 : 3;
    static constexpr int M = 64 * E;           // complex points
    static constexpr int N = 2 * M;            // real samples per frame
    static constexpr int LGN = LGE + 7;
    static constexpr int NS1 = LGE;            // stages per phase
    static constexpr int NS2 = (E == 16) ? 4 : 3;
    static constexpr int NS3 = (LGE + 6) - NS1 - NS2;
    static constexpr int DATA_LEN = M + (M >> LGE);             // padded complex tile
    static constexpr int TW_LEN = (N / 2 + 1) + ((N / 2 + 1) >> 5) + 1;
};

template <int E> __device__ __forceinline__ int bd_pad_d(int p) { return p + (p >> bd_fft_cfg<E>::LGE); }
__device__ __forceinline__ int bd_pad_t(int i) { return i + (i >> 5); }

template <int E> __device__ __forceinline__ float2 bd_tw_global(int idx)
{
    if (E == 16) return make_float2(BD_EXT_TW2048[idx][0], BD_EXT_TW2048[idx][1]);
    return make_float2(BD_EXT_TW1024[idx][0], BD_EXT_TW1024[idx][1]);
}

__device__ __forceinline__ int bd_rev_bits(int v, int bits)
{
    return (int)(__builtin_bitreverse32((unsigned)v) >> (32 - bits));
}

// general butterfly: t = w * b with w = (c, -s); a' = a + t, b' = a - t
__device__ __forceinline__ void bd_bfly(float &ar, float &ai, float &br, float &bi, float c, float s)
{
    const float wi = -s;
    const float p0 = br * c, p1 = br * wi;
    const float tr = __builtin_fmaf(-bi, wi, p0);
    const float ti = __builtin_fmaf(bi, c, p1);
    const float xr = ar, xi = ai;
    ar = xr + tr; ai = xi + ti;
    br = xr - tr; bi = xi - ti;
}

// NS radix-2 stages on CNT = 2^NS points held in registers; point i has global
// index base + i * 2^S0 and `lowj` = base mod 2^S0.  Global stage s = S0 + t + 1.
template <int E, int NS, int S0, int STRIDE>
__device__ __forceinline__ void bd_fft_stages(float *zr, float *zi, int lowj, const float2 *tw_lds)
{
    constexpr int CNT = 1 << NS;
    constexpr int LGN = bd_fft_cfg<E>::LGN;
#pragma unroll
    for (int t = 0; t < NS; t++) {
        constexpr int dummy = 0; (void)dummy;
        const int s = S0 + t + 1;
#pragma unroll
        for (int i = 0; i < CNT; i++) {
            if (i & (1 << t)) continue;
            const int ia = i * STRIDE, ib = (i + (1 << t)) * STRIDE;
            const int jl = i & ((1 << t) - 1);
            if (S0 == 0 && s == 1) {
                const float xr = zr[ia], xi = zi[ia], tr = zr[ib], ti = zi[ib];
                zr[ia] = xr + tr; zi[ia] = xi + ti; zr[ib] = xr - tr; zi[ib] = xi - ti;
            } else if (S0 == 0 && s == 2) {
                const float xr = zr[ia], xi = zi[ia];
                const float tr = jl == 0 ? zr[ib] : zi[ib];
                const float ti = jl == 0 ? zi[ib] : -zr[ib];
                zr[ia] = xr + tr; zi[ia] = xi + ti; zr[ib] = xr - tr; zi[ib] = xi - ti;
            } else if (S0 == 0) {
                const float2 w = bd_tw_global<E>(jl << (LGN - s)); // compile-time constant
                bd_bfly(zr[ia], zi[ia], zr[ib], zi[ib], w.x, w.y);
            } else {
                const int j = lowj + (jl << S0);
                const float2 w = tw_lds[bd_pad_t(j << (LGN - s))];
                bd_bfly(zr[ia], zi[ia], zr[ib], zi[ib], w.x, w.y);
            }
        }
    }
}

__device__ __forceinline__ float bd_xor_tree_sum(float p)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) p = p + __shfl_xor(p, o, 64);
    return p;
}

template <int E>
__global__ __launch_bounds__(256) void bd_stft_flux_kernel(bd_onset_args a)
{
    using C = bd_fft_cfg<E>;
    constexpr int M = C::M, LGE = C::LGE;
    __shared__ float2 tw_lds[C::TW_LEN];
    __shared__ float2 tile_all[4][C::DATA_LEN];
    __shared__ float out_lds[4][2][BD_ONSET_MAX_RUN];

    for (int i = threadIdx.x; i <= C::N / 2; i += 256) tw_lds[bd_pad_t(i)] = bd_tw_global<E>(i);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *tile = tile_all[wave];
    const uint64_t task = (uint64_t)blockIdx.x * 4 + wave;
    if (task >= a.n_tasks) return;
    const uint64_t s = task / a.runs_per_stream;
    const uint64_t f0 = (task - s * a.runs_per_stream) * a.run_frames;
    const uint64_t f_end = f0 + a.run_frames < a.n_frames ? f0 + a.run_frames : a.n_frames;
    const int16_t *sp = a.samples + s * a.stride;
    const int r6 = bd_rev_bits(lane, 6);

    // slot q of this lane holds complex point n_q = rev(q) * 64 + rev6(lane) of the frame
    int nq[E];
    float w0[E], w1[E]; // window at samples 2 n_q and 2 n_q + 1
#pragma unroll
    for (int q = 0; q < E; q++) {
        nq[q] = bd_rev_bits(q, LGE) * 64 + r6;
        if (a.window) {
            const int i0 = 2 * nq[q], i1 = i0 + 1;
            w0[q] = __builtin_fmaf(-0.5f, tw_lds[bd_pad_t(i0 <= M ? i0 : C::N - i0)].x, 0.5f);
            w1[q] = __builtin_fmaf(-0.5f, tw_lds[bd_pad_t(i1 <= M ? i1 : C::N - i1)].x, 0.5f);
        } else {
            w0[q] = 1.f; w1[q] = 1.f;
        }
    }

    float prev[E], prev_m = 0.f; // magnitudes of the previous frame: bins lane + 64 t, and bin M
#pragma unroll
    for (int t = 0; t < E; t++) prev[t] = 0.f;

    // raw points (two i16 per dword) of the current frame
    unsigned raw[E];
    const uint64_t f_first = f0 > 0 ? f0 - 1 : 0; // one frame of warm-up gives the flux its predecessor
    const unsigned *fp = reinterpret_cast<const unsigned *>(sp + f_first * (uint64_t)M);
#pragma unroll
    for (int q = 0; q < E; q++) raw[q] = __builtin_nontemporal_load(fp + nq[q]);

    for (uint64_t f = f_first; f < f_end; f++) {
        float zr[E], zi[E];
#pragma unroll
        for (int q = 0; q < E; q++) {
            const int d = (int)raw[q];
            const float x0 = (float)(short)(d & 0xffff), x1 = (float)(short)(d >> 16);
            zr[q] = a.window ? w0[q] * x0 : x0;
            zi[q] = a.window ? w1[q] * x1 : x1;
        }
        // next frame: its first half is this frame's second half (odd slots -> even slots)
        if (f + 1 < f_end) {
            const unsigned *np = reinterpret_cast<const unsigned *>(sp + (f + 1) * (uint64_t)M);
#pragma unroll
            for (int q = 0; q < E; q += 2) {
                raw[q] = raw[q + 1];
                raw[q + 1] = __builtin_nontemporal_load(np + nq[q + 1]);
            }
        }
        // phase 1: stages 1 .. NS1 on points E*lane .. E*lane + E - 1 (bit-reversed order)
        bd_fft_stages<E, C::NS1, 0, 1>(zr, zi, 0, tw_lds);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < E; q++) tile[bd_pad_d<E>(E * lane + q)] = make_float2(zr[q], zi[q]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // phase 2: stages NS1+1 .. NS1+NS2 on points base + i * E
        {
            constexpr int S0 = C::NS1, CNT = 1 << C::NS2;
            const int low = lane & ((1 << S0) - 1);
            const int base = ((lane >> S0) << (S0 + C::NS2)) + low;
            float yr[CNT], yi[CNT];
#pragma unroll
            for (int i = 0; i < CNT; i++) { const float2 v = tile[bd_pad_d<E>(base + (i << S0))]; yr[i] = v.x; yi[i] = v.y; }
            bd_fft_stages<E, C::NS2, S0, 1>(yr, yi, low, tw_lds);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < CNT; i++) tile[bd_pad_d<E>(base + (i << S0))] = make_float2(yr[i], yi[i]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // phase 3: the last NS3 stages; afterwards slot t holds Z[lane + 64 t]
        {
            constexpr int S0 = C::NS1 + C::NS2, CNT = 1 << C::NS3, GROUPS = E / CNT;
#pragma unroll
            for (int t = 0; t < E; t++) { const float2 v = tile[bd_pad_d<E>(lane + 64 * t)]; zr[t] = v.x; zi[t] = v.y; }
            // group g: points (lane + 64 g) + i * 2^S0  ->  slots g + GROUPS * i
#pragma unroll
            for (int g = 0; g < GROUPS; g++)
                bd_fft_stages<E, C::NS3, S0, GROUPS>(zr + g, zi + g, lane + 64 * g, tw_lds);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < E; t++) tile[bd_pad_d<E>(lane + 64 * t)] = make_float2(zr[t], zi[t]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // real-input post-processing and magnitudes
        float flux_p = 0.f, band_p = 0.f;
        float mag[E];
#pragma unroll
        for (int t = 0; t < E; t++) {
            const int k = lane + 64 * t;
            const float2 q = tile[bd_pad_d<E>((M - k) & (M - 1))];
            const float2 w = tw_lds[bd_pad_t(k)];
            const float er = 0.5f * (zr[t] + q.x), ei = 0.5f * (zi[t] - q.y);
            const float dr = 0.5f * (zr[t] - q.x), di = 0.5f * (zi[t] + q.y);
            const float cdr = w.x * dr, cdi = w.x * di;
            const float pr = __builtin_fmaf(w.y, di, cdr), qi = __builtin_fmaf(-w.y, dr, cdi);
            const float xr = er + qi, xi = ei - pr;
            const float xi2 = xi * xi;
            mag[t] = __builtin_sqrtf(__builtin_fmaf(xr, xr, xi2));
            const float d = mag[t] - prev[t];
            flux_p += d > 0.f ? d : 0.f;
            band_p += (uint32_t)k <= a.kmax ? mag[t] * mag[t] : 0.f;
            prev[t] = mag[t];
        }
        {   // bin M pairs Z[0] with itself, twiddle (cos pi, sin pi)
            const float2 z0 = tile[bd_pad_d<E>(0)];
            const float2 w = tw_lds[bd_pad_t(M)];
            const float er = 0.5f * (z0.x + z0.x), ei = 0.5f * (z0.y - z0.y);
            const float dr = 0.5f * (z0.x - z0.x), di = 0.5f * (z0.y + z0.y);
            const float cdr = w.x * dr, cdi = w.x * di;
            const float pr = __builtin_fmaf(w.y, di, cdr), qi = __builtin_fmaf(-w.y, dr, cdi);
            const float xr = er + qi, xi = ei - pr;
            const float xi2 = xi * xi;
            const float mm = __builtin_sqrtf(__builtin_fmaf(xr, xr, xi2));
            if (lane == 0) {
                const float d = mm - prev_m;
                flux_p += d > 0.f ? d : 0.f;
                band_p += (uint32_t)M <= a.kmax ? mm * mm : 0.f;
            }
            prev_m = mm;
        }
        const float flux = bd_xor_tree_sum(flux_p);
        const float band = bd_xor_tree_sum(band_p);
        if (f >= f0 && lane == 0) {
            out_lds[wave][0][f - f0] = flux;
            out_lds[wave][1][f - f0] = band;
        }
    }
    // results of the run, contiguous
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned n_res = (unsigned)(f_end - f0);
    const uint64_t o0 = s * a.out_stride + f0;
    if (a.flux) {
        __amdgpu_buffer_rsrc_t r = bd_out_rsrc(a.flux + o0, n_res * 4u);
        for (unsigned i = lane; i < n_res; i += 64) bd_store_stream_f32(r, i * 4u, out_lds[wave][0][i]);
    }
    if (a.band) {
        __amdgpu_buffer_rsrc_t r = bd_out_rsrc(a.band + o0, n_res * 4u);
        for (unsigned i = lane; i < n_res; i += 64) bd_store_stream_f32(r, i * 4u, out_lds[wave][1][i]);
    }
}

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
    uint32_t     fft_len, hop, sampling_rate;
    float        delta, lambda;
};

__global__ __launch_bounds__(256) void bd_peak_pick_kernel(bd_pick_args a)
{
    const int lane = threadIdx.x & 63;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
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
            const float mean = sum / (float)(hi - lo + 1);
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
}

extern "C" uint64_t bd_onset_frames(uint64_t n_samples, uint32_t fft_len, uint32_t hop)
{
    if (hop == 0 || n_samples < fft_len) return 0;
    return (n_samples - fft_len) / hop + 1;
}

static int bd_onset_check(const bd_onset_params *p, uint32_t sampling_rate)
{
    if (!p) { bd_set_error("onset parameters must not be NULL"); return BD_ERR_INVALID_ARG; }
    if ((p->fft_len != 1024 && p->fft_len != 2048) || p->hop != p->fft_len / 2) {
        bd_set_error("fft_len must be 1024 or 2048 and hop = fft_len / 2");
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
    if (((uintptr_t)samples % 16) || (stream_stride % 8)) { bd_set_error("bd_onset_stft: rows must be 16-byte aligned"); return BD_ERR_UNSUPPORTED; }
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
    if (grid64 > 0x7fffffffull) { bd_set_error("batch too large for one launch"); return BD_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)hip_stream;
    if (p->fft_len == 2048) hipLaunchKernelGGL(bd_stft_flux_kernel<16>, dim3((unsigned)grid64), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(bd_stft_flux_kernel<8>, dim3((unsigned)grid64), dim3(256), 0, s, a);
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
    bd_pick_args a;
    a.flux = flux; a.cand = candidates; a.count = onset_count; a.frame = onset_frame; a.ms = onset_ms;
    a.out_stride = flux_stride; a.n_frames = n_frames; a.n_streams = n_streams;
    a.mean_frames = p->mean_frames; a.max_frames = p->max_frames; a.min_gap_ms = p->min_gap_ms;
    a.max_onsets = max_onsets; a.fft_len = p->fft_len; a.hop = p->hop; a.sampling_rate = sampling_rate;
    a.delta = p->delta; a.lambda = p->lambda;
    const uint64_t grid64 = (n_streams + 3) / 4;
    if (grid64 > 0x7fffffffull) { bd_set_error("too many streams for one launch"); return BD_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(bd_peak_pick_kernel, dim3((unsigned)grid64), dim3(256), 0, (hipStream_t)hip_stream, a);
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
