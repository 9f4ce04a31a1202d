/*
 * Definition of the extension onset analysis on the CPU.  TEST INFRASTRUCTURE
 * ONLY; NO REFERENCE COUNTERPART (see bd_oracle_ext.h).
 *
 * Arithmetic specification (all f32; compile with -ffp-contract=off so that the
 * only fused operations are the fmaf calls written below):
 *
 * 1. Frame of N real samples x[i] (i16 converted exactly to f32), periodic Hann
 *    window w[i] = fma(-0.5, c, 0.5) with c = BD_EXT_TW<N>[i <= N/2 ? i : N - i][0]
 *    (= cos(2 pi i / N)), or w = 1: u[i] = w[i] * x[i].
 * 2. Complex sequence z[n] = (u[2n], u[2n+1]), n < M = N/2.
 * 3. Radix-2 decimation-in-time FFT of z: bit-reversal permutation, then stages
 *    s = 1 .. log2 M.  Butterfly on (a, b) with twiddle (wr, wi) =
 *    (cos, -sin)(2 pi j / 2^s) taken from BD_EXT_TW<N>[j * N / 2^s].
 *    In the first log2 M - 6 stages (4 for N = 2048, 3 for N = 1024) the twiddles (1, 0)
 *    (j = 0) and (0, -1) (j = 2^s / 4) are exact:
 *        t = b  resp.  t = (b.im, -b.re);      a' = a + t,  b' = a - t
 *    Every other butterfly of these stages, and every butterfly of the later stages
 *    whatever its twiddle, is the fused form (three packed instructions on the GPU):
 *        u  = (fma(b.re, wr, a.re),   fma(b.re, wi, a.im))
 *        a' = (fma(-b.im, wi, u.re),  fma(b.im, wr, u.im))           (= a + w b)
 *        b' = (fma(2, a.re, -a'.re),  fma(2, a.im, -a'.im))          (= a - w b)
 * 4. Real-input post-processing, k = 0 .. M, with P = Z[k mod M], Q = Z[(M - k) mod M],
 *    (c, s) = BD_EXT_TW<N>[k].  The halving of the textbook formula is deferred: the
 *    spectrum is carried at twice its value (Y = 2 X) and the factor is applied once to
 *    each sum.
 *        er = P.re + Q.re              ei = P.im - Q.im
 *        dr = P.re - Q.re              di = P.im + Q.im
 *        Y.re = fma(-s, dr, fma(c, di, er))
 *        Y.im = fma(-s, di, fma(-c, dr, ei))
 *        m2[k] = sqrt(fma(Y.re, Y.re, Y.im * Y.im))             (= 2 |X[k]|)
 * 5. Sums over bins (flux, band energy) are taken as 64 interleaved partial sums
 *    (partial l adds the terms of k = l, l + 64, ... in ascending order, the term of
 *    k = M last into partial 0) combined by the butterfly p[l] += p[l ^ o],
 *    o = 1, 2, 4, 8, 16, 32.
 *        flux[j] = 0.5 * sum_k max(0, m2_j[k] - m2_{j-1}[k])     (m2_{-1} = 0)
 *        band_energy[j] = 0.25 * sum_{k <= kmax} m2_j[k]^2        (terms fma-free: m * m)
 *    and |X[k]| = 0.5 * m2[k] where a magnitude is reported.
 * 6. With flux_power set no square root is taken: p2[k] = fma(Y.re, Y.re, Y.im * Y.im) (= 4 |X[k]|^2),
 *        flux[j] = 0.25 * sum_k max(0, p2_j[k] - p2_{j-1}[k])    (p2_{-1} = 0)
 *        band_energy[j] = 0.25 * sum_{k <= kmax} p2_j[k]
 */
#include "bd_oracle_ext.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bd_ext_tables.inc"

void bdo_onset_default_params(bdo_onset_params *p)
{
    p->fft_len = 2048;
    p->hop = 1024;
    p->window = 1;
    p->band_max_hz = 90.0f;
    p->mean_frames = 8;
    p->max_frames = 3;
    p->delta = 0.0f;
    p->lambda = 1.5f;
    p->min_gap_ms = 100;
    p->use_median = 0;
    p->flux_power = 0;
}

size_t bdo_onset_frames(size_t n_samples, uint32_t fft_len, uint32_t hop)
{
    if (hop == 0 || n_samples < fft_len) return 0;
    return (n_samples - fft_len) / hop + 1;
}

uint32_t bdo_onset_frame_ms(size_t j, uint32_t fft_len, uint32_t hop, uint32_t sampling_rate)
{
    return (uint32_t)((((uint64_t)j * hop + fft_len / 2) * 1000u) / sampling_rate);
}

static const float (*tw_table(uint32_t n))[2]
{
    return n == 1024 ? BD_EXT_TW1024 : BD_EXT_TW2048;
}

static int log2u(uint32_t v) { int r = 0; while ((1u << r) < v) r++; return r; }

/* m2[k] = 2 |X[k]| of one frame (items 1 - 4), or its square 4 |X[k]|^2 (item 6) */
static int rfft_mag2(const int16_t *frame, uint32_t n, uint32_t window, float *mag, int power)
{
    if (n != 1024 && n != 2048) return -1;
    const float (*tw)[2] = tw_table(n);
    const uint32_t m = n / 2;
    const int lg = log2u(m);
    float *re = (float *)malloc(sizeof(float) * m), *im = (float *)malloc(sizeof(float) * m);
    for (uint32_t i = 0; i < m; i++) {
        uint32_t r = 0;
        for (int b = 0; b < lg; b++) if (i & (1u << b)) r |= 1u << (lg - 1 - b);
        float u0 = (float)frame[2 * i], u1 = (float)frame[2 * i + 1];
        if (window) {
            const uint32_t i0 = 2 * i, i1 = 2 * i + 1;
            const float w0 = fmaf(-0.5f, tw[i0 <= m ? i0 : n - i0][0], 0.5f);
            const float w1 = fmaf(-0.5f, tw[i1 <= m ? i1 : n - i1][0], 0.5f);
            u0 = w0 * u0; u1 = w1 * u1;
        }
        re[r] = u0; im[r] = u1;
    }
    /* The first lg - 6 stages (4 for N = 2048, 3 for N = 1024) take the exact forms X = a + t,
     * Y = a - t for the twiddles (1, 0) and (0, -1); every other butterfly, and every butterfly
     * of the later stages, is the fused form
     *     u = (fma(b.re, wr, a.re), fma(b.re, wi, a.im))
     *     X = (fma(-b.im, wi, u.re), fma(b.im, wr, u.im))
     *     Y = (fma(2, a.re, -X.re), fma(2, a.im, -X.im))                                  */
    const int exact_stages = lg - 6;
    for (int s = 1; s <= lg; s++) {
        const uint32_t len = 1u << s, half = len >> 1;
        for (uint32_t base = 0; base < m; base += len) {
            for (uint32_t j = 0; j < half; j++) {
                const uint32_t a = base + j, b = a + half;
                const float ar = re[a], ai = im[a];
                if (s <= exact_stages && (j == 0 || 4 * j == len)) {
                    float tr, ti;
                    if (j == 0) { tr = re[b]; ti = im[b]; } else { tr = im[b]; ti = -re[b]; }
                    re[a] = ar + tr; im[a] = ai + ti;
                    re[b] = ar - tr; im[b] = ai - ti;
                } else {
                    const float wr = tw[j * (n >> s)][0], wi = -tw[j * (n >> s)][1];
                    const float ur = fmaf(re[b], wr, ar), ui = fmaf(re[b], wi, ai);
                    const float xr = fmaf(-im[b], wi, ur), xi = fmaf(im[b], wr, ui);
                    re[a] = xr; im[a] = xi;
                    re[b] = fmaf(2.0f, ar, -xr); im[b] = fmaf(2.0f, ai, -xi);
                }
            }
        }
    }
    for (uint32_t k = 0; k <= m; k++) {
        const uint32_t kp = k % m, kq = (m - k) % m;
        const float c = tw[k][0], s = tw[k][1];
        const float er = re[kp] + re[kq], ei = im[kp] - im[kq];
        const float dr = re[kp] - re[kq], di = im[kp] + im[kq];
        const float xr = fmaf(-s, dr, fmaf(c, di, er));
        const float xi = fmaf(-s, di, fmaf(-c, dr, ei));
        const float xi2 = xi * xi;
        mag[k] = power ? fmaf(xr, xr, xi2) : sqrtf(fmaf(xr, xr, xi2));
    }
    free(re); free(im);
    return 0;
}

int bdo_ext_rfft_mag(const int16_t *frame, uint32_t n, uint32_t window, float *mag)
{
    if (rfft_mag2(frame, n, window, mag, 0) != 0) return -1;
    for (uint32_t k = 0; k <= n / 2; k++) mag[k] = 0.5f * mag[k];
    return 0;
}

int bdo_ext_rfft_mag_f64(const int16_t *frame, uint32_t n, uint32_t window, double *mag)
{
    if (n != 1024 && n != 2048) return -1;
    const double pi = 3.14159265358979323846;
    for (uint32_t k = 0; k <= n / 2; k++) {
        double r = 0.0, i = 0.0;
        for (uint32_t t = 0; t < n; t++) {
            const double w = window ? 0.5 - 0.5 * cos(2.0 * pi * t / n) : 1.0;
            const double a = -2.0 * pi * (double)((uint64_t)k * t % n) / n;
            r += w * frame[t] * cos(a);
            i += w * frame[t] * sin(a);
        }
        mag[k] = sqrt(r * r + i * i);
    }
    return 0;
}

/* 64 interleaved partial sums + xor butterfly (item 5 of the specification) */
static float lane_sum(const float *term, uint32_t m /* terms for k = 0 .. m */)
{
    float p[64];
    for (int l = 0; l < 64; l++) {
        float s = 0.0f;
        for (uint32_t k = (uint32_t)l; k < m; k += 64) s += term[k];
        p[l] = s;
    }
    p[0] += term[m];
    for (int o = 1; o <= 32; o <<= 1) {
        float q[64];
        for (int l = 0; l < 64; l++) q[l] = p[l] + p[l ^ o];
        memcpy(p, q, sizeof(p));
    }
    return p[0];
}

int bdo_onset_stft(const int16_t *samples, size_t n_samples, uint32_t sampling_rate,
                   const bdo_onset_params *p, float *flux, float *band_energy)
{
    const uint32_t n = p->fft_len, m = n / 2;
    if ((n != 1024 && n != 2048) || (p->hop != n / 2 && p->hop != n / 4)) return -1;
    const size_t frames = bdo_onset_frames(n_samples, n, p->hop);
    float *prev = (float *)calloc(m + 1, sizeof(float)), *cur = (float *)malloc((m + 1) * sizeof(float));
    float *term = (float *)malloc((m + 1) * sizeof(float));
    /* bins of the band: k * (fs / N) <= band_max_hz, f32 as in the Spectrum strategy */
    const float res = (float)sampling_rate / (float)n;
    uint32_t kmax = 0;
    for (uint32_t k = 0; k <= m; k++) { if ((float)k * res <= p->band_max_hz) kmax = k; else break; }
    for (size_t j = 0; j < frames; j++) {
        const int power = p->flux_power != 0;
        rfft_mag2(samples + j * p->hop, n, p->window, cur, power);
        if (flux) {
            for (uint32_t k = 0; k <= m; k++) { const float d = cur[k] - prev[k]; term[k] = d > 0.0f ? d : 0.0f; }
            flux[j] = (power ? 0.25f : 0.5f) * lane_sum(term, m);
        }
        if (band_energy) {
            for (uint32_t k = 0; k <= m; k++) term[k] = k <= kmax ? (power ? cur[k] : cur[k] * cur[k]) : 0.0f;
            band_energy[j] = 0.25f * lane_sum(term, m);
        }
        float *t = prev; prev = cur; cur = t;
    }
    free(prev); free(cur); free(term);
    return 0;
}

size_t bdo_onset_pick(const float *flux, size_t n_frames, uint32_t sampling_rate,
                      const bdo_onset_params *p, uint8_t *candidates,
                      uint32_t *onset_frame, uint32_t *onset_ms, size_t max_onsets)
{
    size_t count = 0;
    uint32_t last_ms = 0;
    int have_last = 0;
    for (size_t j = 0; j < n_frames; j++) {
        const size_t lo = j > p->mean_frames ? j - p->mean_frames : 0;
        const size_t hi = j + p->mean_frames < n_frames - 1 ? j + p->mean_frames : n_frames - 1;
        float sum = 0.0f;
        for (size_t i = lo; i <= hi; i++) sum += flux[i];
        float mean = sum / (float)(hi - lo + 1);
        if (p->use_median) {
            /* element c/2 of the sorted window; equal values keep their order (any choice gives the same value) */
            const size_t c = hi - lo + 1;
            for (size_t i = lo; i <= hi; i++) {
                size_t rank = 0;
                for (size_t k = lo; k <= hi; k++)
                    if (flux[k] < flux[i] || (flux[k] == flux[i] && k < i)) rank++;
                if (rank == c / 2) mean = flux[i];
            }
        }
        const float thr = fmaf(p->lambda, mean, p->delta);
        const size_t mlo = j > p->max_frames ? j - p->max_frames : 0;
        const size_t mhi = j + p->max_frames < n_frames - 1 ? j + p->max_frames : n_frames - 1;
        int is_max = 1;
        for (size_t i = mlo; i <= mhi; i++) {
            if (i < j && flux[i] >= flux[j]) is_max = 0;   /* ties: the earliest frame wins */
            if (i > j && flux[i] > flux[j]) is_max = 0;
        }
        const int cand = is_max && flux[j] > thr;
        if (candidates) candidates[j] = (uint8_t)cand;
        if (!cand) continue;
        const uint32_t ms = bdo_onset_frame_ms(j, p->fft_len, p->hop, sampling_rate);
        if (have_last && ms < last_ms + p->min_gap_ms) continue;
        have_last = 1; last_ms = ms;
        if (count < max_onsets) {
            if (onset_frame) onset_frame[count] = (uint32_t)j;
            if (onset_ms) onset_ms[count] = ms;
        }
        count++;
    }
    return count;
}

/* ======================================================================
 * Carried-state biquad envelope.  Arithmetic specification (all f32, fma
 * exactly where written):
 *
 * 1. Design: RBJ low-pass in float64 (w0 = 2 pi fc / fs, alpha = sin w0 / (2 Q),
 *    b0 = b2 = (1 - cos w0) / 2, b1 = 1 - cos w0, a0 = 1 + alpha, a1 = -2 cos w0,
 *    a2 = 1 - alpha), normalised by a0 and rounded to f32: b0, b1, b2, na1 = -a1/a0,
 *    na2 = -a2/a0; c1b = na1^2 + na2, c2b = na1 na2 from the rounded coefficients in float64,
 *    rounded once.  A = [[na1, na2], [1, 0]] (from the rounded coefficients, as
 *    float64), P = A^64; pw[k] = P^(2^k), q[l] = P^(l+1), computed in float64 by
 *    repeated multiplication and rounded to f32.
 * 2. u[n] = fma(b1, x[n-1], fma(b2, x[n-2], b0 * x[n])), x[-1] = x[-2] = 0.
 * 3. The stream is cut into chunks of 64 samples; 64 consecutive chunks form a
 *    step (4096 samples).  A trailing partial step is evaluated with the missing
 *    chunks' input taken as zero (their results are not reported).
 *    The recurrence advances two samples at a time from (p1, p2) = (y[2i-1], y[2i-2]):
 *        w_i     = fma(na1, u[2i], u[2i+1])
 *        y[2i]   = fma(na1, p1, fma(na2, p2, u[2i]))
 *        y[2i+1] = fma(c1b, p1, fma(c2b, p2, w_i))
 *    Zero-state response of chunk l of a step: z[-1] = z[-2] = 0, z advanced as above;
 *    r_l = (z[63], z[62]).
 * 4. Inclusive scan over the 64 chunks of a step (Hillis-Steele, k = 0..5):
 *    v_l <- M_k v_{l - 2^k} + v_l for l >= 2^k, with M_k = pw[k] and
 *    (M v + w).a = fma(M01, v.b, fma(M00, v.a, w.a)), .b = fma(M11, v.b, fma(M10, v.a, w.b)).
 * 5. State at the end of chunk l: e_l = q[l] s_in + v_l (same fma pattern), where
 *    s_in is the state at the start of the step (zero for the first step, e_63 of
 *    the previous step otherwise).  Chunk l starts from s_l = e_{l-1} (s_0 = s_in).
 * 6. Output of chunk l: y[-1] = s_l.a, y[-2] = s_l.b, y advanced as in item 3.
 * 7. Per complete window: m = max |x| with -32768 counted as 32767; value = max |y|;
 *    thr = (int)(ratio * (float)m) (truncation); flag = m >= min_amplitude && value >= (float)thr.
 * ====================================================================== */
void bdo_biquad_default_params(bdo_biquad_params *p)
{
    p->cutoff_hz = 120; p->q_factor = 0.70710678f; p->window_len = 1024; p->ratio = 0.77f; p->min_amplitude = 9830;
}

static void mat2_mul(const double a[4], const double b[4], double c[4])
{
    const double r0 = a[0] * b[0] + a[1] * b[2], r1 = a[0] * b[1] + a[1] * b[3];
    const double r2 = a[2] * b[0] + a[3] * b[2], r3 = a[2] * b[1] + a[3] * b[3];
    c[0] = r0; c[1] = r1; c[2] = r2; c[3] = r3;
}

int bdo_biquad_design_lowpass(uint32_t sampling_rate, uint32_t cutoff_hz, float q_factor, bdo_biquad_design *d)
{
    if (!d || sampling_rate == 0 || cutoff_hz == 0 || 2.0 * cutoff_hz >= (double)sampling_rate || !(q_factor > 0.0f)) return -1;
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
    const double a[4] = {(double)d->na1, (double)d->na2, 1.0, 0.0};
    double p[4] = {1.0, 0.0, 0.0, 1.0};
    for (int i = 0; i < BDO_BQ_CHUNK; i++) mat2_mul(a, p, p);           /* P = A^64 */
    double m[4] = {p[0], p[1], p[2], p[3]};
    for (int k = 0; k < 6; k++) {
        for (int j = 0; j < 4; j++) d->pw[k][j] = (float)m[j];
        mat2_mul(m, m, m);                                              /* P^(2^k) */
    }
    double ql[4] = {p[0], p[1], p[2], p[3]};
    for (int l = 0; l < 64; l++) {
        for (int j = 0; j < 4; j++) d->q[l][j] = (float)ql[j];
        mat2_mul(p, ql, ql);                                            /* P^(l+1) */
    }
    return 0;
}

static inline void affine2(const float m[4], float va, float vb, float wa, float wb, float *oa, float *ob)
{
    *oa = fmaf(m[1], vb, fmaf(m[0], va, wa));
    *ob = fmaf(m[3], vb, fmaf(m[2], va, wb));
}

int bdo_biquad_envelope(const int16_t *samples, size_t n_samples, const bdo_biquad_design *d,
                        const bdo_biquad_params *p, float *values, uint8_t *flags, float *y_out)
{
    const uint32_t w = p->window_len;
    if (w < 64 || w > 4096 || (w & (w - 1))) return -1;
    const size_t n_win = n_samples / w, n_used = n_win * w;
    const size_t step = 64 * BDO_BQ_CHUNK, n_steps = (n_used + step - 1) / step;
    float *u = (float *)malloc(step * sizeof(float)), *y = (float *)malloc(step * sizeof(float));
    float sa = 0.0f, sb = 0.0f;   /* s_in */
    float wmax_y = 0.0f; int wmax_x = 0;
    for (size_t st = 0; st < n_steps; st++) {
        const size_t base = st * step;
        /* item 2 (samples beyond the analysed range count as zero) */
        for (size_t i = 0; i < step; i++) {
            const size_t n = base + i;
            const float x0 = n < n_used ? (float)samples[n] : 0.0f;
            const float x1 = (n >= 1 && n - 1 < n_used) ? (float)samples[n - 1] : 0.0f;
            const float x2 = (n >= 2 && n - 2 < n_used) ? (float)samples[n - 2] : 0.0f;
            u[i] = fmaf(d->b1, x1, fmaf(d->b2, x2, d->b0 * x0));
        }
        float va[64], vb[64];
        for (int l = 0; l < 64; l++) {        /* item 3 */
            float z1 = 0.0f, z2 = 0.0f;
            for (int i = 0; i < BDO_BQ_CHUNK; i += 2) {
                const float u0 = u[l * BDO_BQ_CHUNK + i], u1 = u[l * BDO_BQ_CHUNK + i + 1];
                const float w = fmaf(d->na1, u0, u1);
                const float za = fmaf(d->na1, z1, fmaf(d->na2, z2, u0));
                const float zb = fmaf(d->c1b, z1, fmaf(d->c2b, z2, w));
                z2 = za; z1 = zb;
            }
            va[l] = z1; vb[l] = z2;
        }
        for (int k = 0; k < 6; k++) {          /* item 4 */
            float na[64], nb[64];
            for (int l = 0; l < 64; l++) {
                if (l >= (1 << k)) affine2(d->pw[k], va[l - (1 << k)], vb[l - (1 << k)], va[l], vb[l], &na[l], &nb[l]);
                else { na[l] = va[l]; nb[l] = vb[l]; }
            }
            memcpy(va, na, sizeof(va)); memcpy(vb, nb, sizeof(vb));
        }
        float ea[64], eb[64];
        for (int l = 0; l < 64; l++) affine2(d->q[l], sa, sb, va[l], vb[l], &ea[l], &eb[l]);   /* item 5 */
        for (int l = 0; l < 64; l++) {        /* item 6 */
            float y1 = l ? ea[l - 1] : sa, y2 = l ? eb[l - 1] : sb;
            for (int i = 0; i < BDO_BQ_CHUNK; i += 2) {
                const float u0 = u[l * BDO_BQ_CHUNK + i], u1 = u[l * BDO_BQ_CHUNK + i + 1];
                const float w = fmaf(d->na1, u0, u1);
                const float ya = fmaf(d->na1, y1, fmaf(d->na2, y2, u0));
                const float yb = fmaf(d->c1b, y1, fmaf(d->c2b, y2, w));
                y[l * BDO_BQ_CHUNK + i] = ya; y[l * BDO_BQ_CHUNK + i + 1] = yb;
                y2 = ya; y1 = yb;
            }
        }
        sa = ea[63]; sb = eb[63];
        for (size_t i = 0; i < step && base + i < n_used; i++) {   /* item 7 */
            const size_t n = base + i;
            if (y_out) y_out[n] = y[i];
            int ax = samples[n] == INT16_MIN ? 32767 : (samples[n] < 0 ? -samples[n] : samples[n]);
            if (ax > wmax_x) wmax_x = ax;
            const float ay = fabsf(y[i]);
            if (ay > wmax_y) wmax_y = ay;
            if ((n + 1) % w == 0) {
                const size_t wi = n / w;
                const int thr = (int)(p->ratio * (float)wmax_x);
                if (values) values[wi] = wmax_y;
                if (flags) flags[wi] = (uint8_t)((uint32_t)wmax_x >= p->min_amplitude && wmax_y >= (float)thr);
                wmax_x = 0; wmax_y = 0.0f;
            }
        }
    }
    free(u); free(y);
    return 0;
}

void bdo_biquad_filter_f64(const int16_t *samples, size_t n_samples, uint32_t sampling_rate, uint32_t cutoff_hz,
                           double q_factor, double *y)
{
    const double pi = 3.14159265358979323846;
    const double w0 = 2.0 * pi * (double)cutoff_hz / (double)sampling_rate;
    const double cw = cos(w0), alpha = sin(w0) / (2.0 * q_factor), a0 = 1.0 + alpha;
    const double b0 = ((1.0 - cw) / 2.0) / a0, b1 = (1.0 - cw) / a0, a1 = (-2.0 * cw) / a0, a2 = (1.0 - alpha) / a0;
    double x1 = 0, x2 = 0, y1 = 0, y2 = 0;
    for (size_t n = 0; n < n_samples; n++) {
        const double x0 = samples[n];
        const double v = b0 * x0 + b1 * x1 + b0 * x2 - a1 * y1 - a2 * y2;
        y[n] = v; x2 = x1; x1 = x0; y2 = y1; y1 = v;
    }
}
