// Deterministic synthetic click tracks, integer arithmetic only, so that the
// host and the GPU produce the same bits on any machine.
//
//   sample(s, n) = clamp_i16( noise(s, n) + click(s, n) )
//   noise  : a 32-bit counter hash of (stream key, n) mapped to [-500, 500]
//   click  : a decaying ~60 Hz burst (4096-sample template, peak 32767) placed
//            every `period` samples from `offset`; tempo 60..180 BPM per
//            stream, base level 40..95 % of full scale per stream, each click
//            scaled by a further 75..100 %.
#ifndef BD_SYNTH_H
#define BD_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define BD_HD __host__ __device__ __forceinline__
#else
#define BD_HD static inline
#endif

#define BD_CLICK_LEN 4096

BD_HD uint64_t bd_mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

BD_HD uint32_t bd_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

typedef struct bd_synth_stream {
    uint32_t key;       // noise key
    uint32_t click_key; // per-click level key
    uint32_t period;    // samples between clicks
    uint32_t offset;    // position of the first click
    uint32_t level_q15; // base click level, 32768 = full scale
} bd_synth_stream;

BD_HD bd_synth_stream bd_synth_stream_params(uint64_t seed, uint64_t stream, uint32_t sampling_rate)
{
    bd_synth_stream p;
    uint64_t h = bd_mix64(seed ^ bd_mix64(stream));
    p.key = (uint32_t)h;
    p.click_key = (uint32_t)(h >> 32);
    uint64_t h2 = bd_mix64(h);
    uint32_t bpm = 60u + (uint32_t)(h2 % 121u);
    p.period = (uint32_t)(((uint64_t)sampling_rate * 60u) / bpm);
    if (p.period == 0) p.period = 1;
    p.offset = (uint32_t)((h2 >> 20) % p.period);
    p.level_q15 = 13107u + (uint32_t)((h2 >> 44) % 18023u);
    return p;
}

// `click` points at the BD_CLICK_LEN-entry template.
BD_HD int16_t bd_synth_sample(const bd_synth_stream &p, const int16_t *click, uint64_t n)
{
    uint32_t h = bd_mix32(p.key + (uint32_t)n * 0x9E3779B1u + (uint32_t)(n >> 32));
    int32_t v = (int32_t)(((h >> 16) * 1001u) >> 16) - 500;
    if (n >= p.offset) {
        uint64_t rel = n - p.offset;
        uint64_t ci = rel / p.period;
        uint32_t phase = (uint32_t)(rel - ci * p.period);
        if (phase < BD_CLICK_LEN) {
            uint32_t jitter = 768u + bd_mix32(p.click_key ^ (uint32_t)ci) % 257u; // /1024
            int64_t level = ((int64_t)p.level_q15 * jitter) >> 10;               // q15
            v += (int32_t)(((int64_t)click[phase] * level) >> 15);
        }
    }
    if (v > 32767) v = 32767;
    if (v < -32768) v = -32768;
    return (int16_t)v;
}

// Fills the click template: two-pole resonator in fixed point
// (60 Hz, 30 ms decay at 44.1 kHz), normalised to a peak of 32767, linear
// fade over the last quarter.  Host only.
static inline void bd_synth_make_click(int16_t *click)
{
    const int64_t c1 = 2145782662ll; // round(2 r cos w * 2^30)
    const int64_t c2 = 1072119858ll; // round(r^2 * 2^30)
    static int64_t raw[BD_CLICK_LEN];
    int64_t y1 = 0, y2 = 0, peak = 1;
    for (int i = 0; i < BD_CLICK_LEN; i++) {
        int64_t y = (i == 0) ? (1ll << 20) : ((c1 * y1 - c2 * y2) >> 30);
        raw[i] = (i == 0) ? 0 : y1; // delay by one so the burst starts at zero
        y2 = y1; y1 = y;
        int64_t a = raw[i] < 0 ? -raw[i] : raw[i];
        if (a > peak) peak = a;
    }
    for (int i = 0; i < BD_CLICK_LEN; i++) {
        int64_t v = raw[i] * 32767 / peak;
        if (i >= 3 * BD_CLICK_LEN / 4) v = v * (BD_CLICK_LEN - i) / (BD_CLICK_LEN / 4);
        click[i] = (int16_t)v;
    }
}

#endif
