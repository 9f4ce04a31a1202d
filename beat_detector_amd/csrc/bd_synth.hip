// Synthetic click-track generator (device and host twins of one integer-only
// definition, see bd_synth.h) and the live-path ingest converters.
#include "bd_internal.h"
#include "bd_synth.h"

#include <mutex>
#include <vector>

static __device__ int16_t bd_click_dev[BD_CLICK_LEN];

static const int16_t *bd_click_host()
{
    static int16_t click[BD_CLICK_LEN];
    static std::once_flag once;
    std::call_once(once, [] { bd_synth_make_click(click); });
    return click;
}

#define BD_SYNTH_ROUNDS 16 /* groups of 8 samples per thread: the per-stream set-up is paid once for all of them */

__global__ __launch_bounds__(256) void bd_synth_kernel(int16_t *samples, uint64_t n_streams,
                                                       uint64_t n_samples, uint64_t stride,
                                                       uint64_t first_stream, uint64_t seed,
                                                       uint32_t sampling_rate, uint64_t groups_per_stream,
                                                       uint64_t threads_per_stream)
{
    // A thread writes the groups j, j + T, j + 2 T, ... of 8 consecutive samples of one stream (T =
    // threads per stream): consecutive threads write consecutive 16-byte units in every round, and the
    // stream's parameters (three 64-bit hashes, three divisions) are derived once per thread.
    const uint64_t t = bd_block_index() * blockDim.x + threadIdx.x;
    if (t >= n_streams * threads_per_stream) return;
    const uint64_t s = t / threads_per_stream;
    const uint64_t j = t - s * threads_per_stream;
    const bd_synth_stream p = bd_synth_stream_params(seed, first_stream + s, sampling_rate);
    for (int r = 0; r < BD_SYNTH_ROUNDS; r++) {
        const uint64_t g = j + (uint64_t)r * threads_per_stream;
        if (g >= groups_per_stream) break;
        const uint64_t n0 = g * 8;
        // bd_synth_sample() for n0 .. n0 + 7 with the position inside the click pattern carried from
        // sample to sample (one division per group instead of one per sample); same bits
        int16_t v[8];
        uint64_t ci = 0;
        uint32_t phase = 0, level = 0;
        bool level_known = false;
        // samples before `skip` precede the first click (8: the whole group does)
        const int skip = n0 >= p.offset ? 0 : (p.offset - n0 < 8 ? (int)(p.offset - n0) : 8);
        if (skip < 8) {
            const uint64_t rel = n0 + (uint64_t)skip - p.offset;
            ci = rel >> 32 ? rel / p.period : (uint64_t)((uint32_t)rel / p.period);
            phase = (uint32_t)(rel - ci * p.period);
        }
        // the noise hash takes n in 32-bit halves; the high half is constant inside a group of 8
        uint32_t hk = p.key + (uint32_t)n0 * 0x9E3779B1u + (uint32_t)(n0 >> 32);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t h = bd_mix32(hk);
            hk += 0x9E3779B1u;
            int32_t x = (int32_t)(((h >> 16) * 1001u) >> 16) - 500;
            if (i >= skip) {
                if (phase < BD_CLICK_LEN) {
                    if (!level_known) {
                        const uint32_t jitter = 768u + bd_mix32(p.click_key ^ (uint32_t)ci) % 257u;
                        level = (uint32_t)(((int64_t)p.level_q15 * jitter) >> 10);
                        level_known = true;
                    }
                    x += (int32_t)(((int64_t)bd_click_dev[phase] * (int64_t)level) >> 15);
                }
                if (++phase == p.period) { phase = 0; ci++; level_known = false; }
            }
            x = x > 32767 ? 32767 : x;
            x = x < -32768 ? -32768 : x;
            v[i] = (int16_t)x;
        }
        int16_t *dst = samples + s * stride + n0;
        if (n0 + 8 <= n_samples && ((uintptr_t)dst % 16 == 0)) {
            uint4 pk;
            pk.x = (uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16);
            pk.y = (uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
            pk.z = (uint16_t)v[4] | ((uint32_t)(uint16_t)v[5] << 16);
            pk.w = (uint16_t)v[6] | ((uint32_t)(uint16_t)v[7] << 16);
            *reinterpret_cast<uint4 *>(dst) = pk;
        } else {
            for (int i = 0; i < 8; i++)
                if (n0 + i < n_samples) dst[i] = v[i];
        }
    }
}

extern "C" int bd_synth_fill(int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                             uint64_t stream_stride, uint64_t first_stream, uint64_t seed,
                             uint32_t sampling_rate, void *hip_stream)
{
    if (!samples || stream_stride < n_samples || sampling_rate == 0) {
        bd_set_error("bd_synth_fill: bad arguments"); return BD_ERR_INVALID_ARG;
    }
    if (n_streams == 0 || n_samples == 0) return BD_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    // the template is uploaded once per device and process
    static std::mutex mu;
    static std::vector<int> done;
    int dev = 0;
    BD_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(mu);
        bool have = false;
        for (int d : done) have = have || d == dev;
        if (!have) {
            BD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(bd_click_dev), bd_click_host(),
                                     sizeof(int16_t) * BD_CLICK_LEN, 0, hipMemcpyHostToDevice));
            done.push_back(dev);
        }
    }
    const uint64_t groups = (n_samples + 7) / 8;
    const uint64_t threads_per_stream = (groups + BD_SYNTH_ROUNDS - 1) / BD_SYNTH_ROUNDS;
    const uint64_t grid64 = (n_streams * threads_per_stream + 255) / 256;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("bd_synth_fill: batch too large for one launch"); return BD_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(bd_synth_kernel, grid_dims, dim3(256), 0, st,
                       samples, n_streams, n_samples, stream_stride, first_stream, seed,
                       sampling_rate, groups, threads_per_stream);
    return bd_check_hip(hipGetLastError(), "launch bd_synth_kernel");
}

extern "C" int bd_synth_fill_host(int16_t *samples, uint64_t n_streams, uint64_t n_samples,
                                  uint64_t stream_stride, uint64_t first_stream, uint64_t seed,
                                  uint32_t sampling_rate)
{
    if (!samples || stream_stride < n_samples || sampling_rate == 0) {
        bd_set_error("bd_synth_fill_host: bad arguments"); return BD_ERR_INVALID_ARG;
    }
    const int16_t *click = bd_click_host();
    for (uint64_t s = 0; s < n_streams; s++) {
        const bd_synth_stream p = bd_synth_stream_params(seed, first_stream + s, sampling_rate);
        int16_t *dst = samples + s * stream_stride;
        for (uint64_t n = 0; n < n_samples; n++) dst[n] = bd_synth_sample(p, click, n);
    }
    return BD_OK;
}

// ---- ingest converters, src/record.rs:151-166 -------------------------------
// Streaming element-wise kernels: with 16-byte aligned buffers every thread converts the eight
// samples of one 16-byte output unit (16-byte non-temporal loads, one 16-byte store) and one more
// thread the up to seven samples after the last whole unit; other buffers take one sample per thread.
__device__ __forceinline__ int16_t bd_cvt_u16(uint16_t v)
{
    const int32_t x = (int32_t)v - (int32_t)(INT16_MAX / 2);    // record.rs:154-155
    return (int16_t)(uint16_t)(uint32_t)x;                       // :156 wrapping cast
}
__device__ __forceinline__ int16_t bd_cvt_f32(float v)
{
    const float x = v * 32767.0f;                                // record.rs:163
    // `as i16`: truncate toward zero, saturate, NaN -> 0        // :164
    float t = truncf(x);
    t = (t != t) ? 0.f : fminf(fmaxf(t, -32768.f), 32767.f);
    return (int16_t)(int)t;
}
// Down-mix of one interleaved stereo frame as the reference's test harness does it
// (src/lib.rs:310-337: `((l as i32 + r as i32) as f32 / 2.0) as i16`, i.e. the sum halved
// with truncation toward zero; the sum of two i16 is exact in f32).
__device__ __forceinline__ int16_t bd_cvt_stereo(int16_t l, int16_t r)
{
    return (int16_t)(((int32_t)l + (int32_t)r) / 2); // C division truncates toward zero, like the f32 -> i16 cast
}
__device__ __forceinline__ unsigned bd_pack16(int16_t lo, int16_t hi)
{
    return (unsigned)(uint16_t)lo | ((unsigned)(uint16_t)hi << 16);
}
__device__ __forceinline__ void bd_store_unit(int16_t *out, uint64_t unit, const unsigned (&w)[4])
{
    bd_u32x4 v = {w[0], w[1], w[2], w[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<bd_u32x4 *>(out) + unit);
}

// VEC: 16-byte aligned buffers
template <bool VEC>
__global__ __launch_bounds__(256) void bd_u16_to_i16_kernel(const uint16_t *in, int16_t *out, uint64_t n)
{
    const uint64_t i = bd_block_index() * blockDim.x + threadIdx.x;
    if (!VEC) { if (i < n) out[i] = bd_cvt_u16(in[i]); return; }
    const uint64_t units = n / 8;
    if (i < units) {
        const uint4 v = bd_load_stream16(reinterpret_cast<const uint4 *>(in) + i);
        const unsigned s[4] = {v.x, v.y, v.z, v.w};
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = bd_pack16(bd_cvt_u16((uint16_t)(s[q] & 0xffff)), bd_cvt_u16((uint16_t)(s[q] >> 16)));
        bd_store_unit(out, i, w);
    } else if (i == units) {
        for (uint64_t e = units * 8; e < n; e++) out[e] = bd_cvt_u16(in[e]);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bd_f32_to_i16_kernel(const float *in, int16_t *out, uint64_t n)
{
    const uint64_t i = bd_block_index() * blockDim.x + threadIdx.x;
    if (!VEC) { if (i < n) out[i] = bd_cvt_f32(in[i]); return; }
    const uint64_t units = n / 8;
    if (i < units) {
        const uint4 a = bd_load_stream16(reinterpret_cast<const uint4 *>(in) + 2 * i);
        const uint4 b = bd_load_stream16(reinterpret_cast<const uint4 *>(in) + 2 * i + 1);
        const unsigned s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            w[q] = bd_pack16(bd_cvt_f32(__builtin_bit_cast(float, s[2 * q])), bd_cvt_f32(__builtin_bit_cast(float, s[2 * q + 1])));
        bd_store_unit(out, i, w);
    } else if (i == units) {
        for (uint64_t e = units * 8; e < n; e++) out[e] = bd_cvt_f32(in[e]);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bd_stereo_to_mono_kernel(const int16_t *in, int16_t *out, uint64_t n_frames)
{
    const uint64_t i = bd_block_index() * blockDim.x + threadIdx.x;
    if (!VEC) { if (i < n_frames) out[i] = bd_cvt_stereo(in[2 * i], in[2 * i + 1]); return; }
    const uint64_t units = n_frames / 8;
    if (i < units) {
        const uint4 a = bd_load_stream16(reinterpret_cast<const uint4 *>(in) + 2 * i);
        const uint4 b = bd_load_stream16(reinterpret_cast<const uint4 *>(in) + 2 * i + 1);
        const unsigned s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};   // one frame (l, r) per dword
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            w[q] = bd_pack16(bd_cvt_stereo((int16_t)(s[2 * q] & 0xffff), (int16_t)(s[2 * q] >> 16)),
                             bd_cvt_stereo((int16_t)(s[2 * q + 1] & 0xffff), (int16_t)(s[2 * q + 1] >> 16)));
        bd_store_unit(out, i, w);
    } else if (i == units) {
        for (uint64_t e = units * 8; e < n_frames; e++) out[e] = bd_cvt_stereo(in[2 * e], in[2 * e + 1]);
    }
}

// threads of a launch: one per 16-byte unit plus one for the tail, or one per sample
static bool bd_cvt_grid(const void *in, const void *out, uint64_t n, bool *vec, dim3 *grid)
{
    *vec = ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const uint64_t threads = *vec ? n / 8 + 1 : n;
    const uint64_t grid64 = (threads + 255) / 256;
    dim3 grid_dims;
    if (!bd_grid_dims(grid64, &grid_dims)) { bd_set_error("buffer too large for one launch"); return false; }
    *grid = grid_dims;
    return true;
}

extern "C" int bd_u16_data_to_i16(const uint16_t *in, int16_t *out, uint64_t n, void *hip_stream)
{
    if ((!in || !out) && n) { bd_set_error("bd_u16_data_to_i16: null pointer"); return BD_ERR_INVALID_ARG; }
    if (n == 0) return BD_OK;
    bool vec; dim3 grid;
    if (!bd_cvt_grid(in, out, n, &vec, &grid)) return BD_ERR_UNSUPPORTED;
    if (vec) hipLaunchKernelGGL(bd_u16_to_i16_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, in, out, n);
    else     hipLaunchKernelGGL(bd_u16_to_i16_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, in, out, n);
    return bd_check_hip(hipGetLastError(), "launch bd_u16_to_i16_kernel");
}

extern "C" int bd_f32_data_to_i16(const float *in, int16_t *out, uint64_t n, void *hip_stream)
{
    if ((!in || !out) && n) { bd_set_error("bd_f32_data_to_i16: null pointer"); return BD_ERR_INVALID_ARG; }
    if (n == 0) return BD_OK;
    bool vec; dim3 grid;
    if (!bd_cvt_grid(in, out, n, &vec, &grid)) return BD_ERR_UNSUPPORTED;
    if (vec) hipLaunchKernelGGL(bd_f32_to_i16_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, in, out, n);
    else     hipLaunchKernelGGL(bd_f32_to_i16_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, in, out, n);
    return bd_check_hip(hipGetLastError(), "launch bd_f32_to_i16_kernel");
}

extern "C" int bd_stereo_to_mono(const int16_t *interleaved, int16_t *mono, uint64_t n_frames, void *hip_stream)
{
    if ((!interleaved || !mono) && n_frames) { bd_set_error("bd_stereo_to_mono: null pointer"); return BD_ERR_INVALID_ARG; }
    if (n_frames == 0) return BD_OK;
    bool vec; dim3 grid;
    if (!bd_cvt_grid(interleaved, mono, n_frames, &vec, &grid)) return BD_ERR_UNSUPPORTED;
    if (vec) hipLaunchKernelGGL(bd_stereo_to_mono_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, interleaved, mono, n_frames);
    else     hipLaunchKernelGGL(bd_stereo_to_mono_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, interleaved, mono, n_frames);
    return bd_check_hip(hipGetLastError(), "launch bd_stereo_to_mono_kernel");
}

// ---- read-bandwidth probe -----------------------------------------------------
// The ceiling the sample-streaming kernels are measured against: the same 16-byte
// non-temporal loads (256 contiguous bytes per 16 lanes), no arithmetic beyond an
// xor, one dword written per wavefront.  DEPTH loads per lane are in flight; with
// `contiguous` every wavefront walks its own contiguous share of the buffer (as
// the analysis kernels walk their runs), otherwise the wavefronts sweep the
// buffer side by side.
template <int DEPTH>
__global__ __launch_bounds__(256) void bd_probe_read_kernel(const uint4 *src, uint64_t n_units, uint32_t *sink, int contiguous)
{
    const uint64_t wave = (bd_block_index() * 256 + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    constexpr uint64_t STEP = 64 * DEPTH; // units per wavefront step
    uint32_t acc = 0;
    uint64_t base, end, stride;
    if (contiguous) {
        const uint64_t share = (n_units / n_waves) / STEP * STEP;
        base = wave * share; end = base + share; stride = STEP;
    } else {
        base = wave * STEP; end = n_units / STEP * STEP; stride = n_waves * STEP;
    }
    for (; base + STEP <= end; base += stride) {
        uint4 v[DEPTH];
#pragma unroll
        for (int c = 0; c < DEPTH; c++) v[c] = bd_load_stream16(src + base + 64 * c + lane);
#pragma unroll
        for (int c = 0; c < DEPTH; c++) acc ^= v[c].x ^ v[c].y ^ v[c].z ^ v[c].w;
    }
    acc = (uint32_t)__builtin_amdgcn_readfirstlane((int)bd_row_sum((int)acc));
    if (lane == 0 && sink) sink[wave] = acc;
}

extern "C" int bd_probe_read_ex(const void *src, uint64_t bytes, uint32_t *sink, uint32_t waves_per_cu,
                                uint32_t depth, uint32_t contiguous, void *hip_stream)
{
    if (!src || ((uintptr_t)src % 16)) { bd_set_error("bd_probe_read: src must be a 16-byte aligned device pointer"); return BD_ERR_INVALID_ARG; }
    if (waves_per_cu == 0 || waves_per_cu > 32) waves_per_cu = 16;
    const unsigned grid = (256u * waves_per_cu + 3) / 4u; // 256 CUs, 4 wavefronts per block
    const uint4 *p = reinterpret_cast<const uint4 *>(src);
    hipStream_t st = (hipStream_t)hip_stream;
    if (depth == 8) hipLaunchKernelGGL(bd_probe_read_kernel<8>, dim3(grid), dim3(256), 0, st, p, bytes / 16, sink, (int)contiguous);
    else if (depth == 16) hipLaunchKernelGGL(bd_probe_read_kernel<16>, dim3(grid), dim3(256), 0, st, p, bytes / 16, sink, (int)contiguous);
    else if (depth == 2) hipLaunchKernelGGL(bd_probe_read_kernel<2>, dim3(grid), dim3(256), 0, st, p, bytes / 16, sink, (int)contiguous);
    else hipLaunchKernelGGL(bd_probe_read_kernel<4>, dim3(grid), dim3(256), 0, st, p, bytes / 16, sink, (int)contiguous);
    return bd_check_hip(hipGetLastError(), "launch bd_probe_read_kernel");
}

extern "C" int bd_probe_read(const void *src, uint64_t bytes, uint32_t *sink, uint32_t waves_per_cu, void *hip_stream)
{
    return bd_probe_read_ex(src, bytes, sink, waves_per_cu, 4, 0, hip_stream);
}
