// Internal declarations shared by the translation units of libbeatdet_hip.so.
#ifndef BD_INTERNAL_H
#define BD_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/beat_detector_hip.h"

#define BD_RING_LEN 1024  // src/strategies/spectrum.rs:43
#define BD_HALF_BLOCK 512 // granule of the fast Spectrum kernel

// Everything the kernels need, derived once on the host from
// (kind, sampling_rate, bd_params) with the reference's f32 arithmetic.
struct bd_plan {
    int      kind;
    uint32_t sampling_rate;
    float    ms_per_sample;   // src/strategies/mod.rs:64
    uint32_t min_amp;
    uint32_t gap_ms;
    // Spectrum
    int      first_bin, last_bin, n_bins;
    float    divisor;         // 0 => no scaling
    float    threshold;
    uint32_t dc_packs_nyquist;
    // LPF
    float    alpha;
    int      alpha_valid;
    float    lpf_ratio;
};

int  bd_make_plan(int kind, uint32_t sampling_rate, const bd_params *params, bd_plan *plan);
void bd_clock_step(const bd_plan &plan, uint32_t callback_len, uint32_t *step_ms, uint32_t *half_ms);

void bd_set_error(const char *fmt, ...);
void bd_set_last_kernel(const char *name);
int  bd_check_hip(hipError_t e, const char *what);

// Experiment switches.  They are read from the environment once, when the library first needs
// them, never on a launch path; bd_tuning_reload() (tests and experiment drivers) reads them again.
// Defaults (every variable unset) are the measured best and are what a program gets.
struct bd_tuning {
    int      spec_force_granule;     // BD_SPEC_FORCE_GRANULE: 512 / 1024 through the granule kernel
    int      spec_run;               // BD_SPEC_RUN: half-blocks per task of the half-block kernel (0 = automatic)
    int      spec_lds_pad;           // BD_SPEC_LDS_PAD: unused dynamic LDS, caps the half-block kernel's residency (-1 = default)
    int      spec_no_split;          // BD_SPEC_NO_SPLIT: no split kernel
    int      spec_no_quarter;        // BD_SPEC_NO_QUARTER: 256-sample callbacks through the granule kernel
    int      refractory_wave;        // BD_REFRACTORY_WAVE: the refractory pass with a wavefront per stream for all rows
    int      spec_no_granule;        // BD_SPEC_NO_GRANULE: generic kernel for everything off the half-block path
    int      spec_no_small_granule;  // BD_SPEC_NO_SMALL_GRANULE: no granules below 64 samples
    int      spec_depth;             // BD_SPEC_DEPTH: register buffers of the half-block kernel (2, 3, 4; 0 = default)
    int      spec_store;             // BD_SPEC_STORE: result-store variant of the half-block kernel (0 = default)
    int      split_no_rem;           // BD_SPLIT_NO_REM: remainder as a masked piece of the main loop
    int      split_subruns;          // BD_SPLIT_SUBRUNS: runs of 64 callbacks per wavefront of the split kernel (0 = automatic)
    int      split_stage;            // BD_SPLIT_STAGE: 1 = every length through the staged copy, 2 = rows straight from memory wherever possible
    int      lpf_variant;            // BD_LPF_VARIANT: chunk size / wavefronts per workgroup
    int      lpf_groups;             // BD_LPF_GROUPS: groups of 64 windows per wavefront (aligned staging), 0 = chosen by the launcher
    unsigned host_threads;           // BD_HOST_THREADS: packing threads of bd_analyze_batch_host (0 = default)
    uint64_t host_chunk_bytes;       // BD_HOST_CHUNK_BYTES: staging chunk of bd_analyze_batch_host (0 = default)
};
const bd_tuning &bd_tuning_get();

#define BD_HIP(call) do { int _s = bd_check_hip((call), #call); if (_s != BD_OK) return _s; } while (0)

// stage launchers (device pointers, enqueue only)
int bd_launch_spectrum(const bd_plan &plan, const int16_t *samples, uint64_t n_streams,
                       uint64_t n_samples, uint64_t stride, uint32_t callback_len,
                       float *values, uint8_t *flags, uint64_t out_stride, hipStream_t st);
#define BD_GR_MAX_BINS 11 // non-DC bins the granule kernel can evaluate (granules of 64 samples)
uint32_t bd_spectrum_granule_size(uint32_t callback_len);
int bd_spectrum_granule_max_bins(uint32_t gran);
int bd_launch_spectrum_granule(const bd_plan &plan, const int16_t *samples, uint64_t n_streams,
                               uint64_t n_samples, uint64_t stride, uint32_t callback_len,
                               float *values, uint8_t *flags, uint64_t out_stride, hipStream_t st);
bool bd_spectrum_split_applies(const bd_plan &plan, uint32_t callback_len);
int bd_launch_spectrum_split(const bd_plan &plan, const int16_t *samples, uint64_t n_streams, uint64_t n_cb_eval,
                             uint64_t stride, uint32_t callback_len, float *values, uint8_t *flags,
                             uint64_t out_stride, hipStream_t st);
int bd_launch_lpf(const bd_plan &plan, const int16_t *samples, uint64_t n_streams,
                  uint64_t n_samples, uint64_t stride, uint32_t callback_len,
                  float *values, uint8_t *flags, uint64_t out_stride, hipStream_t st);
int bd_launch_refractory(const bd_plan &plan, uint32_t callback_len, const uint8_t *flags,
                         uint64_t flags_stride, uint64_t n_streams, uint64_t n_cb, uint32_t *beat_count,
                         uint32_t *beat_idx, uint32_t *beat_ms, uint32_t max_beats,
                         hipStream_t st, const uint64_t *lengths = nullptr);

// One grid dimension holds fewer than 2^32 threads (blocks x threads per block); the runtime
// cuts a larger launch short without an error.  Launches therefore fold their blocks into two
// dimensions, rows of at most BD_GRID_ROW_BLOCKS blocks (of at most 256 threads), and kernels
// number their blocks with bd_block_index(); blocks past the last task find nothing to do.
#define BD_GRID_ROW_BLOCKS 0x400000ull
static inline bool bd_grid_dims(uint64_t blocks, dim3 *grid)
{
    if (blocks == 0) blocks = 1;
    if (blocks <= BD_GRID_ROW_BLOCKS) { *grid = dim3((unsigned)blocks, 1, 1); return true; }
    const uint64_t rows = (blocks + BD_GRID_ROW_BLOCKS - 1) / BD_GRID_ROW_BLOCKS;
    if (rows > 65535ull) return false;
    *grid = dim3((unsigned)((blocks + rows - 1) / rows), (unsigned)rows, 1);
    return true;
}

// Rust `as` casts on the host
static inline uint32_t bd_f32_as_u32(float x)
{
    if (x != x) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 4294967296.0f) return 4294967295u;
    return (uint32_t)x;
}

// ---- device helpers -------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ uint64_t bd_block_index()
{
    return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
}

// 16-byte streaming load (data is read once: keep it out of the way in the caches)
typedef unsigned int bd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 bd_load_stream16(const void *p)
{
    const bd_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const bd_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Result stores.  Outputs are a trickle (0.5 % of the bytes) inside a saturated
// read stream; written with the default policy they cost 3 - 15 % of the kernel's
// time, depending on where the driver happened to place the sample buffer.
// Non-temporal, write-through stores (nt sc1) remove most of that cost
// (15.8 -> 13.8 ms in the bad placement, 13.6 -> 13.3 ms in the good one, hop-512
// Spectrum workload).  The policy bits only exist on buffer/global instructions,
// so stores go through a buffer descriptor over the wavefront's output range
// [base, base + bytes): `base` must be wave-uniform.
#define BD_AUX_STREAM_STORE 18 /* nt (2) | sc1 (16) */
// A descriptor over [base, base + bytes), both the same in every lane of the wavefront.  Its four
// dwords go through readfirstlane: a base that the compiler keeps in vector registers (anything
// computed with a 64-bit division, for one) otherwise wraps every access in a loop over the distinct
// descriptors of the wavefront -- four readfirstlanes, two 64-bit compares and a branch per access.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bd_uniform_rsrc(const void *base, unsigned bytes)
{
    const uint64_t b = reinterpret_cast<uintptr_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const uint32_t n = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>((uintptr_t)(((uint64_t)hi << 32) | lo)), 0, (int)n,
                                             0x00020000);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bd_out_rsrc(void *base, unsigned bytes)
{
    return bd_uniform_rsrc(base, bytes);
}
__device__ __forceinline__ void bd_store_stream_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, BD_AUX_STREAM_STORE);
}
__device__ __forceinline__ void bd_store_stream_u8(__amdgpu_buffer_rsrc_t r, unsigned byte_off, uint8_t v)
{
    __builtin_amdgcn_raw_buffer_store_b8(v, r, byte_off, 0, BD_AUX_STREAM_STORE);
}

template <int CTRL>
__device__ __forceinline__ float bd_dpp_f(float v)
{
    return __builtin_bit_cast(float,
        __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int bd_dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

// all-reduce inside each row of 16 lanes (row_ror 8,4,2,1)
__device__ __forceinline__ float bd_row_sum(float v)
{
    v += bd_dpp_f<0x128>(v); v += bd_dpp_f<0x124>(v);
    v += bd_dpp_f<0x122>(v); v += bd_dpp_f<0x121>(v);
    return v;
}
__device__ __forceinline__ int bd_row_sum(int v)
{
    v += bd_dpp_i<0x128>(v); v += bd_dpp_i<0x124>(v);
    v += bd_dpp_i<0x122>(v); v += bd_dpp_i<0x121>(v);
    return v;
}
__device__ __forceinline__ float bd_row_max(float v)
{
    v = fmaxf(v, bd_dpp_f<0x128>(v)); v = fmaxf(v, bd_dpp_f<0x124>(v));
    v = fmaxf(v, bd_dpp_f<0x122>(v)); v = fmaxf(v, bd_dpp_f<0x121>(v));
    return v;
}

// value of lane 15 of the previous row for rows 1..3; `row0` for row 0
__device__ __forceinline__ float bd_prev_row(float cur, float row0)
{
    return __builtin_bit_cast(float,
        __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, row0), __builtin_bit_cast(int, cur),
                                    0x142 /*row_bcast:15*/, 0xe, 0xf, false));
}
__device__ __forceinline__ int bd_prev_row(int cur, int row0)
{
    return __builtin_amdgcn_update_dpp(row0, cur, 0x142, 0xe, 0xf, false);
}
__device__ __forceinline__ float bd_lane63(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int bd_lane63(int v) { return __builtin_amdgcn_readlane(v, 63); }

// full-wave reductions (butterfly through the LDS crossbar; used off the hot path)
__device__ __forceinline__ float bd_wave_sum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float bd_wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

#endif // __HIPCC__
#endif
