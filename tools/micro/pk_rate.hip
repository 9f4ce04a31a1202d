// Issue rate of packed and scalar f32 instructions on gfx950 (tools, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/pk_rate tools/micro/pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHAINS 16
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    v2f a[CHAINS], w = {seed, 1.0f - seed};
    for (int i = 0; i < CHAINS; i++) a[i] = v2f{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
            if (MODE == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
            if (MODE == 3) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(a[(i+1)%CHAINS].y), "v"(w.y));
            if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(a[(i+1)%CHAINS].y), "v"(w.y));
            if (MODE == 6) asm volatile("v_fmamk_f32 %0, %1, 0x3f000123, %0" : "+v"(a[i].x) : "v"(a[(i+1)%CHAINS].y));
            if (MODE == 7) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(a[(i+1)%CHAINS].y), "s"(seed));
            if (MODE == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 10) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 11) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
            if (MODE == 12) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
            if (MODE == 13) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i].x));
            if (MODE == 14) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 15) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(a[i].x));
            if (MODE == 16) asm volatile("v_ashrrev_i32 %0, 16, %0" : "+v"(a[i].x));
            if (MODE == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 18) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 19) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 20) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
            if (MODE == 21) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
            if (MODE == 22) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
            if (MODE == 23) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x));
            if (MODE == 24) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 25) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 26) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));
            if (MODE == 27) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (MODE == 28) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (MODE == 29) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x));
            if (MODE == 30) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].x));
            if (MODE == 31) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 32) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i].x) : "v"(w.x));
            if (MODE == 33) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i].x) : "v"(w.x));
        }
    }
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, float *out)
{
    const int iters = 4096, blocks = 256 * 8;   // 8 blocks of 4 waves per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 16, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * iters * CHAINS
    const double per_simd = (double)blocks * 4 / 1024 * iters * CHAINS;
    printf("%-32s %8.3f ms  %.2f ns per wave-instruction per SIMD (4 cycles at 2.4 GHz = 1.67 ns)\n", name, ms, ms * 1e6 / per_simd);
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32 a=a*w+a", out);
    run<1>("v_fma_f32 a=w*w2+a", out);
    run<2>("v_fmac_f32 a+=w*w2", out);
    run<3>("v_fmac_f32 a+=w*w", out);
    run<4>("v_fmac_f32 a+=b*w (b varies)", out);
    run<5>("v_fma_f32 a=b*w+a (b varies)", out);
    run<6>("v_fmamk_f32 a=b*K+a", out);
    run<7>("v_fma_f32 a=b*s+a (sgpr)", out);
    run<8>("v_mul_f32", out);
    run<9>("v_add_f32", out);
    run<10>("v_max_f32", out);
    run<11>("v_max3_f32", out);
    run<12>("v_max3_f32 |abs|", out);
    run<13>("v_cvt_f32_i32", out);
    run<14>("v_cvt_f32_i32_sdwa", out);
    run<15>("v_bfe_i32", out);
    run<16>("v_ashrrev_i32", out);
    run<17>("v_and_b32", out);
    run<18>("v_add_u32", out);
    run<19>("v_cndmask_b32", out);
    run<20>("v_alignbit_b32", out);
    run<21>("v_dot2c_i32_i16", out);
    run<22>("v_dot2_i32_i16 (vop3p)", out);
    run<23>("v_add_f32_dpp row_ror", out);
    run<24>("v_mov_b32_dpp", out);
    run<25>("v_mov_b32", out);
    run<26>("v_pk_fma_f32", out);
    run<27>("v_pk_add_f32", out);
    run<28>("v_pk_mul_f32", out);
    run<29>("v_sqrt_f32", out);
    run<30>("v_rcp_f32", out);
    run<31>("v_mul_lo_u32", out);
    run<32>("v_mad_u32_u24", out);
    run<33>("v_lshl_add_u32", out);
    return 0;
}
