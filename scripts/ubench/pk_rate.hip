// Issue rate of packed fp32 VALU instructions on gfx950: independent chains, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float c32 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(c32 *out, int iters, c32 seed)
{
    c32 a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (float) (threadIdx.x + i);
    const c32 w = seed + 1.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
                if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
                if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(w.x));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(w));
            }
    }
    c32 s = a[0];
#pragma unroll
    for (int i = 1; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, c32 *out, int wg_per_cu)
{
    const int iters = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, 10, c32{1e-3f, 2e-3f});
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, iters, c32{1e-3f, 2e-3f});
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double) iters * 64 * wg_per_cu;          // one wave of each workgroup per SIMD
    printf("%-22s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz)\n", name,
           wg_per_cu, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
    c32 *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(c32));
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_pk_fma_f32", out, w);
        run<4>("v_pk_fma_f32 op_sel", out, w);
        run<1>("v_pk_add_f32", out, w);
        run<2>("v_pk_mul_f32", out, w);
        run<3>("v_fma_f32", out, w);
    }
    return 0;
}
