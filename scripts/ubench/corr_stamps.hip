// Phase timing inside the plane-generation kernel (k_correlate_rr, 8192-point plan, one block
// per workgroup looping over 16 templates): a diagnostic build of the same kernel records the
// shader clock at each phase boundary of every transform, for wave 0 of every workgroup.
// Random inputs of the cfg2 shape (201 templates x 1123 blocks); results are not checked.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I presto_amd/csrc -I include -o corr_stamps corr_stamps.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "accel_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NPH = 10, NIT = 16;
struct CycleStamp {
    unsigned long long *buf; // [workgroup][iteration][phase]
    int iter;
    __device__ __forceinline__ void begin(int i) { iter = i; }
    __device__ __forceinline__ void operator()(int k) const
    {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0 && iter < NIT)
            buf[((size_t) blockIdx.x * NIT + iter) * NPH + k] = t;
        __builtin_amdgcn_sched_barrier(0);
    }
};

typedef afft::Plan<32, 16, 16, 256> P8K;
using accelk::c32;

int main(int argc, char **argv)
{
    const int numz = 201, nblk = 1123, chunk = 16, halfwidth = 114;
    const int lds_pad = argc > 1 ? atoi(argv[1]) : 0;
    const int64_t rlen = 2 * (4194303 + 7470);
    c32 *bank, *blockfft, *pw1, *pw2;
    float *plane;
    int64_t *col;
    int *ncols;
    unsigned long long *stamps;
    std::vector<c32> h((size_t) numz * 8192);
    srand(1);
    for (auto &x : h) x = c32{(float) rand() / RAND_MAX - 0.5f, (float) rand() / RAND_MAX - 0.5f};
    CK(hipMalloc(&bank, h.size() * 8));
    CK(hipMemcpy(bank, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    h.resize((size_t) nblk * 4096);
    CK(hipMalloc(&blockfft, h.size() * 8));
    CK(hipMemcpy(blockfft, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&pw1, 16 * 256 * 8));
    CK(hipMalloc(&pw2, 16 * 16 * 8));
    CK(hipMemcpy(pw1, h.data(), 16 * 256 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(pw2, h.data(), 16 * 16 * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&plane, (size_t) rlen * numz * 4));
    std::vector<int64_t> hc(nblk);
    std::vector<int> hn(nblk, 7470);
    for (int b = 0; b < nblk; b++) hc[b] = 1072 + (int64_t) b * 7470;
    CK(hipMalloc(&col, nblk * 8));
    CK(hipMalloc(&ncols, nblk * 4));
    CK(hipMemcpy(col, hc.data(), nblk * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(ncols, hn.data(), nblk * 4, hipMemcpyHostToDevice));
    const int nchunks = (numz + chunk - 1) / chunk, tz = 4, tc = 16;
    const long long tiles = (long long) ((nblk + tz - 1) / tz) * ((nchunks + tc - 1) / tc);
    const unsigned grid = (unsigned) (((tiles + 7) / 8) * 8 * tz * tc);
    CK(hipMalloc(&stamps, (size_t) grid * NIT * NPH * 8));
    CK(hipMemset(stamps, 0, (size_t) grid * NIT * NPH * 8));
    auto kern = accelk::k_correlate_rr<P8K, false, 16, true, CycleStamp>;
    auto plain = accelk::k_correlate_rr<P8K, false, 16, true>;
    CK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8K::LDS_BYTES + 20000));
    CK(hipFuncSetAttribute((const void *) plain, hipFuncAttributeMaxDynamicSharedMemorySize, P8K::LDS_BYTES + 20000));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_plain = 0, ms_stamp = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(plain, dim3(grid), dim3(256), P8K::LDS_BYTES + lds_pad, 0, bank, blockfft, col, ncols, nblk, chunk,
                           plane, rlen, halfwidth, pw1, pw2, numz, tz, tc, afft::NoStamp(), (unsigned short *) nullptr, (unsigned short *) nullptr, (unsigned short *) nullptr, (int64_t) 0, (const int2 *) nullptr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_plain, e0, e1));
    }
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), P8K::LDS_BYTES + lds_pad, 0, bank, blockfft, col, ncols, nblk, chunk,
                           plane, rlen, halfwidth, pw1, pw2, numz, tz, tc, CycleStamp{stamps, 0}, (unsigned short *) nullptr, (unsigned short *) nullptr, (unsigned short *) nullptr, (int64_t) 0, (const int2 *) nullptr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_stamp, e0, e1));
    }
    std::vector<unsigned long long> hs((size_t) grid * NIT * NPH);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    // median over workgroups and iterations 2..13 of each phase's length
    const char *names[NPH] = {"", "template loads + products + pass 1", "first image writes", "barrier", "first image reads issued",
                              "reads landed + barrier", "pass 2 + second image writes", "barrier",
                              "second image reads + pass 3 + stores issued", "trailing barrier"};
    printf("plain kernel %.3f ms, stamped kernel %.3f ms (LDS pad %d)\n", ms_plain, ms_stamp, lds_pad);
    double total = 0;
    for (int k = 1; k < NPH; k++) {
        std::vector<double> d;
        for (unsigned w = 0; w < grid; w++)
            for (int it = 2; it < 12; it++) {
                const unsigned long long *p = &hs[((size_t) w * NIT + it) * NPH];
                if (p[k] > p[k - 1] && p[0]) d.push_back((double) (p[k] - p[k - 1]));
            }
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        const double med = d[d.size() / 2];
        total += med;
        printf("  %-46s median %7.0f  p10 %7.0f  p90 %7.0f cycles\n", names[k], med, d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    printf("  sum of medians %.0f cycles per transform\n", total);
    return 0;
}
