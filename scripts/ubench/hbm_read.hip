// Read-only HBM bandwidth of this part, the ceiling the harmonic sum is priced against: a
// buffer of the cfg2 plane's size (6.76 GB, far beyond the 256 MB Infinity Cache) is summed by
// kernels that differ in load width, in how a workgroup's addresses are laid out and in how
// many loads a lane keeps in flight.  Prints GB/s per variant (best of 5 launches).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o hbm_read hbm_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// grid-stride over 16-byte elements, U loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_stream16(const f4 *__restrict__ p, size_t n, float *out)
{
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++)
            acc += v[u];
    }
    for (; i < n; i += stride)
        acc += p[i];
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f)
        out[0] = s;
}

// each workgroup owns one contiguous span (the harmonic sum's layout along a row): dword or
// 16-byte loads, U in flight
template <int U, typename T>
__global__ __launch_bounds__(256) void k_span(const T *__restrict__ p, size_t per_wg, float *out)
{
    const T *q = p + (size_t) blockIdx.x * per_wg;
    T acc = {};
    size_t i = threadIdx.x;
    for (; i + (U - 1) * 256 < per_wg; i += U * 256) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            v[u] = q[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++)
            acc += v[u];
    }
    float s;
    if constexpr (sizeof(T) == 16)
        s = acc.x + acc.y + acc.z + acc.w;
    else
        s = acc;
    if (s == 123.456f)
        out[0] = s;
}

// the harmonic sum's own shape: one column per lane, walking rows of a [rows][rlen] plane
template <int U>
__global__ __launch_bounds__(256) void k_columns(const float *__restrict__ p, int rows, size_t rlen, float *out)
{
    const size_t c = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (c >= rlen)
        return;
    float acc = 0;
    int r = 0;
    for (; r + U <= rows; r += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            v[u] = p[(size_t) (r + u) * rlen + c];
#pragma unroll
        for (int u = 0; u < U; u++)
            acc += v[u];
    }
    for (; r < rows; r++)
        acc += p[(size_t) r * rlen + c];
    if (acc == 123.456f)
        out[0] = acc;
}

template <typename F>
static void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best)
            best = ms;
    }
    CK(hipGetLastError());
    printf("  %-52s %7.3f ms  %7.1f GB/s  (%.1f %% of 8 TB/s)\n", name, best, bytes / best * 1e-6, bytes / best * 1e-6 / 80.0);
    fflush(stdout);
}

int main()
{
    const int rows = 201;
    const size_t rlen = 2 * (4194303 + 7470); // floats per row, cfg2
    const size_t nfl = (size_t) rows * rlen;
    const double bytes = (double) nfl * 4;
    float *buf, *out;
    CK(hipMalloc(&buf, nfl * 4 + 4096));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, nfl * 4 + 4096));
    printf("read-only sum over %.3f GB\n", bytes * 1e-9);
    const size_t n16 = nfl / 4;
    for (int wgs_per_cu : {4, 8, 16, 32}) {
        const unsigned grid = 256u * wgs_per_cu;
        char nm[96];
        snprintf(nm, sizeof nm, "grid-stride 16 B x1, %d WG/CU", wgs_per_cu);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_stream16<1, false>), dim3(grid), dim3(256), 0, 0, (const f4 *) buf, n16, out); });
        snprintf(nm, sizeof nm, "grid-stride 16 B x4, %d WG/CU", wgs_per_cu);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_stream16<4, false>), dim3(grid), dim3(256), 0, 0, (const f4 *) buf, n16, out); });
        snprintf(nm, sizeof nm, "grid-stride 16 B x4 nontemporal, %d WG/CU", wgs_per_cu);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_stream16<4, true>), dim3(grid), dim3(256), 0, 0, (const f4 *) buf, n16, out); });
        snprintf(nm, sizeof nm, "grid-stride 16 B x8, %d WG/CU", wgs_per_cu);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_stream16<8, false>), dim3(grid), dim3(256), 0, 0, (const f4 *) buf, n16, out); });
    }
    for (size_t span_kb : {64, 256, 1024, 4096}) {
        const size_t per16 = span_kb * 1024 / 16, per4 = span_kb * 1024 / 4;
        const unsigned grid = (unsigned) (n16 / per16);
        char nm[96];
        snprintf(nm, sizeof nm, "span %zu KB per WG, 16 B x4", span_kb);
        timeit(nm, (double) grid * per16 * 16, [&] { hipLaunchKernelGGL((k_span<4, f4>), dim3(grid), dim3(256), 0, 0, (const f4 *) buf, per16, out); });
        snprintf(nm, sizeof nm, "span %zu KB per WG, 4 B x8", span_kb);
        timeit(nm, (double) grid * per4 * 4, [&] { hipLaunchKernelGGL((k_span<8, float>), dim3(grid), dim3(256), 0, 0, (const float *) buf, per4, out); });
    }
    {
        const unsigned grid = (unsigned) ((rlen + 255) / 256);
        timeit("column walk (one column per lane), 4 rows in flight", bytes, [&] { hipLaunchKernelGGL((k_columns<4>), dim3(grid), dim3(256), 0, 0, buf, rows, rlen, out); });
        timeit("column walk, 8 rows in flight", bytes, [&] { hipLaunchKernelGGL((k_columns<8>), dim3(grid), dim3(256), 0, 0, buf, rows, rlen, out); });
        timeit("column walk, 16 rows in flight", bytes, [&] { hipLaunchKernelGGL((k_columns<16>), dim3(grid), dim3(256), 0, 0, buf, rows, rlen, out); });
    }
    return 0;
}
