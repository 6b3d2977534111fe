// How v_dot2c_f32_bf16 rounds when it is used as "sum += bfloat16": for random sums and random
// bfloat16 values (the second element of the pair zero, the other operand (1, 1)) the result is
// compared with the float add of the same two numbers; reports the largest difference in units
// of the result's last place, and what infinities, NaNs and subnormal values do.
// The bounded harmonic sum (k_harmsum_b* in presto_amd/csrc/accel_kernels.hpp) allows 2^-23 per
// instruction, i.e. one unit.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__global__ void k_dot2(const float *acc, const uint32_t *h16, float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h16[i]), __builtin_bit_cast(bf16x2, 0x3f803f80u),
                                                 acc[i], false);
}

static uint64_t rng = 88172645463325252ull;
static uint32_t rnd()
{
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (uint32_t) (rng >> 16);
}

int main()
{
    const int n = 1 << 24;
    std::vector<float> acc(n), out(n);
    std::vector<uint32_t> h(n);
    for (int i = 0; i < n; i++) {
        // sums and values over the exponents powers take (2^-20 .. 2^20), some far apart, some close
        const uint32_t ea = 107 + rnd() % 40, eh = 107 + rnd() % 40;
        uint32_t abits = (ea << 23) | (rnd() & 0x7FFFFF);
        uint32_t hbits = ((eh << 7) | (rnd() & 0x7F)) & 0xFFFF;
        if (i % 16 == 0) abits = 0;                    // the first add of a chain
        memcpy(&acc[i], &abits, 4);
        h[i] = hbits;
    }
    // special values at the front
    const uint32_t sp_h[] = {0x7F80, 0x7FC0, 0x0001, 0x007F, 0x0000, 0x7F7F, 0x7F7F, 0x3F80};
    const float sp_a[] = {1.f, 1.f, 0.f, 1e-39f, 5.f, 3.3e38f, 1.f, INFINITY};
    for (int i = 0; i < 8; i++) { h[i] = sp_h[i]; acc[i] = sp_a[i]; }
    float *d_acc, *d_out; uint32_t *d_h;
    hipMalloc(&d_acc, n * 4); hipMalloc(&d_out, n * 4); hipMalloc(&d_h, n * 4);
    hipMemcpy(d_acc, acc.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_h, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dot2, dim3(n / 256), dim3(256), 0, 0, d_acc, d_h, d_out, n);
    if (hipMemcpy(out.data(), d_out, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    for (int i = 0; i < 8; i++) {
        uint32_t ob; memcpy(&ob, &out[i], 4);
        printf("special %d: sum %g + bf16 0x%04x -> %g (0x%08x)\n", i, sp_a[i], sp_h[i], out[i], ob);
    }
    long exact = 0, below = 0, above = 0; int64_t maxulp = 0;
    for (int i = 8; i < n; i++) {
        uint32_t vb = h[i] << 16; float v; memcpy(&v, &vb, 4);
        const float want = acc[i] + v; // round to nearest even
        int32_t wb, ob; memcpy(&wb, &want, 4); memcpy(&ob, &out[i], 4);
        const int64_t d = (int64_t) ob - (int64_t) wb;
        if (d == 0) exact++; else if (d < 0) below++; else above++;
        if (llabs(d) > maxulp) maxulp = llabs(d);
    }
    printf("%d adds: %ld equal to the float add, %ld below it, %ld above it; largest difference %lld units in the last place\n",
           n - 8, exact, below, above, (long long) maxulp);
    return maxulp <= 1 ? 0 : 2;
}
