// spectrum_kernels.hpp -- kernels that prepare the input spectrum on the device:
//
//   k_bigfft_cols / k_bigfft_cols_reg / k_bigfft_rows
//                      one long complex FFT as N = N1 * N2 (two passes of the LDS
//                      transforms of fft_lds.hpp, twiddle multiply in between)
//   k_real_unpack      the real-to-half-complex step of realfft()
//                      (reference src/fastffts.c:198-270): PRESTO's packed .fft layout
//   k_red_medians / k_red_apply
//                      deredden() (reference src/accel_utils.c:1301-1374): medians
//                      of logarithmically growing blocks, linear interpolation of
//                      the local mean between block centres, fast inverse square root
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_lds.hpp"

namespace speck {

using afft::c32;

// exp(-2*pi*i * (p mod n) / n) in double, returned as float2
__device__ __forceinline__ c32 unit_root_neg(int64_t p, int64_t n)
{
    double s, c;
    sincospi(-2.0 * (double) (p % n) / (double) n, &s, &c);
    return c32{(float) c, (float) s};
}

constexpr int kColsPerGroup = 8; // adjacent columns share 64-byte lines: one group touches each line once in L2

// Pass 1, N1 = P::N >= 512: x is viewed as [N1][N2]; for each column b the
// N1-point transform over a (stride N2), times exp(-2*pi*i*b*k1/N), in place.
template <class P>
__global__ __launch_bounds__(P::NT) void k_bigfft_cols(c32 *__restrict__ x, int N2,
                                                      const c32 *__restrict__ tw1,
                                                      const c32 *__restrict__ tw2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    const int64_t n = (int64_t) P::N * N2;
    for (int cc = 0; cc < kColsPerGroup; cc++) {
        const int col = blockIdx.x * kColsPerGroup + cc;
        if (col >= N2)
            break;
        afft::fft3<P, -1>(
            lds, tw1, tw2,
            [&](int, int a, int b) { return x[(size_t) (P::M1 * a + b) * N2 + col]; },
            [&](int k1, c32 v) {
                x[(size_t) k1 * N2 + col] = afft::cmul(v, unit_root_neg((int64_t) col * k1, n));
            });
        __syncthreads();
    }
}

// Pass 1 for short columns, N1 = R <= 32: one thread per column, transform in registers.
template <int R>
__global__ __launch_bounds__(256) void k_bigfft_cols_reg(c32 *__restrict__ x, int N2)
{
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= N2)
        return;
    const int64_t n = (int64_t) R * N2;
    c32 v[R];
#pragma unroll
    for (int a = 0; a < R; a++)
        v[a] = x[(size_t) a * N2 + col];
    afft::Dft<R, -1>::run(v);
#pragma unroll
    for (int k1 = 0; k1 < R; k1++)
        x[(size_t) k1 * N2 + col] = afft::cmul(v[k1], unit_root_neg((int64_t) col * k1, n));
}

// Pass 2: for each k1 the N2-point transform of row y[k1][.]; output element
// k1 + N1*k2 of the long transform.
template <class P>
__global__ __launch_bounds__(P::NT) void k_bigfft_rows(const c32 *__restrict__ y,
                                                      c32 *__restrict__ out, int N1,
                                                      const c32 *__restrict__ tw1,
                                                      const c32 *__restrict__ tw2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    for (int rr = 0; rr < kColsPerGroup; rr++) {
        const int k1 = blockIdx.x * kColsPerGroup + rr;
        if (k1 >= N1)
            break;
        const c32 *row = y + (size_t) k1 * P::N;
        afft::fft3<P, -1>(
            lds, tw1, tw2, [&](int, int a, int b) { return row[P::M1 * a + b]; },
            [&](int k2, c32 v) { out[(size_t) k2 * N1 + k1] = v; });
        __syncthreads();
    }
}

// Half-complex unpack of a real series of n points whose even/odd samples were
// transformed as one complex series of n/2 points.  One thread per pair
// (il, n/2 - il), il = 1 .. n/4; thread 0 also packs DC and Nyquist into bin 0.
// Arithmetic follows the reference statement by statement (float sums, double
// products, rounded to float on store); the rotation exp(-2*pi*i*il/n) is
// evaluated directly instead of by the reference's trigonometric recurrence.
__global__ __launch_bounds__(256) void k_real_unpack(c32 *__restrict__ data, int64_t n)
{
    const int64_t nby2 = n >> 1;
    const int64_t il = (int64_t) blockIdx.x * 256 + threadIdx.x + 1;
    if (il == 1) {
        const c32 d0 = data[0];
        data[0] = c32{d0.x + d0.y, d0.x - d0.y};
    }
    if (il > (n >> 2))
        return;
    const int64_t ih = nby2 - il;
    const c32 lo = data[il], hi = data[ih];
    double wi, wr;
    sincospi(-2.0 * (double) il / (double) n, &wi, &wr);
    const double cc = -0.5;
    const double h1r = 0.5 * (double) (lo.x + hi.x);
    const double h1i = 0.5 * (double) (lo.y - hi.y);
    const double h2r = -cc * (double) (lo.y + hi.y);
    const double h2i = cc * (double) (lo.x - hi.x);
    const double h2rwr = h2r * wr, h2rwi = h2r * wi, h2iwr = h2i * wr, h2iwi = h2i * wi;
    if (il != ih)
        data[il] = c32{(float) (h1r + h2rwr - h2iwi), (float) (h1i + h2iwr + h2rwi)};
    data[ih] = c32{(float) (h1r - h2rwr + h2iwi), (float) (-h1i + h2iwr + h2rwi)};
}

// ---------------------------------------------------------------------------
// Lengths that are not powers of two: the chirp-z identity
//   X[k] = w[k] * sum_j (x[j] w[j]) conj(w[k-j]),   w[k] = exp(-i*pi*k^2/L),
// i.e. one circular convolution of length M >= 2L-1 (a power of two) done with
// the long transforms above.  k^2 is reduced mod 2L in integers before the
// phase is formed, so the chirp is exact to fp64 round-off at any k.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void chirp(int64_t k, int64_t L, double *c, double *s)
{
    const int64_t k2 = (k * k) % (2 * L);
    sincospi(-(double) k2 / (double) L, s, c);
}

// a[k] = x[k] * w[k] (k < L), 0 up to M
__global__ __launch_bounds__(256) void k_blu_pre(const c32 *__restrict__ x, c32 *__restrict__ a,
                                                int64_t L, int64_t M)
{
    const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= M)
        return;
    c32 v = c32{0.f, 0.f};
    if (k < L) {
        double c, s;
        chirp(k, L, &c, &s);
        const c32 d = x[k];
        v = c32{(float) ((double) d.x * c - (double) d.y * s), (float) ((double) d.x * s + (double) d.y * c)};
    }
    a[k] = v;
}

// b[k] = conj(w[k]) for |k| < L (negative k wrapped to M - |k|), 0 elsewhere
__global__ __launch_bounds__(256) void k_blu_filter(c32 *__restrict__ b, int64_t L, int64_t M)
{
    const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= M)
        return;
    const int64_t j = k < L ? k : (M - k < L ? M - k : -1);
    c32 v = c32{0.f, 0.f};
    if (j >= 0) {
        double c, s;
        chirp(j, L, &c, &s);
        v = c32{(float) c, (float) -s};
    }
    b[k] = v;
}

// a[k] = conj(a[k] * b[k]): the inverse transform is conj(FFT(conj(.))) / M
__global__ __launch_bounds__(256) void k_blu_mul(c32 *__restrict__ a, const c32 *__restrict__ b, int64_t M)
{
    const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= M)
        return;
    const c32 p = afft::cmul(a[k], b[k]);
    a[k] = c32{p.x, -p.y};
}

// X[k] = w[k] * conj(c[k]) / M
__global__ __launch_bounds__(256) void k_blu_post(const c32 *__restrict__ cbuf, c32 *__restrict__ X,
                                                 int64_t L, int64_t M)
{
    const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= L)
        return;
    double c, s;
    chirp(k, L, &c, &s);
    const double inv = 1.0 / (double) M;
    const double re = (double) cbuf[k].x * inv, im = -(double) cbuf[k].y * inv;
    X[k] = c32{(float) (re * c - im * s), (float) (re * s + im * c)};
}

// ---------------------------------------------------------------------------
// Red-noise removal.
// ---------------------------------------------------------------------------
constexpr int kMaxRedBlock = 1024; // longest median block (16 values per lane)

// mean[j] = median(|fft|^2 over block j) / ln 2.  One wavefront per block; the
// median is the order statistic (len-1)/2 found by a bitwise radix select.
__global__ __launch_bounds__(256) void k_red_medians(const c32 *__restrict__ fft,
                                                    const int64_t *__restrict__ blk_start,
                                                    const int *__restrict__ blk_len, int nblk,
                                                    float *__restrict__ mean)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= nblk)
        return; // whole wavefront leaves together
    const int lane = threadIdx.x & 63;
    const int64_t start = blk_start[j];
    const int len = blk_len[j];
    uint32_t pw[kMaxRedBlock / 64];
#pragma unroll
    for (int e = 0; e < kMaxRedBlock / 64; e++) {
        const int idx = lane + 64 * e;
        pw[e] = 0xFFFFFFFFu;
        if (idx < len) {
            const c32 v = fft[start + idx];
            pw[e] = __float_as_uint(afft::strict::add(afft::strict::mul(v.x, v.x), afft::strict::mul(v.y, v.y)));
        }
    }
    const int k = (len - 1) / 2;
    uint32_t prefix = 0;
    for (int bit = 30; bit >= 0; bit--) {
        const uint32_t cand = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int e = 0; e < kMaxRedBlock / 64; e++)
            c += __popcll(__ballot(pw[e] < cand));
        if (c <= k)
            prefix = cand;
    }
    if (lane == 0)
        mean[j] = (float) ((double) __uint_as_float(prefix) / 0.69314718055994530942);
}

struct RedSegment {
    int64_t start, end; // bins [start, end) scaled by this segment
    int mean_idx;       // block whose mean is the value at the segment's first bin
    int slope_a;        // slope = (mean[slope_a+1] - mean[slope_a]) / (0.5*lensum); -1: zero; -2: one
    int lensum;         // lengths of the two blocks the slope runs between
    int copy_re;        // 1: imaginary part takes re*norm (last chunk of the rednoise program)
};

// One wavefront per segment: fft[ind] *= invsqrtf(mean_old + dslope*(ind-start)).
__global__ __launch_bounds__(256) void k_red_apply(c32 *__restrict__ fft,
                                                  const RedSegment *__restrict__ seg, int nseg,
                                                  const float *__restrict__ mean)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nseg)
        return;
    const int lane = threadIdx.x & 63;
    const RedSegment sg = seg[s];
    const float mean_old = mean[sg.mean_idx];
    float dslope = 0.0f;
    if (sg.slope_a == -2)
        dslope = 1.0f;
    else if (sg.slope_a >= 0)
        dslope = (float) ((double) afft::strict::sub(mean[sg.slope_a + 1], mean[sg.slope_a]) /
                          (0.5 * (double) sg.lensum));
    for (int64_t ind = sg.start + lane; ind < sg.end; ind += 64) {
        const float x = afft::strict::add(mean_old, afft::strict::mul(dslope, (float) (int) (ind - sg.start)));
        const float y = __int_as_float(0x5f3759df - (__float_as_int(x) >> 1));
        const float nrm = afft::strict::mul(y, afft::strict::sub(1.5f, afft::strict::mul(afft::strict::mul(afft::strict::mul(0.5f, x), y), y)));
        c32 v = fft[ind];
        const float re = afft::strict::mul(v.x, nrm);
        v.y = sg.copy_re ? re : afft::strict::mul(v.y, nrm);
        v.x = re;
        fft[ind] = v;
    }
    if (s == 0 && lane == 0)
        fft[0] = c32{1.0f, 0.0f};
}

// ---------------------------------------------------------------------------
// zapbirds() (reference src/zapping.c:29-81): every bin of a birdie range is set
// to (avgamp, 0), avgamp = sqrt(mean of the median powers of the 200 bins below
// and the 200 bins at the top of the range / ln 2).  Birds are applied in order
// and a later one sees what earlier ones wrote, so one workgroup walks the list;
// wavefront 0 and 1 find the two medians, then all threads write the range.
// ---------------------------------------------------------------------------
__device__ inline float wave_median200(const c32 *__restrict__ fft, int64_t numbins, int64_t start)
{
    const int lane = threadIdx.x & 63;
    uint32_t pw[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int idx = lane + 64 * e;
        pw[e] = 0xFFFFFFFFu;
        if (idx < 200) {
            const int64_t b = start + idx;
            float p = 0.0f;
            if (b >= 0 && b < numbins)
                p = afft::strict::norm2(fft[b].x, fft[b].y);
            pw[e] = __float_as_uint(p);
        }
    }
    const int k = (200 - 1) / 2;
    uint32_t prefix = 0;
    for (int bit = 30; bit >= 0; bit--) {
        const uint32_t cand = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int e = 0; e < 4; e++)
            c += __popcll(__ballot(pw[e] < cand));
        if (c <= k)
            prefix = cand;
    }
    return __uint_as_float(prefix);
}

__global__ __launch_bounds__(256) void k_zapbirds(c32 *__restrict__ fft, int64_t numbins,
                                                 const double *__restrict__ lobins,
                                                 const double *__restrict__ hibins, int n)
{
    __shared__ float s_med[2];
    const int wave = threadIdx.x >> 6;
    const double top = (double) numbins;
    for (int k = 0; k < n; k++) {
        const double lobin = lobins[k];
        double hibin = hibins[k];
        if (lobin >= top - 1)
            break;
        if (hibin >= top)
            hibin = top - 1;
        if (!(lobin - 300.0 > 1))
            continue;
        const int ilobin = (int) floor(lobin), ihibin = (int) ceil(hibin);
        if (wave == 0) {
            const float m = wave_median200(fft, numbins, (int64_t) (int) (lobin - 200) - 100);
            if (threadIdx.x == 0) s_med[0] = m;
        } else if (wave == 1) {
            const float m = wave_median200(fft, numbins, (int64_t) (int) hibin - 100);
            if ((threadIdx.x & 63) == 0) s_med[1] = m;
        }
        __syncthreads();
        // float sum of the medians, then double arithmetic, rounded to float
        const float avgamp = (float) sqrt(0.5 * (double) afft::strict::add(s_med[0], s_med[1]) / 0.69314718055994530942);
        for (int64_t b = (int64_t) ilobin + threadIdx.x; b < ihibin && b < numbins; b += 256)
            fft[b] = c32{avgamp, 0.0f};
        __threadfence_block();
        __syncthreads(); // the next bird may read what this one wrote
    }
}

} // namespace speck
