// accel_kernels.hpp -- gfx950 kernels of the acceleration-search hot path.
//
//   k_gen_zresp      z-response templates in fp64 (gen_z_response / gen_r_response,
//                    reference src/responses.c:165-322) placed for correlation
//                    (place_complex_kernel, src/corr_prep.c:58-80)
//   k_fft_rows       forward FFT of each placed template (init_kernel, src/accel_utils.c:150)
//   k_block_prep     per block: window fetch, exact block-median normalisation,
//                    forward FFT (src/accel_utils.c:929-983)
//   k_correlate_rr   per (z, block): template multiply, inverse FFT, |.|^2 / N^2,
//   k_correlate      row store into the plane (src/accel_utils.c:1002-1051, 1076-1090);
//                    twiddles in registers / read from tables
//   k_harmsum        in-memory harmonic sum + threshold + wave-ballot compaction
//                    (src/accel_utils.c:1160-1208, 1259-1299)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_lds.hpp"

namespace accelk {

using afft::c32;

constexpr int USELEN = 7470;

constexpr int kUseLen = 7470; // columns per block (ACCELGPU_USELEN)
// entries per block and row in the plane of group maxima (see k_prefilter): 1868 whether groups start at column 0 or 2
constexpr int kGpb = (kUseLen + 2 + 3) >> 2;

struct DevHit {          // 16 bytes, written by k_harmsum
    float power;
    uint32_t stage_zind; // stage << 16 | zind
    int64_t rind;
};

// ---------------------------------------------------------------------------
// Fresnel integrals in fp64 (Cephes rational approximations, the same ones the
// reference's src/fresnl.c:175-216 evaluates).
// ---------------------------------------------------------------------------
// Coefficient tables live in constant memory at namespace scope and the Horner
// loops unroll at compile time: a table declared inside the function would be
// rebuilt in private memory on every call.
namespace fres {
__device__ __constant__ const double sn[6] = {-2.99181919401019853726E3, 7.08840045257738576863E5,
                                              -6.29741486205862506537E7, 2.54890880573376359104E9,
                                              -4.42979518059697779103E10, 3.18016297876567817986E11};
__device__ __constant__ const double sd[6] = {2.81376268889994315696E2, 4.55847810806532581675E4,
                                              5.17343888770096400730E6, 4.19320245898111231129E8,
                                              2.24411795645340920940E10, 6.07366389490084639049E11};
__device__ __constant__ const double cn[6] = {-4.98843114573573548651E-8, 9.50428062829859605134E-6,
                                              -6.45191435683965050962E-4, 1.88843319396703850064E-2,
                                              -2.05525900955013891793E-1, 9.99999999999999998822E-1};
__device__ __constant__ const double cd[7] = {3.99982968972495980367E-12, 9.15439215774657478799E-10,
                                              1.25001862479598821474E-7, 1.22262789024179030997E-5,
                                              8.68029542941784300606E-4, 4.12142090722199792936E-2,
                                              1.00000000000000000118E0};
__device__ __constant__ const double fn[10] = {4.21543555043677546506E-1, 1.43407919780758885261E-1,
                                               1.15220955073585758835E-2, 3.45017939782574027900E-4,
                                               4.63613749287867322088E-6, 3.05568983790257605827E-8,
                                               1.02304514164907233465E-10, 1.72010743268161828879E-13,
                                               1.34283276233062758925E-16, 3.76329711269987889006E-20};
__device__ __constant__ const double fd[10] = {7.51586398353378947175E-1, 1.16888925859191382142E-1,
                                               6.44051526508858611005E-3, 1.55934409164153020873E-4,
                                               1.84627567348930545870E-6, 1.12699224763999035261E-8,
                                               3.60140029589371370404E-11, 5.88754533621578410010E-14,
                                               4.52001434074129701496E-17, 1.25443237090011264384E-20};
__device__ __constant__ const double gn[11] = {5.04442073643383265887E-1, 1.97102833525523411709E-1,
                                               1.87648584092575249293E-2, 6.84079380915393090172E-4,
                                               1.15138826111884280931E-5, 9.82852443688422223854E-8,
                                               4.45344415861750144738E-10, 1.08268041139020870318E-12,
                                               1.37555460633261799868E-15, 8.36354435630677421531E-19,
                                               1.86958710162783235106E-22};
__device__ __constant__ const double gd[11] = {1.47495759925128324529E0, 3.37748989120019970451E-1,
                                               2.53603741420338795122E-2, 8.14679107184306179049E-4,
                                               1.27545075667729118702E-5, 1.04314589657571990585E-7,
                                               4.60680728146520428211E-10, 1.10273215066240270757E-12,
                                               1.38796531259578871258E-15, 8.39158816283118707363E-19,
                                               1.86958710162783236342E-22};
} // namespace fres

template <int N> __device__ __forceinline__ double horner(double x, const double (&c)[N])
{
    double acc = c[0];
#pragma unroll
    for (int k = 1; k < N; k++)
        acc = acc * x + c[k];
    return acc;
}
template <int N> __device__ __forceinline__ double horner1(double x, const double (&c)[N])
{
    double acc = x + c[0];
#pragma unroll
    for (int k = 1; k < N; k++)
        acc = acc * x + c[k];
    return acc;
}

__device__ inline void fresnel(double xin, double *S, double *C)
{
    using namespace fres;
    const double PI = 3.14159265358979323846;
    const double x = fabs(xin), x2 = x * x;
    double ss, cc;
    if (x2 < 2.5625) {
        const double t = x2 * x2;
        ss = x * x2 * horner(t, sn) / horner1(t, sd);
        cc = x * horner(t, cn) / horner(t, cd);
    } else if (x > 36974.0) {
        ss = cc = 0.5;
    } else {
        double t = PI * x2;
        const double u = 1.0 / (t * t);
        t = 1.0 / t;
        const double f = 1.0 - u * horner(u, fn) / horner1(u, fd);
        const double g = t * horner(u, gn) / horner1(u, gd);
        double s, c;
        sincospi(0.5 * x2, &s, &c); // 0.5*pi*x^2 with the reduction done on x^2/2
        const double pix = PI * x;
        cc = 0.5 + (f * s - g * c) / pix;
        ss = 0.5 - (f * c + g * s) / pix;
    }
    *S = xin < 0.0 ? -ss : ss;
    *C = xin < 0.0 ? -cc : cc;
}

// One thread per template sample.  Sample ii of kernel zi (z = -zmax + 2*zi,
// numkern = 4*m(z) samples at half-bin spacing, roffset = 0) goes to its wrapped
// position in a zero-filled row of `fftlen` points: the second half of the
// response first, the first half at the tail.
__global__ void k_gen_zresp(c32 *__restrict__ bank, const int *__restrict__ halfwidths, int zmax,
                            int fftlen, int max_numkern)
{
    const int zi = blockIdx.y;
    const int ii = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = halfwidths[zi];
    const int numkern = 4 * m;
    if (ii >= numkern)
        return;
    const double PI = 3.14159265358979323846;
    const double z = (double) (-zmax + 2 * zi);
    const double absz = fabs(z);
    double re, im;
    if (absz < 1e-4) {
        // sinc interpolation kernel, gen_r_response with roffset = 0
        const double r = PI * (numkern / 4.0) + (double) ii * (-PI / 2.0);
        double s, c;
        sincos(r, &s, &c);
        const double sinc = (r == 0.0) ? 1.0 : s / r;
        re = c * sinc;
        im = s * sinc;
        if (ii == numkern / 2) { // series value at the removable singularity
            re = 1.0;
            im = 0.0;
        }
    } else {
        const double signz = z < 0.0 ? -1.0 : 1.0;
        const double zd = signz * 1.4142135623730950488 / sqrt(absz);
        const double cons = zd / 2.0;
        const double pibyz = PI / z;
        // the reference walks r by repeated addition of -0.5 from startr; every
        // partial sum is exact in fp64 (multiples of 0.5), so this is the same r
        const double r = (-0.5 * z + numkern / 4.0) - 0.5 * (double) ii;
        const double yy = r * zd;
        const double zz = yy + z * zd;
        double s, c, sy, cy, sz, cz;
        sincos(pibyz * r * r, &s, &c);
        fresnel(yy, &sy, &cy);
        fresnel(zz, &sz, &cz);
        const double tr = signz * (cz - cy);
        const double ti = sy - sz;
        re = (tr * c - ti * s) * cons;
        im = -(tr * s + ti * c) * cons;
    }
    const int half = numkern / 2;
    const int pos = (ii >= half) ? (ii - half) : (fftlen - half + ii);
    bank[(size_t) zi * fftlen + pos] = c32{(float) re, (float) im};
}

// In-place transform of each row of `data` (N points per row).
template <class P, int DIR>
__global__ __launch_bounds__(P::NT) void k_fft_rows(c32 *__restrict__ data,
                                                   const c32 *__restrict__ tw1,
                                                   const c32 *__restrict__ tw2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    c32 *row = data + (size_t) blockIdx.x * P::N;
    // all loads of the row happen in pass 1, all stores in pass 3, with
    // barriers in between: in place is safe
    afft::fft3<P, DIR>(
        lds, tw1, tw2, [&](int, int a, int b) { return row[P::M1 * a + b]; },
        [&](int n, c32 v) { row[n] = v; });
}

// ---------------------------------------------------------------------------
// Block preparation.  One workgroup per block of the reference grid.
//   window  : NF = fftlen/2 bins starting at lobin_b = rlo_b - halfwidth, zero
//             outside the file (get_fourier_amplitudes)
//   norm    : 1/sqrt(median(|d|^2 over the first numdata bins)/ln 2) in fp64,
//             applied to those numdata bins only; bins beyond keep raw values
//             exactly as the reference leaves them.  nph_norm > 0 replaces the
//             median by the -photon constant.
//   forward : NF-point FFT.  The reference zero-interleaves to fftlen points
//             first; that transform is this one repeated twice, X'[k] = X[k mod NF].
// blk_nice[b] <= NF is the window length the reference reads
// (next2_to_n(numdata)); a null pointer means NF.  It is shorter than NF only
// for sub-harmonics whose reference transform is shorter than the shortest plan
// here: those run zero-padded in a longer one.
// The median is the exact order statistic (numdata-1)/2 (lower median), found
// by a bitwise radix select on the IEEE bit patterns of the non-negative powers.
// ---------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(P::NT) void k_block_prep(
    const c32 *__restrict__ fft, int64_t numbins, int64_t file_lobin,
    const int64_t *__restrict__ blk_lobin, const int *__restrict__ blk_numdata,
    c32 *__restrict__ blockfft, double nph_norm, int locpow, const int *__restrict__ blk_nice,
    const c32 *__restrict__ tw1, const c32 *__restrict__ tw2, const int2 *__restrict__ zrange = nullptr)
{
    if (zrange && zrange[blockIdx.x].y <= zrange[blockIdx.x].x)
        return; // a series split over several GPUs: no row of this block is wanted here
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    // wave partial counts live behind the FFT image (launch with LDS_BYTES + 64)
    int (*s_cnt)[P::NT / 64] = reinterpret_cast<int (*)[P::NT / 64]>(smem + P::LDS_BYTES);
    static_assert(P::B1 == 1, "block prep expects one pass-1 butterfly per thread");
    static_assert(2 * (P::NT / 64) * 4 <= 64, "count scratch does not fit");

    const int t = threadIdx.x;
    const int blk = blockIdx.x;
    const int64_t lobin = blk_lobin[blk];
    const int numdata = blk_numdata[blk];
    const int nice_n = blk_nice ? blk_nice[blk] : P::N;

    c32 d[P::R1];
    uint32_t pw[P::R1];
#pragma unroll
    for (int a = 0; a < P::R1; a++) {
        const int idx = P::M1 * a + t;
        const int64_t fbin = lobin + idx - file_lobin;
        c32 x = c32{0.f, 0.f};
        if (fbin >= 0 && fbin < numbins && idx < nice_n)
            x = fft[fbin];
        d[a] = x;
        // r*r + i*i with each operation rounded on its own (no fma), so the
        // selected median is bit-identical to a plain C evaluation
        const float p = afft::strict::add(afft::strict::mul(x.x, x.x), afft::strict::mul(x.y, x.y));
        pw[a] = (idx < numdata) ? __float_as_uint(p) : 0xFFFFFFFFu;
    }

    double norm = 0.0;
    if (nph_norm > 0.0) {
        norm = nph_norm;
    } else if (locpow) {
        // -locpow: divide each bin by the root of the mean power in the ring of
        // 10 + 10 bins at distance 6..15 (corr_loc_pow, reference
        // src/corr_routines.c:309-366; the ends copy their nearest full value),
        // through the same bit-trick inverse square root (src/misc_utils.c:255-266).
        // Sums run in one fixed order and without fused ops, so the factors are
        // reproducible bit for bit.
        float *pl = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int a = 0; a < P::R1; a++) {
            const int idx = P::M1 * a + t;
            pl[idx] = afft::strict::add(afft::strict::mul(d[a].x, d[a].x), afft::strict::mul(d[a].y, d[a].y));
        }
        __syncthreads();
        const int n = nice_n, khw = 15;
        const float scale = afft::strict::mul(2.0f / (float) n, (float) ((n / 2) / 20));
#pragma unroll
        for (int a = 0; a < P::R1; a++) {
            const int idx = P::M1 * a + t;
            if (idx < numdata) {
                const int c = min(max(idx, khw), n - 1 - khw);
                float acc = 0.0f;
                for (int k = -15; k <= -6; k++) acc = afft::strict::add(acc, pl[c + k]);
                for (int k = 6; k <= 15; k++) acc = afft::strict::add(acc, pl[c + k]);
                const float x = afft::strict::mul(acc, scale);
                const float y = __int_as_float(0x5f3759df - (__float_as_int(x) >> 1));
                const float nrm = afft::strict::mul(y, afft::strict::sub(1.5f, afft::strict::mul(afft::strict::mul(afft::strict::mul(0.5f, x), y), y)));
                d[a].x = afft::strict::mul(d[a].x, nrm);
                d[a].y = afft::strict::mul(d[a].y, nrm);
            }
        }
        __syncthreads(); // the FFT reuses this LDS
    } else {
        const int k = (numdata - 1) / 2;
        uint32_t prefix = 0;
        int slot = 0;
        for (int bit = 30; bit >= 0; bit--) {
            const uint32_t cand = prefix | (1u << bit);
            int c = 0;
#pragma unroll
            for (int a = 0; a < P::R1; a++)
                c += (pw[a] < cand) ? 1 : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                c += __shfl_xor(c, off);
            if ((t & 63) == 0)
                s_cnt[slot][t >> 6] = c;
            __syncthreads();
            int total = 0;
#pragma unroll
            for (int w = 0; w < P::NT / 64; w++)
                total += s_cnt[slot][w];
            if (total <= k)
                prefix = cand;
            slot ^= 1; // the other slot is free: everyone passed the barrier after reading it
        }
        const double med = (double) __uint_as_float(prefix);
        norm = 1.0 / sqrt(med / 0.69314718055994530942);
    }
    if (!locpow || nph_norm > 0.0) {
#pragma unroll
        for (int a = 0; a < P::R1; a++) {
            const int idx = P::M1 * a + t;
            if (idx < numdata) {
                d[a].x = (float) ((double) d[a].x * norm);
                d[a].y = (float) ((double) d[a].y * norm);
            }
        }
    }
    __syncthreads();
    // The spectrum is stored times 1/(2*P::N), one over the length of the correlation it
    // feeds: a power of two, so every later product and sum is the unscaled one times that
    // power exactly, and the correlate kernels' |x|^2 needs no 1/N^2 any more.
    c32 *out = blockfft + (size_t) blk * P::N;
    constexpr float kScale = 1.0f / (2.0f * (float) P::N);
    afft::fft3<P, -1>(
        lds, tw1, tw2, [&](int, int a, int) { return d[a]; }, [&](int n, c32 v) { out[n] = v * kScale; });
}

// |v|^2 as every plane-generation kernel forms it: x*x + (y*y rounded), one fused step -- spelled out,
// so that all variants of the kernels write the same bits whatever the compiler would contract.
__device__ __forceinline__ float power_of(c32 v) { return __builtin_fmaf(v.x, v.x, __fmul_rn(v.y, v.y)); }

// ---------------------------------------------------------------------------
// Stores of one transform's kept outputs into a plane row, through a buffer
// descriptor over the block's columns [out, out + ncols): the hardware range check
// drops the outputs that fall before the first or after the last column, so a store
// is one instruction with a 32-bit byte offset and no compare, mask or 64-bit address.
// A thread's outputs are n = t + c, c a compile-time multiple of NT; its byte offset
// is 4*(n - shift).  Outputs with c < 2048 can lie before the block (shift <= 2048 is
// checked on the host): each of those has its own offset register, so a negative offset
// reaches the range check as a large unsigned number; the others share one register per
// 4096 bytes (never negative) and carry the rest as the instruction's immediate.
// ---------------------------------------------------------------------------
// Outputs n >= 2048 + kUseLen lie behind every block's columns (the shift is at most 2048, a block has at most
// kUseLen columns): of a 16384-point transform that is 13 of a thread's 32 outputs, which callers skip when the
// kernel is compiled (KEPT) -- their last-pass arithmetic, powers and stores with them.
template <class P, int ES = 4> struct PlaneStore { // ES: bytes per stored element
    static constexpr int KEPT = P::N < 2048 + kUseLen ? P::N : 2048 + kUseLen; // outputs at or past this are never stored
    static constexpr int NEAR = (P::N < 2048 ? P::N : 2048) / P::NT, NG = KEPT >= 1024 ? (KEPT + 1023) / 1024 : 1;
    int vnear[NEAR], vgrp[NG];
    __device__ __forceinline__ PlaneStore(int t, int shift)
    {
        const int voff = (t - shift) * ES;
#pragma unroll
        for (int i = 0; i < NEAR; i++) {
            vnear[i] = voff + ES * P::NT * i;
            asm("" : "+v"(vnear[i])); // opaque: the sum must not be split into base + immediate again
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            vgrp[g] = voff + ES * 1024 * g;
            asm("" : "+v"(vgrp[g]));
        }
    }
    // Called once per block: the registers become opaque again, so the constant parts
    // added below stay instruction immediates instead of being hoisted out of the block
    // loop into one register per output.
    __device__ __forceinline__ void refresh()
    {
#pragma unroll
        for (int g = 0; g < NG; g++)
            asm volatile("" : "+v"(vgrp[g]));
    }
    __device__ __forceinline__ void operator()(__amdgpu_buffer_rsrc_t rsrc, int c, float p) const
    {
        static_assert(ES == 4, "float elements");
        if (c < 2048)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(p), rsrc, vnear[c / P::NT], 0, 0);
        else
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(p), rsrc, vgrp[c / 1024] + 4 * (c % 1024), 0, 0);
    }
    // the power as two 16-bit halves, into the planes of upper and of lower halves (see k_harmsum_b)
    __device__ __forceinline__ void split(__amdgpu_buffer_rsrc_t hi, __amdgpu_buffer_rsrc_t lo, int c, float p) const
    {
        static_assert(ES == 2, "16-bit elements");
        const unsigned int b = __float_as_uint(p);
        if (c < 2048) {
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short) (b >> 16), hi, vnear[c / P::NT], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short) b, lo, vnear[c / P::NT], 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short) (b >> 16), hi, vgrp[c / 1024] + 2 * (c % 1024), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short) b, lo, vgrp[c / 1024] + 2 * (c % 1024), 0, 0);
        }
    }
};

// The same for a thread that owns output pairs (n, n + 1), n = 2t + STEP*f (the paired last pass
// of afft::fft3x): a pair of powers is one 8-byte store into the float plane, or -- split planes --
// one 4-byte store of the two upper halves and one of the two lower halves.
// Columns are even (n - shift with both even) and a block's first column and width are even,
// so a pair is kept or dropped as a whole and every access is aligned to its size.
// Offsets that can be negative (n < 2048) again have a register of their own.
template <class P> struct PairStore {
    static constexpr int STEP = P::R1 * P::R2, NF = P::R3;
    static constexpr int NEARF = 2048 / STEP < NF ? 2048 / STEP : NF;   // outputs below 2048: exact offsets
    static constexpr int G4 = 4096 / (4 * STEP) > 0 ? 4096 / (4 * STEP) : 1; // f's per offset register, powers
    static constexpr int G2 = 4096 / (2 * STEP) > 0 ? 4096 / (2 * STEP) : 1; // f's per offset register, 16-bit planes
    static constexpr int NG4 = (NF - NEARF + G4 - 1) / G4, NG2 = (NF - NEARF + G2 - 1) / G2;
    int near4[NEARF], grp4[NG4 > 0 ? NG4 : 1];
    int near2[NEARF], grp2[NG2 > 0 ? NG2 : 1];
    int col0;  // column of the thread's first pair
    int gslot; // the first pair's group of four (n >> 2) among the transform's
    __device__ __forceinline__ PairStore(int t, int shift, bool halves, bool maxima = false)
    {
        col0 = 2 * t - shift;
        gslot = maxima ? t >> 1 : 0;
        const int c0 = 2 * t - shift; // column of the thread's first pair
#pragma unroll
        for (int f = 0; f < NEARF; f++) {
            near4[f] = 4 * (c0 + STEP * f);
            asm("" : "+v"(near4[f])); // opaque: the sum must not be split into base + immediate again
            near2[f] = halves ? 2 * (c0 + STEP * f) : 0;
            asm("" : "+v"(near2[f]));
        }
#pragma unroll
        for (int g = 0; g < NG4; g++) {
            grp4[g] = 4 * (c0 + STEP * (NEARF + G4 * g));
            asm("" : "+v"(grp4[g]));
        }
#pragma unroll
        for (int g = 0; g < NG2; g++) {
            grp2[g] = halves ? 2 * (c0 + STEP * (NEARF + G2 * g)) : 0;
            asm("" : "+v"(grp2[g]));
        }
    }
    // The largest upper half of the four outputs n >> 2 == const that this lane and its neighbour
    // (lane ^ 1) hold, into the workgroup's table of the transform's group maxima in LDS (both lanes write the
    // same value to the same place); outputs outside the block's columns do not count.  What goes to the plane
    // of maxima is formed from this table once the whole transform has written it: see window_maxima().
    template <int f> __device__ __forceinline__ void maxima(unsigned short *gmx, float p0, float p1, int ncols) const
    {
        float m = fmaxf(p0, p1);
        if (f < NEARF || STEP * (f + 1) > kUseLen) // pairs that can lie before or behind the block's columns
            m = (unsigned int) (col0 + STEP * f) < (unsigned int) ncols ? m : 0.f;
        // max(m, m of lane ^ 1) as one instruction.  (Through fmaxf and a DPP move the compiler spends three: the
        // move, a canonicalising max of the moved value with itself, the max.  The wait states a DPP read of a
        // register just written needs are spelled out, since the hazard check does not look inside.)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(m) : "v"(m));
        gmx[gslot + (STEP / 4) * f] = (unsigned short) (__float_as_uint(m) >> 16);
    }
    __device__ __forceinline__ void refresh(bool halves)
    {
        if (halves) {
#pragma unroll
            for (int g = 0; g < NG2; g++)
                asm volatile("" : "+v"(grp2[g]));
        } else {
#pragma unroll
            for (int g = 0; g < NG4; g++)
                asm volatile("" : "+v"(grp4[g]));
        }
    }
    // f is the compile-time index of the pair (its outputs are 2t + STEP*f and the next one)
    __device__ __forceinline__ void powers(__amdgpu_buffer_rsrc_t rsrc, int f, float p0, float p1) const
    {
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const u2 d = {__float_as_uint(p0), __float_as_uint(p1)};
        if (f < NEARF)
            __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, near4[f], 0, 0);
        else
            __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, grp4[(f - NEARF) / G4] + 4 * STEP * ((f - NEARF) % G4), 0, 0);
    }
    __device__ __forceinline__ void split(__amdgpu_buffer_rsrc_t hi, __amdgpu_buffer_rsrc_t lo, int f, float p0, float p1) const
    {
        // the two upper halves side by side, and the two lower halves
        const unsigned int h = __builtin_amdgcn_perm(__float_as_uint(p1), __float_as_uint(p0), 0x07060302u);
        const unsigned int l = __builtin_amdgcn_perm(__float_as_uint(p1), __float_as_uint(p0), 0x05040100u);
        if (f < NEARF) {
            __builtin_amdgcn_raw_buffer_store_b32(h, hi, near2[f], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(l, lo, near2[f], 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(h, hi, grp2[(f - NEARF) / G2] + 2 * STEP * ((f - NEARF) % G2), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(l, lo, grp2[(f - NEARF) / G2] + 2 * STEP * ((f - NEARF) % G2), 0, 0);
        }
    }
};

// An entry of the plane of maxima for the largest upper half h of its columns: h + 1, which as the upper half of a
// float is greater than every power with upper half <= h (the bound the sums work with: see bound_value), so that
// readers need no arithmetic; 0xffff (a NaN either way) stays.
__device__ __forceinline__ unsigned short bound_entry(unsigned int h) { return (unsigned short) (h + (h < 0xffffu ? 1u : 0u)); }

// The plane of maxima holds, per entry, the largest upper half of a WINDOW of two consecutive groups of four
// outputs: entry j of a block stands for the groups j and j + 1 (up to eight columns), so that any run of up to
// five consecutive columns is bounded by ONE entry -- the one its first column's group has (see the coarse bound
// in front of the harmonic sum).  A thread holds the maxima of its own groups only, and a group's neighbour may
// sit in another wave, so the transform's group maxima go through a table in LDS (4 KB, behind the FFT image;
// two entries of padding, kept zero): written during the last pass, read here after the barrier that ends the
// transform -- thread t forms the windows of the groups 2t, 2t + 1 (or 2t + 1, 2t + 2), the same 2 * NT on, ...
// and stores them, consecutive threads to consecutive pairs of entries.  `s4` = shift >> 2: plane entry j is the transform's group j + s4; entries before the
// block's first group and after its last are dropped by the descriptor's range check.  The block's last
// windows (they reach into the next block) are completed by k_group_fix.
// (two entries at a time: the table read as 32-bit words, the two maxima and the two increments each one packed
// instruction, one 4-byte store; which pairs are aligned in the plane depends on the parity of s4)
template <class P, int J, bool ODD>
__device__ __forceinline__ void window_maxima_one(const uint32_t *gmx32, __amdgpu_buffer_rsrc_t rs, int t, int o0, int o1)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    constexpr int NP = P::N / 8; // pairs of groups
    // (a block has kGpb entries and starts at most 512 groups into the transform: pairs beyond that are never stored)
    constexpr int NJ = NP / P::NT < ((kGpb + 512 + 2) / 2 + P::NT - 1) / P::NT ? NP / P::NT : ((kGpb + 512 + 2) / 2 + P::NT - 1) / P::NT;
    if constexpr (J < NJ) {
        const int pidx = t + P::NT * J;
        const uint32_t d0 = gmx32[pidx], d1 = gmx32[pidx + 1];   // groups 2p, 2p + 1 | 2p + 2, 2p + 3
        const uint32_t mid = __builtin_amdgcn_alignbit(d1, d0, 16); // groups 2p + 1, 2p + 2
        // windows 2p, 2p + 1 (even s4) or 2p + 1, 2p + 2 (odd)
        us2 w = __builtin_elementwise_max(__builtin_bit_cast(us2, ODD ? mid : d0), __builtin_bit_cast(us2, ODD ? d1 : mid));
        w = __builtin_elementwise_add_sat(w, us2{1, 1}); // (bound_entry of both)
        if constexpr (J == 0)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, w), rs, o0, 0, 0);
        else
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, w), rs, o1 + 4 * P::NT * (J - 1), 0, 0); // (the constant becomes the instruction's offset field)
        window_maxima_one<P, J + 1, ODD>(gmx32, rs, t, o0, o1);
    }
}
template <class P> __device__ __forceinline__ void window_maxima(const unsigned short *gmx, unsigned short *out_mx, int t, int s4)
{
    static_assert((P::N / 8) % P::NT == 0, "pairs of groups divide evenly among the threads");
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out_mx, 0, kGpb * 2, 0x00020000);
    const uint32_t *gmx32 = reinterpret_cast<const uint32_t *>(gmx);
    // the first store's offset can be negative (groups before the block's first column), the others cannot:
    // a register each, so that a negative offset reaches the range check as a large unsigned number
    const int odd = s4 & 1;
    int o0 = 2 * (2 * t + odd - s4), o1 = 2 * (2 * (t + P::NT) + odd - s4);
    asm("" : "+v"(o0), "+v"(o1));
    if (odd)
        window_maxima_one<P, 0, true>(gmx32, rs, t, o0, o1);
    else
        window_maxima_one<P, 0, false>(gmx32, rs, t, o0, o1);
}
constexpr int kMaximaLdsBytes = 2 * (8192 / 4 + 2); // the table above for the 8192-point plan
constexpr int kMaximaLdsBytes16k = 2 * (16384 / 4 + 2); // ... and for the 16384-point plan

// The same table for a plan whose last pass hands a thread single outputs n = t + c (c a compile-time multiple of the
// workgroup's size; the 16384-point plan): the four outputs of a group sit in four neighbouring lanes, two DPP
// steps bring their largest to all four, and all four write it (the same value to the same place).  Outputs
// outside the block's columns do not count; outputs past anything a block can keep (c >= 2048 + kUseLen: the
// shift is at most 2048) are skipped when the kernel is compiled.
template <class P> __device__ __forceinline__ void single_maxima(unsigned short *gmx, int t, int c, float p, int col_t, int ncols)
{
    if (c >= 2048 + kUseLen)
        return;
    float m = p;
    if (c < 2048 || c + P::NT > kUseLen) // outputs that can lie before or behind the block's columns
        m = (unsigned int) (col_t + c) < (unsigned int) ncols ? m : 0.f;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(m) : "v"(m));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(m) : "v"(m));
    gmx[(t >> 2) + (c >> 2)] = (unsigned short) (__float_as_uint(m) >> 16);
}

// The bank as k_correlate_rr reads it: folded[z][k] = K_z[k] + K_z[k + n/2], folded[z][k + n/2] = K_z[k] - K_z[k + n/2]
// for k < n/2 (n = the transform length).  One thread per k < n/2 and row.
__global__ __launch_bounds__(256) void k_fold_bank(const c32 *__restrict__ bank, c32 *__restrict__ folded, int n)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n / 2)
        return;
    const size_t row = (size_t) blockIdx.y * (size_t) n;
    const c32 a = bank[row + k], b = bank[row + k + n / 2];
    folded[row + k] = a + b;
    folded[row + k + n / 2] = a - b;
}

// ---------------------------------------------------------------------------
// Correlation: one workgroup per (template z, chunk of blocks).  The template's
// spectrum stays in registers for the whole chunk.  For each block:
//   X[k] = D[k mod N/2] * conj(K_z[k]);  x = IFFT_N(X) (unnormalised);
//   plane[z][col_b + j] = |x[j + 2*halfwidth]|^2 / N^2,  0 <= j < ncols_b
// (the 1/N comes with D: k_block_prep stores the block spectra already scaled)
// ---------------------------------------------------------------------------
template <class P, bool KEEPK>
__global__ __launch_bounds__(P::NT, 2) void k_correlate(
    const c32 *__restrict__ bank, const c32 *__restrict__ blockfft,
    const int64_t *__restrict__ blk_col, const int *__restrict__ blk_ncols, int nblk, int chunk,
    float *__restrict__ plane, int64_t rlen, int halfwidth, const c32 *__restrict__ tw1,
    const c32 *__restrict__ tw2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    static_assert(P::B1 == 1, "correlate expects one pass-1 butterfly per thread");
    constexpr int NF = P::N / 2;
    constexpr int H = P::R1 / 2; // M1 * H == NF: inputs a and a + H are the same bin

    const int t = threadIdx.x;
    const int z = blockIdx.x;
    const int b0 = blockIdx.y * chunk;
    const int b1 = min(b0 + chunk, nblk);

    // KEEPK: the template's spectrum stays in registers for the whole chunk;
    // otherwise it is re-read (from L2) for every block to save 2*R1 registers
    const c32 *krow = bank + (size_t) z * P::N;
    c32 kreg[KEEPK ? P::R1 : 1];
    if constexpr (KEEPK) {
#pragma unroll
        for (int a = 0; a < P::R1; a++)
            kreg[a] = krow[P::M1 * a + t];
    }
    float *prow = plane + (size_t) z * (size_t) rlen;
    const int shift = 2 * halfwidth;
    PlaneStore<P> put(t, shift);

    for (int blk = b0; blk < b1; blk++) {
        // The tables are loop-invariant; left alone the compiler hoists every
        // load of them out of the block loop and runs out of registers.  Launder
        // the pointers so each block re-reads them (L1/L2 hits).
        const c32 *tw1_0 = tw1, *tw2_0 = tw2, *krow_0 = krow;
        asm volatile("" : "+s"(tw1_0), "+s"(tw2_0), "+s"(krow_0));
        // the laundering hides that these point to global memory: say so again, or the
        // loads become flat loads, which also count against the LDS wait counter
        typedef const c32 __attribute__((address_space(1))) *gptr;
        const gptr tw1_l = (gptr) tw1_0, tw2_l = (gptr) tw2_0, krow_l = (gptr) krow_0;
        const c32 *D = blockfft + (size_t) blk * NF;
        c32 dreg[H];
#pragma unroll
        for (int a = 0; a < H; a++)
            dreg[a] = D[P::M1 * a + t];
        float *out = prow + blk_col[blk];
        const int ncols = blk_ncols[blk];
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, ncols * 4, 0x00020000);
        put.refresh();
        afft::fft3<P, +1>(
            lds, tw1_l, tw2_l,
            [&](int, int a, int) {
                if constexpr (KEEPK)
                    return afft::cmul_conj_v(dreg[a % H], kreg[a]);
                else
                    return afft::cmul_conj_v(dreg[a % H], krow_l[P::M1 * a + t]);
            },
            [&](int n, c32 v) { put(rsrc, n - t, power_of(v)); });
        __syncthreads(); // pass-3 reads done before the next block's pass-1 writes
    }
}

// ---------------------------------------------------------------------------
// Correlation, register-resident form.  Same arithmetic as k_correlate, but every
// twiddle of passes 1 and 2 is expanded once per workgroup from two short exact tables
// and stays in registers for the whole chunk of transforms, and (ZL) so does the block
// spectrum: a workgroup owns one block and loops over a chunk of templates.  Global
// traffic per transform drops from ~160 KB to the 64 KB template row.  KEEPK (template
// row in registers too, for the block loop) does not fit the register file and is kept
// only as a compile-time experiment.
// `bank` is the FOLDED bank (k_fold_bank): inputs k and k + N/2 of the inverse transform are products of the
// same bin of the block spectrum (the zero-interleaved block's spectrum has period N/2), and the transform's
// innermost butterflies are exactly the pairs (k, k + N/2); so D * conj(K[k]) +- D * conj(K[k + N/2]) =
// D * conj(K[k] +- K[k + N/2]) is formed with the sum and the difference taken once per bank instead of once
// per transform (16 packed additions and 16 subtractions fewer per thread and transform).
//   pw1: [R1/4+2][M1]  rows 0..R1/4-2: exp(+2*pi*i*b*4*hi/N), hi = 1..R1/4-1;
//                      then three rows exp(+2*pi*i*b*lo/N), lo = 1..3
//   pw2: [R2/4+2][R3]  the same for exp(+2*pi*i*d*e/M1)
// ---------------------------------------------------------------------------
// SPLIT: the plane is two planes of 16-bit values, the upper and the lower halves of every float
// (`hi`, `lo`, each of the plane's shape; `plane` is not used): the harmonic sum reads the upper
// halves alone -- half the bytes -- and both only where it matters (see k_harmsum_b).
// PAIR (8192-point plan): the last pass hands a thread adjacent outputs, stored two at a time.
template <class P, bool KEEPK, int SER, bool ZL = false, class Stamp = afft::NoStamp, bool SPLIT = false, bool PAIR = false,
          bool MAXIMA = false>
__global__ __launch_bounds__(P::NT, 2) void k_correlate_rr(
    const c32 *__restrict__ bank, const c32 *__restrict__ blockfft,
    const int64_t *__restrict__ blk_col, const int *__restrict__ blk_ncols, int nblk, int chunk,
    float *__restrict__ plane, int64_t rlen, int halfwidth, const c32 *__restrict__ pw1,
    const c32 *__restrict__ pw2, int numz, int tile_z, int tile_c, Stamp stamp = Stamp(),
    unsigned short *__restrict__ hi = nullptr, unsigned short *__restrict__ lo = nullptr,
    unsigned short *__restrict__ mx = nullptr, int64_t mxlen = 0, const int2 *__restrict__ zrange = nullptr)
{
    static_assert(!MAXIMA || (SPLIT && ZL), "group maxima come with the template loop's stores into split planes");
    static_assert(!MAXIMA || PAIR || (P::B3 == 1 && P::NT % 4 == 0), "single stores: a group's four outputs in four neighbouring lanes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    static_assert(P::B1 == 1, "one pass-1 butterfly per thread");
    static_assert(P::NT % P::R3 == 0, "pass-2 butterflies of a thread must share d");
    constexpr int NF = P::N / 2;
    constexpr int H = P::R1 / 2;

    const int t = threadIdx.x;
    // ZL: a workgroup owns one block and a chunk of templates (the block's spectrum stays
    // in registers, the template row changes per transform); otherwise one template and a
    // chunk of blocks.  `ia` indexes the owned item, `cy` the chunk of the others.
    const int na = ZL ? nblk : numz, nitems = ZL ? numz : nblk;
    const int nchunks = (nitems + chunk - 1) / chunk;
    int ia, cy;
    if (tile_z > 0) {
        // One-dimensional grid, dealt round-robin over the 8 XCDs by the hardware: the
        // workgroups an XCD holds at one time come from one tile of tile_z owned items x
        // tile_c chunks, so its L2 keeps the few template rows and block spectra of that
        // tile instead of cycling through the whole bank.
        // The tiles beyond the last whole eight -- 281 tiles are 35 eights and one -- are dealt workgroup by
        // workgroup over all XCDs instead: as a whole tile each they would leave one XCD with a tile more than the
        // others (1/36 of the kernel with seven eighths of the device idle).
        const int tiles_a = (na + tile_z - 1) / tile_z, tiles_c = (nchunks + tile_c - 1) / tile_c;
        const int ntiles = tiles_a * tiles_c, eights = ntiles / 8;
        const int per = tile_z * tile_c;
        const int x = blockIdx.x % 8, sq = blockIdx.x / 8;
        int tile, p;
        if (sq / per < eights) {
            tile = (sq / per) * 8 + x;
            p = sq % per;
        } else {
            const int j = (sq % per) * 8 + x;
            tile = eights * 8 + j / per;
            p = j % per;
            if (tile >= ntiles)
                return;
        }
        ia = (tile % tiles_a) * tile_z + p % tile_z;
        cy = (tile / tiles_a) * tile_c + p / tile_z;
        if (ia >= na || cy >= nchunks)
            return;
    } else {
        ia = blockIdx.x;
        cy = blockIdx.y;
    }
    int b0 = cy * chunk;
    int b1 = min(b0 + chunk, nitems);
    if constexpr (ZL) {
        // a series split over several GPUs: of a block that only serves as a source of sub-harmonic look-ups,
        // only the rows those can reach are wanted (zrange[block] = [first row, one past the last))
        if (zrange) {
            const int2 zr = zrange[ia];
            b0 = max(b0, zr.x);
            b1 = min(b1, zr.y);
        }
    }
    if (b0 >= b1)
        return;
    const int z = ZL ? b0 : ia; // ZL: the first template of the chunk

    const c32 *krow = bank + (size_t) z * P::N;
    c32 kreg[KEEPK ? P::R1 : 1];
    afft::TwoLevel<P::R1> w1;
    afft::TwoLevel<P::R2> w2;
    {
        if constexpr (KEEPK) {
#pragma unroll
            for (int a = 0; a < P::R1; a++)
                kreg[a] = krow[P::M1 * a + t];
        }
#pragma unroll
        for (int k = 0; k < w1.NA; k++)
            w1.A[k] = pw1[k * P::M1 + t];
#pragma unroll
        for (int k = 0; k < 3; k++)
            w1.B[k] = pw1[(w1.NA + k) * P::M1 + t];
#pragma unroll
        for (int k = 0; k < w2.NA; k++)
            w2.A[k] = pw2[k * P::R3 + (t % P::R3)];
#pragma unroll
        for (int k = 0; k < 3; k++)
            w2.B[k] = pw2[(w2.NA + k) * P::R3 + (t % P::R3)];
    }
    const int shift = 2 * halfwidth;
    PlaneStore<P> put(t, PAIR || SPLIT ? 0 : shift);
    PlaneStore<P, 2> put16(t, SPLIT && !PAIR ? shift : 0);
    PairStore<P> put2(t, PAIR ? shift : 0, SPLIT, MAXIMA);
    // every twiddle of passes 1 and 2, formed once and kept in registers for the whole chunk
    c32 W1[P::R1 - 1], W2[P::R2 - 1];
    afft::expand_twiddles<P::R1>(W1, w1);
    afft::expand_twiddles<P::R2>(W2, w2);
    c32 dreg[H];
    {
        const c32 *D = blockfft + (size_t) (ZL ? ia : b0) * NF;
#pragma unroll
        for (int a = 0; a < H; a++)
            dreg[a] = D[P::M1 * a + t];
    }
    // MAXIMA: the transform's group maxima in LDS (see window_maxima), and where the previous transform's go
    unsigned short *gmx = reinterpret_cast<unsigned short *>(smem + P::LDS_BYTES);
    unsigned short *mx_prev = nullptr;
    if constexpr (MAXIMA) {
        if (t < 2)
            gmx[P::N / 4 + t] = 0; // the padding entries (visible after the first transform's barriers)
    }

    for (int it = b0; it < b1; it++) {
        stamp.begin(it - b0);
        stamp(0);
        const int blk = ZL ? ia : it;
        const size_t cell0 = (size_t) (ZL ? it : z) * (size_t) rlen + (size_t) blk_col[blk]; // the block's first cell in row z
        const int ncols = blk_ncols[blk];
        float *out = SPLIT ? nullptr : plane + cell0;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, SPLIT ? 0 : ncols * 4, 0x00020000);
        unsigned short *out_hi = SPLIT ? hi + cell0 : nullptr, *out_lo = SPLIT ? lo + cell0 : nullptr;
        const __amdgpu_buffer_rsrc_t rsrc_hi = __builtin_amdgcn_make_buffer_rsrc(out_hi, 0, SPLIT ? ncols * 2 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_lo = __builtin_amdgcn_make_buffer_rsrc(out_lo, 0, SPLIT ? ncols * 2 : 0, 0x00020000);
        // MAXIMA: the block's run of the plane of group maxima in row z (the launch covers the whole block grid:
        // a block's index is its place in that plane)
        unsigned short *out_mx = MAXIMA ? mx + (size_t) (ZL ? it : z) * (size_t) mxlen + (size_t) blk * kGpb : nullptr;
        if constexpr (PAIR)
            put2.refresh(SPLIT);
        else if constexpr (SPLIT)
            put16.refresh();
        else
            put.refresh();
        // launder the pointer: the loads must stay inside the loop (hoisted, they
        // would pin another 2*R1 registers across the whole transform)
        const c32 *krow_l = ZL ? bank + (size_t) it * P::N : krow;
        asm volatile("" : "+s"(krow_l));
        // The template's spectrum: all loads issued before the first product (buffer
        // loads with the row offset in the scalar operand: no address arithmetic); the
        // products start as the values arrive, in order.  (Typed as plain global loads the
        // compiler places them one by one right before their use, each with a full wait:
        // 5.6 ms against 3.3 ms.)
        c32 kv[KEEPK ? 1 : P::R1];
        if constexpr (!KEEPK) {
            const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void *) krow_l, 0, P::N * 8, 0x00020000);
#pragma unroll
            for (int a = 0; a < P::R1; a++) {
                typedef unsigned int u2 __attribute__((ext_vector_type(2)));
                const u2 r = __builtin_amdgcn_raw_buffer_load_b64(krs, t * 8, P::M1 * 8 * a, 0);
                kv[a] = c32{__uint_as_float(r.x), __uint_as_float(r.y)};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        auto load = [&](int, int a, int) {
            if constexpr (KEEPK)
                return afft::cmul_conj_v(dreg[a % H], kreg[a]);
            else
                return afft::cmul_conj_v(dreg[a % H], kv[a]);
        };
        auto tw1 = [&](int, int, c32 *v) { afft::apply_twiddles_full<P::R1>(v, W1); };
        auto tw2 = [&](int, int, c32 *v) { afft::apply_twiddles_full<P::R2>(v, W2); };
        auto mid = [&] {
            // the block spectrum is used up by pass 1: fetch the next block's into the
            // same registers now, so that it arrives during passes 2 and 3
            if (!ZL && it + 1 < b1) {
                const c32 *D = blockfft + (size_t) (it + 1) * NF;
#pragma unroll
                for (int a = 0; a < H; a++)
                    dreg[a] = D[P::M1 * a + t];
            }
            // the previous transform's maxima: its table is complete (the barrier that ended it) and is not
            // written again before this transform's last pass, three barriers on
            if constexpr (MAXIMA) {
                if (mx_prev)
                    window_maxima<P>(gmx, mx_prev, t, shift >> 2);
            }
        };
        if constexpr (PAIR) {
            afft::fft3x<P, +1, SER, true, true>(
                lds, load, tw1, tw2, mid,
                [&](auto fc, c32 v, c32 w) {
                    constexpr int f = decltype(fc)::value;
                    const float p0 = power_of(v), p1 = power_of(w);
                    if constexpr (SPLIT)
                        put2.split(rsrc_hi, rsrc_lo, f, p0, p1);
                    else
                        put2.powers(rsrc, f, p0, p1);
                    if constexpr (MAXIMA)
                        put2.template maxima<f>(gmx, p0, p1, ncols);
                },
                stamp);
        } else {
            afft::fft3x<P, +1, SER, false, true>(
                lds, load, tw1, tw2, mid,
                [&](auto uc, auto fc, c32 v) { // output n = t + c
                    constexpr int c = P::NT * decltype(uc)::value + P::R1 * P::R2 * decltype(fc)::value;
                    if constexpr (c < PlaneStore<P>::KEPT) {
                        const float pw = power_of(v);
                        if constexpr (SPLIT)
                            put16.split(rsrc_hi, rsrc_lo, c, pw);
                        else
                            put(rsrc, c, pw);
                        if constexpr (MAXIMA)
                            single_maxima<P>(gmx, t, c, pw, t - shift, ncols);
                    }
                },
                stamp);
        }
        __syncthreads();
        stamp(9);
        mx_prev = out_mx;
    }
    if constexpr (MAXIMA) {
        if (mx_prev)
            window_maxima<P>(gmx, mx_prev, t, shift >> 2);
    }
}

// ---------------------------------------------------------------------------
// Harmonic sum + threshold.  A workgroup owns 512*J consecutive fundamental
// columns (absolute half-bin index rind = col0 + c) and walks every z row; a
// thread owns J column pairs (2t, 2t+1) + 512*j, so the fundamental comes in
// as 8-byte loads.  For stage s >= 1 and odd h < 2^s the sub-harmonic cell is
//   row  rowoff[zi][harm]  = 4 * rlen * index_from_z(calc_required_z(h/2^s, z))
//   col  (int)(rind * h / 2^s + 0.5) = (rind*h + 2^(s-1)) >> s   (exact: dyadic)
// and the running float sum is accumulated in the reference's order
// (fundamental, 1/2, 1/4, 3/4, 1/8, ...).  After each stage the sum is compared
// with powcut[stage]; crossings are compacted per wavefront with a ballot and
// one atomic per wave.  Row offsets are wave-uniform (scalar loads, scalar base
// + 32-bit lane offset addressing), so the loop is loads and adds only.
// DUMP: instead of thresholding, write the sums after stage `dump_stage`.
// ---------------------------------------------------------------------------
struct HarmParams {
    float powcut[6];
};

template <int NSTAGES> struct HarmShape {
    static constexpr int NH = (1 << (NSTAGES - 1)) - 1; // sub-harmonics summed in total
    static constexpr int NHS = NH > 0 ? NH : 1;         // row stride of the offset table
};

// C = columns per thread: 1 (dword loads, least registers, most waves in
// flight) or an even number 2*J (8-byte fundamental loads, J pairs 512 apart).
// Workgroup -> tile mapping: blockIdx.x is dealt round-robin over the 8 XCDs, so
// tile = (b % 8) * ceil(T/8) + b / 8 hands each XCD one contiguous run of tiles:
// neighbouring tiles then share source lines in that XCD's L2 and each XCD
// streams long contiguous pieces of every plane row.
template <int NSTAGES, int C, bool DUMP>
__global__ __launch_bounds__(256) void k_harmsum(
    const float *__restrict__ plane, int64_t rlen, int numz, int64_t col0, int64_t ncols,
    const int64_t *__restrict__ rowoff, const uint32_t *__restrict__ rowchg, HarmParams prm,
    DevHit *__restrict__ hits, unsigned int *__restrict__ count, unsigned int max_hits,
    float *__restrict__ dump, int dump_stage, int xcd_remap)
{
    constexpr int NH = HarmShape<NSTAGES>::NH, NHS = HarmShape<NSTAGES>::NHS;
    constexpr int J = C > 1 ? C / 2 : 1;
    constexpr int TILE = 256 * C;
    static_assert(C == 1 || C % 2 == 0, "C is 1 or even");
    const int lane = threadIdx.x & 63;
    unsigned int tile = blockIdx.x;
    if (xcd_remap) {
        const unsigned int per = (gridDim.x + 7) / 8;
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if ((int64_t) tile * TILE >= ncols)
            return; // padding slot of the remapped grid (whole workgroup)
    }
    const int64_t cbase = (int64_t) tile * TILE + (C > 1 ? 2 : 1) * threadIdx.x;

    int64_t rind[C];
    bool live[C];
    uint32_t off[NHS][C]; // byte offset of each source column inside its row
#pragma unroll
    for (int j = 0; j < J; j++)
#pragma unroll
        for (int k = 0; k < (C > 1 ? 2 : 1); k++) {
            const int64_t c = cbase + 512 * j + k;
            live[2 * j + k] = c < ncols;
            rind[2 * j + k] = col0 + (c < ncols ? c : (int64_t) k); // dead lanes read a valid cell
        }
    {
        int i = 0;
#pragma unroll
        for (int s = 1; s < NSTAGES; s++)
#pragma unroll
            for (int h = 1; h < (1 << s); h += 2) {
#pragma unroll
                for (int c = 0; c < C; c++)
                    off[i][c] = (uint32_t) (((rind[c] * h + (1 << (s - 1))) >> s) << 2);
                i++;
            }
    }

    const char *pbytes = reinterpret_cast<const char *>(plane);
    // Sub-harmonic values live in registers across z rows: harmonic h/2^s maps
    // about 2^s/h consecutive rows to one source row, so a value is re-read only
    // when rowchg[zi] says its source row moved (wave-uniform branch).  Every
    // plane cell is then fetched once per use instead of relying on L2 retention.
    float v[NHS][C];
    for (int zi = 0; zi < numz; zi++) {
        const float *frow = plane + (size_t) zi * (size_t) rlen;
        float p[C];
        if constexpr (C == 1) {
            p[0] = frow[rind[0]];
        } else {
#pragma unroll
            for (int j = 0; j < J; j++) {
                const float2 f = *reinterpret_cast<const float2 *>(frow + rind[2 * j]);
                p[2 * j] = f.x;
                p[2 * j + 1] = f.y;
            }
        }
        const uint32_t chg = NH > 0 ? rowchg[zi] : 0u;
        // the row's offsets all at once (wide scalar loads, one latency) rather than one
        // scalar load and wait inside each taken branch
        int64_t ro[NHS];
#pragma unroll
        for (int i = 0; i < NH; i++)
            ro[i] = rowoff[(size_t) zi * NHS + i];
#pragma unroll
        for (int i = 0; i < NH; i++) {
            if (chg & (1u << i)) {
                const char *row = pbytes + ro[i];
#pragma unroll
                for (int c = 0; c < C; c++)
                    v[i][c] = *reinterpret_cast<const float *>(row + off[i][c]);
            }
        }
        float ps[NSTAGES][C];
        bool any = false;
        {
            int i = 0;
#pragma unroll
            for (int s = 0; s < NSTAGES; s++) {
                if (s > 0) {
#pragma unroll
                    for (int h = 1; h < (1 << s); h += 2) {
#pragma unroll
                        for (int c = 0; c < C; c++)
                            p[c] += v[i][c];
                        i++;
                    }
                }
#pragma unroll
                for (int c = 0; c < C; c++) {
                    ps[s][c] = p[c];
                    any |= live[c] && (p[c] > prm.powcut[s]);
                }
            }
        }
        if constexpr (DUMP) {
#pragma unroll
            for (int s = 0; s < NSTAGES; s++)
#pragma unroll
                for (int c = 0; c < C; c++)
                    if (s == dump_stage && live[c])
                        dump[(size_t) zi * ncols + (rind[c] - col0)] = ps[s][c];
        } else {
            if (__ballot(any)) { // rare: some lane of this wave crossed a threshold
#pragma unroll
                for (int s = 0; s < NSTAGES; s++)
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const bool hit = live[c] && (ps[s][c] > prm.powcut[s]);
                        const unsigned long long mask = __ballot(hit);
                        if (mask) {
                            unsigned int base = 0;
                            const int leader = __ffsll((long long) mask) - 1;
                            if (lane == leader)
                                base = atomicAdd(count, (unsigned int) __popcll(mask));
                            base = __shfl(base, leader);
                            if (hit) {
                                const unsigned int pos =
                                    base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
                                if (pos < max_hits)
                                    hits[pos] = DevHit{ps[s][c], ((uint32_t) s << 16) | (uint32_t) zi,
                                                       rind[c]};
                            }
                        }
                    }
            }
        }
    }
}


// ---------------------------------------------------------------------------
// Harmonic sum over 16-bit upper bounds, exact only where it can matter.
//
// The plane is kept as two planes of the same shape: `hi` holds the upper 16 bits of every
// power's float, `lo` the lower 16 (plane generation writes both: k_correlate_rr<.., SPLIT>).
// For a power p >= 0 with upper half h, (h + 1) << 16 read as a float is greater than p: an
// upper bound in bfloat16 precision, from half the bytes.  The kernel walks the columns exactly
// as k_harmsum does, but reads `hi` alone and sums those bounds.  Floating-point addition is
// monotonic (a <= a', b <= b' => fl(a+b) <= fl(a'+b')), so the sums of the bounds, taken in
// the same order, bound the exact sums of every stage from above: a row of a wave in which no
// bound sum crosses its threshold has no crossing.  Where one does (a hit, or a cell within
// 2^-7 of its threshold) -- or is not <= the threshold at all: an infinite power's bound is
// a NaN -- the wave recomputes that row from both halves and compacts the exact crossings:
// the hits are those of k_harmsum on the float plane, bit for bit.
// Negative powers do not come out of plane generation; a host can write them into the plane
// (k_split_span sets flags[0] when it sees one), and then every row takes the exact path.
// ---------------------------------------------------------------------------
// The control words of the coarse bound in front of these sums (k_prefilter_*): word kPfFlag set = its list is of
// no use and the sum over the whole plane has to run.
constexpr int kPfCtlWords = 48;
constexpr int kPfFlag = 16;
constexpr int harm_stage_of(int i) // i-th sub-harmonic in summing order -> its stage s
{
    int s = 1;
    while ((1 << s) - 1 <= i)
        s++;
    return s;
}
constexpr int harm_num_of(int i) // -> its odd numerator h
{
    const int s = harm_stage_of(i);
    return 2 * (i + 1 - (1 << (s - 1))) + 1;
}
__device__ __forceinline__ float bound_value(uint32_t h) { return __uint_as_float((h + 1u) << 16); }
__device__ __forceinline__ float join_halves(uint32_t h, uint32_t l) { return __uint_as_float((h << 16) | l); }

// Floats <-> the two planes of halves, for the entry points that hand plane cells to or take
// them from a host: `n` consecutive cells of `rows` rows, row strides src_stride / rlen.
__global__ __launch_bounds__(256) void k_split_span(const float *__restrict__ src, int64_t src_stride,
                                                    unsigned short *__restrict__ hi, unsigned short *__restrict__ lo,
                                                    int64_t rlen, int64_t n, unsigned int *__restrict__ flags)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (i >= n)
        return;
    const float p = src[r * src_stride + i];
    const uint32_t b = __float_as_uint(p);
    hi[r * rlen + i] = (unsigned short) (b >> 16);
    lo[r * rlen + i] = (unsigned short) b;
    if (b >> 31)       // (the sign bit, so that -0.0 counts: as a 16-bit number its upper half beats every positive one)
        flags[0] = 1u; // the bounds of k_harmsum_b hold for p >= 0 only
}
__global__ __launch_bounds__(256) void k_join_span(float *__restrict__ dst, int64_t dst_stride,
                                                   const unsigned short *__restrict__ hi,
                                                   const unsigned short *__restrict__ lo, int64_t rlen, int64_t n)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (i < n)
        dst[r * dst_stride + i] = join_halves(hi[r * rlen + i], lo[r * rlen + i]);
}

template <int NSTAGES>
__global__ __launch_bounds__(256) void k_harmsum_b(
    const unsigned short *__restrict__ hi, const unsigned short *__restrict__ lo, int64_t rlen, int numz,
    int64_t col0, int64_t ncols, const int64_t *__restrict__ rowoff, const uint32_t *__restrict__ rowchg,
    HarmParams prm, DevHit *__restrict__ hits, unsigned int *__restrict__ count, unsigned int max_hits,
    int xcd_remap, const unsigned int *__restrict__ flags, const unsigned int *__restrict__ gate)
{
    if (gate && gate[kPfFlag] == 0)
        return; // the listed cells were summed instead
    constexpr int NH = HarmShape<NSTAGES>::NH, NHS = HarmShape<NSTAGES>::NHS;
    const int lane = threadIdx.x & 63;
    unsigned int tile = blockIdx.x;
    if (xcd_remap) {
        const unsigned int per = (gridDim.x + 7) / 8;
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if ((int64_t) tile * 256 >= ncols)
            return;
    }
    const int64_t c = (int64_t) tile * 256 + threadIdx.x;
    const bool live = c < ncols;
    const int64_t rind = col0 + (live ? c : 0); // dead lanes read a valid cell
    uint32_t off[NHS];                          // byte offset of each source column in a row of 16-bit halves
    {
        int i = 0;
#pragma unroll
        for (int s = 1; s < NSTAGES; s++)
#pragma unroll
            for (int h = 1; h < (1 << s); h += 2)
                off[i++] = (uint32_t) (((rind * h + (1 << (s - 1))) >> s) << 1);
    }
    const char *bbytes = reinterpret_cast<const char *>(hi);
    const char *lbytes = reinterpret_cast<const char *>(lo);
    const bool all_exact = flags[0] != 0; // a host has written a negative power into the plane
    uint32_t v[NHS]; // the sources' upper halves, 16 bits each
    for (int zi = 0; zi < numz; zi++) {
        const uint32_t fh = hi[(size_t) zi * (size_t) rlen + rind];
        float p = bound_value(fh);
        const uint32_t chg = NH > 0 ? rowchg[zi] : 0u;
        int64_t ro[NHS]; // byte offsets of the sources' rows in a plane of halves (2 * rlen * source row)
#pragma unroll
        for (int i = 0; i < NH; i++)
            ro[i] = rowoff[(size_t) zi * NHS + i];
        // The column offsets are made opaque once per row: left alone, the compiler adds each of them
        // to the plane's address outside the loop and keeps 15 64-bit addresses (30 registers)
        // instead of 15 offsets; this way a load is scalar row address + 32-bit lane offset.
#pragma unroll
        for (int i = 0; i < NH; i++)
            asm volatile("" : "+v"(off[i]));
#pragma unroll
        for (int i = 0; i < NH; i++) {
            if (chg & (1u << i)) {
                const int64_t r = ro[i];
                // kept as loaded; widened where it is added, so that no branch waits for its load
                v[i] = *reinterpret_cast<const unsigned short *>(bbytes + r + off[i]);
            }
        }
        bool any = all_exact;
        {
            int i = 0;
#pragma unroll
            for (int s = 0; s < NSTAGES; s++) {
                if (s > 0) {
#pragma unroll
                    for (int h = 1; h < (1 << s); h += 2)
                        p += bound_value(v[i++]);
                }
                any |= live && !(p <= prm.powcut[s]); // a NaN (the bound of an infinite power) counts as a crossing
            }
        }
        if (__ballot(any)) { // rare: this row again, exactly: the upper halves are in registers, the lower are fetched
            float q = join_halves(fh, lo[(size_t) zi * (size_t) rlen + rind]);
            float ps[NSTAGES];
            ps[0] = q;
            // four sources at a time, so that this path's loads do not set the kernel's register count
#pragma unroll
            for (int i0 = 0; i0 < NH; i0 += 4) {
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (i0 + k < NH) {
                        // opaque, and tied to the sum so far: twice the offset must not become a register
                        // per source outside the loop, and a group's loads must not start before the
                        // group before it has been added
                        uint32_t o = off[i0 + k];
                        asm volatile("" : "+v"(o) : "v"(q));
                        // (with 31 sources the upper half is fetched again: used from its register here, the
                        // kernel is allotted 111 registers instead of 96, four waves per SIMD instead of five)
                        const uint32_t h16 = NSTAGES >= 6 ? (uint32_t) *reinterpret_cast<const unsigned short *>(bbytes + ro[i0 + k] + o)
                                                          : v[i0 + k];
                        e[k] = join_halves(h16, *reinterpret_cast<const unsigned short *>(lbytes + ro[i0 + k] + o));
                    }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (i0 + k < NH) {
                        const int i = i0 + k;
                        q += e[k];
                        if (((i + 2) & (i + 1)) == 0) // i + 2 a power of two: the last source of its stage
                            ps[harm_stage_of(i)] = q;
                    }
            }
#pragma unroll
            for (int s = 0; s < NSTAGES; s++) {
                const bool hit = live && (ps[s] > prm.powcut[s]);
                const unsigned long long mask = __ballot(hit);
                if (mask) {
                    unsigned int base = 0;
                    const int leader = __ffsll((long long) mask) - 1;
                    if (lane == leader)
                        base = atomicAdd(count, (unsigned int) __popcll(mask));
                    base = __shfl(base, leader);
                    if (hit) {
                        const unsigned int pos = base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
                        if (pos < max_hits)
                            hits[pos] = DevHit{ps[s], ((uint32_t) s << 16) | (uint32_t) zi, rind};
                    }
                }
            }
        }
    }
}

// The same with NC columns per lane, 64 apart (a wave owns 64*NC consecutive columns): for
// harmonic h/2^s the k-th column's source cell lies exactly 64*k*h/2^s cells after the
// first's (an integer for s <= 6), so all columns share one offset register per source and
// their loads differ by an immediate offset.  NC times the loads in flight per wave for 15
// registers per extra column: the kernel is bound by load latency at full occupancy.
template <int NSTAGES, int NC>
__global__ __launch_bounds__(256) void k_harmsum_bw(
    const unsigned short *__restrict__ hi, const unsigned short *__restrict__ lo, int64_t rlen, int numz,
    int64_t col0, int64_t ncols, const int64_t *__restrict__ rowoff, const uint32_t *__restrict__ rowchg,
    HarmParams prm, DevHit *__restrict__ hits, unsigned int *__restrict__ count, unsigned int max_hits,
    int xcd_remap, const unsigned int *__restrict__ flags, const unsigned int *__restrict__ gate)
{
    if (gate && gate[kPfFlag] == 0)
        return; // the listed cells were summed instead
    constexpr int NH = HarmShape<NSTAGES>::NH, NHS = HarmShape<NSTAGES>::NHS;
    constexpr int TILE = 256 * NC;
    constexpr int GRP = 7; // sources fetched together on the exact path (8 would cost the 15-source kernel a wave per SIMD)
    const int lane = threadIdx.x & 63;
    unsigned int tile = blockIdx.x;
    if (xcd_remap) {
        const unsigned int per = (gridDim.x + 7) / 8;
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if ((int64_t) tile * TILE >= ncols)
            return;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int64_t c0 = (int64_t) tile * TILE + 64 * NC * wave; // the wave's first column
    if (c0 >= ncols)
        return; // nothing to do for the whole wave (the kernel has no barriers)
    const int64_t R0 = col0 + c0;
    bool live[NC];
#pragma unroll
    for (int k = 0; k < NC; k++)
        live[k] = c0 + lane + 64 * k < ncols;
    // dead lanes read valid cells: the wave's first column, and the plane extends a block past the last searched column
    const int t = live[0] ? lane : 0;
    const uint32_t toff = 2u * (uint32_t) t; // the lane's byte offset in the wave's first 128 bytes of a row of halves
    uint32_t off[NHS]; // byte offset of the first column's source cell in a row of 16-bit halves
    {
        int i = 0;
#pragma unroll
        for (int s = 1; s < NSTAGES; s++)
#pragma unroll
            for (int h = 1; h < (1 << s); h += 2)
                off[i++] = (uint32_t) ((((R0 + t) * h + (1 << (s - 1))) >> s) << 1);
    }
    const char *bbytes = reinterpret_cast<const char *>(hi);
    const char *lbytes = reinterpret_cast<const char *>(lo);
    const bool all_exact = flags[0] != 0; // a host has written a negative power into the plane
    uint32_t v[NC][NHS]; // the sources' upper halves, 16 bits each, per column
    for (int zi = 0; zi < numz; zi++) {
        const char *frow = bbytes + 2 * ((size_t) zi * (size_t) rlen + (size_t) R0); // scalar
        uint32_t f[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            f[k] = *reinterpret_cast<const unsigned short *>(frow + toff + 128 * k);
        const uint32_t chg = NH > 0 ? rowchg[zi] : 0u;
        int64_t ro[NHS]; // byte offsets of the sources' rows in a plane of halves (2 * rlen * source row)
#pragma unroll
        for (int i = 0; i < NH; i++)
            ro[i] = rowoff[(size_t) zi * NHS + i];
        // opaque once per row: a load is then scalar row address + 32-bit lane offset, not one of
        // 15 64-bit addresses kept in registers
#pragma unroll
        for (int i = 0; i < NH; i++)
            asm volatile("" : "+v"(off[i]));
#pragma unroll
        for (int i = 0; i < NH; i++) {
            if (chg & (1u << i)) {
                const int s = harm_stage_of(i), h = harm_num_of(i);
                const char *src = bbytes + ro[i] + off[i];
#pragma unroll
                for (int k = 0; k < NC; k++)
                    v[k][i] = *reinterpret_cast<const unsigned short *>(src + 2 * ((64 * k * h) >> s));
            }
        }
        // one column after the other: summed side by side the compiler packs the columns' 16-bit
        // values into shared registers, which costs a wait for its load inside every reload branch
        bool any = all_exact;
#pragma unroll
        for (int k = 0; k < NC; k++) {
            float p = bound_value(f[k]);
            int i = 0;
#pragma unroll
            for (int s = 0; s < NSTAGES; s++) {
                if (s > 0) {
#pragma unroll
                    for (int h = 1; h < (1 << s); h += 2)
                        p += bound_value(v[k][i++]);
                }
                any |= live[k] && !(p <= prm.powcut[s]); // a NaN (the bound of an infinite power) counts as a crossing
            }
        }
        if (__ballot(any)) { // rare: this row again, exactly, one column after the other: the upper halves are
                             // in registers already, the lower halves are fetched
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const bool lv = c0 + lane + 64 * k < ncols;
                const int64_t rind = R0 + t + 64 * k;
                const size_t fcell = (size_t) zi * (size_t) rlen + (size_t) rind;
                float q = join_halves(f[k], lo[fcell]);
                float ps[NSTAGES];
                ps[0] = q;
#pragma unroll
                for (int i0 = 0; i0 < NH; i0 += GRP) { // GRP sources at a time: few registers of its own
                    float e[GRP];
#pragma unroll
                    for (int j = 0; j < GRP; j++)
                        if (i0 + j < NH) {
                            // tied to the sum so far: a group's loads start when the group before is added
                            const int sI = harm_stage_of(i0 + j), hI = harm_num_of(i0 + j); // constants once unrolled
                            uint32_t o = off[i0 + j];
                            asm volatile("" : "+v"(o) : "v"(q));
                            o += 2u * (uint32_t) ((64 * k * hI) >> sI);
                            e[j] = join_halves(v[k][i0 + j], *reinterpret_cast<const unsigned short *>(lbytes + ro[i0 + j] + o));
                        }
#pragma unroll
                    for (int j = 0; j < GRP; j++)
                        if (i0 + j < NH) {
                            const int i = i0 + j;
                            q += e[j];
                            if (((i + 2) & (i + 1)) == 0) // the last source of its stage
                                ps[harm_stage_of(i)] = q;
                        }
                }
#pragma unroll
                for (int s = 0; s < NSTAGES; s++) {
                    const bool hit = lv && (ps[s] > prm.powcut[s]);
                    const unsigned long long mask = __ballot(hit);
                    if (mask) {
                        unsigned int base = 0;
                        const int leader = __ffsll((long long) mask) - 1;
                        if (lane == leader)
                            base = atomicAdd(count, (unsigned int) __popcll(mask));
                        base = __shfl(base, leader);
                        if (hit) {
                            const unsigned int pos = base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
                            if (pos < max_hits)
                                hits[pos] = DevHit{ps[s], ((uint32_t) s << 16) | (uint32_t) zi, rind};
                        }
                    }
                }
            }
        }
    }
}

// The same again for many harmonics (31 sources), where registers limit the waves in flight.
// What a column costs here is its 31 values and little else:
// * the lanes' source-column offsets are kept as k_harmsum_lean keeps them: for harmonic h/2^s,
//   (rind*h + 2^(s-1)) >> s = Q + ((B + t*h) >> s) with Q, B from the wave's first column R0 and
//   t the lane; the lane part is at most 62 cells, a byte offset below 128, four to a register;
// * the wave-uniform part -- the address of cell Q of the source's row -- is kept across the
//   LANES of one register pair: lane i holds source i's (plane + its row's byte offset + 2*Q),
//   read with v_readlane at the moment of a reload.  The row's 31 byte offsets are one vector
//   load (lane i fetches source i's), requested a row ahead, instead of 62 scalar registers.
// NC columns 64 apart share all of it, as in k_harmsum_bw.
template <int NSTAGES, int NC>
__global__ __launch_bounds__(256) void k_harmsum_bl(
    const unsigned short *__restrict__ hi, const unsigned short *__restrict__ lo, int64_t rlen, int numz,
    int64_t col0, int64_t ncols, const int64_t *__restrict__ rowoff, const uint32_t *__restrict__ rowchg,
    HarmParams prm, DevHit *__restrict__ hits, unsigned int *__restrict__ count, unsigned int max_hits,
    int xcd_remap, const unsigned int *__restrict__ flags, const unsigned int *__restrict__ gate)
{
    if (gate && gate[kPfFlag] == 0)
        return; // the listed cells were summed instead
    constexpr int NH = HarmShape<NSTAGES>::NH, NHS = HarmShape<NSTAGES>::NHS;
    static_assert(NH >= 1 && NH <= 32 && NSTAGES <= 6, "one lane and one mask bit per source");
    constexpr int NPK = (NH + 3) / 4;
    constexpr int TILE = 256 * NC;
    constexpr int GRP = 2; // sources fetched together on the exact path
    // (addresses rebuilt from lane reads are integers: told to be global memory, or the loads are flat ones)
    typedef __attribute__((address_space(1))) char gbyte;
    typedef __attribute__((address_space(1))) unsigned short gushort;
    const int lane = threadIdx.x & 63;
    unsigned int tile = blockIdx.x;
    if (xcd_remap) {
        const unsigned int per = (gridDim.x + 7) / 8;
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if ((int64_t) tile * TILE >= ncols)
            return;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int64_t c0 = (int64_t) tile * TILE + 64 * NC * wave; // the wave's first column
    if (c0 >= ncols)
        return; // nothing to do for the whole wave (the kernel has no barriers)
    const int64_t R0 = col0 + c0;
    bool live[NC];
#pragma unroll
    for (int k = 0; k < NC; k++)
        live[k] = c0 + lane + 64 * k < ncols;
    const int t = live[0] ? lane : 0; // dead lanes shadow the wave's first column
    const uint32_t toff = 2u * (uint32_t) t;
    uint32_t pk[NPK];
#pragma unroll
    for (int q = 0; q < NPK; q++)
        pk[q] = 0;
#pragma unroll
    for (int i = 0; i < NH; i++) {
        const int s = harm_stage_of(i), h = harm_num_of(i);
        const uint32_t B = (uint32_t) ((R0 * h + (1 << (s - 1))) & ((1 << s) - 1));
        pk[i >> 2] |= (((B + (uint32_t) (t * h)) >> s) << 1) << (8 * (i & 3));
    }
    // lane i < NH: source i of the summing order (stage s_i, numerator h_i)
    const int src_i = lane < NH ? lane : 0;
    const int s_i = 32 - __clz(src_i + 1);
    const int h_i = 2 * (src_i + 1 - (1 << (s_i - 1))) + 1;
    const uint64_t cellq = (uint64_t) reinterpret_cast<uintptr_t>(hi) + 2u * (uint64_t) ((R0 * h_i + (1 << (s_i - 1))) >> s_i);
    const uint64_t lo_minus_hi = (uint64_t) reinterpret_cast<uintptr_t>(lo) - (uint64_t) reinterpret_cast<uintptr_t>(hi);
    const char *bbytes = reinterpret_cast<const char *>(hi);
    const bool all_exact = flags[0] != 0; // a host has written a negative power into the plane
    uint32_t v[NC][NHS]; // the sources' upper halves, 16 bits each, per column
    uint32_t roff = 8u * (uint32_t) src_i; // the lane's byte offset in a row of the offsets' table
    uint32_t tb = toff;
    int64_t ro_next = rowoff[src_i];
    for (int zi = 0; zi < numz; zi++) {
        const uint64_t addr = cellq + (uint64_t) ro_next; // lane i: where source i's cell Q of this row is
        uint32_t alo = (uint32_t) addr, ahi = (uint32_t) (addr >> 32);
        // (lane offsets opaque once per row: a load is then scalar row address + 32-bit lane offset, not a
        // 64-bit address kept in two registers)
        asm volatile("" : "+v"(roff), "+v"(tb));
        ro_next = *reinterpret_cast<const int64_t *>(reinterpret_cast<const char *>(rowoff) +
                                                     (size_t) (zi + 1 < numz ? zi + 1 : zi) * (NHS * 8) + roff);
        const size_t frow = 2 * ((size_t) zi * (size_t) rlen + (size_t) R0); // scalar: the wave's first fundamental cell
        uint32_t f[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            f[k] = *reinterpret_cast<const unsigned short *>(bbytes + frow + 128 * k + tb);
        const uint32_t chg = rowchg[zi];
        // unpacked at each use: hoisted out of the loop, the unpacked offsets would take the
        // registers the packing saved
#pragma unroll
        for (int q = 0; q < NPK; q++)
            asm volatile("" : "+v"(pk[q]));
#pragma unroll
        for (int i = 0; i < NH; i++) {
            if (chg & (1u << i)) {
                const int s = harm_stage_of(i), h = harm_num_of(i);
                // (opaque here, so that the lane reads stay inside this branch: ahead of the branches
                // they would be 62 scalar registers again)
                asm volatile("" : "+v"(alo), "+v"(ahi));
                const uint64_t base = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) ahi, i) << 32) |
                                      (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) alo, i);
                const gbyte *src = (const gbyte *) base + ((pk[i >> 2] >> (8 * (i & 3))) & 0xFFu);
#pragma unroll
                for (int k = 0; k < NC; k++)
                    v[k][i] = *(const gushort *) (src + 2 * ((64 * k * h) >> s));
            }
        }
        bool any = all_exact;
#pragma unroll
        for (int k = 0; k < NC; k++) { // one column after the other (see k_harmsum_bw)
            float p = bound_value(f[k]);
            int i = 0;
#pragma unroll
            for (int s = 0; s < NSTAGES; s++) {
                if (s > 0) {
#pragma unroll
                    for (int h = 1; h < (1 << s); h += 2)
                        p += bound_value(v[k][i++]);
                }
                any |= live[k] && !(p <= prm.powcut[s]); // a NaN (the bound of an infinite power) counts as a crossing
            }
        }
        if (__ballot(any)) { // rare: this row again, exactly, one column after the other, both halves fetched
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const bool lv = c0 + lane + 64 * k < ncols;
                const int64_t rind = R0 + t + 64 * k;
                float q = join_halves(f[k], *reinterpret_cast<const unsigned short *>(
                                                reinterpret_cast<const char *>(lo) + frow + 128 * k + tb));
                // a stage's crossings are compacted as soon as its sum is complete (the order k_harmsum
                // writes them in), so that this path holds no sums beside the running one
                auto emit = [&](int s, float p) {
                    const bool hit = lv && (p > prm.powcut[s]);
                    const unsigned long long mask = __ballot(hit);
                    if (mask) {
                        unsigned int base = 0;
                        const int leader = __ffsll((long long) mask) - 1;
                        if (lane == leader)
                            base = atomicAdd(count, (unsigned int) __popcll(mask));
                        base = __shfl(base, leader);
                        if (hit) {
                            const unsigned int pos = base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
                            if (pos < max_hits)
                                hits[pos] = DevHit{p, ((uint32_t) s << 16) | (uint32_t) zi, rind};
                        }
                    }
                };
                emit(0, q);
#pragma unroll
                for (int i0 = 0; i0 < NH; i0 += GRP) {
                    float e[GRP];
#pragma unroll
                    for (int j = 0; j < GRP; j++)
                        if (i0 + j < NH) {
                            const int i = i0 + j;
                            const int s = harm_stage_of(i), h = harm_num_of(i); // constants once unrolled
                            // tied to the sum so far: a group's loads start when the group before is added
                            asm volatile("" : "+v"(pk[i >> 2]), "+v"(alo), "+v"(ahi) : "v"(q));
                            const uint64_t base = (((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) ahi, i) << 32) |
                                                   (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) alo, i)) +
                                                  2 * ((64 * k * h) >> s);
                            const uint32_t o = (pk[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                            e[j] = join_halves(*(const gushort *) ((const gbyte *) base + o),
                                               *(const gushort *) ((const gbyte *) (base + lo_minus_hi) + o));
                        }
#pragma unroll
                    for (int j = 0; j < GRP; j++)
                        if (i0 + j < NH) {
                            const int i = i0 + j;
                            q += e[j];
                            if (((i + 2) & (i + 1)) == 0) // the last source of its stage
                                emit(harm_stage_of(i), q);
                        }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// A coarser bound in front of the sums: the largest upper half of every eight consecutive columns, in steps of
// four, and the harmonic sum of those.
//
// The groups follow the transforms that wrote the plane: within a block, output n of the transform (column
// blk_col + n - shift) belongs to group n >> 2.  The plane of maxima has kGpb = 1868 16-bit entries per block
// and row; entry (b, g) stands for the WINDOW of the groups g and g + 1 of block b, i.e. the columns l = n - shift
// with (l + A) >> 2 in {g, g + 1}, A = shift & 3 -- eight columns, fewer where a block begins (A = 2) or ends
// (A = 0) with a group of two -- and holds their largest upper half plus one (bound_entry): shifted up 16 bits
// that is the float the sums use as the bound.  The windows of a block's last two entries reach over the next block's first two
// groups (k_group_fix sees to that after plane generation).  So any run of up to five consecutive columns,
// whatever block boundary it crosses, lies within the window of ONE entry: that of the group its first column
// is in.
//
// k_prefilter_lanes gives a lane one group of four fundamental columns (or two, 256 groups apart).  For harmonic
// h/2^s the source columns of those four lie within five consecutive columns, so one entry of the source's row is
// an upper bound for the source cell of each of the four, and -- floating-point addition being monotonic -- the
// sums of these bounds, taken in the reference's order, bound every stage's sum of every one of the four columns.
// A (row, group) whose bound sums stay at or below the thresholds has no crossing; the others are written to a
// list, and k_flagged_exact sums their columns exactly from both halves.
//
// Harmonics up to 1/4 take their bound from the plane of upper halves itself: the source cells of four
// consecutive columns are then at most two cells, the larger of which is a far tighter bound than a window of
// eight, and these sources move to a new row so rarely (once in 4 to 16 rows) that their two loads cost little.
// That tightness is what lets everything else -- the group's own entry and the sources above 1/4 -- do with one
// load from the windows: windows throughout list over 1 % of the (row, group)s of a 16-harmonic search in noise,
// this mixture 0.1 %, as few as the maxima of single groups did at two loads per source (the kernel is bound by
// the number of its load instructions).
// ---------------------------------------------------------------------------
struct GroupGrid {
    int64_t prepop_col0; // first column of the first block below the search region
    int64_t search_col0; // first column of the first search block
    int nprepop;         // blocks below the search region
    int nblk;            // blocks in all
    int a;               // shift & 3: where groups start relative to a block's first column
    int64_t mxlen;       // entries per row
};
__device__ __forceinline__ uint32_t group_entry_of(const GroupGrid &g, uint32_t c) // c >= prepop_col0
{
    const bool srch = c >= (uint32_t) g.search_col0;
    const uint32_t x = c - (srch ? (uint32_t) g.search_col0 : (uint32_t) g.prepop_col0);
    const uint32_t b = (srch ? (uint32_t) g.nprepop : 0u) + x / (uint32_t) kUseLen, l = x % (uint32_t) kUseLen;
    return b * (uint32_t) kGpb + ((l + (uint32_t) g.a) >> 2);
}
// Does harmonic h/2^s take its bound from the cells (true) or from the windows?
__host__ __device__ constexpr bool pf_from_cells(int s, int h) { return 4 * h <= (1 << s); }
// Where the bound of harmonic h/2^s for the fundamental columns cf..cl (at most four consecutive ones) is read,
// as a byte offset within the source's row: of the first source cell in the plane of upper halves, bit 0 set if
// the second cell is the next one (from cells); or of the entry of the first source column's group in the plane
// of maxima (from windows).  false: not bounded that way (a source below the plane's first block, a run across
// the gap between the blocks below the search region and the search blocks) -- such a group is always listed.
__device__ __forceinline__ bool source_offset(const GroupGrid &g, uint32_t cf, uint32_t cl, int s, int h, uint32_t &off)
{
    const uint32_t s0 = (uint32_t) (((uint64_t) cf * h + (1u << (s - 1))) >> s), s1 = (uint32_t) (((uint64_t) cl * h + (1u << (s - 1))) >> s);
    if (pf_from_cells(s, h)) {
        off = 2u * s0 + (s1 > s0 ? 1u : 0u);
        return s1 - s0 <= 1u;
    }
    off = 0;
    if (s0 < (uint32_t) g.prepop_col0)
        return false;
    const uint32_t e0 = group_entry_of(g, s0);
    const uint32_t d = group_entry_of(g, s1) - e0;
    off = 2u * e0;
    // the next entry's group is in the window; from a block's last two entries on, so are the next block's first two
    return d <= 1 || (d == 2 && e0 % (uint32_t) kGpb >= (uint32_t) kGpb - 2);
}

// The plane of maxima from the plane of upper halves (a host has written into the plane, or plane generation
// did not keep the maxima): one thread per entry, its window straight from the cells.
__global__ __launch_bounds__(256) void k_group_max(const unsigned short *__restrict__ hi, int64_t rlen,
                                                   const int64_t *__restrict__ blk_col, const int *__restrict__ blk_ncols,
                                                   GroupGrid g, unsigned short *__restrict__ mx)
{
    constexpr unsigned int per_block = (kGpb + 255) / 256;
    const int lg = (int) (blockIdx.x % per_block) * 256 + threadIdx.x;
    const int b = (int) (blockIdx.x / per_block), zi = blockIdx.z;
    if (lg >= kGpb)
        return;
    const unsigned short *row = hi + (size_t) zi * (size_t) rlen;
    const unsigned short *cells = row + blk_col[b];
    const int ncols = min(blk_ncols[b], kUseLen);
    unsigned int w = 0;
    for (int l = max(4 * lg - g.a, 0); l < min(4 * lg - g.a + 8, ncols); l++) // the groups lg and lg + 1
        w = max(w, (unsigned int) cells[l]);
    if (lg >= kGpb - 2 && b + 1 < g.nblk) { // ... and, from the block's last two entries, the next block's groups 0 and 1
        const unsigned short *next = row + blk_col[b + 1];
        for (int l = 0; l < min(8 - g.a, min(blk_ncols[b + 1], kUseLen)); l++)
            w = max(w, (unsigned int) next[l]);
    }
    mx[(size_t) zi * (size_t) g.mxlen + (size_t) b * kGpb + lg] = bound_entry(w);
}

// After plane generation has written the windows within every block: the windows of a block's last two
// entries also take in the next block's first entry (its groups 0 and 1).  That entry is not changed here, so
// the order in which the blocks are visited does not matter.
__global__ __launch_bounds__(256) void k_group_fix(unsigned short *__restrict__ mx, GroupGrid g, int numz)
{
    const int b = blockIdx.x * 256 + threadIdx.x, zi = blockIdx.y;
    if (b + 1 >= g.nblk)
        return;
    unsigned short *e = mx + (size_t) zi * (size_t) g.mxlen + (size_t) b * kGpb + (kGpb - 2);
    const unsigned short nx = e[2]; // entry 0 of block b + 1
    e[0] = max(e[0], nx);
    e[1] = max(e[1], nx);
}

struct WorkItem {
    uint32_t zi;
    uint32_t group; // index of the group of four fundamental columns within the search region
};
// work[0..]: the list.  ctl is kPfCtlWords 32-bit words, zeroed before every launch: ctl[0] = entries wanted in
// the list so far; ctl[kPfFlag], on a cache line of its own: 1 when the list is of no use (more entries than
// `limit`, or negative powers in the plane) and the plain sum has to run; ctl[32..33], on another line, one
// 64-bit word of k_prefilter_sample: bits 0..39 its samples that would be listed, bits 40..63 its workgroups
// that have finished.
//
// The list pays only while it is short: every entry costs four exact column sums, and a walk that ends
// in a list of no use has cost its whole time for nothing.  Looking while walking is what does not work:
// operations on one cache line that is being written are served one after the other, about 45 ns each (a
// returning atomic per workgroup every 16 rows added 0.9 ms to a 1.7 ms walk, a poll by every wave 10 ms);
// any way out of the row loop, even a changed loop bound, costs the kernels 10 to 30 registers and with them
// resident workgroups; and cutting the walk into shorter workgroups that report as they end multiplies a
// fixed cost per workgroup that is a third of the whole (4, 8, 16 row chunks: 3.4, 5.7, 8.9 ms).  So the
// question is asked BEFORE the walk: k_prefilter_sample works out the bound sums of some tens of thousands of
// (row, group)s spread evenly over the plane (one per lane: 30 microseconds), its last workgroup scales the
// count to the whole plane and raises the flag if the list would be longer than `limit`, and the walk's
// workgroups look at the flag when they begin.
// Entries of a workgroup's own list in LDS: six per thread (48 KB for the 1024 threads large planes are walked
// with: two workgroups per CU), 2048 at least.  A workgroup that walks a bright signal's columns lists thousands
// of (row, group)s, and every entry past its own list costs its wave an atomic on the global counter, in the row
// loop, while the rest of the device waits for that workgroup at the end of the kernel: with 2048 entries the
// walk of the benchmark's spectrum (two injected pulsars) took 1.10 ms, with 6144 1.04.
template <int NT> struct PfCap {
    static constexpr int value = 6 * NT > 2048 ? 6 * NT : 2048;
};
template <int CAP> struct PfShared {
    WorkItem lst[CAP];
    unsigned int ln;    // entries this workgroup wants to list (may exceed CAP: the rest went to work[] directly)
    unsigned int gbase;
    unsigned int dead;  // the walk had been given up when this workgroup began
};
__device__ __forceinline__ unsigned int pf_peek(const unsigned int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pf_give_up(unsigned int *ctl)
{
    __hip_atomic_store(&ctl[kPfFlag], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A workgroup begins: false if the walk was given up before it started.  (Contains a barrier.)
template <int CAP> __device__ __forceinline__ bool pf_begin(PfShared<CAP> &sh, const unsigned int *ctl)
{
    if (threadIdx.x == 0) {
        sh.ln = 0;
        sh.dead = pf_peek(&ctl[kPfFlag]);
    }
    __syncthreads();
    return sh.dead == 0;
}
// One row's flagged lanes of a wave (`mask` = ballot of `any`) into the workgroup's list first: one atomic on
// the global counter per workgroup, not per row.  A workgroup whose list is full (a bright signal fills the rows
// of its columns) sends the rest of the wave's row straight to the global list, again with one atomic.
template <int CAP>
__device__ __forceinline__ void pf_insert(PfShared<CAP> &sh, unsigned long long mask, bool any, int lane, const WorkItem &w,
                                          WorkItem *__restrict__ work, unsigned int *__restrict__ ctl, unsigned int limit)
{
    unsigned int base = 0;
    const int leader = __ffsll((long long) mask) - 1;
    if (lane == leader)
        base = atomicAdd(&sh.ln, (unsigned int) __popcll(mask));
    base = __shfl(base, leader);
    const unsigned int pos = base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
    if (any && pos < (unsigned int) CAP)
        sh.lst[pos] = w;
    const unsigned long long over = __ballot(any && pos >= (unsigned int) CAP);
    if (over) {
        unsigned int gb = 0;
        const int l2 = __ffsll((long long) over) - 1;
        if (lane == l2)
            gb = atomicAdd(&ctl[0], (unsigned int) __popcll(over));
        gb = __shfl(gb, l2);
        if (any && pos >= (unsigned int) CAP) {
            const unsigned int gpos = gb + (unsigned int) __popcll(over & ((1ull << lane) - 1ull));
            if (gpos < limit)
                work[gpos] = w;
            else // the global list is full too: the plain sum will run
                pf_give_up(ctl);
        }
    }
}
// The workgroup's list goes to the global one (after the barrier that ends the walk).  Whether it goes is
// decided by one thread: the flag can change between two threads' looks at it, and there is a barrier below.
constexpr unsigned int kPfNoBase = 0xFFFFFFFFu;
template <int NT, int CAP>
__device__ __forceinline__ void pf_flush(PfShared<CAP> &sh, WorkItem *__restrict__ work, unsigned int *__restrict__ ctl, unsigned int limit)
{
    const unsigned int n = min(sh.ln, (unsigned int) CAP);
    if (threadIdx.x == 0) {
        unsigned int base = kPfNoBase;
        if (n && !pf_peek(&ctl[kPfFlag])) {
            base = atomicAdd(&ctl[0], n);
            if (base + n > limit || base + n < base) { // the list is of no use: nobody will read it
                pf_give_up(ctl);
                base = kPfNoBase;
            }
        }
        sh.gbase = base;
    }
    __syncthreads();
    const unsigned int base = sh.gbase;
    if (base == kPfNoBase)
        return;
    for (unsigned int i = threadIdx.x; i < n; i += NT)
        work[base + i] = sh.lst[i];
}
// NG: groups per lane, 256 apart (a workgroup walks 256 * NG consecutive groups): the row's scalar work -- which
// sources moved, their rows' offsets, the branches -- is shared by NG times the loads.
// rowpf[zi][0 .. NHS]: the address of each source's row for fundamental row zi, in the plane of upper halves
// (sources bounded from cells) or in the plane of maxima (from windows), and in the last place the mask of the
// sources whose row moved since row zi - 1: everything a row's scalar side needs, in one run of memory.
template <int NSTAGES, int NG, int NT = 256>
__global__ __launch_bounds__(NT) void k_prefilter_lanes(
    const unsigned short *__restrict__ mx, int numz, GroupGrid g, int64_t ngroups, const int64_t *__restrict__ rowpf,
    HarmParams prm, WorkItem *__restrict__ work, unsigned int *__restrict__ ctl, unsigned int limit, int xcd_remap,
    const unsigned int *__restrict__ flags, uint32_t group0)
{
    constexpr int NH = HarmShape<NSTAGES>::NH, NHS = HarmShape<NSTAGES>::NHS;
    __shared__ PfShared<PfCap<NT>::value> sh;
    const int lane = threadIdx.x & 63;
    unsigned int tile = blockIdx.x;
    if (xcd_remap) {
        const unsigned int per = (gridDim.x + 7) / 8; // 8 XCDs
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if ((int64_t) tile * (NT * NG) >= ngroups)
            return;
    }
    if (flags[0] != 0) { // negative powers in the plane: the bounds do not hold
        if (blockIdx.x == 0 && threadIdx.x == 0)
            pf_give_up(ctl);
        return;
    }
    if (!pf_begin(sh, ctl))
        return; // given up before this workgroup started
    // the lane's groups among the `ngroups` walked, which begin with group0 of the search region (a context that
    // searches a range of blocks only)
    bool live[NG], force[NG]; // force: a source that one entry (or two cells) does not bound: always listed
    uint32_t gl[NG], foff[NG]; // the group's number in the search region; its own entry's byte offset in a row
    uint32_t off[NG][NHS];     // see source_offset()
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const int64_t gid = (int64_t) tile * (NT * NG) + NT * k + threadIdx.x;
        live[k] = gid < ngroups;
        gl[k] = group0 + (live[k] ? (uint32_t) gid : 0u);
        // the group's columns within the search region
        const uint32_t sb = gl[k] / (uint32_t) kGpb;
        const int lg = (int) (gl[k] % (uint32_t) kGpb);
        const int lfirst = max(4 * lg - g.a, 0), llast = min(4 * lg - g.a + 3, kUseLen - 1);
        const uint32_t cf = (uint32_t) g.search_col0 + sb * (uint32_t) kUseLen + (uint32_t) lfirst, cl = cf + (uint32_t) (llast - lfirst);
        foff[k] = 2u * ((uint32_t) (g.nprepop + sb) * (uint32_t) kGpb + (uint32_t) lg);
        force[k] = false;
        int i = 0;
#pragma unroll
        for (int s = 1; s < NSTAGES; s++)
#pragma unroll
            for (int h = 1; h < (1 << s); h += 2) {
                force[k] = !source_offset(g, cf, cl, s, h, off[k][i]) || force[k];
                i++;
                __builtin_amdgcn_sched_barrier(0); // one source after the other: interleaved, the set-up sets the register count
            }
    }
    unsigned long long live_mask[NG], force_mask[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        live_mask[k] = __ballot(live[k]);
        force_mask[k] = __ballot(force[k]);
    }
    const char *mbytes = reinterpret_cast<const char *>(mx);
    const int64_t rowb = 2 * g.mxlen;
    // Values are loaded into the UPPER half of their registers (the lower half is and stays zero): an entry of the
    // plane of maxima is then the float bound as it stands, a cell's upper half the float it is the upper half of.
    // The compiler has no such load for these parts (it assumes a 16-bit load may not leave the other half alone),
    // so the loads are written out, and so is the one wait per row -- the compiler does not count them.
    uint32_t fm[NG], v[NG][NHS], v2[NG][NHS]; // (v2: the second cell of the sources bounded from cells; unused otherwise)
#pragma unroll
    for (int k = 0; k < NG; k++) {
        fm[k] = 0;
        asm volatile("" : "+v"(fm[k]));
        int i = 0;
#pragma unroll
        for (int s = 1; s < NSTAGES; s++)
#pragma unroll
            for (int h = 1; h < (1 << s); h += 2) {
                v[k][i] = 0;
                asm volatile("" : "+v"(v[k][i]));
                if (pf_from_cells(s, h)) {
                    v2[k][i] = 0;
                    asm volatile("" : "+v"(v2[k][i]));
                }
                i++;
            }
    }
#define PF_LOAD_HI(dst, voff, sbase) asm volatile("global_load_short_d16_hi %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase))
    // (the first row loads every source -- what rowchg[0] says anyway: spelt out, the compiler knows that no source
    // value enters the loop undefined, and the kernel needs fewer registers)
    int zfirst = 0;
    asm volatile("" : "+s"(zfirst));
    for (int zi = zfirst; zi < numz; zi++) {
        const char *own = mbytes + (size_t) zi * (size_t) rowb;
#pragma unroll
        for (int k = 0; k < NG; k++)
            PF_LOAD_HI(fm[k], foff[k], own);
        const int64_t *rp = rowpf + (size_t) zi * (NHS + 1);
        int64_t ro[NHS];
#pragma unroll
        for (int i = 0; i < NH; i++)
            ro[i] = rp[i];
        const uint32_t chg = NH == 0 ? 0u : zi == zfirst ? ~0u : (uint32_t) rp[NHS];
#pragma unroll
        for (int k = 0; k < NG; k++)
#pragma unroll
            for (int i = 0; i < NH; i++)
                asm volatile("" : "+v"(off[k][i]));
        {
            int i = 0;
#pragma unroll
            for (int s = 1; s < NSTAGES; s++)
#pragma unroll
                for (int h = 1; h < (1 << s); h += 2) {
                    if (chg & (1u << i)) {
                        if (pf_from_cells(s, h)) {
                            const char *src = reinterpret_cast<const char *>(ro[i]);
#pragma unroll
                            for (int k = 0; k < NG; k++) {
                                const uint32_t o0 = off[k][i] & ~1u, o1 = o0 + ((off[k][i] & 1u) << 1);
                                PF_LOAD_HI(v[k][i], o0, src);
                                PF_LOAD_HI(v2[k][i], o1, src);
                            }
                        } else {
                            const char *src = reinterpret_cast<const char *>(ro[i]);
#pragma unroll
                            for (int k = 0; k < NG; k++)
                                PF_LOAD_HI(v[k][i], off[k][i], src);
                        }
                    }
                    i++;
                }
        }
        asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
        for (int k = 0; k < NG; k++) { // (ties every loaded register to the wait: the sums below come after it)
            asm volatile("" : "+v"(fm[k]));
            int i = 0;
#pragma unroll
            for (int s = 1; s < NSTAGES; s++)
#pragma unroll
                for (int h = 1; h < (1 << s); h += 2) {
                    asm volatile("" : "+v"(v[k][i]));
                    if (pf_from_cells(s, h))
                        asm volatile("" : "+v"(v2[k][i]));
                    i++;
                }
        }
#pragma unroll
        for (int k = 0; k < NG; k++) {
            float p = __uint_as_float(fm[k]);
            // (one compare per stage, each straight into a lane mask: as a bool per lane, or-ed up, the compiler spends a
            // dozen vector instructions on the five bits)
            unsigned long long mask = force_mask[k];
            int i = 0;
#pragma unroll
            for (int s = 0; s < NSTAGES; s++) {
                if (s > 0) {
#pragma unroll
                    for (int h = 1; h < (1 << s); h += 2) {
                        p += __uint_as_float(pf_from_cells(s, h) ? max(v[k][i], v2[k][i]) + 0x10000u : v[k][i]);
                        i++;
                    }
                }
                mask |= __ballot(!(p <= prm.powcut[s])); // a NaN (the bound of an infinite power) counts as a crossing
            }
            mask &= live_mask[k];
            if (mask)
                pf_insert(sh, mask, (mask >> lane) & 1ull, lane, WorkItem{(uint32_t) zi, gl[k]}, work, ctl, limit);
        }
    }
#undef PF_LOAD_HI
    __syncthreads();
    pf_flush<NT>(sh, work, ctl, limit);
}

// How long would the list be?  Asked before the walk (see the note on the control words): every lane takes one
// (row, group) -- sample k of nsamples: group k * gstep mod ngroups, gstep odd and near the golden section of
// ngroups, so that the samples are spread evenly over the columns; one row per wave, the waves going round the
// rows, so that a wave reads each source row's address once -- and works out its bound sums exactly as
// k_prefilter_lanes does for that row.  The last workgroup to finish scales the count: more than sample_limit of
// the samples listed, and the flag goes up.
template <int NSTAGES>
__global__ __launch_bounds__(256) void k_prefilter_sample(
    const unsigned short *__restrict__ mx, const unsigned short *__restrict__ hi, int numz, GroupGrid g, int64_t ngroups,
    const int64_t *__restrict__ rowpf, HarmParams prm, unsigned int *__restrict__ ctl, unsigned int nsamples, unsigned int gstep,
    unsigned int sample_limit, const unsigned int *__restrict__ flags, uint32_t group0)
{
    constexpr int NHS = HarmShape<NSTAGES>::NHS;
    __shared__ unsigned int found;
    if (flags[0] != 0) { // negative powers in the plane: the bounds do not hold
        if (blockIdx.x == 0 && threadIdx.x == 0)
            pf_give_up(ctl);
        return;
    }
    if (threadIdx.x == 0)
        found = 0;
    __syncthreads();
    const unsigned int k = blockIdx.x * 256 + threadIdx.x;
    const bool live = k < nsamples;
    const uint32_t gl = group0 + (uint32_t) (((uint64_t) (live ? k : 0u) * gstep) % (uint64_t) ngroups);
    const int zi = __builtin_amdgcn_readfirstlane((int) ((k >> 6) % (uint32_t) numz));
    const uint32_t sb = gl / (uint32_t) kGpb;
    const int lg = (int) (gl % (uint32_t) kGpb);
    const int lfirst = max(4 * lg - g.a, 0), llast = min(4 * lg - g.a + 3, kUseLen - 1);
    const uint32_t cf = (uint32_t) g.search_col0 + sb * (uint32_t) kUseLen + (uint32_t) lfirst, cl = cf + (uint32_t) (llast - lfirst);
    const char *mbytes = reinterpret_cast<const char *>(mx);
    const size_t rowb = 2 * (size_t) g.mxlen;
    const uint32_t fm = *reinterpret_cast<const unsigned short *>(mbytes + (size_t) zi * rowb +
                                                                  2 * ((size_t) (g.nprepop + sb) * kGpb + (size_t) lg));
    float p = __uint_as_float(fm << 16); // (an entry is the bound's upper half: bound_entry)
    bool any = !(p <= prm.powcut[0]);
    {
        int i = 0;
#pragma unroll
        for (int s = 1; s < NSTAGES; s++) {
#pragma unroll
            for (int h = 1; h < (1 << s); h += 2) {
                uint32_t off;
                any = !source_offset(g, cf, cl, s, h, off) || any; // not bounded by one entry or two cells: always listed
                const char *src = reinterpret_cast<const char *>(rowpf[(size_t) zi * (NHS + 1) + i]);
                if (pf_from_cells(s, h)) {
                    const uint32_t o0 = off & ~1u, o1 = o0 + ((off & 1u) << 1);
                    const uint32_t a = *reinterpret_cast<const unsigned short *>(src + o0);
                    const uint32_t b = *reinterpret_cast<const unsigned short *>(src + o1);
                    p += bound_value(max(a, b));
                } else {
                    p += __uint_as_float((uint32_t) *reinterpret_cast<const unsigned short *>(src + off) << 16);
                }
                i++;
            }
            any |= !(p <= prm.powcut[s]); // a NaN (the bound of an infinite power) counts as a crossing
        }
    }
    const unsigned long long mask = __ballot(any && live);
    if (mask && (threadIdx.x & 63) == 0)
        atomicAdd(&found, (unsigned int) __popcll(mask));
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long old = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(ctl + 32),
                                                              (unsigned long long) found | (1ull << 40), __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT);
        // the last one sees everybody's count: they all went through this one word
        if ((old >> 40) + 1 == gridDim.x && (old & ((1ull << 40) - 1)) + found > sample_limit)
            pf_give_up(ctl);
    }
}

// The listed (row, group)s, exactly: one thread per column, sums and order as k_harmsum's.
template <int NSTAGES>
__global__ __launch_bounds__(256) void k_flagged_exact(
    const unsigned short *__restrict__ hi, const unsigned short *__restrict__ lo, int64_t rlen, GroupGrid g,
    const int64_t *__restrict__ rowhalf, HarmParams prm, const WorkItem *__restrict__ work,
    const unsigned int *__restrict__ ctl, unsigned int work_cap, DevHit *__restrict__ hits,
    unsigned int *__restrict__ count, unsigned int max_hits)
{
    constexpr int NHS = HarmShape<NSTAGES>::NHS;
    // what became of the list, for the host, beside the launch's hit count: count[4] entries, [5] of no use, [8] = 1
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        count[4] = ctl[0];
        count[5] = ctl[kPfFlag];
        count[8] = 1u;
    }
    if (ctl[kPfFlag] != 0)
        return; // the plain sum runs instead
    const unsigned int n = min(ctl[0], work_cap);
    const int lane = threadIdx.x & 63;
    for (size_t tid = (size_t) blockIdx.x * 256 + threadIdx.x; (tid >> 2) < (((size_t) n + 63) & ~(size_t) 63);
         tid += (size_t) gridDim.x * 256) { // whole waves stay in the loop for the ballots
        const size_t e = tid >> 2;
        const int j = (int) (tid & 3);
        bool lv = e < n;
        const WorkItem w = work[lv ? e : 0];
        const int64_t sb = w.group / kGpb;
        const int lg = (int) (w.group % kGpb);
        const int l = 4 * lg - g.a + j;
        lv = lv && l >= 0 && l < kUseLen;
        const int64_t rind = g.search_col0 + sb * kUseLen + (lv ? l : max(4 * lg - g.a, 0));
        const int zi = (int) w.zi;
        const size_t fcell = (size_t) zi * (size_t) rlen + (size_t) rind;
        // First the column's own bounded sum, from the upper halves alone: nine columns in ten of the listed groups
        // stay below every threshold on those (the list comes from bounds of eight columns at a time), and need no
        // lower half at all.
        unsigned short uh[NHS + 1];
        uh[0] = hi[fcell];
        float qb = bound_value(uh[0]);
        bool cross = lv && !(qb <= prm.powcut[0]);
        {
            int i = 0;
#pragma unroll
            for (int s = 1; s < NSTAGES; s++) {
#pragma unroll
                for (int h = 1; h < (1 << s); h += 2) {
                    const size_t cell = (size_t) (rowhalf[(size_t) zi * NHS + i] >> 1) + (size_t) ((rind * h + (1 << (s - 1))) >> s);
                    uh[i + 1] = hi[cell];
                    qb += bound_value(uh[i + 1]);
                    i++;
                }
                cross = cross || (lv && !(qb <= prm.powcut[s])); // a NaN (the bound of an infinite power) counts
            }
        }
        if (!__any(cross))
            continue;
        lv = cross;
        float q = join_halves(uh[0], lo[fcell]);
        float ps[NSTAGES];
        ps[0] = q;
        {
            int i = 0;
#pragma unroll
            for (int s = 1; s < NSTAGES; s++) {
#pragma unroll
                for (int h = 1; h < (1 << s); h += 2) {
                    const size_t cell = (size_t) (rowhalf[(size_t) zi * NHS + i] >> 1) + (size_t) ((rind * h + (1 << (s - 1))) >> s);
                    q += join_halves(uh[i + 1], cross ? lo[cell] : (unsigned short) 0);
                    i++;
                }
                ps[s] = q;
            }
        }
#pragma unroll
        for (int s = 0; s < NSTAGES; s++) {
            const bool hit = lv && (ps[s] > prm.powcut[s]);
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                unsigned int base = 0;
                const int leader = __ffsll((long long) mask) - 1;
                if (lane == leader)
                    base = atomicAdd(count, (unsigned int) __popcll(mask));
                base = __shfl(base, leader);
                if (hit) {
                    const unsigned int pos = base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
                    if (pos < max_hits)
                        hits[pos] = DevHit{ps[s], ((uint32_t) s << 16) | (uint32_t) zi, rind};
                }
            }
        }
    }
}

// The summed powers of one stage (accelgpu_summed_block) from a plane kept as halves: the same
// sums as k_harmsum<.., DUMP>, one thread per column, nothing clever.
template <int NSTAGES>
__global__ __launch_bounds__(256) void k_dump_split(
    const unsigned short *__restrict__ hi, const unsigned short *__restrict__ lo, int64_t rlen, int numz,
    int64_t col0, int64_t ncols, const int64_t *__restrict__ rowoff, float *__restrict__ dump, int dump_stage)
{
    constexpr int NHS = HarmShape<NSTAGES>::NHS;
    const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols)
        return;
    const int64_t rind = col0 + c;
    for (int zi = 0; zi < numz; zi++) {
        const size_t fcell = (size_t) zi * (size_t) rlen + (size_t) rind;
        float p = join_halves(hi[fcell], lo[fcell]);
        float keep = p;
        int i = 0;
        for (int s = 1; s < NSTAGES; s++) {
            for (int h = 1; h < (1 << s); h += 2) {
                const size_t cell = (size_t) (rowoff[(size_t) zi * NHS + i] >> 2) + (size_t) ((rind * h + (1 << (s - 1))) >> s);
                p += join_halves(hi[cell], lo[cell]);
                i++;
            }
            if (s <= dump_stage)
                keep = p; // the sum after stage s
        }
        dump[(size_t) zi * ncols + c] = keep;
    }
}

// ---------------------------------------------------------------------------
// The one-column-per-thread harmonic sum with the source-column bookkeeping
// squeezed out of the vector registers, for many harmonics (where registers,
// not bytes, limit the waves in flight).  Same sums, same order, same hits.
//
// A wave owns 64 consecutive columns rind = R0 + t (R0 wave-uniform, t the lane).
// For harmonic h/2^s:  (rind*h + 2^(s-1)) >> s  =  Q + ((B + t*h) >> s)  with
// A = R0*h + 2^(s-1), Q = A >> s, B = A & (2^s - 1): Q is scalar arithmetic done
// at the moment of a reload, and the lane part is < 64, kept as a byte offset
// (<= 252) in one byte of a packed register: NH/4 registers instead of NH.
// ---------------------------------------------------------------------------

template <int NSTAGES, bool DUMP>
__global__ __launch_bounds__(256) void k_harmsum_lean(
    const float *__restrict__ plane, int64_t rlen, int numz, int64_t col0, int64_t ncols,
    const int64_t *__restrict__ rowoff, const uint32_t *__restrict__ rowchg, HarmParams prm,
    DevHit *__restrict__ hits, unsigned int *__restrict__ count, unsigned int max_hits,
    float *__restrict__ dump, int dump_stage, int xcd_remap)
{
    constexpr int NH = HarmShape<NSTAGES>::NH, NHS = HarmShape<NSTAGES>::NHS;
    constexpr int NPK = (NH + 3) / 4 > 0 ? (NH + 3) / 4 : 1;
    const int lane = threadIdx.x & 63;
    unsigned int tile = blockIdx.x;
    if (xcd_remap) {
        const unsigned int per = (gridDim.x + 7) / 8;
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int64_t c0 = (int64_t) tile * 256 + 64 * wave; // first column of this wave
    if (c0 >= ncols)
        return; // nothing to do for the whole wave (the kernel has no barriers)
    const int64_t R0 = col0 + c0;
    const bool live = c0 + lane < ncols;
    const int t = live ? lane : 0; // dead lanes shadow the wave's first column

    uint32_t pk[NPK];
#pragma unroll
    for (int q = 0; q < NPK; q++)
        pk[q] = 0;
#pragma unroll
    for (int i = 0; i < NH; i++) {
        const int s = harm_stage_of(i), h = harm_num_of(i);
        const uint32_t B = (uint32_t) ((R0 * h + (1 << (s - 1))) & ((1 << s) - 1));
        pk[i >> 2] |= (((B + (uint32_t) (t * h)) >> s) << 2) << (8 * (i & 3));
    }

    const char *pbytes = reinterpret_cast<const char *>(plane);
    const uint32_t toff = 4u * (uint32_t) t;
    float v[NHS];
    for (int zi = 0; zi < numz; zi++) {
        const char *frow = pbytes + 4 * ((size_t) zi * (size_t) rlen + (size_t) R0);
        float p = *reinterpret_cast<const float *>(frow + toff);
        const uint32_t chg = NH > 0 ? rowchg[zi] : 0u;
        // the packed offsets must be unpacked at each use: hoisted out of the loop,
        // the unpacked values would take the registers the packing saved
#pragma unroll
        for (int q = 0; q < NPK; q++)
            asm volatile("" : "+v"(pk[q]));
#pragma unroll
        for (int i = 0; i < NH; i++) {
            if (chg & (1u << i)) {
                const int s = harm_stage_of(i), h = harm_num_of(i);
                const int64_t Q = (R0 * h + (1 << (s - 1))) >> s;
                const char *row = pbytes + rowoff[(size_t) zi * NHS + i] + 4 * Q;
                v[i] = *reinterpret_cast<const float *>(row + ((pk[i >> 2] >> (8 * (i & 3))) & 0xFFu));
            }
        }
        float ps[NSTAGES];
        bool any = false;
        {
            int i = 0;
#pragma unroll
            for (int s = 0; s < NSTAGES; s++) {
                if (s > 0) {
#pragma unroll
                    for (int h = 1; h < (1 << s); h += 2) {
                        p += v[i];
                        i++;
                    }
                }
                ps[s] = p;
                any |= live && (p > prm.powcut[s]);
            }
        }
        if constexpr (DUMP) {
#pragma unroll
            for (int s = 0; s < NSTAGES; s++)
                if (s == dump_stage && live)
                    dump[(size_t) zi * ncols + (c0 + lane)] = ps[s];
        } else {
            if (__ballot(any)) { // rare: some lane of this wave crossed a threshold
#pragma unroll
                for (int s = 0; s < NSTAGES; s++) {
                    const bool hit = live && (ps[s] > prm.powcut[s]);
                    const unsigned long long mask = __ballot(hit);
                    if (mask) {
                        unsigned int base = 0;
                        const int leader = __ffsll((long long) mask) - 1;
                        if (lane == leader)
                            base = atomicAdd(count, (unsigned int) __popcll(mask));
                        base = __shfl(base, leader);
                        if (hit) {
                            const unsigned int pos =
                                base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
                            if (pos < max_hits)
                                hits[pos] = DevHit{ps[s], ((uint32_t) s << 16) | (uint32_t) zi,
                                                   R0 + lane};
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Block-granular forms of the same steps, for hosts that keep the reference's
// per-block call sequence (one block of numz x USELEN powers at a time).
// ---------------------------------------------------------------------------
// blockpow[zi][j] += plane[zind[zi]][(int)((2*rlo + j) * h / 2^s + 0.5)]
__global__ __launch_bounds__(256) void k_block_add_harm(float *__restrict__ blockpow,
                                                       const float *__restrict__ plane,
                                                       int64_t rlen, int rlo, int s, int h,
                                                       const int *__restrict__ zind)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int zi = blockIdx.y;
    if (j >= USELEN)
        return;
    const int64_t rr = (int64_t) 2 * rlo + j;
    const int64_t src = (rr * h + (1 << (s - 1))) >> s;
    blockpow[(size_t) zi * USELEN + j] += plane[(size_t) zind[zi] * rlen + src];
}

// the same from a plane kept as halves
__global__ __launch_bounds__(256) void k_block_add_harm_split(float *__restrict__ blockpow,
                                                             const unsigned short *__restrict__ hi,
                                                             const unsigned short *__restrict__ lo,
                                                             int64_t rlen, int rlo, int s, int h,
                                                             const int *__restrict__ zind)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int zi = blockIdx.y;
    if (j >= USELEN)
        return;
    const int64_t rr = (int64_t) 2 * rlo + j;
    const int64_t src = (rr * h + (1 << (s - 1))) >> s;
    const size_t cell = (size_t) zind[zi] * rlen + src;
    blockpow[(size_t) zi * USELEN + j] += join_halves(hi[cell], lo[cell]);
}

// Threshold scan of one block: every cell > powcut becomes a hit.
__global__ __launch_bounds__(256) void k_block_threshold(const float *__restrict__ blockpow,
                                                        float powcut, int stage, int64_t col0,
                                                        DevHit *__restrict__ hits,
                                                        unsigned int *__restrict__ count,
                                                        unsigned int max_hits)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int zi = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const float p = j < USELEN ? blockpow[(size_t) zi * USELEN + j] : 0.f;
    const bool hit = j < USELEN && p > powcut;
    const unsigned long long mask = __ballot(hit);
    if (mask) {
        unsigned int base = 0;
        const int leader = __ffsll((long long) mask) - 1;
        if (lane == leader)
            base = atomicAdd(count, (unsigned int) __popcll(mask));
        base = __shfl(base, leader);
        if (hit) {
            const unsigned int pos = base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < max_hits)
                hits[pos] = DevHit{p, ((uint32_t) stage << 16) | (uint32_t) zi, col0 + j};
        }
    }
}


// Block descriptors handed over as kernel arguments and written to device memory by the
// kernel itself: no host-to-device copy, so nothing to stage and nothing to wait for.
constexpr int DESC_MAX = 16;
struct BlockDescArgs {
    int64_t lobin[DESC_MAX], col[DESC_MAX];
    int numdata[DESC_MAX], ncols[DESC_MAX];
};
__global__ void k_write_desc(BlockDescArgs a, int n, int64_t *__restrict__ lobin,
                             int64_t *__restrict__ col, int *__restrict__ numdata,
                             int *__restrict__ ncols)
{
    const int i = threadIdx.x;
    if (i < n) {
        lobin[i] = a.lobin[i];
        col[i] = a.col[i];
        numdata[i] = a.numdata[i];
        ncols[i] = a.ncols[i];
    }
}

// ---------------------------------------------------------------------------
// Standard (not in-memory) mode, batched over nb blocks.  Block matrices are
// laid out [z][block][column] so that one correlate launch fills all blocks.
// ---------------------------------------------------------------------------
// rinds[b][j] = index_from_r(calc_required_r(f, startr_b + j/2), sub_rlo_b)
// (reference src/accel_utils.c:910-917, 102-117), in fp64 with every operation
// rounded on its own so the indices equal the host's.
__global__ __launch_bounds__(256) void k_make_rinds(unsigned short *__restrict__ rinds,
                                                   const double *__restrict__ startr,
                                                   const int *__restrict__ sub_rlo,
                                                   double harm_fract)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= USELEN)
        return;
    const double rr = afft::strict::add(startr[b], afft::strict::mul((double) j, 0.5));
    const double t = afft::strict::add(afft::strict::mul(afft::strict::mul(2.0, rr), harm_fract), 0.5);
    const double subr = afft::strict::mul((double) (int) t, 0.5);
    const double q = afft::strict::add(afft::strict::mul(afft::strict::add(subr, -(double) sub_rlo[b]), 2.0), 1e-14);
    rinds[(size_t) b * USELEN + j] = (unsigned short) (int) q;
}

// fund[zi][b][j] += sub[zind[zi]][b][rinds[b][j]]   (add_ffdotpows, :1119-1134)
__global__ __launch_bounds__(256) void k_add_sub(float *__restrict__ fund,
                                                const float *__restrict__ sub, int sub_stride,
                                                const unsigned short *__restrict__ rinds,
                                                const int *__restrict__ zind, int nb)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int zi = blockIdx.y, b = blockIdx.z;
    if (j >= USELEN)
        return;
    const int src = rinds[(size_t) b * USELEN + j];
    fund[((size_t) zi * nb + b) * USELEN + j] +=
        sub[((size_t) zind[zi] * nb + b) * sub_stride + src];
}

// One harmonic stage of the standard mode in one pass over the fundamental blocks: the
// NS sub-harmonics of the stage are added in the reference's order (harmnum 1, 3, 5, ...,
// each its own float add), the sum is stored and scanned against the stage's threshold.
// Sub-harmonic s lives at sub + s*sub_elems ([z][block][sub_stride]), its column table at
// rinds + s*rinds_elems ([block][USELEN]) and its row table at zind + s*numz.
template <int NS>
__global__ __launch_bounds__(256) void k_add_stage(
    float *__restrict__ fund, const float *__restrict__ sub, size_t sub_elems, int sub_stride,
    const unsigned short *__restrict__ rinds, size_t rinds_elems, const int *__restrict__ zind,
    int numz, int nb, float powcut, int stage, const int64_t *__restrict__ blk_col0,
    DevHit *__restrict__ hits, unsigned int *__restrict__ count, unsigned int max_hits)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int zi = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    float p = 0.f;
    if (j < USELEN) {
        const size_t cell = ((size_t) zi * nb + b) * USELEN + j;
        p = fund[cell];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int src = rinds[s * rinds_elems + (size_t) b * USELEN + j];
            p += sub[s * sub_elems + ((size_t) zind[s * numz + zi] * nb + b) * sub_stride + src];
        }
        fund[cell] = p;
    }
    const bool hit = j < USELEN && p > powcut;
    const unsigned long long mask = __ballot(hit);
    if (mask) {
        unsigned int base = 0;
        const int leader = __ffsll((long long) mask) - 1;
        if (lane == leader)
            base = atomicAdd(count, (unsigned int) __popcll(mask));
        base = __shfl(base, leader);
        if (hit) {
            const unsigned int pos = base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < max_hits)
                hits[pos] = DevHit{p, ((uint32_t) stage << 16) | (uint32_t) zi, blk_col0[b] + j};
        }
    }
}

// Threshold scan of nb blocks laid out [z][block][column].
__global__ __launch_bounds__(256) void k_batch_threshold(const float *__restrict__ pow,
                                                        float powcut, int stage,
                                                        const int64_t *__restrict__ blk_col0,
                                                        int nb, DevHit *__restrict__ hits,
                                                        unsigned int *__restrict__ count,
                                                        unsigned int max_hits)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int zi = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const float p = j < USELEN ? pow[((size_t) zi * nb + b) * USELEN + j] : 0.f;
    const bool hit = j < USELEN && p > powcut;
    const unsigned long long mask = __ballot(hit);
    if (mask) {
        unsigned int base = 0;
        const int leader = __ffsll((long long) mask) - 1;
        if (lane == leader)
            base = atomicAdd(count, (unsigned int) __popcll(mask));
        base = __shfl(base, leader);
        if (hit) {
            const unsigned int pos = base + (unsigned int) __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < max_hits)
                hits[pos] = DevHit{p, ((uint32_t) stage << 16) | (uint32_t) zi, blk_col0[b] + j};
        }
    }
}

} // namespace accelk
