// optimize_kernels.hpp -- candidate optimisation on the device: the (r, z) of
// every candidate is refined by maximising the (harmonic-summed) interpolated
// Fourier power with a downhill simplex, then local power, power, phase and
// their derivatives are measured at the optimum for every harmonic.
//
// Reference: optimize_accelcand() src/accel_utils.c:465-525;
// max_rz_arr_harmonics()/max_rz_arr() src/maximize_rz.c:22-92, 129-231;
// amoeba() src/amoeba.c; rz_interp() src/rzinterp.c:144-236; gen_z_response()/
// gen_r_response() src/responses.c:165-322; get_localpower3d()/get_derivs3d()
// src/characteristics.c:77-190.
//
// One workgroup per optimisation problem.  The simplex logic runs redundantly in
// every thread on identical values; only the objective is parallel: all threads
// share the taps of every harmonic's interpolation sum (each tap is a
// Fresnel-integral response sample in fp64 times one spectrum bin).  Local
// powers and derivatives are measured one harmonic per wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "accel_kernels.hpp"

namespace optk {

using afft::c32;
typedef double c64 __attribute__((ext_vector_type(2)));

constexpr int kMaxHarm = 32;
constexpr double kPi = 3.1415926535897932384626433832795;
constexpr double kTwoPi = 6.2831853071795864769252867665590;
constexpr double kZScale = 4.0; // src/maximize_rz.c:3

struct Problem {
    double rin, zin; // start, r relative to the first bin of the spectrum array
    int nharm;       // harmonics summed in the objective
    int r_offset;    // Fourier bin of the array's first element
    int polish;      // 1: harmonic-summed objective normalised by local powers
    int cand;        // candidate this problem belongs to
    int harm;        // 0-based harmonic it reports to when polish == 0
    int pad;
};

struct Rderivs { // layout of rderivs, include/presto.h:161-169
    double pow, phs, dpow, dphs, d2pow, d2phs, locpow;
};

struct Result {
    double rout, zout;
    int nfunk[2], pad[2]; // objective evaluations spent by the two simplex runs
    double pows[kMaxHarm];
    Rderivs derivs[kMaxHarm];
};

__device__ __forceinline__ int halfwidth_low(double z) // z_resp_halfwidth(z, LOWACC)
{
    z = fabs(z);
    int m = (int) (long long) (z * (0.00089 * z + 0.3131) + 16);
    m = m < 16 ? 16 : m;
    if (z > 100 && m > 0.6 * z)
        m = (int) (0.6 * z);
    return m;
}
__device__ __forceinline__ int halfwidth_high(double z) // z_resp_halfwidth(z, HIGHACC)
{
    z = fabs(z);
    int m = (int) (long long) (z * (0.002057 * z + 0.0377) + 16 * 3);
    m += 10 + 5;
    if (z > 100 && m > 1.2 * z)
        m = (int) (1.2 * z);
    return m;
}

// 1/x to ~1e-15 relative: hardware estimate plus two Newton steps (a full IEEE
// division costs several times as much; response samples are rounded to float).
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

// Fresnel integrals S(x), C(x) (the rational approximations of the reference's
// src/fresnl.c:175-216) given sin and cos of pi*x^2/2, which the caller has.
__device__ __forceinline__ void fresnel_with_trig(double xin, double s, double c, double *S,
                                                  double *C)
{
    using namespace accelk::fres;
    const double x = fabs(xin), x2 = x * x;
    double ss, cc;
    if (x2 < 2.5625) {
        const double t = x2 * x2;
        const double sdv = accelk::horner1(t, sd), cdv = accelk::horner(t, cd);
        const double q = fast_rcp(sdv * cdv);
        ss = x * x2 * accelk::horner(t, sn) * cdv * q;
        cc = x * accelk::horner(t, cn) * sdv * q;
    } else if (x > 36974.0) {
        ss = cc = 0.5;
    } else {
        const double t = fast_rcp(kPi * x2);
        const double u = t * t;
        const double fdv = accelk::horner1(u, fd), gdv = accelk::horner1(u, gd);
        const double q = fast_rcp(fdv * gdv);
        const double f = 1.0 - u * accelk::horner(u, fn) * gdv * q;
        const double g = t * accelk::horner(u, gn) * fdv * q;
        const double tx = t * x; // 1 / (pi x)
        cc = 0.5 + (f * s - g * c) * tx;
        ss = 0.5 - (f * c + g * s) * tx;
    }
    *S = xin < 0.0 ? -ss : ss;
    *C = xin < 0.0 ? -cc : cc;
}

// One sample of the response for |z| >= 1e-4 (see TapGen).  Kept out of line: the
// simplex kernel reaches it from many places and inlining every copy costs more
// registers than the call does.
__device__ __attribute__((noinline)) c64 chirp_tap(int ii, int numkern, double z, double signz,
                                                   double zd, double cons, double inv_absz,
                                                   double startr, double startroffset, double sA,
                                                   double cA)
{
    double re, im;
    const double r = startr - (double) ii;
    const double yy = r * zd;
    const double zz = yy + z * zd;
    double s1, c1, sy, cy, sz, cz;
    sincospi(r * r * inv_absz, &s1, &c1);
    const double s2 = s1 * cA + c1 * sA, c2 = c1 * cA - s1 * sA;
    fresnel_with_trig(yy, s1, c1, &sy, &cy);
    fresnel_with_trig(zz, s2, c2, &sz, &cz);
    const double tr = signz * (cz - cy);
    const double ti = sy - sz;
    const double s = signz * s1; // sin(pi r^2 / z)
    re = (tr * c1 - ti * s) * cons;
    im = -(tr * s + ti * c1) * cons;
    if (startroffset < 1e-3 && fabs(z) < 1e-3 && ii == numkern / 2) {
        const double z2 = z * z, x2 = startroffset * startroffset;
        re = 1.0 - 0.16449340668482264365 * z2;
        im = -0.5235987755982988731 * z;
        re += startroffset * 1.6449340668482264365 * z;
        im += startroffset * (kPi - 0.5167712780049970029 * z2);
        re += x2 * (-6.579736267392905746 + 0.9277056288952613070 * z2);
        im += x2 * (3.1006276680299820175 * z);
    }
    return c64{re, im};
}

// gen_z_response(roffset, numbetween = 1, z, numkern), one sample at a time, in
// fp64.  (The reference rounds its response array and the interpolation sum to
// float; that makes its objective noisy at the 1e-8 level, below which its
// simplex cannot converge: harmonic-summed candidates then wander until the
// 5000-evaluation cap.  Keeping the objective smooth lets the same simplex
// converge in ~150 evaluations onto the maximum the reference wanders around.)
// Everything that does
// not depend on the sample index is computed once.  With r = startr - ii,
// yy = r*zd, zz = (r + z)*zd and zd^2 = 2/|z|:
//   pi*yy^2/2 = pi*r^2/|z|                       (also the outer phase, up to sign)
//   pi*zz^2/2 = pi*r^2/|z| + pi*(2*sgn(z)*startr + |z|)   (mod 2*pi)
// so one sincospi per sample serves the phase factor and both Fresnel integrals.
struct TapGen {
    bool sinc_mode;
    int numkern;
    double roffset, z, signz, zd, cons, inv_absz, startr, startroffset, sA, cA;

    __device__ void init(double roffset_, double z_, int numkern_)
    {
        roffset = roffset_;
        z = z_;
        numkern = numkern_;
        const double absz = fabs(z);
        sinc_mode = absz < 1e-4;
        if (sinc_mode)
            return;
        startr = roffset - 0.5 * z;
        double ipart;
        const double fr = modf(startr, &ipart);
        startroffset = startr < 0 ? 1.0 + fr : fr;
        signz = z < 0.0 ? -1.0 : 1.0;
        zd = signz * 1.4142135623730950488 / sqrt(absz);
        cons = zd / 2.0;
        inv_absz = 1.0 / absz;
        startr += (double) (numkern / 2);
        sincospi(2.0 * signz * startr + absz, &sA, &cA);
    }

    __device__ c64 tap(int ii) const
    {
        double re, im;
        if (sinc_mode) { // gen_r_response
            const double x = (double) (numkern / 2 - ii) + roffset; // r / pi
            double s, c;
            sincospi(x, &s, &c);
            const double r = kPi * x;
            const double sinc = (r == 0.0) ? 1.0 : s / r;
            re = c * sinc;
            im = s * sinc;
            if (roffset < 1e-3 && ii == numkern / 2) {
                const double r2 = roffset * roffset;
                re = 1 - 6.579736267392905746 * r2;
                im = roffset * (kPi - 10.335425560099940058 * r2);
            }
            return c64{re, im};
        }
        return chirp_tap(ii, numkern, z, signz, zd, cons, inv_absz, startr, startroffset, sA, cA);
    }
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off);
    return v;
}

// rz_interp() by one wavefront; every lane returns the answer (rounded to float
// complex like the reference's result; the sum itself runs in fp64).
__device__ inline c32 wave_rz_interp(const c32 *__restrict__ data, int64_t numdata, double r,
                                     double z, int khw)
{
    const int lane = threadIdx.x & 63;
    if (r > (double) numdata - 1.0 || r < 0.0)
        return c32{0.f, 0.f};
    double dint;
    const double frac = modf(r, &dint);
    const int64_t intfreq = (int64_t) dint;
    if (fabs(z) < 1e-4) {
        if (frac < 1e-5)
            return data[intfreq];
        if (1.0 - frac < 1e-5)
            return data[intfreq + 1];
    }
    const int numkern = 2 * khw;
    int64_t lodata = intfreq - khw;
    int loresp = 0, hiresp = numkern;
    if (lodata < 0) {
        loresp = (int) -lodata;
        lodata = 0;
    }
    const int64_t hidata = intfreq + khw - 1;
    if (hidata > numdata - 1)
        hiresp = (int) (numkern - hidata + numdata - 1);
    double ar = 0.0, ai = 0.0;
    TapGen gen;
    gen.init(frac, z, numkern);
    for (int ii = loresp + lane; ii < hiresp; ii += 64) {
        const c64 p = gen.tap(ii);
        const c32 d = data[lodata + (ii - loresp)];
        ar += (double) d.x * p.x + (double) d.y * p.y;
        ai += (double) d.y * p.x - p.y * (double) d.x;
    }
    ar = wave_sum(ar);
    ai = wave_sum(ai);
    return c32{(float) ar, (float) ai};
}

__device__ __forceinline__ double cpower(c32 a)
{
    return (double) a.x * (double) a.x + (double) a.y * (double) a.y;
}

// get_localpower3d() with w = 0 by one wavefront.
__device__ inline double wave_localpower(const c32 *__restrict__ data, int64_t numdata, double r,
                                         double z)
{
    const int khw = halfwidth_low(z);
    const double nd = (double) numdata;
    double lo1 = r - 5 - 10, hi1 = lo1 + 10, lo2 = r + 5, hi2 = lo2 + 10, sum = 0.0;
    lo1 = fmin(fmax(lo1, 0.0), nd);
    hi1 = fmin(fmax(hi1, 0.0), nd);
    lo2 = fmin(fmax(lo2, 0.0), nd);
    hi2 = fmin(fmax(hi2, 0.0), nd);
    for (double f = lo1; f < hi1; f += 1.0)
        sum += cpower(wave_rz_interp(data, numdata, f, z, khw));
    for (double f = lo2; f < hi2; f += 1.0)
        sum += cpower(wave_rz_interp(data, numdata, f, z, khw));
    return sum / 20.0;
}

// get_derivs3d() with w = 0 by one wavefront.
__device__ inline Rderivs wave_derivs(const c32 *__restrict__ data, int64_t numdata, double r,
                                      double z, double localpower)
{
    const double h = 0.003, twoh = 0.006, twohsqrd = 0.000036;
    const int khw = halfwidth_high(z);
    double pwr[5], phs[5];
    double f = r - twoh;
#pragma unroll
    for (int ii = 0; ii < 5; ii++, f += h) {
        const c32 a = wave_rz_interp(data, numdata, f, z, khw);
        pwr[ii] = cpower(a);
        const double t = atan2((double) a.y, (double) a.x);
        phs[ii] = t > 0.0 ? t : t + kTwoPi;
    }
#pragma unroll
    for (int ii = 0; ii < 4; ii++)
        if (fabs(phs[ii + 1] - phs[ii]) > kPi)
            phs[ii + 1] -= kTwoPi;
    Rderivs d;
    d.pow = pwr[2];
    d.phs = phs[2];
    d.dpow = (pwr[3] - pwr[1]) / twoh;
    d.dphs = (phs[3] - phs[1]) / twoh;
    d.d2pow = (pwr[4] - 2.0 * pwr[2] + pwr[0]) / twohsqrd;
    d.d2phs = (phs[4] - 2.0 * phs[2] + phs[0]) / twohsqrd;
    d.locpow = localpower;
    return d;
}

// One thread's share (taps lane, lane + stride, ...) of the rz_interp() sum; *done
// is set when the reference returns early (r outside the data, or z ~ 0 at an
// integer bin) and then the full answer is returned.
__device__ inline void rz_interp_share(const c32 *__restrict__ data, int64_t numdata, double r,
                                       double z, int khw, int first, int stride, double *ar,
                                       double *ai, bool *done)
{
    *ar = *ai = 0.0;
    *done = true;
    if (r > (double) numdata - 1.0 || r < 0.0)
        return;
    double dint;
    const double frac = modf(r, &dint);
    const int64_t intfreq = (int64_t) dint;
    if (fabs(z) < 1e-4) {
        if (frac < 1e-5) {
            *ar = data[intfreq].x;
            *ai = data[intfreq].y;
            return;
        }
        if (1.0 - frac < 1e-5) {
            *ar = data[intfreq + 1].x;
            *ai = data[intfreq + 1].y;
            return;
        }
    }
    *done = false;
    const int numkern = 2 * khw;
    int64_t lodata = intfreq - khw;
    int loresp = 0, hiresp = numkern;
    if (lodata < 0) {
        loresp = (int) -lodata;
        lodata = 0;
    }
    const int64_t hidata = intfreq + khw - 1;
    if (hidata > numdata - 1)
        hiresp = (int) (numkern - hidata + numdata - 1);
    double sr = 0.0, si = 0.0;
    TapGen gen;
    gen.init(frac, z, numkern);
    for (int ii = loresp + first; ii < hiresp; ii += stride) {
        const c64 p = gen.tap(ii);
        const c32 d = data[lodata + (ii - loresp)];
        sr += (double) d.x * p.x + (double) d.y * p.y;
        si += (double) d.y * p.x - p.y * (double) d.x;
    }
    *ar = sr;
    *ai = si;
}

template <int NT> struct Objective {
    static constexpr int NW = NT / 64;
    const c32 *data;
    int64_t numdata;
    int nharm, r_offset, khw, polish;
    double *s_red;         // [kMaxHarm][NW][2] in LDS: per-wave partial sums
    double *s_pow;         // [kMaxHarm] in LDS
    const float *s_locpow; // [kMaxHarm] in LDS

    // -(summed power) at rz = (r, z / ZSCALE); same value in every thread.
    // nharm <= NW: each harmonic's interpolation sum is shared by NW/nharm
    // wavefronts; otherwise every wavefront takes nharm/NW harmonics in turn
    // (nharm and NW are powers of two).  Partial sums meet in LDS in a fixed order.
    __device__ double operator()(double rz0, double rz1) const
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int G = nharm <= NW ? NW / nharm : 1;  // wavefronts per harmonic
        const int per = nharm <= NW ? 1 : nharm / NW; // harmonics per wavefront
        for (int k = 0; k < per; k++) {
            const int h = (wave / G) * per + k; // 0-based harmonic
            const int g = wave % G;
            const int i = h + 1;
            double ar, ai;
            bool done;
            rz_interp_share(data, numdata, ((double) r_offset + rz0) * i - r_offset,
                            rz1 * kZScale * i, khw, g * 64 + lane, G * 64, &ar, &ai, &done);
            if (done) { // every thread holds the whole answer: count it once
                if (g != 0) ar = ai = 0.0;
            } else {
                ar = wave_sum(ar);
                ai = wave_sum(ai);
            }
            if (lane == 0) {
                s_red[(h * NW + g) * 2] = ar;
                s_red[(h * NW + g) * 2 + 1] = ai;
            }
        }
        __syncthreads();
        if ((int) threadIdx.x < nharm) {
            double re = 0.0, im = 0.0;
            for (int g = 0; g < G; g++) {
                re += s_red[(threadIdx.x * NW + g) * 2];
                im += s_red[(threadIdx.x * NW + g) * 2 + 1];
            }
            const double p = re * re + im * im;
            s_pow[threadIdx.x] = polish ? p / (double) s_locpow[threadIdx.x] : p;
        }
        __syncthreads();
        double sum = 0.0;
        for (int i = 0; i < nharm; i++)
            sum += s_pow[i];
        __syncthreads();
        return -sum;
    }
};

// amoeba(), src/amoeba.c:18-92, for a two-dimensional simplex.
template <class F>
__device__ inline int simplex(double p[3][2], double y[3], double ftol, const F &funk)
{
    int nfunk = 0;
    double psum[2] = {p[0][0] + p[1][0] + p[2][0], p[0][1] + p[1][1] + p[2][1]};
    auto attempt = [&](int ihi, double fac) {
        const double fac1 = 0.5 * (1.0 - fac), fac2 = fac1 - fac;
        const double t0 = psum[0] * fac1 - p[ihi][0] * fac2;
        const double t1 = psum[1] * fac1 - p[ihi][1] * fac2;
        const double ytry = funk(t0, t1);
        if (ytry < y[ihi]) {
            y[ihi] = ytry;
            psum[0] += t0 - p[ihi][0];
            p[ihi][0] = t0;
            psum[1] += t1 - p[ihi][1];
            p[ihi][1] = t1;
        }
        return ytry;
    };
    for (;;) {
        int ilo = 0, ihi, inhi;
        if (y[0] > y[1]) {
            inhi = 1;
            ihi = 0;
        } else {
            inhi = 0;
            ihi = 1;
        }
        for (int ii = 0; ii <= 2; ii++) {
            if (y[ii] <= y[ilo])
                ilo = ii;
            if (y[ii] > y[ihi]) {
                inhi = ihi;
                ihi = ii;
            } else if (y[ii] > y[inhi] && ii != ihi) {
                inhi = ii;
            }
        }
        const double rtol = 2.0 * fabs(y[ihi] - y[ilo]) / (fabs(y[ihi]) + fabs(y[ilo]) + 1.0e-15);
        if (rtol < ftol) {
            double t = y[0]; y[0] = y[ilo]; y[ilo] = t;
            t = p[0][0]; p[0][0] = p[ilo][0]; p[ilo][0] = t;
            t = p[0][1]; p[0][1] = p[ilo][1]; p[ilo][1] = t;
            return nfunk;
        }
        if (nfunk >= 5000)
            return nfunk;
        nfunk += 2;
        double ytry = attempt(ihi, -1.0);
        if (ytry <= y[ilo]) {
            ytry = attempt(ihi, 2.0);
        } else if (ytry >= y[inhi]) {
            const double ysave = y[ihi];
            ytry = attempt(ihi, 0.5);
            if (ytry >= ysave) {
                for (int ii = 0; ii <= 2; ii++)
                    if (ii != ilo) {
                        p[ii][0] = 0.5 * (p[ii][0] + p[ilo][0]);
                        p[ii][1] = 0.5 * (p[ii][1] + p[ilo][1]);
                        y[ii] = funk(p[ii][0], p[ii][1]);
                    }
                nfunk += 2;
                psum[0] = p[0][0] + p[1][0] + p[2][0];
                psum[1] = p[0][1] + p[1][1] + p[2][1];
            }
        } else {
            --nfunk;
        }
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_optimize(const c32 *__restrict__ data, int64_t numdata,
                                                const Problem *__restrict__ problems,
                                                Result *__restrict__ results)
{
    constexpr int NW = NT / 64;
    __shared__ double s_red[kMaxHarm * NW * 2];
    __shared__ double s_pow[kMaxHarm];
    __shared__ float s_locpow[kMaxHarm];
    const Problem pb = problems[blockIdx.x];
    Result *res = results + blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nharm = pb.nharm, off = pb.r_offset;
    const double rin = pb.rin, zin = pb.zin;

    if (pb.polish) {
        for (int i = 1 + wave; i <= nharm; i += NW) {
            const double lp = wave_localpower(data, numdata, ((double) off + rin) * i - off, zin * i);
            if (lane == 0)
                s_locpow[i - 1] = (float) lp;
        }
    }
    __syncthreads();

    Objective<NT> obj{data, numdata, nharm, off, halfwidth_low(fabs(zin * nharm) + 4.0), pb.polish,
                      s_red, s_pow, s_locpow};
    const double step = 0.4;
    double x[3][2] = {{rin - step, zin / kZScale - step},
                      {rin - step, zin / kZScale + step},
                      {rin + step, zin / kZScale}};
    double y[3];
    for (int i = 0; i < 3; i++)
        y[i] = obj(x[i][0], x[i][1]);
    const int nf1 = simplex(x, y, 1.0e-7, obj);

    obj.khw = halfwidth_high(fabs(x[0][1] * nharm) + 4.0);
    x[1][0] = x[0][0] + 0.01;
    x[1][1] = x[0][1];
    x[2][0] = x[0][0];
    x[2][1] = x[0][1] + 0.01;
    for (int i = 0; i < 3; i++)
        y[i] = obj(x[i][0], x[i][1]);
    const int nf2 = simplex(x, y, 1.0e-10, obj);

    const double rout = x[0][0], zout = x[0][1] * kZScale;
    if (threadIdx.x == 0) {
        res->rout = rout;
        res->zout = zout;
        res->nfunk[0] = nf1;
        res->nfunk[1] = nf2;
    }
    if (!pb.polish) {
        if (wave == 0) {
            const float lp = (float) wave_localpower(data, numdata, rout, zout);
            const Rderivs d = wave_derivs(data, numdata, rout, zout, (double) lp);
            if (lane == 0) {
                res->derivs[0] = d;
                res->pows[0] = -y[0];
            }
        }
        return;
    }
    for (int i = 1 + wave; i <= nharm; i += NW) {
        const double rh = ((double) off + rout) * i - off;
        const float lp = (float) wave_localpower(data, numdata, rh, zout * i);
        const c32 a = wave_rz_interp(data, numdata, rh, (zout / kZScale * i) * kZScale, obj.khw);
        const Rderivs d = wave_derivs(data, numdata, rh, zout * i, (double) lp);
        if (lane == 0) {
            res->pows[i - 1] = cpower(a);
            res->derivs[i - 1] = d;
        }
    }
}

} // namespace optk
