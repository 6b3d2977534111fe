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
// every thread on identical values; only the objective is parallel: wavefront w
// evaluates harmonics w+1, w+1+NW, ... and within a wavefront the 64 lanes share
// the taps of one interpolation sum (each tap is a Fresnel-integral response
// sample in fp64 times one spectrum bin).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "accel_kernels.hpp"

namespace optk {

using afft::c32;

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

// Sample ii of gen_z_response(roffset, numbetween = 1, z, numkern), rounded to
// float like the reference's response array.
__device__ inline c32 zresp_tap(double roffset, double z, int numkern, int ii)
{
    const double absz = fabs(z);
    double re, im;
    if (absz < 1e-4) { // gen_r_response
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
        return c32{(float) re, (float) im};
    }
    double startr = roffset - 0.5 * z;
    double ipart;
    const double fr = modf(startr, &ipart);
    const double startroffset = startr < 0 ? 1.0 + fr : fr;
    const double signz = z < 0.0 ? -1.0 : 1.0;
    const double zd = signz * 1.4142135623730950488 / sqrt(absz);
    const double cons = zd / 2.0;
    const double pibyz = kPi / z;
    startr += (double) (numkern / 2);
    const double r = startr - (double) ii;
    const double yy = r * zd;
    const double zz = yy + z * zd;
    double s, c, sy, cy, sz, cz;
    sincos(pibyz * r * r, &s, &c);
    accelk::fresnel(yy, &sy, &cy);
    accelk::fresnel(zz, &sz, &cz);
    const double tr = signz * (cz - cy);
    const double ti = sy - sz;
    re = (tr * c - ti * s) * cons;
    im = -(tr * s + ti * c) * cons;
    if (startroffset < 1e-3 && absz < 1e-3 && ii == numkern / 2) {
        const double z2 = z * z, x2 = startroffset * startroffset;
        re = 1.0 - 0.16449340668482264365 * z2;
        im = -0.5235987755982988731 * z;
        re += startroffset * 1.6449340668482264365 * z;
        im += startroffset * (kPi - 0.5167712780049970029 * z2);
        re += x2 * (-6.579736267392905746 + 0.9277056288952613070 * z2);
        im += x2 * (3.1006276680299820175 * z);
    }
    return c32{(float) re, (float) im};
}

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
    for (int ii = loresp + lane; ii < hiresp; ii += 64) {
        const c32 p = zresp_tap(frac, z, numkern, ii);
        const c32 d = data[lodata + (ii - loresp)];
        ar += (double) d.x * (double) p.x + (double) d.y * (double) p.y;
        ai += (double) d.y * (double) p.x - (double) p.y * (double) d.x;
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

template <int NT> struct Objective {
    static constexpr int NW = NT / 64;
    const c32 *data;
    int64_t numdata;
    int nharm, r_offset, khw, polish;
    double *s_part;        // [NW] in LDS
    const float *s_locpow; // [kMaxHarm] in LDS

    // -(summed power) at rz = (r, z / ZSCALE); same value in every thread.
    __device__ double operator()(double rz0, double rz1) const
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        double tot = 0.0;
        for (int i = 1 + wave; i <= nharm; i += NW) {
            const c32 a = wave_rz_interp(data, numdata, ((double) r_offset + rz0) * i - r_offset,
                                         rz1 * kZScale * i, khw);
            const double p = cpower(a);
            tot += polish ? p / (double) s_locpow[i - 1] : p;
        }
        if (lane == 0)
            s_part[wave] = tot;
        __syncthreads();
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w++)
            sum += s_part[w];
        __syncthreads();
        return -sum;
    }
};

// amoeba(), src/amoeba.c:18-92, for a two-dimensional simplex.
template <class F>
__device__ inline void simplex(double p[3][2], double y[3], double ftol, const F &funk)
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
            return;
        }
        if (nfunk >= 5000)
            return;
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
    __shared__ double s_part[NW];
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
                      s_part, s_locpow};
    const double step = 0.4;
    double x[3][2] = {{rin - step, zin / kZScale - step},
                      {rin - step, zin / kZScale + step},
                      {rin + step, zin / kZScale}};
    double y[3];
    for (int i = 0; i < 3; i++)
        y[i] = obj(x[i][0], x[i][1]);
    simplex(x, y, 1.0e-7, obj);

    obj.khw = halfwidth_high(fabs(x[0][1] * nharm) + 4.0);
    x[1][0] = x[0][0] + 0.01;
    x[1][1] = x[0][1];
    x[2][0] = x[0][0];
    x[2][1] = x[0][1] + 0.01;
    for (int i = 0; i < 3; i++)
        y[i] = obj(x[i][0], x[i][1]);
    simplex(x, y, 1.0e-10, obj);

    const double rout = x[0][0], zout = x[0][1] * kZScale;
    if (threadIdx.x == 0) {
        res->rout = rout;
        res->zout = zout;
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
