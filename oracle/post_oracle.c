/*
 * post_oracle.c -- CPU restatement of what accelsearch does with the candidate
 * list after the search loop (TEST INFRASTRUCTURE, see accel_oracle.h):
 * sorting, harmonic elimination, (r, z) optimisation by Fourier interpolation
 * and downhill simplex, and the derived signal properties.
 *
 * PARITY STATUS: unpinned (no reference fixtures exist for these steps); the
 * arithmetic follows the cited reference lines statement by statement,
 * including the places where the reference accumulates in single precision.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "accel_oracle.h"

#define POST_PI 3.1415926535897932384626433832795028841971693993751
#define POST_TWOPI 6.2831853071795864769252867665590057683943387987502
#define NUMFINTBINS 16   /* include/presto.h:108 */
#define NUMLOCPOWAVG 20  /* include/presto.h:100 */
#define DELTAAVGBINS 5   /* include/presto.h:104 */
#define ZSCALE 4.0       /* src/maximize_rz.c:3 */

/* z_resp_halfwidth(z, HIGHACC), src/responses.c:45-54 */
int orc_z_resp_halfwidth_high(double z)
{
    z = fabs(z);
    int m = (int) (long) (z * (0.002057 * z + 0.0377) + NUMFINTBINS * 3);
    m += ((NUMLOCPOWAVG >> 1) + DELTAAVGBINS);
    if (z > 100 && m > 1.2 * z)
        m = (int) (1.2 * z);
    return m;
}

/* rz_interp(), src/rzinterp.c:144-236.  The answer is accumulated in a float
 * complex, as there. */
void orc_rz_interp(const orc_cf32 *data, long numdata, double r, double z, int kern_half_width,
                   orc_cf32 *ans)
{
    if (r > numdata - 1.0 || r < 0.0) {
        ans->r = 0.0f;
        ans->i = 0.0f;
        return;
    }
    double dintfreq;
    const double fracfreq = modf(r, &dintfreq);
    const int intfreq = (int) dintfreq;
    if (fabs(z) < 1E-4) {
        if (fracfreq < 1E-5) {
            *ans = data[intfreq];
            return;
        }
        if ((1.0 - fracfreq) < 1E-5) {
            *ans = data[intfreq + 1];
            return;
        }
    }
    const int numkern = 2 * kern_half_width;
    orc_cf32 *response = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) numkern);
    orc_gen_z_response(fracfreq, 1, z, numkern, response);
    int lodata = intfreq - kern_half_width, loresp, hiresp;
    if (lodata < 0) {
        loresp = abs(lodata);
        lodata = 0;
    } else {
        loresp = 0;
    }
    const int hidata = intfreq + kern_half_width - 1;
    if (hidata > numdata - 1)
        hiresp = (int) (numkern - hidata + numdata - 1);
    else
        hiresp = numkern;
    const int nsum = hiresp - loresp;
    const orc_cf32 *d = data + lodata, *p = response + loresp;
    float ar = 0.0f, ai = 0.0f;
    for (int ii = 0; ii < nsum; ii++) {
        const double tmpd = d[ii].r, tmpr = p[ii].r;
        const float di = d[ii].i, pi = p[ii].i;
        ar = (float) ((double) ar + (tmpd * tmpr + (double) (di * pi)));
        ai = (float) ((double) ai + ((double) di * tmpr - (double) pi * tmpd));
    }
    ans->r = ar;
    ans->i = ai;
    free(response);
}

static double cpower(orc_cf32 a) /* POWER(), include/presto.h:72 (double arguments) */
{
    const double r = a.r, i = a.i;
    return r * r + i * i;
}

/* get_localpower3d() with w = 0, src/characteristics.c:77-135 */
double orc_get_localpower3d(const orc_cf32 *data, long numdata, double r, double z)
{
    const int binsperside = NUMLOCPOWAVG / 2;
    const int khw = orc_z_resp_halfwidth(z);
    double lo1 = r - DELTAAVGBINS - binsperside, hi1 = lo1 + binsperside;
    double lo2 = r + DELTAAVGBINS, hi2 = lo2 + binsperside, sum = 0.0, freq;
    orc_cf32 ans;
    if (lo1 < 0.0) lo1 = 0.0;
    if (lo2 < 0.0) lo2 = 0.0;
    if (hi1 < 0.0) hi1 = 0.0;
    if (hi2 < 0.0) hi2 = 0.0;
    if (lo1 > numdata) lo1 = (double) numdata;
    if (lo2 > numdata) lo2 = (double) numdata;
    if (hi1 > numdata) hi1 = (double) numdata;
    if (hi2 > numdata) hi2 = (double) numdata;
    for (freq = lo1; freq < hi1; freq += 1.0) {
        orc_rz_interp(data, numdata, freq, z, khw, &ans);
        sum += cpower(ans);
    }
    for (freq = lo2; freq < hi2; freq += 1.0) {
        orc_rz_interp(data, numdata, freq, z, khw, &ans);
        sum += cpower(ans);
    }
    return sum / (double) NUMLOCPOWAVG;
}

/* get_derivs3d() with w = 0, src/characteristics.c:138-190 */
void orc_get_derivs3d(const orc_cf32 *data, long numdata, double r, double z, double localpower,
                      orc_rderivs *res)
{
    const double h = 0.003, twoh = 0.006, twohsqrd = 0.000036;
    double pwr[5], phs[5], f;
    const int khw = orc_z_resp_halfwidth_high(z);
    orc_cf32 ans;
    int ii;
    for (ii = 0, f = r - twoh; ii < 5; ii++, f += h) {
        orc_rz_interp(data, numdata, f, z, khw, &ans);
        pwr[ii] = cpower(ans);
        const double t = atan2((double) ans.i, (double) ans.r); /* RADIAN_PHASE, include/presto.h:89-91 */
        phs[ii] = t > 0.0 ? t : t + POST_TWOPI;
    }
    for (ii = 0; ii < 4; ii++)
        if (fabs(phs[ii + 1] - phs[ii]) > POST_PI)
            phs[ii + 1] -= POST_TWOPI;
    res->pow = pwr[2];
    res->phs = phs[2];
    res->dpow = (pwr[3] - pwr[1]) / twoh;
    res->dphs = (phs[3] - phs[1]) / twoh;
    res->d2pow = (pwr[4] - 2.0 * pwr[2] + pwr[0]) / twohsqrd;
    res->d2phs = (phs[4] - 2.0 * phs[2] + phs[0]) / twohsqrd;
    res->locpow = localpower;
}

/* ---- downhill simplex in two dimensions, src/amoeba.c ------------------------ */
typedef struct {
    const orc_cf32 *data;
    long numdata;
    int nharm, r_offset, khw, divide;
    float locpow[ORC_MAXHARM];
} objective;

/* power_call_rz_harmonics() / power_call_rz(), src/maximize_rz.c:11-20, 113-127 */
static double objective_call(const objective *o, const double rz[2])
{
    double total = 0.0;
    orc_cf32 ans;
    for (int i = 1; i <= o->nharm; i++) {
        orc_rz_interp(o->data, o->numdata, (o->r_offset + rz[0]) * i - o->r_offset,
                      rz[1] * ZSCALE * i, o->khw, &ans);
        total += o->divide ? cpower(ans) / o->locpow[i - 1] : cpower(ans);
    }
    return -total;
}

static double amotry(double p[3][2], double *y, double *psum, const objective *o, int ihi,
                     double fac) /* src/amoeba.c:74-92 */
{
    const double fac1 = 0.5 * (1.0 - fac), fac2 = fac1 - fac;
    double ptry[2];
    ptry[0] = psum[0] * fac1 - p[ihi][0] * fac2;
    ptry[1] = psum[1] * fac1 - p[ihi][1] * fac2;
    const double ytry = objective_call(o, ptry);
    if (ytry < y[ihi]) {
        y[ihi] = ytry;
        psum[0] += ptry[0] - p[ihi][0];
        p[ihi][0] = ptry[0];
        psum[1] += ptry[1] - p[ihi][1];
        p[ihi][1] = ptry[1];
    }
    return ytry;
}

static void amoeba(double p[3][2], double *y, double ftol, const objective *o, int *nfunk)
{                               /* src/amoeba.c:18-71 */
    int ii, ihi, ilo, inhi;
    double rtol, ysave, ytry, psum[2], t;
    *nfunk = 0;
    psum[0] = p[0][0] + p[1][0] + p[2][0];
    psum[1] = p[0][1] + p[1][1] + p[2][1];
    for (;;) {
        ilo = 0;
        if (y[0] > y[1]) { inhi = 1; ihi = 0; } else { inhi = 0; ihi = 1; }
        for (ii = 0; ii <= 2; ii++) {
            if (y[ii] <= y[ilo])
                ilo = ii;
            if (y[ii] > y[ihi]) {
                inhi = ihi;
                ihi = ii;
            } else if (y[ii] > y[inhi] && ii != ihi)
                inhi = ii;
        }
        rtol = 2.0 * fabs(y[ihi] - y[ilo]) / (fabs(y[ihi]) + fabs(y[ilo]) + 1.0e-15);
        if (rtol < ftol) {
            t = y[0]; y[0] = y[ilo]; y[ilo] = t;
            t = p[0][0]; p[0][0] = p[ilo][0]; p[ilo][0] = t;
            t = p[0][1]; p[0][1] = p[ilo][1]; p[ilo][1] = t;
            break;
        }
        if (*nfunk >= 5000)
            return;
        *nfunk += 2;
        ytry = amotry(p, y, psum, o, ihi, -1.0);
        if (ytry <= y[ilo])
            ytry = amotry(p, y, psum, o, ihi, 2.0);
        else if (ytry >= y[inhi]) {
            ysave = y[ihi];
            ytry = amotry(p, y, psum, o, ihi, 0.5);
            if (ytry >= ysave) {
                for (ii = 0; ii <= 2; ii++) {
                    if (ii != ilo) {
                        p[ii][0] = psum[0] = 0.5 * (p[ii][0] + p[ilo][0]);
                        p[ii][1] = psum[1] = 0.5 * (p[ii][1] + p[ilo][1]);
                        y[ii] = objective_call(o, psum);
                    }
                }
                *nfunk += 2;
                psum[0] = p[0][0] + p[1][0] + p[2][0];
                psum[1] = p[0][1] + p[1][1] + p[2][1];
            }
        } else
            --(*nfunk);
    }
}

/* max_rz_arr_harmonics(), src/maximize_rz.c:129-231 (all harmonics read the same
 * spectrum with the same offset, as optimize_accelcand() calls it).  With
 * polish == 0 and nharm == 1 this is max_rz_arr(), :22-92: the same two simplex
 * runs on the raw power of one harmonic. */
void orc_max_rz(const orc_cf32 *data, long numdata, int nharm, int r_offset, double rin, double zin,
                int polish, double *rout, double *zout, orc_rderivs *derivs, double *power)
{
    objective o;
    double y[3], x[3][2];
    const double step = 0.4;
    int numeval, i;

    o.data = data;
    o.numdata = numdata;
    o.nharm = nharm;
    o.r_offset = r_offset;
    o.divide = polish;
    if (polish)
        for (i = 1; i <= nharm; i++)
            o.locpow[i - 1] = (float) orc_get_localpower3d(data, numdata,
                                                           (r_offset + rin) * i - r_offset, zin * i);
    o.khw = orc_z_resp_halfwidth(fabs(zin * nharm) + 4.0);
    x[0][0] = rin - step; x[0][1] = zin / ZSCALE - step;
    x[1][0] = rin - step; x[1][1] = zin / ZSCALE + step;
    x[2][0] = rin + step; x[2][1] = zin / ZSCALE;
    for (i = 0; i < 3; i++)
        y[i] = objective_call(&o, x[i]);
    numeval = 0;
    amoeba(x, y, 1.0e-7, &o, &numeval);
    o.khw = orc_z_resp_halfwidth_high(fabs(x[0][1] * nharm) + 4.0);
    x[1][0] = x[0][0] + 0.01; x[1][1] = x[0][1];
    x[2][0] = x[0][0];        x[2][1] = x[0][1] + 0.01;
    for (i = 0; i < 3; i++)
        y[i] = objective_call(&o, x[i]);
    numeval = 0;
    amoeba(x, y, 1.0e-10, &o, &numeval);
    *rout = x[0][0];
    *zout = x[0][1] * ZSCALE;
    if (!polish) {
        const float locpow = (float) orc_get_localpower3d(data, numdata, *rout, *zout);
        orc_get_derivs3d(data, numdata, *rout, *zout, locpow, &derivs[0]);
        power[0] = -y[0];
        return;
    }
    for (i = 1; i <= nharm; i++) {
        const double rh = (r_offset + *rout) * i - r_offset;
        const float locpow = (float) orc_get_localpower3d(data, numdata, rh, (*zout) * i);
        orc_cf32 ans;
        /* power_call_rz() with the width left over from the second simplex run */
        orc_rz_interp(data, numdata, rh, (*zout / ZSCALE * i) * ZSCALE, o.khw, &ans);
        power[i - 1] = cpower(ans);
        orc_get_derivs3d(data, numdata, rh, (*zout) * i, locpow, &derivs[i - 1]);
    }
}

/* optimize_accelcand(), src/accel_utils.c:465-525 (spectrum in memory). */
void orc_optimize_cand(const orc_obs *obs, int polish, orc_optcand *c)
{
    const int nh = c->numharm;
    if (polish) {
        double r, z;
        orc_max_rz(obs->fft, (long) obs->numbins, nh, (int) obs->lobin, c->r - obs->lobin, c->z, 1,
                   &r, &z, c->derivs, c->pows);
        for (int ii = 0; ii < nh; ii++) {
            c->hirs[ii] = (r + obs->lobin) * (ii + 1);
            c->hizs[ii] = z * (ii + 1);
        }
    } else {
        for (int ii = 0; ii < nh; ii++) {
            orc_max_rz(obs->fft, (long) obs->numbins, 1, 0, c->r * (ii + 1) - obs->lobin,
                       c->z * (ii + 1), 0, &c->hirs[ii], &c->hizs[ii], &c->derivs[ii], &c->pows[ii]);
            c->hirs[ii] += obs->lobin;
        }
    }
    int stage = 0;
    for (int n = nh; n > 1; n >>= 1)
        stage++;
    c->sigma = (float) orc_candidate_sigma(c->power, nh, (double) obs->numindep[stage]);
}

/* sort_accelcands() + eliminate_harmonics(), src/accel_utils.c:278-298, 383-462.
 * cands: n entries in list order; returns the number kept, sorted by decreasing
 * sigma (ties: decreasing power; the sort is stable like g_slist_sort). */
static int cmp_sigma(const orc_cand *a, const orc_cand *b)
{
    const int r = (a->sigma < b->sigma) - (a->sigma > b->sigma);
    if (r)
        return r;
    return (a->power < b->power) - (a->power > b->power);
}

int orc_sort_and_eliminate(orc_cand *cands, int n, int eliminate)
{
    /* stable insertion sort (n is a few hundred) */
    for (int i = 1; i < n; i++) {
        const orc_cand key = cands[i];
        int j = i - 1;
        while (j >= 0 && cmp_sigma(&cands[j], &key) > 0) {
            cands[j + 1] = cands[j];
            j--;
        }
        cands[j + 1] = key;
    }
    if (!eliminate || n < 2)
        return n;
    static const double ratios[][2] = {
        {3, 2}, {5, 2}, {2, 3}, {4, 3}, {5, 3}, {3, 4}, {5, 4}, {2, 5}, {3, 5}, {4, 5}, {5, 6}, {2, 7},
        {3, 7}, {4, 7}, {3, 8}, {5, 8}, {2, 9}, {3, 10}, {2, 11}, {3, 11}, {2, 13}, {3, 13}, {2, 15}};
    const double tooclose = 1.5;
    const int maxharm = 16;
    char *dead = (char *) calloc((size_t) n, 1);
    /* the reference walks "current" over the surviving list; "other" runs over
     * the survivors after it */
    for (int cur = 0; cur < n; cur++) {
        if (dead[cur])
            continue;
        /* the walk stops when current has no successor */
        int has_next = 0;
        for (int k = cur + 1; k < n; k++)
            if (!dead[k]) { has_next = 1; break; }
        if (!has_next)
            break;
        for (int oth = cur + 1; oth < n; oth++) {
            if (dead[oth])
                continue;
            const double cr = cands[cur].r, orr = cands[oth].r;
            int remove = 0;
            for (int ii = 1; ii <= maxharm && !remove; ii++)
                if (fabs(cr / ii - orr) < tooclose)
                    remove = 1;
            for (int ii = 1; ii <= maxharm && !remove; ii++)
                if (fabs(cr * ii - orr) < tooclose)
                    remove = 1;
            for (size_t k = 0; k < sizeof(ratios) / sizeof(ratios[0]) && !remove; k++)
                if (fabs(cr * ratios[k][0] / ratios[k][1] - orr) < tooclose)
                    remove = 1;
            if (remove)
                dead[oth] = 1;
        }
    }
    int m = 0;
    for (int i = 0; i < n; i++)
        if (!dead[i])
            cands[m++] = cands[i];
    free(dead);
    return m;
}
