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
    double (*plain)(double *); /* when set: the function minimised (orc_amoeba, used to pin the simplex) */
} objective;

/* power_call_rz_harmonics() / power_call_rz(), src/maximize_rz.c:11-20, 113-127 */
static double objective_call(const objective *o, const double rz[2])
{
    double total = 0.0;
    if (o->plain) {
        double x[2] = {rz[0], rz[1]};
        return o->plain(x);
    }
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

/* The simplex routine on a caller's function, with the reference's signature
 * (src/amoeba.c:18-19): lets the tests run it beside the reference's own amoeba(). */
void orc_amoeba(double p[3][2], double *y, double ftol, double (*funk)(double *), int *nfunk)
{
    objective o;
    memset(&o, 0, sizeof o);
    o.plain = funk;
    amoeba(p, y, ftol, &o, nfunk);
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

    o.plain = NULL;
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

/* ------------------------------------------------------------------------
 * Derived properties and the ACCEL text / .cand files.
 * ---------------------------------------------------------------------- */
#define POST_DBLCORRECT 1e-14
#define POST_SOL 299792458.0

/* calc_props(), src/characteristics.c:193-256.  Returns 0, or -1 where the
 * reference prints and exits (power or local power out of bounds). */
int orc_calc_props(const orc_rderivs *data, double r, double z, double w, orc_fourierprops *res)
{
    if (fabs(data->pow) < POST_DBLCORRECT || data->locpow < POST_DBLCORRECT)
        return -1;
    const double tmppow = data->pow / data->locpow;
    const double tmpd2pow = data->d2pow / data->locpow;
    res->pur = sqrt(1.5 * fabs(tmpd2pow) / tmppow) / POST_PI;
    if (fabs(tmppow) < POST_DBLCORRECT)
        res->pur = 1.0;
    res->purerr = 1.0 / (res->pur * sqrt(10.0 * tmppow));
    res->r = r;
    res->rerr = 3.0 / (POST_PI * res->pur * sqrt(6.0 * tmppow));
    res->z = z;
    res->zerr = 3.0 * sqrt(10.0) / (POST_PI * res->pur * res->pur * sqrt(tmppow));
    res->w = w;
    res->werr = 6.0 * sqrt(105.0) / (POST_PI * res->pur * res->pur * res->pur * sqrt(tmppow));
    res->pow = tmppow;
    res->powerr = sqrt(2.0 * tmppow);
    res->sig = sqrt(2.0 * tmppow - log(POST_PI * tmppow));
    res->rawpow = data->pow;
    res->phs = data->phs;
    res->phserr = 1.0 / res->powerr;
    res->cen = -data->dphs / POST_TWOPI;
    res->cenerr = 1.0 / sqrt(24.0 * tmppow);
    res->locpow = data->locpow;
    return 0;
}

/* calc_rzwerrs(), src/characteristics.c:351-394 */
void orc_calc_rzwerrs(const orc_fourierprops *props, double T, orc_rzwerrs *res)
{
    const double T2 = T * T, T3 = T2 * T;
    const double r2 = props->r * props->r, sr2 = props->rerr * props->rerr, r4 = r2 * r2;
    const double z2 = props->z * props->z, sz2 = props->zerr * props->zerr;
    const double tmp = props->r * props->w - 3 * z2;
    res->f = props->r / T;
    res->ferr = props->rerr / T;
    res->fd = props->z / T2;
    res->fderr = props->zerr / T2;
    res->fdd = props->w / T3;
    res->fdderr = props->werr / T3;
    res->p = T / props->r;
    res->perr = T * props->rerr / r2;
    res->pd = -props->z / r2;
    res->pderr = sqrt(4 * z2 * sr2 / (r4 * r2) + sz2 / r4);
    if (props->w == 0.0)
        res->pdd = 0.0;
    else
        res->pdd = (2 * z2 - props->r * props->w) / (T * r2 * props->r);
    res->pdderr = sqrt((props->werr * props->werr * r4 + 16 * sz2 * r2 + 4 * tmp * tmp * sr2) /
                       (r4 * r4 * T2));
}

static double copysign_of(double a, double b) { return b < 0.0 ? -fabs(a) : fabs(a); }

/* nice_output_1(), src/output.c:6-114: value with a one-digit error in parentheses. */
int orc_nice_output_1(char *output, double val, double err, int len)
{
    int nint, nfrac, totprec, errexp, errval, outexp;
    double rndval, outmant;
    char temp[64];

    sprintf(temp, "There is a problem with 'nice_output()'.\n");
    errexp = (fabs(err) == 0.0) ? 0 : (int) floor(log10(fabs(err)));
    errval = (int) floor(fabs(err) * pow(10.0, (double) (-errexp)) + POST_DBLCORRECT + 0.5);
    if (errval == 10) {
        errval = 1;
        errexp++;
    }
    rndval = pow(10.0, (double) errexp) * floor(val * pow(10.0, (double) (-errexp)) + 0.5);
    if (fabs(val) == 0.0) {
        nint = 1;
    } else {
        nint = (int) (ceil(log10(fabs(val))));
        if (nint == 0)
            nint++;
    }
    nfrac = -errexp;
    totprec = nint + nfrac;
    outexp = (fabs(rndval) == 0.0) ? 0 : (int) floor(log10(fabs(rndval)));
    outmant = rndval * pow(10.0, (double) (-outexp));
    if (fabs(1.0 - outmant) < POST_DBLCORRECT || fabs(-1.0 - outmant) < POST_DBLCORRECT)
        totprec++;
    if ((outexp >= 0 && errexp > 0) && outexp > errexp)
        sprintf(temp, "% .*f(%d)x10^%d", totprec - 1, copysign_of(outmant, rndval), errval, outexp);
    else if ((outexp >= 0 && errexp > 0) && outexp == errexp)
        sprintf(temp, "% d(%d)x10^%d", (int) (copysign_of(outmant, rndval)), errval, outexp);
    else if (outexp < -4 && outexp >= errexp)
        sprintf(temp, "% .*f(%d)x10^%d", totprec - 1, copysign_of(outmant, rndval), errval, outexp);
    else if (outexp < errexp && errexp != 0)
        sprintf(temp, "% d(%d)x10^%d", (int) (copysign_of(outmant, rndval) + POST_DBLCORRECT),
                errval, errexp);
    else if (nfrac == 0 && fabs(rndval) < 1.0e-15)
        sprintf(temp, "% d(%d)", (int) fabs(rndval), errval);
    else if (fabs(rndval) <= POST_DBLCORRECT && errexp < -5)
        sprintf(temp, "0.0(%d)x10^%d", errval, errexp + 1);
    else
        sprintf(temp, "% .*f(%d)", nfrac, rndval, errval);
    if (len == 0)
        sprintf(output, "%s", temp);
    else
        sprintf(output, "%*s", len, temp);
    return (int) strlen(output);
}

/* nice_output_2(), src/output.c:117-219: two-digit error.  Returns -1 where the
 * reference prints "undefined condition" and exits. */
int orc_nice_output_2(char *output, double val, double err, int len)
{
    int errexp, errval, outexp, valexp, tmp;
    double rndval, outmant;
    char temp[64];

    errexp = (fabs(err) == 0.0) ? 0 : (int) floor(log10(fabs(err)));
    errval = (int) floor(fabs(err) * pow(10.0, (double) (-errexp + 1)) + POST_DBLCORRECT + 0.5);
    if (errval == 100) {
        errval = 10;
        errexp++;
    }
    rndval = pow(10.0, (double) (errexp - 1)) * floor(val * pow(10.0, (double) (-errexp + 1)) + 0.5);
    outexp = (rndval == 0.0) ? 0 : (int) floor(log10(fabs(rndval)));
    valexp = (val == 0.0) ? 0 : (int) floor(log10(fabs(val)));
    outmant = rndval * pow(10.0, (double) (-outexp));
    if (outexp < -4) {
        tmp = outexp - errexp + 1;
        if (tmp < 2) {
            if (tmp == 0)
                sprintf(temp, "% .1f(%d.%d)x10^%d", outmant / 10, errval / 10, errval % 10, outexp + 1);
            else
                sprintf(temp, "% .1f(%d.%d)x10^%d", outmant, errval / 10, errval % 10, outexp);
        } else
            sprintf(temp, "% .*f(%.2d)x10^%d", tmp, outmant, errval, outexp);
    } else if (errexp - valexp > 1 && (errexp != 0 && errexp != 1)) {
        sprintf(temp, "0.0(%d.%d)x10^%d", errval / 10, errval % 10, errexp);
    } else if (errexp == 0) {
        if (fabs(rndval) < 0.1)
            rndval = fabs(rndval);
        sprintf(temp, "% .1f(%d.%d)", rndval, errval / 10, errval % 10);
    } else if (errexp < 0) {
        if (fabs(rndval) * pow(10.0, (double) (-errexp + 1)) < 0.1)
            rndval = fabs(rndval);
        if (fabs(rndval) <= POST_DBLCORRECT && errexp < -4)
            sprintf(temp, "0.0(%d.%d)x10^%d", errval / 10, errval % 10, errexp);
        else
            sprintf(temp, "% .*f(%.2d)", -errexp + 1, rndval, errval);
    } else if (errexp - outexp == 0) {
        if (errexp == 1)
            sprintf(temp, "% d(%.2d)", (int) rndval, errval);
        else
            sprintf(temp, "% .1f(%d.%d)x10^%d", outmant, errval / 10, errval % 10, errexp);
    } else if (errexp - outexp == 1) {
        if (fabs(outmant) < 0.99999999999999)
            outmant = fabs(outmant);
        sprintf(temp, "% .1f(%d.%d)x10^%d", outmant / 10.0, errval / 10, errval % 10, errexp);
    } else if ((errexp > 1 && outexp > 1) && errexp - outexp <= -1) {
        sprintf(temp, "% .*f(%.2d)x10^%d", outexp - errexp + 1, outmant, errval, outexp);
    } else if (errexp == 1 && outexp > 0) {
        sprintf(temp, "% d(%.2d)", (int) rndval, errval);
    } else {
        return -1;
    }
    if (len == 0)
        sprintf(output, "%s", temp);
    else
        sprintf(output, "%*s", len, temp);
    return (int) strlen(output);
}

static void center_string(char *out, const char *in, int width) /* src/accel_utils.c:526-544 */
{
    const int len = (int) strlen(in);
    memset(out, ' ', (size_t) width);
    out[width] = '\0';
    if (len >= width)
        strncpy(out, in, (size_t) width);
    else
        memcpy(out + (width - len) / 2, in, (size_t) len);
}

static void write_val_with_err(FILE *f, double val, double err, int numerr, int width)
{                               /* :547-562 */
    char tmpstr[64], ctrstr[64];
    if (numerr == 1)
        orc_nice_output_1(tmpstr, val, err, 0);
    else
        orc_nice_output_2(tmpstr, val, err, 0);
    center_string(ctrstr, tmpstr, width);
    fprintf(f, "%s  ", ctrstr);
}

/* compare_rzw_cands(), src/accel_utils.c:87-117 */
static void compare_rzw_cands(const orc_fourierprops *list, int nlist, char *notes)
{
    char tmp[32];
    for (int ii = 0; ii < nlist; ii++)
        for (int jj = 0; jj < nlist; jj++) {
            if (ii == jj)
                continue;
            if (fabs(list[ii].r - list[jj].r) < 15.0 && fabs(list[ii].z - list[jj].z) > 1.0 &&
                list[ii].pow > list[jj].pow) {
                if (strncmp(notes + jj * 20, "                      ", 20) == 0) {
                    sprintf(tmp, "SL? of Cand %d", ii + 1);
                    strncpy(notes + jj * 20, tmp, 20);
                }
                continue;
            }
            for (int kk = 1; kk < 61; kk++)
                if ((fabs(list[ii].r - list[jj].r / kk) < list[jj].rerr * 3) &&
                    (fabs(list[ii].z - list[jj].z / kk) < list[jj].zerr * 2)) {
                    if (strncmp(notes + jj * 20, "                      ", 20) == 0) {
                        sprintf(tmp, "H %d of Cand %d", kk, ii + 1);
                        strncpy(notes + jj * 20, tmp, 20);
                        break;
                    }
                }
        }
}

static void header_rows(FILE *f, const char **t1, const char **t2, const int *w, int ncols,
                        const char *alt3)
{
    char ctr[64], dash[64];
    int ii;
    for (ii = 0; ii < ncols - 1; ii++) {
        center_string(ctr, t1[ii], w[ii]);
        fprintf(f, "%s  ", ctr);
    }
    center_string(ctr, t1[ii], w[ii]);
    fprintf(f, "%s\n", ctr);
    for (ii = 0; ii < ncols - 1; ii++) {
        center_string(ctr, (alt3 && ii == 3) ? alt3 : t2[ii], w[ii]);
        fprintf(f, "%s  ", ctr);
    }
    center_string(ctr, t2[ii], w[ii]);
    fprintf(f, "%s\n", ctr);
    for (ii = 0; ii < ncols - 1; ii++) {
        memset(dash, '-', (size_t) w[ii]);
        dash[w[ii]] = '\0';
        fprintf(f, "%s--", dash);
    }
    memset(dash, '-', (size_t) w[ii]);
    dash[w[ii]] = '\0';
    fprintf(f, "%s\n", dash);
}

/* The tail of accelsearch (src/accelsearch.c:256-286): fundamental properties,
 * output_fundamentals() and output_harmonics() (src/accel_utils.c:565-835; the
 * pulsar-database comparison is skipped as for telescope "None"), the .inf text
 * appended, and the .cand file of fourierprops.  Returns 0 or -1. */
int orc_write_accel_files(const orc_obs *obs, orc_optcand *cands, int numcands,
                          const char *accelnm, const char *candnm, const char *inftext)
{
    static const char *ft1[] = {"", "", "Summed", "Coherent", "Num", "Period", "Frequency",
                                "FFT 'r'", "Freq Deriv", "FFT 'z'", "Accel", ""};
    static const char *ft2[] = {"Cand", "Sigma", "Power", "Power", "Harm", "(ms)", "(Hz)", "(bin)",
                                "(Hz/s)", "(bins)", "(m/s^2)", "Notes"};
    static const int fw[12] = {4, 5, 6, 8, 4, 16, 15, 15, 15, 11, 15, 20};
    static const char *ht1[] = {"", "", "", "Power /", "Raw", "FFT 'r'", "Pred 'r'", "FFT 'z'",
                                "Pred 'z'", "Phase", "Centroid", "Purity", ""};
    static const char *ht2[] = {"Cand", "Harm", "Sigma", "Loc Pow", "Power", "(bin)", "(bin)",
                                "(bins)", "(bins)", "(rad)", "(0-1)", "<p> = 1", "Notes"};
    static const int hw[13] = {5, 4, 5, 15, 11, 18, 13, 12, 9, 12, 10, 10, 20};
    char tmpstr[64], ctrstr[64];
    orc_fourierprops *props = (orc_fourierprops *) calloc((size_t) numcands, sizeof(orc_fourierprops));
    orc_rzwerrs errs;
    int ii, jj;

    for (ii = 0; ii < numcands; ii++) {                      /* accelsearch.c:258-270 */
        if (orc_calc_props(&cands[ii].derivs[0], cands[ii].r, cands[ii].z, 0.0, props + ii)) {
            free(props);
            return -1;
        }
        props[ii].rerr = (float) (ORC_DR) / cands[ii].numharm;
        props[ii].zerr = (float) (ORC_DZ) / cands[ii].numharm;
    }
    FILE *f = fopen(accelnm, "w");
    if (!f) {
        free(props);
        return -1;
    }
    char *notes = (char *) malloc((size_t) numcands * 20 + 1);
    memset(notes, ' ', (size_t) numcands * 20);
    compare_rzw_cands(props, numcands, notes);
    header_rows(f, ft1, ft2, fw, 12, NULL);
    for (ii = 0; ii < numcands; ii++) {
        const orc_optcand *cand = &cands[ii];
        double coherent_r = 0.0, coherent_i = 0.0;
        orc_calc_rzwerrs(props + ii, obs->T, &errs);
        const double phs0 = cand->derivs[0].phs;
        for (jj = 0; jj < cand->numharm; jj++) {
            const orc_rderivs harm = cand->derivs[jj];
            const double amp = obs->nph > 0.0 ? sqrt(harm.pow / obs->nph) : sqrt(harm.pow / harm.locpow);
            const double phscorr = phs0 - fmod((jj + 1.0) * phs0, POST_TWOPI);
            coherent_r += amp * cos(harm.phs + phscorr);
            coherent_i += amp * sin(harm.phs + phscorr);
        }
        const double coherent_pow = coherent_r * coherent_r + coherent_i * coherent_i;
        sprintf(tmpstr, "%-4d", ii + 1);
        center_string(ctrstr, tmpstr, fw[0]);
        fprintf(f, "%s  ", ctrstr);
        sprintf(tmpstr, "%.2f", cand->sigma);
        center_string(ctrstr, tmpstr, fw[1]);
        fprintf(f, "%s  ", ctrstr);
        sprintf(tmpstr, "%.2f", cand->power);
        center_string(ctrstr, tmpstr, fw[2]);
        fprintf(f, "%s  ", ctrstr);
        sprintf(tmpstr, "%.2f", coherent_pow);
        center_string(ctrstr, tmpstr, fw[3]);
        fprintf(f, "%s  ", ctrstr);
        sprintf(tmpstr, "%d", cand->numharm);
        center_string(ctrstr, tmpstr, fw[4]);
        fprintf(f, "%s  ", ctrstr);
        write_val_with_err(f, errs.p * 1000.0, errs.perr * 1000.0, 1, fw[5]);
        write_val_with_err(f, errs.f, errs.ferr, 1, fw[6]);
        write_val_with_err(f, props[ii].r, props[ii].rerr, 2, fw[7]);
        write_val_with_err(f, errs.fd, errs.fderr, 1, fw[8]);
        write_val_with_err(f, props[ii].z, props[ii].zerr, 2, fw[9]);
        const double accel = props[ii].z * POST_SOL / (obs->T * obs->T * errs.f);
        const double accelerr = props[ii].zerr * POST_SOL / (obs->T * obs->T * errs.f);
        write_val_with_err(f, accel, accelerr, 2, fw[10]);
        fprintf(f, "  %.20s\n", notes + ii * 20);
    }
    fprintf(f, "\n\n");
    free(notes);

    header_rows(f, ht1, ht2, hw, 13, obs->nph > 0.0 ? "NumPhot" : NULL);
    for (ii = 0; ii < numcands; ii++) {
        orc_optcand *cand = &cands[ii];
        for (jj = 0; jj < cand->numharm; jj++) {
            orc_fourierprops hp;
            orc_rderivs d = cand->derivs[jj];
            if (obs->nph > 0.0)
                d.locpow = obs->nph;
            if (orc_calc_props(&d, cand->hirs[jj], cand->hizs[jj], 0.0, &hp)) {
                fclose(f);
                free(props);
                return -1;
            }
            orc_calc_rzwerrs(&hp, obs->T, &errs);
            if (jj == 0)
                sprintf(tmpstr, " %-4d", ii + 1);
            else
                sprintf(tmpstr, "     ");
            center_string(ctrstr, tmpstr, hw[0]);
            fprintf(f, "%s  ", ctrstr);
            sprintf(tmpstr, "%-4d", jj + 1);
            center_string(ctrstr, tmpstr, hw[1]);
            fprintf(f, "%s  ", ctrstr);
            sprintf(tmpstr, "%.2f", orc_candidate_sigma(hp.pow, 1, 1));
            center_string(ctrstr, tmpstr, hw[2]);
            fprintf(f, "%s  ", ctrstr);
            write_val_with_err(f, hp.pow, hp.powerr, 2, hw[3]);
            sprintf(tmpstr, "%.3g", hp.rawpow);
            center_string(ctrstr, tmpstr, hw[4]);
            fprintf(f, "%s  ", ctrstr);
            write_val_with_err(f, hp.r, hp.rerr, 2, hw[5]);
            sprintf(tmpstr, "%.2f", cand->r * (jj + 1));
            center_string(ctrstr, tmpstr, hw[6]);
            fprintf(f, "%s  ", ctrstr);
            write_val_with_err(f, hp.z, hp.zerr, 2, hw[7]);
            sprintf(tmpstr, "%.2f", cand->z * (jj + 1));
            center_string(ctrstr, tmpstr, hw[8]);
            fprintf(f, "%s  ", ctrstr);
            write_val_with_err(f, hp.phs, hp.phserr, 2, hw[9]);
            write_val_with_err(f, hp.cen, hp.cenerr, 2, hw[10]);
            write_val_with_err(f, hp.pur, hp.purerr, 2, hw[11]);
            fprintf(f, "  %.20s\n", "                    ");
        }
    }
    fprintf(f, "\n\n");
    if (inftext)
        fputs(inftext, f);                                   /* the "cat root.inf >>" of :831-833 */
    fclose(f);
    f = fopen(candnm, "wb");                                  /* accelsearch.c:282-286 */
    if (!f) {
        free(props);
        return -1;
    }
    fwrite(props, sizeof(orc_fourierprops), (size_t) numcands, f);
    fclose(f);
    free(props);
    return 0;
}
