/*
 * accel_oracle.c -- CPU restatement of PRESTO's accelsearch hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see accel_oracle.h): the checker for the HIP
 * library, and the "port" CPU baseline of bench.py.  Plain C, float32 data,
 * double index/normalisation math exactly where the reference uses double.
 * Build with -ffp-contract=off so that float expressions round the way the
 * source says (the HIP kernels do the same for the values that feed the exact
 * block median).
 *
 * "parity unpinned" for the end-to-end path (no reference goldens exist and
 * the reference program needs FFTW, absent here); see the header for the
 * pieces that are pinned against oracle/_ref.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "accel_oracle.h"

#define ORC_PI 3.1415926535897932384626433832795028841971693993751
#define ORC_SQRT2 1.4142135623730950488016887242096980785696718753769
#define ORC_DBLCORRECT 1e-14 /* include/presto.h:20-22 */
#define ORC_NUMFINTBINS 16   /* include/presto.h:108   */

/* ------------------------------------------------------------------------
 * Fresnel integrals S(x), C(x): Cephes 2.1 rational approximations, as used
 * by the reference (src/fresnl.c:175-216; coefficient tables :77-172).
 * ---------------------------------------------------------------------- */
static double horner(double x, const double *c, int n) /* c[0] x^(n-1) + ... + c[n-1] */
{
    double acc = c[0];
    for (int k = 1; k < n; k++)
        acc = acc * x + c[k];
    return acc;
}

static double horner_monic(double x, const double *c, int n) /* x^n + c[0] x^(n-1) + ... */
{
    double acc = x + c[0];
    for (int k = 1; k < n; k++)
        acc = acc * x + c[k];
    return acc;
}

static const double FS_N[6] = { -2.99181919401019853726E3, 7.08840045257738576863E5,
    -6.29741486205862506537E7, 2.54890880573376359104E9, -4.42979518059697779103E10,
    3.18016297876567817986E11 };
static const double FS_D[6] = { 2.81376268889994315696E2, 4.55847810806532581675E4,
    5.17343888770096400730E6, 4.19320245898111231129E8, 2.24411795645340920940E10,
    6.07366389490084639049E11 };
static const double FC_N[6] = { -4.98843114573573548651E-8, 9.50428062829859605134E-6,
    -6.45191435683965050962E-4, 1.88843319396703850064E-2, -2.05525900955013891793E-1,
    9.99999999999999998822E-1 };
static const double FC_D[7] = { 3.99982968972495980367E-12, 9.15439215774657478799E-10,
    1.25001862479598821474E-7, 1.22262789024179030997E-5, 8.68029542941784300606E-4,
    4.12142090722199792936E-2, 1.00000000000000000118E0 };
static const double FF_N[10] = { 4.21543555043677546506E-1, 1.43407919780758885261E-1,
    1.15220955073585758835E-2, 3.45017939782574027900E-4, 4.63613749287867322088E-6,
    3.05568983790257605827E-8, 1.02304514164907233465E-10, 1.72010743268161828879E-13,
    1.34283276233062758925E-16, 3.76329711269987889006E-20 };
static const double FF_D[10] = { 7.51586398353378947175E-1, 1.16888925859191382142E-1,
    6.44051526508858611005E-3, 1.55934409164153020873E-4, 1.84627567348930545870E-6,
    1.12699224763999035261E-8, 3.60140029589371370404E-11, 5.88754533621578410010E-14,
    4.52001434074129701496E-17, 1.25443237090011264384E-20 };
static const double FG_N[11] = { 5.04442073643383265887E-1, 1.97102833525523411709E-1,
    1.87648584092575249293E-2, 6.84079380915393090172E-4, 1.15138826111884280931E-5,
    9.82852443688422223854E-8, 4.45344415861750144738E-10, 1.08268041139020870318E-12,
    1.37555460633261799868E-15, 8.36354435630677421531E-19, 1.86958710162783235106E-22 };
static const double FG_D[11] = { 1.47495759925128324529E0, 3.37748989120019970451E-1,
    2.53603741420338795122E-2, 8.14679107184306179049E-4, 1.27545075667729118702E-5,
    1.04314589657571990585E-7, 4.60680728146520428211E-10, 1.10273215066240270757E-12,
    1.38796531259578871258E-15, 8.39158816283118707363E-19, 1.86958710162783236342E-22 };

void orc_fresnel(double xin, double *S, double *C)
{
    const double x = fabs(xin);
    const double x2 = x * x;
    double ss, cc;

    if (x2 < 2.5625) {          /* power-series region */
        const double t = x2 * x2;
        ss = x * x2 * horner(t, FS_N, 6) / horner_monic(t, FS_D, 6);
        cc = x * horner(t, FC_N, 6) / horner(t, FC_D, 7);
    } else if (x > 36974.0) {   /* both integrals have converged to 1/2 */
        ss = cc = 0.5;
    } else {                    /* auxiliary functions f, g */
        double t = ORC_PI * x2;
        const double u = 1.0 / (t * t);
        t = 1.0 / t;
        const double f = 1.0 - u * horner(u, FF_N, 10) / horner_monic(u, FF_D, 10);
        const double g = t * horner(u, FG_N, 11) / horner_monic(u, FG_D, 11);
        const double arg = 0.5 * ORC_PI * x2;
        const double c = cos(arg), s = sin(arg);
        const double pix = ORC_PI * x;
        cc = 0.5 + (f * s - g * c) / pix;
        ss = 0.5 - (f * c + g * s) / pix;
    }
    if (xin < 0.0) {
        cc = -cc;
        ss = -ss;
    }
    *S = ss;
    *C = cc;
}

/* ------------------------------------------------------------------------
 * Index helpers (src/accel_utils.c:36, 53-84) and sizes.
 * ---------------------------------------------------------------------- */
static inline int nearest_int(double x) /* NEAREST_INT, src/accel_utils.c:36 */
{
    return (int) (x < 0 ? x - 0.5 : x + 0.5);
}

int orc_calc_required_z(double harm_fract, double zfull) /* :53-59 */
{
    return nearest_int(ORC_RDZ * zfull * harm_fract) * ORC_DZ;
}

double orc_calc_required_r(double harm_fract, double rfull) /* :62-68 */
{
    return (int) (ORC_RDR * rfull * harm_fract + 0.5) * ORC_DR;
}

int orc_index_from_z(double z, double loz) /* :79-84 */
{
    return (int) ((z - loz) * ORC_RDZ + ORC_DBLCORRECT);
}

/* z_resp_halfwidth(z, LOWACC): src/responses.c:29-65 (LOWACC branch :55-63). */
int orc_z_resp_halfwidth(double z)
{
    z = fabs(z);
    int m = (int) (long) (z * (0.00089 * z + 0.3131) + ORC_NUMFINTBINS);
    if (m < ORC_NUMFINTBINS)
        m = ORC_NUMFINTBINS;
    if (z > 100 && m > 0.6 * z)
        m = (int) (0.6 * z);
    return m;
}

long long orc_next2_to_n(long long x) /* src/misc_utils.c:269-277 */
{
    long long p = 1;
    while (p < x)
        p <<= 1;
    return p;
}

int orc_calc_fftlen(int numharm, int harmnum, int max_zfull) /* src/accel_utils.c:119-132 */
{
    const double harm_fract = (double) harmnum / (double) numharm;
    const int bins_needed = (ORC_USELEN * harmnum) / numharm + 2;
    const int end_effects = 2 * ORC_NUMBETWEEN *
        orc_z_resp_halfwidth(orc_calc_required_z(harm_fract, max_zfull));
    return (int) orc_next2_to_n(bins_needed + end_effects);
}

/* ------------------------------------------------------------------------
 * Exact order statistic arr[(n-1)/2] by quickselect; permutes arr.
 * Follows src/median.c:12-69 (median-of-three pivot, Hoare-style nibbling).
 * ---------------------------------------------------------------------- */
static inline void fswap(float *a, float *b) { float t = *a; *a = *b; *b = t; }

float orc_median(float *arr, int n)
{
    int lo = 0, hi = n - 1;
    const int k = (lo + hi) / 2;
    for (;;) {
        if (hi <= lo)
            return arr[k];
        if (hi == lo + 1) {
            if (arr[lo] > arr[hi])
                fswap(&arr[lo], &arr[hi]);
            return arr[k];
        }
        const int mid = (lo + hi) / 2;
        if (arr[mid] > arr[hi]) fswap(&arr[mid], &arr[hi]);
        if (arr[lo] > arr[hi])  fswap(&arr[lo], &arr[hi]);
        if (arr[mid] > arr[lo]) fswap(&arr[mid], &arr[lo]);
        fswap(&arr[mid], &arr[lo + 1]);
        int ll = lo + 1, hh = hi;
        for (;;) {
            do ll++; while (arr[lo] > arr[ll]);
            do hh--; while (arr[hh] > arr[lo]);
            if (hh < ll)
                break;
            fswap(&arr[ll], &arr[hh]);
        }
        fswap(&arr[lo], &arr[hh]);
        if (hh <= k) lo = ll;
        if (hh >= k) hi = hh - 1;
    }
}

/* ------------------------------------------------------------------------
 * Significance (src/characteristics.c:397-604).  The reference evaluates the
 * chi^2 and normal distributions with DCDFLIB (cdfchi/cdfnor, src/dcdflib.c).
 * Here: dof is always even (2*numharm), so the chi^2 survival function is the
 * closed-form Poisson sum Q(x|2k) = exp(-x/2) sum_{j<k} (x/2)^j / j!, and the
 * normal tail is erfc.  Inverses are refined to double round-off (DCDFLIB stops
 * its chi^2 inversion at a relative 1e-8, so the last float bit of powcut may
 * differ; tests compare both).
 * ---------------------------------------------------------------------- */
static double chi2_sf_even(double x, int dof) /* Q(x | dof), dof even */
{
    const int k = dof / 2;
    const double h = 0.5 * x;
    double term = 1.0, sum = 1.0;
    for (int j = 1; j < k; j++) {
        term *= h / j;
        sum += term;
    }
    return exp(-h) * sum;
}

static double log_asym_inc_gamma(double a, double z) /* A&S 6.5.32; :418-436 */
{
    double x = 1.0, part = 1.0, term = 1.0;
    int ii = 1;
    while (fabs(part) > 1e-15) {
        term *= (a - ii);
        part = term / pow(z, ii);
        x += part;
        ii += 1;
    }
    return (a - 1.0) * log(z) - z + log(x);
}

static double log_asym_gamma(double z) /* A&S 6.1.41; :438-453 */
{
    double x = (z - 0.5) * log(z) - z + 0.91893853320467267;
    const double y = 1.0 / (z * z);
    x += (((-5.9523809523809529e-4 * y + 7.9365079365079365079365e-4) * y
           - 2.7777777777777777777778e-3) * y + 8.3333333333333333333333e-2) / z;
    return x;
}

double orc_chi2_logp(double chi2, int dof) /* :494-528 */
{
    if (chi2 <= 0.0)
        return -INFINITY;
    if (chi2 / dof > 15.0 || (dof > 150 && chi2 / dof > 6.0))
        return log_asym_inc_gamma(0.5 * dof, 0.5 * chi2) - log_asym_gamma(0.5 * dof);
    return log(chi2_sf_even(chi2, dof));
}

static double norm_sf(double x) { return 0.5 * erfc(x / ORC_SQRT2); }

/* x with Q(x) = q, 0 < q < 1: Newton on log Q. */
static double norm_isf(double q)
{
    if (q == 0.5)
        return 0.0;
    const int flip = q > 0.5;
    const double qq = flip ? 1.0 - q : q;
    /* A&S 26.2.23 start */
    double t = sqrt(-2.0 * log(qq));
    double x = t - (2.515517 + t * (0.802853 + t * 0.010328)) /
        (1.0 + t * (1.432788 + t * (0.189269 + t * 0.001308)));
    const double lq = log(qq);
    for (int it = 0; it < 60; it++) {
        const double Q = norm_sf(x);
        const double pdf = exp(-0.5 * x * x) / sqrt(2.0 * ORC_PI);
        const double dx = (log(Q) - lq) * Q / pdf;
        x += dx;
        if (fabs(dx) <= 4e-16 * fmax(1.0, fabs(x)))
            break;
    }
    return flip ? -x : x;
}

static double ext_equiv_gauss_sigma(double logp) /* A&S 26.2.23; :397-415 */
{
    const double t = sqrt(-2.0 * logp);
    const double num = 2.515517 + t * (0.802853 + t * 0.010328);
    const double den = 1.0 + t * (1.432788 + t * (0.189269 + t * 0.001308));
    return t - num / den;
}

double orc_equivalent_gaussian_sigma(double logp) /* :456-491 */
{
    double x;
    if (logp < -600.0) {
        x = ext_equiv_gauss_sigma(logp);
    } else {
        const double q = exp(logp);
        const double p = 1.0 - q;
        if (p <= 0.0 || p > 1.0)       /* cdfnor status -2 -> 0.0   (:474-475) */
            x = 0.0;
        else if (q <= 0.0 || q > 1.0)  /* cdfnor status -3 -> 38.5  (:476-477) */
            x = 38.5;
        else
            x = norm_isf(q);
    }
    return x < 0.0 ? 0.0 : x;
}

double orc_candidate_sigma(double power, int numsum, double numtrials) /* :548-568 */
{
    if (power <= 0.0)
        return 0.0;
    double logp = orc_chi2_logp(2.0 * power, 2 * numsum);
    logp += log(numtrials);
    return orc_equivalent_gaussian_sigma(logp);
}

double orc_power_for_sigma(double sigma, int numsum, double numtrials) /* :571-604 */
{
    const double q = norm_sf(sigma) / numtrials;
    const int dof = 2 * numsum;
    /* Solve Q(x | dof) = q by bisection on log Q, then Newton polish. */
    const double lq = log(q);
    double lo = 0.0, hi = 2.0 * dof + 10.0;
    while (log(chi2_sf_even(hi, dof)) > lq)
        hi *= 2.0;
    for (int it = 0; it < 200; it++) {
        const double mid = 0.5 * (lo + hi);
        if (log(chi2_sf_even(mid, dof)) > lq)
            lo = mid;
        else
            hi = mid;
        if (hi - lo <= 1e-15 * hi)
            break;
    }
    return 0.5 * (0.5 * (lo + hi));
}

/* ------------------------------------------------------------------------
 * Complex float32 FFT (stands where the reference calls FFTW through
 * COMPLEXFFT/fftwcall, src/fftcalls.c:53-152, and fftwf_execute_dft,
 * src/accel_utils.c:1034).  isign=-1 forward, +1 inverse, both unnormalised
 * (include/ransomfft.h:61-63).  Stockham autosort, radix 4 with a trailing
 * radix 2, float arithmetic, twiddles rounded from double.
 * ---------------------------------------------------------------------- */
#define ORC_MAXPLANS 8
static struct { long n; orc_cf32 *w; } g_plans[ORC_MAXPLANS];
static int g_nplans = 0;

static const orc_cf32 *fft_twiddles(long n)
{
    const orc_cf32 *found = NULL;
#ifdef _OPENMP
#pragma omp critical(orc_fft_plan)
#endif
    {
        for (int i = 0; i < g_nplans; i++)
            if (g_plans[i].n == n)
                found = g_plans[i].w;
        if (!found) {
            orc_cf32 *w = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) n);
            for (long k = 0; k < n; k++) {
                const double a = -2.0 * ORC_PI * (double) k / (double) n;
                w[k].r = (float) cos(a);
                w[k].i = (float) sin(a);
            }
            if (g_nplans < ORC_MAXPLANS) {
                g_plans[g_nplans].n = n;
                g_plans[g_nplans].w = w;
                g_nplans++;
            }
            found = w; /* leaks only if more than ORC_MAXPLANS sizes are used */
        }
    }
    return found;
}

static inline orc_cf32 cmulf(orc_cf32 a, orc_cf32 b)
{
    orc_cf32 c = { a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r };
    return c;
}

static void fft_pow2(orc_cf32 *data, long n, int isign)
{
    if (n <= 1)
        return;
    const orc_cf32 *W = fft_twiddles(n);
    orc_cf32 *scratch = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) n);
    orc_cf32 *x = data, *y = scratch;
    const float sg = (isign < 0) ? -1.0f : 1.0f; /* sign of the twiddle exponent */
    long len = n, s = 1;

    while (len >= 4) {
        const long m = len / 4;
        for (long p = 0; p < m; p++) {
            orc_cf32 w1 = W[p * s], w2 = W[2 * p * s], w3 = W[3 * p * s];
            if (isign > 0) { w1.i = -w1.i; w2.i = -w2.i; w3.i = -w3.i; }
            const orc_cf32 *x0 = x + s * p, *x1 = x + s * (p + m);
            const orc_cf32 *x2 = x + s * (p + 2 * m), *x3 = x + s * (p + 3 * m);
            orc_cf32 *y0 = y + s * (4 * p);
            for (long q = 0; q < s; q++) {
                const orc_cf32 a = x0[q], b = x1[q], c = x2[q], d = x3[q];
                const orc_cf32 apc = { a.r + c.r, a.i + c.i };
                const orc_cf32 amc = { a.r - c.r, a.i - c.i };
                const orc_cf32 bpd = { b.r + d.r, b.i + d.i };
                /* sg * i * (b - d) */
                const orc_cf32 jbmd = { -sg * (b.i - d.i), sg * (b.r - d.r) };
                const orc_cf32 t1 = { amc.r + jbmd.r, amc.i + jbmd.i };
                const orc_cf32 t2 = { apc.r - bpd.r, apc.i - bpd.i };
                const orc_cf32 t3 = { amc.r - jbmd.r, amc.i - jbmd.i };
                y0[q].r = apc.r + bpd.r;
                y0[q].i = apc.i + bpd.i;
                y0[q + s] = cmulf(t1, w1);
                y0[q + 2 * s] = cmulf(t2, w2);
                y0[q + 3 * s] = cmulf(t3, w3);
            }
        }
        orc_cf32 *t = x; x = y; y = t;
        len = m;
        s *= 4;
    }
    if (len == 2) {
        for (long q = 0; q < s; q++) {
            const orc_cf32 a = x[q], b = x[q + s];
            y[q].r = a.r + b.r;       y[q].i = a.i + b.i;
            y[q + s].r = a.r - b.r;   y[q + s].i = a.i - b.i;
        }
        orc_cf32 *t = x; x = y; y = t;
    }
    if (x != data)
        memcpy(data, x, sizeof(orc_cf32) * (size_t) n);
    free(scratch);
}

/* ------------------------------------------------------------------------
 * Response templates (src/responses.c:165-231, 234-322).
 * ---------------------------------------------------------------------- */
void orc_gen_r_response(double roffset, int numbetween, int numkern, orc_cf32 *out)
{
    /* sinc interpolation kernel; cos/sin advanced by the same three-term
     * recurrence as the reference (:199-221) so round-off follows it. */
    const double startr = ORC_PI * (numkern / (double) (2 * numbetween) + roffset);
    const double delta = -ORC_PI / numbetween;
    double tmp = sin(0.5 * delta);
    const double alpha = -2.0 * tmp * tmp;
    const double beta = sin(delta);
    double c = cos(startr), s = sin(startr), r = startr;

    for (int ii = 0; ii < numkern; ii++, r += delta) {
        const double sinc = (r == 0.0) ? 1.0 : s / r;
        out[ii].r = (float) (c * sinc);
        out[ii].i = (float) (s * sinc);
        tmp = c;
        c = alpha * tmp - beta * s + c;
        s = alpha * s + beta * tmp + s;
    }
    if (roffset < 1E-3) { /* series about the removable singularity (:225-229) */
        const double r2 = roffset * roffset;
        out[numkern / 2].r = (float) (1 - 6.579736267392905746 * r2);
        out[numkern / 2].i = (float) (roffset * (ORC_PI - 10.335425560099940058 * r2));
    }
}

void orc_gen_z_response(double roffset, int numbetween, double z, int numkern, orc_cf32 *out)
{
    const double absz = fabs(z);
    if (absz < 1E-4) { /* :274-278 */
        orc_gen_r_response(roffset, numbetween, numkern, out);
        return;
    }
    double ipart;
    double startr = roffset - 0.5 * z;
    const double startroffset = (startr < 0) ? 1.0 + modf(startr, &ipart) : modf(startr, &ipart);
    const int signz = (z < 0.0) ? -1 : 1;
    const double zd = signz * ORC_SQRT2 / sqrt(absz);
    const double cons = zd / 2.0;
    const double pibyz = ORC_PI / z;
    const double delta = -1.0 / numbetween;
    startr += numkern / (double) (2 * numbetween);

    double r = startr;
    for (int ii = 0; ii < numkern; ii++, r += delta) {
        const double yy = r * zd;
        const double zz = yy + z * zd;
        const double xx = pibyz * r * r;
        const double c = cos(xx), s = sin(xx);
        double sy, cy, sz, cz;
        orc_fresnel(yy, &sy, &cy);
        orc_fresnel(zz, &sz, &cz);
        const double re = signz * (cz - cy);
        const double im = sy - sz;
        out[ii].r = (float) ((re * c - im * s) * cons);
        out[ii].i = (float) (-(re * s + im * c) * cons);
    }
    if (startroffset < 1E-3 && absz < 1E-3) { /* :309-320 (unreachable for |z| >= 1e-4 ... 1e-3 band only) */
        const double z2 = z * z, x2 = startroffset * startroffset;
        const int h = numkern / 2;
        double re = 1.0 - 0.16449340668482264365 * z2;
        double im = -0.5235987755982988731 * z;
        re += startroffset * 1.6449340668482264365 * z;
        im += startroffset * (ORC_PI - 0.5167712780049970029 * z2);
        re += x2 * (-6.579736267392905746 + 0.9277056288952613070 * z2);
        im += x2 * (3.1006276680299820175 * z);
        out[h].r = (float) re;
        out[h].i = (float) im;
    }
}

void orc_place_complex_kernel(const orc_cf32 *kernel, int numkernel, orc_cf32 *result,
                              int numresult) /* src/corr_prep.c:58-80 */
{
    const int half = numkernel / 2;
    memset(result, 0, sizeof(orc_cf32) * (size_t) numresult);
    memcpy(result, kernel + half, sizeof(orc_cf32) * (size_t) half);
    memcpy(result + numresult - half, kernel, sizeof(orc_cf32) * (size_t) half);
}

void orc_init_kernel(int z, int fftlen, orc_cf32 *out, int *halfwidth) /* src/accel_utils.c:135-151 */
{
    const int hw = orc_z_resp_halfwidth((double) z);
    const int numkern = 2 * ORC_NUMBETWEEN * hw;
    orc_cf32 *tmp = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) numkern);
    orc_gen_z_response(0.0, ORC_NUMBETWEEN, (double) z, numkern, tmp);
    orc_place_complex_kernel(tmp, numkern, out, fftlen);
    free(tmp);
    orc_fft(out, fftlen, -1);
    if (halfwidth)
        *halfwidth = hw;
}

/* Lengths that are not powers of two (time series of any even length): the
 * reference hands them to FFTW, whose results are the exact transform up to
 * float round-off.  Here: the chirp-z identity in double precision on top of a
 * radix-2 double FFT, rounded to float at the end. */
static void dfft_pow2(double *re, double *im, long n, int isign)
{
    for (long i = 1, j = 0; i < n; i++) {                     /* bit reversal */
        long bit = n >> 1;
        for (; j & bit; bit >>= 1)
            j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (long len = 2; len <= n; len <<= 1) {
        const long half = len >> 1;
        for (long k = 0; k < half; k++) {
            const double a = isign * 2.0 * ORC_PI * (double) k / (double) len;
            const double wr = cos(a), wi = sin(a);
            for (long i = k; i < n; i += len) {
                const long j = i + half;
                const double tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
                re[j] = re[i] - tr; im[j] = im[i] - ti;
                re[i] += tr;        im[i] += ti;
            }
        }
    }
}

static void fft_any(orc_cf32 *data, long n, int isign)
{
    long m = 1;
    while (m < 2 * n - 1)
        m <<= 1;
    double *ar = (double *) calloc((size_t) m, sizeof(double)), *ai = (double *) calloc((size_t) m, sizeof(double));
    double *br = (double *) calloc((size_t) m, sizeof(double)), *bi = (double *) calloc((size_t) m, sizeof(double));
    double *wr = (double *) malloc(sizeof(double) * (size_t) n), *wi = (double *) malloc(sizeof(double) * (size_t) n);
    for (long k = 0; k < n; k++) {                            /* w[k] = exp(isign*i*pi*k^2/n) */
        const long long k2 = ((long long) k * k) % (2 * n);
        const double a = isign * ORC_PI * (double) k2 / (double) n;
        wr[k] = cos(a);
        wi[k] = sin(a);
        ar[k] = data[k].r * wr[k] - data[k].i * wi[k];
        ai[k] = data[k].r * wi[k] + data[k].i * wr[k];
        br[k] = wr[k];
        bi[k] = -wi[k];
        if (k) {
            br[m - k] = wr[k];
            bi[m - k] = -wi[k];
        }
    }
    dfft_pow2(ar, ai, m, -1);
    dfft_pow2(br, bi, m, -1);
    for (long k = 0; k < m; k++) {
        const double tr = ar[k] * br[k] - ai[k] * bi[k];
        ai[k] = ar[k] * bi[k] + ai[k] * br[k];
        ar[k] = tr;
    }
    dfft_pow2(ar, ai, m, +1);
    for (long k = 0; k < n; k++) {
        const double cr = ar[k] / (double) m, ci = ai[k] / (double) m;
        data[k].r = (float) (cr * wr[k] - ci * wi[k]);
        data[k].i = (float) (cr * wi[k] + ci * wr[k]);
    }
    free(ar); free(ai); free(br); free(bi); free(wr); free(wi);
}

void orc_fft(orc_cf32 *data, long n, int isign)
{
    if (n <= 1)
        return;
    if ((n & (n - 1)) == 0)
        fft_pow2(data, n, isign);
    else
        fft_any(data, n, isign);
}

/* ------------------------------------------------------------------------
 * Search set-up: create_accelobs(), src/accel_utils.c:1524-1670.
 * flo/fhi always apply because the generated parser marks them present
 * (src/accelsearch_cmd.c:54-60 -> branches :1581-1613).
 * ---------------------------------------------------------------------- */
static int twon_to_index(int n) /* :42-51 */
{
    int x = 0;
    while (n > 1) { n >>= 1; x++; }
    return x;
}

int orc_obs_init(orc_obs *obs, const orc_cf32 *fft, long long numbins, long long N,
                 double dt, int zmax, int numharm, double sigma, double flo, double fhi,
                 long long lobin, int norm_type, int photon, int alloc_plane)
{
    memset(obs, 0, sizeof(*obs));
    obs->fft = fft;
    obs->numbins = numbins;
    obs->nthreads = 1;
    if (zmax % ORC_DZ)                                   /* :1526-1527 */
        zmax = (zmax / ORC_DZ + 1) * ORC_DZ;
    obs->N = N;
    obs->nph = photon ? fft[0].r : 0.0f;                 /* :1531-1539 */
    obs->norm_type = norm_type;
    obs->lobin = lobin;
    if (obs->lobin > 0) {                                /* :1559-1566 */
        obs->nph = 0.0f;
        if (lobin > numbins - 1)
            return -1;
    }
    if (numharm != 1 && numharm != 2 && numharm != 4 && numharm != 8 &&
        numharm != 16 && numharm != 32)                  /* :1567-1574 */
        return -2;
    obs->numharmstages = twon_to_index(numharm) + 1;
    obs->zmax = zmax;
    obs->numz = (zmax / ORC_DZ) * 2 + 1;
    obs->dt = dt;
    obs->T = dt * (double) N;
    obs->rlo = floor(flo * obs->T);                      /* :1581-1589 */
    if (obs->rlo < obs->lobin)
        obs->rlo = (double) obs->lobin;
    if (obs->rlo > numbins - 1)
        return -3;
    obs->highestbin = (long long) ceil(fhi * obs->T);    /* :1603-1613 */
    if (obs->highestbin > numbins - 1)
        obs->highestbin = numbins - 1;
    obs->rhi = (double) obs->highestbin;
    if (obs->highestbin < obs->rlo)
        return -4;
    obs->zhi = zmax;
    obs->zlo = -zmax;
    obs->sigma = (float) sigma;
    for (int ii = 0; ii < obs->numharmstages; ii++) {    /* :1631-1641 */
        if (obs->numz == 1)
            obs->numindep[ii] = (long long) ((obs->rhi - obs->rlo) / (1 << ii));
        else
            obs->numindep[ii] = (long long) ((obs->rhi - obs->rlo) * (obs->numz + 1) *
                                             ((double) ORC_DZ / 6.95) / (1 << ii));
        obs->powcut[ii] = (float) orc_power_for_sigma(obs->sigma, 1 << ii,
                                                      (double) obs->numindep[ii]);
    }
    obs->fftlen = orc_calc_fftlen(1, 1, zmax);
    obs->inmem = 1;
    obs->rlen = (obs->highestbin + ORC_USELEN) * ORC_RDR; /* :1081, :1658-1659 */
    if (alloc_plane) {
        /* The reference leaves the plane uninitialised (gen_fvect, :1664);
         * cells never written (r < 8) are defined as 0 here. */
        obs->ffdotplane = (float *) calloc((size_t) obs->rlen * (size_t) obs->numz, sizeof(float));
        if (!obs->ffdotplane)
            return -5;
    }
    return 0;
}

void orc_obs_free(orc_obs *obs)
{
    free(obs->ffdotplane);
    free(obs->kern);
    obs->ffdotplane = NULL;
    obs->kern = NULL;
}

void orc_build_kernels(orc_obs *obs) /* create_subharminfos, fundamental only (:181-193, 160-178) */
{
    const int numkern = (obs->zmax / ORC_DZ) * 2 + 1;
    free(obs->kern);
    obs->kern = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) numkern * (size_t) obs->fftlen);
    (void) fft_twiddles(obs->fftlen);
    for (int ii = 0; ii < numkern; ii++) {
        int hw;
        orc_init_kernel(-obs->zmax + ii * ORC_DZ, obs->fftlen,
                        obs->kern + (size_t) ii * obs->fftlen, &hw);
        if (ii == 0)
            obs->halfwidth = hw;
    }
}

/* get_fourier_amplitudes(), mmap branch: src/accel_utils.c:848-877 */
static void fetch_amplitudes(const orc_obs *obs, long long lobin, int numbins, orc_cf32 *out)
{
    long long offset = 0;
    if (lobin - obs->lobin < 0) {
        offset = llabs(lobin - obs->lobin);
        if (offset > numbins) offset = numbins;
        memset(out, 0, sizeof(orc_cf32) * (size_t) offset);
    }
    const long long firstbin = (lobin - obs->lobin) + offset;
    long long count = numbins - offset;
    if (firstbin + count > obs->numbins) {
        long long numpad = firstbin + count - obs->numbins;
        if (numpad > count) numpad = count;
        count -= numpad;
        memset(out + numbins - numpad, 0, sizeof(orc_cf32) * (size_t) numpad);
    }
    if (count > 0)
        memcpy(out + offset, obs->fft + firstbin, sizeof(orc_cf32) * (size_t) count);
}

/* Normalisation of one data window, src/accel_utils.c:942-977. */
static void normalise_window(const orc_obs *obs, orc_cf32 *data, int numdata, int nice_numdata)
{
    if (obs->nph > 0.0) {                                        /* :942-948 */
        const double norm = 1.0 / sqrt(obs->nph);
        for (int ii = 0; ii < numdata; ii++) {
            data[ii].r *= norm;
            data[ii].i *= norm;
        }
    } else if (obs->norm_type == 1) {                            /* :962-977 -locpow */
        /* corr_loc_pow(), src/corr_routines.c:309-366: mean power in a ring of
         * 10 + 10 bins at distance 6..15 (the FFT correlation there, scaled by
         * 2/n twice over an integer-truncated kernel height, is this sum times
         * (2/n)*((n/2)/20)); the first and last 15 values copy their neighbour;
         * then the bit-trick inverse square root, src/misc_utils.c:255-266. */
        const int n = nice_numdata, khw = 15;
        float *pw = (float *) malloc(sizeof(float) * (size_t) n);
        float *loc = (float *) malloc(sizeof(float) * (size_t) n);
        const float scale = (2.0f / (float) n) * (float) ((n / 2) / 20);
        for (int ii = 0; ii < n; ii++)
            pw[ii] = data[ii].r * data[ii].r + data[ii].i * data[ii].i;
        for (int ii = khw; ii < n - khw; ii++) {
            float acc = 0.0f;
            for (int k = -15; k <= -6; k++) acc += pw[ii + k];
            for (int k = 6; k <= 15; k++) acc += pw[ii + k];
            loc[ii] = acc * scale;
        }
        for (int ii = 0; ii < khw; ii++) {
            loc[ii] = loc[khw];
            loc[n - 1 - ii] = loc[n - 1 - khw];
        }
        for (int ii = 0; ii < numdata; ii++) {
            const float x = loc[ii];
            union { float f; int i; } u;
            u.f = x;
            u.i = 0x5f3759df - (u.i >> 1);
            const float y = u.f;
            const float nrm = y * (1.5f - 0.5f * x * y * y);
            data[ii].r *= nrm;
            data[ii].i *= nrm;
        }
        free(pw);
        free(loc);
    } else {                                                     /* :949-961 block median */
        float *pw = (float *) malloc(sizeof(float) * (size_t) numdata);
        for (int ii = 0; ii < numdata; ii++)
            pw[ii] = data[ii].r * data[ii].r + data[ii].i * data[ii].i;
        const double norm = 1.0 / sqrt(orc_median(pw, numdata) / log(2.0));
        free(pw);
        for (int ii = 0; ii < numdata; ii++) {
            data[ii].r *= norm;
            data[ii].i *= norm;
        }
    }
}

/* subharm_ffdot_plane(1, 1, startr, startr + USELEN*DR - DR): :879-1056 */
void orc_fund_block(const orc_obs *obs, double startr, float *powers, int *rlo_out)
{
    const double fullrhi = startr + ORC_USELEN * ORC_DR - ORC_DR;
    const double drlo = orc_calc_required_r(1.0, startr);
    const double drhi = orc_calc_required_r(1.0, fullrhi);
    const int rlo = (int) floor(drlo);
    const int numrs = ORC_USELEN;
    const int numzs = obs->numz;
    const int binoffset = obs->halfwidth;
    const int fftlen = obs->fftlen;
    const int lobin = rlo - binoffset;
    const int hibin = (int) ceil(drhi) + binoffset;
    const int numdata = hibin - lobin + 1;
    const int nice_numdata = (int) orc_next2_to_n(numdata);

    if (nice_numdata != fftlen / ORC_NUMBETWEEN)
        printf("WARNING!!:  nice_numdata != fftlen/2 in orc_fund_block()!\n");
    orc_cf32 *data = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) nice_numdata);
    fetch_amplitudes(obs, lobin, nice_numdata, data);

    normalise_window(obs, data, numdata, nice_numdata);

    /* spread_no_pad + forward FFT: :980-983, src/corr_prep.c:3-40 */
    orc_cf32 *pdata = (orc_cf32 *) calloc((size_t) fftlen, sizeof(orc_cf32));
    {
        int numtoplace = fftlen / ORC_NUMBETWEEN;
        if (numtoplace > nice_numdata) numtoplace = nice_numdata;
        for (int ii = 0; ii < numtoplace; ii++)
            pdata[ii * ORC_NUMBETWEEN] = data[ii];
    }
    orc_fft(pdata, fftlen, -1);

    /* correlate with every template: :1002-1051 */
    const float norm = 1.0 / ((double) fftlen * (double) fftlen);
    const int offset = binoffset * ORC_NUMBETWEEN;
#ifdef _OPENMP
#pragma omp parallel num_threads(obs->nthreads)
#endif
    {
        orc_cf32 *tmp = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) fftlen);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int zi = 0; zi < numzs; zi++) {
            const orc_cf32 *k = obs->kern + (size_t) zi * fftlen;
            for (int jj = 0; jj < fftlen; jj++) {
                const float dr = pdata[jj].r, di = pdata[jj].i;
                const float kr = k[jj].r, ki = k[jj].i;
                tmp[jj].r = dr * kr + di * ki;
                tmp[jj].i = di * kr - dr * ki;
            }
            orc_fft(tmp, fftlen, +1);
            float *out = powers + (size_t) zi * numrs;
            for (int jj = 0; jj < numrs; jj++) {
                const orc_cf32 v = tmp[jj + offset];
                out[jj] = (v.r * v.r + v.i * v.i) * norm;
            }
        }
        free(tmp);
    }
    free(data);
    free(pdata);
    *rlo_out = rlo;
}

void orc_fund_to_ffdotplane(orc_obs *obs, const float *powers, int rlo) /* :1076-1090 */
{
    for (int ii = 0; ii < obs->numz; ii++) {
        float *dst = obs->ffdotplane + (size_t) ii * (size_t) obs->rlen + (size_t) rlo * ORC_RDR;
        memcpy(dst, powers + (size_t) ii * ORC_USELEN, sizeof(float) * ORC_USELEN);
    }
}

void orc_inmem_add_ffdotpows(const orc_obs *obs, float *powers, int rlo, int numharm,
                             int harmnum) /* :1160-1208 */
{
    const int numrs = ORC_USELEN, numzs = obs->numz;
    const double harm_fract = (double) harmnum / (double) numharm;
    const int zlo = (int) obs->zlo;
    int *indices = (int *) malloc(sizeof(int) * numrs);
    for (int ii = 0, rrint = ORC_RDR * rlo; ii < numrs; ii++, rrint++)
        indices[ii] = (int) (rrint * harm_fract + 0.5);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(obs->nthreads)
#endif
    for (int ii = 0; ii < numzs; ii++) {
        const int zz = zlo + ii * ORC_DZ;
        const int subz = orc_calc_required_z(harm_fract, zz);
        const int zind = orc_index_from_z(subz, zlo);
        const float *in = obs->ffdotplane + (size_t) zind * (size_t) obs->rlen;
        float *out = powers + (size_t) ii * numrs;
        for (int jj = 0; jj < numrs; jj++)
            out[jj] += in[indices[jj]];
    }
    free(indices);
}

/* ------------------------------------------------------------------------
 * Candidate list: insert_new_accelcand(), src/accel_utils.c:301-381, on a
 * plain singly linked list (the reference uses glib's GSList).
 * ---------------------------------------------------------------------- */
typedef struct cnode { orc_cand c; struct cnode *next; } cnode;

static cnode *new_node(float power, float sigma, int numharm, double rr, double zz)
{
    cnode *n = (cnode *) malloc(sizeof(cnode));
    n->c.power = power; n->c.sigma = sigma; n->c.numharm = numharm;
    n->c.r = rr; n->c.z = zz; n->next = NULL;
    return n;
}

static void set_node(cnode *n, float power, float sigma, int numharm, double rr, double zz)
{
    n->c.power = power; n->c.sigma = sigma; n->c.numharm = numharm;
    n->c.r = rr; n->c.z = zz;
}

static cnode *list_unlink(cnode *head, cnode *victim)
{
    if (head == victim) {
        head = victim->next;
    } else {
        cnode *p = head;
        while (p && p->next != victim) p = p->next;
        if (p) p->next = victim->next;
    }
    free(victim);
    return head;
}

static cnode *insert_cand(cnode *list, float power, float sigma, int numharm, double rr,
                          double zz, int *added)
{
    *added = 0;
    if (!list) {
        *added = 1;
        return new_node(power, sigma, numharm, rr, zz);
    }
    cnode *cur = list, *prev = NULL;
    while (cur->next && cur->c.r < rr) {
        prev = cur;
        cur = cur->next;
    }
    const double next_diff = fabs(rr - cur->c.r);
    const double prev_diff = prev ? fabs(rr - prev->c.r) : ORC_CLOSEST_R + 1.0;

    if (prev_diff < ORC_CLOSEST_R) {
        if (prev->c.sigma < sigma) {
            set_node(prev, power, sigma, numharm, rr, zz);
            *added = 1;
        }
        if (next_diff < ORC_CLOSEST_R && cur->c.sigma < sigma) {
            if (*added) {
                list = list_unlink(list, cur);
            } else {
                set_node(cur, power, sigma, numharm, rr, zz);
                *added = 1;
            }
        }
    } else if (next_diff < ORC_CLOSEST_R) {
        if (cur->c.sigma < sigma) {
            set_node(cur, power, sigma, numharm, rr, zz);
            *added = 1;
        }
    } else {
        cnode *n = new_node(power, sigma, numharm, rr, zz);
        *added = 1;
        if (!cur->next && cur->c.r < (rr - ORC_CLOSEST_R)) {
            cur->next = n;
            return list;
        }
        if (prev) {
            prev->next = n;
            n->next = cur;
        } else {
            n->next = list;
            return n;
        }
    }
    return list;
}

typedef struct {
    orc_hit *hits; long long max_hits, n_hits;
    cnode *list;
} search_state;

/* search_ffdotpows(), :1259-1299 (serial order: z rows ascending, then r). */
static void search_block(const orc_obs *obs, const float *powers, int rlo, int numharm,
                         int block, search_state *st)
{
    const int stage = twon_to_index(numharm);
    const float powcut = obs->powcut[stage];
    const long long numindep = obs->numindep[stage];
    const int zlo = (int) obs->zlo;
    for (int ii = 0; ii < obs->numz; ii++) {
        const float *row = powers + (size_t) ii * ORC_USELEN;
        for (int jj = 0; jj < ORC_USELEN; jj++) {
            if (row[jj] > powcut) {
                const float pow = row[jj];
                const float sig = (float) orc_candidate_sigma(pow, numharm, (double) numindep);
                const double rr = (rlo + jj * (double) ORC_DR) / (double) numharm;
                const double zz = (zlo + ii * (double) ORC_DZ) / (double) numharm;
                int added = 0;
                st->list = insert_cand(st->list, pow, sig, numharm, rr, zz, &added);
                if (st->n_hits < st->max_hits) {
                    orc_hit *h = &st->hits[st->n_hits];
                    h->power = pow; h->sigma = sig; h->numharm = numharm;
                    h->block = block; h->zind = ii; h->added = added;
                    h->rind = (long long) rlo * ORC_RDR + jj;
                    h->r = rr; h->z = zz;
                }
                st->n_hits++;
            }
        }
    }
}

int orc_num_search_blocks(const orc_obs *obs) /* src/accelsearch.c:165-168 */
{
    int n = 0;
    double startr = obs->rlo;
    while (startr + ORC_USELEN * ORC_DR < obs->highestbin) {
        n++;
        startr += ORC_USELEN * ORC_DR;
    }
    return n;
}

int orc_num_prepop_blocks(const orc_obs *obs) /* src/accelsearch.c:143-161 */
{
    int n = 0;
    double startr = 8;
    while (startr < obs->rlo) {
        n++;
        startr += ORC_USELEN * ORC_DR;
    }
    return n;
}

static void prepopulate(orc_obs *obs, float *powers) /* src/accelsearch.c:143-161 */
{
    double startr = 8;
    while (startr < obs->rlo) {
        int rlo;
        orc_fund_block(obs, startr, powers, &rlo);
        orc_fund_to_ffdotplane(obs, powers, rlo);
        startr += ORC_USELEN * ORC_DR;
    }
}

void orc_fill_plane(orc_obs *obs)
{
    float *powers = (float *) malloc(sizeof(float) * (size_t) obs->numz * ORC_USELEN);
    prepopulate(obs, powers);
    double startr = obs->rlo;
    while (startr + ORC_USELEN * ORC_DR < obs->highestbin) {
        int rlo;
        orc_fund_block(obs, startr, powers, &rlo);
        orc_fund_to_ffdotplane(obs, powers, rlo);
        startr += ORC_USELEN * ORC_DR;
    }
    free(powers);
}

static int drain_list(search_state *st, orc_cand *cands, int max_cands, int *n_cands)
{
    int n = 0;
    for (cnode *p = st->list; p;) {
        cnode *nx = p->next;
        if (cands && n < max_cands)
            cands[n] = p->c;
        n++;
        free(p);
        p = nx;
    }
    st->list = NULL;
    if (n_cands)
        *n_cands = n;
    return n;
}

/* Main loop, src/accelsearch.c:136-221, in-memory branch.  from_plane != 0:
 * take the fundamental block from obs->ffdotplane instead of recomputing it
 * (the plane holds exactly those powers once filled). */
static int run_loop(orc_obs *obs, int from_plane, orc_hit *hits, long long max_hits,
                    long long *n_hits, orc_cand *cands, int max_cands, int *n_cands)
{
    search_state st = { hits, max_hits, 0, NULL };
    float *powers = (float *) malloc(sizeof(float) * (size_t) obs->numz * ORC_USELEN);
    if (!from_plane)
        prepopulate(obs, powers);
    double startr = obs->rlo;
    int block = 0;
    while (startr + ORC_USELEN * ORC_DR < obs->highestbin) {
        int rlo;
        if (from_plane) {
            rlo = (int) floor(orc_calc_required_r(1.0, startr));
            for (int ii = 0; ii < obs->numz; ii++)
                memcpy(powers + (size_t) ii * ORC_USELEN,
                       obs->ffdotplane + (size_t) ii * (size_t) obs->rlen + (size_t) rlo * ORC_RDR,
                       sizeof(float) * ORC_USELEN);
        } else {
            orc_fund_block(obs, startr, powers, &rlo);
        }
        search_block(obs, powers, rlo, 1, block, &st);
        if (obs->numharmstages > 1) {
            if (!from_plane)
                orc_fund_to_ffdotplane(obs, powers, rlo);
            for (int stage = 1; stage < obs->numharmstages; stage++) {
                const int harmtosum = 1 << stage;
                for (int harm = 1; harm < harmtosum; harm += 2)
                    orc_inmem_add_ffdotpows(obs, powers, rlo, harmtosum, harm);
                search_block(obs, powers, rlo, harmtosum, block, &st);
            }
        }
        startr += ORC_USELEN * ORC_DR;
        block++;
    }
    free(powers);
    *n_hits = st.n_hits;
    drain_list(&st, cands, max_cands, n_cands);
    return 0;
}

int orc_run(orc_obs *obs, orc_hit *hits, long long max_hits, long long *n_hits,
            orc_cand *cands, int max_cands, int *n_cands)
{
    return run_loop(obs, 0, hits, max_hits, n_hits, cands, max_cands, n_cands);
}

int orc_search_plane(orc_obs *obs, orc_hit *hits, long long max_hits, long long *n_hits,
                     orc_cand *cands, int max_cands, int *n_cands)
{
    return run_loop(obs, 1, hits, max_hits, n_hits, cands, max_cands, n_cands);
}

/* ------------------------------------------------------------------------
 * Standard (not in-memory) mode: every sub-harmonic gets its own F-Fdot block
 * from its own data window, its own median and its own kernel bank, and is added
 * through the rinds/zinds look-up.  subharm_ffdot_plane() for any harmonic
 * (src/accel_utils.c:879-1056), init_subharminfo() (:160-178), add_ffdotpows()
 * (:1119-1134), and the non-inmem branch of the main loop
 * (src/accelsearch.c:199-210).
 * ---------------------------------------------------------------------- */
typedef struct {
    int numharm, harmnum, zmax, numkern, fftlen, halfwidth0;
    orc_cf32 *kern;
} sub_bank;

#define ORC_MAXBANKS 64
static sub_bank g_banks[ORC_MAXBANKS];
static int g_nbanks = 0;

static const sub_bank *get_bank(int numharm, int harmnum, int zmax_full)
{
    const double f = (double) harmnum / (double) numharm;
    const int zmax = orc_calc_required_z(f, zmax_full);
    const int fftlen = orc_calc_fftlen(numharm, harmnum, zmax_full);
    for (int i = 0; i < g_nbanks; i++)
        if (g_banks[i].numharm == numharm && g_banks[i].harmnum == harmnum &&
            g_banks[i].zmax == zmax && g_banks[i].fftlen == fftlen)
            return &g_banks[i];
    if (g_nbanks == ORC_MAXBANKS)
        g_nbanks = 0; /* recycle (leaks old kernels; test infrastructure) */
    sub_bank *b = &g_banks[g_nbanks++];
    b->numharm = numharm; b->harmnum = harmnum; b->zmax = zmax; b->fftlen = fftlen;
    b->numkern = (zmax / ORC_DZ) * 2 + 1;
    b->kern = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) b->numkern * (size_t) fftlen);
    for (int ii = 0; ii < b->numkern; ii++) {
        int hw;
        orc_init_kernel(-zmax + ii * ORC_DZ, fftlen, b->kern + (size_t) ii * fftlen, &hw);
        if (ii == 0)
            b->halfwidth0 = hw;
    }
    return b;
}

/* powers must hold numzs*numrs floats (at most numz * (USELEN+2)); rinds USELEN shorts. */
void orc_subharm_block(const orc_obs *obs, int numharm, int harmnum, double fullrlo,
                       double fullrhi, float *powers, int *rlo_out, int *zlo_out, int *numrs_out,
                       int *numzs_out, unsigned short *rinds)
{
    const sub_bank *bank = get_bank(numharm, harmnum, obs->zmax);
    const double harm_fract = (double) harmnum / (double) numharm;
    const double drlo = orc_calc_required_r(harm_fract, fullrlo);
    const double drhi = orc_calc_required_r(harm_fract, fullrhi);
    const int rlo = (int) floor(drlo);
    const int zlo = orc_calc_required_z(harm_fract, obs->zlo);
    if (numharm > 1 && rinds) {                               /* :910-917 */
        for (int ii = 0; ii < ORC_USELEN; ii++) {
            const double rr = fullrlo + ii * ORC_DR;
            const double subr = orc_calc_required_r(harm_fract, rr);
            rinds[ii] = (unsigned short) (int) ((subr - rlo) * ORC_RDR + ORC_DBLCORRECT);
        }
    }
    int numrs = (int) ((ceil(drhi) - floor(drlo)) * ORC_RDR + ORC_DBLCORRECT) + 1; /* :919-927 */
    if (numharm == 1 && harmnum == 1)
        numrs = ORC_USELEN;
    else if (numrs % ORC_RDR)
        numrs = (numrs / ORC_RDR + 1) * ORC_RDR;
    const int numzs = bank->numkern;
    const int binoffset = bank->halfwidth0;
    const int fftlen = bank->fftlen;
    const int lobin = rlo - binoffset;
    const int hibin = (int) ceil(drhi) + binoffset;
    const int numdata = hibin - lobin + 1;
    const int nice_numdata = (int) orc_next2_to_n(numdata);
    if (nice_numdata != fftlen / ORC_NUMBETWEEN)
        printf("WARNING!!:  nice_numdata != fftlen/2 in orc_subharm_block()!\n");
    orc_cf32 *data = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) nice_numdata);
    fetch_amplitudes(obs, lobin, nice_numdata, data);
    normalise_window(obs, data, numdata, nice_numdata);
    orc_cf32 *pdata = (orc_cf32 *) calloc((size_t) fftlen, sizeof(orc_cf32));
    {
        int numtoplace = fftlen / ORC_NUMBETWEEN;
        if (numtoplace > nice_numdata) numtoplace = nice_numdata;
        for (int ii = 0; ii < numtoplace; ii++)
            pdata[ii * ORC_NUMBETWEEN] = data[ii];
    }
    orc_fft(pdata, fftlen, -1);
    const float norm = 1.0 / ((double) fftlen * (double) fftlen);
    const int offset = binoffset * ORC_NUMBETWEEN;
    orc_cf32 *tmp = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) fftlen);
    for (int zi = 0; zi < numzs; zi++) {
        const orc_cf32 *k = bank->kern + (size_t) zi * fftlen;
        for (int jj = 0; jj < fftlen; jj++) {
            const float dr = pdata[jj].r, di = pdata[jj].i, kr = k[jj].r, ki = k[jj].i;
            tmp[jj].r = dr * kr + di * ki;
            tmp[jj].i = di * kr - dr * ki;
        }
        orc_fft(tmp, fftlen, +1);
        float *out = powers + (size_t) zi * numrs;
        for (int jj = 0; jj < numrs; jj++) {
            const orc_cf32 v = tmp[jj + offset];
            out[jj] = (v.r * v.r + v.i * v.i) * norm;
        }
    }
    free(tmp); free(data); free(pdata);
    *rlo_out = rlo; *zlo_out = zlo; *numrs_out = numrs; *numzs_out = numzs;
}

/* add_ffdotpows(), :1119-1134 */
void orc_add_ffdotpows(const orc_obs *obs, float *fund, const float *sub, int sub_zlo,
                       int sub_numrs, const unsigned short *rinds, int numharm, int harmnum)
{
    const double harm_fract = (double) harmnum / (double) numharm;
    const int zlo = (int) obs->zlo;
    for (int ii = 0; ii < obs->numz; ii++) {
        const int zz = zlo + ii * ORC_DZ;
        const int subz = orc_calc_required_z(harm_fract, zz);
        const int zind = orc_index_from_z(subz, sub_zlo);
        float *out = fund + (size_t) ii * ORC_USELEN;
        const float *in = sub + (size_t) zind * sub_numrs;
        for (int jj = 0; jj < ORC_USELEN; jj++)
            out[jj] += in[rinds[jj]];
    }
}

/* Main loop, standard branch (src/accelsearch.c:165-218 with obs.inmem == 0). */
int orc_run_standard(orc_obs *obs, orc_hit *hits, long long max_hits, long long *n_hits,
                     orc_cand *cands, int max_cands, int *n_cands)
{
    search_state st = { hits, max_hits, 0, NULL };
    float *powers = (float *) malloc(sizeof(float) * (size_t) obs->numz * ORC_USELEN);
    float *sub = (float *) malloc(sizeof(float) * (size_t) obs->numz * (ORC_USELEN + 2));
    unsigned short *rinds = (unsigned short *) malloc(sizeof(unsigned short) * ORC_USELEN);
    double startr = obs->rlo;
    int block = 0;
    while (startr + ORC_USELEN * ORC_DR < obs->highestbin) {
        const double lastr = startr + ORC_USELEN * ORC_DR - ORC_DR;
        int rlo, zlo, numrs, numzs;
        orc_subharm_block(obs, 1, 1, startr, lastr, powers, &rlo, &zlo, &numrs, &numzs, NULL);
        search_block(obs, powers, rlo, 1, block, &st);
        for (int stage = 1; stage < obs->numharmstages; stage++) {
            const int harmtosum = 1 << stage;
            for (int harm = 1; harm < harmtosum; harm += 2) {
                int srlo, szlo, snumrs, snumzs;
                orc_subharm_block(obs, harmtosum, harm, startr, lastr, sub, &srlo, &szlo, &snumrs,
                                  &snumzs, rinds);
                orc_add_ffdotpows(obs, powers, sub, szlo, snumrs, rinds, harmtosum, harm);
            }
            search_block(obs, powers, rlo, harmtosum, block, &st);
        }
        startr += ORC_USELEN * ORC_DR;
        block++;
    }
    free(powers); free(sub); free(rinds);
    *n_hits = st.n_hits;
    drain_list(&st, cands, max_cands, n_cands);
    return 0;
}

/* ------------------------------------------------------------------------
 * Time-series input: realfft() (src/fastffts.c:198-270, forward direction) and
 * deredden() (src/accel_utils.c:1301-1374), the two steps create_accelobs()
 * applies to a .dat file (:1429-1484).
 * ---------------------------------------------------------------------- */
void orc_realfft_forward(float *idata, long n)
{
    const long nby2 = n >> 1;
    orc_cf32 *data = (orc_cf32 *) idata;
    const double cc = -0.5, theta = -6.2831853071795864769252867665590057683943387987502 / (double) n;
    orc_fft(data, nby2, -1);
    double wr = cos(theta), wi = sin(theta), tmp1 = sin(0.5 * theta);
    const double wpr = -2.0 * tmp1 * tmp1, wpi = wi;
    long il = 1, ih = nby2 - il;
    for (; il <= (n >> 2); il++, ih--) {
        const double h1r = 0.5 * (data[il].r + data[ih].r);
        const double h1i = 0.5 * (data[il].i - data[ih].i);
        const double h2r = -cc * (data[il].i + data[ih].i);
        const double h2i = cc * (data[il].r - data[ih].r);
        const double h2rwr = h2r * wr, h2rwi = h2r * wi, h2iwr = h2i * wr, h2iwi = h2i * wi;
        data[il].r = h1r + h2rwr - h2iwi;
        data[il].i = h1i + h2iwr + h2rwi;
        data[ih].r = h1r - h2rwr + h2iwi;
        data[ih].i = -h1i + h2iwr + h2rwi;
        tmp1 = wr;
        wr = tmp1 * wpr - wi * wpi + wr;
        wi = wi * wpr + tmp1 * wpi + wi;
    }
    tmp1 = data[0].r;
    data[0].r = tmp1 + data[0].i;
    data[0].i = tmp1 - data[0].i;
}

/* invsqrtf(), src/misc_utils.c:255-266 */
static float orc_invsqrtf(float x)
{
    union { float f; int i; } u;
    u.f = x;
    u.i = 0x5f3759df - (u.i >> 1);
    const float y = u.f;
    return y * (1.5f - 0.5f * x * y * y);
}

void orc_deredden(orc_cf32 *fft, int numamps)
{
    int ii, ind;
    const int initialbuflen = 6, maxbuflen = 200;
    int buflen, lastbuflen, binnum = 1, numwrote = 1;
    float mean_old, mean_new, dslope = 1.0, norm;

    fft[0].r = 1.0;
    fft[0].i = 0.0;
    float *powbuf = (float *) malloc(sizeof(float) * (size_t) numamps);
    for (ii = 0; ii < numamps; ii++)
        powbuf[ii] = fft[ii].r * fft[ii].r + fft[ii].i * fft[ii].i;
    buflen = initialbuflen;
    mean_old = orc_median(powbuf + binnum, buflen) / log(2.0);
    norm = orc_invsqrtf(mean_old);
    for (ind = numwrote; ind < binnum + buflen / 2; ind++) {
        fft[ind].r *= norm;
        fft[ind].i *= norm;
    }
    numwrote += buflen / 2;
    binnum += buflen;
    lastbuflen = buflen;
    buflen = initialbuflen * log(binnum);
    if (buflen > maxbuflen)
        buflen = maxbuflen;
    while (binnum + buflen < numamps) {
        mean_new = orc_median(powbuf + binnum, buflen) / log(2.0);
        dslope = (mean_new - mean_old) / (0.5 * (lastbuflen + buflen));
        for (ii = 0, ind = numwrote; ind < binnum + buflen / 2; ii++, ind++) {
            norm = orc_invsqrtf(mean_old + dslope * ii);
            fft[ind].r *= norm;
            fft[ind].i *= norm;
        }
        numwrote += ii;
        binnum += buflen;
        lastbuflen = buflen;
        mean_old = mean_new;
        buflen = initialbuflen * log(binnum);
        if (buflen > maxbuflen)
            buflen = maxbuflen;
    }
    for (ii = 0, ind = numwrote; ind < numamps; ii++, ind++) {
        norm = orc_invsqrtf(mean_old + dslope * ii);
        fft[ind].r *= norm;
        fft[ind].i *= norm;
    }
    free(powbuf);
}

/* The stand-alone rednoise filter (src/rednoise.c:57-188) on an in-memory
 * spectrum: same block walk as the program's read loop, output in place.  The
 * program writes the imaginary parts of its last half block from the real
 * parts (:176); so does this. */
void orc_rednoise(orc_cf32 *fft, long numamps, int startwidth, int endwidth, double endfreq,
                  double Tsec)
{
    const float T = (float) Tsec;
    long binnum = 1, pos;
    int bufflen = startwidth, nblk_old, nblk_new, mid_old, mid_new, ii, ind;
    float mean_old, mean_new, dslope = 1.0, norm;
    float *powbuf = (float *) malloc(sizeof(float) * 4096);
    orc_cf32 *out = (orc_cf32 *) malloc(sizeof(orc_cf32) * (size_t) numamps);
    long numwrote = 0, old_start;

    out[0].r = 1.0; out[0].i = 0.0;
    numwrote = 1;
    pos = 1;
    nblk_old = bufflen;
    old_start = pos;
    pos += nblk_old;
    mid_old = nblk_old / 2;
    for (ii = 0; ii < nblk_old; ii++)
        powbuf[ii] = fft[old_start + ii].r * fft[old_start + ii].r + fft[old_start + ii].i * fft[old_start + ii].i;
    mean_old = orc_median(powbuf, nblk_old) / log(2.0);
    norm = orc_invsqrtf(mean_old);
    for (ii = 0; ii < mid_old; ii++) {
        out[numwrote + ii].r = fft[old_start + ii].r * norm;
        out[numwrote + ii].i = fft[old_start + ii].i * norm;
    }
    numwrote += mid_old;
    binnum += nblk_old;
    bufflen = startwidth * log(binnum);
    for (;;) {
        nblk_new = bufflen;
        if (pos + nblk_new > numamps)
            nblk_new = (int) (numamps - pos);
        if (nblk_new <= 0)
            break;
        const long new_start = pos;
        pos += nblk_new;
        mid_new = nblk_new / 2;
        for (ii = 0; ii < nblk_new; ii++)
            powbuf[ii] = fft[new_start + ii].r * fft[new_start + ii].r + fft[new_start + ii].i * fft[new_start + ii].i;
        mean_new = orc_median(powbuf, nblk_new) / log(2.0);
        dslope = (mean_new - mean_old) / (0.5 * (nblk_old + nblk_new));
        for (ii = 0, ind = mid_old; ind < nblk_old; ii++, ind++) {
            norm = orc_invsqrtf(mean_old + dslope * ii);
            out[numwrote + ii].r = fft[old_start + ind].r * norm;
            out[numwrote + ii].i = fft[old_start + ind].i * norm;
        }
        for (ind = 0; ind < mid_new; ii++, ind++) {
            norm = orc_invsqrtf(mean_old + dslope * ii);
            out[numwrote + ii].r = fft[new_start + ind].r * norm;
            out[numwrote + ii].i = fft[new_start + ind].i * norm;
        }
        numwrote += ii;
        binnum += nblk_new;
        if ((float) binnum / T < endfreq)
            bufflen = startwidth * log(binnum);
        else
            bufflen = endwidth;
        old_start = new_start;
        nblk_old = nblk_new;
        mean_old = mean_new;
        mid_old = mid_new;
    }
    for (ii = 0, ind = mid_old; ind < nblk_old; ii++, ind++) {
        norm = orc_invsqrtf(mean_old + dslope * ii);
        out[numwrote + ii].r = fft[old_start + ind].r * norm;
        out[numwrote + ii].i = fft[old_start + ind].r * norm;
    }
    numwrote += nblk_old - mid_old;
    memcpy(fft, out, sizeof(orc_cf32) * (size_t) numwrote);
    free(out);
    free(powbuf);
}

/* zapbirds(), src/zapping.c:29-81, applied to an in-memory spectrum the way the
 * zapbirds program applies it to the file (src/zapbirds.c:293-300): birds in
 * order of increasing lobin, each one seeing what the earlier ones wrote.  The
 * two 200-bin median windows start at (int)(lobin - 200) - 100 and (int)hibin - 100
 * (get_rawbins(), :19-27); bins outside the spectrum read as zero. */
#define ORC_MEDIANBINS 200
static float zap_window_median(const orc_cf32 *fft, long numbins, long start)
{
    float pw[ORC_MEDIANBINS];
    for (int ii = 0; ii < ORC_MEDIANBINS; ii++) {
        const long b = start + ii;
        pw[ii] = (b >= 0 && b < numbins) ? fft[b].r * fft[b].r + fft[b].i * fft[b].i : 0.0f;
    }
    return orc_median(pw, ORC_MEDIANBINS);
}

void orc_zapbirds(orc_cf32 *fft, long numbins, const double *lobins, const double *hibins, int n)
{
    const double top = (double) numbins;                     /* idata.N / 2 */
    for (int k = 0; k < n; k++) {
        const double lobin = lobins[k];
        double hibin = hibins[k];
        if (lobin >= top - 1)
            break;
        if (hibin >= top)
            hibin = top - 1;
        const int ilobin = (int) floor(lobin), ihibin = (int) ceil(hibin);
        const int binstozap = ihibin - ilobin;
        if (!(lobin - 1.5 * ORC_MEDIANBINS > 1))
            continue;
        const float median_lo = zap_window_median(fft, numbins, (long) (int) (lobin - ORC_MEDIANBINS) - ORC_MEDIANBINS / 2);
        const float median_hi = zap_window_median(fft, numbins, (long) (int) hibin - ORC_MEDIANBINS / 2);
        const float avgamp = sqrt(0.5 * (median_lo + median_hi) / -log(0.5));
        for (int ii = 0; ii < binstozap; ii++) {
            const long b = (long) ilobin + ii;
            if (b < numbins) {
                fft[b].r = avgamp;
                fft[b].i = 0.0f;
            }
        }
    }
}
