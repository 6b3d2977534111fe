// accel_host.cpp -- host-side logic of libaccelgpu: search planning, detection
// statistics and the candidate-list replay.  No GPU calls in this file.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "accel_host.hpp"

namespace accelhost {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

// ---- sizes ---------------------------------------------------------------
int z_resp_halfwidth(double z)
{
    // low-accuracy rule of the reference (src/responses.c:55-63): quadratic in
    // |z| on top of 16 bins, capped at 0.6|z| for |z| > 100
    z = std::fabs(z);
    int m = (int) (long) (z * (0.00089 * z + 0.3131) + 16);
    m = std::max(m, 16);
    if (z > 100 && m > 0.6 * z)
        m = (int) (0.6 * z);
    return m;
}

static int nearest_int(double x) { return (int) (x < 0 ? x - 0.5 : x + 0.5); }

int calc_required_z(double harm_fract, double zfull)
{
    return nearest_int(0.5 * zfull * harm_fract) * 2;
}

double calc_required_r(double harm_fract, double rfull)
{
    return (int) (2 * rfull * harm_fract + 0.5) * 0.5;
}

int index_from_z(double z, double loz) { return (int) ((z - loz) * 0.5 + 1e-14); }

int calc_fftlen(int numharm, int harmnum, int max_zfull)
{
    const double f = (double) harmnum / (double) numharm;
    const int bins_needed = (ACCELGPU_USELEN * harmnum) / numharm + 2;
    const int end_effects = 4 * z_resp_halfwidth(calc_required_z(f, max_zfull));
    long long n = 1;
    while (n < bins_needed + end_effects)
        n <<= 1;
    return (int) n;
}

// ---- detection statistics --------------------------------------------------
// Summed power of numsum harmonics is chi^2 with 2*numsum degrees of freedom
// (in units of power = chi^2/2).  For even dof the survival function is a
// finite Poisson sum, so no incomplete-gamma library is needed.
static double chi2_sf_even(double x, int dof)
{
    const int k = dof / 2;
    const double h = 0.5 * x;
    double term = 1.0, sum = 1.0;
    for (int j = 1; j < k; j++) {
        term *= h / j;
        sum += term;
    }
    return std::exp(-h) * sum;
}

static double chi2_logp(double chi2, int dof)
{
    if (chi2 <= 0.0)
        return -INFINITY;
    if (chi2 / dof > 15.0 || (dof > 150 && chi2 / dof > 6.0)) {
        // asymptotic forms used by the reference for large chi^2
        // (src/characteristics.c:418-453, 504-509; A&S 6.5.32 and 6.1.41)
        const double a = 0.5 * dof, z = 0.5 * chi2;
        double x = 1.0, part = 1.0, term = 1.0;
        for (int ii = 1; std::fabs(part) > 1e-15; ii++) {
            term *= (a - ii);
            part = term / std::pow(z, ii);
            x += part;
        }
        const double lig = (a - 1.0) * std::log(z) - z + std::log(x);
        double lg = (a - 0.5) * std::log(a) - a + 0.91893853320467267;
        const double y = 1.0 / (a * a);
        lg += (((-5.9523809523809529e-4 * y + 7.9365079365079365079365e-4) * y -
                2.7777777777777777777778e-3) * y + 8.3333333333333333333333e-2) / a;
        return lig - lg;
    }
    // log(exp(-h) * sum) with the exponential taken analytically: one libm call instead of
    // two, and no underflow for large h
    const int k = dof / 2;
    const double h = 0.5 * chi2;
    double term = 1.0, sum = 1.0;
    for (int j = 1; j < k; j++) {
        term *= h / j;
        sum += term;
    }
    return std::log(sum) - h;
}

static double norm_sf(double x) { return 0.5 * std::erfc(x / M_SQRT2); }

// Upper-tail quantile of the normal distribution from log q, 0 < q <= 1/2: Wichura's
// rational approximations (Applied Statistics algorithm AS 241, routine PPND16; relative
// error below 1e-15, i.e. at the rounding noise of solving Q(x) = q through erfc).  Taking
// log q keeps the full precision of tail probabilities far below the double range of q.
static double norm_isf_log(double lq)
{
    auto horner8 = [](const double (&c)[8], double r) {
        double v = c[7];
        for (int k = 6; k >= 0; k--)
            v = v * r + c[k];
        return v;
    };
    static const double A[8] = {3.3871328727963666080, 1.3314166789178437745e2, 1.9715909503065514427e3,
                                1.3731693765509461125e4, 4.5921953931549871457e4, 6.7265770927008700853e4,
                                3.3430575583588128105e4, 2.5090809287301226727e3};
    static const double B[8] = {1.0, 4.2313330701600911252e1, 6.8718700749205790830e2, 5.3941960214247511077e3,
                                2.1213794301586595867e4, 3.9307895800092710610e4, 2.8729085735721942674e4,
                                5.2264952788528545610e3};
    static const double Cc[8] = {1.42343711074968357734, 4.63033784615654529590, 5.76949722146069140550,
                                 3.64784832476320460504, 1.27045825245236838258, 2.41780725177450611770e-1,
                                 2.27238449892691845833e-2, 7.74545014278341407640e-4};
    static const double Dd[8] = {1.0, 2.05319162663775882187, 1.67638483018380384940, 6.89767334985100004550e-1,
                                 1.48103976427480074590e-1, 1.51986665636164571966e-2, 5.47593808499534494600e-4,
                                 1.05075007164441684324e-9};
    static const double Ee[8] = {6.65790464350110377720, 5.46378491116411436990, 1.78482653991729133580,
                                 2.96560571828504891230e-1, 2.65321895265761230930e-2, 1.24266094738807843860e-3,
                                 2.71155556874348757815e-5, 2.01033439929228813265e-7};
    static const double Ff[8] = {1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2,
                                 7.86869131145613259100e-4, 1.84631831751005468180e-5, 1.42151175831644588870e-7,
                                 2.04426310338993978564e-15};
    if (lq > -2.5902671654458267) { // q > 0.075: the central form works on q itself
        const double d = 0.5 - std::exp(lq), r = 0.180625 - d * d;
        return d * horner8(A, r) / horner8(B, r);
    }
    double r = std::sqrt(-lq);
    if (r <= 5.0) {
        r -= 1.6;
        return horner8(Cc, r) / horner8(Dd, r);
    }
    r -= 5.0;
    return horner8(Ee, r) / horner8(Ff, r);
}

static double equivalent_gaussian_sigma(double logp)
{
    double x;
    if (logp < -600.0) {
        // A&S 26.2.23 on log p (src/characteristics.c:397-415)
        const double t = std::sqrt(-2.0 * logp);
        x = t - (2.515517 + t * (0.802853 + t * 0.010328)) /
                    (1.0 + t * (1.432788 + t * (0.189269 + t * 0.001308)));
    } else if (logp >= 0.0) {
        x = 0.0;  // p = 1 - q <= 0: the reference maps this DCDFLIB range error to 0 (:474-475)
                  // (its other range error, q <= 0 -> 38.5 (:476-477), cannot occur above -600)
    } else if (logp > -0.69314718055994530942) {
        x = -norm_isf_log(std::log1p(-std::exp(logp))); // q > 1/2: by symmetry from 1 - q; clamped to 0 below
    } else {
        x = norm_isf_log(logp);
    }
    return x < 0.0 ? 0.0 : x;
}

static double candidate_sigma_logtrials(double power, int numsum, double log_numtrials)
{
    if (power <= 0.0)
        return 0.0;
    const double logp = chi2_logp(2.0 * power, 2 * numsum) + log_numtrials;
    return equivalent_gaussian_sigma(logp);
}

double candidate_sigma(double power, int numsum, double numtrials)
{
    return candidate_sigma_logtrials(power, numsum, std::log(numtrials));
}

double power_for_sigma(double sigma, int numsum, double numtrials)
{
    const double lq = std::log(norm_sf(sigma) / numtrials);
    const int dof = 2 * numsum;
    double lo = 0.0, hi = 2.0 * dof + 10.0;
    while (std::log(chi2_sf_even(hi, dof)) > lq)
        hi *= 2.0;
    for (int it = 0; it < 200 && hi - lo > 1e-15 * hi; it++) {
        const double mid = 0.5 * (lo + hi);
        (std::log(chi2_sf_even(mid, dof)) > lq ? lo : hi) = mid;
    }
    return 0.25 * (lo + hi);
}

// ---- planning --------------------------------------------------------------
int plan(int64_t numbins, int64_t N, double dt, int zmax, int numharm, double sigma, double flo,
         double fhi, int64_t lobin, accelgpu_setup *s)
{
    std::memset(s, 0, sizeof(*s));
    if (numbins <= 0 || N <= 0 || dt <= 0 || zmax < 0 || zmax > 1200) {
        set_error("accelgpu_plan: bad numbins/N/dt/zmax");
        return ACCELGPU_ERR_ARG;
    }
    if (zmax % 2)
        zmax = (zmax / 2 + 1) * 2;
    if (numharm != 1 && numharm != 2 && numharm != 4 && numharm != 8 && numharm != 16 &&
        numharm != 32) {
        set_error("'numharm' = %d must be a power-of-two!", numharm);
        return ACCELGPU_ERR_ARG;
    }
    s->N = N;
    s->numbins = numbins;
    s->lobin = lobin;
    if (lobin > numbins - 1) {
        set_error("'lobin' is greater than the total number of frequencies in the data set.");
        return ACCELGPU_ERR_RANGE;
    }
    s->zmax = zmax;
    s->numz = (zmax / 2) * 2 + 1;
    s->numharm = numharm;
    int st = 0;
    for (int n = numharm; n > 1; n >>= 1)
        st++;
    s->numharmstages = st + 1;
    s->dt = dt;
    s->T = dt * (double) N;
    s->rlo = std::floor(flo * s->T);
    if (s->rlo < (double) lobin)
        s->rlo = (double) lobin;
    if (s->rlo > numbins - 1) {
        set_error("Low frequency to search 'flo' is greater than the highest available frequency.");
        return ACCELGPU_ERR_RANGE;
    }
    s->highestbin = (int64_t) std::ceil(fhi * s->T);
    if (s->highestbin > numbins - 1)
        s->highestbin = numbins - 1;
    s->rhi = (double) s->highestbin;
    if (s->highestbin < s->rlo) {
        set_error("High frequency to search 'fhi' is less than the lowest frequency to search 'flo'.");
        return ACCELGPU_ERR_RANGE;
    }
    s->zhi = zmax;
    s->zlo = -zmax;
    s->sigma = (float) sigma;
    for (int ii = 0; ii < s->numharmstages; ii++) {
        if (s->numz == 1)
            s->numindep[ii] = (int64_t) ((s->rhi - s->rlo) / (1 << ii));
        else
            s->numindep[ii] =
                (int64_t) ((s->rhi - s->rlo) * (s->numz + 1) * (2.0 / 6.95) / (1 << ii));
        s->powcut[ii] = (float) power_for_sigma(s->sigma, 1 << ii, (double) s->numindep[ii]);
    }
    s->fftlen = calc_fftlen(1, 1, zmax);
    s->halfwidth = z_resp_halfwidth((double) zmax);
    s->rlen = (s->highestbin + ACCELGPU_USELEN) * 2;
    s->plane_bytes = s->rlen * (int64_t) s->numz * 4;
    const double step = ACCELGPU_USELEN * 0.5;
    for (double r = 8; r < s->rlo; r += step)
        s->num_prepop_blocks++;
    for (double r = s->rlo; r + step < (double) s->highestbin; r += step)
        s->num_search_blocks++;
    return ACCELGPU_OK;
}

// ---- candidate list --------------------------------------------------------
// The reference keeps an r-ordered singly linked list and, for every hit, walks
// it from the head to the first entry with r >= rr (or the last entry).  The
// same walk on a contiguous array gives the same list; entries can fall out of
// order after an in-place overwrite, so the scan is linear on purpose.
struct Entry {
    double r;
    float sigma, power;
    int numharm;
    int order;
    double z;
};

static std::atomic<bool> g_replay_linear{false};
static std::atomic<int> g_replay_threads{-1};

void block_grid(double rlo, int64_t highestbin, int halfwidth, BlockGrid *g)
{
    *g = BlockGrid();
    const double step = ACCELGPU_USELEN * 0.5;
    auto describe = [&](double startr, int64_t *lobin, int *brlo, int *numdata) {
        const double fullrhi = startr + ACCELGPU_USELEN * 0.5 - 0.5;
        const double drlo = calc_required_r(1.0, startr), drhi = calc_required_r(1.0, fullrhi);
        *brlo = (int) std::floor(drlo);
        *lobin = (int64_t) *brlo - halfwidth;
        const int hibin = (int) std::ceil(drhi) + halfwidth;
        *numdata = (int) (hibin - *lobin + 1);
    };
    g->search_col0 = (int64_t) std::floor(calc_required_r(1.0, rlo)) * 2;
    auto add = [&](double r, bool prepop) {
        int64_t lobin;
        int brlo, numdata;
        describe(r, &lobin, &brlo, &numdata);
        g->lobin.push_back(lobin);
        g->rlo.push_back(brlo);
        g->numdata.push_back(numdata);
        g->col.push_back((int64_t) brlo * 2);
        // columns at or above the first search column belong to the search blocks
        g->ncols.push_back(prepop ? (int) std::max<int64_t>(0, std::min<int64_t>(ACCELGPU_USELEN, g->search_col0 - (int64_t) brlo * 2))
                                  : ACCELGPU_USELEN);
    };
    for (double r = 8; r < rlo; r += step) {
        add(r, true);
        g->nprepop++;
    }
    for (double r = rlo; r + step < (double) highestbin; r += step) {
        add(r, false);
        g->nsearch++;
    }
}

int block_range_need(const accelgpu_setup *s, const BlockGrid &g, int first, int n, std::vector<int> *z0, std::vector<int> *z1,
                     int64_t *rows_generated)
{
    const int nblk = g.nprepop + g.nsearch, numz = s->numz;
    z0->assign((size_t) nblk, 0);
    z1->assign((size_t) nblk, 0);
    auto want = [&](int b, int lo, int hi) { // rows [lo, hi) of block b as well
        if ((*z1)[b] == 0) {
            (*z0)[b] = lo;
            (*z1)[b] = hi;
        } else {
            (*z0)[b] = std::min((*z0)[b], lo);
            (*z1)[b] = std::max((*z1)[b], hi);
        }
    };
    if (n > 0) {
        for (int k = 0; k < n; k++)
            want(g.nprepop + first + k, 0, numz);
        const int64_t ca = g.search_col0 + (int64_t) first * ACCELGPU_USELEN;
        const int64_t cb = ca + (int64_t) n * ACCELGPU_USELEN - 1;
        // the block a plane column lies in: pre-population blocks are USELEN columns apart from their first
        // column on, search blocks from search_col0 on; columns below the first block belong to none
        auto block_of = [&](int64_t c) -> int {
            if (c >= g.search_col0)
                return g.nprepop + (int) std::min<int64_t>((c - g.search_col0) / ACCELGPU_USELEN, g.nsearch - 1);
            if (g.nprepop == 0 || c < g.col[0])
                return -1;
            return (int) std::min<int64_t>((c - g.col[0]) / ACCELGPU_USELEN, g.nprepop - 1);
        };
        for (int st = 1; st < s->numharmstages; st++)
            for (int h = 1; h < (1 << st); h += 2) {
                const double f = (double) h / (double) (1 << st);
                const int64_t lo = (ca * h + (1 << (st - 1))) >> st, hi = (cb * h + (1 << (st - 1))) >> st;
                // source rows: monotonic in z, so the ends give the range
                const int ra = index_from_z(calc_required_z(f, s->zlo), s->zlo);
                const int rb = index_from_z(calc_required_z(f, s->zlo + 2.0 * (numz - 1)), s->zlo);
                const int rlo = std::min(ra, rb), rhi = std::max(ra, rb) + 1;
                int b0 = block_of(lo);
                const int b1 = block_of(hi);
                if (b1 < 0)
                    continue; // below every block: cells nothing writes (they read as zero)
                b0 = std::max(b0 - 1, 0); // (a block either side: the test below is the exact one)
                for (int b = b0; b <= std::min(b1 + 1, nblk - 1); b++)
                    if (g.ncols[b] > 0 && g.col[b] <= hi && g.col[b] + g.ncols[b] - 1 >= lo)
                        want(b, rlo, rhi);
            }
    }
    int blocks = 0;
    int64_t rows = 0;
    for (int b = 0; b < nblk; b++)
        if ((*z1)[b] > (*z0)[b]) {
            blocks++;
            rows += (*z1)[b] - (*z0)[b];
        }
    if (rows_generated)
        *rows_generated = rows;
    return blocks;
}

void replay_options(int linear_walk, int threads)
{
    g_replay_linear.store(linear_walk != 0);
    g_replay_threads.store(threads);
}

int replay(const accelgpu_setup *s, const accelgpu_hit *hits, int64_t n_hits,
           accelgpu_cand *cands, int64_t max_cands, int64_t *n_cands, float *sigma_out,
           int32_t *added_out)
{
    const double CLOSEST = 15.0;
    std::vector<Entry> list;
    std::vector<double> pmax; // running maximum of list[..].r, same length as list
    list.reserve(1024);
    pmax.reserve(1024);
    // replay_options(): the plain list walk (the check of the binary search below) and the
    // number of threads that compute sigmas; neither changes a result
    const bool linear_walk = g_replay_linear.load(std::memory_order_relaxed);
    const int env_threads = g_replay_threads.load(std::memory_order_relaxed);
    const double zlo = s->zlo;
    for (int64_t i = 0; i < n_hits; i++)
        if (hits[i].stage < 0 || hits[i].stage >= s->numharmstages) {
            set_error("accelgpu_replay_candidates: hit %lld has stage %d", (long long) i,
                      hits[i].stage);
            return ACCELGPU_ERR_ARG;
        }
    // The significance of a hit depends on that hit alone (and costs ~0.2 us of libm
    // calls): computed for all hits up front, on several threads when there are
    // thousands.  The list walk below stays sequential, in the reference's order.
    std::vector<float> sigs((size_t) n_hits);
    double log_trials[ACCELGPU_MAXSTAGES];
    for (int st = 0; st < s->numharmstages; st++)
        log_trials[st] = std::log((double) s->numindep[st]);
    auto sigma_range = [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; i++)
            sigs[(size_t) i] = (float) candidate_sigma_logtrials(hits[i].power, 1 << hits[i].stage,
                                                                log_trials[hits[i].stage]);
    };
    int nthreads = (int) std::min<int64_t>(n_hits / 2048, 8);
    if (env_threads >= 0)
        nthreads = env_threads;
    static const int host_cores = (int) std::max(1u, std::thread::hardware_concurrency()); // asked once: it reads sysfs
    nthreads = std::min<int>(nthreads, host_cores);
    if (nthreads <= 1) {
        sigma_range(0, n_hits);
    } else {
        std::vector<std::thread> pool;
        const int64_t per = (n_hits + nthreads - 1) / nthreads;
        for (int t = 1; t < nthreads; t++)
            pool.emplace_back(sigma_range, std::min(n_hits, t * per), std::min(n_hits, (t + 1) * per));
        sigma_range(0, std::min(n_hits, per));
        for (auto &th : pool)
            th.join();
    }
    for (int64_t i = 0; i < n_hits; i++) {
        const accelgpu_hit &h = hits[i];
        const int numharm = 1 << h.stage;
        const float sig = sigs[(size_t) i];
        const double rr = ((double) h.rind * 0.5) / (double) numharm;
        const double zz = (zlo + h.zind * 2.0) / (double) numharm;
        if (sigma_out)
            sigma_out[i] = sig;
        const Entry e{rr, sig, h.power, numharm, (int) i, zz};
        int added = 0;
        if (list.empty()) {
            list.push_back(e);
            pmax.push_back(rr);
            added = 1;
        } else {
            const size_t last = list.size() - 1;
            // first entry with r >= rr, or the last one: the reference's walk from the head.
            // pmax[i] = max r of entries 0..i never decreases, and its first value >= rr sits
            // at the first entry >= rr, so the walk is a binary search.
            size_t cur;
            if (linear_walk) {
                cur = 0;
                while (cur < last && list[cur].r < rr)
                    cur++;
            } else {
                cur = (size_t) (std::lower_bound(pmax.begin(), pmax.begin() + (std::ptrdiff_t) last, rr) -
                                pmax.begin());
            }
            const bool has_prev = cur > 0;
            const double next_diff = std::fabs(rr - list[cur].r);
            const double prev_diff = has_prev ? std::fabs(rr - list[cur - 1].r) : CLOSEST + 1.0;
            size_t dirty = list.size(); // first position whose prefix maximum may have changed
            if (prev_diff < CLOSEST) {
                if (list[cur - 1].sigma < sig) {
                    list[cur - 1] = e;
                    dirty = cur - 1;
                    added = 1;
                }
                if (next_diff < CLOSEST && list[cur].sigma < sig) {
                    if (added) {
                        list.erase(list.begin() + (std::ptrdiff_t) cur);
                        pmax.pop_back();
                    } else {
                        list[cur] = e;
                        dirty = cur;
                        added = 1;
                    }
                }
            } else if (next_diff < CLOSEST) {
                if (list[cur].sigma < sig) {
                    list[cur] = e;
                    dirty = cur;
                    added = 1;
                }
            } else {
                added = 1;
                if (cur == last && list[cur].r < rr - CLOSEST) {
                    list.push_back(e);
                    dirty = list.size() - 1;
                } else {
                    list.insert(list.begin() + (std::ptrdiff_t) cur, e); // before cur (also the new-head case)
                    dirty = cur;
                }
                pmax.push_back(0.0);
            }
            for (size_t k = dirty; k < list.size(); k++)
                pmax[k] = k == 0 ? list[0].r : std::max(pmax[k - 1], list[k].r);
        }
        if (added_out)
            added_out[i] = added;
    }
    *n_cands = (int64_t) list.size();
    if ((int64_t) list.size() > max_cands) {
        set_error("accelgpu_replay_candidates: %zu candidates, room for %lld", list.size(),
                  (long long) max_cands);
        return ACCELGPU_ERR_OVERFLOW;
    }
    for (size_t i = 0; i < list.size(); i++) {
        cands[i].power = list[i].power;
        cands[i].sigma = list[i].sigma;
        cands[i].numharm = list[i].numharm;
        cands[i].added_order = list[i].order;
        cands[i].r = list[i].r;
        cands[i].z = list[i].z;
    }
    return ACCELGPU_OK;
}

// ---- candidate list trimming -------------------------------------------------
// sort_accelcands() (reference src/accel_utils.c:278-298): decreasing sigma, ties
// by decreasing power, stable.  eliminate_harmonics() (:383-462): walking the
// sorted list from the most significant candidate, every later candidate whose r
// lies within 1.5 bins of r/k, r*k (k = 1..16) or r*a/b for the listed small
// ratios is dropped.
int64_t sort_and_eliminate(accelgpu_cand *cands, int64_t n, bool eliminate)
{
    std::stable_sort(cands, cands + n, [](const accelgpu_cand &a, const accelgpu_cand &b) {
        if (a.sigma != b.sigma)
            return a.sigma > b.sigma;
        return a.power > b.power;
    });
    if (!eliminate || n < 2)
        return n;
    static const int num[] = {3, 5, 2, 4, 5, 3, 5, 2, 3, 4, 5, 2, 3, 4, 3, 5, 2, 3, 2, 3, 2, 3, 2};
    static const int den[] = {2, 2, 3, 3, 3, 4, 4, 5, 5, 5, 6, 7, 7, 7, 8, 8, 9, 10, 11, 11, 13, 13, 15};
    const double tooclose = 1.5;
    auto related = [&](double cr, double orr) {
        for (int k = 1; k <= 16; k++)
            if (std::fabs(cr / k - orr) < tooclose)
                return true;
        for (int k = 1; k <= 16; k++)
            if (std::fabs(cr * k - orr) < tooclose)
                return true;
        for (size_t k = 0; k < sizeof(num) / sizeof(num[0]); k++)
            if (std::fabs(cr * (double) num[k] / (double) den[k] - orr) < tooclose)
                return true;
        return false;
    };
    std::vector<accelgpu_cand> kept;
    std::vector<accelgpu_cand> rest(cands, cands + n);
    // each pass keeps the head and filters what follows it, as the list walk does
    while (!rest.empty()) {
        const accelgpu_cand cur = rest.front();
        kept.push_back(cur);
        std::vector<accelgpu_cand> next;
        for (size_t i = 1; i < rest.size(); i++)
            if (!related(cur.r, rest[i].r))
                next.push_back(rest[i]);
        rest.swap(next);
    }
    std::copy(kept.begin(), kept.end(), cands);
    return (int64_t) kept.size();
}

// get_birdies(), reference src/birdzap.c:6-88: topocentric lines are shifted by
// (1 + baryv), every range is made at least one bin wide, the list comes back
// sorted by its low edge.
int read_birds(const char *path, double T, double baryv, double *lobins, double *hibins,
               int32_t max_birds, int32_t *n_birds)
{
    FILE *f = fopen(path, "r");
    if (!f) {
        set_error("cannot open birdie file '%s'", path);
        return ACCELGPU_ERR_ARG;
    }
    std::vector<std::pair<double, double>> birds;
    char line[256];
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#')
            continue;
        double freq = 0.0, width = 0.0, lo, hi;
        if (line[0] == 'B') {
            if (sscanf(line, "B%lf %lf", &freq, &width) != 2)
                continue;
            lo = (freq - 0.5 * width) * T;
            hi = (freq + 0.5 * width) * T;
        } else {
            if (sscanf(line, "%lf %lf", &freq, &width) != 2)
                continue;
            lo = (freq - 0.5 * width) * T * (1.0 + baryv);
            hi = (freq + 0.5 * width) * T * (1.0 + baryv);
        }
        if (hi - lo < 1.0) {
            const double mid = 0.5 * (hi + lo);
            lo = mid - 0.5;
            hi = mid + 0.5;
        }
        birds.emplace_back(lo, hi);
    }
    fclose(f);
    std::stable_sort(birds.begin(), birds.end(),
                     [](const std::pair<double, double> &a, const std::pair<double, double> &b) {
                         return a.first < b.first;
                     });
    *n_birds = (int32_t) birds.size();
    if ((int32_t) birds.size() > max_birds) {
        set_error("%zu birds but room for %d", birds.size(), max_birds);
        return ACCELGPU_ERR_OVERFLOW;
    }
    for (size_t i = 0; i < birds.size(); i++) {
        lobins[i] = birds[i].first;
        hibins[i] = birds[i].second;
    }
    return ACCELGPU_OK;
}

} // namespace accelhost
