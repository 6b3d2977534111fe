// accel_output.cpp -- host side of the search's tail: signal properties derived
// from the optimised candidates and the two result files consumers read, the
// ACCEL text table and the .cand array of fourierprops.  No GPU calls.
//
// Reference: calc_props()/calc_rzwerrs() src/characteristics.c:193-256, 351-394;
// nice_output_1()/nice_output_2() src/output.c:6-219; output_fundamentals()/
// output_harmonics()/compare_rzw_cands() src/accel_utils.c:87-117, 565-835;
// the tail of main(), src/accelsearch.c:256-286.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "accel_host.hpp"

namespace accelhost {

namespace {

const double kPi = 3.1415926535897932384626433832795028841971693993751;
const double kTwoPi = 6.2831853071795864769252867665590057683943387987502;
const double kTiny = 1e-14;          // DBLCORRECT, include/presto.h:21
const double kSol = 299792458.0;     // SOL, include/presto.h:39

std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2)));
std::string fmt(const char *f, ...)
{
    char buf[128];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return buf;
}

int decade(double x) { return x == 0.0 ? 0 : (int) std::floor(std::log10(std::fabs(x))); }
double p10(int e) { return std::pow(10.0, (double) e); }
double with_sign_of(double mag, double ref) { return ref < 0.0 ? -std::fabs(mag) : std::fabs(mag); }

// A cell: text centred in `width` characters (left-biased when the padding is
// odd, cut when too long), as center_string() of src/accel_utils.c:526-544.
std::string centred(const std::string &text, int width)
{
    std::string out((size_t) width, ' ');
    const int len = (int) text.size();
    if (len >= width)
        return text.substr(0, (size_t) width);
    out.replace((size_t) ((width - len) / 2), (size_t) len, text);
    return out;
}

struct Column {
    const char *top, *bottom;
    int width;
};

void write_header(FILE *f, const std::vector<Column> &cols, const char *bottom3 = nullptr)
{
    const size_t n = cols.size();
    for (int row = 0; row < 3; row++)
        for (size_t c = 0; c < n; c++) {
            std::string cell;
            if (row == 2)
                cell.assign((size_t) cols[c].width, '-');
            else
                cell = centred(row == 0 ? cols[c].top : (bottom3 && c == 3 ? bottom3 : cols[c].bottom),
                               cols[c].width);
            const char *sep = c + 1 == n ? "\n" : (row == 2 ? "--" : "  ");
            fprintf(f, "%s%s", cell.c_str(), sep);
        }
}

} // namespace

// ---- value(error) strings ------------------------------------------------------
// One significant digit of error.  src/output.c:6-114.
std::string nice_output_1(double val, double err)
{
    if (!std::isfinite(val) || !std::isfinite(err)) // the reference's integer conversions are undefined here
        return fmt("% g(%g)", val, err);
    int errexp = decade(err);
    int errval = (int) std::floor(std::fabs(err) * p10(-errexp) + kTiny + 0.5);
    if (errval == 10) {
        errval = 1;
        errexp++;
    }
    const double rnd = p10(errexp) * std::floor(val * p10(-errexp) + 0.5);
    int nint = 1;
    if (std::fabs(val) != 0.0) {
        nint = (int) std::ceil(std::log10(std::fabs(val)));
        if (nint == 0)
            nint = 1;
    }
    const int nfrac = -errexp;
    int totprec = nint + nfrac;
    const int outexp = decade(rnd);
    const double mant = rnd * p10(-outexp);
    if (std::fabs(1.0 - mant) < kTiny || std::fabs(-1.0 - mant) < kTiny)
        totprec++;
    const double smant = with_sign_of(mant, rnd);
    const bool big = outexp >= 0 && errexp > 0;
    if (big && outexp > errexp)
        return fmt("% .*f(%d)x10^%d", totprec - 1, smant, errval, outexp);
    if (big && outexp == errexp)
        return fmt("% d(%d)x10^%d", (int) smant, errval, outexp);
    if (outexp < -4 && outexp >= errexp)
        return fmt("% .*f(%d)x10^%d", totprec - 1, smant, errval, outexp);
    if (outexp < errexp && errexp != 0)
        return fmt("% d(%d)x10^%d", (int) (smant + kTiny), errval, errexp);
    if (nfrac == 0 && std::fabs(rnd) < 1.0e-15)
        return fmt("% d(%d)", (int) std::fabs(rnd), errval);
    if (std::fabs(rnd) <= kTiny && errexp < -5)
        return fmt("0.0(%d)x10^%d", errval, errexp + 1);
    return fmt("% .*f(%d)", nfrac, rnd, errval);
}

// Two significant digits of error.  src/output.c:117-219.  `ok` is cleared for the
// combination the reference calls undefined (it prints and exits there).
std::string nice_output_2(double val, double err, bool *ok)
{
    if (ok)
        *ok = true;
    if (!std::isfinite(val) || !std::isfinite(err)) // as in nice_output_1
        return fmt("% g(%g)", val, err);
    int errexp = decade(err);
    int errval = (int) std::floor(std::fabs(err) * p10(-errexp + 1) + kTiny + 0.5);
    if (errval == 100) {
        errval = 10;
        errexp++;
    }
    double rnd = p10(errexp - 1) * std::floor(val * p10(-errexp + 1) + 0.5);
    const int outexp = decade(rnd), valexp = decade(val);
    double mant = rnd * p10(-outexp);
    const int e1 = errval / 10, e2 = errval % 10;
    if (outexp < -4) {
        const int digits = outexp - errexp + 1;
        if (digits == 0)
            return fmt("% .1f(%d.%d)x10^%d", mant / 10, e1, e2, outexp + 1);
        if (digits < 2)
            return fmt("% .1f(%d.%d)x10^%d", mant, e1, e2, outexp);
        return fmt("% .*f(%.2d)x10^%d", digits, mant, errval, outexp);
    }
    if (errexp - valexp > 1 && errexp != 0 && errexp != 1)
        return fmt("0.0(%d.%d)x10^%d", e1, e2, errexp);
    if (errexp == 0) {
        if (std::fabs(rnd) < 0.1)
            rnd = std::fabs(rnd);
        return fmt("% .1f(%d.%d)", rnd, e1, e2);
    }
    if (errexp < 0) {
        if (std::fabs(rnd) * p10(-errexp + 1) < 0.1)
            rnd = std::fabs(rnd);
        if (std::fabs(rnd) <= kTiny && errexp < -4)
            return fmt("0.0(%d.%d)x10^%d", e1, e2, errexp);
        return fmt("% .*f(%.2d)", -errexp + 1, rnd, errval);
    }
    if (errexp == outexp) {
        if (errexp == 1)
            return fmt("% d(%.2d)", (int) rnd, errval);
        return fmt("% .1f(%d.%d)x10^%d", mant, e1, e2, errexp);
    }
    if (errexp - outexp == 1) {
        if (std::fabs(mant) < 0.99999999999999)
            mant = std::fabs(mant);
        return fmt("% .1f(%d.%d)x10^%d", mant / 10.0, e1, e2, errexp);
    }
    if (errexp > 1 && outexp > 1 && errexp - outexp <= -1)
        return fmt("% .*f(%.2d)x10^%d", outexp - errexp + 1, mant, errval, outexp);
    if (errexp == 1 && outexp > 0)
        return fmt("% d(%.2d)", (int) rnd, errval);
    if (ok)
        *ok = false;
    return "undefined";
}

// ---- derived properties -----------------------------------------------------------
// Middleditch, Deich & Kulkarni estimators.  src/characteristics.c:193-256.
int calc_props(const accelgpu_rderivs &d, double r, double z, double w, accelgpu_fourierprops *out)
{
    if (std::fabs(d.pow) < kTiny) {
        set_error("data.pow = %f (out-of-bounds) in calc_props()", d.pow);
        return ACCELGPU_ERR_ARG;
    }
    if (d.locpow < kTiny) {
        set_error("data.locpow = %f (out-of-bounds) in calc_props()", d.locpow);
        return ACCELGPU_ERR_ARG;
    }
    const double P = d.pow / d.locpow, P2 = d.d2pow / d.locpow;
    accelgpu_fourierprops &o = *out;
    o.pur = std::sqrt(1.5 * std::fabs(P2) / P) / kPi;
    if (std::fabs(P) < kTiny)
        o.pur = 1.0;
    o.purerr = 1.0 / (o.pur * std::sqrt(10.0 * P));
    o.r = r;
    o.rerr = 3.0 / (kPi * o.pur * std::sqrt(6.0 * P));
    o.z = z;
    o.zerr = 3.0 * std::sqrt(10.0) / (kPi * o.pur * o.pur * std::sqrt(P));
    o.w = w;
    o.werr = 6.0 * std::sqrt(105.0) / (kPi * o.pur * o.pur * o.pur * std::sqrt(P));
    o.pow = P;
    o.powerr = std::sqrt(2.0 * P);
    o.sig = std::sqrt(2.0 * P - std::log(kPi * P));
    o.rawpow = d.pow;
    o.phs = d.phs;
    o.phserr = 1.0 / o.powerr;
    o.cen = -d.dphs / kTwoPi;
    o.cenerr = 1.0 / std::sqrt(24.0 * P);
    o.locpow = d.locpow;
    return ACCELGPU_OK;
}

namespace {

// Periods, frequencies and their errors.  src/characteristics.c:351-394.
struct RzwErrs {
    double p, pd, pdd, f, fd, fdd;
    float perr, pderr, pdderr, ferr, fderr, fdderr;
};

RzwErrs calc_rzwerrs(const accelgpu_fourierprops &q, double T)
{
    RzwErrs e;
    const double T2 = T * T, T3 = T2 * T, r2 = q.r * q.r, r4 = r2 * r2;
    const double sr2 = (double) q.rerr * q.rerr, z2 = q.z * q.z, sz2 = (double) q.zerr * q.zerr;
    const double t = q.r * q.w - 3 * z2;
    e.f = q.r / T;
    e.ferr = q.rerr / T;
    e.fd = q.z / T2;
    e.fderr = q.zerr / T2;
    e.fdd = q.w / T3;
    e.fdderr = q.werr / T3;
    e.p = T / q.r;
    e.perr = T * q.rerr / r2;
    e.pd = -q.z / r2;
    e.pderr = std::sqrt(4 * z2 * sr2 / (r4 * r2) + sz2 / r4);
    e.pdd = q.w == 0.0 ? 0.0 : (2 * z2 - q.r * q.w) / (T * r2 * q.r);
    e.pdderr = std::sqrt(((double) q.werr * q.werr * r4 + 16 * sz2 * r2 + 4 * t * t * sr2) / (r4 * r4 * T2));
    return e;
}

// "SL? of Cand n" / "H k of Cand n" notes.  src/accel_utils.c:87-117.
std::vector<std::string> self_notes(const std::vector<accelgpu_fourierprops> &p)
{
    const int n = (int) p.size();
    std::vector<std::string> notes((size_t) n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            if (i == j)
                continue;
            if (std::fabs(p[i].r - p[j].r) < 15.0 && std::fabs(p[i].z - p[j].z) > 1.0 && p[i].pow > p[j].pow) {
                if (notes[j].empty())
                    notes[j] = fmt("SL? of Cand %d", i + 1);
                continue;
            }
            for (int k = 1; k < 61; k++)
                if (std::fabs(p[i].r - p[j].r / k) < p[j].rerr * 3 &&
                    std::fabs(p[i].z - p[j].z / k) < p[j].zerr * 2 && notes[j].empty()) {
                    notes[j] = fmt("H %d of Cand %d", k, i + 1);
                    break;
                }
        }
    return notes;
}

// One "value(error)" cell followed by the two-space separator.
int put_measured(FILE *f, double val, double err, int digits, int width)
{
    bool ok = true;
    const std::string s = digits == 1 ? nice_output_1(val, err) : nice_output_2(val, err, &ok);
    if (!ok) {
        set_error("value %g with error %g cannot be formatted (undefined condition in nice_output2)", val, err);
        return ACCELGPU_ERR_ARG;
    }
    fprintf(f, "%s  ", centred(s, width).c_str());
    return ACCELGPU_OK;
}

void put_text(FILE *f, const std::string &s, int width) { fprintf(f, "%s  ", centred(s, width).c_str()); }

} // namespace

int write_txtcand(const accelgpu_setup *su, const accelgpu_hit *hits, const float *sigma, const int32_t *added, int64_t n,
                  const char *path)
{
    if (!su || !path || (n > 0 && (!hits || !sigma || !added)) || !(su->T > 0.0)) {
        set_error("write_txtcand: bad arguments");
        return ACCELGPU_ERR_ARG;
    }
    FILE *w = fopen(path, "w");
    if (!w) {
        set_error("cannot open '%s' for writing", path);
        return ACCELGPU_ERR_ARG;
    }
    // one line per hit the list took, in the order it took them (src/accel_utils.c:1290-1294)
    for (int64_t i = 0; i < n; i++)
        if (added[i]) {
            const int nh = 1 << hits[i].stage;
            const double rr = hits[i].rind * 0.5 / nh, zz = (su->zlo + hits[i].zind * 2.0) / nh;
            fprintf(w, "%-7.2f  %-7.4f  %-2d  %-14.4f  %-14.9f  %-10.4f\n", hits[i].power, sigma[i], nh, rr, rr / su->T, zz);
        }
    if (fclose(w) != 0) {
        set_error("error writing '%s'", path);
        return ACCELGPU_ERR_ARG;
    }
    return ACCELGPU_OK;
}

int write_accel_files(const accelgpu_setup *su, const accelgpu_optcand *cands, int64_t n, float nph,
                      const char *accel_path, const char *cand_path, const char *inf_text)
{
    if (!su || (n > 0 && !cands) || !accel_path || !cand_path || !(su->T > 0.0)) {
        set_error("write_accel_files: bad arguments");
        return ACCELGPU_ERR_ARG;
    }
    // properties of the fundamentals; errors overridden by the search grid (src/accelsearch.c:258-270)
    std::vector<accelgpu_fourierprops> props((size_t) n);
    for (int64_t i = 0; i < n; i++) {
        memset(&props[(size_t) i], 0, sizeof(accelgpu_fourierprops));
        int rc = calc_props(cands[i].derivs[0], cands[i].r, cands[i].z, 0.0, &props[(size_t) i]);
        if (rc)
            return rc;
        props[(size_t) i].rerr = 0.5f / cands[i].numharm;
        props[(size_t) i].zerr = 2.0f / cands[i].numharm;
    }
    FILE *f = fopen(accel_path, "w");
    if (!f) {
        set_error("cannot open '%s' for writing", accel_path);
        return ACCELGPU_ERR_ARG;
    }
    const std::vector<std::string> notes = self_notes(props);
    const double T = su->T;
    int rc = ACCELGPU_OK;

    static const std::vector<Column> fund = {
        {"", "Cand", 4}, {"", "Sigma", 5}, {"Summed", "Power", 6}, {"Coherent", "Power", 8},
        {"Num", "Harm", 4}, {"Period", "(ms)", 16}, {"Frequency", "(Hz)", 15}, {"FFT 'r'", "(bin)", 15},
        {"Freq Deriv", "(Hz/s)", 15}, {"FFT 'z'", "(bins)", 11}, {"Accel", "(m/s^2)", 15}, {"", "Notes", 20}};
    write_header(f, fund);
    for (int64_t i = 0; i < n && !rc; i++) {
        const accelgpu_optcand &c = cands[i];
        const accelgpu_fourierprops &q = props[(size_t) i];
        const RzwErrs e = calc_rzwerrs(q, T);
        // coherently summed power, phases referred to the fundamental's
        double cr = 0.0, ci = 0.0;
        const double phs0 = c.derivs[0].phs;
        for (int h = 0; h < c.numharm; h++) {
            const accelgpu_rderivs &d = c.derivs[h];
            const double amp = std::sqrt(d.pow / (nph > 0.0f ? (double) nph : d.locpow));
            const double corr = phs0 - std::fmod((h + 1.0) * phs0, kTwoPi);
            cr += amp * std::cos(d.phs + corr);
            ci += amp * std::sin(d.phs + corr);
        }
        put_text(f, fmt("%-4d", (int) i + 1), fund[0].width);
        put_text(f, fmt("%.2f", c.sigma), fund[1].width);
        put_text(f, fmt("%.2f", c.power), fund[2].width);
        put_text(f, fmt("%.2f", cr * cr + ci * ci), fund[3].width);
        put_text(f, fmt("%d", c.numharm), fund[4].width);
        const double acc = q.z * kSol / (T * T * e.f), accerr = q.zerr * kSol / (T * T * e.f);
        if ((rc = put_measured(f, e.p * 1000.0, e.perr * 1000.0, 1, fund[5].width))) break;
        if ((rc = put_measured(f, e.f, e.ferr, 1, fund[6].width))) break;
        if ((rc = put_measured(f, q.r, q.rerr, 2, fund[7].width))) break;
        if ((rc = put_measured(f, e.fd, e.fderr, 1, fund[8].width))) break;
        if ((rc = put_measured(f, q.z, q.zerr, 2, fund[9].width))) break;
        if ((rc = put_measured(f, acc, accerr, 2, fund[10].width))) break;
        // an empty note is twenty blanks; a written one ends where its text ends
        // (the reference's strncpy fills the rest of the field with NULs)
        fprintf(f, "  %.20s\n", notes[(size_t) i].empty() ? "                    " : notes[(size_t) i].c_str());
    }
    fprintf(f, "\n\n");

    static const std::vector<Column> harm = {
        {"", "Cand", 5}, {"", "Harm", 4}, {"", "Sigma", 5}, {"Power /", "Loc Pow", 15}, {"Raw", "Power", 11},
        {"FFT 'r'", "(bin)", 18}, {"Pred 'r'", "(bin)", 13}, {"FFT 'z'", "(bins)", 12},
        {"Pred 'z'", "(bins)", 9}, {"Phase", "(rad)", 12}, {"Centroid", "(0-1)", 10},
        {"Purity", "<p> = 1", 10}, {"", "Notes", 20}};
    if (!rc)
        write_header(f, harm, nph > 0.0f ? "NumPhot" : nullptr);
    for (int64_t i = 0; i < n && !rc; i++) {
        const accelgpu_optcand &c = cands[i];
        for (int h = 0; h < c.numharm && !rc; h++) {
            accelgpu_rderivs d = c.derivs[h];
            if (nph > 0.0f)
                d.locpow = nph;
            accelgpu_fourierprops q;
            memset(&q, 0, sizeof(q));
            if ((rc = calc_props(d, c.hirs[h], c.hizs[h], 0.0, &q))) break;
            put_text(f, h == 0 ? fmt(" %-4d", (int) i + 1) : std::string("     "), harm[0].width);
            put_text(f, fmt("%-4d", h + 1), harm[1].width);
            put_text(f, fmt("%.2f", candidate_sigma(q.pow, 1, 1)), harm[2].width);
            if ((rc = put_measured(f, q.pow, q.powerr, 2, harm[3].width))) break;
            put_text(f, fmt("%.3g", q.rawpow), harm[4].width);
            if ((rc = put_measured(f, q.r, q.rerr, 2, harm[5].width))) break;
            put_text(f, fmt("%.2f", c.r * (h + 1)), harm[6].width);
            if ((rc = put_measured(f, q.z, q.zerr, 2, harm[7].width))) break;
            put_text(f, fmt("%.2f", c.z * (h + 1)), harm[8].width);
            if ((rc = put_measured(f, q.phs, q.phserr, 2, harm[9].width))) break;
            if ((rc = put_measured(f, q.cen, q.cenerr, 2, harm[10].width))) break;
            if ((rc = put_measured(f, q.pur, q.purerr, 2, harm[11].width))) break;
            fprintf(f, "  %.20s\n", "                    ");
        }
    }
    if (!rc) {
        fprintf(f, "\n\n");
        if (inf_text)
            fputs(inf_text, f); // the reference appends the .inf file here (src/accel_utils.c:831-833)
    }
    fclose(f);
    if (rc)
        return rc;
    f = fopen(cand_path, "wb");
    if (!f) {
        set_error("cannot open '%s' for writing", cand_path);
        return ACCELGPU_ERR_ARG;
    }
    if (n > 0 && fwrite(props.data(), sizeof(accelgpu_fourierprops), (size_t) n, f) != (size_t) n) {
        fclose(f);
        set_error("short write on '%s'", cand_path);
        return ACCELGPU_ERR_ARG;
    }
    fclose(f);
    return ACCELGPU_OK;
}

namespace {

// leading number of a value(error) string
double leading_number(const std::string &t) { return std::strtod(t.c_str(), nullptr); }

// '12.3(4)' or '1.2(3)x10^5' -> value; an exponent that does not read as a number counts as 5
double printed_power(const std::string &t)
{
    double v = leading_number(t);
    const size_t hat = t.find('^');
    if (hat != std::string::npos) {
        const char *b = t.c_str() + hat + 1;
        char *e = nullptr;
        double ex = std::strtod(b, &e);
        while (e && (*e == ' ' || *e == '\t' || *e == '\n'))
            e++;
        if (e == b || (e && *e != '\0'))
            ex = 5.0;
        v *= std::pow(10.0, ex);
    }
    return v;
}

} // namespace

int sift_records(const accelgpu_optcand *cands, int64_t n, float nph, accelgpu_siftrec *out)
{
    if (n < 0 || (n > 0 && (!cands || !out))) {
        set_error("sift_records: bad arguments");
        return ACCELGPU_ERR_ARG;
    }
    for (int64_t i = 0; i < n; i++) {
        const accelgpu_optcand &c = cands[i];
        accelgpu_siftrec &o = out[i];
        memset(&o, 0, sizeof(o));
        if (c.numharm < 1 || c.numharm > ACCELGPU_MAXHARM) {
            set_error("sift_records: candidate %lld has numharm %d", (long long) i, c.numharm);
            return ACCELGPU_ERR_ARG;
        }
        double cr = 0.0, ci = 0.0;
        const double phs0 = c.derivs[0].phs;
        bool ok = true;
        for (int h = 0; h < c.numharm; h++) {
            accelgpu_rderivs d = c.derivs[h];
            const double amp = std::sqrt(d.pow / (nph > 0.0f ? (double) nph : d.locpow));
            const double corr = phs0 - std::fmod((h + 1.0) * phs0, kTwoPi);
            cr += amp * std::cos(d.phs + corr);
            ci += amp * std::sin(d.phs + corr);
            if (nph > 0.0f)
                d.locpow = nph;
            accelgpu_fourierprops q;
            memset(&q, 0, sizeof(q));
            int rc = calc_props(d, c.hirs[h], c.hizs[h], 0.0, &q);
            if (rc)
                return rc;
            o.harm_pow[h] = printed_power(centred(nice_output_2(q.pow, q.powerr, &ok), 15));
            o.harm_phs[h] = leading_number(centred(nice_output_2(q.phs, q.phserr, &ok), 12));
        }
        o.numharm = c.numharm;
        // each cell as the table holds it: centred in its column and cut at the column's width
        o.sigma = leading_number(centred(fmt("%.2f", c.sigma), 5));
        o.ipow = leading_number(centred(fmt("%.2f", c.power), 6));
        o.cpow = leading_number(centred(fmt("%.2f", cr * cr + ci * ci), 8));
        o.r = leading_number(centred(nice_output_2(c.r, 0.5f / c.numharm, &ok), 15));
        o.z = leading_number(centred(nice_output_2(c.z, 2.0f / c.numharm, &ok), 11));
        if (!ok) {
            set_error("This is an undefined condition in nice_output2().");
            return ACCELGPU_ERR_ARG;
        }
    }
    return ACCELGPU_OK;
}

} // namespace accelhost
