// accelsearch_hip -- command-line front end over libaccelgpu with the search
// flags of PRESTO's accelsearch (reference clig/accelsearch_cmd.cli:13-47):
//
//   accelsearch_hip [-zmax Z] [-numharm H] [-sigma S] [-flo F] [-fhi F] [-lobin B]
//                   [-photon] [-locpow] [-median] [-inmem] [-standard] [-device D]
//                   file.fft | file.dat | file.sdat
//
// A time series (.dat floats, .sdat shorts) is transformed and de-reddened on the
// GPU first, as the reference does on the host (src/accel_utils.c:1429-1484); no
// work file is written for it (src/accel_utils.c:1290).
//
// Reads file.fft + file.inf, runs the Fourier-domain acceleration search on the
// GPU and writes the working candidates file <root>_ACCEL_<zmax>.txtcand with the
// reference's line format (src/accel_utils.c:1291-1294), then sorts the
// candidates, removes harmonically related ones, optimises the rest on the GPU and
// writes <root>_ACCEL_<zmax> (text) and <root>_ACCEL_<zmax>.cand (fourierprops),
// the tail of the reference's main() (src/accelsearch.c:225-292).
// -rlo/-rhi are accepted and ignored, as in the reference (its parser marks
// -flo/-fhi as always present, src/accelsearch_cmd.c:54-60).
// Mode selection follows src/accel_utils.c:1652-1670 with HBM in place of
// MAXRAMUSE: the in-memory search runs when the plane can be allocated (or
// -inmem insists); otherwise, or with -standard, the standard search runs.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "accelgpu.h"

static void die(const char *what)
{
    printf("\n%s: %s\n\n", what, accelgpu_last_error());
    exit(1);
}

// "<key text>  =  <value>" lines of a .inf file (reference src/ioinf.c:121-175)
static bool inf_value(const std::string &path, const char *key, double *out)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f)
        return false;
    char line[512];
    bool ok = false;
    while (fgets(line, sizeof(line), f)) {
        if (strstr(line, key)) {
            const char *eq = strchr(line, '=');
            if (eq) {
                *out = atof(eq + 1);
                ok = true;
            }
            break;
        }
    }
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    int zmax = 200, numharm = 8, device = 0, norm_type = 0;
    long long lobin = 0;
    double sigma = 2.0, flo = 1.0, fhi = 10000.0, baryv = 0.0;
    std::string zaplist;
    bool photon = false, force_inmem = false, force_standard = false, polish = true, harmremove = true;
    bool showopts = false, median = false, otheropt = false;
    int ncpus = 1;
    double rlo = 0.0, rhi = 0.0;
    int nfiles = 0;
    std::string infile;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *flag) {
            if (i + 1 >= argc) {
                printf("\nMissing value for %s\n\n", flag);
                exit(1);
            }
            return argv[++i];
        };
        if (a == "-zmax") zmax = atoi(need("-zmax"));
        else if (a == "-numharm") numharm = atoi(need("-numharm"));
        else if (a == "-sigma") sigma = atof(need("-sigma"));
        else if (a == "-flo") flo = atof(need("-flo"));
        else if (a == "-fhi") fhi = atof(need("-fhi"));
        else if (a == "-lobin") lobin = atoll(need("-lobin"));
        // -rlo/-rhi never take effect in the reference either: its parser marks -flo and
        // -fhi as always present, and they win (src/accel_utils.c:1581-1624)
        else if (a == "-rlo") rlo = atof(need("-rlo"));
        else if (a == "-rhi") rhi = atof(need("-rhi"));
        else if (a == "-ncpus") ncpus = atoi(need("-ncpus"));
        else if (a == "-showopts") showopts = true; // prints what was understood and stops (tests)
        else if (a == "-zaplist") zaplist = need("-zaplist");
        else if (a == "-baryv") baryv = atof(need("-baryv"));
        else if (a == "-device") device = atoi(need("-device"));
        else if (a == "-photon") photon = true;
        else if (a == "-locpow") norm_type = 1;
        else if (a == "-median") median = true;
        else if (a == "-inmem") force_inmem = true;
        else if (a == "-standard") force_standard = true;
        else if (a == "-noharmpolish") polish = false;
        else if (a == "-noharmremove") harmremove = false;
        else if (a == "-otheropt") otheropt = true;
        else if (a[0] == '-') {
            printf("\nUnknown option '%s'\n\n", a.c_str());
            return 1;
        } else {
            infile = a;
            nfiles++;
        }
    }
    // the limits of the reference's generated parser (src/accelsearch_cmd.c:1040-1180)
    if (nfiles != 1) {
        printf("\nthere should be exactly 1 non-option argument (the .fft or .[s]dat file)\n\n");
        return 1;
    }
    if (ncpus < 1 || lobin < 0 || numharm < 1 || numharm > 32 || rlo < 0.0 || rhi < 0.0 || flo < 0.0 || fhi < 0.0) {
        printf("\n-ncpus must be >= 1, -lobin, -rlo, -rhi, -flo and -fhi >= 0 and -numharm in 1..32\n\n");
        return 1;
    }
    auto ends_with = [&](const char *suf) {
        const size_t n = strlen(suf);
        return infile.size() > n && infile.compare(infile.size() - n, n, suf) == 0;
    };
    const bool is_sdat = ends_with(".sdat"), is_dat = !is_sdat && ends_with(".dat");
    const bool dat_input = is_dat || is_sdat;
    if (!dat_input && !ends_with(".fft")) {
        // src/accel_utils.c:1397-1422
        printf("\nInput file ('%s') must be an '.fft' or '.[s]dat' file!\n\n", infile.c_str());
        return 1;
    }
    if (baryv < -0.1 || baryv > 0.1) {
        printf("\n-baryv must be in -0.1..0.1\n\n");
        return 1;
    }
    if (zmax < 0 || zmax > 1200 || sigma < 1.0 || sigma > 30.0) {
        printf("\n-zmax must be in 0..1200 and -sigma in 1..30\n\n");
        return 1;
    }
    if (showopts) {
        printf("ncpus=%d lobin=%lld numharm=%d zmax=%d sigma=%.9g rlo=%.17g rhi=%.17g flo=%.17g fhi=%.17g inmem=%d "
               "photon=%d median=%d locpow=%d zaplist=%s baryv=%.17g otheropt=%d noharmpolish=%d noharmremove=%d file=%s\n",
               ncpus, lobin, numharm, zmax, (double) (float) sigma, rlo, rhi, flo, fhi, (int) force_inmem, (int) photon,
               (int) median, norm_type, zaplist.empty() ? "-" : zaplist.c_str(), baryv, (int) otheropt, (int) !polish,
               (int) !harmremove, infile.c_str());
        return 0;
    }
    const std::string root = infile.substr(0, infile.size() - (is_sdat ? 5 : 4));
    double Nd = 0, dt = 0;
    if (!inf_value(root + ".inf", "Number of bins in the time series", &Nd) ||
        !inf_value(root + ".inf", "Width of each time series bin", &dt)) {
        printf("\nCannot read N and dt from '%s.inf'\n\n", root.c_str());
        return 1;
    }
    FILE *f = fopen(infile.c_str(), "rb");
    if (!f) {
        perror(infile.c_str());
        return 1;
    }
    fseek(f, 0, SEEK_END);
    const long long filebytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    long long numbins = filebytes / 8;
    std::vector<float> fft;
    if (dat_input) {
        const long long nsamp = filebytes / (is_sdat ? 2 : 4);
        if (nsamp > 67108864) {
            printf("\nThe input time series is too large.  Use 'realfft' first.\n\n");
            return 0;
        }
        fft.resize((size_t) nsamp);
        if (is_sdat) {
            std::vector<short> tmp((size_t) nsamp);
            if (fread(tmp.data(), 2, (size_t) nsamp, f) != (size_t) nsamp) {
                printf("\nShort read on '%s'\n\n", infile.c_str());
                return 1;
            }
            for (long long i = 0; i < nsamp; i++)
                fft[(size_t) i] = (float) tmp[(size_t) i];
        } else if (fread(fft.data(), 4, (size_t) nsamp, f) != (size_t) nsamp) {
            printf("\nShort read on '%s'\n\n", infile.c_str());
            return 1;
        }
        numbins = nsamp / 2;
    } else {
        fft.resize((size_t) numbins * 2);
        if (fread(fft.data(), 8, (size_t) numbins, f) != (size_t) numbins) {
            printf("\nShort read on '%s'\n\n", infile.c_str());
            return 1;
        }
    }
    fclose(f);

    printf("\n\n    Fourier-Domain Acceleration Search Routine (MI355X)\n\n");
    const auto t0 = std::chrono::steady_clock::now();
    accelgpu_setup su;
    if (accelgpu_plan(numbins, (long long) Nd, dt, zmax, numharm, sigma, flo, fhi, lobin, &su))
        die("accelgpu_plan");
    printf("Searching with up to %d harmonics summed:\n", su.numharm);
    printf("  f = %.1f to %.1f Hz\n", su.rlo / su.T, su.rhi / su.T);
    printf("  r = %.1f to %.1f Fourier bins\n", su.rlo, su.rhi);
    printf("  z = %.1f to %.1f Fourier bins drifted\n\n", su.zlo, su.zhi);

    accelgpu_ctx *gpu = nullptr;
    if (accelgpu_create(&gpu, device)) die("accelgpu_create");
    if (accelgpu_build_kernels(gpu, su.zmax)) die("accelgpu_build_kernels");
    printf("  Harmonic  1/1  has %3d kernel(s) from z = %4d to %4d,  FFT length = %d\n\n", su.numz,
           -su.zmax, su.zmax, su.fftlen);
    float nph = 0.f;
    if (dat_input) {
        printf("Reading and FFTing the time series...");
        if (accelgpu_load_dat(gpu, fft.data(), (int64_t) fft.size(), 0)) die("accelgpu_load_dat");
        printf("done.\n");
        printf("Removing red-noise...");
        if (accelgpu_deredden(gpu)) die("accelgpu_deredden");
        printf("done.\n\n");
        // Birdie zapping applies to a spectrum made in memory from a time series only
        // (src/accelsearch.c:82-105); ranges are cut at the Nyquist bin as there.
        if (!zaplist.empty()) {
            int32_t nb = 0;
            std::vector<double> lo(1), hi(1);
            int zrc = accelgpu_read_birds(zaplist.c_str(), su.T, baryv, lo.data(), hi.data(), 0, &nb);
            if (zrc == ACCELGPU_ERR_OVERFLOW) {
                lo.resize((size_t) nb);
                hi.resize((size_t) nb);
                zrc = accelgpu_read_birds(zaplist.c_str(), su.T, baryv, lo.data(), hi.data(), nb, &nb);
            }
            if (zrc) die("accelgpu_read_birds");
            printf("Zapping them using a barycentric velocity of %.5gc.\n\n", baryv);
            const double hibin = (double) ((long long) Nd / 2);
            int32_t use = 0;
            for (; use < nb && lo[(size_t) use] < hibin; use++)
                if (hi[(size_t) use] >= hibin)
                    hi[(size_t) use] = hibin - 1;
            if (accelgpu_zapbirds(gpu, lo.data(), hi.data(), use)) die("accelgpu_zapbirds");
        }
        // The reference reads the photon count from bin 0 after de-reddening has
        // set it to 1 (src/accel_utils.c:1313, 1531-1533): so does this.
        if (photon) nph = 1.0f;
        // without -photon bin 0 becomes (1, 1), so that the lowest blocks' windows do not
        // pick up the DC term (src/accel_utils.c:1553-1556)
        else if (accelgpu_set_bin0(gpu, 1.0f, 1.0f)) die("accelgpu_set_bin0");
    } else {
        if (accelgpu_load_fft(gpu, fft.data(), numbins)) die("accelgpu_load_fft");
        if (photon && lobin == 0) nph = fft[0];
    }
    if (photon)
        printf("Normalizing powers using %.0f photons.\n\n", nph);
    printf("Full f-fdot plane would need %.2f GB: ", su.plane_bytes / 1073741824.0);
    int rc = force_standard ? ACCELGPU_ERR_NOMEM : accelgpu_set_search(gpu, &su, norm_type, nph);
    bool inmem = rc == ACCELGPU_OK;
    if (rc == ACCELGPU_ERR_NOMEM && !force_inmem) {
        if (accelgpu_set_search_standard(gpu, &su, norm_type, nph)) die("accelgpu_set_search_standard");
    } else if (rc) {
        die("accelgpu_set_search");
    }
    printf("using %s accelsearch.\n\n", inmem ? "in-memory" : "standard");

    std::vector<accelgpu_hit> hits((size_t) 1 << 20);
    int64_t nhits = 0;
    auto run = inmem ? accelgpu_run : accelgpu_run_standard;
    while ((rc = run(gpu, hits.data(), (int64_t) hits.size(), &nhits)) == ACCELGPU_ERR_OVERFLOW)
        hits.resize((size_t) nhits + 1024);
    if (rc) die(inmem ? "accelgpu_run" : "accelgpu_run_standard");
    std::vector<accelgpu_cand> cands((size_t) nhits + 1);
    std::vector<float> sig((size_t) nhits + 1);
    std::vector<int32_t> added((size_t) nhits + 1);
    int64_t ncands = 0;
    if (accelgpu_replay_candidates(&su, hits.data(), nhits, cands.data(), (int64_t) cands.size(), &ncands,
                                   sig.data(), added.data()))
        die("accelgpu_replay_candidates");

    char work[4096];
    snprintf(work, sizeof(work), "%s_ACCEL_%d.txtcand", root.c_str(), zmax); // as given: the reference names
                                                                                // its files before it rounds zmax up
                                                                                // (src/accel_utils.c:1492-1494, 1526)
    FILE *w = dat_input ? nullptr : fopen(work, "w");
    if (!w && !dat_input) {
        perror(work);
        return 1;
    }
    for (int64_t i = 0; w && i < nhits; i++)
        if (added[i]) {
            const int nh = 1 << hits[i].stage;
            const double rr = hits[i].rind * 0.5 / nh, zz = (su.zlo + hits[i].zind * 2.0) / nh;
            fprintf(w, "%-7.2f  %-7.4f  %-2d  %-14.4f  %-14.9f  %-10.4f\n", hits[i].power, sig[i], nh, rr,
                    rr / su.T, zz);
        }
    if (w) {
        fclose(w);
        printf("  Working candidates in a test format are in '%s'.\n\n", work);
    }

    printf("Found %lld candidates (%lld cells above threshold).\n", (long long) ncands, (long long) nhits);
    if (ncands > 0) {
        int64_t kept = 0;
        if (accelgpu_sort_and_eliminate(cands.data(), ncands, su.numharm > 1 && harmremove, &kept))
            die("accelgpu_sort_and_eliminate");
        if (kept < ncands)
            printf("Removed %lld likely harmonically related candidates.\n", (long long) (ncands - kept));
        ncands = kept;
        std::vector<accelgpu_optcand> opt((size_t) ncands);
        if (accelgpu_optimize_cands(gpu, cands.data(), ncands, polish ? 1 : 0, opt.data()))
            die("accelgpu_optimize_cands");
        std::string inftext;
        if (FILE *fi = fopen((root + ".inf").c_str(), "rb")) {
            char buf[4096];
            size_t got;
            while ((got = fread(buf, 1, sizeof(buf), fi)) > 0)
                inftext.append(buf, got);
            fclose(fi);
        }
        char accelnm[4096], candnm[4096];
        snprintf(accelnm, sizeof(accelnm), "%s_ACCEL_%d", root.c_str(), zmax);
        snprintf(candnm, sizeof(candnm), "%s_ACCEL_%d.cand", root.c_str(), zmax);
        if (accelgpu_write_accel_files(&su, opt.data(), ncands, nph, accelnm, candnm, inftext.c_str()))
            die("accelgpu_write_accel_files");
        printf("Optimized %lld candidates; results in '%s' and '%s'.\n", (long long) ncands, accelnm, candnm);
    } else {
        printf("No candidates above sigma = %.2f were found.\n\n", su.sigma);
    }
    printf("Strongest:\n");
    for (int64_t i = 0; i < std::min<int64_t>(ncands, 20); i++)
        printf("sigma: %-7.4f  pow: %-7.2f  harm: %-2d  r: %-14.4f  z: %-10.4f\n", cands[i].sigma,
               cands[i].power, cands[i].numharm, cands[i].r, cands[i].z);
    printf("\nSearched the following approx numbers of independent points:\n");
    for (int i = 0; i < su.numharmstages; i++)
        printf("  %d harmonic%s  %9lld\n", 1 << i, i ? "s: " : ":  ", (long long) su.numindep[i]);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("\nTiming summary:\n  Total time: %.3f sec\n\n", secs);
    accelgpu_destroy(gpu);
    return 0;
}
