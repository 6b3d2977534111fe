// Host-side code of the library (accel_host.cpp, accel_output.cpp: no GPU calls) under
// AddressSanitizer and UndefinedBehaviorSanitizer: planning, detection statistics, the
// candidate-list replay, list trimming, the birdie-list reader, value(error) formatting,
// signal properties and the result-file writer, driven with random and edge-case inputs.
// Built and run by tests/test_host_sanitizers_cpu.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined
//       -I include -I presto_amd/csrc tests/tools/host_sanitize.cpp
//       presto_amd/csrc/accel_host.cpp presto_amd/csrc/accel_output.cpp -pthread
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "accel_host.hpp"

#define REQUIRE(x) do { if (!(x)) { fprintf(stderr, "REQUIRE failed: %s (line %d): %s\n", #x, __LINE__, accelhost::last_error()); exit(2); } } while (0)

int main(int argc, char **argv)
{
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    std::mt19937_64 rng(12345);
    auto uni = [&](double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); };
    auto irand = [&](int64_t a, int64_t b) { return std::uniform_int_distribution<int64_t>(a, b)(rng); };

    // sizes and index rules
    for (int z = -1300; z <= 1300; z += 7) {
        REQUIRE(accelhost::z_resp_halfwidth(z) >= 16);
        for (double f : {1.0 / 32, 0.5, 15.0 / 16, 1.0})
            (void) accelhost::index_from_z(accelhost::calc_required_z(f, z), -1200.0);
    }
    for (int nh : {1, 2, 4, 8, 16, 32})
        for (int h = 1; h <= nh; h++)
            REQUIRE(accelhost::calc_fftlen(nh, h, 1200) >= 512);

    // detection statistics over the whole range they can see, incl. the asymptotic branches
    for (int nh : {1, 2, 4, 8, 16, 32})
        for (double nt : {1.0, 1e3, 2.4e8, 5.8e9}) {
            for (double p : {-1.0, 0.0, 1e-9, 0.5, 3.0, 14.9 * nh, 15.1 * nh, 40.0 * nh, 1e3, 1e5, 1e7})
                REQUIRE(std::isfinite(accelhost::candidate_sigma(p, nh, nt)));
            for (double s : {1.0, 2.0, 6.0, 30.0})
                REQUIRE(accelhost::power_for_sigma(s, nh, nt) > 0.0);
        }

    // planning: valid set-ups and everything the reference rejects
    accelgpu_setup su;
    REQUIRE(accelhost::plan(1 << 22, 1 << 23, 64e-6, 200, 16, 2.0, 1.0, 1e9, 0, &su) == ACCELGPU_OK);
    REQUIRE(su.numharmstages == 5 && su.numz == 201 && su.num_search_blocks == 1122);
    REQUIRE(accelhost::plan(1 << 19, 1 << 20, 1e-4, 51, 8, 2.0, 1.0, 1e4, 0, &su) == ACCELGPU_OK && su.zmax == 52);
    REQUIRE(accelhost::plan(1 << 19, 1 << 20, 1e-4, 0, 1, 2.0, 0.0, 1e9, 1000, &su) == ACCELGPU_OK);
    REQUIRE(accelhost::plan(0, 1 << 20, 1e-4, 0, 1, 2.0, 1.0, 1e9, 0, &su) != ACCELGPU_OK);
    REQUIRE(accelhost::plan(1 << 19, 1 << 20, 1e-4, 1300, 8, 2.0, 1.0, 1e9, 0, &su) != ACCELGPU_OK);
    REQUIRE(accelhost::plan(1 << 19, 1 << 20, 1e-4, 20, 3, 2.0, 1.0, 1e9, 0, &su) != ACCELGPU_OK);
    REQUIRE(accelhost::plan(1 << 19, 1 << 20, 1e-4, 20, 8, 2.0, 1.0, 1e9, 1 << 19, &su) != ACCELGPU_OK);
    REQUIRE(accelhost::plan(1 << 19, 1 << 20, 1e-4, 20, 8, 2.0, 5000.0, 6000.0, 0, &su) != ACCELGPU_OK);
    REQUIRE(accelhost::plan(4000, 8000, 1e-4, 20, 8, 2.0, 1.0, 1e9, 0, &su) != ACCELGPU_OK || su.num_search_blocks >= 0);

    // candidate-list replay: clustered hits, shuffled hits, both walks, every thread count,
    // an output array that is too small, an invalid stage
    REQUIRE(accelhost::plan(1 << 22, 1 << 23, 64e-6, 200, 16, 2.0, 1.0, 1e9, 0, &su) == ACCELGPU_OK);
    for (int trial = 0; trial < 6; trial++) {
        const int64_t n = trial == 0 ? 0 : irand(1, 30000);
        std::vector<accelgpu_hit> hits((size_t) n);
        const int ncen = (int) irand(1, 200);
        std::vector<double> cen((size_t) ncen);
        for (auto &c : cen) c = uni(1e3, 8e6);
        for (auto &h : hits) {
            h.stage = (int32_t) irand(0, 4);
            h.power = (float) (su.powcut[h.stage] * uni(1.0, 3.0));
            h.zind = (int32_t) irand(0, 200);
            h.rind = (int64_t) (cen[(size_t) irand(0, ncen - 1)] + (double) irand(-40, 40));
            h.block = (int32_t) (h.rind / ACCELGPU_USELEN);
        }
        if (trial % 2)
            std::sort(hits.begin(), hits.end(), [](const accelgpu_hit &a, const accelgpu_hit &b) { return a.rind < b.rind; });
        std::vector<accelgpu_cand> c0((size_t) n + 1), c1((size_t) n + 1);
        std::vector<float> s0((size_t) n + 1), s1((size_t) n + 1);
        std::vector<int32_t> a0((size_t) n + 1), a1((size_t) n + 1);
        int64_t n0 = 0, n1 = 0;
        accelhost::replay_options(1, 1);
        REQUIRE(accelhost::replay(&su, hits.data(), n, c0.data(), n + 1, &n0, s0.data(), a0.data()) == ACCELGPU_OK);
        accelhost::replay_options(0, (int) irand(0, 8));
        REQUIRE(accelhost::replay(&su, hits.data(), n, c1.data(), n + 1, &n1, s1.data(), a1.data()) == ACCELGPU_OK);
        REQUIRE(n0 == n1);
        for (int64_t i = 0; i < n0; i++)
            REQUIRE(c0[(size_t) i].r == c1[(size_t) i].r && c0[(size_t) i].sigma == c1[(size_t) i].sigma &&
                    c0[(size_t) i].added_order == c1[(size_t) i].added_order);
        for (int64_t i = 0; i < n; i++)
            REQUIRE(a0[(size_t) i] == a1[(size_t) i] && s0[(size_t) i] == s1[(size_t) i]);
        accelhost::replay_options(0, -1);
        if (n0 > 1) {
            int64_t nn = 0;
            REQUIRE(accelhost::replay(&su, hits.data(), n, c1.data(), n0 - 1, &nn, nullptr, nullptr) == ACCELGPU_ERR_OVERFLOW);
            REQUIRE(nn == n0);
        }
        // list trimming on what came out
        for (int elim = 0; elim < 2; elim++) {
            std::vector<accelgpu_cand> cc(c0.begin(), c0.begin() + n0);
            const int64_t kept = accelhost::sort_and_eliminate(cc.data(), n0, elim != 0);
            REQUIRE(kept >= (n0 > 0 ? 1 : 0) && kept <= n0);
            for (int64_t i = 1; i < kept; i++)
                REQUIRE(cc[(size_t) i - 1].sigma >= cc[(size_t) i].sigma);
        }
        if (n > 0) {
            hits[0].stage = 7;
            int64_t nn = 0;
            REQUIRE(accelhost::replay(&su, hits.data(), n, c1.data(), n + 1, &nn, nullptr, nullptr) == ACCELGPU_ERR_ARG);
        }
    }

    // birdie lists: comments, barycentric lines, blank lines, garbage, a missing file
    {
        const std::string path = tmp + "/sanitize_birds.zaplist";
        FILE *f = fopen(path.c_str(), "w");
        REQUIRE(f != nullptr);
        fprintf(f, "# comment\n\n60.0 0.1\nB 120.0 0.25\n   \nnot a number here\n0.5 0.01\n1e9 1.0\n");
        for (int i = 0; i < 300; i++)
            fprintf(f, "%.6f %.4f\n", uni(1.0, 4000.0), uni(0.001, 0.5));
        fclose(f);
        int32_t nb = 0;
        std::vector<double> lo(1), hi(1);
        int rc = accelhost::read_birds(path.c_str(), 536.870912, 1e-4, lo.data(), hi.data(), 0, &nb);
        REQUIRE(rc == ACCELGPU_ERR_OVERFLOW && nb > 300);
        lo.resize((size_t) nb);
        hi.resize((size_t) nb);
        REQUIRE(accelhost::read_birds(path.c_str(), 536.870912, 1e-4, lo.data(), hi.data(), nb, &nb) == ACCELGPU_OK);
        for (int32_t i = 1; i < nb; i++)
            REQUIRE(lo[(size_t) i - 1] <= lo[(size_t) i]);
        REQUIRE(accelhost::read_birds((tmp + "/no_such_file.zaplist").c_str(), 1.0, 0.0, lo.data(), hi.data(), nb, &nb) != ACCELGPU_OK);
        remove(path.c_str());
    }

    // value(error) strings over many magnitudes, signs and degenerate errors
    for (int i = 0; i < 20000; i++) {
        const double val = (i % 3 ? 1.0 : -1.0) * std::pow(10.0, uni(-12, 12)) * (i % 97 == 0 ? 0.0 : 1.0);
        const double err = i % 11 == 0 ? 0.0 : std::pow(10.0, uni(-14, 10));
        bool ok = true;
        REQUIRE(!accelhost::nice_output_1(val, err).empty());
        REQUIRE(!accelhost::nice_output_2(val, err, &ok).empty());
    }
    (void) accelhost::nice_output_1(NAN, 1.0);
    (void) accelhost::nice_output_1(1.0, INFINITY);

    // signal properties, sifting records and the result files from random optimised candidates
    {
        const int64_t n = 40;
        std::vector<accelgpu_optcand> oc((size_t) n);
        for (int64_t i = 0; i < n; i++) {
            accelgpu_optcand &c = oc[(size_t) i];
            memset(&c, 0, sizeof(c));
            c.numharm = 1 << irand(0, 5);
            c.power = (float) uni(20, 5000);
            c.sigma = (float) uni(2, 80);
            c.r = uni(100, 4e6);
            c.z = uni(-200, 200);
            for (int h = 0; h < c.numharm; h++) {
                c.pows[h] = uni(0.0, 300.0);
                c.hirs[h] = c.r * (h + 1) + uni(-0.5, 0.5);
                c.hizs[h] = c.z * (h + 1) + uni(-1, 1);
                accelgpu_rderivs &d = c.derivs[h];
                d.pow = c.pows[h] * uni(0.5, 2.0);
                d.phs = uni(-3.2, 3.2);
                d.dpow = uni(-1, 1);
                d.dphs = uni(-1, 1);
                d.d2pow = -std::fabs(uni(0.01, 30.0));
                d.d2phs = uni(-1, 1);
                d.locpow = uni(0.5, 2.0);
                accelgpu_fourierprops fp;
                (void) accelhost::calc_props(d, c.hirs[h], c.hizs[h], 0.0, &fp);
            }
        }
        std::vector<accelgpu_siftrec> sr((size_t) n);
        for (float nph : {0.0f, 12345.0f}) {
            REQUIRE(accelhost::sift_records(oc.data(), n, nph, sr.data()) == ACCELGPU_OK);
            const std::string a = tmp + "/sanitize_ACCEL_200", c = a + ".cand";
            REQUIRE(accelhost::write_accel_files(&su, oc.data(), n, nph, a.c_str(), c.c_str(), "inf text\n") == ACCELGPU_OK);
            REQUIRE(accelhost::write_accel_files(&su, oc.data(), 0, nph, a.c_str(), c.c_str(), nullptr) == ACCELGPU_OK);
            remove(a.c_str());
            remove(c.c_str());
        }
        {   // a local power of zero is refused (the reference prints and exits there)
            accelgpu_optcand bad = oc[0];
            bad.derivs[0].locpow = 0.0;
            REQUIRE(accelhost::sift_records(&bad, 1, 0.f, sr.data()) != ACCELGPU_OK);
        }
        REQUIRE(accelhost::write_accel_files(&su, oc.data(), n, 0.f, (tmp + "/no/such/dir/x").c_str(),
                                             (tmp + "/no/such/dir/x.cand").c_str(), nullptr) != ACCELGPU_OK);
    }
    printf("host sanitizer run ok\n");
    return 0;
}
