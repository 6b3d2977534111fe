// accel_host.hpp -- host-side helpers of libaccelgpu (no GPU calls).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "accelgpu.h"

namespace accelhost {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *last_error();

int z_resp_halfwidth(double z);
int calc_required_z(double harm_fract, double zfull);
double calc_required_r(double harm_fract, double rfull);
int index_from_z(double z, double loz);
int calc_fftlen(int numharm, int harmnum, int max_zfull);
double candidate_sigma(double power, int numsum, double numtrials);
double power_for_sigma(double sigma, int numsum, double numtrials);
int plan(int64_t numbins, int64_t N, double dt, int zmax, int numharm, double sigma, double flo,
         double fhi, int64_t lobin, accelgpu_setup *out);
int replay(const accelgpu_setup *s, const accelgpu_hit *hits, int64_t n_hits, accelgpu_cand *cands,
           int64_t max_cands, int64_t *n_cands, float *sigma_out, int32_t *added_out);

// The reference's block grid for a search (src/accelsearch.c:143-161 pre-population blocks anchored at r = 8,
// :165-218 search blocks anchored at rlo), each block as subharm_ffdot_plane() describes it for the fundamental
// (src/accel_utils.c:903-907, 929-934): first bin of its window, bins of data in it, first plane column, and the
// plane columns it owns (pre-population blocks stop where the search blocks begin).
struct BlockGrid {
    std::vector<int64_t> lobin, col;
    std::vector<int> rlo, numdata, ncols;
    int nprepop = 0, nsearch = 0;
    int64_t search_col0 = 0;
};
void block_grid(double rlo, int64_t highestbin, int halfwidth, BlockGrid *g);

// One long series split over several GPUs: what a rank that searches the search blocks [first, first + n) has
// to generate.  Its own blocks, every z row; and of every other block a sub-harmonic look-up of its columns
// falls in -- columns (int)(c*h/2^s + 0.5), rows index_from_z(calc_required_z(h/2^s, z)), src/accel_utils.c:1173-1197
// -- the rows those look-ups can reach (a harmonic h/2^s only looks at rows |z| <= h/2^s * zmax).
// z0[b], z1[b]: the rows [z0, z1) of plane block b (0, 0: none).  Returns the number of blocks with any.
int block_range_need(const accelgpu_setup *s, const BlockGrid &g, int first, int n, std::vector<int> *z0, std::vector<int> *z1,
                     int64_t *rows_generated);

// How replay() works, not what it returns: linear_walk != 0 finds a hit's place in the list by
// the reference's walk from the head instead of the binary search; threads >= 0 fixes the number
// of threads that compute sigmas (-1: by the number of hits).
void replay_options(int linear_walk, int threads);

// Sort by decreasing sigma, optionally drop harmonically related candidates;
// returns the number kept (in place, at the front).
int64_t sort_and_eliminate(accelgpu_cand *cands, int64_t n, bool eliminate);

// Birdie list (frequency and width in Hz per line; 'B' lines are barycentric and
// not Doppler-shifted; '#' comments) as sorted Fourier-bin ranges.
int read_birds(const char *path, double T, double baryv, double *lobins, double *hibins,
               int32_t max_birds, int32_t *n_birds);

// value(error) strings, derived signal properties and the result files
std::string nice_output_1(double val, double err);
std::string nice_output_2(double val, double err, bool *ok);
int calc_props(const accelgpu_rderivs &d, double r, double z, double w, accelgpu_fourierprops *out);
int sift_records(const accelgpu_optcand *cands, int64_t n, float nph, accelgpu_siftrec *out);
int write_txtcand(const accelgpu_setup *su, const accelgpu_hit *hits, const float *sigma, const int32_t *added, int64_t n,
                  const char *path);
int write_accel_files(const accelgpu_setup *su, const accelgpu_optcand *cands, int64_t n, float nph,
                      const char *accel_path, const char *cand_path, const char *inf_text);

} // namespace accelhost
