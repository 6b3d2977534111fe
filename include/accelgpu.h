/*
 * accelgpu.h -- C ABI of libaccelgpu.so: the MI355X (gfx950) implementation of
 * the accelsearch hot path (F-Fdot plane generation, in-memory harmonic
 * summing, thresholding), plus the host-side candidate bookkeeping that the
 * reference keeps inside search_ffdotpows().
 *
 * Plain C types only.  Every function returns ACCELGPU_OK (0) or a negative
 * code; accelgpu_last_error() gives the text for the calling thread.  The
 * reference prints and exit()s on errors (e.g. src/accel_utils.c:895-898,
 * 1571-1573); a C host maps non-zero returns to that behaviour.
 *
 * "ref:" comments name the reference interface each entry point stands in for
 * (paths relative to the reference tree).
 *
 * Threading: one context is used from one host thread at a time.  Work is
 * queued on the context's HIP stream; calls that return host data synchronise
 * that stream.
 */
#ifndef ACCELGPU_H
#define ACCELGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACCELGPU_OK            0
#define ACCELGPU_ERR_ARG      -1   /* bad argument / call order                       */
#define ACCELGPU_ERR_HIP      -2   /* a HIP runtime call failed                       */
#define ACCELGPU_ERR_NOMEM    -3   /* device or host allocation failed                */
#define ACCELGPU_ERR_OVERFLOW -4   /* more hits than the caller's buffer can hold     */
#define ACCELGPU_ERR_RANGE    -5   /* search range errors of create_accelobs()        */

#define ACCELGPU_USELEN    7470    /* ref: include/accel.h:13 ACCEL_USELEN            */
#define ACCELGPU_MAXSTAGES 6       /* numharm 1,2,4,8,16,32                           */

typedef struct accelgpu_ctx accelgpu_ctx;

/* One F-Fdot cell whose (summed) power exceeded powcut[stage].
 * ref: the body of the scan in search_ffdotpows(), src/accel_utils.c:1272-1283.
 *   numharm = 1 << stage
 *   r = (rind * 0.5) / numharm          z = (zlo + 2 * zind) / numharm          */
typedef struct accelgpu_hit {
    float   power;
    int32_t stage;
    int32_t zind;
    int32_t block;      /* search block index, (rind - 2*rlo) / ACCELGPU_USELEN      */
    int64_t rind;       /* absolute column in half bins: 2*block_rlo + jj            */
} accelgpu_hit;

/* ref: struct accelcand, include/accel.h:73-83 (search-time fields). */
typedef struct accelgpu_cand {
    float   power;
    float   sigma;
    int32_t numharm;
    int32_t added_order; /* index of the hit that last (re)wrote this candidate      */
    double  r;
    double  z;
} accelgpu_cand;

/* Derived search set-up.  ref: the search-range part of create_accelobs(),
 * src/accel_utils.c:1524-1670, and struct accelobs, include/accel.h:33-71. */
typedef struct accelgpu_setup {
    int64_t N, numbins, lobin, highestbin;
    int32_t zmax, numz, numharm, numharmstages, fftlen, halfwidth;
    int32_t num_search_blocks, num_prepop_blocks;
    double  dt, T, rlo, rhi, zlo, zhi;
    float   sigma;
    float   powcut[ACCELGPU_MAXSTAGES];
    int64_t numindep[ACCELGPU_MAXSTAGES];
    int64_t rlen;        /* plane row length in floats: (highestbin + USELEN) * 2    */
    int64_t plane_bytes;
} accelgpu_setup;

const char *accelgpu_last_error(void);
const char *accelgpu_version(void);

/* ---- context ----------------------------------------------------------- */
/* Binds a context (own stream, own device buffers) to HIP device `device`. */
int accelgpu_create(accelgpu_ctx **ctx, int device);
void accelgpu_destroy(accelgpu_ctx *ctx);
/* Use the caller's hipStream_t instead of the context's own (NULL restores it). */
int accelgpu_set_stream(accelgpu_ctx *ctx, void *hip_stream);
/* Makes the context's GPU the current device of the calling thread: for a helper thread (a file
 * reader) that calls accelgpu_host_alloc() for the context's uploads.  Every entry point that
 * takes a context does the same for its own thread. */
int accelgpu_bind_thread(accelgpu_ctx *ctx);
/* Waits for everything the context has queued, uploads included. */
int accelgpu_synchronize(accelgpu_ctx *ctx);

/* Launch-shape knobs, by name.  The defaults are the measured optima; a context reads
 * the environment overrides ACCELGPU_<NAME IN CAPITALS> once, in accelgpu_create(), and
 * nothing on the launch path reads the environment afterwards.  Results never depend on
 * a knob (tests compare them bit for bit); only with nothing in flight.
 *   corr_rr        1   plane generation: register-twiddle kernel (0: table form)
 *   corr_zl        1   ... one block per workgroup, loop over templates (0: the reverse)
 *   corr_chunk     0   ... transforms per workgroup (0: 16, fewer on small grids)
 *   corr_tz, _tc  -1,16 ... XCD tile of the grid: owned items x chunks (-1: 4 on large
 *                      grids, plain 2-D grid on small ones; 0: plain grid)
 *   hs_xcd_remap   1   harmonic sum: one contiguous run of columns per XCD
 *   hs_lds         0   ... bytes of dynamic LDS per workgroup (caps the waves per CU)
 *   hs_lean_from   6   ... register-lean kernel from this many stages on
 *   hs_j          -1   ... column pairs per thread (-1: by number of stages)
 *   hs_chunk_blocks 0  fill and sum the plane in chunks of this many search blocks
 *                      (0: the whole plane at once)
 *   std_table      0   standard mode: table-form correlate at the fundamental's length too
 *   std_blocks     0   standard mode: search blocks per pass (0: what the scratch holds)
 *   opt_nt       512   candidate optimisation: threads per workgroup
 *   use_graph      1   searches with a plane of at most 64 MB: the whole launch is captured once
 *                      per (hit buffer, spectrum buffer) as a HIP graph and replayed with one call
 *   corr_lds       0   plane generation: extra bytes of dynamic LDS per workgroup (experiments)
 *   corr_pair      1   plane generation, 8192-point transforms: adjacent outputs stored in pairs
 *   hs_bound       1   the in-memory plane is kept as two planes of 16-bit halves of every power's
 *                      float and the harmonic sum reads the upper halves alone, as upper bounds,
 *                      fetching the lower halves wherever a bound sum crosses a threshold: the
 *                      same hits from half the bytes.  On large planes searched with 2 to 16
 *                      harmonics (8192-point transforms; 16384-point ones, i.e. zmax above 300,
 *                      with 8 or 16 harmonics) a coarser bound runs in front: plane
 *                      generation also writes the largest upper half of every eight columns, in
 *                      steps of four; sums of those (and, for harmonics up to 1/4, of the source
 *                      cells' own upper halves) single out the few (row, four columns) that could
 *                      cross a threshold, and only these are summed (exactly): the same hits again,
 *                      from an eighth of the bytes.  0: float plane and float sums; 2 / 3: the bounded
 *                      sum alone with one / several columns per lane; 4: the coarse bound wherever
 *                      it can run.  Changing between 0 and the others sets the search up afresh
 *                      (an empty plane).  Read-only: "hs_bound_active" (1 if the current search
 *                      keeps its plane as halves), "hs_prefilter_active" (1 if the coarse bound
 *                      runs), "pf_listed" ((row, four columns) the last search had to sum
 *                      exactly; -1: too many, the bounded sum ran over the whole plane instead)
 *   pf_list_cap    0   capacity of that list in entries (0: 1/64 of all there are)
 *   pf_limit_ppm 15000 entries per million (row, four columns) beyond which the list costs more than the
 *                      bounded sum over the whole plane it replaces (less where the kernel's own lists hold
 *                      less: two groups per lane, many rows); pf_sample 1: a sample of 2^15
 *                      (row, four columns) is asked before the walk (30 microseconds), and a list
 *                      that would be longer is not started; pf_memory 1: a context whose list was
 *                      given up leaves the coarse bound out of its next 4 launches (8, 16 ... 64 when
 *                      the one after gives up again).  Read-only: "pf_last_listed" (the last
 *                      COLLECTED launch: entries listed, -1 given up, -2 the coarse bound did not
 *                      run), "pf_skip" (launches still to go without it).  hs_bound 4 ignores the
 *                      limit and the memory.
 *   pf_groups      0   groups of four columns a lane of the coarse bound's kernel takes: 0 two with up
 *                      to 8 harmonics and one with 16 (two cost it its waves); 1 / 2: always that many
 *   pf_nt          0   threads per workgroup of that kernel: 0 by the size of the search (1024 where
 *                      that still makes several workgroups per CU); 256 / 512 / 1024: that many
 *   hits_cap   2^20    records in each device hit buffer (set: reallocates both; they
 *                      grow on their own when a search overflows them)
 * ref: none (the reference's only run-time knob is -ncpus). */
int accelgpu_set_tuning(accelgpu_ctx *ctx, const char *name, int value);
int accelgpu_get_tuning(accelgpu_ctx *ctx, const char *name, int *value);

/* ---- host-side scalar helpers (no GPU needed) --------------------------- */
/* Can the coarse bound in front of the harmonic sum run on a search of this shape (tuning knobs aside)?
 * numharmstages 2..5 unless `forced` (hs_bound = 4: 2..6), at least 2^22 search columns unless forced,
 * and a plane row of fewer than 2^32 columns in any case: its kernels count columns, groups and list
 * entries in 32 bits (the sums proper index in 64).  The kernels' workgroup-to-XCD mappings assume the
 * 8 XCDs of an MI355X.
 * ref: none (a device-side strategy; the reference's scan is src/accel_utils.c:1259-1299). */
int accelgpu_coarse_bound_applies(int numharmstages, int64_t search_columns, int64_t rlen, int fftlen, int forced);
/* ref: z_resp_halfwidth(z, LOWACC), src/responses.c:29-65 */
int accelgpu_z_resp_halfwidth(double z);
/* ref: calc_fftlen(), src/accel_utils.c:119-132 */
int accelgpu_calc_fftlen(int numharm, int harmnum, int max_zfull);
/* ref: candidate_sigma(), src/characteristics.c:548-568 */
double accelgpu_candidate_sigma(double power, int numsum, double numtrials);
/* ref: power_for_sigma(), src/characteristics.c:571-604 */
double accelgpu_power_for_sigma(double sigma, int numsum, double numtrials);
/* ref: create_accelobs() range/threshold logic, src/accel_utils.c:1524-1641,
 * with the -flo/-fhi semantics of the generated parser (both always present,
 * src/accelsearch_cmd.c:54-60).  Fills `out`; no GPU work. */
int accelgpu_plan(int64_t numbins, int64_t N, double dt, int zmax, int numharm, double sigma,
                  double flo, double fhi, int64_t lobin, accelgpu_setup *out);

/* ---- kernel bank -------------------------------------------------------- */
/* Builds (or reuses, if zmax is unchanged) the bank of FFT'd z-response
 * templates for the fundamental: numz = 2*(zmax/2)+1 kernels of fftlen points.
 * ref: create_subharminfos()/init_subharminfo()/init_kernel()/gen_z_response()/
 * place_complex_kernel(), src/accel_utils.c:135-210, src/responses.c:165-322,
 * src/corr_prep.c:58-80, in-memory branch (fundamental only, :195). */
int accelgpu_build_kernels(accelgpu_ctx *ctx, int zmax);
/* Parity getter: copies kernel `zind` (fftlen complex floats) to the host.
 * ref: kernel.data, include/accel.h:85-92 */
int accelgpu_get_kernel(accelgpu_ctx *ctx, int zind, float *out_complex);

/* ---- input spectrum ----------------------------------------------------- */
/* Uploads numbins complex floats (PRESTO .fft layout); bin 0 of the array is
 * Fourier bin `lobin`.  The spectrum becomes the one the next launch reads.  The
 * library keeps several spectrum buffers: with searches in flight
 * (accelgpu_run_launch) the upload goes to one they do not read and runs beside
 * them, as do the in-place steps below on a spectrum loaded that way.
 * ref: the mmap of the .fft, src/accel_utils.c:1497-1522,
 * and get_fourier_amplitudes(), :848-877. */
int accelgpu_load_fft(accelgpu_ctx *ctx, const float *fft_complex_host, int64_t numbins);
/* Same, for a spectrum already in device memory (not copied, not owned; it must stay
 * valid and unchanged until the search launched on it has been collected). */
int accelgpu_load_fft_device(accelgpu_ctx *ctx, const void *fft_complex_dev, int64_t numbins);
/* Page-locked host memory, the kind accelgpu_load_fft_async() copies from at the full
 * PCIe rate without a staging copy: read the .fft file straight into it.
 * ref: the reference mmaps the file instead (src/accel_utils.c:1497-1522). */
int accelgpu_host_alloc(void **ptr, size_t bytes);
int accelgpu_host_free(void *ptr);
/* accelgpu_load_fft() that returns at once: the copy runs on the copy engine beside
 * whatever searches are in flight, and the launch that reads this spectrum waits for it
 * on the device, so the host never does.  `fft_complex_host` must be page-locked
 * (accelgpu_host_alloc, hipHostMalloc, hipHostRegister) and stay unchanged until the
 * search launched on it has been collected (or accelgpu_synchronize() has returned).
 * ref: as accelgpu_load_fft. */
int accelgpu_load_fft_async(accelgpu_ctx *ctx, const float *fft_complex_host, int64_t numbins);

/* Time-series input: uploads n float samples, transforms them on the device into
 * the packed .fft layout (bin 0 = DC, Nyquist) and, if `deredden` is non-zero,
 * removes red noise; the result becomes the current spectrum (numbins = n/2).
 * n is even, 1024 <= n <= 67108864 (the reference's limit); power-of-two lengths go
 * through the LDS transforms directly, other lengths through a chirp-z
 * convolution on top of them.
 * ref: the .dat branch of create_accelobs(), src/accel_utils.c:1429-1484;
 * realfft(), src/fastffts.c:198-270; deredden(), src/accel_utils.c:1301-1374. */
int accelgpu_load_dat(accelgpu_ctx *ctx, const float *series, int64_t n, int deredden);
/* deredden() on the current spectrum in place (it must be library-owned:
 * accelgpu_load_fft or accelgpu_load_dat).  ref: src/accel_utils.c:1301-1374. */
int accelgpu_deredden(accelgpu_ctx *ctx);
/* Overwrites bin 0 (the packed DC and Nyquist terms) of the current, library-owned
 * spectrum.  A host that searches a time series without -photon calls it with (1, 1)
 * after accelgpu_load_dat(.., deredden), as the reference does so that the lowest
 * blocks' windows do not pick up the DC term.  ref: src/accel_utils.c:1553-1556. */
int accelgpu_set_bin0(accelgpu_ctx *ctx, float re, float im);
/* The stand-alone rednoise filter on the current spectrum in place: median
 * blocks of startwidth*ln(bin) bins up to frequency endfreq (Hz), endwidth
 * beyond; T = N*dt in seconds.  Reproduces the program's output file bin for
 * bin, including its last half block, whose imaginary parts are written from
 * the real parts.  ref: src/rednoise.c:57-188 (defaults 6, 100, 6.0:
 * clig/rednoise_cmd.cli:13-21). */
int accelgpu_rednoise(accelgpu_ctx *ctx, int startwidth, int endwidth, double endfreq, double T);
/* Replaces every bin of the given Fourier-bin ranges (sorted by lobin) by the
 * local noise amplitude (amp, 0), amp^2 = mean of the median powers of the 200
 * bins below and at the top of the range, over ln 2; ranges below bin 301 are
 * left alone, ranges past the top are cut, as the zapbirds program does.
 * In place on the current, library-owned spectrum.
 * ref: zapbirds(), src/zapping.c:29-81; the loop of src/zapbirds.c:293-300. */
int accelgpu_zapbirds(accelgpu_ctx *ctx, const double *lobins, const double *hibins, int32_t n);
/* Reads a birdie list (frequency and width in Hz per line, 'B' prefix for
 * barycentric lines, '#' comments) into sorted bin ranges for an observation of
 * T seconds at average velocity baryv (in units of c).  *n_birds is the number
 * in the file; ACCELGPU_ERR_OVERFLOW if it exceeds max_birds.  Host only.
 * ref: get_birdies(), src/birdzap.c:6-88. */
int accelgpu_read_birds(const char *zapfile, double T, double baryv, double *lobins,
                        double *hibins, int32_t max_birds, int32_t *n_birds);
/* Copies the current spectrum (numbins complex floats) to the host. */
int accelgpu_get_fft(accelgpu_ctx *ctx, float *out_complex, int64_t numbins);

/* ---- search ------------------------------------------------------------- */
/* Fixes the search geometry and thresholds.  `setup` comes from
 * accelgpu_plan() (or is filled by a host that ran create_accelobs() itself:
 * rlo, highestbin, lobin, zmax, numharmstages, powcut[] are what is read).
 * norm_type 0 = block median (default), 1 = -locpow local-power normalisation
 * (corr_loc_pow, src/corr_routines.c:309-366); nph > 0 selects -photon normalisation.
 * Allocates the in-HBM plane.  ref: accelobs fields, include/accel.h:33-71;
 * plane allocation src/accel_utils.c:1652-1670 (sized for HBM, not MAXRAMUSE). */
int accelgpu_set_search(accelgpu_ctx *ctx, const accelgpu_setup *setup, int norm_type, float nph);

/* One long series split over several GPUs: restricts this context to the search
 * blocks [first_block, first_block + num_blocks) of the reference grid.  The
 * following fill generates only the plane blocks those columns and their
 * sub-harmonic look-ups (columns (int)(rind*h/n + 0.5), src/accel_utils.c:1174)
 * fall in -- recomputed locally instead of exchanged; of blocks outside the range only the z
 * rows those look-ups can reach -- and the search covers only those columns; hit block
 * indices stay global.  *blocks_generated
 * (optional) receives the number of plane blocks a fill will compute.
 * accelgpu_set_search() resets the range to all blocks.
 * ref: the loop bounds of src/accelsearch.c:165-168 (the reference has no
 * counterpart; it searches the whole range in one process). */
int accelgpu_set_block_range(accelgpu_ctx *ctx, int32_t first_block, int32_t num_blocks,
                             int32_t *blocks_generated);
/* The same question without a GPU, for whoever divides the blocks among ranks: the plane blocks
 * a fill would compute for that range and the (block, z row)s in them -- of a block that only
 * serves as a source, a harmonic h/n reaches the rows |z| <= h/n * zmax alone, and only those
 * are generated.  `setup` as accelgpu_plan() returns it.
 * ref: none (see accelgpu_set_block_range). */
int accelgpu_block_range_need(const accelgpu_setup *setup, int32_t first_block, int32_t num_blocks,
                              int32_t *blocks_generated, int64_t *rows_generated);

/* Generates the whole fundamental plane: every pre-population block and every
 * search block of the reference grid (startr = 8 + k*3735 below rlo,
 * startr = rlo + k*3735 above).  ref: subharm_ffdot_plane(1,1,...) +
 * fund_to_ffdotplane() over the loops of src/accelsearch.c:143-161,168-188. */
int accelgpu_fill_plane(accelgpu_ctx *ctx);

/* Harmonic summing and thresholding of every search block, from the plane.
 * Writes at most max_hits records, sorted in the order the reference visits
 * cells (block, stage, z row, r); *n_hits is the true count; returns
 * ACCELGPU_ERR_OVERFLOW if it exceeds max_hits (re-run with a larger buffer).
 * ref: inmem_add_ffdotpows() + search_ffdotpows() over src/accelsearch.c:176-214. */
int accelgpu_search_plane(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits,
                          int64_t *n_hits);

/* accelgpu_fill_plane() followed by accelgpu_search_plane().  With a buffer that is too
 * small it returns ACCELGPU_ERR_OVERFLOW and the count in *n_hits, and nothing stays
 * queued: call it again with room for that many.  Refused while launches of the
 * asynchronous pair below are uncollected (so are accelgpu_set_search() and a change of
 * zmax by accelgpu_build_kernels(): queued trials belong to the geometry they were
 * launched with). */
int accelgpu_run(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits);

/* Asynchronous halves of accelgpu_run() for pipelined batches: _launch queues
 * all GPU work of one trial on the stream and returns; _collect waits for it
 * and returns the sorted hits.  At most two launches are in flight; they are
 * collected in launch order.  A launched search keeps reading the spectrum that
 * was current at its launch: in-place changes of that spectrum are refused until
 * it is collected, loading another one is not.
 * A trial leaves the queue only when its hits have been handed over: if _collect
 * returns ACCELGPU_ERR_OVERFLOW (*n_hits > max_hits) the same call with a larger
 * buffer returns them.  A trial with more hits than the device buffer holds is run
 * again by _collect itself into a larger one (regenerating its plane from its own
 * spectrum if a later trial has been launched meanwhile), so the caller never sees
 * that case. */
int accelgpu_run_launch(accelgpu_ctx *ctx);
int accelgpu_run_collect(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits,
                         int64_t *n_hits);

/* Host replay of the candidate list from sorted hits: sigma for every hit and
 * the r-sorted list with 15-bin proximity replacement.
 * `added` (optional, n_hits ints) receives the reference's *added flag per hit
 * (a line in the .txtcand work file).  ref: candidate_sigma() call and
 * insert_new_accelcand(), src/accel_utils.c:1281-1290, 301-381. */
int accelgpu_replay_candidates(const accelgpu_setup *setup, const accelgpu_hit *hits,
                               int64_t n_hits, accelgpu_cand *cands, int64_t max_cands,
                               int64_t *n_cands, float *sigma_out, int32_t *added);

/* How the replay works, never what it returns (process-wide): linear_walk != 0 finds every
 * hit's place in the list by the reference's walk from the head (src/accel_utils.c:318-330)
 * instead of a binary search over the list's running maximum -- the check of one against
 * the other; threads >= 0 fixes the number of host threads that compute sigmas (-1, the
 * default: by the number of hits, at most 8).  ref: none. */
int accelgpu_replay_options(int linear_walk, int threads);

/* ---- after the search: candidate list trimming and optimisation -------------- */
#define ACCELGPU_MAXHARM 32
/* ref: rderivs, include/presto.h:161-169 */
typedef struct accelgpu_rderivs {
    double pow, phs, dpow, dphs, d2pow, d2phs, locpow;
} accelgpu_rderivs;
/* ref: accelcand after optimize_accelcand(), include/accel.h:73-83 */
typedef struct accelgpu_optcand {
    float   power, sigma;        /* summed power from the search; sigma recomputed in double */
    int32_t numharm, pad;
    double  r, z;                /* search-time position (not overwritten, as in the reference) */
    double  pows[ACCELGPU_MAXHARM], hirs[ACCELGPU_MAXHARM], hizs[ACCELGPU_MAXHARM];
    accelgpu_rderivs derivs[ACCELGPU_MAXHARM];
} accelgpu_optcand;
/* Sorts by decreasing sigma (ties: decreasing power; stable) and, if `eliminate`,
 * drops the less significant of harmonically related candidates.  Host only.
 * ref: sort_accelcands(), eliminate_harmonics(), src/accel_utils.c:278-298, 383-462. */
int accelgpu_sort_and_eliminate(accelgpu_cand *cands, int64_t n, int eliminate, int64_t *n_kept);
/* Optimises every candidate against the current spectrum on the GPU: simplex
 * maximisation of the interpolated power (harmonic-summed and normalised by
 * local powers when `polish`, each harmonic on its own otherwise), then local
 * power, power, phase and derivatives per harmonic.  out[i] belongs to cands[i].
 * ref: optimize_accelcand(), src/accel_utils.c:465-525; max_rz_arr_harmonics(),
 * max_rz_arr(), src/maximize_rz.c; amoeba(), src/amoeba.c. */
int accelgpu_optimize_cands(accelgpu_ctx *ctx, const accelgpu_cand *cands, int64_t n, int polish,
                            accelgpu_optcand *out);
/* The same in two halves, so that a host working through many trials keeps going while
 * the device optimises: launch copies the candidates and queues the work, collect
 * waits for it and fills out[0..n) (n as launched).  One optimisation at a time.
 * With searches in flight (accelgpu_run_launch) both this pair and
 * accelgpu_optimize_cands work on the spectrum of the search collected last, not on
 * the one loaded for the next launch, and run beside the queued searches.
 * ref: the per-candidate loop of src/accelsearch.c:240-254, one call at a time there. */
int accelgpu_optimize_launch(accelgpu_ctx *ctx, const accelgpu_cand *cands, int64_t n, int polish);
int accelgpu_optimize_collect(accelgpu_ctx *ctx, accelgpu_optcand *out, int64_t n);

/* ref: fourierprops, include/presto.h:171-189 (88 bytes; the record of .cand files) */
typedef struct accelgpu_fourierprops {
    double r;  float rerr;
    double z;  float zerr;
    double w;  float werr;
    float  pow, powerr, sig, rawpow, phs, phserr, cen, cenerr, pur, purerr, locpow;
} accelgpu_fourierprops;
/* Signal properties from power, phase and their derivatives.  Host only.
 * ref: calc_props(), src/characteristics.c:193-256. */
int accelgpu_calc_props(const accelgpu_rderivs *derivs, double r, double z, double w,
                        accelgpu_fourierprops *out);
/* "value(error)" as in journals with 1 or 2 error digits, left-justified, into
 * out (at most outlen-1 characters).  Host only.
 * ref: nice_output_1(), nice_output_2(), src/output.c:6-219. */
int accelgpu_nice_output(int error_digits, double val, double err, char *out, int outlen);
/* Writes the two result files of a search from the optimised candidates (in the
 * order given: the caller passes them sorted as accelgpu_sort_and_eliminate()
 * left them): the ACCEL text table of fundamentals and harmonics with `inf_text`
 * (the contents of the .inf file, may be NULL) appended, and the .cand file of
 * fourierprops.  nph > 0 selects the -photon columns.  The pulsar-database notes
 * of the reference are not produced (as for telescope "None"); the
 * candidate-against-candidate notes are.  Host only.
 * ref: src/accelsearch.c:256-286; output_fundamentals(), output_harmonics(),
 * src/accel_utils.c:565-835. */
int accelgpu_write_accel_files(const accelgpu_setup *setup, const accelgpu_optcand *cands,
                               int64_t n, float nph, const char *accel_path,
                               const char *cand_path, const char *inf_text);

/* The work file of a search (<root>_ACCEL_<zmax>.txtcand): one line per hit that the candidate list took,
 * in the order it took them -- hits, sigma and added as accelgpu_replay_candidates() returned them.  Host only.
 * ref: the fprintf in search_ffdotpows(), src/accel_utils.c:1290-1294. */
int accelgpu_write_txtcand(const accelgpu_setup *setup, const accelgpu_hit *hits, const float *sigma,
                           const int32_t *added, int64_t n, const char *path);

/* What the sifting step reads back from an ACCEL file for one fundamental, taken from
 * the optimised candidate at the precision the file prints it with (so that a list
 * built in memory equals one parsed from the file): sigma, summed and coherent power
 * ("%.2f"), r and z (two error digits), and per harmonic the normalised power and the
 * phase as their value(error) strings read.  Host only.
 * ref: the parser of lib/python/sifting.py:1099-1200 over the table written by
 * output_fundamentals()/output_harmonics(), src/accel_utils.c:565-835. */
typedef struct accelgpu_siftrec {
    double sigma, ipow, cpow, r, z;
    int32_t numharm, pad;
    double harm_pow[ACCELGPU_MAXHARM];
    double harm_phs[ACCELGPU_MAXHARM];
} accelgpu_siftrec;
int accelgpu_sift_records(const accelgpu_optcand *cands, int64_t n, float nph, accelgpu_siftrec *out);

/* ---- per-block forms (hosts that keep the reference's call sequence) ------ */
/* Copies a host block (numz*USELEN floats, [z][r]) into the in-HBM plane at
 * column 2*rlo.  ref: fund_to_ffdotplane(), src/accel_utils.c:1076-1090. */
int accelgpu_fund_to_ffdotplane(accelgpu_ctx *ctx, const float *powers, int32_t rlo);
/* powers[z][j] += plane[zind(z)][(int)((2*rlo+j)*harmnum/numharm + 0.5)] for one
 * sub-harmonic, on the GPU; `powers` is the host block, updated in place.
 * ref: inmem_add_ffdotpows(), src/accel_utils.c:1160-1208. */
int accelgpu_inmem_add_ffdotpows(accelgpu_ctx *ctx, float *powers, int32_t rlo, int numharm,
                                 int harmnum);
/* Threshold scan of a host block against powcut[log2 numharm] on the GPU; hits
 * come back in the reference's scan order (z row, then r).
 * ref: the scan of search_ffdotpows(), src/accel_utils.c:1272-1283. */
int accelgpu_search_ffdotpows(accelgpu_ctx *ctx, const float *powers, int32_t rlo, int numharm,
                              accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits);

/* ---- standard (not in-memory) mode ----------------------------------------- */
/* Same as accelgpu_set_search() but without the plane: for searches whose plane
 * does not fit (the reference's rule is MAXRAMUSE, src/accel_utils.c:1652-1670)
 * or hosts that keep the reference's default mode.  The in-memory entry points
 * (fill/search/run, fund_to_ffdotplane, inmem_add_ffdotpows, the plane getters)
 * then return ACCELGPU_ERR_ARG. */
int accelgpu_set_search_standard(accelgpu_ctx *ctx, const accelgpu_setup *setup, int norm_type,
                                 float nph);
/* The F-Fdot block of harmonic harmnum/numharm for the fundamental columns
 * fullrlo .. fullrhi (Fourier bins): own data window, own normalisation, own bank
 * of templates (built on first use and cached per harmonic).  `powers` receives
 * numzs*numrs floats, [z][r] (room for powers_cap floats; at most
 * numz*(ACCELGPU_USELEN+2)); `rinds` (ACCELGPU_USELEN entries, may be NULL, not
 * written for numharm 1) is the column look-up the reference stores in
 * subharminfo.rinds.  ref: subharm_ffdot_plane(), src/accel_utils.c:879-1056;
 * init_subharminfo(), :160-178. */
int accelgpu_subharm_ffdot_plane(accelgpu_ctx *ctx, int numharm, int harmnum, double fullrlo,
                                 double fullrhi, float *powers, int64_t powers_cap, int32_t *rlo,
                                 int32_t *zlo, int32_t *numrs, int32_t *numzs, uint16_t *rinds);
/* fund[z][j] += sub[zind(z)][rinds[j]] on the GPU; `fund` (numz*USELEN floats) is
 * updated in place.  ref: add_ffdotpows(), src/accel_utils.c:1119-1134. */
int accelgpu_add_ffdotpows(accelgpu_ctx *ctx, float *fund, const float *sub, int32_t sub_zlo,
                           int32_t sub_numrs, int32_t sub_numzs, const uint16_t *rinds,
                           int numharm, int harmnum);
/* The whole main loop in standard mode, device resident: for every search block
 * the fundamental, every sub-harmonic block, the adds and the threshold scans,
 * batched over groups of blocks.  Hits as accelgpu_search_plane() returns them.
 * ref: src/accelsearch.c:165-218 with obs.inmem == 0. */
int accelgpu_run_standard(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits,
                          int64_t *n_hits);

/* ---- parity getters ------------------------------------------------------ */
/* One fundamental block computed on its own: numz*USELEN floats, row-major
 * [z][r], exactly the ffdotpows.powers matrix.  ref: subharm_ffdot_plane(1,1,
 * startr, startr+3734.5, ...), src/accel_utils.c:879-1056; *rlo_out = ffdot->rlo. */
int accelgpu_fund_block(accelgpu_ctx *ctx, double startr, float *powers, int32_t *rlo_out);
/* No counterpart in the reference.  For callers that cannot allocate device memory or
 * wait on the host: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  accelgpu_fund_block() for 1..ACCELGPU_NOALLOC_MAX_BLOCKS blocks at
 * once, startr[i] each, of the current spectrum: the powers go to the caller's DEVICE
 * buffer, nblocks*numz*USELEN floats laid out [block][z][r]; rlo_out (host, optional)
 * receives each block's rlo.  The block spectra and descriptors live in a fixed array in
 * the library's device image (1 MiB + 512 B per GPU, held until the process ends), so the
 * call makes no hipMalloc/hipFree and no host-device copy; its kernels are queued on the
 * context's stream and it returns without waiting for them.  More blocks than the array
 * holds is ACCELGPU_ERR_ARG.  Every context on a GPU shares that GPU's array: calls that
 * overlap in time on one GPU race; calls on one stream are ordered and safe. */
#define ACCELGPU_NOALLOC_MAX_BLOCKS 16
int accelgpu_fund_blocks_noalloc(accelgpu_ctx *ctx, const double *startr, int32_t nblocks,
                                 float *powers_dev, int32_t *rlo_out);
/* The block at search index `block` after harmonic stage `stage` (0 = plain
 * fundamental), summed from the plane as it stands.  ref: the `fundamental`
 * matrix after the inmem_add_ffdotpows() calls of stage `stage`,
 * src/accelsearch.c:189-214. */
int accelgpu_summed_block(accelgpu_ctx *ctx, int block, int stage, float *powers);
/* Copies `count` floats of plane row `zind` starting at column `col`. */
int accelgpu_get_plane(accelgpu_ctx *ctx, int zind, int64_t col, int64_t count, float *out);
/* Overwrites the same region (tests feed both implementations one plane). */
int accelgpu_put_plane(accelgpu_ctx *ctx, int zind, int64_t col, int64_t count, const float *in);

/* ---- measurement --------------------------------------------------------- */
/* With profiling on, every launch of the three hot kernels is bracketed by HIP
 * events on the context's stream.  accelgpu_get_timings() synchronises and
 * returns, per kernel, launches since the last reset and their summed ms:
 * index 0 block-prep, 1 correlate (plane generation), 2 harmonic-sum+threshold. */
int accelgpu_set_profiling(accelgpu_ctx *ctx, int on);
int accelgpu_get_timings(accelgpu_ctx *ctx, int64_t launches[3], double total_ms[3], int reset);

#ifdef __cplusplus
}
#endif
#endif
