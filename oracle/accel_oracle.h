/*
 * accel_oracle.h -- CPU restatement of the accelsearch hot path (TEST INFRASTRUCTURE).
 *
 * This is the checker the HIP library is compared against.  It is NOT part of
 * the shipped product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.
 *
 * PARITY STATUS: "parity unpinned" for the path as a whole -- the reference
 * ships no golden vectors or assertions for accelsearch, and the reference
 * program cannot be built in this image (every file that includes presto.h
 * needs FFTW's fftw3.h, which is absent).  Pieces that ARE pinned against the
 * reference's own code, compiled unmodified into oracle/_ref/libprestoref.so:
 *   fresnl()  src/fresnl.c:175-216      <-> orc_fresnel()
 *   median()  src/median.c:12-69        <-> orc_median()
 *   cdfchi()/cdfnor() src/dcdflib.c     <-> orc_power_for_sigma()/orc_candidate_sigma()
 *   amoeba()  src/amoeba.c:18-75        <-> orc_amoeba() (the simplex of the candidate optimisation)
 * and the reference's generated command-line parsers, also unmodified
 * (oracle/_ref/libaccelsearch_cmd.so, librednoise_cmd.so), which pin the product's flag
 * names, defaults and limits:
 *   parseCmdline() src/accelsearch_cmd.c:1028-1203 <-> accelsearch_hip's command line
 *   parseCmdline() src/rednoise_cmd.c              <-> the rednoise defaults 6 / 100 / 6.0
 * The z-response is additionally pinned against direct quadrature of its
 * defining integral, and the FFT against a double-precision DFT.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference tree).
 */
#ifndef ACCEL_ORACLE_H
#define ACCEL_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_USELEN 7470      /* include/accel.h:13  ACCEL_USELEN     */
#define ORC_NUMBETWEEN 2     /* include/accel.h:19  ACCEL_NUMBETWEEN */
#define ORC_DR 0.5           /* include/accel.h:21 */
#define ORC_RDR 2            /* include/accel.h:23 */
#define ORC_DZ 2             /* include/accel.h:25 */
#define ORC_RDZ 0.5          /* include/accel.h:27 */
#define ORC_CLOSEST_R 15.0   /* include/accel.h:29 */
#define ORC_MAXSTAGES 6

typedef struct { float r, i; } orc_cf32;

/* One cell above threshold, in the order search_ffdotpows() visits them. */
typedef struct {
    float power;         /* summed power                                    */
    float sigma;         /* (float) candidate_sigma()                       */
    int numharm;         /* 1,2,4,8,16,32                                   */
    int block;           /* index of the search block (0-based)             */
    int zind;            /* row in the block (z = zlo + 2*zind)             */
    int added;           /* 1 if insert_new_accelcand() set *added          */
    long long rind;      /* absolute half-bin column: 2*block_rlo + jj      */
    double r;            /* (rlo + jj*0.5)/numharm                          */
    double z;            /* (zlo + zind*2)/numharm                          */
} orc_hit;

typedef struct {
    float power, sigma;
    int numharm;
    double r, z;
} orc_cand;

typedef struct orc_obs {
    /* search set-up: the part of accelobs (include/accel.h:33-71) the path uses */
    long long N, numbins, lobin, highestbin;
    int fftlen, numharmstages, numz, zmax, norm_type, inmem;
    double dt, T, rlo, rhi, zlo, zhi;
    float nph, sigma;
    float powcut[ORC_MAXSTAGES];
    long long numindep[ORC_MAXSTAGES];
    const orc_cf32 *fft;     /* not owned */
    float *ffdotplane;       /* numz x rlen, owned */
    long long rlen;          /* (highestbin + USELEN) * 2 */
    /* kernel bank for the fundamental: numz kernels of fftlen points */
    int halfwidth;           /* kern[0].kern_half_width */
    orc_cf32 *kern;          /* numz * fftlen, owned */
    int nthreads;            /* OpenMP threads for the block loops (1 = serial) */
} orc_obs;

/* ---- scalar helpers ---------------------------------------------------- */
void orc_fresnel(double x, double *S, double *C);
int orc_z_resp_halfwidth(double z);
long long orc_next2_to_n(long long x);
int orc_calc_required_z(double harm_fract, double zfull);
double orc_calc_required_r(double harm_fract, double rfull);
int orc_index_from_z(double z, double loz);
int orc_calc_fftlen(int numharm, int harmnum, int max_zfull);
float orc_median(float *arr, int n);
double orc_candidate_sigma(double power, int numsum, double numtrials);
double orc_power_for_sigma(double sigma, int numsum, double numtrials);
double orc_chi2_logp(double chi2, int dof);
double orc_equivalent_gaussian_sigma(double logp);

/* ---- kernels / FFT ----------------------------------------------------- */
void orc_fft(orc_cf32 *data, long n, int isign);
void orc_gen_r_response(double roffset, int numbetween, int numkern, orc_cf32 *out);
void orc_gen_z_response(double roffset, int numbetween, double z, int numkern, orc_cf32 *out);
void orc_place_complex_kernel(const orc_cf32 *kernel, int numkernel, orc_cf32 *result, int numresult);
void orc_init_kernel(int z, int fftlen, orc_cf32 *out, int *halfwidth);

/* ---- search ------------------------------------------------------------ */
/* create_accelobs(), search-range part.  Returns 0 or a negative error code
 * (the reference prints and exits). */
int orc_obs_init(orc_obs *obs, const orc_cf32 *fft, long long numbins, long long N,
                 double dt, int zmax, int numharm, double sigma, double flo, double fhi,
                 long long lobin, int norm_type, int photon, int alloc_plane);
void orc_obs_free(orc_obs *obs);
void orc_build_kernels(orc_obs *obs);
/* subharm_ffdot_plane(1,1,startr,startr+3735-0.5): powers is numz*USELEN. */
void orc_fund_block(const orc_obs *obs, double startr, float *powers, int *rlo_out);
void orc_fund_to_ffdotplane(orc_obs *obs, const float *powers, int rlo);
void orc_inmem_add_ffdotpows(const orc_obs *obs, float *powers, int rlo, int numharm, int harmnum);
/* Whole main loop of accelsearch.c:136-221 (in-memory semantics).
 * hits: caller buffer of max_hits; *n_hits gets the TRUE count (may exceed max).
 * cands: final list in list order (r-sorted as the reference keeps it). */
int orc_run(orc_obs *obs, orc_hit *hits, long long max_hits, long long *n_hits,
            orc_cand *cands, int max_cands, int *n_cands);
/* Number of search blocks / pre-population blocks for this set-up. */
int orc_num_search_blocks(const orc_obs *obs);
int orc_num_prepop_blocks(const orc_obs *obs);
/* Only the plane-generation half of orc_run (pre-pop + all search blocks). */
void orc_fill_plane(orc_obs *obs);
/* Only the sum+threshold half, on whatever obs->ffdotplane holds. */
int orc_search_plane(orc_obs *obs, orc_hit *hits, long long max_hits, long long *n_hits,
                     orc_cand *cands, int max_cands, int *n_cands);

/* ---- standard (not in-memory) mode ---------------------------------------- */
void orc_subharm_block(const orc_obs *obs, int numharm, int harmnum, double fullrlo,
                       double fullrhi, float *powers, int *rlo_out, int *zlo_out, int *numrs_out,
                       int *numzs_out, unsigned short *rinds);
void orc_add_ffdotpows(const orc_obs *obs, float *fund, const float *sub, int sub_zlo,
                       int sub_numrs, const unsigned short *rinds, int numharm, int harmnum);
int orc_run_standard(orc_obs *obs, orc_hit *hits, long long max_hits, long long *n_hits,
                     orc_cand *cands, int max_cands, int *n_cands);

/* ---- time-series input ----------------------------------------------------- */
/* In-place forward real FFT into PRESTO's packed layout (bin 0 = DC, Nyquist). */
void orc_realfft_forward(float *idata, long n);
void orc_deredden(orc_cf32 *fft, int numamps);
void orc_zapbirds(orc_cf32 *fft, long numbins, const double *lobins, const double *hibins, int n);
void orc_rednoise(orc_cf32 *fft, long numamps, int startwidth, int endwidth, double endfreq,
                  double Tsec);

/* ---- after the search: candidate optimisation (post_oracle.c) ---------------- */
#define ORC_MAXHARM 32
typedef struct {           /* rderivs, include/presto.h:161-169 */
    double pow, phs, dpow, dphs, d2pow, d2phs, locpow;
} orc_rderivs;
typedef struct {           /* accelcand with its optimisation results, include/accel.h:73-83 */
    float power, sigma;
    int numharm, pad;
    double r, z;
    double pows[ORC_MAXHARM], hirs[ORC_MAXHARM], hizs[ORC_MAXHARM];
    orc_rderivs derivs[ORC_MAXHARM];
} orc_optcand;
int orc_z_resp_halfwidth_high(double z);
void orc_rz_interp(const orc_cf32 *data, long numdata, double r, double z, int kern_half_width,
                   orc_cf32 *ans);
double orc_get_localpower3d(const orc_cf32 *data, long numdata, double r, double z);
void orc_get_derivs3d(const orc_cf32 *data, long numdata, double r, double z, double localpower,
                      orc_rderivs *res);
void orc_amoeba(double p[3][2], double *y, double ftol, double (*funk)(double *), int *nfunk);
void orc_max_rz(const orc_cf32 *data, long numdata, int nharm, int r_offset, double rin, double zin,
                int polish, double *rout, double *zout, orc_rderivs *derivs, double *power);
void orc_optimize_cand(const orc_obs *obs, int polish, orc_optcand *c);
int orc_sort_and_eliminate(orc_cand *cands, int n, int eliminate);
typedef struct {           /* fourierprops, include/presto.h:171-189 (88 bytes) */
    double r; float rerr; double z; float zerr; double w; float werr;
    float pow, powerr, sig, rawpow, phs, phserr, cen, cenerr, pur, purerr, locpow;
} orc_fourierprops;
typedef struct {           /* rzwerrs, include/presto.h:234-247 */
    double p, pd, pdd, f, fd, fdd;
    float perr, pderr, pdderr, ferr, fderr, fdderr;
} orc_rzwerrs;
int orc_calc_props(const orc_rderivs *data, double r, double z, double w, orc_fourierprops *res);
void orc_calc_rzwerrs(const orc_fourierprops *props, double T, orc_rzwerrs *res);
int orc_nice_output_1(char *output, double val, double err, int len);
int orc_nice_output_2(char *output, double val, double err, int len);
int orc_write_accel_files(const orc_obs *obs, orc_optcand *cands, int numcands,
                          const char *accelnm, const char *candnm, const char *inftext);

#ifdef __cplusplus
}
#endif
#endif
