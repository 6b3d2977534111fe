/*
 * accel_shim.c -- the accel_utils.c function seam of PRESTO's accelsearch,
 * implemented on libaccelgpu (MI355X).  Linking this file in place of the
 * corresponding functions of src/accel_utils.c lets src/accelsearch.c:116-224
 * run unchanged with every F-Fdot computation on the GPU:
 *
 *   create_subharminfos   src/accel_utils.c:181-210   -> kernel bank on the GPU
 *   free_subharminfos     :225-244
 *   subharm_ffdot_plane   :879-1056   (any harmonic; the fundamental alone in -inmem runs)
 *   fund_to_ffdotplane    :1076-1090  -> plane lives in HBM
 *   inmem_add_ffdotpows   :1160-1208
 *   add_ffdotpows         :1119-1134  (standard mode)
 *   search_ffdotpows      :1259-1299  (GPU threshold scan; sigma and the list stay on the host)
 *   free_ffdotpows        :1112-1117
 *   fund_to_ffdotplane_trans  :1093-1109, inmem_add_ffdotpows_trans  :1211-1256,
 *   add_ffdotpows_ptrs    :1136-1157  (what accelsearch.c calls instead with -otheropt: the same
 *                         results from another memory layout or loop form; the plane's layout is
 *                         invisible behind this seam, so they are the same GPU calls)
 *
 * Errors follow the reference's convention: print, then exit.
 * Build inside PRESTO with its own headers; outside (tests) with
 * -DACCELGPU_SHIM_STANDALONE, which takes the type declarations from
 * include/presto_compat.h and plain malloc for list nodes.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef ACCELGPU_SHIM_STANDALONE
#include "presto_compat.h"
#define NODE_ALLOC() ((GSList *) calloc(1, sizeof(GSList)))
#define NODE_FREE(n) free(n)
#else
#include "accel.h"
#define NODE_ALLOC() g_slist_alloc()
#define NODE_FREE(n) g_slist_free_1(n)
#endif
#include "accelgpu.h"

static accelgpu_ctx *g_ctx = NULL;
static accelgpu_setup g_setup;

static void fail(const char *where)
{
    printf("\n%s: %s\n\n", where, accelgpu_last_error());
    exit(1);
}

static int stage_of(int numharm)
{
    int st = 0;
    while (numharm > 1) {
        numharm >>= 1;
        st++;
    }
    return st;
}

/* Host-side description of one harmonic: the fields accelsearch.c and the
 * functions below read.  The FFT'd templates themselves live on the GPU. */
static void describe_harmonic(int numharm, int harmnum, int zmax_full, subharminfo *shi)
{
    int ii;
    const double frac = (double) harmnum / (double) numharm;
    const int fftlen = accelgpu_calc_fftlen(numharm, harmnum, zmax_full);
    /* calc_required_z(): nearest even z of the scaled zmax */
    const double zs = 0.5 * zmax_full * frac;
    const int zmax = (int) (zs < 0 ? zs - 0.5 : zs + 0.5) * 2;

    shi->numharm = numharm;
    shi->harmnum = harmnum;
    shi->zmax = zmax;
    shi->numkern = (zmax / ACCEL_DZ) * 2 + 1;
    shi->rinds = numharm > 1 ? (unsigned short *) malloc(ACCEL_USELEN * sizeof(unsigned short)) : NULL;
    shi->kern = (kernel *) calloc(shi->numkern, sizeof(kernel));
    for (ii = 0; ii < shi->numkern; ii++) {
        kernel *k = &shi->kern[ii];
        k->z = -zmax + ii * ACCEL_DZ;
        k->fftlen = fftlen;
        k->numbetween = 2;
        k->kern_half_width = accelgpu_z_resp_halfwidth((double) k->z);
        k->numgoodbins = k->fftlen - 4 * k->kern_half_width;
        k->data = NULL;
    }
    if (numharm == 1)
        printf("  Harmonic  1/1  has %3d kernel(s) from z = %4d to %4d,  FFT length = %d\n",
               shi->numkern, -zmax, zmax, fftlen);
    else
        printf("  Harmonic %2d/%-2d has %3d kernel(s) from z = %4d to %4d,  FFT length = %d\n",
               harmnum, numharm, shi->numkern, -zmax, zmax, fftlen);
}

subharminfo **create_subharminfos(accelobs *obs)
{
    int ii, jj;
    const int zmax = (int) obs->zhi;
    subharminfo **shis;

    if (!obs->fft) {
        printf("\nThe GPU search needs the FFT in memory (mmap or .dat input).\n\n");
        exit(1);
    }
    if (!g_ctx && accelgpu_create(&g_ctx, 0))
        fail("accelgpu_create");
    if (accelgpu_build_kernels(g_ctx, zmax))
        fail("accelgpu_build_kernels");
    if (accelgpu_load_fft(g_ctx, (const float *) obs->fft, obs->numbins))
        fail("accelgpu_load_fft");

    memset(&g_setup, 0, sizeof(g_setup));
    g_setup.N = obs->N;
    g_setup.numbins = obs->numbins;
    g_setup.lobin = obs->lobin;
    g_setup.highestbin = obs->highestbin;
    g_setup.zmax = zmax;
    g_setup.numz = obs->numz;
    g_setup.numharmstages = obs->numharmstages;
    g_setup.numharm = 1 << (obs->numharmstages - 1);
    g_setup.dt = obs->dt;
    g_setup.T = obs->T;
    g_setup.rlo = obs->rlo;
    g_setup.rhi = obs->rhi;
    g_setup.zlo = obs->zlo;
    g_setup.zhi = obs->zhi;
    g_setup.sigma = obs->sigma;
    for (ii = 0; ii < obs->numharmstages; ii++) {
        g_setup.powcut[ii] = obs->powcut[ii];
        g_setup.numindep[ii] = obs->numindep[ii];
    }
    /* -inmem: the whole plane in HBM; otherwise per-harmonic blocks and no plane */
    if (obs->inmem ? accelgpu_set_search(g_ctx, &g_setup, obs->norm_type, obs->nph)
                   : accelgpu_set_search_standard(g_ctx, &g_setup, obs->norm_type, obs->nph))
        fail("accelgpu_set_search");

    shis = (subharminfo **) malloc(obs->numharmstages * sizeof(subharminfo *));
    shis[0] = (subharminfo *) calloc(2, sizeof(subharminfo));
    describe_harmonic(1, 1, zmax, &shis[0][0]);
    if (!obs->inmem)
        for (ii = 1; ii < obs->numharmstages; ii++) {
            const int harmtosum = 1 << ii;
            shis[ii] = (subharminfo *) calloc(harmtosum, sizeof(subharminfo));
            for (jj = 1; jj < harmtosum; jj += 2)
                describe_harmonic(harmtosum, jj, zmax, &shis[ii][jj - 1]);
        }
    return shis;
}

void free_subharminfos(accelobs *obs, subharminfo **shis)
{
    int ii, jj;

    if (!obs->inmem)
        for (ii = 1; ii < obs->numharmstages; ii++) {
            for (jj = 1; jj < (1 << ii); jj += 2) {
                free(shis[ii][jj - 1].kern);
                free(shis[ii][jj - 1].rinds);
            }
            free(shis[ii]);
        }
    free(shis[0][0].kern);
    free(shis[0]);
    free(shis);
    if (g_ctx) {
        accelgpu_destroy(g_ctx);
        g_ctx = NULL;
    }
}

ffdotpows *subharm_ffdot_plane(int numharm, int harmnum, double fullrlo, double fullrhi,
                               subharminfo *shi, accelobs *obs)
{
    int ii;
    int32_t rlo = 0, zlo = 0, numrs = 0, numzs = 0;
    ffdotpows *ffd;
    float *block;
    const int64_t cap = (int64_t) obs->numz * (ACCEL_USELEN + 2);

    if (numharm != 1 && obs->inmem) {
        printf("subharm_ffdot_plane(): sub-harmonic blocks are not computed in an in-memory search.\n");
        printf("Exiting.\n\n");
        exit(0);
    }
    ffd = (ffdotpows *) malloc(sizeof(ffdotpows));
    block = (float *) malloc(sizeof(float) * (size_t) cap);
    if (accelgpu_subharm_ffdot_plane(g_ctx, numharm, harmnum, fullrlo, fullrhi, block, cap, &rlo,
                                     &zlo, &numrs, &numzs, numharm > 1 ? shi->rinds : NULL))
        fail("accelgpu_subharm_ffdot_plane");
    ffd->rlo = rlo;
    ffd->zlo = zlo;
    ffd->numrs = numrs;
    ffd->numzs = numzs;
    ffd->rinds = numharm > 1 ? shi->rinds : NULL;
    ffd->powers = (float **) malloc(sizeof(float *) * ffd->numzs);
    for (ii = 0; ii < ffd->numzs; ii++)
        ffd->powers[ii] = block + (size_t) ii * ffd->numrs;
    return ffd;
}

void add_ffdotpows(ffdotpows *fundamental, ffdotpows *subharmonic, int numharm, int harmnum)
{
    if (accelgpu_add_ffdotpows(g_ctx, fundamental->powers[0], subharmonic->powers[0],
                               subharmonic->zlo, subharmonic->numrs, subharmonic->numzs,
                               subharmonic->rinds, numharm, harmnum))
        fail("accelgpu_add_ffdotpows");
}

void free_ffdotpows(ffdotpows *ffd)
{
    free(ffd->powers[0]);
    free(ffd->powers);
    free(ffd);
}

void fund_to_ffdotplane(ffdotpows *ffd, accelobs *obs)
{
    (void) obs;
    if (accelgpu_fund_to_ffdotplane(g_ctx, ffd->powers[0], ffd->rlo))
        fail("accelgpu_fund_to_ffdotplane");
}

void inmem_add_ffdotpows(ffdotpows *fundamental, accelobs *obs, int numharm, int harmnum)
{
    (void) obs;
    if (accelgpu_inmem_add_ffdotpows(g_ctx, fundamental->powers[0], fundamental->rlo, numharm, harmnum))
        fail("accelgpu_inmem_add_ffdotpows");
}

/* -otheropt (src/accelsearch.c:154-155, 184-185, 193-195, 204-206): the reference keeps the plane
 * transposed (r-major) or walks the rows through pointers; the sums are the same numbers.  The plane
 * here lives in HBM in one layout whatever the caller asks for, so these are the calls above.  Without
 * them a host linked against this shim would run the reference's own versions on an obs->ffdotplane
 * that nothing fills. */
void fund_to_ffdotplane_trans(ffdotpows *ffd, accelobs *obs)
{
    fund_to_ffdotplane(ffd, obs);
}

void inmem_add_ffdotpows_trans(ffdotpows *fundamental, accelobs *obs, int numharm, int harmnum)
{
    inmem_add_ffdotpows(fundamental, obs, numharm, harmnum);
}

void add_ffdotpows_ptrs(ffdotpows *fundamental, ffdotpows *subharmonic, int numharm, int harmnum)
{
    add_ffdotpows(fundamental, subharmonic, numharm, harmnum);
}

#ifdef ACCELGPU_SHIM_STANDALONE
void free_accelcand(void *data, void *user_data)
{
    (void) user_data;
    free(data);
}
#endif

static accelcand *new_cand(float power, float sigma, int numharm, double r, double z)
{
    accelcand *c = (accelcand *) calloc(1, sizeof(accelcand));
    c->power = power;
    c->sigma = sigma;
    c->numharm = numharm;
    c->r = r;
    c->z = z;
    return c;
}

/* Same list discipline as the reference's (file-static) insert_new_accelcand(),
 * src/accel_utils.c:301-381: r-ordered walk, replacement inside 15 bins when
 * the newcomer has the higher sigma. */
static GSList *list_insert(GSList *list, float power, float sigma, int numharm, double rr,
                           double zz, int *added)
{
    GSList *cur = list, *prev = NULL, *node;
    double next_diff, prev_diff = ACCEL_CLOSEST_R + 1.0;

    *added = 0;
    if (!list) {
        node = NODE_ALLOC();
        node->data = new_cand(power, sigma, numharm, rr, zz);
        *added = 1;
        return node;
    }
    while (cur->next && ((accelcand *) cur->data)->r < rr) {
        prev = cur;
        cur = cur->next;
    }
    next_diff = fabs(rr - ((accelcand *) cur->data)->r);
    if (prev)
        prev_diff = fabs(rr - ((accelcand *) prev->data)->r);
    if (prev_diff < ACCEL_CLOSEST_R) {
        if (((accelcand *) prev->data)->sigma < sigma) {
            free_accelcand(prev->data, NULL);
            prev->data = new_cand(power, sigma, numharm, rr, zz);
            *added = 1;
        }
        if (next_diff < ACCEL_CLOSEST_R && ((accelcand *) cur->data)->sigma < sigma) {
            free_accelcand(cur->data, NULL);
            if (*added) {           /* the newcomer already sits in prev: drop the neighbour */
                prev->next = cur->next;
                NODE_FREE(cur);
            } else {
                cur->data = new_cand(power, sigma, numharm, rr, zz);
                *added = 1;
            }
        }
    } else if (next_diff < ACCEL_CLOSEST_R) {
        if (((accelcand *) cur->data)->sigma < sigma) {
            free_accelcand(cur->data, NULL);
            cur->data = new_cand(power, sigma, numharm, rr, zz);
            *added = 1;
        }
    } else {
        node = NODE_ALLOC();
        node->data = new_cand(power, sigma, numharm, rr, zz);
        *added = 1;
        if (!cur->next && ((accelcand *) cur->data)->r < rr - ACCEL_CLOSEST_R) {
            cur->next = node;
        } else if (prev) {
            prev->next = node;
            node->next = cur;
        } else {
            node->next = list;
            list = node;
        }
    }
    return list;
}

GSList *search_ffdotpows(ffdotpows *ffdot, int numharm, accelobs *obs, GSList *cands)
{
    static accelgpu_hit *hits = NULL;
    static int64_t cap = 0;
    int64_t n = 0, ii;
    const int stage = stage_of(numharm);
    int rc;

    if (!hits) {
        cap = 1 << 16;
        hits = (accelgpu_hit *) malloc(sizeof(accelgpu_hit) * cap);
    }
    rc = accelgpu_search_ffdotpows(g_ctx, ffdot->powers[0], ffdot->rlo, numharm, hits, cap, &n);
    if (rc == ACCELGPU_ERR_OVERFLOW) {
        cap = n + 1024;
        hits = (accelgpu_hit *) realloc(hits, sizeof(accelgpu_hit) * cap);
        rc = accelgpu_search_ffdotpows(g_ctx, ffdot->powers[0], ffdot->rlo, numharm, hits, cap, &n);
    }
    if (rc)
        fail("accelgpu_search_ffdotpows");
    for (ii = 0; ii < n; ii++) {
        const float pow = hits[ii].power;
        const float sig = (float) accelgpu_candidate_sigma(pow, numharm, (double) obs->numindep[stage]);
        const double rr = ((double) hits[ii].rind * ACCEL_DR) / (double) numharm;
        const double zz = (ffdot->zlo + hits[ii].zind * (double) ACCEL_DZ) / (double) numharm;
        int added = 0;
        cands = list_insert(cands, pow, sig, numharm, rr, zz, &added);
        if (added && !obs->dat_input && obs->workfile)
            fprintf(obs->workfile, "%-7.2f  %-7.4f  %-2d  %-14.4f  %-14.9f  %-10.4f\n", pow, sig,
                    numharm, rr, rr / obs->T, zz);
    }
    return cands;
}
