/*
 * mainloop_test.c -- drives the shim through the same call sequence as the
 * reference's search loop (src/accelsearch.c:116-224, in-memory or standard
 * branch) on a .fft file, and prints the candidate list.  Test program only.
 *   usage: mainloop_test <file.fft> <N> <dt> <zmax> <numharm> <workfile> [inmem=1] [otheropt=0]
 *   otheropt = 1 takes the branches of cmd->otheroptP (src/accelsearch.c:154-155, 184-185, 193-195, 204-206).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "presto_compat.h"
#include "accelgpu.h"

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s file.fft N dt zmax numharm workfile\n", argv[0]);
        return 2;
    }
    accelobs obs;
    accelgpu_setup s;
    memset(&obs, 0, sizeof(obs));
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    const long long numbins = ftell(f) / (long long) sizeof(fcomplex);
    fseek(f, 0, SEEK_SET);
    obs.fft = (fcomplex *) malloc(sizeof(fcomplex) * numbins);
    if (fread(obs.fft, sizeof(fcomplex), numbins, f) != (size_t) numbins) return 2;
    fclose(f);
    /* the set-up values a host gets from create_accelobs() */
    if (accelgpu_plan(numbins, atoll(argv[2]), atof(argv[3]), atoi(argv[4]), atoi(argv[5]), 2.0, 1.0,
                      1e9, 0, &s)) {
        fprintf(stderr, "%s\n", accelgpu_last_error());
        return 2;
    }
    obs.N = s.N; obs.numbins = numbins; obs.lobin = 0; obs.highestbin = s.highestbin;
    obs.fftlen = s.fftlen; obs.numharmstages = s.numharmstages; obs.numz = s.numz;
    obs.numbetween = 2; obs.inmem = argc > 7 ? atoi(argv[7]) : 1; obs.norm_type = 0; obs.mmap_file = 1;
    obs.dt = s.dt; obs.T = s.T; obs.rlo = s.rlo; obs.rhi = s.rhi; obs.dr = 0.5;
    obs.zlo = s.zlo; obs.zhi = s.zhi; obs.dz = 2; obs.sigma = s.sigma;
    obs.powcut = s.powcut; obs.numindep = (long long *) s.numindep;
    obs.workfile = fopen(argv[6], "w");

    const int otheropt = argc > 8 ? atoi(argv[8]) : 0;

    subharminfo **shis = create_subharminfos(&obs);
    GSList *cands = NULL;
    double startr, lastr, nextr;
    ffdotpows *fund;
    for (startr = 8; obs.inmem && startr < obs.rlo; startr = nextr) {     /* pre-population */
        nextr = startr + ACCEL_USELEN * ACCEL_DR;
        lastr = nextr - ACCEL_DR;
        fund = subharm_ffdot_plane(1, 1, startr, lastr, &shis[0][0], &obs);
        if (otheropt)
            fund_to_ffdotplane_trans(fund, &obs);
        else
            fund_to_ffdotplane(fund, &obs);
        free_ffdotpows(fund);
    }
    for (startr = obs.rlo; startr + ACCEL_USELEN * ACCEL_DR < obs.highestbin; startr = nextr) {
        nextr = startr + ACCEL_USELEN * ACCEL_DR;
        lastr = nextr - ACCEL_DR;
        fund = subharm_ffdot_plane(1, 1, startr, lastr, &shis[0][0], &obs);
        cands = search_ffdotpows(fund, 1, &obs, cands);
        if (obs.numharmstages > 1) {
            if (obs.inmem) {
                if (otheropt)
                    fund_to_ffdotplane_trans(fund, &obs);
                else
                    fund_to_ffdotplane(fund, &obs);
            }
            for (int stage = 1; stage < obs.numharmstages; stage++) {
                const int harmtosum = 1 << stage;
                for (int harm = 1; harm < harmtosum; harm += 2) {
                    if (obs.inmem) {
                        if (otheropt)
                            inmem_add_ffdotpows_trans(fund, &obs, harmtosum, harm);
                        else
                            inmem_add_ffdotpows(fund, &obs, harmtosum, harm);
                    } else {
                        ffdotpows *sub = subharm_ffdot_plane(harmtosum, harm, startr, lastr,
                                                             &shis[stage][harm - 1], &obs);
                        if (otheropt)
                            add_ffdotpows_ptrs(fund, sub, harmtosum, harm);
                        else
                            add_ffdotpows(fund, sub, harmtosum, harm);
                        free_ffdotpows(sub);
                    }
                }
                cands = search_ffdotpows(fund, harmtosum, &obs, cands);
            }
        }
        free_ffdotpows(fund);
    }
    fclose(obs.workfile);
    free_subharminfos(&obs, shis);
    for (GSList *p = cands; p; p = p->next) {
        accelcand *c = (accelcand *) p->data;
        printf("CAND %.9g %.9g %d %.17g %.17g\n", c->power, c->sigma, c->numharm, c->r, c->z);
    }
    return 0;
}
