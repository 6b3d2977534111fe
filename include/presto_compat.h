/*
 * presto_compat.h -- layout-compatible declarations of the PRESTO types that
 * cross the accel_utils.c function seam, for building presto_amd/shim/accel_shim.c
 * outside a PRESTO source tree (compile check, tests).  Inside PRESTO the shim
 * includes the project's own "accel.h" instead and this file is not used.
 *
 * Field order and types follow the reference exactly (include/accel.h:33-110,
 * include/ransomfft.h:13-18, glib's GSList) because the structs are shared by
 * value with the unmodified host program.
 */
#ifndef PRESTO_COMPAT_H
#define PRESTO_COMPAT_H
#include <stdio.h>

typedef struct FCOMPLEX { float r, i; } fcomplex;            /* include/ransomfft.h:13-18 */
typedef struct _GSList { void *data; struct _GSList *next; } GSList; /* glib gslist.h */
typedef struct rderivs_opaque rderivs;                        /* only used through pointers */

typedef struct accelobs {                                     /* include/accel.h:33-71 */
    long long N, numbins, lobin, highestbin;
    int fftlen, numharmstages, numz, numbetween, numzap, dat_input, mmap_file, inmem, norm_type;
    double dt, T, rlo, rhi, dr, zlo, zhi, dz, baryv;
    float nph, sigma;
    float *powcut;
    float *ffdotplane;
    double *lobins, *hibins;
    long long *numindep;
    FILE *fftfile, *workfile;
    fcomplex *fft;
    char *rootfilenm, *candnm, *accelnm, *workfilenm;
    int use_harmonic_polishing;
} accelobs;

typedef struct accelcand {                                    /* include/accel.h:73-83 */
    float power, sigma;
    int numharm;
    double r, z;
    double *pows, *hirs, *hizs;
    rderivs *derivs;
} accelcand;

typedef struct kernel {                                       /* include/accel.h:85-92 */
    int z, fftlen, numgoodbins, numbetween, kern_half_width;
    fcomplex *data;
} kernel;

typedef struct subharminfo {                                  /* include/accel.h:94-101 */
    int numharm, harmnum, zmax, numkern;
    kernel *kern;
    unsigned short *rinds;
} subharminfo;

typedef struct ffdotpows {                                    /* include/accel.h:103-110 */
    int numrs, numzs, rlo, zlo;
    float **powers;
    unsigned short *rinds;
} ffdotpows;

#define ACCEL_USELEN 7470
#define ACCEL_DR 0.5
#define ACCEL_DZ 2
#define ACCEL_CLOSEST_R 15.0

subharminfo **create_subharminfos(accelobs *obs);
void free_subharminfos(accelobs *obs, subharminfo **shis);
void add_ffdotpows(ffdotpows *fundamental, ffdotpows *subharmonic, int numharm, int harmnum);
ffdotpows *subharm_ffdot_plane(int numharm, int harmnum, double fullrlo, double fullrhi,
                               subharminfo *shi, accelobs *obs);
void fund_to_ffdotplane(ffdotpows *ffd, accelobs *obs);
void inmem_add_ffdotpows(ffdotpows *fundamental, accelobs *obs, int numharm, int harmnum);
/* what src/accelsearch.c calls instead with -otheropt (include/accel.h:137-142) */
void fund_to_ffdotplane_trans(ffdotpows *ffd, accelobs *obs);
void inmem_add_ffdotpows_trans(ffdotpows *fundamental, accelobs *obs, int numharm, int harmnum);
void add_ffdotpows_ptrs(ffdotpows *fundamental, ffdotpows *subharmonic, int numharm, int harmnum);
GSList *search_ffdotpows(ffdotpows *ffdot, int numharm, accelobs *obs, GSList *cands);
void free_ffdotpows(ffdotpows *ffd);
void free_accelcand(void *data, void *user_data);
#endif
