// accelgpu.hip -- C ABI of libaccelgpu.so (see include/accelgpu.h).
//
// Data layout in HBM (one context = one search geometry at a time):
//   fft       complex64[numbins]              the input spectrum (.fft layout)
//   bank      complex64[numz][fftlen]         FFT'd z-response templates
//   blockfft  complex64[nblk][fftlen/2]       forward FFT of each normalised block window
//   plane     float[numz][rlen]               fundamental powers, rlen = 2*(highestbin+7470)
//   hits      {float,u32,i64}[cap]            compacted threshold crossings
// Blocks follow the reference grid: pre-population blocks at r = 8 + k*3735 below
// rlo, search blocks at r = rlo + k*3735.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "accel_host.hpp"
#include "accel_kernels.hpp"
#include "accelgpu.h"

using accelhost::set_error;
using accelk::c32;
using accelk::DevHit;

namespace {

typedef afft::Plan<16, 16, 16, 256> P4K;
typedef afft::Plan<32, 16, 16, 256> P8K;
typedef afft::Plan<32, 16, 32, 512> P16K;


#define HIPCHK(expr)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                      __LINE__);                                                         \
            return ACCELGPU_ERR_HIP;                                                     \
        }                                                                                \
    } while (0)

struct Twiddles {
    c32 *tw1 = nullptr, *tw2 = nullptr;
};

struct PowTables { // the two short tables of afft::TwoLevel, per thread column
    c32 *pw1 = nullptr, *pw2 = nullptr;
};

struct EventPair {
    hipEvent_t a, b;
};

} // namespace

struct accelgpu_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;

    // kernel bank
    int zmax = -1, numz = 0, fftlen = 0, halfwidth = 0;
    c32 *d_bank = nullptr;
    Twiddles tw_inv, tw_fwd_full, tw_fwd_half;
    PowTables pw_inv;

    // spectrum
    const c32 *d_fft = nullptr;
    c32 *d_fft_owned = nullptr;
    size_t fft_cap = 0;
    int64_t numbins = 0;

    // search geometry
    bool have_search = false;
    accelgpu_setup setup{};
    int norm_type = 0;
    float nph = 0.f;
    int nblk = 0, nprepop = 0, nsearch = 0;
    int64_t col0 = 0, ncols = 0; // first search column (2*rlo) and number of search columns
    int64_t *d_blk_lobin = nullptr, *d_blk_col = nullptr;
    int *d_blk_numdata = nullptr, *d_blk_ncols = nullptr;
    size_t blk_cap = 0;
    c32 *d_blockfft = nullptr;
    size_t blockfft_cap = 0;
    float *d_plane = nullptr;
    size_t plane_cap = 0;
    int64_t *d_rowoff = nullptr; // [numz][NH] byte offsets of the sub-harmonic source rows
    size_t rowoff_cap = 0;
    uint32_t *d_rowchg = nullptr; // [numz] bit i: source row of harmonic i differs from row zi-1
    size_t rowchg_cap = 0;

    // hits
    // Two hit buffers so that one trial's hits can be collected while the next
    // trial's kernels are already queued.  d_hits/h_hits/d_count/h_count alias
    // the slot currently being launched into or collected from.
    struct HitSlot {
        DevHit *d_hits = nullptr, *h_hits = nullptr; // device buffer, pinned host mirror
        unsigned int *d_count = nullptr, *h_count = nullptr;
        hipEvent_t done = nullptr;                   // recorded after the slot's last copy
        size_t prefetch = 0;                         // hits copied back speculatively at launch
    } slots[2];
    int head = 0, tail = 0, inflight = 0;            // FIFO of launched, uncollected trials
    hipStream_t copy_stream = nullptr;               // late D2H copies that must not queue behind kernels
    DevHit *d_hits = nullptr;
    DevHit *h_hits = nullptr;
    unsigned int *d_count = nullptr, *h_count = nullptr;
    size_t hits_cap = 0;

    // scratch for the single-block getters
    float *d_tmp = nullptr;
    size_t tmp_cap = 0;
    int64_t *d_one_lobin = nullptr, *d_one_col = nullptr;
    int *d_one_numdata = nullptr, *d_one_ncols = nullptr;
    c32 *d_one_blockfft = nullptr;

    // profiling
    bool profiling = false;
    std::vector<EventPair> ev[3];
    std::vector<EventPair> ev_pool;
    int64_t prof_launches[3] = {0, 0, 0};
    double prof_ms[3] = {0, 0, 0};
};

namespace {

int use(accelgpu_ctx *ctx)
{
    if (!ctx) {
        set_error("null context");
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device));
    return ACCELGPU_OK;
}

template <class T> int dev_reserve(T **p, size_t *cap, size_t want, bool zero, hipStream_t st)
{
    if (*cap >= want && *p)
        return ACCELGPU_OK;
    if (*p)
        HIPCHK(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void **) p, want * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", want * sizeof(T), hipGetErrorString(e));
        return ACCELGPU_ERR_NOMEM;
    }
    *cap = want;
    if (zero)
        HIPCHK(hipMemsetAsync(*p, 0, want * sizeof(T), st));
    return ACCELGPU_OK;
}

template <class P> int make_twiddles(Twiddles *tw, int dir, hipStream_t st)
{
    std::vector<c32> t1((size_t) P::R1 * P::M1), t2((size_t) P::R2 * P::R3);
    for (int n1 = 0; n1 < P::R1; n1++)
        for (int b = 0; b < P::M1; b++) {
            const double a = dir * 2.0 * M_PI * (double) ((long long) b * n1 % P::N) / (double) P::N;
            t1[(size_t) n1 * P::M1 + b] = c32{(float) std::cos(a), (float) std::sin(a)};
        }
    for (int e = 0; e < P::R2; e++)
        for (int d = 0; d < P::R3; d++) {
            const double a = dir * 2.0 * M_PI * (double) (e * d % P::M1) / (double) P::M1;
            t2[(size_t) e * P::R3 + d] = c32{(float) std::cos(a), (float) std::sin(a)};
        }
    if (tw->tw1) HIPCHK(hipFree(tw->tw1));
    if (tw->tw2) HIPCHK(hipFree(tw->tw2));
    HIPCHK(hipMalloc((void **) &tw->tw1, t1.size() * sizeof(c32)));
    HIPCHK(hipMalloc((void **) &tw->tw2, t2.size() * sizeof(c32)));
    HIPCHK(hipMemcpyAsync(tw->tw1, t1.data(), t1.size() * sizeof(c32), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(tw->tw2, t2.data(), t2.size() * sizeof(c32), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st)); // host vectors die at return
    return ACCELGPU_OK;
}

template <class P> int make_pow_tables(PowTables *pw, int dir, hipStream_t st)
{
    // rows: w^(4*hi), hi = 1..R/4-1, then w^lo, lo = 1..3 (see afft::TwoLevel)
    const int n1 = P::R1 / 4 + 2, n2 = P::R2 / 4 + 2;
    std::vector<c32> t1((size_t) n1 * P::M1), t2((size_t) n2 * P::R3);
    auto expo = [](int row, int R) { return row < R / 4 - 1 ? 4 * (row + 1) : row - (R / 4 - 1) + 1; };
    for (int k = 0; k < n1; k++)
        for (int b = 0; b < P::M1; b++) {
            const double a = dir * 2.0 * M_PI * (double) (((long long) b * expo(k, P::R1)) % P::N) / (double) P::N;
            t1[(size_t) k * P::M1 + b] = c32{(float) std::cos(a), (float) std::sin(a)};
        }
    for (int k = 0; k < n2; k++)
        for (int d = 0; d < P::R3; d++) {
            const double a = dir * 2.0 * M_PI * (double) ((d * expo(k, P::R2)) % P::M1) / (double) P::M1;
            t2[(size_t) k * P::R3 + d] = c32{(float) std::cos(a), (float) std::sin(a)};
        }
    if (pw->pw1) HIPCHK(hipFree(pw->pw1));
    if (pw->pw2) HIPCHK(hipFree(pw->pw2));
    HIPCHK(hipMalloc((void **) &pw->pw1, t1.size() * sizeof(c32)));
    HIPCHK(hipMalloc((void **) &pw->pw2, t2.size() * sizeof(c32)));
    HIPCHK(hipMemcpyAsync(pw->pw1, t1.data(), t1.size() * sizeof(c32), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(pw->pw2, t2.data(), t2.size() * sizeof(c32), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return ACCELGPU_OK;
}

template <class K> int allow_lds(K kernel, int bytes)
{
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return ACCELGPU_OK;
}

int prof_begin(accelgpu_ctx *ctx, int which, EventPair *ep)
{
    if (!ctx->profiling)
        return ACCELGPU_OK;
    if (!ctx->ev_pool.empty()) {
        *ep = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
    } else {
        HIPCHK(hipEventCreate(&ep->a));
        HIPCHK(hipEventCreate(&ep->b));
    }
    HIPCHK(hipEventRecord(ep->a, ctx->stream));
    (void) which;
    return ACCELGPU_OK;
}

int prof_end(accelgpu_ctx *ctx, int which, EventPair *ep)
{
    if (!ctx->profiling)
        return ACCELGPU_OK;
    HIPCHK(hipEventRecord(ep->b, ctx->stream));
    ctx->ev[which].push_back(*ep);
    return ACCELGPU_OK;
}

// Block descriptors exactly as subharm_ffdot_plane() derives them for
// numharm = harmnum = 1 (reference src/accel_utils.c:903-907, 929-934).
struct BlockDesc {
    int rlo;
    int64_t lobin;
    int numdata;
};

BlockDesc describe_block(double startr, int halfwidth)
{
    const double fullrhi = startr + ACCELGPU_USELEN * 0.5 - 0.5;
    const double drlo = accelhost::calc_required_r(1.0, startr);
    const double drhi = accelhost::calc_required_r(1.0, fullrhi);
    BlockDesc d;
    d.rlo = (int) std::floor(drlo);
    d.lobin = (int64_t) d.rlo - halfwidth;
    const int hibin = (int) std::ceil(drhi) + halfwidth;
    d.numdata = (int) (hibin - d.lobin + 1);
    return d;
}

int launch_block_prep(accelgpu_ctx *ctx, int nblk, const int64_t *d_lobin, const int *d_numdata,
                      c32 *d_out)
{
    const double nph_norm = ctx->nph > 0.f ? 1.0 / std::sqrt((double) ctx->nph) : 0.0;
    EventPair ep;
    int rc = prof_begin(ctx, 0, &ep);
    if (rc) return rc;
    if (ctx->fftlen == 8192) {
        hipLaunchKernelGGL(accelk::k_block_prep<P4K>, dim3(nblk), dim3(P4K::NT),
                           P4K::LDS_BYTES + 64, ctx->stream, ctx->d_fft, ctx->numbins,
                           ctx->setup.lobin, d_lobin, d_numdata, d_out, nph_norm,
                           ctx->tw_fwd_half.tw1, ctx->tw_fwd_half.tw2);
    } else {
        hipLaunchKernelGGL(accelk::k_block_prep<P8K>, dim3(nblk), dim3(P8K::NT),
                           P8K::LDS_BYTES + 64, ctx->stream, ctx->d_fft, ctx->numbins,
                           ctx->setup.lobin, d_lobin, d_numdata, d_out, nph_norm,
                           ctx->tw_fwd_half.tw1, ctx->tw_fwd_half.tw2);
    }
    HIPCHK(hipGetLastError());
    return prof_end(ctx, 0, &ep);
}

int launch_correlate(accelgpu_ctx *ctx, int nblk, const c32 *d_blockfft, const int64_t *d_col,
                     const int *d_ncols, float *d_plane, int64_t rlen)
{
    // Register-resident twiddle form: faster for the 8192-point plan (measured
    // 4.45 vs 5.76 ms on cfg2); the 16384-point plan spills with it and stays on
    // the table form (35.5 vs 41.9 ms).  ACCELGPU_CORR=0/1 forces one for A/B runs.
    const char *variant = getenv("ACCELGPU_CORR");
    const bool rr = variant ? atoi(variant) != 0 : ctx->fftlen == 8192;
    int chunk = rr ? 16 : 8;
    if (const char *e = getenv("ACCELGPU_CORR_CHUNK"))
        chunk = std::max(1, atoi(e));
    const dim3 grid(ctx->numz, (nblk + chunk - 1) / chunk);
    EventPair ep;
    int rc = prof_begin(ctx, 1, &ep);
    if (rc) return rc;
    const int ser = getenv("ACCELGPU_CORR_SER") ? atoi(getenv("ACCELGPU_CORR_SER")) : 16;
#define LAUNCH_RR(PLAN, SER)                                                                   \
    hipLaunchKernelGGL((accelk::k_correlate_rr<PLAN, false, SER>), grid, dim3(PLAN::NT),       \
                       PLAN::LDS_BYTES, ctx->stream, ctx->d_bank, d_blockfft, d_col, d_ncols,   \
                       nblk, chunk, d_plane, rlen, ctx->halfwidth, ctx->pw_inv.pw1,            \
                       ctx->pw_inv.pw2)
    if (rr && ctx->fftlen == 8192) {
        if (ser == 0) LAUNCH_RR(P8K, 0);
        else if (ser == 32) LAUNCH_RR(P8K, 32);
        else LAUNCH_RR(P8K, 16);
    } else if (rr) {
        if (ser == 0) LAUNCH_RR(P16K, 0);
        else if (ser == 32) LAUNCH_RR(P16K, 32);
        else LAUNCH_RR(P16K, 16);
#undef LAUNCH_RR
    } else if (ctx->fftlen == 8192) {
        hipLaunchKernelGGL((accelk::k_correlate<P8K, false>), grid, dim3(P8K::NT), P8K::LDS_BYTES,
                           ctx->stream, ctx->d_bank, d_blockfft, d_col, d_ncols, nblk, chunk,
                           d_plane, rlen, ctx->halfwidth, ctx->tw_inv.tw1, ctx->tw_inv.tw2);
    } else {
        hipLaunchKernelGGL((accelk::k_correlate<P16K, false>), grid, dim3(P16K::NT), P16K::LDS_BYTES,
                           ctx->stream, ctx->d_bank, d_blockfft, d_col, d_ncols, nblk, chunk,
                           d_plane, rlen, ctx->halfwidth, ctx->tw_inv.tw1, ctx->tw_inv.tw2);
    }
    HIPCHK(hipGetLastError());
    return prof_end(ctx, 1, &ep);
}

// J = column pairs per thread of the harmonic-sum kernel.
template <bool DUMP, int J>
int launch_harmsum_j(accelgpu_ctx *ctx, int nstages, int64_t col0, int64_t ncols, float *dump,
                     int dump_stage)
{
    accelk::HarmParams prm;
    for (int i = 0; i < 6; i++)
        prm.powcut[i] = i < ctx->setup.numharmstages ? ctx->setup.powcut[i] : INFINITY;
    constexpr int C = J > 0 ? 2 * J : 1;
    const int remap = getenv("ACCELGPU_NO_XCD_REMAP") ? 0 : 1;
    const int lds_pad = getenv("ACCELGPU_HS_LDS") ? atoi(getenv("ACCELGPU_HS_LDS")) : 0; // occupancy experiments
    unsigned ntiles = (unsigned) ((ncols + 256 * C - 1) / (256 * C));
    if (remap)
        ntiles = ((ntiles + 7) / 8) * 8;
    const dim3 grid(ntiles);
    const unsigned int cap = (unsigned int) std::min<size_t>(ctx->hits_cap, 0xFFFFFFFFu);
    EventPair ep;
    if (!DUMP) {
        int rc = prof_begin(ctx, 2, &ep);
        if (rc) return rc;
    }
    // the offset table is laid out for the configured number of stages; a dump
    // of an earlier stage still runs the full-depth kernel and picks its stage
    if (nstages != ctx->setup.numharmstages) {
        set_error("harmonic-sum launch for %d stages but the table holds %d", nstages,
                  ctx->setup.numharmstages);
        return ACCELGPU_ERR_ARG;
    }
#define LAUNCH_HS(NS)                                                                        \
    hipLaunchKernelGGL((accelk::k_harmsum<NS, C, DUMP>), grid, dim3(256), lds_pad, ctx->stream,     \
                       ctx->d_plane, ctx->setup.rlen, ctx->numz, col0, ncols, ctx->d_rowoff,   \
                       ctx->d_rowchg,                                                         \
                       prm, ctx->d_hits, ctx->d_count, cap, dump, dump_stage, remap)
    switch (nstages) {
    case 1: LAUNCH_HS(1); break;
    case 2: LAUNCH_HS(2); break;
    case 3: LAUNCH_HS(3); break;
    case 4: LAUNCH_HS(4); break;
    case 5: LAUNCH_HS(5); break;
    case 6: LAUNCH_HS(6); break;
    default:
        set_error("bad number of harmonic stages %d", nstages);
        return ACCELGPU_ERR_ARG;
    }
#undef LAUNCH_HS
    HIPCHK(hipGetLastError());
    if (!DUMP)
        return prof_end(ctx, 2, &ep);
    return ACCELGPU_OK;
}

template <bool DUMP>
int launch_harmsum(accelgpu_ctx *ctx, int nstages, int64_t col0, int64_t ncols, float *dump,
                   int dump_stage)
{
    // Few harmonics: little data per thread, so take it in 8-byte pieces, two
    // pairs per thread, to keep enough bytes in flight; many harmonics: one
    // column per thread keeps registers low and waves plentiful.
    int j = nstages <= 2 ? 2 : 0;
    if (const char *e = getenv("ACCELGPU_HARM_J"))
        j = atoi(e);
    if (j == 2)
        return launch_harmsum_j<DUMP, 2>(ctx, nstages, col0, ncols, dump, dump_stage);
    if (j == 1)
        return launch_harmsum_j<DUMP, 1>(ctx, nstages, col0, ncols, dump, dump_stage);
    return launch_harmsum_j<DUMP, 0>(ctx, nstages, col0, ncols, dump, dump_stage);
}

void select_slot(accelgpu_ctx *ctx, int i)
{
    ctx->d_hits = ctx->slots[i].d_hits;
    ctx->h_hits = ctx->slots[i].h_hits;
    ctx->d_count = ctx->slots[i].d_count;
    ctx->h_count = ctx->slots[i].h_count;
}

int reserve_hits(accelgpu_ctx *ctx, size_t want)
{
    if (ctx->hits_cap >= want)
        return ACCELGPU_OK;
    if (ctx->inflight > 0) {
        set_error("cannot grow the hit buffers while %d launches are in flight", ctx->inflight);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->hits_cap = 0;
    for (auto &sl : ctx->slots) {
        if (sl.d_hits) HIPCHK(hipFree(sl.d_hits));
        if (sl.h_hits) HIPCHK(hipHostFree(sl.h_hits));
        sl.d_hits = sl.h_hits = nullptr;
        if (hipMalloc((void **) &sl.d_hits, want * sizeof(DevHit)) != hipSuccess ||
            hipHostMalloc((void **) &sl.h_hits, want * sizeof(DevHit)) != hipSuccess) {
            set_error("cannot allocate room for %zu hits", want);
            return ACCELGPU_ERR_NOMEM;
        }
    }
    ctx->hits_cap = want;
    select_slot(ctx, ctx->head);
    return ACCELGPU_OK;
}

int check_ready(accelgpu_ctx *ctx, bool need_fft)
{
    if (ctx->zmax < 0) {
        set_error("accelgpu_build_kernels() has not been called");
        return ACCELGPU_ERR_ARG;
    }
    if (!ctx->have_search) {
        set_error("accelgpu_set_search() has not been called");
        return ACCELGPU_ERR_ARG;
    }
    if (need_fft && !ctx->d_fft) {
        set_error("no spectrum loaded (accelgpu_load_fft)");
        return ACCELGPU_ERR_ARG;
    }
    if (ctx->setup.zmax != ctx->zmax) {
        set_error("kernel bank built for zmax=%d but the search uses zmax=%d", ctx->zmax,
                  ctx->setup.zmax);
        return ACCELGPU_ERR_ARG;
    }
    return ACCELGPU_OK;
}

int queue_fill(accelgpu_ctx *ctx)
{
    int rc = launch_block_prep(ctx, ctx->nblk, ctx->d_blk_lobin, ctx->d_blk_numdata, ctx->d_blockfft);
    if (rc) return rc;
    return launch_correlate(ctx, ctx->nblk, ctx->d_blockfft, ctx->d_blk_col, ctx->d_blk_ncols,
                            ctx->d_plane, ctx->setup.rlen);
}

int queue_search(accelgpu_ctx *ctx)
{
    HIPCHK(hipMemsetAsync(ctx->d_count, 0, sizeof(unsigned int), ctx->stream));
    int rc = launch_harmsum<false>(ctx, ctx->setup.numharmstages, ctx->col0, ctx->ncols, nullptr, 0);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->h_count, ctx->d_count, sizeof(unsigned int), hipMemcpyDeviceToHost,
                          ctx->stream));
    return ACCELGPU_OK;
}

// Waits for the search launched into slot `si`, re-running it with a larger
// buffer if the compaction overflowed (possible only when no later trial has
// been queued, because the plane must still hold this trial), then sorts into
// the reference's visiting order.
int finish_search(accelgpu_ctx *ctx, int si, accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    accelgpu_ctx::HitSlot &sl = ctx->slots[si];
    select_slot(ctx, si);
    HIPCHK(hipEventSynchronize(sl.done));
    size_t n = *sl.h_count;
    size_t have = std::min(sl.prefetch, n);
    sl.prefetch = 0;
    if (n > ctx->hits_cap) {
        if (ctx->inflight > 0) {
            *n_hits = (int64_t) n;
            set_error("%zu hits overflow the %zu-entry device buffer and a later trial is already "
                      "queued: collect it, then run this trial again on its own", n, ctx->hits_cap);
            return ACCELGPU_ERR_OVERFLOW;
        }
        int rc = reserve_hits(ctx, n + n / 4 + 1024);
        if (rc) return rc;
        select_slot(ctx, si);
        rc = queue_search(ctx);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        n = *ctx->h_count;
        have = 0;
        if (n > ctx->hits_cap) {
            set_error("hit count changed between two runs of the same search");
            return ACCELGPU_ERR_HIP;
        }
    }
    if (n > have) {
        HIPCHK(hipMemcpyAsync(ctx->h_hits + have, ctx->d_hits + have, (n - have) * sizeof(DevHit),
                              hipMemcpyDeviceToHost, ctx->copy_stream));
        HIPCHK(hipStreamSynchronize(ctx->copy_stream));
    }
    *n_hits = (int64_t) n;
    if ((int64_t) n > max_hits) {
        set_error("%zu hits but room for %lld: re-run with a larger buffer", n, (long long) max_hits);
        return ACCELGPU_ERR_OVERFLOW;
    }
    const int64_t col0 = ctx->col0;
    for (size_t i = 0; i < n; i++) {
        const DevHit &d = ctx->h_hits[i];
        accelgpu_hit &h = hits[i];
        h.power = d.power;
        h.stage = (int32_t) (d.stage_zind >> 16);
        h.zind = (int32_t) (d.stage_zind & 0xFFFF);
        h.rind = d.rind;
        h.block = (int32_t) ((d.rind - col0) / ACCELGPU_USELEN);
    }
    std::sort(hits, hits + n, [](const accelgpu_hit &a, const accelgpu_hit &b) {
        if (a.block != b.block) return a.block < b.block;
        if (a.stage != b.stage) return a.stage < b.stage;
        if (a.zind != b.zind) return a.zind < b.zind;
        return a.rind < b.rind;
    });
    return ACCELGPU_OK;
}

} // namespace

// ===========================================================================
extern "C" {

const char *accelgpu_last_error(void) { return accelhost::last_error(); }
const char *accelgpu_version(void) { return "accelgpu 0.1 (gfx950)"; }

int accelgpu_z_resp_halfwidth(double z) { return accelhost::z_resp_halfwidth(z); }
int accelgpu_calc_fftlen(int numharm, int harmnum, int max_zfull)
{
    return accelhost::calc_fftlen(numharm, harmnum, max_zfull);
}
double accelgpu_candidate_sigma(double power, int numsum, double numtrials)
{
    return accelhost::candidate_sigma(power, numsum, numtrials);
}
double accelgpu_power_for_sigma(double sigma, int numsum, double numtrials)
{
    return accelhost::power_for_sigma(sigma, numsum, numtrials);
}
int accelgpu_plan(int64_t numbins, int64_t N, double dt, int zmax, int numharm, double sigma,
                  double flo, double fhi, int64_t lobin, accelgpu_setup *out)
{
    if (!out) {
        set_error("accelgpu_plan: null output");
        return ACCELGPU_ERR_ARG;
    }
    return accelhost::plan(numbins, N, dt, zmax, numharm, sigma, flo, fhi, lobin, out);
}
int accelgpu_replay_candidates(const accelgpu_setup *setup, const accelgpu_hit *hits,
                               int64_t n_hits, accelgpu_cand *cands, int64_t max_cands,
                               int64_t *n_cands, float *sigma_out, int32_t *added)
{
    if (!setup || (!hits && n_hits > 0) || !n_cands || (!cands && max_cands > 0)) {
        set_error("accelgpu_replay_candidates: null argument");
        return ACCELGPU_ERR_ARG;
    }
    return accelhost::replay(setup, hits, n_hits, cands, max_cands, n_cands, sigma_out, added);
}

int accelgpu_create(accelgpu_ctx **out, int device)
{
    if (!out) {
        set_error("accelgpu_create: null output");
        return ACCELGPU_ERR_ARG;
    }
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        set_error("accelgpu_create: device %d of %d", device, ndev);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipSetDevice(device));
    accelgpu_ctx *ctx = new accelgpu_ctx();
    ctx->device = device;
    HIPCHK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (auto &sl : ctx->slots) {
        HIPCHK(hipMalloc((void **) &sl.d_count, sizeof(unsigned int)));
        HIPCHK(hipHostMalloc((void **) &sl.h_count, sizeof(unsigned int)));
        HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    HIPCHK(hipMalloc((void **) &ctx->d_one_lobin, sizeof(int64_t)));
    HIPCHK(hipMalloc((void **) &ctx->d_one_col, sizeof(int64_t)));
    HIPCHK(hipMalloc((void **) &ctx->d_one_numdata, sizeof(int)));
    HIPCHK(hipMalloc((void **) &ctx->d_one_ncols, sizeof(int)));
    int rc;
    if ((rc = allow_lds(accelk::k_block_prep<P4K>, P4K::LDS_BYTES + 64))) return rc;
    if ((rc = allow_lds(accelk::k_block_prep<P8K>, P8K::LDS_BYTES + 64))) return rc;
    if ((rc = allow_lds(accelk::k_correlate<P8K, false>, P8K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_correlate<P16K, false>, P16K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P8K, false, 0>, P8K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P8K, false, 16>, P8K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P8K, false, 32>, P8K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P16K, false, 0>, P16K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P16K, false, 16>, P16K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P16K, false, 32>, P16K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_fft_rows<P8K, -1>, P8K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_fft_rows<P16K, -1>, P16K::LDS_BYTES))) return rc;
    if ((rc = reserve_hits(ctx, (size_t) 1 << 20))) return rc;
    select_slot(ctx, 0);
    *out = ctx;
    return ACCELGPU_OK;
}

void accelgpu_destroy(accelgpu_ctx *ctx)
{
    if (!ctx)
        return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto &v : ctx->ev)
        for (auto &e : v) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto &e : ctx->ev_pool) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    void *dev[] = {ctx->pw_inv.pw1, ctx->pw_inv.pw2, ctx->d_bank, ctx->tw_inv.tw1, ctx->tw_inv.tw2, ctx->tw_fwd_full.tw1,
                   ctx->tw_fwd_full.tw2, ctx->tw_fwd_half.tw1, ctx->tw_fwd_half.tw2,
                   ctx->d_fft_owned, ctx->d_blk_lobin, ctx->d_blk_col, ctx->d_blk_numdata,
                   ctx->d_blk_ncols, ctx->d_blockfft, ctx->d_plane, ctx->d_rowoff, ctx->d_rowchg, ctx->d_tmp, ctx->d_one_lobin, ctx->d_one_col,
                   ctx->d_one_numdata, ctx->d_one_ncols, ctx->d_one_blockfft};
    for (void *p : dev)
        if (p) hipFree(p);
    for (auto &sl : ctx->slots) {
        if (sl.d_hits) hipFree(sl.d_hits);
        if (sl.d_count) hipFree(sl.d_count);
        if (sl.h_hits) hipHostFree(sl.h_hits);
        if (sl.h_count) hipHostFree(sl.h_count);
        if (sl.done) hipEventDestroy(sl.done);
    }
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int accelgpu_set_stream(accelgpu_ctx *ctx, void *hip_stream)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t) hip_stream : ctx->own_stream;
    return ACCELGPU_OK;
}

int accelgpu_synchronize(accelgpu_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_build_kernels(accelgpu_ctx *ctx, int zmax)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (zmax < 0 || zmax > 1200) {
        set_error("accelgpu_build_kernels: zmax %d outside 0..1200", zmax);
        return ACCELGPU_ERR_ARG;
    }
    if (zmax % 2)
        zmax = (zmax / 2 + 1) * 2;
    if (zmax == ctx->zmax)
        return ACCELGPU_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int numz = (zmax / 2) * 2 + 1;
    const int fftlen = accelhost::calc_fftlen(1, 1, zmax);
    if (fftlen != 8192 && fftlen != 16384) {
        set_error("unsupported FFT length %d", fftlen);
        return ACCELGPU_ERR_ARG;
    }
    if (fftlen != ctx->fftlen) {
        if (fftlen == 8192) {
            if ((rc = make_twiddles<P8K>(&ctx->tw_inv, +1, ctx->stream))) return rc;
            if ((rc = make_pow_tables<P8K>(&ctx->pw_inv, +1, ctx->stream))) return rc;
            if ((rc = make_twiddles<P8K>(&ctx->tw_fwd_full, -1, ctx->stream))) return rc;
            if ((rc = make_twiddles<P4K>(&ctx->tw_fwd_half, -1, ctx->stream))) return rc;
        } else {
            if ((rc = make_twiddles<P16K>(&ctx->tw_inv, +1, ctx->stream))) return rc;
            if ((rc = make_pow_tables<P16K>(&ctx->pw_inv, +1, ctx->stream))) return rc;
            if ((rc = make_twiddles<P16K>(&ctx->tw_fwd_full, -1, ctx->stream))) return rc;
            if ((rc = make_twiddles<P8K>(&ctx->tw_fwd_half, -1, ctx->stream))) return rc;
        }
    }
    std::vector<int> hw(numz);
    int maxhw = 0;
    for (int i = 0; i < numz; i++) {
        hw[i] = accelhost::z_resp_halfwidth((double) (-zmax + 2 * i));
        maxhw = std::max(maxhw, hw[i]);
    }
    if (ctx->d_bank) HIPCHK(hipFree(ctx->d_bank));
    ctx->d_bank = nullptr;
    const size_t bank_elems = (size_t) numz * fftlen;
    if (hipMalloc((void **) &ctx->d_bank, bank_elems * sizeof(c32)) != hipSuccess) {
        set_error("cannot allocate the kernel bank (%zu bytes)", bank_elems * sizeof(c32));
        return ACCELGPU_ERR_NOMEM;
    }
    int *d_hw = nullptr;
    HIPCHK(hipMalloc((void **) &d_hw, numz * sizeof(int)));
    HIPCHK(hipMemcpyAsync(d_hw, hw.data(), numz * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_bank, 0, bank_elems * sizeof(c32), ctx->stream));
    const int maxkern = 4 * maxhw;
    hipLaunchKernelGGL(accelk::k_gen_zresp, dim3((maxkern + 127) / 128, numz), dim3(128), 0,
                       ctx->stream, ctx->d_bank, d_hw, zmax, fftlen, maxkern);
    HIPCHK(hipGetLastError());
    if (fftlen == 8192)
        hipLaunchKernelGGL((accelk::k_fft_rows<P8K, -1>), dim3(numz), dim3(P8K::NT),
                           P8K::LDS_BYTES, ctx->stream, ctx->d_bank, ctx->tw_fwd_full.tw1,
                           ctx->tw_fwd_full.tw2);
    else
        hipLaunchKernelGGL((accelk::k_fft_rows<P16K, -1>), dim3(numz), dim3(P16K::NT),
                           P16K::LDS_BYTES, ctx->stream, ctx->d_bank, ctx->tw_fwd_full.tw1,
                           ctx->tw_fwd_full.tw2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(d_hw));
    ctx->zmax = zmax;
    ctx->numz = numz;
    ctx->fftlen = fftlen;
    ctx->halfwidth = hw[0];
    return ACCELGPU_OK;
}

int accelgpu_get_kernel(accelgpu_ctx *ctx, int zind, float *out)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (ctx->zmax < 0 || zind < 0 || zind >= ctx->numz || !out) {
        set_error("accelgpu_get_kernel: no bank or zind %d out of range", zind);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipMemcpyAsync(out, ctx->d_bank + (size_t) zind * ctx->fftlen,
                          (size_t) ctx->fftlen * sizeof(c32), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_load_fft(accelgpu_ctx *ctx, const float *host, int64_t numbins)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!host || numbins <= 0) {
        set_error("accelgpu_load_fft: null spectrum or numbins <= 0");
        return ACCELGPU_ERR_ARG;
    }
    if ((rc = dev_reserve(&ctx->d_fft_owned, &ctx->fft_cap, (size_t) numbins, false, ctx->stream)))
        return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_fft_owned, host, (size_t) numbins * sizeof(c32),
                          hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream)); // the caller may free `host` on return
    ctx->d_fft = ctx->d_fft_owned;
    ctx->numbins = numbins;
    return ACCELGPU_OK;
}

int accelgpu_load_fft_device(accelgpu_ctx *ctx, const void *dev, int64_t numbins)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!dev || numbins <= 0) {
        set_error("accelgpu_load_fft_device: null spectrum or numbins <= 0");
        return ACCELGPU_ERR_ARG;
    }
    ctx->d_fft = (const c32 *) dev;
    ctx->numbins = numbins;
    return ACCELGPU_OK;
}

int accelgpu_set_search(accelgpu_ctx *ctx, const accelgpu_setup *s, int norm_type, float nph)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!s) {
        set_error("accelgpu_set_search: null setup");
        return ACCELGPU_ERR_ARG;
    }
    if (ctx->zmax < 0 || s->zmax != ctx->zmax) {
        set_error("accelgpu_set_search: build the kernel bank for zmax=%d first", s->zmax);
        return ACCELGPU_ERR_ARG;
    }
    if (norm_type != 0) {
        set_error("accelgpu_set_search: only block-median (0) and -photon normalisation are implemented");
        return ACCELGPU_ERR_ARG;
    }
    if (s->numharmstages < 1 || s->numharmstages > ACCELGPU_MAXSTAGES || s->rlo < 0 ||
        s->highestbin < s->rlo) {
        set_error("accelgpu_set_search: inconsistent setup");
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const bool same_geometry = ctx->have_search && ctx->setup.rlo == s->rlo &&
                               ctx->setup.highestbin == s->highestbin &&
                               ctx->setup.lobin == s->lobin && ctx->setup.zmax == s->zmax;
    ctx->setup = *s;
    ctx->setup.numz = ctx->numz;
    ctx->setup.fftlen = ctx->fftlen;
    ctx->setup.halfwidth = ctx->halfwidth;
    ctx->setup.rlen = (s->highestbin + ACCELGPU_USELEN) * 2;
    ctx->setup.plane_bytes = ctx->setup.rlen * (int64_t) ctx->numz * 4;
    ctx->norm_type = norm_type;
    ctx->nph = nph;

    // block grid (reference src/accelsearch.c:143-161 and 165-218)
    std::vector<int64_t> lobin, col;
    std::vector<int> numdata, ncols;
    const double step = ACCELGPU_USELEN * 0.5;
    const int64_t search_col0 = (int64_t) std::floor(accelhost::calc_required_r(1.0, s->rlo)) * 2;
    int nprepop = 0, nsearch = 0;
    for (double r = 8; r < s->rlo; r += step) {
        const BlockDesc d = describe_block(r, ctx->halfwidth);
        lobin.push_back(d.lobin);
        numdata.push_back(d.numdata);
        col.push_back((int64_t) d.rlo * 2);
        // columns at or above the first search column belong to the search blocks
        ncols.push_back((int) std::max<int64_t>(
            0, std::min<int64_t>(ACCELGPU_USELEN, search_col0 - (int64_t) d.rlo * 2)));
        nprepop++;
    }
    for (double r = s->rlo; r + step < (double) s->highestbin; r += step) {
        const BlockDesc d = describe_block(r, ctx->halfwidth);
        lobin.push_back(d.lobin);
        numdata.push_back(d.numdata);
        col.push_back((int64_t) d.rlo * 2);
        ncols.push_back(ACCELGPU_USELEN);
        nsearch++;
    }
    const int nblk = nprepop + nsearch;
    for (int i = 0; i < nblk; i++)
        if (numdata[i] > ctx->fftlen / 2) {
            set_error("block window of %d bins exceeds fftlen/2 = %d", numdata[i], ctx->fftlen / 2);
            return ACCELGPU_ERR_ARG;
        }
    ctx->nblk = nblk;
    ctx->nprepop = nprepop;
    ctx->nsearch = nsearch;
    ctx->setup.num_prepop_blocks = nprepop;
    ctx->setup.num_search_blocks = nsearch;
    ctx->col0 = search_col0;
    ctx->ncols = (int64_t) nsearch * ACCELGPU_USELEN;

    if (nblk > 0) {
        const size_t want = (size_t) nblk;
        if (ctx->blk_cap < want) {
            for (void *p : {(void *) ctx->d_blk_lobin, (void *) ctx->d_blk_col,
                            (void *) ctx->d_blk_numdata, (void *) ctx->d_blk_ncols})
                if (p) HIPCHK(hipFree(p));
            ctx->d_blk_lobin = ctx->d_blk_col = nullptr;
            ctx->d_blk_numdata = ctx->d_blk_ncols = nullptr;
            HIPCHK(hipMalloc((void **) &ctx->d_blk_lobin, want * sizeof(int64_t)));
            HIPCHK(hipMalloc((void **) &ctx->d_blk_col, want * sizeof(int64_t)));
            HIPCHK(hipMalloc((void **) &ctx->d_blk_numdata, want * sizeof(int)));
            HIPCHK(hipMalloc((void **) &ctx->d_blk_ncols, want * sizeof(int)));
            ctx->blk_cap = want;
        }
        HIPCHK(hipMemcpy(ctx->d_blk_lobin, lobin.data(), want * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->d_blk_col, col.data(), want * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->d_blk_numdata, numdata.data(), want * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->d_blk_ncols, ncols.data(), want * sizeof(int), hipMemcpyHostToDevice));
        if ((rc = dev_reserve(&ctx->d_blockfft, &ctx->blockfft_cap,
                              (size_t) nblk * (ctx->fftlen / 2), false, ctx->stream)))
            return rc;
    }

    // the plane: zero-filled so that cells below r = 8, which nothing writes, read as 0
    const size_t plane_elems = (size_t) ctx->setup.rlen * ctx->numz;
    const bool fresh = ctx->plane_cap < plane_elems || !ctx->d_plane;
    if ((rc = dev_reserve(&ctx->d_plane, &ctx->plane_cap, plane_elems, false, ctx->stream)))
        return rc;
    if (fresh || !same_geometry)
        HIPCHK(hipMemsetAsync(ctx->d_plane, 0, plane_elems * sizeof(float), ctx->stream));

    // sub-harmonic source rows, in summing order (stage ascending, odd harmonic
    // ascending), as byte offsets into the plane: [numz][nh]
    const int nh = (1 << (s->numharmstages - 1)) - 1;
    if (nh > 0) {
        std::vector<int64_t> tab((size_t) nh * ctx->numz);
        int i = 0;
        for (int st = 1; st < s->numharmstages; st++)
            for (int h = 1; h < (1 << st); h += 2, i++) {
                const double f = (double) h / (double) (1 << st);
                for (int zi = 0; zi < ctx->numz; zi++) {
                    const int zz = (int) s->zlo + zi * 2;
                    const int subz = accelhost::calc_required_z(f, zz);
                    const int zind = accelhost::index_from_z(subz, s->zlo);
                    tab[(size_t) zi * nh + i] = (int64_t) zind * ctx->setup.rlen * 4;
                }
            }
        if ((rc = dev_reserve(&ctx->d_rowoff, &ctx->rowoff_cap, tab.size(), false, ctx->stream)))
            return rc;
        HIPCHK(hipMemcpy(ctx->d_rowoff, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        std::vector<uint32_t> chg((size_t) ctx->numz);
        for (int zi = 0; zi < ctx->numz; zi++) {
            uint32_t m = 0;
            for (int k = 0; k < nh; k++)
                if (zi == 0 || tab[(size_t) zi * nh + k] != tab[(size_t) (zi - 1) * nh + k])
                    m |= 1u << k;
            chg[zi] = m;
        }
        if ((rc = dev_reserve(&ctx->d_rowchg, &ctx->rowchg_cap, chg.size(), false, ctx->stream)))
            return rc;
        HIPCHK(hipMemcpy(ctx->d_rowchg, chg.data(), chg.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    } else if ((rc = dev_reserve(&ctx->d_rowoff, &ctx->rowoff_cap, 1, false, ctx->stream))) {
        return rc;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->have_search = true;
    return ACCELGPU_OK;
}

int accelgpu_fill_plane(accelgpu_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, true))) return rc;
    if (ctx->nblk == 0)
        return ACCELGPU_OK;
    if ((rc = queue_fill(ctx))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_search_plane(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    if (!n_hits || (!hits && max_hits > 0)) {
        set_error("accelgpu_search_plane: null output");
        return ACCELGPU_ERR_ARG;
    }
    if (ctx->inflight > 0) {
        set_error("accelgpu_search_plane: collect the %d launched trial(s) first", ctx->inflight);
        return ACCELGPU_ERR_ARG;
    }
    *n_hits = 0;
    if (ctx->nsearch == 0)
        return ACCELGPU_OK;
    const int si = ctx->head;
    select_slot(ctx, si);
    ctx->slots[si].prefetch = 0;
    if ((rc = queue_search(ctx))) return rc;
    HIPCHK(hipEventRecord(ctx->slots[si].done, ctx->stream));
    return finish_search(ctx, si, hits, max_hits, n_hits);
}

int accelgpu_run_launch(accelgpu_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, true))) return rc;
    if (ctx->inflight >= 2) {
        set_error("accelgpu_run_launch: two launches are already in flight; collect one first");
        return ACCELGPU_ERR_ARG;
    }
    const int si = ctx->tail;
    accelgpu_ctx::HitSlot &sl = ctx->slots[si];
    select_slot(ctx, si);
    sl.prefetch = 0;
    if (ctx->nsearch > 0) {
        if ((rc = queue_fill(ctx))) return rc;
        if ((rc = queue_search(ctx))) return rc;
        // speculative copy of the head of the hit buffer: usually all of them
        const size_t pre = std::min<size_t>(ctx->hits_cap, 65536);
        HIPCHK(hipMemcpyAsync(sl.h_hits, sl.d_hits, pre * sizeof(DevHit), hipMemcpyDeviceToHost,
                              ctx->stream));
        sl.prefetch = pre;
    }
    HIPCHK(hipEventRecord(sl.done, ctx->stream));
    ctx->tail ^= 1;
    ctx->inflight++;
    return ACCELGPU_OK;
}

int accelgpu_run_collect(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (ctx->inflight == 0) {
        set_error("accelgpu_run_collect: nothing launched");
        return ACCELGPU_ERR_ARG;
    }
    if (!n_hits || (!hits && max_hits > 0)) {
        set_error("accelgpu_run_collect: null output");
        return ACCELGPU_ERR_ARG;
    }
    const int si = ctx->head;
    ctx->head ^= 1;
    ctx->inflight--;
    *n_hits = 0;
    if (ctx->nsearch == 0) {
        HIPCHK(hipEventSynchronize(ctx->slots[si].done));
        return ACCELGPU_OK;
    }
    rc = finish_search(ctx, si, hits, max_hits, n_hits);
    if (rc == ACCELGPU_ERR_OVERFLOW && *n_hits > max_hits && (size_t) *n_hits <= ctx->hits_cap) {
        // the caller's buffer was too small; the hits are still in the slot's pinned
        // mirror, so let a following accelgpu_search_plane() (plane intact) or a
        // larger collect retry succeed: put the slot back
        ctx->head = si;
        ctx->inflight++;
    }
    return rc;
}

int accelgpu_run(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    int rc = accelgpu_run_launch(ctx);
    if (rc) return rc;
    return accelgpu_run_collect(ctx, hits, max_hits, n_hits);
}

int accelgpu_fund_block(accelgpu_ctx *ctx, double startr, float *powers, int32_t *rlo_out)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, true))) return rc;
    if (!powers) {
        set_error("accelgpu_fund_block: null output");
        return ACCELGPU_ERR_ARG;
    }
    const BlockDesc d = describe_block(startr, ctx->halfwidth);
    const int64_t zero = 0;
    const int ncols = ACCELGPU_USELEN;
    const size_t elems = (size_t) ctx->numz * ACCELGPU_USELEN;
    if ((rc = dev_reserve(&ctx->d_tmp, &ctx->tmp_cap, elems, false, ctx->stream))) return rc;
    if (!ctx->d_one_blockfft)
        HIPCHK(hipMalloc((void **) &ctx->d_one_blockfft, (size_t) 8192 * sizeof(c32)));
    HIPCHK(hipMemcpyAsync(ctx->d_one_lobin, &d.lobin, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_one_numdata, &d.numdata, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_one_col, &zero, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_one_ncols, &ncols, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream)); // sources are stack variables
    const bool prof = ctx->profiling;
    ctx->profiling = false;
    rc = launch_block_prep(ctx, 1, ctx->d_one_lobin, ctx->d_one_numdata, ctx->d_one_blockfft);
    if (!rc)
        rc = launch_correlate(ctx, 1, ctx->d_one_blockfft, ctx->d_one_col, ctx->d_one_ncols,
                              ctx->d_tmp, ACCELGPU_USELEN);
    ctx->profiling = prof;
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(powers, ctx->d_tmp, elems * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (rlo_out)
        *rlo_out = d.rlo;
    return ACCELGPU_OK;
}

int accelgpu_summed_block(accelgpu_ctx *ctx, int block, int stage, float *powers)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    if (block < 0 || block >= ctx->nsearch || stage < 0 || stage >= ctx->setup.numharmstages ||
        !powers) {
        set_error("accelgpu_summed_block: block %d / stage %d out of range", block, stage);
        return ACCELGPU_ERR_ARG;
    }
    const size_t elems = (size_t) ctx->numz * ACCELGPU_USELEN;
    if ((rc = dev_reserve(&ctx->d_tmp, &ctx->tmp_cap, elems, false, ctx->stream))) return rc;
    rc = launch_harmsum<true>(ctx, ctx->setup.numharmstages, ctx->col0 + (int64_t) block * ACCELGPU_USELEN,
                              ACCELGPU_USELEN, ctx->d_tmp, stage);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(powers, ctx->d_tmp, elems * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

static int stage_of(int numharm)
{
    int st = 0;
    for (int n = numharm; n > 1; n >>= 1)
        st++;
    return (1 << st) == numharm ? st : -1;
}

// Uploads a host block into the context's block buffer.
static int upload_block(accelgpu_ctx *ctx, const float *powers)
{
    const size_t elems = (size_t) ctx->numz * ACCELGPU_USELEN;
    int rc = dev_reserve(&ctx->d_tmp, &ctx->tmp_cap, elems, false, ctx->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_tmp, powers, elems * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_fund_to_ffdotplane(accelgpu_ctx *ctx, const float *powers, int32_t rlo)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    const int64_t col = (int64_t) rlo * 2;
    if (!powers || col < 0 || col + ACCELGPU_USELEN > ctx->setup.rlen) {
        set_error("accelgpu_fund_to_ffdotplane: block at rlo=%d does not fit the plane", rlo);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipMemcpy2DAsync(ctx->d_plane + col, (size_t) ctx->setup.rlen * sizeof(float), powers,
                            ACCELGPU_USELEN * sizeof(float), ACCELGPU_USELEN * sizeof(float),
                            ctx->numz, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_inmem_add_ffdotpows(accelgpu_ctx *ctx, float *powers, int32_t rlo, int numharm,
                                 int harmnum)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    const int st = stage_of(numharm);
    if (!powers || st < 1 || st > 5 || harmnum < 1 || harmnum >= numharm || rlo < 0 ||
        (int64_t) rlo * 2 + ACCELGPU_USELEN > ctx->setup.rlen) {
        set_error("accelgpu_inmem_add_ffdotpows: bad numharm/harmnum/rlo (%d, %d, %d)", numharm,
                  harmnum, rlo);
        return ACCELGPU_ERR_ARG;
    }
    std::vector<int> zind(ctx->numz);
    const double f = (double) harmnum / (double) numharm;
    for (int zi = 0; zi < ctx->numz; zi++) {
        const int zz = (int) ctx->setup.zlo + zi * 2;
        zind[zi] = accelhost::index_from_z(accelhost::calc_required_z(f, zz), ctx->setup.zlo);
    }
    int *d_zind = nullptr;
    HIPCHK(hipMalloc((void **) &d_zind, zind.size() * sizeof(int)));
    HIPCHK(hipMemcpyAsync(d_zind, zind.data(), zind.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = upload_block(ctx, powers))) return rc;
    hipLaunchKernelGGL(accelk::k_block_add_harm, dim3((ACCELGPU_USELEN + 255) / 256, ctx->numz),
                       dim3(256), 0, ctx->stream, ctx->d_tmp, ctx->d_plane, ctx->setup.rlen, rlo, st,
                       harmnum, d_zind);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(powers, ctx->d_tmp, (size_t) ctx->numz * ACCELGPU_USELEN * sizeof(float),
                          hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(d_zind));
    return ACCELGPU_OK;
}

int accelgpu_search_ffdotpows(accelgpu_ctx *ctx, const float *powers, int32_t rlo, int numharm,
                              accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    const int st = stage_of(numharm);
    if (!powers || !n_hits || (!hits && max_hits > 0) || st < 0 || st >= ctx->setup.numharmstages) {
        set_error("accelgpu_search_ffdotpows: bad arguments (numharm %d)", numharm);
        return ACCELGPU_ERR_ARG;
    }
    const int64_t bcol = (int64_t) rlo * 2;
    for (;;) {
        if ((rc = upload_block(ctx, powers))) return rc;
        HIPCHK(hipMemsetAsync(ctx->d_count, 0, sizeof(unsigned int), ctx->stream));
        const unsigned int cap = (unsigned int) std::min<size_t>(ctx->hits_cap, 0xFFFFFFFFu);
        hipLaunchKernelGGL(accelk::k_block_threshold, dim3((ACCELGPU_USELEN + 255) / 256, ctx->numz),
                           dim3(256), 0, ctx->stream, ctx->d_tmp, ctx->setup.powcut[st], st, bcol,
                           ctx->d_hits, ctx->d_count, cap);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ctx->h_count, ctx->d_count, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (*ctx->h_count <= ctx->hits_cap)
            break;
        if ((rc = reserve_hits(ctx, (size_t) *ctx->h_count + 1024))) return rc;
    }
    const size_t n = *ctx->h_count;
    *n_hits = (int64_t) n;
    if ((int64_t) n > max_hits) {
        set_error("%zu hits but room for %lld: re-run with a larger buffer", n, (long long) max_hits);
        return ACCELGPU_ERR_OVERFLOW;
    }
    if (n) {
        HIPCHK(hipMemcpyAsync(ctx->h_hits, ctx->d_hits, n * sizeof(DevHit), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    const int64_t rel = bcol - ctx->col0;
    const int32_t block = (rel >= 0 && rel % ACCELGPU_USELEN == 0) ? (int32_t) (rel / ACCELGPU_USELEN) : -1;
    for (size_t i = 0; i < n; i++) {
        const DevHit &d = ctx->h_hits[i];
        hits[i].power = d.power;
        hits[i].stage = (int32_t) (d.stage_zind >> 16);
        hits[i].zind = (int32_t) (d.stage_zind & 0xFFFF);
        hits[i].rind = d.rind;
        hits[i].block = block;
    }
    std::sort(hits, hits + n, [](const accelgpu_hit &a, const accelgpu_hit &b) {
        if (a.zind != b.zind) return a.zind < b.zind;
        return a.rind < b.rind;
    });
    return ACCELGPU_OK;
}

static int plane_span(accelgpu_ctx *ctx, int zind, int64_t col, int64_t count)
{
    if (!ctx->have_search || !ctx->d_plane) {
        set_error("no plane allocated (accelgpu_set_search)");
        return ACCELGPU_ERR_ARG;
    }
    if (zind < 0 || zind >= ctx->numz || col < 0 || count < 0 || col + count > ctx->setup.rlen) {
        set_error("plane span z=%d col=%lld count=%lld out of range", zind, (long long) col,
                  (long long) count);
        return ACCELGPU_ERR_ARG;
    }
    return ACCELGPU_OK;
}

int accelgpu_get_plane(accelgpu_ctx *ctx, int zind, int64_t col, int64_t count, float *out)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = plane_span(ctx, zind, col, count))) return rc;
    HIPCHK(hipMemcpyAsync(out, ctx->d_plane + (size_t) zind * ctx->setup.rlen + col,
                          (size_t) count * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_put_plane(accelgpu_ctx *ctx, int zind, int64_t col, int64_t count, const float *in)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = plane_span(ctx, zind, col, count))) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_plane + (size_t) zind * ctx->setup.rlen + col, in,
                          (size_t) count * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_set_profiling(accelgpu_ctx *ctx, int on)
{
    int rc = use(ctx);
    if (rc) return rc;
    ctx->profiling = on != 0;
    return ACCELGPU_OK;
}

int accelgpu_get_timings(accelgpu_ctx *ctx, int64_t launches[3], double total_ms[3], int reset)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 3; k++) {
        for (auto &e : ctx->ev[k]) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
            ctx->prof_ms[k] += ms;
            ctx->prof_launches[k]++;
            ctx->ev_pool.push_back(e);
        }
        ctx->ev[k].clear();
        if (launches) launches[k] = ctx->prof_launches[k];
        if (total_ms) total_ms[k] = ctx->prof_ms[k];
        if (reset) {
            ctx->prof_launches[k] = 0;
            ctx->prof_ms[k] = 0;
        }
    }
    return ACCELGPU_OK;
}

} // extern "C"
