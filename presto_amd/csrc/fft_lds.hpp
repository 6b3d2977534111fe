// fft_lds.hpp -- fixed-size complex float32 FFTs for gfx950, one transform per
// workgroup, data in registers between three in-register radix passes and
// exchanged twice through LDS.
//
//   N = R1 * R2 * R3,  NT threads,  E = N / NT values per thread.
//
//   pass 1: k = M1*a + b (M1 = R2*R3): radix-R1 over a for each b, then the
//           large twiddle w_N^(b*n1) from a table laid out [n1][b] (coalesced).
//   pass 2: b = R3*c + d: radix-R2 over c for each (n1, d), then w_M1^(e*d).
//   pass 3: radix-R3 over d for each (n1, e); output index n1 + R1*(e + R2*f),
//           so that consecutive threads hold consecutive outputs (coalesced
//           global stores).
//
// LDS images are padded so that both the 8-byte writes and the 8-byte reads of
// each exchange are bank-conflict free on CDNA4 (64 banks for ds_read_b64,
// lane groups of 32; 32 banks and lane groups of 16 for ds_write_b64).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace afft {

typedef float c32 __attribute__((ext_vector_type(2)));

// Complex products written on whole vectors: one packed multiply and one packed fused
// multiply-add, the swap of halves and the sign left to operand selects and modifiers.
__device__ __forceinline__ c32 cmul(c32 a, c32 b)
{
    const c32 t = a.xx * b;
    return __builtin_elementwise_fma(a.yy, c32{-b.y, b.x}, t);
}
__device__ __forceinline__ c32 cmul_s(c32 a, c32 b) { return cmul(a, b); }
// The same product for two values that both live in vector registers, with the second
// instruction spelled out: a.y * (-b.y, b.x) + t as one packed fma whose operand selects
// swap b's halves and whose modifier negates the low one.  (For register operands the
// compiler builds (-b.y, b.x) with an xor and a move first; for constants it does not,
// so cmul() stays the form for those.)
__device__ __forceinline__ c32 cmul_v(c32 a, c32 b)
{
    const c32 t = a.xx * b;
    c32 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(b), "v"(t));
    return r;
}
// a * conj(b)
__device__ __forceinline__ c32 cmul_conj(c32 a, c32 b)
{
    const c32 t = a * b.xx;
    return __builtin_elementwise_fma(a.yx, c32{b.y, -b.y}, t);
}
// a * conj(b) for two values in vector registers, the second instruction spelled out as
// in cmul_v(): (a.y, a.x) * (b.y, -b.y) + t.
__device__ __forceinline__ c32 cmul_conj_v(c32 a, c32 b)
{
    const c32 t = a * b.xx;
    c32 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(b), "v"(t));
    return r;
}
template <int DIR> __device__ __forceinline__ c32 mul_i(c32 a)
{
    return DIR > 0 ? c32{-a.y, a.x} : c32{a.y, -a.x};
}

// Single operations that are never contracted into a fused multiply-add, for
// code whose results must equal a plain C evaluation bit for bit.  (The
// __fmul_rn/__fadd_rn intrinsics are plain operators here and do get fused.)
namespace strict {
__device__ __forceinline__ float mul(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double mul(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double add(double a, double b)
{
#pragma clang fp contract(off)
    return a + b;
}
// x*x + y*y, each operation rounded on its own
__device__ __forceinline__ float norm2(float x, float y)
{
#pragma clang fp contract(off)
    const float a = x * x;
    const float b = y * y;
    return a + b;
}
} // namespace strict

// 32nd roots of unity, cos and sin of 2*pi*j/32.
constexpr float kCos32[32] = {
    1.f, 0.980785251f, 0.923879504f, 0.831469595f, 0.707106769f, 0.555570245f, 0.382683426f,
    0.195090324f, 0.f, -0.195090324f, -0.382683426f, -0.555570245f, -0.707106769f,
    -0.831469595f, -0.923879504f, -0.980785251f, -1.f, -0.980785251f, -0.923879504f,
    -0.831469595f, -0.707106769f, -0.555570245f, -0.382683426f, -0.195090324f, 0.f,
    0.195090324f, 0.382683426f, 0.555570245f, 0.707106769f, 0.831469595f, 0.923879504f,
    0.980785251f};
constexpr float kSin32[32] = {
    0.f, 0.195090324f, 0.382683426f, 0.555570245f, 0.707106769f, 0.831469595f, 0.923879504f,
    0.980785251f, 1.f, 0.980785251f, 0.923879504f, 0.831469595f, 0.707106769f, 0.555570245f,
    0.382683426f, 0.195090324f, 0.f, -0.195090324f, -0.382683426f, -0.555570245f,
    -0.707106769f, -0.831469595f, -0.923879504f, -0.980785251f, -1.f, -0.980785251f,
    -0.923879504f, -0.831469595f, -0.707106769f, -0.555570245f, -0.382683426f, -0.195090324f};

// In-register DFT of N = 2,4,...,32 points, natural order in and out:
// v[k] <- sum_n v[n] * exp(DIR * 2*pi*i * n*k / N).  Everything is unrolled at
// compile time so v[] lives in VGPRs.
// SER: sizes >= SER run their two half-size sub-transforms one after the other
// (scheduling barrier between them): fewer live temporaries, less overlap.  0 = off.
// PRE2: the innermost butterflies -- v[j] +- v[j + N/2], the first thing the recursion computes -- have been
// taken already: v[j] comes in as that sum and v[j + N/2] as that difference (a caller whose inputs are
// products with a table can fold the butterfly into the table: see k_fold_bank).
template <int N, int DIR, int SER = 0, bool PRE2 = false> struct Dft {
    template <int K> static __device__ __forceinline__ void combine(c32 *v, const c32 *e, const c32 *o)
    {
        c32 t;
        if constexpr (K == 0) {
            t = o[0];
        } else if constexpr (4 * K == N) {
            // e +- i*o (DIR > 0) or e -+ i*o in one packed add each: the swap of o's halves
            // and the sign ride on the operand selects instead of on register moves (the
            // compiler does not form these from vector code; it does for cmul above)
            c32 p, m;
            const c32 a = e[K], b = o[K];
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(p) : "v"(a), "v"(b));
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(m) : "v"(a), "v"(b));
            v[K] = DIR > 0 ? p : m;
            v[K + N / 2] = DIR > 0 ? m : p;
            if constexpr (K + 1 < N / 2)
                combine<K + 1>(v, e, o);
            return;
        } else {
            // e + w*o as two fused multiply-adds on top of e, then e - w*o = 2e - (e + w*o) as
            // a third: three packed instructions per butterfly instead of four (product in
            // two, sum, difference)
            constexpr int J = K * (32 / N);
            const c32 w = c32{kCos32[J], DIR * kSin32[J]}, wq = c32{-DIR * kSin32[J], kCos32[J]};
            const c32 u = __builtin_elementwise_fma(o[K].xx, w, e[K]);
            const c32 a = __builtin_elementwise_fma(o[K].yy, wq, u);
            v[K] = a;
            v[K + N / 2] = __builtin_elementwise_fma(e[K], c32{2.f, 2.f}, -a);
            if constexpr (K + 1 < N / 2)
                combine<K + 1>(v, e, o);
            return;
        }
        v[K] = e[K] + t;
        v[K + N / 2] = e[K] - t;
        if constexpr (K + 1 < N / 2)
            combine<K + 1>(v, e, o);
    }
    static __device__ __forceinline__ void run(c32 *v)
    {
        c32 e[N / 2], o[N / 2];
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            e[i] = v[2 * i];
            o[i] = v[2 * i + 1];
        }
        Dft<N / 2, DIR, SER, PRE2>::run(e);
        if constexpr (SER > 0 && N >= SER)
            __builtin_amdgcn_sched_barrier(0);
        Dft<N / 2, DIR, SER, PRE2>::run(o);
        if constexpr (SER > 0 && N >= SER)
            __builtin_amdgcn_sched_barrier(0);
        combine<0>(v, e, o);
    }
};
template <int DIR, int SER, bool PRE2> struct Dft<2, DIR, SER, PRE2> {
    static __device__ __forceinline__ void run(c32 *v)
    {
        if constexpr (!PRE2) {
            const c32 a = v[0], b = v[1];
            v[0] = a + b;
            v[1] = a - b;
        }
    }
};
template <int DIR, int SER, bool PRE2> struct Dft<1, DIR, SER, PRE2> {
    static __device__ __forceinline__ void run(c32 *) {}
};

template <int R1_, int R2_, int R3_, int NT_> struct Plan {
    static constexpr int R1 = R1_, R2 = R2_, R3 = R3_, NT = NT_;
    static constexpr int N = R1 * R2 * R3;
    static constexpr int M1 = R2 * R3;
    static constexpr int E = N / NT;           // values per thread
    static constexpr int B1 = M1 / NT;         // radix-R1 butterflies per thread
    static constexpr int B2 = R1 * R3 / NT;    // radix-R2 butterflies per thread
    static constexpr int B3 = R1 * R2 / NT;    // radix-R3 butterflies per thread
    static constexpr int S1 = M1 + 16;         // row stride of the first LDS image
    static constexpr int S2 = R1 + 1;          // row stride of the second LDS image
    static constexpr int LDS_ELEMS = (R1 * S1 > M1 * S2) ? R1 * S1 : M1 * S2;
    static constexpr int LDS_BYTES = LDS_ELEMS * 8;
    static_assert(B1 >= 1 && B2 >= 1 && B3 >= 1, "too many threads for this size");
    static_assert(B1 * R1 == E && B2 * R2 == E && B3 * R3 == E, "uneven split");
};

// Twiddles from two short exact tables.  With n = 4*hi + lo, w^n = A[hi]*B[lo],
// A[hi] = w^(4*hi) (hi = 1..R/4-1) and B[lo] = w^lo (lo = 1..3), all rounded from
// double: every factor is exact or one product of two exact values.  A thread
// keeps R/4+2 complex registers for the whole kernel instead of loading R-1
// twiddles per transform.
constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x / 2); }

template <int R> struct TwoLevel {
    static constexpr int NA = R / 4 - 1, NB = 3;
    c32 A[NA], B[NB];
};

// Makes the register copies opaque to the optimiser at the point of call, so
// products of table entries are formed where they are used (a handful of live
// registers) instead of being hoisted out of the caller's loop (R-1 live values).
template <int R> __device__ __forceinline__ void launder(TwoLevel<R> &w)
{
    // (each value as one 64-bit register pair: laundered as two floats the halves come
    // back in unrelated registers and every packed use needs two moves to pair them)
#pragma unroll
    for (int k = 0; k < TwoLevel<R>::NA; k++)
        asm volatile("" : "+v"(w.A[k]));
#pragma unroll
    for (int k = 0; k < TwoLevel<R>::NB; k++)
        asm volatile("" : "+v"(w.B[k]));
}

// All R-1 twiddles w^1 .. w^(R-1) from the two tables, for kernels that keep them in
// registers for many transforms.
template <int R> __device__ __forceinline__ void expand_twiddles(c32 *full, const TwoLevel<R> &w)
{
#pragma unroll
    for (int n = 1; n < R; n++) {
        const int hi = n >> 2, lo = n & 3;
        full[n - 1] = hi == 0 ? w.B[lo - 1] : lo == 0 ? w.A[hi - 1] : cmul_v(w.A[hi - 1], w.B[lo - 1]);
    }
}
// (in groups of four: the four first halves, then the four fused halves, so that no
// instruction reads the result of the one right before it -- that costs a wait state)
template <int R> __device__ __forceinline__ void apply_twiddles_full(c32 *v, const c32 *full)
{
    static_assert(R % 4 == 0, "radix is a multiple of four");
#pragma unroll
    for (int n0 = 0; n0 < R; n0 += 4) {
        c32 t[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (n0 + k > 0)
                t[k] = v[n0 + k].xx * full[n0 + k - 1];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (n0 + k > 0)
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
                    : "=v"(v[n0 + k])
                    : "v"(v[n0 + k]), "v"(full[n0 + k - 1]), "v"(t[k]));
    }
}

template <int R> __device__ __forceinline__ void apply_twiddles_2level(c32 *v, const TwoLevel<R> &w)
{
#pragma unroll
    for (int n = 1; n < R; n++) {
        const int hi = n >> 2, lo = n & 3;
        c32 t;
        if (hi == 0)
            t = w.B[lo - 1];
        else if (lo == 0)
            t = w.A[hi - 1];
        else
            t = cmul_v(w.A[hi - 1], w.B[lo - 1]);
        v[n] = cmul_v(v[n], t);
    }
}

// One transform, general form.  load(u, a, b) returns input element M1*a + b for
// the thread's u-th pass-1 butterfly (u, a are compile-time after unrolling);
// tw1(u, b, v) multiplies v[n1] by exp(DIR*2*pi*i*b*n1/N), n1 = 1..R1-1;
// tw2(u, d, v) multiplies v[e] by exp(DIR*2*pi*i*e*d/M1), e = 1..R2-1;
// mid() runs after the pass-1 LDS writes and before the first barrier (a place
// to start global loads that should fly during passes 2 and 3);
// store(n, v) receives output n.  Ends with all LDS reads done but no trailing
// barrier: callers that loop must __syncthreads() before reuse.
// An 8-byte LDS read that stays one ds_read_b64: left alone, the compiler pairs neighbouring
// reads into ds_read2_b64, which the LDS serves at half the rate (two accesses of 4 x 16 lanes,
// 8 cycles per wave-instruction, against 2 cycles for a ds_read_b64).
__device__ __forceinline__ c32 lds_read(const c32 *p)
{
    typedef const volatile c32 __attribute__((address_space(3))) *lds_ptr;
    return *(lds_ptr) p;
}

// Phase marker hook of fft3x(): does nothing in product kernels; a diagnostic build passes
// an object that records the cycle counter at each phase boundary.
struct NoStamp {
    __device__ __forceinline__ void begin(int) {}
    __device__ __forceinline__ void operator()(int) const {}
};

// PAIR (plans with two pass-3 butterflies per thread): the thread takes butterflies 2t and 2t + 1
// instead of t and t + NT, so that it owns outputs n and n + 1 for n = 2t + R1*R2*f, and
// store(PairIndex<f>(), v, w) receives both: the caller can store them as one wider access, and
// f can be used where the instruction needs an immediate.
template <int F> struct PairIndex {
    static constexpr int value = F;
};
template <int F, int NF, class Store> __device__ __forceinline__ void pair_loop(Store &store, const c32 *v)
{
    if constexpr (F < NF) {
        store(PairIndex<F>(), v[F], v[NF + F]);
        pair_loop<F + 1, NF>(store, v);
    }
}
// PRE2: load() returns the inputs with pass 1's innermost butterflies taken (see Dft).
// A store functor that takes its output's place as compile-time constants -- store(PairIndex<u>(), PairIndex<f>(), v)
// for output t + NT*u + R1*R2*f -- gets them from a recursion instead of a loop: whatever it derives from them
// (which offset register, whether a range check is needed) is then decided when the kernel is compiled even if
// the optimiser would not unroll the loop.  Functors that take (n, v) are called from the plain loop.
template <class P, int U, int F, class Store> __device__ __forceinline__ void single_loop(Store &store, const c32 *v)
{
    if constexpr (F < P::R3) {
        store(PairIndex<U>(), PairIndex<F>(), v[U * P::R3 + F]);
        single_loop<P, U, F + 1>(store, v);
    } else if constexpr (U + 1 < P::B3) {
        single_loop<P, U + 1, 0>(store, v);
    }
}
template <class P, int DIR, int SER = 0, bool PAIR = false, bool PRE2 = false, class Load, class Tw1, class Tw2, class Mid,
          class Store, class Stamp = NoStamp>
__device__ __forceinline__ void fft3x(c32 *lds, Load load, Tw1 tw1, Tw2 tw2, Mid mid, Store store,
                                      Stamp stamp = Stamp())
{
    const int t = threadIdx.x;
    c32 v[P::E];

#pragma unroll
    for (int u = 0; u < P::B1; u++) {
        const int b = t + P::NT * u;
#pragma unroll
        for (int a = 0; a < P::R1; a++)
            v[u * P::R1 + a] = load(u, a, b);
        Dft<P::R1, DIR, SER, PRE2>::run(&v[u * P::R1]);
        tw1(u, b, &v[u * P::R1]);
        stamp(1); // inputs arrived, pass-1 arithmetic done
#pragma unroll
        for (int n1 = 0; n1 < P::R1; n1++)
            lds[n1 * P::S1 + b] = v[u * P::R1 + n1];
    }
    mid();
    stamp(2); // first image written (issued)
    __syncthreads();
    stamp(3);

#pragma unroll
    for (int u = 0; u < P::B2; u++) {
        const int g = t + P::NT * u;
        const int d = g % P::R3, n1 = g / P::R3;
#pragma unroll
        for (int c = 0; c < P::R2; c++)
            v[u * P::R2 + c] = lds_read(&lds[n1 * P::S1 + P::R3 * c + d]);
    }
    stamp(4); // first image read (issued)
    __syncthreads(); // every read of image 1 is done before image 2 overwrites it
    stamp(5);
#pragma unroll
    for (int u = 0; u < P::B2; u++) {
        const int g = t + P::NT * u;
        const int d = g % P::R3, n1 = g / P::R3;
        Dft<P::R2, DIR, SER>::run(&v[u * P::R2]);
        tw2(u, d, &v[u * P::R2]);
        // PAIR: one more element of padding per e, so that the paired pass-3 reads (lanes 0-15 and
        // 16-31 of a group read rows e and e + 1 at every other n1) fall on disjoint banks
#pragma unroll
        for (int e = 0; e < P::R2; e++)
            lds[(e * P::R3 + d) * P::S2 + n1 + (PAIR ? e : 0)] = v[u * P::R2 + e];
    }
    stamp(6); // pass-2 arithmetic done, second image written (issued)
    __syncthreads();
    stamp(7);

    if constexpr (PAIR) {
        static_assert(P::B3 == 2, "pairs need two pass-3 butterflies per thread");
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int g = 2 * t + u;
            const int n1 = g % P::R1, e = g / P::R1;
#pragma unroll
            for (int d = 0; d < P::R3; d++)
                v[u * P::R3 + d] = lds_read(&lds[(e * P::R3 + d) * P::S2 + n1 + e]);
        }
        static_assert((P::R2 * P::R3 * P::S2 + P::R2) * sizeof(c32) <= P::LDS_BYTES, "padded second image fits");
#pragma unroll
        for (int u = 0; u < 2; u++)
            Dft<P::R3, DIR, SER>::run(&v[u * P::R3]);
        // (the pair's index f reaches the caller as a compile-time constant: PairIndex<f>)
        pair_loop<0, P::R3>(store, v);
    } else {
#pragma unroll
        for (int u = 0; u < P::B3; u++) {
            const int g = t + P::NT * u;
            const int n1 = g % P::R1, e = g / P::R1;
#pragma unroll
            for (int d = 0; d < P::R3; d++)
                v[u * P::R3 + d] = lds_read(&lds[(e * P::R3 + d) * P::S2 + n1]);
            Dft<P::R3, DIR, SER>::run(&v[u * P::R3]);
            if constexpr (std::is_invocable_v<Store, int, c32>) {
#pragma unroll
                for (int f = 0; f < P::R3; f++)
                    store(g + P::R1 * P::R2 * f, v[u * P::R3 + f]);
            }
        }
        if constexpr (!std::is_invocable_v<Store, int, c32>)
            single_loop<P, 0, 0>(store, v);
    }
    stamp(8); // pass-3 arithmetic done, outputs stored (issued)
}

// Table form: tw1 is the [R1][M1] table of exp(DIR*2*pi*i*b*n1/N), tw2 the
// [R2][R3] table of exp(DIR*2*pi*i*e*d/M1).
template <class P, int DIR, class Tab, class Load, class Store>
__device__ __forceinline__ void fft3(c32 *lds, Tab tw1, Tab tw2, Load load, Store store)
{
    fft3x<P, DIR>(
        lds, load,
        [&](int, int b, c32 *v) {
#pragma unroll
            for (int n1 = 1; n1 < P::R1; n1++)
                v[n1] = cmul_v(v[n1], tw1[n1 * P::M1 + b]);
        },
        [&](int, int d, c32 *v) {
#pragma unroll
            for (int e = 1; e < P::R2; e++)
                v[e] = cmul_v(v[e], tw2[e * P::R3 + d]);
        },
        [] {}, store);
}

} // namespace afft
