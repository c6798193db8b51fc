// Micro-benchmark: 64 x 64 Hermitian Jacobi with 1024 threads -- the library's one-block-per-thread
// scheme against a role-split scheme (upper-triangle matrix blocks on 8 waves, eigenvector blocks on
// the other 8, rotations computed once and published through LDS).  Diagnostics only.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../forest-benchmarking_amd/csrc -I../../include jacobi64_bench.hip -o jacobi64_bench
#include "fbx_eigh.hpp"
#include "fbx_eigh64.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
namespace fbx { void set_error(const std::string&) {} int hip_fail(hipError_t, const char*, const char*, int) { return 2; } hipStream_t stream() { return 0; } int ensure_device() { return 0; } int device_epoch() { return 0; } }
using namespace fbx;
#ifndef NT_VALUE
#define NT_VALUE 1024
#endif
constexpr int N = 64, NB = 32, LS = 1024, NT = NT_VALUE;

__device__ int jacobi_eigh_split(cplx* Ms, cplx* Vs, double* rot, double* red, int t) {
    const bool mrole = t < 512;
    // ---- block coordinates: M role: upper blocks u = t (and 512 + t for t < 16); V role: (I, J), (I + 16, J)
    int bI[2], bJ[2];
    bool has[2] = {true, true};
    if (mrole) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = t + 512 * k;
            has[k] = u < 528;
            int I = 0, off = 0;
            if (has[k]) while (off + (NB - I) <= u) { off += NB - I; ++I; }
            bI[k] = has[k] ? I : 0; bJ[k] = has[k] ? I + (u - off) : 0;
        }
    } else {
        const int v = t - 512;
        bI[0] = v / NB; bJ[0] = v % NB; bI[1] = bI[0] + 16; bJ[1] = bJ[0];
    }
    const bool rotrole = !mrole && (t - 512) < NB;         // pivot J = t - 512
    // ---- read / write addresses, fixed for the whole decomposition
    int rd[2], w0[2][4], w1[2][4];                         // w1: adjoint seats (M role, I != J)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        rd[k] = bI[k] * NB + bJ[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int sa = jacobi_seat<N>(2 * bI[k] + (e >> 1)), sb = jacobi_seat<N>(2 * bJ[k] + (e & 1));
            if (mrole) {
                w0[k][e] = ((sa & 1) * 2 + (sb & 1)) * LS + (sa >> 1) * NB + (sb >> 1);
                w1[k][e] = ((sb & 1) * 2 + (sa & 1)) * LS + (sb >> 1) * NB + (sa >> 1);
            } else {
                w0[k][e] = ((e >> 1) * 2 + (sb & 1)) * LS + bI[k] * NB + (sb >> 1);
                w1[k][e] = 0;
            }
        }
    }
    cplx* Mine = mrole ? Ms : Vs;
    if (!mrole) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cplx c; c.re = (2 * bI[k] + (e >> 1) == 2 * bJ[k] + (e & 1)) ? 1.0 : 0.0; c.im = 0.0;
                Vs[e * LS + rd[k]] = c;
            }
    }
    __syncthreads();
    int sweep = 0;
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        {
            double o2 = 0.0, n2 = 0.0;
            if (mrole) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double wgt = !has[k] ? 0.0 : (bI[k] == bJ[k] ? 1.0 : 2.0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const cplx c = Ms[e * LS + rd[k]];
                        const double a2 = wgt * (c.re * c.re + c.im * c.im);
                        n2 += a2;
                        if (!(bI[k] == bJ[k] && (e == 0 || e == 3))) o2 += a2;
                    }
                }
            }
            block_sum2<NT>(o2, n2, red);
            if (!(uniform(o2) > FBX_JACOBI_TOL2 * uniform(n2))) break;
        }
        for (int r = 0; r < N - 1; ++r) {
            cplx x0[4], x1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { x0[e] = Mine[e * LS + rd[0]]; x1[e] = Mine[e * LS + rd[1]]; }
            if (rotrole) {
                const int J = t - 512, dJ = J * NB + J;
                const double a = Ms[0 * LS + dJ].re, d = Ms[3 * LS + dJ].re;
                const cplx b = Ms[1 * LS + dJ];
                const JRot rj = jacobi_rotation(a, d, b.re, b.im);
                rot[3 * J] = rj.c; rot[3 * J + 1] = rj.sr; rot[3 * J + 2] = rj.si;
            }
            __syncthreads();
            if (mrole) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    cplx (&x)[4] = k == 0 ? x0 : x1;
                    if (has[k]) {
                        const int I = bI[k], J = bJ[k];
                        jacobi_apply_m(rot[3 * I], rot[3 * I + 1], rot[3 * I + 2], rot[3 * J], rot[3 * J + 1], rot[3 * J + 2],
                                       x[0], x[1], x[2], x[3]);
                        if (I == J) { x[1].re = x[1].im = 0.0; x[2].re = x[2].im = 0.0; x[0].im = 0.0; x[3].im = 0.0; }
#pragma unroll
                        for (int e = 0; e < 4; ++e) Ms[w0[k][e]] = x[e];
                        if (I != J) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { cplx c = x[e]; c.im = -c.im; Ms[w1[k][e]] = c; }
                        }
                    }
                }
            } else {
                const int J = bJ[0];
                const double cJ = rot[3 * J], sJr = rot[3 * J + 1], sJi = rot[3 * J + 2];
                jacobi_apply_v(cJ, sJr, sJi, x0[0], x0[1], x0[2], x0[3]);
                jacobi_apply_v(cJ, sJr, sJi, x1[0], x1[1], x1[2], x1[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { Vs[w0[0][e]] = x0[e]; Vs[w0[1][e]] = x1[e]; }
            }
            __syncthreads();
        }
    }
    return sweep;
}

// Variant 2: the first 512 threads own TWO blocks each -- (I, J) and (I + 16, J) -- so one rotation
// chain serves two blocks and both row rotations come from lanes of the same wavefront; the other 512
// threads only take part in the barriers (in the library they would keep their registers).
__device__ int jacobi_eigh_two(cplx* Ms, cplx* Vs, double* red, int t) {
    const bool act = t < 512;
    const int I0 = act ? t / NB : 0, J = act ? t % NB : 0, I1 = I0 + 16;
    const int me0 = I0 * NB + J, me1 = I1 * NB + J, dJ = J * NB + J;
    int wm0[4], wm1[4], wv0[4], wv1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sb = jacobi_seat<N>(2 * J + (e & 1));
        const int sa0 = jacobi_seat<N>(2 * I0 + (e >> 1)), sa1 = jacobi_seat<N>(2 * I1 + (e >> 1));
        wm0[e] = ((sa0 & 1) * 2 + (sb & 1)) * LS + (sa0 >> 1) * NB + (sb >> 1);
        wm1[e] = ((sa1 & 1) * 2 + (sb & 1)) * LS + (sa1 >> 1) * NB + (sb >> 1);
        wv0[e] = ((e >> 1) * 2 + (sb & 1)) * LS + I0 * NB + (sb >> 1);
        wv1[e] = ((e >> 1) * 2 + (sb & 1)) * LS + I1 * NB + (sb >> 1);
    }
    const int lane = t & 63, src0 = lane - J + I0, src1 = lane - J + I1;      // lanes (row, J = I0) and (row, J = I1)
    if (act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cplx c; c.im = 0.0;
            c.re = (2 * I0 + (e >> 1) == 2 * J + (e & 1)) ? 1.0 : 0.0; Vs[e * LS + me0] = c;
            c.re = (2 * I1 + (e >> 1) == 2 * J + (e & 1)) ? 1.0 : 0.0; Vs[e * LS + me1] = c;
        }
    }
    __syncthreads();
    int sweep = 0;
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        {
            double o2 = 0.0, n2 = 0.0;
            if (act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const cplx a = Ms[e * LS + me0], b = Ms[e * LS + me1];
                    const double a2 = a.re * a.re + a.im * a.im, b2 = b.re * b.re + b.im * b.im;
                    n2 += a2 + b2;
                    if (!(I0 == J && (e == 0 || e == 3))) o2 += a2;
                    if (!(I1 == J && (e == 0 || e == 3))) o2 += b2;
                }
            }
            block_sum2<NT>(o2, n2, red);
            if (!(uniform(o2) > FBX_JACOBI_TOL2 * uniform(n2))) break;
        }
        for (int r = 0; r < N - 1; ++r) {
            const double aJ = Ms[0 * LS + dJ].re, dJ_ = Ms[3 * LS + dJ].re;
            const cplx bJ = Ms[1 * LS + dJ];
            cplx m0[4], m1[4], v0[4], v1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { m0[e] = Ms[e * LS + me0]; m1[e] = Ms[e * LS + me1]; v0[e] = Vs[e * LS + me0]; v1[e] = Vs[e * LS + me1]; }
            __syncthreads();
            const JRot rJ = jacobi_rotation(aJ, dJ_, bJ.re, bJ.im);
            const double c0 = __shfl(rJ.c, src0), s0r = __shfl(rJ.sr, src0), s0i = __shfl(rJ.si, src0);
            const double c1 = __shfl(rJ.c, src1), s1r = __shfl(rJ.sr, src1), s1i = __shfl(rJ.si, src1);
            jacobi_apply_m(c0, s0r, s0i, rJ.c, rJ.sr, rJ.si, m0[0], m0[1], m0[2], m0[3]);
            jacobi_apply_m(c1, s1r, s1i, rJ.c, rJ.sr, rJ.si, m1[0], m1[1], m1[2], m1[3]);
            jacobi_apply_v(rJ.c, rJ.sr, rJ.si, v0[0], v0[1], v0[2], v0[3]);
            jacobi_apply_v(rJ.c, rJ.sr, rJ.si, v1[0], v1[1], v1[2], v1[3]);
            if (I0 == J) { m0[1].re = m0[1].im = 0.0; m0[2].re = m0[2].im = 0.0; m0[0].im = 0.0; m0[3].im = 0.0; }
            if (I1 == J) { m1[1].re = m1[1].im = 0.0; m1[2].re = m1[2].im = 0.0; m1[0].im = 0.0; m1[3].im = 0.0; }
            if (act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { Ms[wm0[e]] = m0[e]; Ms[wm1[e]] = m1[e]; Vs[wv0[e]] = v0[e]; Vs[wv1[e]] = v1[e]; }
            }
            __syncthreads();
        }
    }
    return sweep;
}


// Variant 3: roles split AND out of phase.  Waves 0-7: one strictly-upper matrix block per thread (496 of them),
// both rotations computed locally from the pivot blocks (no published rotation in the matrix chain); the threads
// of block row 0 also own the pivots' new diagonal entries and publish the rotations.  Waves 8-15: two
// eigenvector blocks per thread, ONE ROUND BEHIND: between the barriers where the matrix waves compute, they
// apply the previous round's rotations and write; while the matrix waves write, they read.
struct Rot3 { double c, sr, si, pad; };
__device__ long long* g_pt = nullptr;       // [16 waves][4]: work A, wait B1, work B, wait B2 (cycles, summed)
#define PT(k) do { const long long _n = __builtin_readcyclecounter(); acc[k] += _n - tl; tl = _n; } while (0)
__device__ int jacobi_eigh_phase(cplx* Ms, cplx* Vs, Rot3* rot, double* red, int t) {
    const bool mrole = t < 512;
    int sweep = 0;
    // ---------------- matrix role set-up
    int Ih = 0, Jh = 1; bool hact = false;
    if (mrole) {
        const int k = t >> 5, c = t & 31;
        if (k == 15) { hact = c < 16; Ih = 15; Jh = hact ? 16 + c : 16; }
        else if (c < 31 - k) { hact = true; Ih = k; Jh = k + 1 + c; }
        else { hact = true; Ih = 30 - k; Jh = c; }
    }
    const bool pub = mrole && hact && Ih == 0;            // owns pivot Jh (and pivot 0 when Jh == 1)
    const bool pub0 = pub && Jh == 1;
    int wm[4]; double sg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sa = jacobi_seat<N>(2 * Ih + (e >> 1)), sb = jacobi_seat<N>(2 * Jh + (e & 1));
        const int lo = sa < sb ? sa : sb, hi = sa < sb ? sb : sa;
        wm[e] = ((lo & 1) * 2 + (hi & 1)) * LS + (lo >> 1) * NB + (hi >> 1);
        sg[e] = sa < sb ? 1.0 : -1.0;
    }
    const int meh = Ih * NB + Jh, dI = Ih * NB + Ih, dJ = Jh * NB + Jh;
    auto diag_addr = [](int s) { return ((s & 1) * 3) * LS + (s >> 1) * NB + (s >> 1); };
    auto off_addr = [](int sa, int sb) {
        const int lo = sa < sb ? sa : sb, hi = sa < sb ? sb : sa;
        return ((lo & 1) * 2 + (hi & 1)) * LS + (lo >> 1) * NB + (hi >> 1);
    };
    const int pJa = diag_addr(jacobi_seat<N>(2 * Jh)), pJd = diag_addr(jacobi_seat<N>(2 * Jh + 1));
    const int pJz = off_addr(jacobi_seat<N>(2 * Jh), jacobi_seat<N>(2 * Jh + 1));
    const int p0a = diag_addr(jacobi_seat<N>(0)), p0d = diag_addr(jacobi_seat<N>(1));
    const int p0z = off_addr(jacobi_seat<N>(0), jacobi_seat<N>(1));
    // ---------------- eigenvector role set-up
    const int v = t - 512, Iv = mrole ? 0 : v / NB, Jv = mrole ? 0 : v % NB;
    const int me0 = Iv * NB + Jv, me1 = (Iv + 16) * NB + Jv;
    int wv0[4], wv1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sb = jacobi_seat<N>(2 * Jv + (e & 1));
        wv0[e] = ((e >> 1) * 2 + (sb & 1)) * LS + Iv * NB + (sb >> 1);
        wv1[e] = ((e >> 1) * 2 + (sb & 1)) * LS + (Iv + 16) * NB + (sb >> 1);
    }
    cplx v0[4], v1[4];
    if (!mrole) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e].im = 0.0; v1[e].im = 0.0;
            v0[e].re = (2 * Iv + (e >> 1) == 2 * Jv + (e & 1)) ? 1.0 : 0.0;
            v1[e].re = (2 * (Iv + 16) + (e >> 1) == 2 * Jv + (e & 1)) ? 1.0 : 0.0;
        }
    }
    if (t < 64) { Rot3 id; id.c = 1.0; id.sr = 0.0; id.si = 0.0; id.pad = 0.0; rot[t] = id; }   // both buffers: identity
    __syncthreads();
    int rr = 0;                                           // global round counter (rotation buffer parity)
    long long acc[4] = {0, 0, 0, 0}, tl = __builtin_readcyclecounter();
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        {
            double o2 = 0.0, n2 = 0.0;
            if (mrole && hact) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const cplx c = Ms[e * LS + meh]; o2 += 2.0 * (c.re * c.re + c.im * c.im); }
                if (pub) {
                    const cplx b = Ms[1 * LS + dJ]; const double a = Ms[0 * LS + dJ].re, d = Ms[3 * LS + dJ].re;
                    o2 += 2.0 * (b.re * b.re + b.im * b.im); n2 += a * a + d * d;
                }
                if (pub0) {
                    const cplx b = Ms[1 * LS + dI]; const double a = Ms[0 * LS + dI].re, d = Ms[3 * LS + dI].re;
                    o2 += 2.0 * (b.re * b.re + b.im * b.im); n2 += a * a + d * d;
                }
                n2 += o2;
            }
            block_sum2<NT>(o2, n2, red);
            if (!(uniform(o2) > FBX_JACOBI_TOL2 * uniform(n2))) break;
        }
        tl = __builtin_readcyclecounter();
        for (int r = 0; r < N - 1; ++r, ++rr) {
            cplx m00, m01, m10, m11; JRot rI, rJ;
            if (mrole) {
                if (hact) {
                    const double aJ = Ms[0 * LS + dJ].re, dJ_ = Ms[3 * LS + dJ].re; const cplx bJ = Ms[1 * LS + dJ];
                    const double aI = Ms[0 * LS + dI].re, dI_ = Ms[3 * LS + dI].re; const cplx bI = Ms[1 * LS + dI];
                    m00 = Ms[0 * LS + meh]; m01 = Ms[1 * LS + meh]; m10 = Ms[2 * LS + meh]; m11 = Ms[3 * LS + meh];
                    rJ = jacobi_rotation(aJ, dJ_, bJ.re, bJ.im);
                    rI = jacobi_rotation(aI, dI_, bI.re, bI.im);
                    if (pub) { Rot3 o; o.c = rJ.c; o.sr = rJ.sr; o.si = rJ.si; o.pad = 0.0; rot[(rr & 1) * 32 + Jh] = o; }
                    if (pub0) { Rot3 o; o.c = rI.c; o.sr = rI.sr; o.si = rI.si; o.pad = 0.0; rot[(rr & 1) * 32] = o; }
                    jacobi_apply_m(rI.c, rI.sr, rI.si, rJ.c, rJ.sr, rJ.si, m00, m01, m10, m11);
                }
            } else if (rr > 0) {
                const Rot3 q = rot[((rr + 1) & 1) * 32 + Jv];      // rotation of the PREVIOUS round, in that round's seats
                jacobi_apply_v(q.c, q.sr, q.si, v0[0], v0[1], v0[2], v0[3]);
                jacobi_apply_v(q.c, q.sr, q.si, v1[0], v1[1], v1[2], v1[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { Vs[wv0[e]] = v0[e]; Vs[wv1[e]] = v1[e]; }
            }
            PT(0);
            __syncthreads();
            PT(1);
            if (mrole) {
                if (hact) {
                    m00.im *= sg[0]; m01.im *= sg[1]; m10.im *= sg[2]; m11.im *= sg[3];
                    Ms[wm[0]] = m00; Ms[wm[1]] = m01; Ms[wm[2]] = m10; Ms[wm[3]] = m11;
                    if (pub) {
                        cplx a; a.re = rJ.an; a.im = 0.0; cplx d; d.re = rJ.dn; d.im = 0.0; cplx z; z.re = 0.0; z.im = 0.0;
                        Ms[pJa] = a; Ms[pJd] = d; Ms[pJz] = z;
                    }
                    if (pub0) {
                        cplx a; a.re = rI.an; a.im = 0.0; cplx d; d.re = rI.dn; d.im = 0.0; cplx z; z.re = 0.0; z.im = 0.0;
                        Ms[p0a] = a; Ms[p0d] = d; Ms[p0z] = z;
                    }
                }
            } else if (rr > 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = Vs[e * LS + me0]; v1[e] = Vs[e * LS + me1]; }
            }
            PT(2);
            __syncthreads();
            PT(3);
        }
    }
    if ((t & 63) == 0 && g_pt) { for (int k = 0; k < 4; ++k) atomicAdd((unsigned long long*)&g_pt[(t >> 6) * 4 + k], (unsigned long long)acc[k]); }
    // the eigenvector waves are one rotation behind
    if (!mrole && rr > 0) {
        const Rot3 q = rot[((rr + 1) & 1) * 32 + Jv];
        jacobi_apply_v(q.c, q.sr, q.si, v0[0], v0[1], v0[2], v0[3]);
        jacobi_apply_v(q.c, q.sr, q.si, v1[0], v1[1], v1[2], v1[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { Vs[wv0[e]] = v0[e]; Vs[wv1[e]] = v1[e]; }
    }
    __syncthreads();
    return sweep;
}

template <int MODE>
__global__ void __launch_bounds__(NT) k64(const double* A, double* W, double* Vout, long long* cyc, int* sw, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Ms = (cplx*)smem; cplx* Vs = Ms + N * N;
    double* red = (double*)(Vs + N * N); double* rot = red + 64; Rot3* rot3 = (Rot3*)(rot + 3 * 32 + 8);
    const int t = threadIdx.x, item = blockIdx.x;
    long long total = 0; int sweeps = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int idx = t; idx < N * N; idx += NT) {
            cplx c; c.re = A[((size_t)item * N * N + idx) * 2]; c.im = A[((size_t)item * N * N + idx) * 2 + 1];
            Ms[sys_index<N>(idx / N, idx % N)] = c;
        }
        __syncthreads();
        long long t0 = __builtin_readcyclecounter();
        if constexpr (MODE == 0 && NT >= 1024) sweeps += jacobi_eigh_simple<N, (NT >= 1024 ? NT : 1024)>(Ms, Vs, t, true, red);
        else if constexpr (MODE == 1) sweeps += jacobi_eigh_split(Ms, Vs, rot, red, t);
        else if constexpr (MODE == 3) sweeps += jacobi_eigh_phase(Ms, Vs, rot3, red, t);
        else if constexpr (MODE == 4) { if constexpr (NT >= 1024) sweeps += jacobi_eigh64<1024>(Ms, Vs, t, true, red); }
        else sweeps += jacobi_eigh_two(Ms, Vs, red, t);
        total += __builtin_readcyclecounter() - t0;
        __syncthreads();
    }
    if (t < N) W[item * N + t] = Ms[sys_index<N>(t, t)].re;
    for (int idx = t; idx < N * N; idx += NT) {
        const cplx c = Vs[sys_index<N>(idx / N, idx % N)];
        Vout[((size_t)item * N * N + idx) * 2] = c.re; Vout[((size_t)item * N * N + idx) * 2 + 1] = c.im;
    }
    if (t == 0) { cyc[item] = total; sw[item] = sweeps; }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, B = 256, reps = 4;
    std::vector<double> A((size_t)B * N * N * 2);
    srand(1);
    for (int b = 0; b < B; ++b) {
        std::vector<double> g(N * N * 2);
        for (auto& x : g) x = (rand() / (double)RAND_MAX) - 0.5;
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
            A[((size_t)b * N * N + i * N + j) * 2] = g[(i * N + j) * 2] + g[(j * N + i) * 2];
            A[((size_t)b * N * N + i * N + j) * 2 + 1] = g[(i * N + j) * 2 + 1] - g[(j * N + i) * 2 + 1];
        }
    }
    double *dA, *dW, *dV; long long* dc; int* ds;
    (void)hipMalloc(&dA, A.size() * 8); (void)hipMalloc(&dW, B * N * 8); (void)hipMalloc(&dV, A.size() * 8);
    (void)hipMalloc(&dc, B * 8); (void)hipMalloc(&ds, B * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = 2 * sizeof(cplx) * N * N + sizeof(double) * (64 + 3 * 32 + 8) + 64 * 32;
    (void)hipFuncSetAttribute((const void*)k64<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k64<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k64<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k64<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k64<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    { long long* dp; (void)hipMalloc(&dp, 64 * 8); (void)hipMemset(dp, 0, 64 * 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pt), &dp, sizeof dp); }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 2; ++it) {
        (void)hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k64<0>, dim3(B), dim3(NT), lds, 0, dA, dW, dV, dc, ds, reps);
        else if (mode == 1) hipLaunchKernelGGL(k64<1>, dim3(B), dim3(NT), lds, 0, dA, dW, dV, dc, ds, reps);
        else if (mode == 3) hipLaunchKernelGGL(k64<3>, dim3(B), dim3(NT), lds, 0, dA, dW, dV, dc, ds, reps);
        else if (mode == 4) hipLaunchKernelGGL(k64<4>, dim3(B), dim3(NT), lds, 0, dA, dW, dV, dc, ds, reps);
        else hipLaunchKernelGGL(k64<2>, dim3(B), dim3(NT), lds, 0, dA, dW, dV, dc, ds, reps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> c(B); std::vector<int> s(B); std::vector<double> W(B * N), V(A.size());
    (void)hipMemcpy(c.data(), dc, B * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(s.data(), ds, B * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(W.data(), dW, B * N * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(V.data(), dV, A.size() * 8, hipMemcpyDeviceToHost);
    double csum = 0, ssum = 0; for (int b = 0; b < B; ++b) { csum += c[b]; ssum += s[b]; }
    double res = 0, orth = 0;
    for (int i = 0; i < N; ++i) for (int k = 0; k < N; ++k) {
        double re = 0, im = 0, o_re = 0, o_im = 0;
        for (int j = 0; j < N; ++j) {
            const double ar = A[(i * N + j) * 2], ai = A[(i * N + j) * 2 + 1], vr = V[(j * N + k) * 2], vi = V[(j * N + k) * 2 + 1];
            re += ar * vr - ai * vi; im += ar * vi + ai * vr;
            const double ur = V[(j * N + i) * 2], ui = V[(j * N + i) * 2 + 1];
            o_re += ur * vr + ui * vi; o_im += ur * vi - ui * vr;
        }
        re -= V[(i * N + k) * 2] * W[k]; im -= V[(i * N + k) * 2 + 1] * W[k];
        res = fmax(res, sqrt(re * re + im * im));
        if (i == k) o_re -= 1.0;
        orth = fmax(orth, sqrt(o_re * o_re + o_im * o_im));
    }
    if (mode == 3) {
        long long h[64]; long long* dp = nullptr;
        (void)hipMemcpyFromSymbol(&dp, HIP_SYMBOL(g_pt), sizeof dp);
        (void)hipMemcpy(h, dp, sizeof h, hipMemcpyDeviceToHost);
        const double rounds = ssum * (N - 1) * 2;     // two launches accumulate
        for (int w : {0, 3, 7, 8, 12, 15}) printf("  wave %2d: work A %.0f  wait B1 %.0f  work B %.0f  wait B2 %.0f cycles per round\n", w,
            h[w * 4] / rounds, h[w * 4 + 1] / rounds, h[w * 4 + 2] / rounds, h[w * 4 + 3] / rounds);
    }
    unsigned long long hash = 1469598103934665603ull;           // FNV-1a over the bits of every eigenvalue and eigenvector entry (A/B builds: bit-identity)
    auto mix = [&](const std::vector<double>& x) { for (double v : x) { unsigned long long u; memcpy(&u, &v, 8); hash = (hash ^ u) * 1099511628211ull; } };
    mix(W); mix(V);
    printf("mode %d: kernel %.3f ms; per eigh %.0f cycles, %.2f sweeps, %.0f cycles/round; residual %.2e, orthogonality %.2e; bits %016llx\n",
           mode, ms, csum / B / reps, ssum / B / reps, csum / ssum / (N - 1), res, orth, hash);
    return 0;
}
