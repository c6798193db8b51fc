// TEST INFRASTRUCTURE -- never loaded by the package, never part of libfbx.so.
// Compiles the per-lane routines of the packed single-qubit PGDB kernel (forest-benchmarking_amd/csrc/
// fbx_pgdb1_core.hpp: plain per-thread C++) for the HOST, so that `-m "not gpu"` can check their algebra --
// the 4 x 4 Jacobi, the Pauli butterflies, the two-matrix Dykstra, the outer loop -- against the oracle
// without a GPU.  The device build differs in its log / rsqrt / reciprocal primitives only; the GPU
// tests (tests/test_pgdb1_gpu.py) are the parity tests proper.
#define FBX_PGDB1_HOST 1
#include "../../forest-benchmarking_amd/csrc/fbx_pgdb1_core.hpp"
#include <vector>

using namespace fbx;

namespace {
struct HostDesign {
    int m, S, unit_coefs;
    const uint32_t* sp; const double* coef; const int* sptr; const double* Ct;
};
struct HostCounts {
    const double* np; const double* nm;
    double plus(int g) const { return np[g]; }
    double minus(int g) const { return nm[g]; }
};
}

extern "C" {

// eigendecomposition of one Hermitian 4 x 4 (row-major complex interleaved in, eigenvalues + eigenvectors out)
int pgdb1_host_eigh(const double* a, double* lam, double* v) {
    H4 A;
    for (int r = 0; r < 4; ++r) A.d[r] = a[(r * 4 + r) * 2];
    for (int r = 0; r < 4; ++r) for (int c = r + 1; c < 4; ++c) { A.re[h4u(r, c)] = a[(r * 4 + c) * 2]; A.im[h4u(r, c)] = a[(r * 4 + c) * 2 + 1]; }
    V4 V;
    const int sw = p1_eigh(A, V);
    for (int k = 0; k < 4; ++k) lam[k] = A.d[k];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { v[(r * 4 + c) * 2] = V.re[r][c]; v[(r * 4 + c) * 2 + 1] = V.im[r][c]; }
    return sw;
}

static H4 load_h4(const double* a) {
    H4 A;
    for (int r = 0; r < 4; ++r) A.d[r] = a[(r * 4 + r) * 2];
    for (int r = 0; r < 4; ++r) for (int c = r + 1; c < 4; ++c) { A.re[h4u(r, c)] = a[(r * 4 + c) * 2]; A.im[h4u(r, c)] = a[(r * 4 + c) * 2 + 1]; }
    return A;
}
static void store_h4(const H4& A, double* o) {
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
        double xr, xi;
        if (r == c) { xr = A.d[r]; xi = 0.0; }
        else if (r < c) { xr = A.re[h4u(r, c)]; xi = A.im[h4u(r, c)]; }
        else { xr = A.re[h4u(c, r)]; xi = -A.im[h4u(c, r)]; }
        o[(r * 4 + c) * 2] = xr; o[(r * 4 + c) * 2 + 1] = xi;
    }
}

void pgdb1_host_choi_to_pauli(const double* a, double* R) { double r[16]; p1_choi_to_pauli(load_h4(a), r); for (int k = 0; k < 16; ++k) R[k] = r[k]; }
void pgdb1_host_pauli_to_choi(const double* R, double* a) { double r[16]; for (int k = 0; k < 16; ++k) r[k] = R[k]; store_h4(p1_pauli_to_choi(r), a); }
int pgdb1_host_proj_physical(const double* a, int tp, double* out) {
    int it = 0, sw = 0, terms = 0;
    P1Basis basis; basis.valid = false; basis.chain = 0;
    store_h4(p1_proj_physical(load_h4(a), tp != 0, it, sw, terms, basis), out);
    return it;
}

// whole reconstructions; e / c in the caller's setting order, `order` maps grouped position -> caller's index
int pgdb1_host_run(int m, int S, int unit_coefs, const uint32_t* sp, const double* coef, const int* sptr, const double* Ct,
                   const int* order, long B, const double* e, const double* c, int tp, int mode, int max_iters,
                   double* choi, int* iters, int* dyk, int* bt, double* cost, int* sweeps) {
    HostDesign des{m, S, unit_coefs, sp, coef, sptr, Ct};
    std::vector<double> np(m), nm(m);
    for (long b = 0; b < B; ++b) {
        double tot = 0.0;
        for (int k = 0; k < m; ++k) tot += c[b * m + k];
        for (int g = 0; g < m; ++g) {
            const int k = order[g];
            const double plus = (1.0 + e[b * m + k]) / 2.0;
            np[g] = (c[b * m + k] * plus) / tot; nm[g] = (c[b * m + k] * (1.0 - plus)) / tot;
        }
        HostCounts nt{np.data(), nm.data()};
        P1State st;
        p1_begin(des, nt, st);
        int d_, b_;
        while (!p1_outer_iteration(des, nt, st, tp != 0, mode, max_iters, d_, b_)) {}
        store_h4(st.est, choi + b * 32);
        if (sweeps) sweeps[b] = st.sweeps;
        iters[b] = st.iters; dyk[b] = st.dyk; bt[b] = st.backtracks; cost[b] = st.new_cost;
    }
    return 0;
}

}
