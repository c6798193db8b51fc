// fbx_shots.hip -- shots -> observable moments (observable_estimation.py:804-853): a byte-stream
// reduction.  Every lane streams 16-byte pieces of the [n_shots][n_qubits] 0/1 byte array (coalesced), masks the
// observable's columns, folds the bytes of each shot with XOR (parity = eigenvalue sign) and counts the -1
// outcomes with popcounts; integer counts are reduced through the wave (and LDS).  One WAVEFRONT per setting
// while a setting's record is short (the usual 1000 shots x 2 qubits = 2 KB: four settings in flight per
// workgroup and no barrier -- a workgroup per setting ran at 1.0 TB/s there), one 256-thread workgroup per
// setting for long records.  HBM-bound integer work: n_qubits bytes per shot in, 16 bytes per setting out.
#include "fbx_common.hpp"

namespace fbx {

// number of shots with odd parity in one 64-bit word holding 8 / NQB shots of NQB masked bytes
template <int NQB>
__device__ __forceinline__ int odd_shots(unsigned long long x) {
    if (NQB == 1) return __popcll(x & 0x0101010101010101ull);
    if (NQB == 2) { x ^= x >> 8; return __popcll(x & 0x0001000100010001ull); }
    if (NQB == 4) { x ^= x >> 16; x ^= x >> 8; return __popcll(x & 0x0000000100000001ull); }
    x ^= x >> 32; x ^= x >> 16; x ^= x >> 8; return (int)(x & 1ull);      // NQB == 8
}

template <int NQB>      // NQB in {1, 2, 4, 8}: bytes per shot, 16-byte vector path
__device__ long long count_minus_vec(const uint8_t* __restrict__ bits, long long n_shots, unsigned long long pat, int tid, int nth) {
    const long long total = n_shots * NQB;
    const long long nvec = total / 16;
    const ulonglong2* v = reinterpret_cast<const ulonglong2*>(bits);
    long long cnt = 0;
    for (long long i = tid; i < nvec; i += nth) {
        const ulonglong2 w = v[i];
        cnt += odd_shots<NQB>(w.x & pat) + odd_shots<NQB>(w.y & pat);
    }
    // tail (fewer than 16 bytes): whole shots, byte-wise, by one lane
    if (tid == 0) {
        for (long long s = nvec * 16 / NQB; s < n_shots; ++s) {
            int par = 0;
            for (int q = 0; q < NQB; ++q) par ^= bits[s * NQB + q] & (int)((pat >> (8 * q)) & 1);
            cnt += par;
        }
    }
    return cnt;
}

// Qubit counts that do not divide 16 (3, 5, 6, 7): every lane takes runs of 16 shots = 16 NQ bytes = 2 NQ 64-bit words,
// masks the observable's columns with a pattern of period NQ laid over the run, and counts a shot as -1 when the
// population of its byte range is odd (the bytes are 0 / 1).  Needs the record 8-byte aligned.
template <int NQ>
__device__ long long count_minus_packed(const uint8_t* __restrict__ bits, long long n_shots, const uint8_t* __restrict__ mask,
                                        int tid, int nth) {
    constexpr int W = 2 * NQ;                       // words per run of 16 shots
    unsigned long long pat[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        unsigned long long m = 0;
#pragma unroll
        for (int byte = 0; byte < 8; ++byte) m |= (unsigned long long)(mask[(8 * w + byte) % NQ] ? 1 : 0) << (8 * byte);
        pat[w] = m;
    }
    const long long runs = n_shots / 16;
    const unsigned long long* v = reinterpret_cast<const unsigned long long*>(bits);
    long long cnt = 0;
    for (long long r = tid; r < runs; r += nth) {
        unsigned long long x[W];
#pragma unroll
        for (int w = 0; w < W; ++w) x[w] = v[r * W + w] & pat[w];
#pragma unroll
        for (int j = 0; j < 16; ++j) {              // shot j: bytes [j NQ, j NQ + NQ)
            constexpr unsigned long long ALL = ~0ull;
            const int lo = j * NQ, hi = lo + NQ - 1, w0 = lo >> 3, w1 = hi >> 3;
            int pop;
            if (w0 == w1) {
                const unsigned long long m = (ALL >> (8 * (7 - (hi & 7)))) & (ALL << (8 * (lo & 7)));
                pop = __popcll(x[w0] & m);
            } else {
                pop = __popcll(x[w0] & (ALL << (8 * (lo & 7)))) + __popcll(x[w1] & (ALL >> (8 * (7 - (hi & 7)))));
            }
            cnt += pop & 1;
        }
    }
    for (long long s = runs * 16 + tid; s < n_shots; s += nth) {       // the last n_shots % 16 shots, byte-wise
        int par = 0;
        for (int q = 0; q < NQ; ++q) par ^= (bits[s * NQ + q] & 1) & (mask[q] ? 1 : 0);
        cnt += par;
    }
    return cnt;
}

__device__ long long count_minus_generic(const uint8_t* __restrict__ bits, long long n_shots, int n,
                                         const uint8_t* __restrict__ mask, int tid, int nth) {
    long long cnt = 0;
    for (long long s = tid; s < n_shots; s += nth) {
        int par = 0;
        for (int q = 0; q < n; ++q) par ^= (bits[s * n + q] & 1) & (mask[q] ? 1 : 0);
        cnt += par;
    }
    return cnt;
}

// Short records whose size is a multiple of 16 bytes (the usual 1000 shots x 2 qubits = 125 vectors): a wavefront per
// setting with the NEXT setting's record requested before the current one is reduced.  A wavefront that loads, waits,
// reduces and stores one 2 KB record at a time keeps 2 KB in flight; 32 wavefronts per CU then hold 16 MB on the chip --
// just the 8 TB/s x 2 us the memory system needs -- and every reduction / store phase is a bubble (2.4-2.9 TB/s).
// VPL = vectors per lane (record <= 1 / 2 / 4 / 8 KB).
template <int NQB, int VPL>
__device__ __forceinline__ void shots_wave_pipeline(long long first, long long stride, long long n_settings, long long n_shots,
                                                    const uint8_t* __restrict__ bits, const uint8_t* __restrict__ obs_mask,
                                                    const double* __restrict__ coefs, int beta_prior,
                                                    double* __restrict__ mean_out, double* __restrict__ var_out, int lane) {
    const long long nvec = n_shots * NQB / 16;
    auto fetch = [&](long long s, ulonglong2 (&buf)[VPL], unsigned long long& pat) __attribute__((always_inline)) {
        const ulonglong2* v = reinterpret_cast<const ulonglong2*>(bits + s * n_shots * NQB);
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const long long i = lane + 64 * k;
            buf[k] = i < nvec ? v[i] : ulonglong2{0ull, 0ull};
        }
        const uint8_t* mk = obs_mask + s * NQB;
        pat = 0;
#pragma unroll
        for (int byte = 0; byte < 8; ++byte) pat |= (unsigned long long)(mk[byte % NQB] ? 1 : 0) << (8 * byte);
    };
    ulonglong2 cur[VPL], nxt[VPL];
    unsigned long long pat = 0, pat_n = 0;
    long long s = first;
    if (s < n_settings) fetch(s, cur, pat);
    while (s < n_settings) {
        const long long sn = s + stride;
        if (sn < n_settings) fetch(sn, nxt, pat_n);
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < VPL; ++k) cnt += odd_shots<NQB>(cur[k].x & pat) + odd_shots<NQB>(cur[k].y & pat);
        const long long n_minus = (long long)wave_sum((double)cnt);
        if (lane == 0) {
            const long long n_plus = n_shots - n_minus;
            const double coef = coefs ? coefs[s] : 1.0;
            double mean, var;
            if (pat == 0) { mean = coef; var = 0.0; }
            else if (beta_prior) {
                const double a = (double)n_plus + 1.0, bb = (double)n_minus + 1.0;
                const double bm = a / (a + bb), bv = a * bb / ((a + bb) * (a + bb) * (a + bb + 1.0));
                mean = coef * (2.0 * bm - 1.0); var = coef * coef * 4.0 * bv;
            } else {
                const double m = ((double)n_plus - (double)n_minus) / (double)n_shots;
                mean = coef * m;
                var = coef * coef * (1.0 - m * m) / (double)n_shots;
            }
            mean_out[s] = mean; var_out[s] = var;
        }
#pragma unroll
        for (int k = 0; k < VPL; ++k) cur[k] = nxt[k];
        pat = pat_n; s = sn;
    }
}

// one instantiation per record size class, so that the short records keep few registers and many wavefronts in flight
template <int NQB, int VPL>
__global__ void __launch_bounds__(256)
shots_pipe_kernel(long long n_settings, long long n_shots, const uint8_t* __restrict__ bits, const uint8_t* __restrict__ obs_mask,
                  const double* __restrict__ coefs, int beta_prior, double* __restrict__ mean_out, double* __restrict__ var_out) {
    shots_wave_pipeline<NQB, VPL>((long long)blockIdx.x * 4 + (threadIdx.x >> 6), (long long)gridDim.x * 4, n_settings, n_shots, bits,
                                  obs_mask, coefs, beta_prior, mean_out, var_out, threadIdx.x & 63);
}

// The same pipeline for the packed qubit counts 3 and 5 (6 and 7 measured no faster than shots_kernel: 110-180 registers): a lane's unit is a run of 16 shots = 2 NQ words; records of at
// most RPL x 64 runs (RPL = runs per lane) plus a byte-wise tail of n_shots % 16 shots, 8-byte aligned.  Own kernels, like
// shots_pipe_kernel: inside shots_kernel the two register buffers cost every path its occupancy.
template <int NQ, int RPL>
__global__ void __launch_bounds__(256)
shots_pipe_packed_kernel(long long n_settings, long long n_shots, const uint8_t* __restrict__ bits, const uint8_t* __restrict__ obs_mask,
                         const double* __restrict__ coefs, int beta_prior, double* __restrict__ mean_out, double* __restrict__ var_out) {
    constexpr int W = 2 * NQ;
    const int lane = threadIdx.x & 63;
    const long long first = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), stride = (long long)gridDim.x * 4;
    const long long runs = n_shots / 16, tail0 = runs * 16;
    unsigned long long cur[RPL][W], nxt[RPL][W];
    int tail_c = 0, tail_n = 0;
    auto fetch = [&](long long s, unsigned long long (&x)[RPL][W], int& tail) __attribute__((always_inline)) {
        const uint8_t* b = bits + s * n_shots * NQ;
        const unsigned long long* v = reinterpret_cast<const unsigned long long*>(b);
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            const long long run = lane + 64 * k;
#pragma unroll
            for (int w = 0; w < W; ++w) x[k][w] = run < runs ? v[run * W + w] : 0ull;
        }
        tail = 0;                                              // the last n_shots % 16 shots: one per lane, masked later
        const long long sh = tail0 + lane;
        if (sh < n_shots) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) tail |= (int)(b[sh * NQ + q] & 1) << q;
        }
    };
    long long s = first;
    if (s < n_settings) fetch(s, cur, tail_c);
    while (s < n_settings) {
        const long long sn = s + stride;
        if (sn < n_settings) fetch(sn, nxt, tail_n);
        const uint8_t* mk = obs_mask + s * NQ;
        unsigned long long pat[W];
        int mbits = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) mbits |= (mk[q] ? 1 : 0) << q;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            unsigned long long m = 0;
#pragma unroll
            for (int byte = 0; byte < 8; ++byte) m |= (unsigned long long)((mbits >> ((8 * w + byte) % NQ)) & 1) << (8 * byte);
            pat[w] = m;
        }
        int cnt = __popc(tail_c & mbits) & 1;
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {                      // shot j of the run: bytes [j NQ, j NQ + NQ)
                constexpr unsigned long long ALL = ~0ull;
                const int lo = j * NQ, hi = lo + NQ - 1, w0 = lo >> 3, w1 = hi >> 3;
                int pop;
                if (w0 == w1) pop = __popcll(cur[k][w0] & pat[w0] & (ALL >> (8 * (7 - (hi & 7)))) & (ALL << (8 * (lo & 7))));
                else pop = __popcll(cur[k][w0] & pat[w0] & (ALL << (8 * (lo & 7)))) + __popcll(cur[k][w1] & pat[w1] & (ALL >> (8 * (7 - (hi & 7)))));
                cnt += pop & 1;
            }
        }
        const long long n_minus = (long long)wave_sum((double)cnt);
        if (lane == 0) {
            const long long n_plus = n_shots - n_minus;
            const double coef = coefs ? coefs[s] : 1.0;
            double mean, var;
            if (mbits == 0) { mean = coef; var = 0.0; }
            else if (beta_prior) {
                const double a = (double)n_plus + 1.0, bb = (double)n_minus + 1.0;
                const double bm = a / (a + bb), bv = a * bb / ((a + bb) * (a + bb) * (a + bb + 1.0));
                mean = coef * (2.0 * bm - 1.0); var = coef * coef * 4.0 * bv;
            } else {
                const double m = ((double)n_plus - (double)n_minus) / (double)n_shots;
                mean = coef * m;
                var = coef * coef * (1.0 - m * m) / (double)n_shots;
            }
            mean_out[s] = mean; var_out[s] = var;
        }
#pragma unroll
        for (int k = 0; k < RPL; ++k)
#pragma unroll
            for (int w = 0; w < W; ++w) cur[k][w] = nxt[k][w];
        tail_c = tail_n; s = sn;
    }
}

template <bool PER_WAVE>
__global__ void __launch_bounds__(256)
shots_kernel(int n, long long n_settings, long long n_shots, const uint8_t* __restrict__ bits,
             const uint8_t* __restrict__ obs_mask, const double* __restrict__ coefs, int beta_prior,
             double* __restrict__ mean_out, double* __restrict__ var_out) {
    __shared__ long long part[4];
    const int tid = PER_WAVE ? (threadIdx.x & 63) : threadIdx.x, nth = PER_WAVE ? 64 : 256;
    const long long first = PER_WAVE ? (long long)blockIdx.x * 4 + (threadIdx.x >> 6) : blockIdx.x;
    const long long stride = PER_WAVE ? (long long)gridDim.x * 4 : gridDim.x;
    for (long long s = first; s < n_settings; s += stride) {
        const uint8_t* mk = obs_mask + s * n;
        const uint8_t* b = bits + s * n_shots * n;
        bool any = false;
        unsigned long long pat = 0;
        for (int q = 0; q < n; ++q) any |= mk[q] != 0;
        long long cnt = 0;
        const bool aligned = ((uintptr_t)b & 15) == 0;
        if (any) {
            if ((n == 1 || n == 2 || n == 4 || n == 8) && aligned) {
                for (int byte = 0; byte < 8; ++byte) pat |= (unsigned long long)(mk[byte % n] ? 1 : 0) << (8 * byte);
                if (n == 1) cnt = count_minus_vec<1>(b, n_shots, pat, tid, nth);
                else if (n == 2) cnt = count_minus_vec<2>(b, n_shots, pat, tid, nth);
                else if (n == 4) cnt = count_minus_vec<4>(b, n_shots, pat, tid, nth);
                else cnt = count_minus_vec<8>(b, n_shots, pat, tid, nth);
            } else if ((n == 3 || n == 5 || n == 6 || n == 7) && ((uintptr_t)b & 7) == 0) {
                if (n == 3) cnt = count_minus_packed<3>(b, n_shots, mk, tid, nth);
                else if (n == 5) cnt = count_minus_packed<5>(b, n_shots, mk, tid, nth);
                else if (n == 6) cnt = count_minus_packed<6>(b, n_shots, mk, tid, nth);
                else cnt = count_minus_packed<7>(b, n_shots, mk, tid, nth);
            } else {
                cnt = count_minus_generic(b, n_shots, n, mk, tid, nth);
            }
        }
        // wave reduction of the integer count (exact in a double: counts stay below 2^53)
        double c = (double)cnt;
        c = wave_sum(c);
        long long n_minus = (long long)c;
        if (!PER_WAVE) {
            __syncthreads();
            if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (long long)c;
            __syncthreads();
            n_minus = part[0] + part[1] + part[2] + part[3];
        }
        if (tid == 0) {
            const long long n_plus = n_shots - n_minus;
            const double coef = coefs ? coefs[s] : 1.0;
            double mean, var;
            if (!any) { mean = coef; var = 0.0; }                       // identity term (:826-827)
            else if (beta_prior) {                                       // :837-846
                const double a = (double)n_plus + 1.0, bb = (double)n_minus + 1.0;
                const double bm = a / (a + bb), bv = a * bb / ((a + bb) * (a + bb) * (a + bb + 1.0));
                mean = coef * (2.0 * bm - 1.0); var = coef * coef * 4.0 * bv;
            } else {                                                     // :848-850
                const double m = ((double)n_plus - (double)n_minus) / (double)n_shots;
                mean = coef * m;
                var = coef * coef * (1.0 - m * m) / (double)n_shots;
            }
            mean_out[s] = mean; var_out[s] = var;
        }
    }
}

// direct fidelity estimate (direct_fidelity_estimation.py:224-307): per experiment the mean of the m
// expectations and the sum of the squared standard errors, mapped to a state / average gate fidelity.
// One wavefront per experiment, coalesced 8-byte loads; HBM-bound (16 B per setting).
__global__ void __launch_bounds__(64)
dfe_kernel(int n_qubits, int process, long long B, long long m, const double* __restrict__ expect,
           const double* __restrict__ std_err, double* __restrict__ mean_out, double* __restrict__ err_out) {
    const int lane = threadIdx.x;
    const double d = (double)(1 << n_qubits);
    for (long long item = blockIdx.x; item < B; item += gridDim.x) {
        const double* e = expect + item * m;
        const double* se = std_err + item * m;
        double s = 0.0, v = 0.0;
        for (long long k = lane; k < m; k += 64) { s += e[k]; const double x = se[k]; v += x * x; }
        s = wave_sum(s); v = wave_sum(v);
        if (lane == 0) {
            const double mean = s / (double)m;
            const double var_mean = v / ((double)m * (double)m);
            if (!process) {
                mean_out[item] = (d - 1.0) / d * mean + 1.0 / d;
                err_out[item] = sqrt((d - 1.0) * (d - 1.0) / (d * d) * var_mean);
            } else {
                const double d2 = d * d;
                const double p_mean = (d2 - 1.0) / d2 * mean + 1.0 / d2;
                mean_out[item] = (d2 * p_mean + d) / (d2 + d);
                err_out[item] = sqrt(d2 / ((d + 1.0) * (d + 1.0)) * (d2 - 1.0) * (d2 - 1.0) / (d2 * d2) * var_mean);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Bootstrap resampling (tomography.py:378-409): e' = 2 Beta(n_plus + prior, n_minus + prior) - 1 with
// n_plus = (e + 1) / 2 * counts.  The reference draws from numpy's global stream; here every output
// element (r, i) owns a counter-based Philox4x32-10 stream keyed by the caller's seed, so the result
// does not depend on the launch shape or on R (parity with the reference is statistical; the
// generator itself is bit-exact against oracle/fbx_oracle/acquisition.py).  Beta = Ga / (Ga + Gb),
// gammas by Marsaglia-Tsang squeeze-free rejection, one Philox block (two 32-bit uniforms for the
// Box-Muller normal, one 53-bit uniform for the acceptance test) per attempt.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
struct PhiloxStream {
    uint32_t k0, k1, e0, e1, draw;
    __device__ __forceinline__ void next(double& u1, double& u2, double& u3) {
        uint32_t c[4] = {e0, e1, draw++, 0u};
        philox4x32_10(c, k0, k1);
        u1 = ((double)c[0] + 0.5) * 0x1p-32;
        u2 = ((double)c[1] + 0.5) * 0x1p-32;
        u3 = ((double)(((unsigned long long)(c[2] >> 5) << 26) | (unsigned long long)(c[3] >> 6)) + 0.5) * 0x1p-53;
    }
};
constexpr int FBX_GAMMA_MAX_TRIES = 64;     // acceptance >= 0.95 per try
__device__ double gamma_marsaglia_tsang(double a, PhiloxStream& s) {
    double boost = 1.0, u1, u2, u3;
    if (a < 1.0) { s.next(u1, u2, u3); boost = pow(u3, 1.0 / a); a += 1.0; }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int tries = 0; tries < FBX_GAMMA_MAX_TRIES; ++tries) {
        s.next(u1, u2, u3);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(u3) < 0.5 * x * x + d - d * v + d * log(v)) return boost * d * v;
    }
    return boost * d;
}
__global__ void __launch_bounds__(256)
beta_resample_kernel(long long n, long long R, const double* __restrict__ expect, const double* __restrict__ counts,
                     double prior, unsigned long long seed, double* __restrict__ out, double* __restrict__ counts_out) {
    const long long total = n * R;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long i = idx % n;
        const double e = expect[i], cnt = counts[i];
        const double n_plus = ((e + 1.0) / 2.0) * cnt, n_minus = cnt - n_plus;
        const double a = n_plus + prior, b = n_minus + prior;
        double val = __builtin_nan("");
        if (a > 0.0 && b > 0.0) {
            PhiloxStream s{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)idx, (uint32_t)((unsigned long long)idx >> 32), 0u};
            const double ga = gamma_marsaglia_tsang(a, s);
            const double gb = gamma_marsaglia_tsang(b, s);
            val = 2.0 * (ga / (ga + gb)) - 1.0;
        }
        out[idx] = val;
        if (counts_out) counts_out[idx] = cnt;
    }
}

// Readout-calibration rescale (observable_estimation.py:1028-1037 + ratio_variance :1052-1090):
// corrected mean = e / c and its standard error sqrt(se^2 / c^2 + e^2 var_c / c^4), element-wise over
// B experiments x m settings; the calibration (c, var_c) of a setting's observable is shared by the batch
// (cal index [m], or NULL for one calibration per setting in order).  16 B in, 16 B out per element.
__global__ void __launch_bounds__(256)
calibrate_kernel(long long B, long long m, const double* __restrict__ expect, const double* __restrict__ std_err,
                 const int* __restrict__ cal_index, const double* __restrict__ cal_mean, const double* __restrict__ cal_var,
                 double* __restrict__ mean_out, double* __restrict__ err_out) {
    const long long total = B * m;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long k = idx % m;
        const int ci = cal_index ? cal_index[k] : (int)k;
        const double a = expect[idx], se = std_err[idx], b = cal_mean[ci], vb = cal_var[ci];
        const double va = se * se;
        const double b2 = b * b;
        mean_out[idx] = a / b;
        err_out[idx] = sqrt(va / b2 + (a * a * vb) / (b2 * b2));
    }
}

}  // namespace fbx

using namespace fbx;

extern "C" {

int fbx_calibrate_expectations_dev(int64_t B, int64_t m, const double* d_expect, const double* d_std_err,
                                   const int32_t* d_cal_index, int64_t n_cal, const double* d_cal_mean,
                                   const double* d_cal_var, double* d_mean_out, double* d_err_out) {
    FBX_REQUIRE(B >= 0 && m >= 0 && n_cal >= 0, "fbx_calibrate_expectations: negative size");
    FBX_REQUIRE(B * m == 0 || (d_expect && d_std_err && d_cal_mean && d_cal_var && d_mean_out && d_err_out),
                "fbx_calibrate_expectations: NULL buffer");
    FBX_REQUIRE(d_cal_index != nullptr || n_cal == m, "fbx_calibrate_expectations: without cal_index there must be one calibration per setting");
    int rc = ensure_device();
    if (rc) return rc;
    if (B * m == 0) return FBX_OK;
    const long long total = (long long)B * m, want = (total + 255) / 256;
    hipLaunchKernelGGL(calibrate_kernel, dim3((unsigned)(want < 256 * 32 ? want : 256 * 32)), dim3(256), 0, stream(),
                       (long long)B, (long long)m, d_expect, d_std_err, (const int*)d_cal_index, d_cal_mean, d_cal_var,
                       d_mean_out, d_err_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_calibrate_expectations(int64_t B, int64_t m, const double* expect, const double* std_err,
                               const int32_t* cal_index, int64_t n_cal, const double* cal_mean,
                               const double* cal_var, double* mean_out, double* err_out) {
    FBX_REQUIRE(B >= 0 && m >= 0 && n_cal >= 0, "fbx_calibrate_expectations: negative size");
    FBX_REQUIRE(B * m == 0 || (expect && std_err && cal_mean && cal_var && mean_out && err_out),
                "fbx_calibrate_expectations: NULL buffer");
    FBX_REQUIRE(cal_index != nullptr || n_cal == m, "fbx_calibrate_expectations: without cal_index there must be one calibration per setting");
    if (cal_index)
        for (int64_t k = 0; k < m; ++k)
            FBX_REQUIRE(cal_index[k] >= 0 && cal_index[k] < n_cal, "fbx_calibrate_expectations: calibration index out of range");
    int rc = ensure_device();
    if (rc) return rc;
    if (B * m == 0) return FBX_OK;
    const size_t n = (size_t)B * m;
    DevBuf de, ds, di, dcm, dcv, dm, dr;
    if ((rc = de.alloc(sizeof(double) * n)) || (rc = ds.alloc(sizeof(double) * n)) || (rc = di.alloc(sizeof(int32_t) * (m ? m : 1))) ||
        (rc = dcm.alloc(sizeof(double) * (n_cal ? n_cal : 1))) || (rc = dcv.alloc(sizeof(double) * (n_cal ? n_cal : 1))) ||
        (rc = dm.alloc(sizeof(double) * n)) || (rc = dr.alloc(sizeof(double) * n)))
        return rc;
    FBX_HIP(hipMemcpyAsync(de.p, expect, sizeof(double) * n, hipMemcpyHostToDevice, stream()));
    FBX_HIP(hipMemcpyAsync(ds.p, std_err, sizeof(double) * n, hipMemcpyHostToDevice, stream()));
    if (cal_index) FBX_HIP(hipMemcpyAsync(di.p, cal_index, sizeof(int32_t) * m, hipMemcpyHostToDevice, stream()));
    FBX_HIP(hipMemcpyAsync(dcm.p, cal_mean, sizeof(double) * n_cal, hipMemcpyHostToDevice, stream()));
    FBX_HIP(hipMemcpyAsync(dcv.p, cal_var, sizeof(double) * n_cal, hipMemcpyHostToDevice, stream()));
    rc = fbx_calibrate_expectations_dev(B, m, de.as<double>(), ds.as<double>(), cal_index ? di.as<int32_t>() : nullptr, n_cal,
                                        dcm.as<double>(), dcv.as<double>(), dm.as<double>(), dr.as<double>());
    if (rc) return rc;
    FBX_HIP(hipMemcpyAsync(mean_out, dm.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipMemcpyAsync(err_out, dr.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

int fbx_shots_to_moments_dev(int n_qubits, int64_t n_settings, int64_t n_shots, const uint8_t* d_bits,
                             const uint8_t* d_obs_mask, const double* d_coefs, int beta_prior,
                             double* d_mean_out, double* d_var_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 64, "fbx_shots_to_moments: n_qubits must be 1..64");
    FBX_REQUIRE(n_settings >= 0 && n_shots >= 1, "fbx_shots_to_moments: need n_settings >= 0 and n_shots >= 1");
    FBX_REQUIRE(n_settings == 0 || (d_bits && d_obs_mask && d_mean_out && d_var_out), "fbx_shots_to_moments: NULL buffer");
    int rc = ensure_device();
    if (rc) return rc;
    if (n_settings == 0) return FBX_OK;
    // short records (below 16 KB per setting): a wavefront per setting, four settings per workgroup
    const bool per_wave = n_shots * n_qubits < 16384 && n_settings >= 4;
    const int64_t units = per_wave ? (n_settings + 3) / 4 : n_settings;
    const unsigned grid = (unsigned)(units < 256 * 16 ? units : 256 * 16);
    // records that are whole 16-byte vectors, at most eight per lane, from a 16-byte aligned base: the pipelined form
    const long long bytes = (long long)n_shots * n_qubits;
    const bool pipe = per_wave && (n_qubits == 1 || n_qubits == 2 || n_qubits == 4 || n_qubits == 8) && (bytes & 15) == 0 &&
                      bytes <= 8192 && (((uintptr_t)d_bits) & 15) == 0;
    if (pipe) {
        const int vpl = (int)((bytes / 16 + 63) / 64);
#define FBX_SHOTS_LAUNCH(NQB, VPL) hipLaunchKernelGGL((shots_pipe_kernel<NQB, VPL>), dim3(grid), dim3(256), 0, stream(), (long long)n_settings, \
                                                      (long long)n_shots, d_bits, d_obs_mask, d_coefs, beta_prior, d_mean_out, d_var_out)
#define FBX_SHOTS_LAUNCH_N(NQB) do { if (vpl <= 1) FBX_SHOTS_LAUNCH(NQB, 1); else if (vpl == 2) FBX_SHOTS_LAUNCH(NQB, 2); \
                                     else if (vpl <= 4) FBX_SHOTS_LAUNCH(NQB, 4); else FBX_SHOTS_LAUNCH(NQB, 8); } while (0)
        if (n_qubits == 1) FBX_SHOTS_LAUNCH_N(1); else if (n_qubits == 2) FBX_SHOTS_LAUNCH_N(2);
        else if (n_qubits == 4) FBX_SHOTS_LAUNCH_N(4); else FBX_SHOTS_LAUNCH_N(8);
#undef FBX_SHOTS_LAUNCH_N
#undef FBX_SHOTS_LAUNCH
    } else if (per_wave && (n_qubits == 3 || n_qubits == 5) && (bytes & 7) == 0 &&      // (6 and 7 qubits: measured no faster than shots_kernel)
               (((uintptr_t)d_bits) & 7) == 0 && n_shots / 16 >= 1 && n_shots / 16 <= 128) {
        const bool one = n_shots / 16 <= 64;
#define FBX_SHOTS_PACKED(NQ) do { if (one) hipLaunchKernelGGL((shots_pipe_packed_kernel<NQ, 1>), dim3(grid), dim3(256), 0, stream(), (long long)n_settings, \
                                                     (long long)n_shots, d_bits, d_obs_mask, d_coefs, beta_prior, d_mean_out, d_var_out); \
                                  else hipLaunchKernelGGL((shots_pipe_packed_kernel<NQ, 2>), dim3(grid), dim3(256), 0, stream(), (long long)n_settings, \
                                                     (long long)n_shots, d_bits, d_obs_mask, d_coefs, beta_prior, d_mean_out, d_var_out); } while (0)
        if (n_qubits == 3) FBX_SHOTS_PACKED(3); else FBX_SHOTS_PACKED(5);
#undef FBX_SHOTS_PACKED
    } else if (per_wave)
        hipLaunchKernelGGL(shots_kernel<true>, dim3(grid), dim3(256), 0, stream(), n_qubits, (long long)n_settings,
                           (long long)n_shots, d_bits, d_obs_mask, d_coefs, beta_prior, d_mean_out, d_var_out);
    else
        hipLaunchKernelGGL(shots_kernel<false>, dim3(grid), dim3(256), 0, stream(), n_qubits, (long long)n_settings,
                           (long long)n_shots, d_bits, d_obs_mask, d_coefs, beta_prior, d_mean_out, d_var_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_shots_to_moments(int n_qubits, int64_t n_settings, int64_t n_shots, const uint8_t* bits,
                         const uint8_t* obs_mask, const double* coefs, int beta_prior,
                         double* mean_out, double* var_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 64, "fbx_shots_to_moments: n_qubits must be 1..64");
    FBX_REQUIRE(n_settings >= 0 && n_shots >= 1, "fbx_shots_to_moments: need n_settings >= 0 and n_shots >= 1");
    FBX_REQUIRE(n_settings == 0 || (bits && obs_mask && mean_out && var_out), "fbx_shots_to_moments: NULL buffer");
    int rc = ensure_device();
    if (rc) return rc;
    if (n_settings == 0) return FBX_OK;
    const size_t nb = (size_t)n_settings * n_shots * n_qubits, nm = (size_t)n_settings * n_qubits;
    DevBuf db, dm, dc, dmean, dvar;
    if ((rc = db.alloc(nb)) || (rc = dm.alloc(nm)) || (rc = dc.alloc(sizeof(double) * n_settings)) ||
        (rc = dmean.alloc(sizeof(double) * n_settings)) || (rc = dvar.alloc(sizeof(double) * n_settings)))
        return rc;
    FBX_HIP(hipMemcpyAsync(db.p, bits, nb, hipMemcpyHostToDevice, stream()));
    FBX_HIP(hipMemcpyAsync(dm.p, obs_mask, nm, hipMemcpyHostToDevice, stream()));
    if (coefs) FBX_HIP(hipMemcpyAsync(dc.p, coefs, sizeof(double) * n_settings, hipMemcpyHostToDevice, stream()));
    rc = fbx_shots_to_moments_dev(n_qubits, n_settings, n_shots, db.as<uint8_t>(), dm.as<uint8_t>(),
                                  coefs ? dc.as<double>() : nullptr, beta_prior, dmean.as<double>(), dvar.as<double>());
    if (rc) return rc;
    FBX_HIP(hipMemcpyAsync(mean_out, dmean.p, sizeof(double) * n_settings, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipMemcpyAsync(var_out, dvar.p, sizeof(double) * n_settings, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

int fbx_dfe_estimate(int n_qubits, int kind, int64_t B, int64_t m, const double* expect, const double* std_err,
                     double* mean_out, double* err_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 30, "fbx_dfe_estimate: n_qubits must be 1..30");
    FBX_REQUIRE(kind == FBX_KIND_STATE || kind == FBX_KIND_PROCESS, "fbx_dfe_estimate: kind must be FBX_KIND_STATE or FBX_KIND_PROCESS");
    FBX_REQUIRE(B >= 0 && m >= 1, "fbx_dfe_estimate: need B >= 0 and m >= 1");
    FBX_REQUIRE(B == 0 || (expect && std_err && mean_out && err_out), "fbx_dfe_estimate: NULL buffer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return FBX_OK;
    const size_t n = (size_t)B * m;
    DevBuf de, ds, dm, dr;
    if ((rc = de.alloc(sizeof(double) * n)) || (rc = ds.alloc(sizeof(double) * n)) ||
        (rc = dm.alloc(sizeof(double) * B)) || (rc = dr.alloc(sizeof(double) * B)))
        return rc;
    FBX_HIP(hipMemcpyAsync(de.p, expect, sizeof(double) * n, hipMemcpyHostToDevice, stream()));
    FBX_HIP(hipMemcpyAsync(ds.p, std_err, sizeof(double) * n, hipMemcpyHostToDevice, stream()));
    const unsigned grid = (unsigned)(B < 65536 ? B : 65536);
    hipLaunchKernelGGL(dfe_kernel, dim3(grid), dim3(64), 0, stream(), n_qubits, kind == FBX_KIND_PROCESS ? 1 : 0,
                       (long long)B, (long long)m, de.as<double>(), ds.as<double>(), dm.as<double>(), dr.as<double>());
    FBX_HIP(hipGetLastError());
    FBX_HIP(hipMemcpyAsync(mean_out, dm.p, sizeof(double) * B, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipMemcpyAsync(err_out, dr.p, sizeof(double) * B, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

int fbx_beta_resample_dev(int64_t n, int64_t R, const double* d_expect, const double* d_counts, double prior_counts,
                          uint64_t seed, double* d_out, double* d_counts_out) {
    FBX_REQUIRE(n >= 0 && R >= 0, "fbx_beta_resample: need n >= 0 and R >= 0");
    FBX_REQUIRE(prior_counts > 0.0, "fbx_beta_resample: prior_counts must be positive");
    FBX_REQUIRE(n * R == 0 || (d_expect && d_counts && d_out), "fbx_beta_resample: NULL buffer");
    int rc = ensure_device();
    if (rc) return rc;
    if (n * R == 0) return FBX_OK;
    const long long total = (long long)n * R, want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < 256 * 32 ? want : 256 * 32);
    hipLaunchKernelGGL(beta_resample_kernel, dim3(grid), dim3(256), 0, stream(), (long long)n, (long long)R, d_expect,
                       d_counts, prior_counts, (unsigned long long)seed, d_out, d_counts_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_beta_resample(int64_t n, int64_t R, const double* expect, const double* counts, double prior_counts,
                      uint64_t seed, double* out) {
    FBX_REQUIRE(n >= 0 && R >= 0, "fbx_beta_resample: need n >= 0 and R >= 0");
    FBX_REQUIRE(prior_counts > 0.0, "fbx_beta_resample: prior_counts must be positive");
    FBX_REQUIRE(n * R == 0 || (expect && counts && out), "fbx_beta_resample: NULL buffer");
    int rc = ensure_device();
    if (rc) return rc;
    if (n * R == 0) return FBX_OK;
    DevBuf de, dc, dout;
    if ((rc = de.alloc(sizeof(double) * n)) || (rc = dc.alloc(sizeof(double) * n)) ||
        (rc = dout.alloc(sizeof(double) * n * R)))
        return rc;
    FBX_HIP(hipMemcpyAsync(de.p, expect, sizeof(double) * n, hipMemcpyHostToDevice, stream()));
    FBX_HIP(hipMemcpyAsync(dc.p, counts, sizeof(double) * n, hipMemcpyHostToDevice, stream()));
    rc = fbx_beta_resample_dev(n, R, de.as<double>(), dc.as<double>(), prior_counts, seed, dout.as<double>(), nullptr);
    if (rc) return rc;
    FBX_HIP(hipMemcpyAsync(out, dout.p, sizeof(double) * n * R, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

}  // extern "C"
