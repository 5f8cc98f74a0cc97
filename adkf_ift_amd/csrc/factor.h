// Workgroup-cooperative, LDS-resident factorisation + inversion of one SPD matrix (n <= NMAX).
//
// Square-root-free Cholesky (A = L D L^T, no pivoting: the pivots are exactly the squares of the
// Cholesky diagonal, so "k-th pivot <= 0" is the reference's NotPSD condition) carried out as ONE
// right-looking sweep over the augmented system [A | I]: after step k the rows below k hold the Schur
// complement (buf0, lower triangle) and the running rows of the unit-lower inverse X = L^-1 (buf1).
// Column k of buf0 and row k of buf1 are read-only during step k, so one barrier per step suffices.
// Then A^-1 = X^T D^-1 X (register-tiled product), log|A| = sum log d_k.
#pragma once
#include "device_utils.h"

namespace adkf {

template <int NMAX>
struct FactorShape {
    static constexpr int LD = NMAX + 1;  // odd leading dimension: column walks are bank-conflict-free
    static constexpr int ELEMS = NMAX * LD;
};

// In: buf0 lower triangle (incl. diagonal) = A, buf1 = anything.  Out: buf1 rows = X scaled by
// 1/sqrt(d_k) (so A^-1 = buf1^T buf1), dinv[k] = 1/d_k, returns info (0 or first bad pivot + 1) and
// log-determinant through `logdet`.  All threads must call; all get the same return values.
template <int NMAX, int NT>
__device__ __forceinline__ int ldl_sweep(float* __restrict__ buf0, float* __restrict__ buf1, float* __restrict__ dinv,
                                         int n, float& logdet, float* red) {
    constexpr int LD = FactorShape<NMAX>::LD;
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;
    constexpr int NY = NT / 32;
    // buf1 = I (lower part)
    for (int e = tid; e < n * n; e += NT) {
        const int i = e / n, j = e - i * n;
        if (j <= i) buf1[i * LD + j] = (i == j) ? 1.f : 0.f;
    }
    int info = 0;
    float ld_acc = 0.f;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        const float p = buf0[k * LD + k];
        if (!(p > 0.f)) {  // also catches NaN
            if (info == 0) info = k + 1;
        }
        const float ip = 1.f / p;
        if (tid == 0) dinv[k] = ip;
        ld_acc += logf(p);
        for (int i = k + 1 + ty; i < n; i += NY) {
            const float f = buf0[i * LD + k] * ip;
            // Schur complement, columns (k, i]
            for (int j = k + 1 + tx; j <= i; j += 32) buf0[i * LD + j] -= f * buf0[j * LD + k];
            // inverse rows, columns [0, k]
            for (int j = tx; j <= k; j += 32) buf1[i * LD + j] -= f * buf1[k * LD + j];
        }
    }
    __syncthreads();
    // scale row k of X by sqrt(1/d_k)
    for (int e = tid; e < n * n; e += NT) {
        const int i = e / n, j = e - i * n;
        if (j <= i) buf1[i * LD + j] *= sqrtf(dinv[i]);
    }
    __syncthreads();
    logdet = ld_acc;
    (void)red;
    return info;
}

// out (full symmetric, LD) = Y^T Y with Y = buf1 lower-triangular [n x n]:  out_ij = sum_{k>=max(i,j)} Y_ki Y_kj
template <int NMAX, int NT>
__device__ __forceinline__ void ata_lower(const float* __restrict__ Y, float* __restrict__ out, int n) {
    constexpr int LD = FactorShape<NMAX>::LD;
    const int nb = (n + 3) >> 2;               // 4x4 register tiles
    const int ntiles = nb * (nb + 1) / 2;
    for (int tl = threadIdx.x; tl < ntiles; tl += NT) {
        // tl -> (bi >= bj)
        int bi = (int)((sqrtf(8.f * tl + 1.f) - 1.f) * 0.5f);
        while ((bi + 1) * (bi + 2) / 2 <= tl) ++bi;
        while (bi * (bi + 1) / 2 > tl) --bi;
        const int bj = tl - bi * (bi + 1) / 2;
        const int i0 = bi * 4, j0 = bj * 4;
        float acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
        for (int k = i0; k < n; ++k) {
            float yi[4], yj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                yi[a] = (i0 + a <= k) ? Y[k * LD + i0 + a] : 0.f;  // Y is lower-triangular: Y_k,c = 0 for c > k
                yj[a] = Y[k * LD + j0 + a];                        // j0 + a <= i0 + 3; entries above the diagonal masked by yi
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += yi[a] * ((j0 + b <= k) ? yj[b] : 0.f);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = i0 + a, j = j0 + b;
                if (i < n && j < n) {
                    out[i * LD + j] = acc[a][b];
                    out[j * LD + i] = acc[a][b];
                }
            }
    }
}

}  // namespace adkf
