// libadkf_gp_cpu.so - the CPU twin of the GP entry points of include/adkf_gp.h ("CPU twins of each for
// testing without a GPU"; section 8(d)(ii): the second CPU baseline, so that GPU-vs-CPU is not only GPU vs slow Python).
//
// Same C ABI, HOST pointers: adkf_batch_t, priors, phi layouts, info codes and flags exactly as in the header; `ws` /
// `ws_bytes` / `stream` are accepted and ignored (adkf_workspace_bytes returns 0 here).  Non-ARD batches of any size.
// Plain loops, OpenMP over the tasks of a batch, float64 arithmetic inside and float32 at the boundary - a Cholesky-based
// restatement of the same staged closed-form algebra the HIP kernels run (stage names as in DESIGN.md section 3 and
// oracle/closed_form.py): kernel matrices from difference-form squared distances, A = L L^T, A^-1, the analytic 3 x 3 Hessian,
// the joint predictive NLL with its cotangents Omega, M_A, M_B, v = H^-1 grad f_out, the mixed-partial weights and the chain
// through D^2 to dL/dZ.  The inner fit is the same quasi-Newton state machine as csrc/inner.h (BFGS, Armijo backtracking
// with safeguarded quadratic interpolation, the reference's gtol / ftol rules).
//
// Used ONLY by bench.py's cpu_baseline leg (entry "kind": "twin") and by tests/test_cpu_twin.py, which pins it to the golden
// fixtures; the product package (adkf_ift_amd.gp_ops) never loads it and keeps refusing CPU tensors.
//   g++ -O3 -fopenmp -shared -fPIC -std=c++17 -I include adkf_ift_amd/csrc/cpu/adkf_gp_cpu.cpp -o adkf_ift_amd/libadkf_gp_cpu.so
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "adkf_gp.h"

namespace {

using Mat = std::vector<double>;   // row-major
constexpr double NOISE_LB = 1e-4, LOG_2PI = 1.8378770664093453, SQRT5 = 2.23606797749979;

inline double softplus(double x) { return x > 30.0 ? x : std::log1p(std::exp(x)); }
inline double sigmoid(double x) { return 1.0 / (1.0 + std::exp(-x)); }
inline double inv_softplus(double y) { return y > 30.0 ? y : y + std::log(-std::expm1(-y)); }

inline void kappa(int kind, double u, double& k0, double& k1, double& k2) {
    if (kind == ADKF_KERNEL_RBF) { k0 = std::exp(-0.5 * u); k1 = -0.5 * k0; k2 = 0.25 * k0; return; }
    const double r = std::sqrt(u), e = std::exp(-SQRT5 * r);
    k0 = (1.0 + SQRT5 * r + (5.0 / 3.0) * u) * e; k1 = -(5.0 / 6.0) * (1.0 + SQRT5 * r) * e; k2 = (25.0 / 12.0) * e;
}

// D2[i][j] = |x_i - y_j|^2 (difference form: exact zero on the diagonal of a matrix with itself)
Mat sqdist(const float* X, int n, const float* Y, int m, int d) {
    Mat D((size_t)n * m);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            const float *x = X + (size_t)i * d, *y = Y + (size_t)j * d;
            for (int k = 0; k < d; ++k) { const double t = (double)x[k] - (double)y[k]; s += t * t; }
            D[(size_t)i * m + j] = s;
        }
    return D;
}

// in-place lower Cholesky; returns 0 or (index + 1) of the first non-positive pivot; logdet = log|A|
int cholesky(Mat& A, int n, double& logdet) {
    logdet = 0.0;
    for (int j = 0; j < n; ++j) {
        double p = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) p -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(p > 0.0)) return j + 1;
        const double l = std::sqrt(p);
        A[(size_t)j * n + j] = l;
        logdet += 2.0 * std::log(l);
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / l;
        }
    }
    return 0;
}

// A^-1 from its Cholesky factor L (lower, in `L`): X = L^-T L^-1
Mat inverse_from_chol(const Mat& L, int n) {
    Mat Li((size_t)n * n, 0.0);   // L^-1, lower
    for (int j = 0; j < n; ++j) {
        Li[(size_t)j * n + j] = 1.0 / L[(size_t)j * n + j];
        for (int i = j + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s -= L[(size_t)i * n + k] * Li[(size_t)k * n + j];
            Li[(size_t)i * n + j] = s / L[(size_t)i * n + i];
        }
    }
    Mat X((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int k = i; k < n; ++k) s += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
            X[(size_t)i * n + j] = s; X[(size_t)j * n + i] = s;
        }
    return X;
}

Mat matmul(const Mat& A, int n, int k, const Mat& B, int m) {   // [n,k] x [k,m]
    Mat C((size_t)n * m, 0.0);
    for (int i = 0; i < n; ++i)
        for (int p = 0; p < k; ++p) {
            const double a = A[(size_t)i * k + p];
            if (a == 0.0) continue;
            const double* b = &B[(size_t)p * m];
            double* c = &C[(size_t)i * m];
            for (int j = 0; j < m; ++j) c[j] += a * b[j];
        }
    return C;
}
Mat matmul_nt(const Mat& A, int n, int k, const Mat& B, int m) {   // [n,k] x [m,k]^T
    Mat C((size_t)n * m);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            const double *a = &A[(size_t)i * k], *b = &B[(size_t)j * k];
            for (int p = 0; p < k; ++p) s += a[p] * b[p];
            C[(size_t)i * m + j] = s;
        }
    return C;
}
std::vector<double> matvec(const Mat& A, int n, int m, const std::vector<double>& x) {
    std::vector<double> y(n, 0.0);
    for (int i = 0; i < n; ++i) { double s = 0.0; for (int j = 0; j < m; ++j) s += A[(size_t)i * m + j] * x[j]; y[i] = s; }
    return y;
}
std::vector<double> matvec_t(const Mat& A, int n, int m, const std::vector<double>& x) {   // A^T x
    std::vector<double> y(m, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) y[j] += A[(size_t)i * m + j] * x[i];
    return y;
}
inline double dot(const std::vector<double>& a, const std::vector<double>& b) { double s = 0.0; for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i]; return s; }
inline double frob(const Mat& A, const Mat& B) { double s = 0.0; for (size_t i = 0; i < A.size(); ++i) s += A[i] * B[i]; return s; }

void lognormal_terms(double x, double loc, double scale, double& lp, double& d1, double& d2) {
    const double lx = std::log(x), z = (lx - loc) / (scale * scale);
    lp = -lx - std::log(scale) - 0.5 * LOG_2PI - 0.5 * (lx - loc) * z;
    d1 = (-1.0 - z) / x;
    d2 = (1.0 + z - 1.0 / (scale * scale)) / (x * x);
}

struct Inner {
    int n = 0, info = 0;
    double noise, s, l, d1[3], d2[3], f_in, g_in[3], H[9], logdet;
    Mat Ainv, K, G, u, k1, k2, P;
    std::vector<double> alpha, gamma, beta, delta;
};

// oracle/closed_form.py::inner_stage
Inner inner_stage(const Mat& D2, const float* y_, int n, const double* phi, const float* pri, int kind, bool want_hess, bool want_mats) {
    Inner o; o.n = n;
    for (int q = 0; q < 3; ++q) { const double sg = sigmoid(phi[q]); o.d1[q] = sg; o.d2[q] = sg * (1.0 - sg); }
    o.noise = softplus(phi[0]) + NOISE_LB; o.s = softplus(phi[1]); o.l = softplus(phi[2]);
    const double il2 = 1.0 / (o.l * o.l);
    std::vector<double> y(n);
    for (int i = 0; i < n; ++i) y[i] = y_[i];
    o.u.resize((size_t)n * n); o.k1.resize((size_t)n * n); o.k2.resize((size_t)n * n); o.K.resize((size_t)n * n); o.G.resize((size_t)n * n);
    Mat A((size_t)n * n);
    for (size_t e = 0; e < (size_t)n * n; ++e) {
        const double u = D2[e] * il2;
        double k0, k1, k2; kappa(kind, u, k0, k1, k2);
        o.u[e] = u; o.k1[e] = k1; o.k2[e] = k2; o.K[e] = o.s * k0; o.G[e] = o.s * k1 * (-2.0 * u / o.l);
        A[e] = o.K[e];
    }
    for (int i = 0; i < n; ++i) A[(size_t)i * n + i] += o.noise;
    o.info = cholesky(A, n, o.logdet);
    if (o.info) { o.f_in = std::numeric_limits<double>::infinity(); for (double& g : o.g_in) g = 0.0; return o; }
    o.Ainv = inverse_from_chol(A, n);
    o.alpha = matvec(o.Ainv, n, n, y);
    const double ya = dot(y, o.alpha), aa = dot(o.alpha, o.alpha);
    const double nll = 0.5 * ya + 0.5 * o.logdet + 0.5 * n * LOG_2PI;
    double lpn = 0, dpn = 0, d2pn = 0, lpl = 0, dpl = 0, d2pl = 0;
    if (pri[1] > 0.f) lognormal_terms(o.noise, pri[0], pri[1], lpn, dpn, d2pn);
    if (pri[3] > 0.f) lognormal_terms(o.l, pri[2], pri[3], lpl, dpl, d2pl);
    o.f_in = (nll - lpn - lpl) / n;
    double trAinv = 0.0;
    for (int i = 0; i < n; ++i) trAinv += o.Ainv[(size_t)i * n + i];
    const double trAinvG = frob(o.Ainv, o.G);
    o.beta = matvec(o.G, n, n, o.alpha);
    const double aGa = dot(o.alpha, o.beta);
    const double g_noise = 0.5 * trAinv - 0.5 * aa;
    const double g_s = (0.5 * (n - o.noise * trAinv) - 0.5 * (ya - o.noise * aa)) / o.s;
    const double g_l = 0.5 * trAinvG - 0.5 * aGa;
    o.g_in[0] = (g_noise - dpn) * o.d1[0] / n; o.g_in[1] = g_s * o.d1[1] / n; o.g_in[2] = (g_l - dpl) * o.d1[2] / n;
    if (!want_hess) { if (!want_mats) { o.u.clear(); o.k2.clear(); } return o; }
    o.P = matmul(o.Ainv, n, n, o.G, n);
    o.gamma = matvec(o.Ainv, n, n, o.alpha);
    o.delta = matvec(o.Ainv, n, n, o.beta);
    double trA2 = frob(o.Ainv, o.Ainv), trPA = frob(o.P, o.Ainv), trPP = 0.0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) trPP += o.P[(size_t)i * n + j] * o.P[(size_t)j * n + i];
    const double ag = dot(o.alpha, o.gamma), bg = dot(o.beta, o.gamma), bd = dot(o.beta, o.delta), ab = dot(o.alpha, o.beta);
    double trAinvKll = 0.0, aKlla = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const size_t e = (size_t)i * n + j;
            const double kll = o.s * (o.k2[e] * 4.0 * o.u[e] * o.u[e] * il2 + o.k1[e] * 6.0 * o.u[e] * il2);
            trAinvKll += o.Ainv[e] * kll; aKlla += o.alpha[i] * kll * o.alpha[j];
        }
    double h[3][3];
    const double noise = o.noise, s = o.s;
    h[0][0] = ag - 0.5 * trA2 - d2pn;
    h[0][1] = ((aa - noise * ag) - 0.5 * (trAinv - noise * trA2)) / s;
    h[0][2] = bg - 0.5 * trPA;
    h[1][1] = ((ya - 2 * noise * aa + noise * noise * ag) - 0.5 * (n - 2 * noise * trAinv + noise * noise * trA2)) / (s * s);
    h[1][2] = ((ab - noise * bg) - 0.5 * (trAinvG - noise * trPA)) / s - (0.5 * aGa - 0.5 * trAinvG) / s;
    h[2][2] = bd - 0.5 * aKlla - 0.5 * trPP + 0.5 * trAinvKll - d2pl;
    h[1][0] = h[0][1]; h[2][0] = h[0][2]; h[2][1] = h[1][2];
    const double gt[3] = {g_noise - dpn, g_s, g_l - dpl};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.H[i * 3 + j] = (h[i][j] * o.d1[i] * o.d1[j] + (i == j ? gt[i] * o.d2[i] : 0.0)) / n;
    return o;
}

struct Outer {
    int info = 0;
    double f_out, g_out[3];
    Mat W_ss, W_qs, W_qq, S, C;
    std::vector<double> mean;
};

// oracle/closed_form.py::outer_stage  (level 1: mean / covariance only)
Outer outer_stage(const Mat& D2qs, const Mat& D2qq, const float* yq_, int m, const Inner& in, int kind, bool want_grads) {
    Outer o;
    const int n = in.n;
    const double il2 = 1.0 / (in.l * in.l), s = in.s, l = in.l;
    Mat B((size_t)m * n), kqs1((size_t)m * n), uqs((size_t)m * n);
    for (size_t e = 0; e < (size_t)m * n; ++e) { double k0, k1, k2; uqs[e] = D2qs[e] * il2; kappa(kind, uqs[e], k0, k1, k2); B[e] = s * k0; kqs1[e] = k1; }
    o.C = matmul(B, m, n, in.Ainv, n);
    o.mean = matvec(B, m, n, in.alpha);
    Mat CBt = matmul_nt(o.C, m, n, B, m);
    Mat Kqq((size_t)m * m), kqq1((size_t)m * m), uqq((size_t)m * m);
    o.S.resize((size_t)m * m);
    for (size_t e = 0; e < (size_t)m * m; ++e) { double k0, k1, k2; uqq[e] = D2qq[e] * il2; kappa(kind, uqq[e], k0, k1, k2); Kqq[e] = s * k0; kqq1[e] = k1; o.S[e] = Kqq[e] - CBt[e]; }
    for (int i = 0; i < m; ++i) o.S[(size_t)i * m + i] += in.noise;
    for (int i = 0; i < m; ++i) for (int j = 0; j < i; ++j) { const double a = 0.5 * (o.S[(size_t)i * m + j] + o.S[(size_t)j * m + i]); o.S[(size_t)i * m + j] = a; o.S[(size_t)j * m + i] = a; }
    if (!yq_) return o;
    Mat L = o.S;
    double logdetS;
    const int bad = cholesky(L, m, logdetS);
    if (bad) { o.info = ADKF_INFO_OUTER_BASE + bad; o.f_out = std::numeric_limits<double>::infinity(); return o; }
    Mat Sinv = inverse_from_chol(L, m);
    std::vector<double> r(m);
    for (int i = 0; i < m; ++i) r[i] = (double)yq_[i] - o.mean[i];
    std::vector<double> e = matvec(Sinv, m, m, r);
    o.f_out = 0.5 * dot(r, e) + 0.5 * logdetS + 0.5 * m * LOG_2PI;
    if (!want_grads) return o;
    Mat Om((size_t)m * m);
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Om[(size_t)i * m + j] = 0.5 * (Sinv[(size_t)i * m + j] - e[i] * e[j]);
    Mat OC = matmul(Om, m, m, o.C, n);
    Mat M_B((size_t)m * n);
    for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) M_B[(size_t)i * n + j] = -2.0 * OC[(size_t)i * n + j] - e[i] * in.alpha[j];
    std::vector<double> Cte = matvec_t(o.C, m, n, e);
    Mat M_A((size_t)n * n, 0.0);
    for (int p = 0; p < m; ++p)
        for (int i = 0; i < n; ++i) {
            const double c = o.C[(size_t)p * n + i];
            for (int j = 0; j < n; ++j) M_A[(size_t)i * n + j] += c * OC[(size_t)p * n + j];
        }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M_A[(size_t)i * n + j] += 0.5 * (Cte[i] * in.alpha[j] + in.alpha[i] * Cte[j]);
    double trOm = 0.0, trMA = 0.0;
    for (int i = 0; i < m; ++i) trOm += Om[(size_t)i * m + i];
    for (int i = 0; i < n; ++i) trMA += M_A[(size_t)i * n + i];
    double g_s = frob(M_A, in.K) + frob(M_B, B) + frob(Om, Kqq), g_l = frob(M_A, in.G);
    for (size_t q = 0; q < (size_t)m * n; ++q) g_l += M_B[q] * (s * kqs1[q] * (-2.0 * uqs[q] / l));
    for (size_t q = 0; q < (size_t)m * m; ++q) g_l += Om[q] * (s * kqq1[q] * (-2.0 * uqq[q] / l));
    o.g_out[0] = (trOm + trMA) * in.d1[0]; o.g_out[1] = g_s / s * in.d1[1]; o.g_out[2] = g_l * in.d1[2];
    o.W_ss.resize((size_t)n * n); o.W_qs.resize((size_t)m * n); o.W_qq.resize((size_t)m * m);
    for (size_t q = 0; q < (size_t)n * n; ++q) o.W_ss[q] = M_A[q] * s * in.k1[q] * il2;
    for (size_t q = 0; q < (size_t)m * n; ++q) o.W_qs[q] = M_B[q] * s * kqs1[q] * il2;
    for (size_t q = 0; q < (size_t)m * m; ++q) o.W_qq[q] = Om[q] * s * kqq1[q] * il2;
    return o;
}

// oracle/closed_form.py::mixed_stage
Mat mixed_stage(const double* v, const Inner& in) {
    const int n = in.n;
    const double noise = in.noise, s = in.s, l = in.l, il2 = 1.0 / (l * l);
    const double cn = v[0] * in.d1[0], cs = v[1] * in.d1[1] / s, cl = v[2] * in.d1[2];
    Mat X((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        const size_t e = (size_t)i * n + j;
        X[e] = cn * in.Ainv[e] + cs * ((i == j ? 1.0 : 0.0) - noise * in.Ainv[e]) + cl * in.P[e];
    }
    std::vector<double> w(n);
    for (int i = 0; i < n; ++i) w[i] = cn * in.gamma[i] + cs * (in.alpha[i] - noise * in.gamma[i]) + cl * in.delta[i];
    Mat XA = matmul(X, n, n, in.Ainv, n);
    Mat W((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        const size_t e = (size_t)i * n + j;
        const double dg_dA = (-0.5 * XA[e] + 0.5 * (w[i] * in.alpha[j] + in.alpha[i] * w[j])) / n;
        const double Q = 0.5 * (in.Ainv[e] - in.alpha[i] * in.alpha[j]) / n;
        const double dBv_du = cs * s * in.k1[e] + cl * s * (-2.0 / l) * (in.k1[e] + in.u[e] * in.k2[e]);
        W[e] = dg_dA * s * in.k1[e] * il2 + Q * dBv_du * il2;
    }
    return W;
}

// oracle/closed_form.py::dz_from_weights: D2_ij = |z_i - z_j|^2
void dz_from_weights(const float* Zs, int n, const float* Zq, int m, int d, const Mat* W_ss, const Mat* W_qs, const Mat* W_qq,
                     float* dZs, float* dZq, int ld_d) {
    (void)ld_d;
    if (dZs) {
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < d; ++k) {
                double s = 0.0;
                if (W_ss) for (int j = 0; j < n; ++j) s += 2.0 * ((*W_ss)[(size_t)i * n + j] + (*W_ss)[(size_t)j * n + i]) * ((double)Zs[(size_t)i * d + k] - (double)Zs[(size_t)j * d + k]);
                if (W_qs) for (int p = 0; p < m; ++p) s += 2.0 * (*W_qs)[(size_t)p * n + i] * ((double)Zs[(size_t)i * d + k] - (double)Zq[(size_t)p * d + k]);
                dZs[(size_t)i * d + k] = (float)s;
            }
    }
    if (dZq) {
        for (int p = 0; p < m; ++p)
            for (int k = 0; k < d; ++k) {
                double s = 0.0;
                if (W_qs) for (int j = 0; j < n; ++j) s += 2.0 * (*W_qs)[(size_t)p * n + j] * ((double)Zq[(size_t)p * d + k] - (double)Zs[(size_t)j * d + k]);
                if (W_qq) for (int q = 0; q < m; ++q) s += 2.0 * ((*W_qq)[(size_t)p * m + q] + (*W_qq)[(size_t)q * m + p]) * ((double)Zq[(size_t)p * d + k] - (double)Zq[(size_t)q * d + k]);
                dZq[(size_t)p * d + k] = (float)s;
            }
    }
}

bool solve3(const double* H, const double* b, double* x) {   // partial pivoting
    double a[3][4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) a[i][j] = H[i * 3 + j]; a[i][3] = b[i]; }
    for (int c = 0; c < 3; ++c) {
        int p = c;
        for (int r = c + 1; r < 3; ++r) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0) return false;
        if (p != c) for (int j = 0; j < 4; ++j) std::swap(a[p][j], a[c][j]);
        for (int r = c + 1; r < 3; ++r) { const double f = a[r][c] / a[c][c]; for (int j = c; j < 4; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int r = 2; r >= 0; --r) { double s = a[r][3]; for (int j = r + 1; j < 3; ++j) s -= a[r][j] * x[j]; x[r] = s / a[r][r]; }
    return true;
}

double median_l0(const Mat& D2, int n) {
    std::vector<double> v;
    v.reserve((size_t)n * (n - 1) / 2);
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { const double x = (double)(float)D2[(size_t)i * n + j]; if (x > 0.0) v.push_back(x); }
    if (v.empty()) return 0.0;
    const size_t k = (v.size() - 1) / 2;
    std::nth_element(v.begin(), v.begin() + k, v.end());
    return std::sqrt(0.5 * v[k]);
}

int check(const adkf_batch_t* b, bool need_q) {
    if (!b || b->T <= 0 || b->ns_max <= 0 || b->nq_max < 0 || b->d <= 0 || !b->Z_s) return ADKF_E_BADARG;
    if (b->kernel != ADKF_KERNEL_RBF && b->kernel != ADKF_KERNEL_MATERN52) return ADKF_E_BADARG;
    if (b->flags & ADKF_BATCH_ARD) return ADKF_E_BADARG;
    if (need_q && (b->nq_max <= 0 || !b->Z_q)) return ADKF_E_BADARG;
    return 0;
}
inline int ns_of(const adkf_batch_t* b, int t) { return b->n_s ? b->n_s[t] : b->ns_max; }
inline int nq_of(const adkf_batch_t* b, int t) { return b->n_q ? b->n_q[t] : b->nq_max; }

// The quasi-Newton driver of csrc/inner.h (Bfgs + fit_advance), in float64.
struct Bfgs {
    double x[3], f, g[3], Hi[3][3], p[3], gp, step; int bt; bool first;
    void reset_H() { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Hi[i][j] = i == j; first = true; }
    bool direction() {
        for (int i = 0; i < 3; ++i) p[i] = -(Hi[i][0] * g[0] + Hi[i][1] * g[1] + Hi[i][2] * g[2]);
        gp = g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
        if (!(gp < 0.0)) { reset_H(); for (int i = 0; i < 3; ++i) p[i] = -g[i]; gp = -(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]); if (!(gp < 0.0)) return false; }
        step = first ? std::min(1.0, 1.0 / (std::fabs(g[0]) + std::fabs(g[1]) + std::fabs(g[2]))) : 1.0;
        step = std::min(step, 16.0 / std::max(std::max(std::fabs(p[0]), std::fabs(p[1])), std::max(std::fabs(p[2]), 1e-30)));
        bt = 0;
        return true;
    }
    void accept(const double* xn, double fn, const double* gn) {
        double s[3], yv[3];
        for (int i = 0; i < 3; ++i) { s[i] = xn[i] - x[i]; yv[i] = gn[i] - g[i]; x[i] = xn[i]; g[i] = gn[i]; }
        f = fn;
        const double sy = s[0] * yv[0] + s[1] * yv[1] + s[2] * yv[2], yy = yv[0] * yv[0] + yv[1] * yv[1] + yv[2] * yv[2], ss = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
        if (sy > 1e-10 * std::sqrt(ss * yy) && yy > 0.0) {
            if (first) { const double sc = sy / yy; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Hi[i][j] = i == j ? sc : 0.0; first = false; }
            const double rho = 1.0 / sy;
            double Hy[3];
            for (int i = 0; i < 3; ++i) Hy[i] = Hi[i][0] * yv[0] + Hi[i][1] * yv[1] + Hi[i][2] * yv[2];
            const double yHy = yv[0] * Hy[0] + yv[1] * Hy[1] + yv[2] * Hy[2];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Hi[i][j] += -rho * (s[i] * Hy[j] + Hy[i] * s[j]) + rho * (rho * yHy + 1.0) * s[i] * s[j];
        }
    }
};

}  // namespace

extern "C" {

const char* adkf_version(void) { return "adkf_gp_cpu 0.1 (float64 inside, OpenMP over tasks)"; }
const char* adkf_last_hip_error(void) { return "no HIP in the CPU twin"; }
int adkf_max_points(void) { return 1 << 20; }
size_t adkf_workspace_bytes(int32_t, int32_t, int32_t, int32_t) { return 0; }

int adkf_median_lengthscale(const adkf_batch_t* b, float* l0, void*, size_t, void*) {
    if (int rc = check(b, false)) return rc;
    if (!l0) return ADKF_E_BADARG;
#pragma omp parallel for schedule(dynamic)
    for (int t = 0; t < b->T; ++t) {
        const int n = ns_of(b, t);
        const float* Z = b->Z_s + (size_t)t * b->ns_max * b->d;
        l0[t] = n > 1 ? (float)median_l0(sqdist(Z, n, Z, n, b->d), n) : 0.f;
    }
    return 0;
}

int adkf_init_params(const adkf_batch_t* b, int32_t numeric, int32_t use_ls_prior, float* phi, float* priors, float* l0, void*, size_t, void*) {
    if (int rc = check(b, false)) return rc;
    if (!phi || !priors) return ADKF_E_BADARG;
#pragma omp parallel for schedule(dynamic)
    for (int t = 0; t < b->T; ++t) {
        const int n = ns_of(b, t);
        const float* Z = b->Z_s + (size_t)t * b->ns_max * b->d;
        const float l = n > 1 ? (float)median_l0(sqdist(Z, n, Z, n, b->d), n) : 0.f;
        if (l0) l0[t] = l;
        const double scale = 0.25, mode = numeric ? 0.01 : 0.1;
        phi[t * 3 + 0] = (float)inv_softplus(mode - NOISE_LB); phi[t * 3 + 1] = 0.f; phi[t * 3 + 2] = (float)inv_softplus((double)l);
        priors[t * 4 + 0] = (float)(std::log(mode) + scale * scale); priors[t * 4 + 1] = (float)scale;
        priors[t * 4 + 2] = use_ls_prior ? (float)(std::log((double)l) + scale * scale) : 0.f; priors[t * 4 + 3] = use_ls_prior ? (float)scale : -1.f;
    }
    return 0;
}

int adkf_mll_value_grad(const adkf_batch_t* b, const float* phi, float* f_in, float* g_phi, float* dZ_s, int32_t* info, void*, size_t, void*) {
    if (int rc = check(b, false)) return rc;
    if (!phi || !f_in || !info || !b->y_s || !b->priors) return ADKF_E_BADARG;
    const int d = b->d;
#pragma omp parallel for schedule(dynamic)
    for (int t = 0; t < b->T; ++t) {
        const int n = ns_of(b, t);
        const float* Z = b->Z_s + (size_t)t * b->ns_max * d;
        const double p[3] = {phi[t * 3], phi[t * 3 + 1], phi[t * 3 + 2]};
        Inner in = inner_stage(sqdist(Z, n, Z, n, d), b->y_s + (size_t)t * b->ns_max, n, p, b->priors + t * 4, b->kernel, false, true);
        info[t] = in.info; f_in[t] = (float)in.f_in;
        if (g_phi) for (int q = 0; q < 3; ++q) g_phi[t * 3 + q] = (float)in.g_in[q];
        if (dZ_s) {
            float* out = dZ_s + (size_t)t * b->ns_max * d;
            std::memset(out, 0, sizeof(float) * (size_t)b->ns_max * d);
            if (!in.info) {
                Mat W((size_t)n * n);
                const double il2 = 1.0 / (in.l * in.l);
                for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { const size_t e = (size_t)i * n + j; W[e] = 0.5 * (in.Ainv[e] - in.alpha[i] * in.alpha[j]) / n * in.s * in.k1[e] * il2; }
                dz_from_weights(Z, n, nullptr, 0, d, &W, nullptr, nullptr, out, nullptr, d);
            }
        }
    }
    return 0;
}

int adkf_fit(const adkf_batch_t* b, float* phi, const adkf_fit_options_t* opt, float* f_final, float* gnorm, int32_t* n_evals, int32_t* info,
             void*, size_t, void*) {
    if (int rc = check(b, false)) return rc;
    if (!phi || !opt || !info || !b->y_s || !b->priors || opt->max_evals < 1) return ADKF_E_BADARG;
    const int d = b->d;
#pragma omp parallel for schedule(dynamic)
    for (int t = 0; t < b->T; ++t) {
        const int n = ns_of(b, t);
        const float* Z = b->Z_s + (size_t)t * b->ns_max * d;
        const Mat D2 = sqdist(Z, n, Z, n, d);
        const float* y = b->y_s + (size_t)t * b->ns_max;
        const float* pri = b->priors + t * 4;
        auto eval = [&](const double* x, double& f, double* g) { Inner in = inner_stage(D2, y, n, x, pri, b->kernel, false, false); f = in.f_in; for (int q = 0; q < 3; ++q) g[q] = in.g_in[q]; return in.info; };
        Bfgs st;
        for (int q = 0; q < 3; ++q) st.x[q] = phi[t * 3 + q];
        st.reset_H();
        double xe[3] = {st.x[0], st.x[1], st.x[2]}, fe, ge[3];
        int evals = 0, ie = eval(xe, fe, ge);
        ++evals;
        st.f = fe; for (int q = 0; q < 3; ++q) st.g[q] = ge[q];
        const int budget = opt->max_evals - 1;
        bool stop = ie != 0 || std::max(std::fabs(ge[0]), std::max(std::fabs(ge[1]), std::fabs(ge[2]))) <= opt->gtol || !st.direction();
        while (!stop && evals < budget) {
            for (int q = 0; q < 3; ++q) xe[q] = st.x[q] + st.step * st.p[q];
            ie = eval(xe, fe, ge);
            ++evals;
            if (ie == 0 && fe <= st.f + 1e-4 * st.step * st.gp) {
                const double fprev = st.f;
                st.accept(xe, fe, ge);
                const double gmax = std::max(std::fabs(ge[0]), std::max(std::fabs(ge[1]), std::fabs(ge[2])));
                if (gmax <= opt->gtol || std::fabs(fprev - fe) <= opt->ftol * std::max(std::max(std::fabs(fprev), std::fabs(fe)), 1.0)) stop = true;
                else if (!st.direction()) stop = true;
            } else {
                const double denom = 2.0 * (fe - st.f - st.gp * st.step);
                const double sq = (denom > 0.0 && std::isfinite(fe)) ? (-st.gp * st.step * st.step / denom) : 0.5 * st.step;
                st.step = std::min(std::max(sq, 0.1 * st.step), 0.5 * st.step);
                if (++st.bt >= 12) stop = true;
            }
        }
        double ff, gf[3];
        const int inf = eval(st.x, ff, gf);     // the output evaluation at the accepted point
        ++evals;
        for (int q = 0; q < 3; ++q) phi[t * 3 + q] = (float)st.x[q];
        info[t] = inf;
        if (f_final) f_final[t] = (float)ff;
        if (gnorm) gnorm[t] = (float)std::max(std::fabs(gf[0]), std::max(std::fabs(gf[1]), std::fabs(gf[2])));
        if (n_evals) n_evals[t] = opt->exact_evals ? opt->max_evals : evals;
    }
    return 0;
}

int adkf_predict(const adkf_batch_t* b, const float* phi, float* mean, float* var, float* cov, int32_t* info, void*, size_t, void*) {
    if (int rc = check(b, true)) return rc;
    if (!phi || !mean || !info || !b->y_s || !b->priors) return ADKF_E_BADARG;
    const int d = b->d;
#pragma omp parallel for schedule(dynamic)
    for (int t = 0; t < b->T; ++t) {
        const int n = ns_of(b, t), m = nq_of(b, t);
        const float *Zs = b->Z_s + (size_t)t * b->ns_max * d, *Zq = b->Z_q + (size_t)t * b->nq_max * d;
        const double p[3] = {phi[t * 3], phi[t * 3 + 1], phi[t * 3 + 2]};
        Inner in = inner_stage(sqdist(Zs, n, Zs, n, d), b->y_s + (size_t)t * b->ns_max, n, p, b->priors + t * 4, b->kernel, false, false);
        info[t] = in.info;
        float* mu = mean + (size_t)t * b->nq_max;
        std::fill(mu, mu + b->nq_max, 0.f);
        if (var) std::fill(var + (size_t)t * b->nq_max, var + (size_t)(t + 1) * b->nq_max, 0.f);
        if (cov) std::fill(cov + (size_t)t * b->nq_max * b->nq_max, cov + (size_t)(t + 1) * b->nq_max * b->nq_max, 0.f);
        if (in.info) continue;
        Outer o = outer_stage(sqdist(Zq, m, Zs, n, d), sqdist(Zq, m, Zq, m, d), nullptr, m, in, b->kernel, false);
        for (int i = 0; i < m; ++i) {
            mu[i] = (float)o.mean[i];
            if (var) var[(size_t)t * b->nq_max + i] = (float)o.S[(size_t)i * m + i];
            if (cov) for (int j = 0; j < m; ++j) cov[((size_t)t * b->nq_max + i) * b->nq_max + j] = (float)o.S[(size_t)i * m + j];
        }
    }
    return 0;
}

static int outer_common(const adkf_batch_t* b, const float* phi, int flags, bool with_hessian, float* f_out, float* g_phi, float* dZ_s, float* dZ_q,
                        float* v_out, float* H_out, int32_t* info) {
    const int d = b->d;
    const bool direct = !(flags & ADKF_IGNORE_DIRECT_GRAD), corr = with_hessian && !(flags & ADKF_IGNORE_GRAD_CORRECTION);
#pragma omp parallel for schedule(dynamic)
    for (int t = 0; t < b->T; ++t) {
        const int n = ns_of(b, t), m = nq_of(b, t);
        const float *Zs = b->Z_s + (size_t)t * b->ns_max * d, *Zq = b->Z_q + (size_t)t * b->nq_max * d;
        const double p[3] = {phi[t * 3], phi[t * 3 + 1], phi[t * 3 + 2]};
        Inner in = inner_stage(sqdist(Zs, n, Zs, n, d), b->y_s + (size_t)t * b->ns_max, n, p, b->priors + t * 4, b->kernel, with_hessian, true);
        info[t] = in.info;
        if (dZ_s) std::memset(dZ_s + (size_t)t * b->ns_max * d, 0, sizeof(float) * (size_t)b->ns_max * d);
        if (dZ_q) std::memset(dZ_q + (size_t)t * b->nq_max * d, 0, sizeof(float) * (size_t)b->nq_max * d);
        if (in.info) { f_out[t] = std::numeric_limits<float>::infinity(); continue; }
        Outer o = outer_stage(sqdist(Zq, m, Zs, n, d), sqdist(Zq, m, Zq, m, d), b->y_q + (size_t)t * b->nq_max, m, in, b->kernel, true);
        f_out[t] = (float)o.f_out;
        if (o.info) { info[t] = o.info; continue; }
        if (g_phi) for (int q = 0; q < 3; ++q) g_phi[t * 3 + q] = (float)o.g_out[q];
        double v[3] = {0, 0, 0};
        if (with_hessian) {
            solve3(in.H, o.g_out, v);
            if (v_out) for (int q = 0; q < 3; ++q) v_out[t * 3 + q] = (float)v[q];
            if (H_out) for (int q = 0; q < 9; ++q) H_out[t * 9 + q] = (float)in.H[q];
        }
        Mat Wss((size_t)n * n, 0.0);
        if (direct) Wss = o.W_ss;
        if (corr) { Mat Wm = mixed_stage(v, in); for (size_t q = 0; q < Wss.size(); ++q) Wss[q] -= Wm[q]; }
        dz_from_weights(Zs, n, Zq, m, d, &Wss, direct ? &o.W_qs : nullptr, direct ? &o.W_qq : nullptr,
                        dZ_s ? dZ_s + (size_t)t * b->ns_max * d : nullptr, dZ_q ? dZ_q + (size_t)t * b->nq_max * d : nullptr, d);
    }
    return 0;
}

int adkf_outer_nll_value_grad(const adkf_batch_t* b, const float* phi, float* f_out, float* g_phi, float* dZ_s, float* dZ_q, int32_t* info,
                              void*, size_t, void*) {
    if (int rc = check(b, true)) return rc;
    if (!phi || !f_out || !info || !b->y_s || !b->y_q || !b->priors) return ADKF_E_BADARG;
    return outer_common(b, phi, 0, false, f_out, g_phi, dZ_s, dZ_q, nullptr, nullptr, info);
}

int adkf_ift_hypergrad(const adkf_batch_t* b, const float* phi, int32_t flags, float* f_out, float* dZ_s, float* dZ_q, float* g_phi_out, float* v,
                       float* H, int32_t* info, void*, size_t, void*) {
    if (int rc = check(b, true)) return rc;
    if (!phi || !f_out || !dZ_s || !dZ_q || !info || !b->y_s || !b->y_q || !b->priors) return ADKF_E_BADARG;
    return outer_common(b, phi, flags, true, f_out, g_phi_out, dZ_s, dZ_q, v, H, info);
}

}  // extern "C"
