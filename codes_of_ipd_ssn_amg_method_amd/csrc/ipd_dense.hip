// Dense Cholesky solves on the device: MATLAB's `\` on the (small, cold-path) symmetric positive
// definite systems of the reference --
//   * inner_solver = 1: zeta = Jk \ (-Fk_old)            Class1/APD_SsN_Class1.m:146-148,
//                                                         Class2/APD_SsN_Class2.m:152-156
//   * ideal interpolation (inter = 2): W = -Aff \ Afc     AMG/transfer.m:57-58
// MATLAB factors these sparse matrices with CHOLMOD (closed source, fill-reducing ordering); here
// the matrix is expanded to a dense row-major array and factored A = L*L' by a right-looking
// blocked Cholesky (64 x 64 blocks), then L*Y = B and L'*X = Y by blocked substitution.  Same
// solution up to rounding; sizes are bounded by the 2 GiB dense-scratch limit (n <= 16384).
// Cold paths: written for correctness and a sane run time (n = 4096: a few ms), not tuned.
#include "ipd_amg_internal.h"

namespace {

constexpr int DNB = 64;   // block size

// Cholesky of the diagonal block at (k0, k0), nb <= 64 rows, by one workgroup in LDS
__global__ __launch_bounds__(256) void k_potrf_diag(double* __restrict__ A, int ld, int k0, int nb,
                                                    int* __restrict__ bad) {
    __shared__ double t[DNB][DNB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        t[i][j] = j <= i ? A[(size_t)(k0 + i) * ld + k0 + j] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const double d = t[j][j];
        if (!(d > 0.0)) {           // not positive definite (or NaN): flag and stop
            if (tid == 0) atomicOr(bad, 1);
            return;
        }
        const double ljj = sqrt(d);
        __syncthreads();
        for (int i = j + tid; i < nb; i += 256) t[i][j] = i == j ? ljj : t[i][j] / ljj;
        __syncthreads();
        // trailing update of the block: t[i][c] -= t[i][j]*t[c][j], j < c <= i
        const int rem = nb - j - 1;
        for (int e = tid; e < rem * rem; e += 256) {
            const int i = j + 1 + e / rem, c = j + 1 + e % rem;
            if (c <= i) t[i][c] -= t[i][j] * t[c][j];
        }
        __syncthreads();
    }
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        if (j <= i) A[(size_t)(k0 + i) * ld + k0 + j] = t[i][j];
    }
}

// panel below the diagonal block: A_ik := A_ik * L_kk^-T, one thread per row
__global__ __launch_bounds__(256) void k_trsm_panel(double* __restrict__ A, int ld, int k0, int nb, int n) {
    __shared__ double l[DNB][DNB + 1];
    for (int e = threadIdx.x; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        l[i][j] = j <= i ? A[(size_t)(k0 + i) * ld + k0 + j] : 0.0;
    }
    __syncthreads();
    const int i = k0 + nb + blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double* row = A + (size_t)i * ld + k0;
    double x[DNB];
#pragma unroll
    for (int j = 0; j < DNB; ++j) x[j] = j < nb ? row[j] : 0.0;
#pragma unroll
    for (int j = 0; j < DNB; ++j) {
        if (j < nb) {
            double s = x[j];
#pragma unroll
            for (int t = 0; t < j; ++t) s -= x[t] * l[j][t];
            x[j] = s / l[j][j];
        }
    }
#pragma unroll
    for (int j = 0; j < DNB; ++j)
        if (j < nb) row[j] = x[j];
}

// C(64 x 64 tile) -= X(64 x nb) * Y(64 x nb)' ; each of the 256 threads owns a 4 x 4 sub-tile and the
// inner index is walked in two halves of 32 (two 64 x 33 LDS panels: 33 KB).
constexpr int DKH = DNB / 2;
struct TileAcc {
    double a[4][4];
};
__device__ __forceinline__ void tile_accumulate(TileAcc& acc, const double (*xs)[DKH + 1],
                                                const double (*ys)[DKH + 1]) {
    const int tr = (threadIdx.x / 16) * 4, tc = (threadIdx.x % 16) * 4;
    for (int t = 0; t < DKH; ++t) {
        double xa[4], yb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xa[u] = xs[tr + u][t];
            yb[u] = ys[tc + u][t];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc.a[u][v] += xa[u] * yb[v];
    }
}
__device__ __forceinline__ void tile_store(const TileAcc& acc, double* __restrict__ C, int ldc, int rows,
                                           int cols, bool lower_only, int r0, int c0) {
    const int tr = (threadIdx.x / 16) * 4, tc = (threadIdx.x % 16) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = tr + u, c = tc + v;
            if (r < rows && c < cols && (!lower_only || c0 + c <= r0 + r)) C[(size_t)r * ldc + c] -= acc.a[u][v];
        }
}

// trailing update A_ij -= L_ik * L_jk' over the lower-triangle tiles (i >= j > k)
__global__ __launch_bounds__(256) void k_syrk_tiles(double* __restrict__ A, int ld, int k0, int nb, int n) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    __shared__ double xs[DNB][DKH + 1], ys[DNB][DKH + 1];
    const int base = k0 + nb;
    const int r0 = base + bi * DNB, c0 = base + bj * DNB;
    TileAcc acc = {};
    for (int h = 0; h < 2; ++h) {
        for (int e = threadIdx.x; e < DNB * DKH; e += 256) {
            const int i = e / DKH, t = h * DKH + e % DKH;
            xs[i][e % DKH] = (r0 + i < n && t < nb) ? A[(size_t)(r0 + i) * ld + k0 + t] : 0.0;
            ys[i][e % DKH] = (c0 + i < n && t < nb) ? A[(size_t)(c0 + i) * ld + k0 + t] : 0.0;
        }
        __syncthreads();
        tile_accumulate(acc, xs, ys);
        __syncthreads();
    }
    tile_store(acc, A + (size_t)r0 * ld + c0, ld, min(DNB, n - r0), min(DNB, n - c0), true, r0, c0);
}

// diagonal-block solve of the substitution, one thread per right-hand side column:
// forward  (trans = 0): Y_k = L_kk^-1 B_k ;  backward (trans = 1): X_k = L_kk^-T Y_k
__global__ __launch_bounds__(256) void k_trsm_diag(const double* __restrict__ L, int ld, int k0, int nb,
                                                   double* __restrict__ B, int ldb, int nrhs, int trans) {
    __shared__ double l[DNB][DNB + 1];
    for (int e = threadIdx.x; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        l[i][j] = j <= i ? L[(size_t)(k0 + i) * ld + k0 + j] : 0.0;
    }
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nrhs) return;
    double x[DNB];
#pragma unroll
    for (int j = 0; j < DNB; ++j) x[j] = j < nb ? B[(size_t)(k0 + j) * ldb + c] : 0.0;
    if (!trans) {
#pragma unroll
        for (int j = 0; j < DNB; ++j)
            if (j < nb) {
                double s = x[j];
#pragma unroll
                for (int t = 0; t < j; ++t) s -= l[j][t] * x[t];
                x[j] = s / l[j][j];
            }
    } else {
#pragma unroll
        for (int jj = 0; jj < DNB; ++jj) {
            const int j = DNB - 1 - jj;
            if (j < nb) {
                double s = x[j];
#pragma unroll
                for (int t = j + 1; t < DNB; ++t)
                    if (t < nb) s -= l[t][j] * x[t];
                x[j] = s / l[j][j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DNB; ++j)
        if (j < nb) B[(size_t)(k0 + j) * ldb + c] = x[j];
}

// off-diagonal update of the substitution on 64 x 64 tiles of B:
// forward:  B_i -= L_ik * Y_k   (row blocks i below k; blockIdx.y counts them from k0 + nb)
// backward: B_i -= L_ki' * X_k  (row blocks i above k; blockIdx.y counts them from 0)
__global__ __launch_bounds__(256) void k_subst_tiles(const double* __restrict__ L, int ld, int k0, int nb,
                                                     int n, double* __restrict__ B, int ldb, int nrhs,
                                                     int trans) {
    __shared__ double xs[DNB][DKH + 1], ys[DNB][DKH + 1];
    const int r0 = (trans ? 0 : k0 + nb) + blockIdx.y * DNB;
    const int c0 = blockIdx.x * DNB;
    const int rend = trans ? k0 : n;
    if (r0 >= rend) return;   // uniform per workgroup
    TileAcc acc = {};
    for (int h = 0; h < 2; ++h) {
        for (int e = threadIdx.x; e < DNB * DKH; e += 256) {
            const int i = e / DKH, t = h * DKH + e % DKH;
            double lv = 0.0;
            if (r0 + i < rend && t < nb)
                lv = trans ? L[(size_t)(k0 + t) * ld + r0 + i] : L[(size_t)(r0 + i) * ld + k0 + t];
            xs[i][e % DKH] = lv;
            ys[i][e % DKH] = (c0 + i < nrhs && t < nb) ? B[(size_t)(k0 + t) * ldb + c0 + i] : 0.0;   // (Y_k)'
        }
        __syncthreads();
        tile_accumulate(acc, xs, ys);
        __syncthreads();
    }
    tile_store(acc, B + (size_t)r0 * ldb + c0, ldb, min(DNB, rend - r0), min(DNB, nrhs - c0), false, r0, c0);
}

__global__ void k_transpose_to_rows(int n, int nrhs, const double* __restrict__ cm,
                                    double* __restrict__ rm) {
    // cm: column-major n x nrhs (MATLAB), rm: row-major
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)n * nrhs;
         e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / nrhs, c = e % nrhs;
        rm[e] = cm[i + c * (size_t)n];
    }
}
__global__ void k_transpose_to_cols(int n, int nrhs, const double* __restrict__ rm,
                                    double* __restrict__ cm) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)n * nrhs;
         e += (size_t)gridDim.x * blockDim.x) {
        const size_t c = e / n, i = e % n;
        cm[e] = rm[i * nrhs + c];
    }
}

}  // namespace

// A (row-major, ld, lower triangle read) := L with A = L*L'.  Throws IPD_E_NUMERIC when a pivot is
// not positive (MATLAB's `\` would fall back to another factorisation there; the systems of the
// reference are positive definite by construction: bk1 > 0, principal blocks of Ae).
void dense_chol_factor(ipd_ctx* ctx, double* A, int n, int ld) {
    int* bad = ctx->scratch->alloc<int>(1);
    IPD_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    for (int k0 = 0; k0 < n; k0 += DNB) {
        const int nb = std::min(DNB, n - k0);
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), 0, ctx->stream, A, ld, k0, nb, bad);
        const int below = n - k0 - nb;
        if (below > 0) {
            hipLaunchKernelGGL(k_trsm_panel, dim3(cdiv(below, 256)), dim3(256), 0, ctx->stream, A, ld, k0,
                               nb, n);
            const int nt = cdiv(below, DNB);
            hipLaunchKernelGGL(k_syrk_tiles, dim3(nt, nt), dim3(256), 0, ctx->stream, A, ld, k0, nb, n);
        }
        IPD_KERNEL_CHECK();
    }
    IPD_REQUIRE(ctx->fetch1(bad) == 0, IPD_E_NUMERIC,
                "direct solve: the matrix is not symmetric positive definite (Cholesky pivot <= 0)");
}

// B (row-major n x nrhs, ldb) := (L*L')^-1 B
void dense_chol_solve(ipd_ctx* ctx, const double* L, int n, int ld, double* B, int nrhs, int ldb) {
    if (n == 0 || nrhs == 0) return;
    const int ct = cdiv(nrhs, DNB);
    for (int k0 = 0; k0 < n; k0 += DNB) {                       // L*Y = B
        const int nb = std::min(DNB, n - k0);
        hipLaunchKernelGGL(k_trsm_diag, dim3(cdiv(nrhs, 256)), dim3(256), 0, ctx->stream, L, ld, k0, nb, B,
                           ldb, nrhs, 0);
        const int below = n - k0 - nb;
        if (below > 0)
            hipLaunchKernelGGL(k_subst_tiles, dim3(ct, cdiv(below, DNB)), dim3(256), 0, ctx->stream, L, ld,
                               k0, nb, n, B, ldb, nrhs, 0);
        IPD_KERNEL_CHECK();
    }
    const int last = ((n - 1) / DNB) * DNB;
    for (int k0 = last; k0 >= 0; k0 -= DNB) {                   // L'*X = Y
        const int nb = std::min(DNB, n - k0);
        hipLaunchKernelGGL(k_trsm_diag, dim3(cdiv(nrhs, 256)), dim3(256), 0, ctx->stream, L, ld, k0, nb, B,
                           ldb, nrhs, 1);
        if (k0 > 0)
            hipLaunchKernelGGL(k_subst_tiles, dim3(ct, cdiv(k0, DNB)), dim3(256), 0, ctx->stream, L, ld, k0,
                               nb, n, B, ldb, nrhs, 1);
        IPD_KERNEL_CHECK();
    }
}

// x = A \ b for a sparse symmetric positive definite A on the device (b, x: n vectors)
void spd_solve_dev(ipd_ctx* ctx, const Csr& A, const double* b, double* x) {
    IPD_REQUIRE(A.nr == A.nc, IPD_E_ARG, "direct solve: the matrix must be square");
    const int n = A.nr;
    IPD_REQUIRE((size_t)n * n * 8 <= (size_t(2) << 30), IPD_E_LIMIT,
                "direct solve: dense factor above 2 GiB (n > 16384)");
    Arena& tmp = *ctx->scratch;
    double* D = tmp.alloc<double>((size_t)n * n);
    IPD_HIP(hipMemsetAsync(D, 0, sizeof(double) * (size_t)n * n, ctx->stream));
    csr_expand_dense(ctx, A, D, n);
    dense_chol_factor(ctx, D, n, n);
    if (x != b) IPD_HIP(hipMemcpyAsync(x, b, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    dense_chol_solve(ctx, D, n, n, x, 1, 1);
}

// X = A \ B, MATLAB's mldivide for sparse symmetric positive definite A and dense B (n x nrhs,
// column-major like MATLAB).  The reference's call sites: APD_SsN_Class1.m:148, APD_SsN_Class2.m:155,
// AMG/transfer.m:58.
extern "C" int ipd_spd_solve(ipd_ctx* ctx, const ipd_csc* A, const double* B, int64_t nrhs, double* X) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && B && X && nrhs >= 1, IPD_E_ARG, "NULL or empty argument");
        IPD_REQUIRE(A->nrows == A->ncols, IPD_E_ARG, "mldivide: A must be square");
        IPD_REQUIRE(nrhs <= (1 << 20), IPD_E_LIMIT, "mldivide: too many right-hand sides");
        CallScope scope(ctx);
        Arena& tmp = *ctx->scratch;
        Csr a;
        csr_upload_from_csc(ctx, tmp, A, true, &a);
        const int n = a.nr;
        IPD_REQUIRE((size_t)n * n * 8 <= (size_t(2) << 30) && (size_t)n * (size_t)nrhs * 8 <= (size_t(2) << 30),
                    IPD_E_LIMIT, "mldivide: dense scratch above 2 GiB");
        double* D = tmp.alloc<double>((size_t)n * n);
        IPD_HIP(hipMemsetAsync(D, 0, sizeof(double) * (size_t)n * n, ctx->stream));
        csr_expand_dense(ctx, a, D, n);
        dense_chol_factor(ctx, D, n, n);
        const size_t len = (size_t)n * (size_t)nrhs;
        double* cm = tmp.alloc<double>(len);
        double* rm = tmp.alloc<double>(len);
        ctx->upload(cm, B, len);
        const int g = (int)std::min<size_t>((len + 255) / 256, 4096);
        hipLaunchKernelGGL(k_transpose_to_rows, dim3(g), dim3(256), 0, ctx->stream, n, (int)nrhs,
                           (const double*)cm, rm);
        IPD_KERNEL_CHECK();
        dense_chol_solve(ctx, D, n, n, rm, (int)nrhs, (int)nrhs);
        hipLaunchKernelGGL(k_transpose_to_cols, dim3(g), dim3(256), 0, ctx->stream, n, (int)nrhs,
                           (const double*)rm, cm);
        IPD_KERNEL_CHECK();
        ctx->fetch(cm, X, len);
    });
}
