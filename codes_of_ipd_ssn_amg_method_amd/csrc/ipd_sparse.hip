// Device sparse-matrix utilities: upload/download in MATLAB's CSC layout,
// deterministic transpose, ordered SpGEMM, scans, generic CSR SpMV.
//
// Arithmetic order contract: every setup-phase product
// accumulates each output entry in ascending inner index with a separate
// multiply and add -- the order MATLAB's column-Gustavson sparse mtimes uses,
// restated row-wise -- so hierarchy matrices are bit-identical to the oracle's.
// This TU is compiled with -ffp-contract=off.
#pragma clang fp contract(off)

#include "ipd_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

// ---------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------
template <class T>
__global__ void k_fill(T* p, T v, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

template <class T>
static void fill_t(ipd_ctx* ctx, T* p, T v, size_t n) {
    if (n == 0) return;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_fill<T>, dim3(blocks), dim3(256), 0, ctx->stream, p, v, n);
    IPD_KERNEL_CHECK();
}
void fill_f64(ipd_ctx* ctx, double* p, double v, size_t n) { fill_t(ctx, p, v, n); }

// Exclusive scan of n ints by ONE workgroup (n is a row/column count, at most a few thousand on this path);
// out[n] receives the total.  in == out is allowed.  Most scans ride at the end of the launch that produces the
// counts (ScanTail, ipd_internal.h); these launches serve the rest.
__global__ __launch_bounds__(1024) void k_exscan(const int* __restrict__ in, int* out, int n,
                                                 volatile unsigned* box, unsigned ticket, int* extra) {
    __shared__ int wsum[16];
    const int carry = ipd_scan_counts(in, out, n, wsum);
    if (threadIdx.x == 0 && extra) *extra = carry;
    if (threadIdx.x == 0 && box) {   // post the total to the host mailbox (ipd_ctx::mailbox_wait)
        box[16] = (unsigned)carry;
        __threadfence_system();
        box[0] = ticket;
    }
}
void exclusive_scan_i32(ipd_ctx* ctx, const int* in, int* out, int n, int* total_dev) {
    hipLaunchKernelGGL(k_exscan, dim3(1), dim3(1024), 0, ctx->stream, in, out, n,
                       (volatile unsigned*)nullptr, 0u, total_dev);
    IPD_KERNEL_CHECK();
}

// scan + total on the host in one launch (the total sizes the next allocation)
int exclusive_scan_total(ipd_ctx* ctx, const int* in, int* out, int n) {
    unsigned ticket = 0;
    if (!ctx->mailbox_begin(&ticket)) {
        exclusive_scan_i32(ctx, in, out, n);
        return ctx->fetch1(out + n);
    }
    hipLaunchKernelGGL(k_exscan, dim3(1), dim3(1024), 0, ctx->stream, in, out, n, ctx->mailbox,
                       ticket, (int*)nullptr);
    IPD_KERNEL_CHECK();
    int total = 0;
    ctx->mailbox_wait(ticket, &total, sizeof(int));
    return total;
}

// ---------------------------------------------------------------------------
// allocation / copies
// ---------------------------------------------------------------------------
Csr csr_alloc(Arena& a, int nr, int nc, int nnz) {
    Csr m;
    m.nr = nr;
    m.nc = nc;
    m.nnz = nnz;
    m.rp = a.alloc<int>((size_t)nr + 1);
    m.ci = a.alloc<int>((size_t)nnz);
    m.va = a.alloc<double>((size_t)nnz);
    return m;
}

// the three arrays in one launch (three copies cost three dispatches and their gaps)
__global__ __launch_bounds__(256) void k_csr_copy(int nr1, int nnz, const int* __restrict__ rp,
                                                  const int* __restrict__ ci, const double* __restrict__ va,
                                                  int* __restrict__ orp, int* __restrict__ oci,
                                                  double* __restrict__ ova) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)nnz; i += stride) {
        oci[i] = ci[i];
        ova[i] = va[i];
        if (i < (size_t)nr1) orp[i] = rp[i];
    }
    for (size_t i = (size_t)nnz + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)nr1; i += stride)
        orp[i] = rp[i];
}

void csr_copy(ipd_ctx* ctx, Arena& dst, const Csr& A, Csr* out) {
    Csr m = csr_alloc(dst, A.nr, A.nc, A.nnz);
    const size_t n = std::max<size_t>((size_t)A.nnz, (size_t)A.nr + 1);
    hipLaunchKernelGGL(k_csr_copy, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0,
                       ctx->stream, A.nr + 1, A.nnz, A.rp, A.ci, A.va, m.rp, m.ci, m.va);
    IPD_KERNEL_CHECK();
    *out = m;
}

// ---------------------------------------------------------------------------
// deterministic transpose through a per-column row bitmap
// ---------------------------------------------------------------------------
// Row r of A sets bit r of column c's bitmap for every stored (r,c); the slot of
// (r,c) inside column c is the number of set bits below r.  Integer atomics
// (OR) give a placement that is independent of execution order, so the result
// has ascending row indices without any sort.
__global__ void k_tr_mark(int nr, const int* __restrict__ rp, const int* __restrict__ ci,
                          unsigned* __restrict__ bits, int wpc) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < nr; r += nwaves) {
        const int b = rp[r], e = rp[r + 1];
        for (int t = b + lane; t < e; t += 64)
            atomicOr(&bits[(size_t)ci[t] * wpc + (r >> 5)], 1u << (r & 31));
    }
}

// one wave per column: exclusive prefix of popcounts over the column's words
__global__ void k_tr_prefix(int nc, const unsigned* __restrict__ bits, int* __restrict__ pref,
                            int* __restrict__ cnt, int wpc) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int c = wave; c < nc; c += nwaves) {
        int carry = 0;
        for (int base = 0; base < wpc; base += 64) {
            const int w = base + lane;
            const int v = w < wpc ? __popc(bits[(size_t)c * wpc + w]) : 0;
            int x = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                int y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            if (w < wpc) pref[(size_t)c * wpc + w] = carry + x - v;
            carry += __shfl(x, 63);
        }
        if (lane == 0) cnt[c] = carry;
    }
}

// head_cnt != NULL: the transpose's row pointers are still the column counts -- every workgroup scans them for
// itself and workgroup 0 stores them at head_rp (scan_head, ipd_internal.h; at most SCAN_HEAD_MAX columns)
__global__ __launch_bounds__(256) void k_tr_scatter(int nr, const int* __restrict__ rp,
                                                    const int* __restrict__ ci, const double* __restrict__ va,
                                                    const unsigned* __restrict__ bits,
                                                    const int* __restrict__ pref, int wpc, const int* trp,
                                                    int* __restrict__ tci, double* __restrict__ tva,
                                                    const int* __restrict__ head_cnt, int ncols, int* head_rp) {
    __shared__ ScanHeadLds L;
    if (head_cnt) {
        scan_head(head_cnt, ncols, head_rp, nullptr, L);
        trp = L.rp;
    }
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < nr; r += nwaves) {
        const int b = rp[r], e = rp[r + 1];
        const unsigned below = (1u << (r & 31)) - 1u;
        for (int t = b + lane; t < e; t += 64) {
            const int c = ci[t];
            const size_t w = (size_t)c * wpc + (r >> 5);
            const int pos = trp[c] + pref[w] + __popc(bits[w] & below);
            tci[pos] = r;
            tva[pos] = va[t];
        }
    }
}

void csr_transpose(ipd_ctx* ctx, Arena& dst, const Csr& A, Csr* At) {
    Arena& tmp = *ctx->scratch;
    const int wpc = (A.nr + 31) / 32;
    const size_t words = (size_t)A.nc * (size_t)(wpc ? wpc : 1);
    IPD_REQUIRE(words * 8 <= (size_t(2) << 30), IPD_E_LIMIT,
                "csr_transpose: matrix too large for the bitmap transpose (limit 2 GiB)");
    Csr T = csr_alloc(dst, A.nc, A.nr, A.nnz);
    unsigned* bits = zeroed<unsigned>(ctx, words);
    int* pref = tmp.alloc<int>(words);
    int* cnt = tmp.alloc<int>((size_t)A.nc + 1);
    const int rows_blocks = std::max(1, std::min(cdiv(A.nr, 4), 2048));
    const int cols_blocks = std::max(1, std::min(cdiv(A.nc, 4), 2048));
    if (A.nnz) {
        hipLaunchKernelGGL(k_tr_mark, dim3(rows_blocks), dim3(256), 0, ctx->stream, A.nr, A.rp,
                           A.ci, bits, wpc);
        IPD_KERNEL_CHECK();
    }
    hipLaunchKernelGGL(k_tr_prefix, dim3(cols_blocks), dim3(256), 0, ctx->stream, A.nc, bits, pref,
                       cnt, wpc);
    IPD_KERNEL_CHECK();
    // the column counts become the transpose's row pointers inside the scatter (scan_head), or in a launch of
    // their own when there are too many columns for that (or nothing to scatter)
    const bool head = A.nnz > 0 && A.nc <= SCAN_HEAD_MAX;
    if (!head) exclusive_scan_i32(ctx, cnt, T.rp, A.nc);
    if (A.nnz) {
        hipLaunchKernelGGL(k_tr_scatter, dim3(rows_blocks), dim3(256), 0, ctx->stream, A.nr, A.rp,
                           A.ci, A.va, bits, pref, wpc, (const int*)T.rp, T.ci, T.va,
                           head ? (const int*)cnt : (const int*)nullptr, A.nc, T.rp);
        IPD_KERNEL_CHECK();
    }
    *At = T;
}

// ---------------------------------------------------------------------------
// host <-> device in MATLAB's CSC layout
// ---------------------------------------------------------------------------
void csr_upload_from_csc(ipd_ctx* ctx, Arena& a, const ipd_csc* A, bool symmetric, Csr* out) {
    IPD_REQUIRE(A && A->jc && (A->nnz == 0 || (A->ir && A->pr)), IPD_E_ARG,
                "sparse input: NULL array");
    IPD_REQUIRE(A->nrows >= 0 && A->ncols >= 0 && A->nnz >= 0, IPD_E_ARG,
                "sparse input: negative dimension");
    IPD_REQUIRE(A->nrows < (int64_t(1) << 30) && A->ncols < (int64_t(1) << 30) &&
                    A->nnz < (int64_t(1) << 31) - 64,
                IPD_E_LIMIT, "sparse input: dimensions exceed the int32 device index range");
    const int nr = (int)A->nrows, nc = (int)A->ncols, nnz = (int)A->nnz;
    IPD_REQUIRE(A->jc[0] == 0 && A->jc[nc] == nnz, IPD_E_ARG, "sparse input: bad column pointers");
    std::vector<int> jc((size_t)nc + 1), ir((size_t)nnz);
    for (int c = 0; c <= nc; ++c) jc[c] = (int)A->jc[c];
    for (int c = 0; c < nc; ++c) {
        IPD_REQUIRE(A->jc[c] >= 0 && A->jc[c] <= A->jc[c + 1] && A->jc[c + 1] <= nnz, IPD_E_ARG,
                    "sparse input: column pointers decrease or exceed nnz");
        for (int64_t t = A->jc[c]; t < A->jc[c + 1]; ++t) {
            const int64_t r = A->ir[t];
            IPD_REQUIRE(r >= 0 && r < nr, IPD_E_ARG, "sparse input: row index out of range");
            IPD_REQUIRE(t == A->jc[c] || A->ir[t - 1] < r, IPD_E_ARG,
                        "sparse input: row indices must be strictly ascending per column");
            ir[t] = (int)r;
        }
    }
    // The CSC arrays of A are the CSR arrays of A'.
    Arena& where = symmetric ? a : *ctx->scratch;
    Csr T = csr_alloc(where, nc, nr, nnz);
    ctx->upload(T.rp, jc.data(), (size_t)nc + 1);
    if (nnz) {
        ctx->upload(T.ci, ir.data(), (size_t)nnz);
        ctx->upload(T.va, A->pr, (size_t)nnz);
    }
    if (symmetric) {
        IPD_REQUIRE(nr == nc, IPD_E_ARG, "symmetric sparse input must be square");
        *out = T;
    } else {
        csr_transpose(ctx, a, T, out);
    }
}

void csr_download_as_csc(ipd_ctx* ctx, const Csr& m, bool is_transposed, ipd_csc_out* out) {
    IPD_REQUIRE(out, IPD_E_ARG, "output matrix is NULL");
    Csr T = m;  // CSR arrays of A' == CSC arrays of A
    if (!is_transposed) csr_transpose(ctx, *ctx->scratch, m, &T);
    const int ncols = T.nr, nrows = T.nc, nnz = T.nnz;
    std::vector<int> rp((size_t)ncols + 1), ci((size_t)nnz);
    out->nrows = nrows;
    out->ncols = ncols;
    out->nnz = nnz;
    out->jc = (int64_t*)std::malloc(sizeof(int64_t) * ((size_t)ncols + 1));
    out->ir = (int64_t*)std::malloc(sizeof(int64_t) * (size_t)(nnz ? nnz : 1));
    out->pr = (double*)std::malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    if (!out->jc || !out->ir || !out->pr) {
        ipd_csc_free(out);
        throw IpdError(IPD_E_NOMEM, "out of host memory");
    }
    ctx->fetch(T.rp, rp.data(), (size_t)ncols + 1);
    if (nnz) {
        ctx->fetch(T.ci, ci.data(), (size_t)nnz);
        ctx->fetch(T.va, out->pr, (size_t)nnz);
    }
    for (int c = 0; c <= ncols; ++c) out->jc[c] = rp[c];
    for (int t = 0; t < nnz; ++t) out->ir[t] = ci[t];
}

// ---------------------------------------------------------------------------
// generic CSR SpMV  y = A*x   (utility / ipd_spmv_dev; the cycle has fused forms)
// ---------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256) void k_spmv(int nr, const int* __restrict__ rp,
                                              const int* __restrict__ ci,
                                              const double* __restrict__ va,
                                              const double* __restrict__ x,
                                              double* __restrict__ y) {
    const int gl = threadIdx.x & (L - 1);
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) / L;
    const int ngrp = (gridDim.x * blockDim.x) / L;
    for (int r = grp; r < nr; r += ngrp) {
        const int b = rp[r], e = rp[r + 1];
        double s = 0.0;
        for (int t = b + gl; t < e; t += L) s += va[t] * x[ci[t]];
#pragma unroll
        for (int d = L >> 1; d > 0; d >>= 1) s += __shfl_xor(s, d);
        if (gl == 0) y[r] = s;
    }
}

void csr_spmv(ipd_ctx* ctx, const Csr& A, const double* x, double* y) {
    if (A.nr == 0) return;
    const double avg = (double)A.nnz / (double)A.nr;
    auto launch = [&](auto kern, int L) {
        const long long threads = (long long)A.nr * L;
        const int blocks = (int)std::max<long long>(1, std::min<long long>((threads + 255) / 256, 8192));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, ctx->stream, A.nr, A.rp, A.ci, A.va, x,
                           y);
        IPD_KERNEL_CHECK();
    };
    if (avg <= 6)
        launch(k_spmv<4>, 4);
    else if (avg <= 24)
        launch(k_spmv<16>, 16);
    else
        launch(k_spmv<64>, 64);
}

// ---------------------------------------------------------------------------
// ordered SpGEMM  C = X*Y
// ---------------------------------------------------------------------------
// One workgroup per output row keeps a dense accumulator row in LDS.  The inner-index loop is
// sequential (ascending k = ascending stored order of X's row) and lanes only split the columns
// of Y's row k, which are distinct, so every C(i,j) receives its terms one at a time in
// ascending k: bit-identical to a sequential Gustavson product.  The row goes to a dense scratch
// matrix; a second kernel compacts rows (exact zeros dropped, as MATLAB's sparse mtimes does).
//
// Wide, sparse outputs (level 2 of an m = n = 8192 problem: 16384 columns, tens of entries per
// row) would spend their time zeroing and scanning the accumulator, so the row keeps a bitmap of
// the 64-column blocks it touched: only those are written out, re-zeroed and later compacted
// (`rowbits`: one bit per block, ceil(nc/4096) words per row).
__global__ __launch_bounds__(256) void k_spgemm_rows(int nr, int nc, const int* __restrict__ xrp,
                                                     const int* __restrict__ xci,
                                                     const double* __restrict__ xva,
                                                     const int* __restrict__ yrp,
                                                     const int* __restrict__ yci,
                                                     const double* __restrict__ yva,
                                                     double* __restrict__ dense,
                                                     int* rowcnt,
                                                     unsigned long long* __restrict__ rowbits,
                                                     const ScanTail st) {
    // blockDim.x = 64 (short rows of Y) or 256 (long rows): more lanes per inner step
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int wcnt[4];
    __shared__ unsigned long long touched[4];   // nc <= 16384: at most 256 blocks
    double* acc = reinterpret_cast<double*>(smem_raw);
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, W = T >> 6;
    const int nb = (nc + 63) >> 6, nw = (nb + 63) >> 6;
    for (int j = tid; j < nc; j += T) acc[j] = 0.0;
    if (tid < 4) touched[tid] = 0ull;
    __syncthreads();
    for (int i = blockIdx.x; i < nr; i += gridDim.x) {
        const int xb = xrp[i], xe = xrp[i + 1];
        for (int e = xb; e < xe; ++e) {
            const int k = xci[e];
            const double a = xva[e];
            const int yb = yrp[k], ye = yrp[k + 1];
            for (int t = yb + tid; t < ye; t += T) {
                const int j = yci[t];
                const double prod = a * yva[t];
                acc[j] = acc[j] + prod;
                const unsigned long long bit = 1ull << ((j >> 6) & 63);
                if (!(touched[j >> 12] & bit)) atomicOr(&touched[j >> 12], bit);
            }
            __syncthreads();
        }
        if (xb == xe) __syncthreads();
        // write out, count and re-zero the touched blocks: wave w takes every W-th of them
        int nz = 0;
        double* drow = dense + (size_t)i * nc;
        int seen = 0;
        for (int w = 0; w < nw; ++w) {
            unsigned long long m = touched[w];
            while (m) {
                const int b = (w << 6) + __builtin_ctzll(m);
                m &= m - 1;
                if ((seen++ % W) != wave) continue;
                const int j = (b << 6) + lane;
                if (j < nc) {
                    const double v = acc[j];
                    drow[j] = v;
                    nz += (v != 0.0);
                    acc[j] = 0.0;
                }
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nz += __shfl_xor(nz, d);
        if (lane == 0) wcnt[wave] = nz;
        __syncthreads();
        if (tid < nw) {
            rowbits[(size_t)i * nw + tid] = touched[tid];
            touched[tid] = 0ull;
        }
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < W; ++w) tot += wcnt[w];
            if (st.out)
                scan_put(rowcnt, i, tot);
            else
                rowcnt[i] = tot;
        }
        __syncthreads();
    }
    scan_tail(st);
}

// one wave per row: ordered compaction of the dense row into CSR; with `rowbits` only the
// 64-column blocks flagged there hold data (ascending bit order = ascending columns)
__global__ __launch_bounds__(256) void k_dense_compact(int nr, int nc, int ld,
                                                       const double* __restrict__ dense,
                                                       const unsigned long long* __restrict__ rowbits,
                                                       const int* rp,
                                                       int* __restrict__ ci,
                                                       double* __restrict__ va,
                                                       const int* __restrict__ head_cnt, int* head_rp,
                                                       int* head_total, const LazyPost post, int* head_max) {
    // head_cnt != NULL: the row pointers are still plain counts -- every workgroup scans them for itself and
    // workgroup 0 stores them at head_rp (scan_head, ipd_internal.h; nr <= SCAN_HEAD_MAX).  post.box != NULL:
    // workgroup 0 then posts post.n device words (the level's lazy counts, this product's total among them)
    // to the host mailbox -- the fetch that would follow, without its launch.
    __shared__ ScanHeadLds L;
    if (head_cnt) {
        const int total = scan_head(head_cnt, nr, head_rp, head_total, L, head_max);
        rp = L.rp;
        if (post.box && blockIdx.x == 0 && threadIdx.x == 0) {
            for (int w = 0; w < post.n; ++w)
                post.box[16 + w] = post.src + w == head_total ? (unsigned)total : (unsigned)post.src[w];
            __threadfence_system();
            post.box[0] = post.ticket;
        }
    }
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nb = (nc + 63) >> 6, nw = (nb + 63) >> 6;
    for (int i = wave; i < nr; i += nwaves) {
        int base = rp[i];
        if (base == rp[i + 1]) continue;
        const double* drow = dense + (size_t)i * ld;
        auto block = [&](int j0) {
            const int j = j0 + lane;
            const double v = j < nc ? drow[j] : 0.0;
            const bool nzf = v != 0.0;
            const unsigned long long mask = __ballot(nzf);
            if (nzf) {
                const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                ci[pos] = j;
                va[pos] = v;
            }
            base += __popcll(mask);
        };
        if (rowbits) {
            for (int w = 0; w < nw; ++w) {
                unsigned long long m = rowbits[(size_t)i * nw + w];
                while (m) {
                    block(((w << 6) + __builtin_ctzll(m)) << 6);
                    m &= m - 1;
                }
            }
        } else {
            for (int j0 = 0; j0 < nc; j0 += 64) block(j0);
        }
    }
}

// ---------------------------------------------------------------------------
// dense-tile variant of the ordered product, for filled-in operands
// ---------------------------------------------------------------------------
// When the masks are dense (BASELINE regime D, early Newton iterations) the level-2 Galerkin
// products are products of nearly full matrices and the row kernel above spends its time on LDS
// read-modify-writes.  Here the operands are expanded to zero-padded dense arrays and a 64x64
// output tile per workgroup accumulates in registers, 4x4 outputs per thread, with the inner
// index still walked strictly in ascending order: every C(i,j) receives x(i,k)*y(k,j) one term
// at a time for k = 0,1,2,...  The terms a structural zero contributes are +0.0, which leave a
// partial sum unchanged, so the result is bit-identical to the row kernel's (this translation
// unit is compiled with -ffp-contract=off: multiply, round, add, round).
constexpr int GT = 64;   // output tile edge
constexpr int GK = 16;   // inner-index tile

// both operands of a tile product in one launch (blockIdx.y picks the matrix)
struct ExpandPair {
    int nr[2], ld[2];
    const int* rp[2];
    const int* ci[2];
    const double* va[2];
    double* dense[2];
};
__global__ __launch_bounds__(256) void k_csr_expand2(const ExpandPair e) {
    const int q = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int* __restrict__ rp = e.rp[q];
    const int* __restrict__ ci = e.ci[q];
    const double* __restrict__ va = e.va[q];
    for (int i = wave; i < e.nr[q]; i += nwaves) {
        double* drow = e.dense[q] + (size_t)i * e.ld[q];
        for (int t = rp[i] + lane; t < rp[i + 1]; t += 64) drow[ci[t]] = va[t];
    }
}

__global__ __launch_bounds__(256) void k_csr_expand(int nr, int ld, const int* __restrict__ rp,
                                                    const int* __restrict__ ci,
                                                    const double* __restrict__ va,
                                                    double* __restrict__ dense) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < nr; i += nwaves) {
        double* drow = dense + (size_t)i * ld;
        for (int t = rp[i] + lane; t < rp[i + 1]; t += 64) drow[ci[t]] = va[t];
    }
}

// R x R outputs per thread, 16 x 16 threads: a (16 R)-edge output tile per workgroup.  R = 4 for large products;
// R = 2 when 64-edge tiles would leave most of the chip idle (a 330 x 1024 x 330 product is 36 of them, each
// walking the 1024 inner indices alone: 29 us; 121 tiles of edge 32 take a third of that).  The order of the
// additions into every C(i,j) does not depend on R.
template <int R>
__global__ __launch_bounds__(256) void k_gemm_ordered(int nkp, int ncp,
                                                      const double* __restrict__ X,
                                                      const double* __restrict__ Y,
                                                      double* __restrict__ C) {
    // X: (rows padded to GT) x nkp, Y: nkp x ncp, C: rows x ncp; nkp % GK == 0, ncp % GT == 0
    constexpr int TE = 16 * R;   // tile edge
    __shared__ __attribute__((aligned(16))) double xs[GK][TE];   // transposed: xs[k][row]
    __shared__ __attribute__((aligned(16))) double ys[GK][TE];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int r0 = blockIdx.y * TE, c0 = blockIdx.x * TE;
    double acc[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < R; ++c) acc[r][c] = 0.0;
    // fetch roles: X tile TE rows x 16 k (R consecutive k per thread), Y tile 16 k x TE columns
    const int xr = tid / (16 / R), xk = (tid % (16 / R)) * R;
    const int yk = tid >> 4, yc = (tid & 15) * R;
    const double* xp = X + (size_t)(r0 + xr) * nkp + xk;
    const double* yp = Y + (size_t)yk * ncp + c0 + yc;
    double xf[R], yf[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        xf[u] = xp[u];
        yf[u] = yp[u];
    }
    for (int k0 = 0; k0 < nkp; k0 += GK) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u) {
            xs[xk + u][xr] = xf[u];
            ys[yk][yc + u] = yf[u];
        }
        __syncthreads();
        if (k0 + GK < nkp) {   // next tile's fetch overlaps this tile's arithmetic
            xp += GK;
            yp += (size_t)GK * ncp;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                xf[u] = xp[u];
                yf[u] = yp[u];
            }
        }
#pragma unroll
        for (int kk = 0; kk < GK; ++kk) {
            double a[R], b[R];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                a[u] = xs[kk][ty * R + u];
                b[u] = ys[kk][tx * R + u];
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const double prod = a[r] * b[c];
                    acc[r][c] = acc[r][c] + prod;
                }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double* crow = C + (size_t)(r0 + ty * R + r) * ncp + c0 + tx * R;
#pragma unroll
        for (int c = 0; c < R; ++c) crow[c] = acc[r][c];
    }
}

// one wave per row: number of nonzeros of a dense row
__global__ __launch_bounds__(256) void k_dense_rowcount(int nr, int nc, int ld,
                                                        const double* __restrict__ dense,
                                                        int* rowcnt, const ScanTail st) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < nr; i += nwaves) {
        const double* drow = dense + (size_t)i * ld;
        int nz = 0;
        for (int j = lane; j < nc; j += 64) nz += (drow[j] != 0.0);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nz += __shfl_xor(nz, d);
        if (lane == 0) {
            if (st.out)
                scan_put(rowcnt, i, nz);
            else
                rowcnt[i] = nz;
        }
    }
    scan_tail(st);
}

// The same product for short rows of Y, one WAVE per output row.  k_spgemm_rows pays three dependent
// global round trips and a barrier per entry of X's row (a hub row of P' with 300 entries: 85 us).
// Here the lanes read the metadata of 64 entries at once (column, value, row range of Y), the rows of Y
// of the next D entries are in flight while the current D are applied, and an entry costs one LDS
// read-modify-write: the wave's LDS operations execute in program order, so every C(i,j) still
// receives its terms one at a time in ascending k.  Rows of Y longer than 64 entries finish in a loop.
__device__ __forceinline__ double sp_readlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int D>
__global__ __launch_bounds__(64) void k_spgemm_rows_w(int nr, int nc, const int* __restrict__ xrp,
                                                      const int* __restrict__ xci,
                                                      const double* __restrict__ xva,
                                                      const int* __restrict__ yrp,
                                                      const int* __restrict__ yci,
                                                      const double* __restrict__ yva,
                                                      double* __restrict__ dense,
                                                      int* __restrict__ rowcnt,
                                                      unsigned long long* __restrict__ rowbits) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ unsigned long long touched[4];   // nc <= 16384: at most 256 blocks
    double* acc = reinterpret_cast<double*>(smem_raw);
    const int lane = threadIdx.x;
    const int nb = (nc + 63) >> 6, nw = (nb + 63) >> 6;
    for (int j = lane; j < nc; j += 64) acc[j] = 0.0;
    if (lane < 4) touched[lane] = 0ull;
    __syncthreads();
    for (int i = blockIdx.x; i < nr; i += gridDim.x) {
        const int xb = xrp[i], xe = xrp[i + 1];
        unsigned long long mybits = 0ull;
        for (int e0 = xb; e0 < xe; e0 += 64) {
            const int cnt = min(64, xe - e0);
            const bool mine = lane < cnt;
            const int kk = mine ? xci[e0 + lane] : 0;
            const double aa = mine ? xva[e0 + lane] : 0.0;
            const int yb = yrp[kk];
            const int yn = mine ? yrp[kk + 1] - yb : 0;
            int jA[D], jB[D];
            double vA[D], vB[D];
            auto load = [&](int u0, int* jj, double* vv) __attribute__((always_inline)) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int u = u0 + d;   // uniform
                    jj[d] = -1;
                    vv[d] = 0.0;
                    if (u < cnt) {
                        const int b = __builtin_amdgcn_readlane(yb, u), n = __builtin_amdgcn_readlane(yn, u);
                        if (lane < n) {
                            jj[d] = yci[b + lane];
                            vv[d] = yva[b + lane];
                        }
                    }
                }
            };
            auto add = [&](int j, double prod) __attribute__((always_inline)) {
                acc[j] = acc[j] + prod;
                const unsigned long long bit = 1ull << ((j >> 6) & 63);
                if (nw == 1)
                    mybits |= bit;   // (one word: kept per lane, OR-ed over the wave at the end of the row)
                else if (!(touched[j >> 12] & bit))
                    atomicOr(&touched[j >> 12], bit);
            };
            auto apply = [&](int u0, const int* jj, const double* vv) __attribute__((always_inline)) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int u = u0 + d;   // uniform
                    if (u < cnt) {
                        const double a = sp_readlane(aa, u);
                        if (jj[d] >= 0) add(jj[d], a * vv[d]);
                        const int n = __builtin_amdgcn_readlane(yn, u);
                        if (n > 64) {   // the rest of a long row of Y (distinct columns: lane order is free)
                            const int b = __builtin_amdgcn_readlane(yb, u);
                            for (int t = b + 64 + lane; t < b + n; t += 64) add(yci[t], a * yva[t]);
                        }
                    }
                }
            };
            load(0, jA, vA);
            for (int u0 = 0; u0 < cnt; u0 += 2 * D) {
                load(u0 + D, jB, vB);
                apply(u0, jA, vA);
                load(u0 + 2 * D, jA, vA);
                apply(u0 + D, jB, vB);
            }
        }
        // write out, count and re-zero the touched blocks
        if (nw == 1) {
            unsigned lo = (unsigned)mybits, hi = (unsigned)(mybits >> 32);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                lo |= __shfl_xor(lo, d);
                hi |= __shfl_xor(hi, d);
            }
            if (lane == 0) touched[0] = ((unsigned long long)hi << 32) | lo;
        }
        __syncthreads();
        int nz = 0;
        double* drow = dense + (size_t)i * nc;
        for (int w = 0; w < nw; ++w) {
            unsigned long long m = touched[w];
            while (m) {
                const int b = (w << 6) + __builtin_ctzll(m);
                m &= m - 1;
                const int j = (b << 6) + lane;
                if (j < nc) {
                    const double v = acc[j];
                    drow[j] = v;
                    nz += (v != 0.0);
                    acc[j] = 0.0;
                }
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nz += __shfl_xor(nz, d);
        __syncthreads();
        if (lane < nw) {
            rowbits[(size_t)i * nw + lane] = touched[lane];
            touched[lane] = 0ull;
        }
        if (lane == 0) rowcnt[i] = nz;   // (plain counts: one-wave workgroups leave the scan to the consumer)
        __syncthreads();
    }
}

static inline size_t round_up(size_t v, size_t q) { return (v + q - 1) / q * q; }

// Which product kernel is expected to finish first (measured on MI355X, round 4).  The row kernels are a
// dependent chain per output row, one step per entry of X's row: the one-wave kernel (short rows of Y) takes
// about 0.16 us per entry with the rows of Y prefetched eight deep, the 256-thread kernel 0.35 us + 0.25 us per
// 256 entries of Y's row with a barrier per step; rows are spread over the CUs as LDS allows.  The tile kernel
// walks the padded rows x inner x columns box 16 inner indices at a time -- about 0.9 us per step and wave of
// 256 64-edge tiles, 0.23 us with the 32-edge tiles small products get -- plus 20-40 us for the expansion of
// the operands and the row count.  IPD_PRODUCT=rows|tiles overrides the choice (tests compare the two bit for bit).
static bool spgemm_prefers_tiles(const Csr& X, const Csr& Y, size_t* bytes, int x_maxrow) {
    const size_t nrp = round_up((size_t)X.nr, GT), nkp = round_up((size_t)X.nc, GT),
                 ncp = round_up((size_t)Y.nc, GT);
    *bytes = 8 * (nrp * nkp + nkp * ncp + nrp * ncp);
    if (X.nr == 0 || X.nc == 0 || Y.nc == 0 || X.nnz == 0) return false;
    if (*bytes > (size_t(12) << 30)) return false;   // 12 GiB of dense scratch at most
    if (const char* e = getenv("IPD_PRODUCT")) {
        if (!strcmp(e, "tiles")) return true;
        if (!strcmp(e, "rows")) return false;
    }
    // (one round of rows: the launch is as slow as its longest row)
    const double ylen = (double)Y.nnz / Y.nr;
    const double xlen = (X.nr <= 256 * 8) ? std::max((double)X.nnz / X.nr, (double)x_maxrow) : (double)X.nnz / X.nr;
    const double lds_rows = std::max(1.0, std::min(ylen < 96.0 ? 32.0 : 8.0, 160.0 * 1024 / (8.0 * Y.nc + 64)));
    const double row_rounds = std::ceil(X.nr / (256.0 * lds_rows));
    const double t_rows = row_rounds * xlen * (ylen < 96.0 ? 0.16 : 0.35 + 0.25 * std::ceil(ylen / 256.0));
    const double tiles = (double)(nrp / GT) * (double)(ncp / GT);
    const double steps = (double)(nkp / GK);
    const double t_walk = tiles >= 256.0 ? 20.0 + std::ceil(tiles / 256.0) * steps * 0.9
                                         : std::ceil(4.0 * tiles / 1024.0) * steps * 0.23;
    const double t_tiles = 20.0 + t_walk + (double)*bytes / 3.0e6;   // (operand block zeroed and written at ~3 TB/s)
    if (const char* dbg = getenv("IPD_DEBUG_LEVELS"); dbg && dbg[0] == '1')
        std::fprintf(stderr, "[ipd] product %d x %d x %d: x row %.1f, y row %.1f entries; model rows %.1f us, tiles %.1f us\n",
                     X.nr, X.nc, Y.nc, xlen, ylen, t_rows, t_tiles);
    return t_tiles < t_rows;
}

void csr_spgemm(ipd_ctx* ctx, Arena& dst, const Csr& X, const Csr& Y, Csr* C, int* total_dev, LazyPost* post,
                int* maxrow_dev, int x_maxrow) {
    if (post) post->box = nullptr;
    IPD_REQUIRE(X.nc == Y.nr, IPD_E_ARG, "spgemm: inner dimensions differ");
    const int nr = X.nr, nc = Y.nc;
    Arena& tmp = *ctx->scratch;
    Csr out;
    out.nr = nr;
    out.nc = nc;
    out.rp = dst.alloc<int>((size_t)nr + 1);
    if (nr == 0) {
        IPD_HIP(hipMemsetAsync(out.rp, 0, sizeof(int), ctx->stream));
        if (total_dev) IPD_HIP(hipMemsetAsync(total_dev, 0, sizeof(int), ctx->stream));
        out.nnz = 0;
        out.ci = dst.alloc<int>(0);
        out.va = dst.alloc<double>(0);
        *C = out;
        return;
    }
    double* dense = nullptr;
    unsigned long long* rowbits = nullptr;   // row kernel only: the 64-column blocks a row touched
    int ld = nc;
    size_t tile_bytes = 0;
    const bool tiles = spgemm_prefers_tiles(X, Y, &tile_bytes, x_maxrow);
    const int threads = (Y.nr > 0 && (double)Y.nnz / Y.nr >= 96.0) ? 256 : 64;
    // The row pointers.  With a lazy count (total_dev) the compaction scans the plain counts on its way in
    // (scan_head).  Otherwise the host needs the total to size the arrays: producers with 256-thread workgroups
    // scan their own counts at the end of the launch and post it (ScanTail); the one-wave row kernel is followed
    // by a scan launch.
    const bool head_ok = total_dev && nr <= SCAN_HEAD_MAX;
    const bool tail = !head_ok && (tiles || threads == 256);
    int* rowcnt = tail ? zeroed<int>(ctx, (size_t)nr + 1) : tmp.alloc<int>((size_t)nr + 1);
    ScanTail st;
    std::unique_ptr<TailTotal> tt;
    if (tail) {
        if (total_dev)
            st = scan_tail_lazy(rowcnt, out.rp, nr, total_dev);
        else {
            tt.reset(new TailTotal(ctx, rowcnt, out.rp, nr));
            st = tt->t;
        }
    }
    if (tiles) {
        const size_t nrp = round_up((size_t)nr, GT), nkp = round_up((size_t)X.nc, GT),
                     ncp = round_up((size_t)nc, GT);
        double* xd = zeroed<double>(ctx, nrp * nkp + nkp * ncp);   // both operands
        double* yd = xd + nrp * nkp;
        dense = tmp.alloc<double>(nrp * ncp);
        ld = (int)ncp;
        ExpandPair ep;
        ep.nr[0] = X.nr, ep.ld[0] = (int)nkp, ep.rp[0] = X.rp, ep.ci[0] = X.ci, ep.va[0] = X.va, ep.dense[0] = xd;
        ep.nr[1] = Y.nr, ep.ld[1] = (int)ncp, ep.rp[1] = Y.rp, ep.ci[1] = Y.ci, ep.va[1] = Y.va, ep.dense[1] = yd;
        hipLaunchKernelGGL(k_csr_expand2, dim3(std::max(1, std::min(cdiv(std::max(X.nr, Y.nr), 4), 4096)), 2),
                           dim3(256), 0, ctx->stream, ep);
        if ((nrp / GT) * (ncp / GT) >= 256)
            hipLaunchKernelGGL(k_gemm_ordered<4>, dim3((unsigned)(ncp / GT), (unsigned)(nrp / GT)),
                               dim3(256), 0, ctx->stream, (int)nkp, (int)ncp, xd, yd, dense);
        else   // few 64-edge tiles: 32-edge ones spread the inner-index walk over four times the CUs
            hipLaunchKernelGGL(k_gemm_ordered<2>, dim3((unsigned)(ncp / 32), (unsigned)(nrp / 32)),
                               dim3(256), 0, ctx->stream, (int)nkp, (int)ncp, xd, yd, dense);
        hipLaunchKernelGGL(k_dense_rowcount, dim3(std::min(cdiv(nr, 4), 4096)), dim3(256), 0,
                           ctx->stream, nr, nc, ld, dense, rowcnt, st);
        IPD_KERNEL_CHECK();
    } else {
        IPD_REQUIRE((size_t)nc * 8 <= 128 * 1024, IPD_E_LIMIT,
                    "spgemm: more than 16384 columns (LDS accumulator row limit)");
        const size_t dense_elems = (size_t)nr * (size_t)(nc ? nc : 1);
        IPD_REQUIRE(dense_elems * 8 <= (size_t(2) << 30), IPD_E_LIMIT,
                    "spgemm: dense scratch above 2 GiB");
        dense = tmp.alloc<double>(dense_elems);
        rowbits = tmp.alloc<unsigned long long>((size_t)nr * (size_t)((nc + 4095) / 4096 + 1));
        const size_t lds = std::max<size_t>((size_t)nc * 8, 16);
        if (threads == 64) {
            IPD_OPTIN_LDS(ctx, k_spgemm_rows_w<8>, 128 * 1024);
            hipLaunchKernelGGL(k_spgemm_rows_w<8>, dim3(std::min(nr, 16384)), dim3(64), lds, ctx->stream, nr,
                               nc, X.rp, X.ci, X.va, Y.rp, Y.ci, Y.va, dense, rowcnt, rowbits);
        } else {
            IPD_OPTIN_LDS(ctx, k_spgemm_rows, 128 * 1024);
            hipLaunchKernelGGL(k_spgemm_rows, dim3(std::min(nr, 16384)), dim3(threads), lds,
                               ctx->stream, nr, nc, X.rp, X.ci, X.va, Y.rp, Y.ci, Y.va, dense,
                               rowcnt, rowbits, st);
        }
        IPD_KERNEL_CHECK();
    }
    const int* head = nullptr;   // plain counts the compaction scans itself
    if (head_ok) {   // lazy count: dense bound, no round trip
        out.nnz = (int)((size_t)nr * (size_t)nc);
        head = rowcnt;
    } else if (tail) {
        if (total_dev) {
            out.nnz = (int)((size_t)nr * (size_t)nc);
        } else {
            int two[2] = {0, 0};
            tt->wait(two);
            out.nnz = two[0];
        }
    } else if (total_dev) {
        out.nnz = (int)((size_t)nr * (size_t)nc);
        exclusive_scan_i32(ctx, rowcnt, out.rp, nr, total_dev);
    } else {
        out.nnz = exclusive_scan_total(ctx, rowcnt, out.rp, nr);
    }
    out.ci = dst.alloc<int>((size_t)out.nnz);
    out.va = dst.alloc<double>((size_t)out.nnz);
    if (out.nnz || head) {
        LazyPost lp;
        if (post && head && post->n <= 32 && ctx->mailbox_begin(&post->ticket)) {
            post->box = ctx->mailbox;
            lp = *post;
        }
        hipLaunchKernelGGL(k_dense_compact, dim3(std::max(1, std::min(cdiv(nr, 4), 4096))), dim3(256),
                           0, ctx->stream, nr, nc, ld, dense, (const unsigned long long*)rowbits,
                           (const int*)out.rp, out.ci, out.va, head, out.rp, total_dev, lp,
                           head ? maxrow_dev : (int*)nullptr);
        IPD_KERNEL_CHECK();
    }
    *C = out;
}

// dense-row helpers shared with the interpolation build (ipd_setup.hip)
void csr_expand_dense(ipd_ctx* ctx, const Csr& A, double* dense, int ld) {
    if (A.nr == 0 || A.nnz == 0) return;
    hipLaunchKernelGGL(k_csr_expand, dim3(std::min(cdiv(A.nr, 4), 4096)), dim3(256), 0, ctx->stream,
                       A.nr, ld, A.rp, A.ci, A.va, dense);
    IPD_KERNEL_CHECK();
}

void dense_rowcount(ipd_ctx* ctx, int nr, int nc, int ld, const double* dense, int* rowcnt, const ScanTail& st) {
    if (nr == 0) return;
    hipLaunchKernelGGL(k_dense_rowcount, dim3(std::min(cdiv(nr, 4), 4096)), dim3(256), 0,
                       ctx->stream, nr, nc, ld, dense, rowcnt, st);
    IPD_KERNEL_CHECK();
}

// dense rows -> the CSR arrays of `out` (row pointers from dense_rowcount + scan), exact zeros dropped
void dense_compact(ipd_ctx* ctx, int nr, int nc, int ld, const double* dense, const Csr& out) {
    if (nr == 0 || out.nnz == 0) return;
    hipLaunchKernelGGL(k_dense_compact, dim3(std::max(1, std::min(cdiv(nr, 4), 4096))), dim3(256), 0,
                       ctx->stream, nr, nc, ld, dense, (const unsigned long long*)nullptr, (const int*)out.rp,
                       out.ci, out.va, (const int*)nullptr, (int*)nullptr, (int*)nullptr, LazyPost(), (int*)nullptr);
    IPD_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// C ABI: device matrices
// ---------------------------------------------------------------------------
extern "C" int ipd_dmat_upload(ipd_ctx* ctx, const ipd_csc* A, int symmetric, ipd_dmat** out) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && out, IPD_E_ARG, "NULL argument");
        CallScope scope(ctx);
        std::unique_ptr<ipd_dmat> d(new ipd_dmat());
        d->ctx = ctx;
        d->arena.reset(new Arena(&ctx->pool));
        csr_upload_from_csc(ctx, *d->arena, A, symmetric != 0, &d->m);
        ctx->sync();
        *out = d.release();
    });
}

extern "C" int ipd_dmat_download(ipd_ctx* ctx, const ipd_dmat* A, ipd_csc_out* out) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && out, IPD_E_ARG, "NULL argument");
        CallScope scope(ctx);
        csr_download_as_csc(ctx, A->m, false, out);
    });
}

extern "C" int ipd_dmat_multiply(ipd_ctx* ctx, const ipd_dmat* A, const ipd_dmat* B,
                                 ipd_dmat** out) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && B && out, IPD_E_ARG, "NULL argument");
        CallScope scope(ctx);
        std::unique_ptr<ipd_dmat> d(new ipd_dmat());
        d->ctx = ctx;
        d->arena.reset(new Arena(&ctx->pool));
        csr_spgemm(ctx, *d->arena, A->m, B->m, &d->m);
        ctx->sync();
        *out = d.release();
    });
}

extern "C" int ipd_dmat_dims(const ipd_dmat* A, int64_t* rows, int64_t* cols, int64_t* nnz) {
    if (!A) return IPD_E_ARG;
    if (rows) *rows = A->m.nr;
    if (cols) *cols = A->m.nc;
    if (nnz) *nnz = A->m.nnz;
    return IPD_OK;
}

extern "C" void ipd_dmat_destroy(ipd_dmat* A) {
    if (!A) return;
    if (A->ctx) {
        (void)hipSetDevice(A->ctx->device);
        (void)hipStreamSynchronize(A->ctx->stream);
    }
    delete A;
}

extern "C" int ipd_spmv_dev(ipd_ctx* ctx, const ipd_dmat* A, const double* x, double* y) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && x && y, IPD_E_ARG, "NULL argument");
        CallScope scope(ctx);
        csr_spmv(ctx, A->m, x, y);
    });
}
