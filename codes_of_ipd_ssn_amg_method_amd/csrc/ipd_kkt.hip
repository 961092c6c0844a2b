// Matrix-free A operators and the active-set KKT assembly:
//   Ax.m:10-13, Aty.m:10-13, ASAt.m:14-19, invAAt.m:13-20, Class2/invHHt.m:7-17.
//
// Layout: x, s, phi are m*n column-major (X(i,j) = x[i + j*m]); the KKT unknowns
// are ordered [0,n) = column constraints, [n,n+m) = row constraints
// (Class1/APD_SsN_Class1.m:33).
//
// Roofline: all of these are single streaming passes -> HBM bound.
//   Ax   : 8*m*n + 16*(m+n) bytes      Aty : 8*m*n + 16*(m+n) bytes
//   ASAt : m*n (mask) + 8*(m+n) + 16*(2E+M) bytes
// This TU is compiled with -ffp-contract=off so that Aty and the ASAt values
// are bit-identical to the oracle (two multiplies and one add per entry).
#pragma clang fp contract(off)

#include "ipd_internal.h"

// ---------------------------------------------------------------------------
// Ax:  y = [X'*p ; X*q]
// ---------------------------------------------------------------------------
// A workgroup stages a 256-row x 16-column tile of X in LDS with coalesced
// column-contiguous loads (16 independent 8-byte loads in flight per lane), then
// forms 256 partial row sums (row i over the 16 columns) and 16 partial column
// sums (column j over the 256 rows) from the LDS copy.  Partials are combined
// by a second kernel in a fixed order (deterministic, no float atomics).
static constexpr int AX_TR = 256;       // tile rows
static constexpr int AX_TC = 16;        // tile columns
static constexpr int AX_LD = AX_TR + 16;  // padded column stride (bank spread)

__global__ __launch_bounds__(256) void k_ax_tiles(const double* __restrict__ x,
                                                  const double* __restrict__ p,
                                                  const double* __restrict__ q, int m, int n,
                                                  double* __restrict__ lpart,
                                                  double* __restrict__ rpart) {
    __shared__ double tile[AX_TC * AX_LD];
    const int tid = threadIdx.x;
    const int ib = blockIdx.x, jb = blockIdx.y;
    const int i = ib * AX_TR + tid;
    const int j0 = jb * AX_TC;
    const bool in_i = i < m;
    double xv[AX_TC];
#pragma unroll
    for (int jj = 0; jj < AX_TC; ++jj) {
        const int j = j0 + jj;
        xv[jj] = (in_i && j < n) ? x[(size_t)j * m + i] : 0.0;
    }
    double lacc = 0.0;
#pragma unroll
    for (int jj = 0; jj < AX_TC; ++jj) {
        const int j = j0 + jj;
        const double qj = j < n ? q[j] : 0.0;
        lacc += xv[jj] * qj;
        tile[jj * AX_LD + tid] = xv[jj];
    }
    if (in_i) lpart[(size_t)jb * m + i] = lacc;
    __syncthreads();
    // column sums: 16 lanes per column, each lane 16 rows, then a 16-lane reduce
    const int jj = tid >> 4, sub = tid & 15;
    double cacc = 0.0;
#pragma unroll
    for (int k = 0; k < AX_TR / 16; ++k) {
        const int r = sub + 16 * k;
        const int gi = ib * AX_TR + r;
        const double pi = gi < m ? p[gi] : 0.0;
        cacc += tile[jj * AX_LD + r] * pi;
    }
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) cacc += __shfl_xor(cacc, d);
    if (sub == 0 && j0 + jj < n) rpart[(size_t)ib * n + j0 + jj] = cacc;
}

// partial sums combined in ascending tile order; eight loads in flight per trip (a load per
// trip made this a chain of 64 L2 round trips: 11 us at m=n=1024 against 5 for the tiles)
__device__ __forceinline__ double ax_sum_strided(const double* __restrict__ part, int count,
                                                 size_t stride, int col) {
    double s = 0.0;
    for (int b0 = 0; b0 < count; b0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u;
            v[u] = part[(size_t)(b < count ? b : 0) * stride + col];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (b0 + u < count) s += v[u];
    }
    return s;
}

__global__ __launch_bounds__(256) void k_ax_final(const double* __restrict__ lpart,
                                                  const double* __restrict__ rpart, int m, int n,
                                                  int nib, int njb, double* __restrict__ y) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n)
        y[t] = ax_sum_strided(rpart, nib, (size_t)n, t);
    else if (t < n + m)
        y[t] = ax_sum_strided(lpart, njb, (size_t)m, t - n);
}

void kkt_ax(ipd_ctx* ctx, const double* x, const double* p, const double* q, int m, int n,
            double* y) {
    if (m <= 0 || n <= 0) return;
    Arena& tmp = *ctx->scratch;
    const int nib = cdiv(m, AX_TR), njb = cdiv(n, AX_TC);
    double* lpart = tmp.alloc<double>((size_t)njb * m);
    double* rpart = tmp.alloc<double>((size_t)nib * n);
    hipLaunchKernelGGL(k_ax_tiles, dim3(nib, njb), dim3(256), 0, ctx->stream, x, p, q, m, n, lpart,
                       rpart);
    IPD_KERNEL_CHECK();
    hipLaunchKernelGGL(k_ax_final, dim3(cdiv(m + n, 256)), dim3(256), 0, ctx->stream, lpart, rpart,
                       m, n, nib, njb, y);
    IPD_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// Aty:  z(i,j) = p_i*y1_j + y2_i*q_j        (pure streaming write)
// ---------------------------------------------------------------------------
static constexpr int ATY_TC = 16;

__global__ __launch_bounds__(256) void k_aty_v2(const double* __restrict__ y,
                                                const double* __restrict__ p,
                                                const double* __restrict__ q, int m, int n,
                                                double* __restrict__ z) {
    // two consecutive rows per lane -> 16-byte stores (m even, z 16-byte aligned)
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= m) return;
    const double p0 = p[i], p1 = p[i + 1];
    const double a0 = y[n + i], a1 = y[n + i + 1];
    const int j0 = blockIdx.y * ATY_TC;
#pragma unroll
    for (int jj = 0; jj < ATY_TC; ++jj) {
        const int j = j0 + jj;
        if (j < n) {
            const double y1 = y[j], qj = q[j];
            double2 v;
            v.x = p0 * y1 + a0 * qj;
            v.y = p1 * y1 + a1 * qj;
            *reinterpret_cast<double2*>(z + (size_t)j * m + i) = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_aty_v1(const double* __restrict__ y,
                                                const double* __restrict__ p,
                                                const double* __restrict__ q, int m, int n,
                                                double* __restrict__ z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double pi = p[i], ai = y[n + i];
    const int j0 = blockIdx.y * ATY_TC;
#pragma unroll
    for (int jj = 0; jj < ATY_TC; ++jj) {
        const int j = j0 + jj;
        if (j < n) z[(size_t)j * m + i] = pi * y[j] + ai * q[j];
    }
}

void kkt_aty(ipd_ctx* ctx, const double* y, const double* p, const double* q, int m, int n,
             double* z) {
    if (m <= 0 || n <= 0) return;
    const bool vec2 = (m % 2 == 0) && ((reinterpret_cast<uintptr_t>(z) & 15) == 0);
    if (vec2) {
        hipLaunchKernelGGL(k_aty_v2, dim3(cdiv(m / 2, 256), cdiv(n, ATY_TC)), dim3(256), 0,
                           ctx->stream, y, p, q, m, n, z);
    } else {
        hipLaunchKernelGGL(k_aty_v1, dim3(cdiv(m, 256), cdiv(n, ATY_TC)), dim3(256), 0, ctx->stream,
                           y, p, q, m, n, z);
    }
    IPD_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// ASAt:  H = A*diag(s)*A'  as CSR (== MATLAB's CSC, H is symmetric)
// ---------------------------------------------------------------------------
// Row j < n      : [ (j,j) , (j, n+i) for every i with Y(i,j)=1, ascending i ]
// Row n+i        : [ (n+i, j) for every j with Y(i,j)=1, ascending j , (n+i,n+i) ]
// A diagonal entry exists iff its row has any off-diagonal (MATLAB stores no
// explicit zeros; a p or q whose square underflows to 0 is handled by a
// zero-dropping pass).
//
// The byte mask is read once, as 64x64 tiles, one tile per wave: a wave ballot
// turns the 64 bytes of a column into a 64-bit row mask; 64 more ballots
// transpose the tile into per-row column masks.  Both bit-packed copies
// (2*m*n/8 bytes) are kept so the fill pass never touches the bytes again.
struct AsatPlan {
    int m, n, nib, njb;
    unsigned long long* colmask;  // [nib][n]  bits = rows of the tile
    unsigned long long* rowmask;  // [njb][m]  bits = columns of the tile
    int* coloff;                  // [nib][n]  entries of column j in tiles above
    int* rowoff;                  // [njb][m]
    int* cntc;                    // [n]
    int* cntr;                    // [m]
    int* rowlen;                  // [n+m]
};

// WIDE: m % 8 == 0 and s 8-byte aligned -- lane l reads the 64 bytes of column j0+l as eight
// 8-byte words and packs each to 8 bits (nonzero byte -> bit), so the lane's column mask needs
// no ballot; otherwise byte loads + one ballot per column.
__device__ __forceinline__ unsigned bytes_to_bits(unsigned long long x) {
    x |= x >> 4;
    x |= x >> 2;
    x |= x >> 1;
    x &= 0x0101010101010101ull;
    return (unsigned)((x * 0x0102040810204080ull) >> 56);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void k_asat_masks(const uint8_t* __restrict__ s, AsatPlan pl) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= pl.nib * pl.njb) return;
    const int ib = tile % pl.nib, jb = tile / pl.nib;
    const int i0 = ib * 64, j0 = jb * 64;
    const int i = i0 + lane;
    unsigned long long colm = 0ull;
    if (WIDE && i0 + 64 <= pl.m) {
        const int j = j0 + lane;
        const unsigned long long* src =
            reinterpret_cast<const unsigned long long*>(s + (size_t)(j < pl.n ? j : 0) * pl.m + i0);
        unsigned long long wd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wd[k] = src[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) colm |= (unsigned long long)bytes_to_bits(wd[k]) << (8 * k);
        if (j >= pl.n) colm = 0ull;
    } else {
#pragma unroll 8
        for (int jj = 0; jj < 64; ++jj) {
            const int j = j0 + jj;
            uint8_t b = 0;
            if (i < pl.m && j < pl.n) b = s[(size_t)j * pl.m + i];
            const unsigned long long mk = __ballot(b != 0);
            if (lane == jj) colm = mk;
        }
    }
    unsigned long long rowm = 0ull;
#pragma unroll 8
    for (int b = 0; b < 64; ++b) {
        const unsigned long long mk = __ballot((colm >> b) & 1ull);
        if (lane == b) rowm = mk;
    }
    if (j0 + lane < pl.n) pl.colmask[(size_t)ib * pl.n + j0 + lane] = colm;
    if (i < pl.m) pl.rowmask[(size_t)jb * pl.m + i] = rowm;
}

// per column / per row: running offsets over the tiles, counts, H row lengths,
// and the "some p_i^2 or q_j^2 is exactly zero" flag (meta[1]).
__global__ __launch_bounds__(256) void k_asat_offsets(AsatPlan pl, const double* __restrict__ p,
                                                      const double* __restrict__ q,
                                                      int* __restrict__ meta_flag) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < pl.n) {
        int run = 0;
        for (int ib = 0; ib < pl.nib; ++ib) {
            pl.coloff[(size_t)ib * pl.n + t] = run;
            run += __popcll(pl.colmask[(size_t)ib * pl.n + t]);
        }
        pl.cntc[t] = run;
        pl.rowlen[t] = run + (run > 0 ? 1 : 0);
        const double qq = q[t] * q[t];
        if (qq == 0.0) *meta_flag = 1;
    } else if (t < pl.n + pl.m) {
        const int i = t - pl.n;
        int run = 0;
        for (int jb = 0; jb < pl.njb; ++jb) {
            pl.rowoff[(size_t)jb * pl.m + i] = run;
            run += __popcll(pl.rowmask[(size_t)jb * pl.m + i]);
        }
        pl.cntr[i] = run;
        pl.rowlen[t] = run + (run > 0 ? 1 : 0);
        const double pp = p[i] * p[i];
        if (pp == 0.0) *meta_flag = 1;
    }
}

// Off-diagonal entries.  One wave per 64x64 tile; for every column (then every row) of the tile
// that has entries, lane b owns bit b of its mask word and writes its entry at the word's base
// position + the number of set bits below b: the 64 stores of a step are consecutive entries
// (one lane walking its own row wrote with a stride of a whole row: 50 us at rho = 1, m=n=1024,
// for 25 MB; this takes 9).
__device__ __forceinline__ unsigned long long wave_bcast64(unsigned long long v, int srclane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, srclane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), srclane);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double wave_bcast_f64(double v, int srclane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), srclane),
                            __builtin_amdgcn_readlane(__double2loint(v), srclane));
}

__global__ __launch_bounds__(256) void k_asat_fill(AsatPlan pl, const double* __restrict__ p,
                                                   const double* __restrict__ q,
                                                   const int* __restrict__ rp,
                                                   int* __restrict__ ci, double* __restrict__ va,
                                                   int cap /* entries allocated */) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= pl.nib * pl.njb) return;
    const int ib = tile % pl.nib, jb = tile / pl.nib;
    const int i0 = ib * 64, j0 = jb * 64;
    const int il = i0 + lane, jl = j0 + lane;
    const bool iok = il < pl.m, jok = jl < pl.n;
    const unsigned long long colm = jok ? pl.colmask[(size_t)ib * pl.n + jl] : 0ull;
    const unsigned long long rowm = iok ? pl.rowmask[(size_t)jb * pl.m + il] : 0ull;
    const int cpos = jok ? rp[jl] + 1 + pl.coloff[(size_t)ib * pl.n + jl] : 0;
    const int rpos = iok ? rp[pl.n + il] + pl.rowoff[(size_t)jb * pl.m + il] : 0;
    const double pv = iok ? p[il] : 0.0, qv = jok ? q[jl] : 0.0;
    const unsigned long long below = (1ull << lane) - 1ull;
    // rows [0,n): column j of Y -> entries (j, n+i), ascending i
    for (unsigned long long nz = __ballot(colm != 0ull); nz; nz &= nz - 1ull) {
        const int jj = __ffsll((long long)nz) - 1;
        const unsigned long long cm = wave_bcast64(colm, jj);
        const int pos0 = __builtin_amdgcn_readlane(cpos, jj);
        const double qj = wave_bcast_f64(qv, jj);
        if ((cm >> lane) & 1ull) {
            const int pos = pos0 + __popcll(cm & below);
            if (pos < cap) {
                ci[pos] = pl.n + il;
                va[pos] = qj * pv;   // q_j * (p_i * Y_ij)
            }
        }
    }
    // rows [n,n+m): row i of Y -> entries (n+i, j), ascending j
    for (unsigned long long nz = __ballot(rowm != 0ull); nz; nz &= nz - 1ull) {
        const int ii = __ffsll((long long)nz) - 1;
        const unsigned long long rm = wave_bcast64(rowm, ii);
        const int pos0 = __builtin_amdgcn_readlane(rpos, ii);
        const double pi = wave_bcast_f64(pv, ii);
        if ((rm >> lane) & 1ull) {
            const int pos = pos0 + __popcll(rm & below);
            if (pos < cap) {
                ci[pos] = jl;
                va[pos] = pi * qv;   // p_i * (Y_ij * q_j)
            }
        }
    }
}

// Diagonal values, accumulated sequentially in ascending index order exactly as the sparse
// products U'*p and Q*q of ASAt.m:19 do.  The order of the ADDS is what the bits fix, not the
// order of the loads.  A lane owns one H row and walks the row's words of the bit mask (coalesced
// across the 64 rows of the wave; not the column indices the fill pass wrote, so no dependent
// gather); per tile, lane l squares entry l of p (or q) -- the next tile's already in flight --
// and the wave loops, with a scalar bit scan, over the bit positions ANY of its rows has: the
// square comes out of lane b's register (v_readlane), a row without the bit adds +0.0, which
// leaves its sum as it is.  Nothing but the mask words and the vector is read.
// (One lane per row walking ci[e] -> p[ci[e]] in memory took 439 us at rho = 1, m=n=1024.)
// Blocks [0, nbc) are the rows [0,n) (column sums, vector p), the others the rows [n,n+m).
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(64) void k_asat_diag(AsatPlan pl, const double* __restrict__ p,
                                                  const double* __restrict__ q, int nbc,
                                                  const int* __restrict__ rp, int* __restrict__ ci,
                                                  double* __restrict__ va, int cap) {
    const int lane = threadIdx.x;
    const int side = blockIdx.x >= nbc ? 1 : 0;                 // uniform
    const int nrow = side ? pl.m : pl.n, len = side ? pl.n : pl.m;
    const int nt = side ? pl.njb : pl.nib;
    const double* __restrict__ vec = side ? q : p;
    const unsigned long long* __restrict__ masks = side ? pl.rowmask : pl.colmask;
    const int idx = (blockIdx.x - (side ? nbc : 0)) * 64 + lane;
    const bool live = idx < nrow;
    const int ic = live ? idx : 0;
    const int cnt = live ? (side ? pl.cntr[ic] : pl.cntc[ic]) : 0;
    double sum = 0.0;
    constexpr int CH = 8;    // tiles whose mask words and vector entries are requested together
    for (int t0 = 0; t0 < nt; t0 += CH) {
        unsigned long long wb[CH];
        double vb[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int t = t0 + k;
            const bool tok = t < nt;
            const unsigned long long wv = masks[(size_t)(tok ? t : 0) * nrow + ic];
            const int e = t * 64 + lane;
            const double vv = vec[(tok && e < len) ? e : 0];
            wb[k] = (tok && live) ? wv : 0ull;
            vb[k] = vv;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const unsigned long long w = wb[k];
            const double sq = vb[k] * vb[k];          // (p_i*Y_ij) * p_i, (Y_ij*q_j) * q_j
            const int slo = __double2loint(sq), shi = __double2hiint(sq);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const unsigned wh = half ? (unsigned)(w >> 32) : (unsigned)w;
                unsigned u = wave_or_u32(wh);         // scalar: positions any row of the wave has
                if (u == 0u) continue;
                if (u == 0xFFFFFFFFu) {
                    // every position present somewhere (dense masks): fixed lanes and bit numbers,
                    // no scalar bit scan; a row without the bit adds +0.0 (mask 0 / ~0 from the bit)
#pragma unroll
                    for (int bpos = 0; bpos < 32; ++bpos) {
                        const int m = __builtin_amdgcn_sbfe(wh, bpos, 1);
                        const int xl = __builtin_amdgcn_readlane(slo, half * 32 + bpos) & m;
                        const int xh = __builtin_amdgcn_readlane(shi, half * 32 + bpos) & m;
                        sum = sum + __hiloint2double(xh, xl);
                    }
                    continue;
                }
                if (__popc(u) <= 6) {
                    // few positions (the realistic, tree-like masks): every lane takes its own bits,
                    // the squares come from the owning lanes' registers (ds_bpermute, no memory)
                    unsigned mine = wh;
                    while (__any(mine != 0u)) {
                        const bool has = mine != 0u;
                        const int bpos = has ? __builtin_ctz(mine) : 0;
                        mine &= mine - 1u;            // 0 stays 0
                        const double y = __shfl(sq, half * 32 + bpos);
                        sum = sum + (has ? y : 0.0);
                    }
                    continue;
                }
                while (u) {
                    const int bpos = __builtin_ctz(u);
                    u &= u - 1u;
                    const int m = __builtin_amdgcn_sbfe(wh, bpos, 1);
                    const int xl = __builtin_amdgcn_readlane(slo, half * 32 + bpos) & m;
                    const int xh = __builtin_amdgcn_readlane(shi, half * 32 + bpos) & m;
                    sum = sum + __hiloint2double(xh, xl);
                }
            }
        }
    }
    if (cnt > 0) {
        const int row = side ? pl.n + idx : idx;
        const int pos = rp[row] + (side ? cnt : 0);
        if (pos < cap) {
            ci[pos] = row;
            va[pos] = sum;
        }
    }
}

// generic "drop exact zeros" (rare path: squares that underflow)
__global__ __launch_bounds__(256) void k_count_nz(int nr, const int* __restrict__ rp,
                                                  const double* __restrict__ va,
                                                  int* __restrict__ cnt) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nr) return;
    int c = 0;
    for (int e = rp[r]; e < rp[r + 1]; ++e) c += (va[e] != 0.0);
    cnt[r] = c;
}
__global__ __launch_bounds__(256) void k_copy_nz(int nr, const int* __restrict__ rp,
                                                 const int* __restrict__ ci,
                                                 const double* __restrict__ va,
                                                 const int* __restrict__ orp, int* __restrict__ oci,
                                                 double* __restrict__ ova) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nr) return;
    int pos = orp[r];
    for (int e = rp[r]; e < rp[r + 1]; ++e)
        if (va[e] != 0.0) {
            oci[pos] = ci[e];
            ova[pos] = va[e];
            ++pos;
        }
}

void csr_drop_zeros(ipd_ctx* ctx, Arena& dst, const Csr& A, Csr* out) {
    Arena& tmp = *ctx->scratch;
    int* cnt = tmp.alloc<int>((size_t)A.nr + 1);
    Csr o;
    o.nr = A.nr;
    o.nc = A.nc;
    o.rp = dst.alloc<int>((size_t)A.nr + 1);
    hipLaunchKernelGGL(k_count_nz, dim3(cdiv(std::max(A.nr, 1), 256)), dim3(256), 0, ctx->stream,
                       A.nr, A.rp, A.va, cnt);
    IPD_KERNEL_CHECK();
    o.nnz = exclusive_scan_total(ctx, cnt, o.rp, A.nr);
    o.ci = dst.alloc<int>((size_t)o.nnz);
    o.va = dst.alloc<double>((size_t)o.nnz);
    hipLaunchKernelGGL(k_copy_nz, dim3(cdiv(std::max(A.nr, 1), 256)), dim3(256), 0, ctx->stream,
                       A.nr, A.rp, A.ci, A.va, o.rp, o.ci, o.va);
    IPD_KERNEL_CHECK();
    *out = o;
}


// Everything after the masks in ONE launch, for small active sets (the drivers' Newton systems:
// E ~ m + n, tree-like): a thread owns a row of H, counts its mask bits, the row pointers come from
// a scan inside the workgroup, and the thread then writes its row -- diagonal, off-diagonal entries
// in ascending column order, the diagonal value summed in ascending index order as ASAt.m:19's
// products do -- and the entry count goes to the host mailbox with the kernel's last store.  The
// general path above is six dependent launches and a read-back, 56-79 us whatever E is (launch-
// bound); this one is k_asat_masks + this kernel.  Same values, same order, same bits.
struct AsatSmallArgs {
    AsatPlan pl;
    const double* p;
    const double* q;
    int* rp;          // M + 2: rp[M] = nnz, rp[M+1] = zero-square flag
    int* ci;
    double* va;
    int cap;
    unsigned long long* agg;   // per workgroup: ticket << 32 | entries of its rows
    volatile unsigned* box;
    unsigned ticket;
};
// G = ceil(M / 256) workgroups of 256 rows; the row pointers come from a chained scan with look-back:
// a workgroup publishes the entry count of its rows, tagged with the call's ticket (so the words
// need no clearing between calls), as soon as it has it, reads the tagged counts of ALL its
// predecessors in one sweep (lane = predecessor, at most 64 of them) and repeats the sweep until every
// tag is there.  All workgroups are on the chip together (G <= 64 of 256 CUs); the spin is bounded
// anyway, and a give-up is reported as an impossible count, which sends the host to the general path.
__global__ __launch_bounds__(256) void k_asat_small(const AsatSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) double asat_pq[];   // p (m), then q (n)
    __shared__ int wsum[4];
    __shared__ int s_flag, s_prefix, s_zpred;
    const AsatPlan& pl = a.pl;
    const int n = pl.n, m = pl.m, M = n + m, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = blockIdx.x, G = gridDim.x;
    const int t = g * 256 + tid;
    const bool frow = t < n, live = t < M;
    const int r = frow ? t : t - n;                               // column j of Y, or row i
    const int nw = frow ? pl.nib : pl.njb, stride = frow ? n : m;
    const unsigned long long* __restrict__ mk = frow ? pl.colmask : pl.rowmask;
    // one burst: the row's mask words, then p and q into LDS (the gathers of the fill)
    unsigned long long w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = (live && k < nw) ? mk[(size_t)k * stride + r] : 0ull;
    double* sp = asat_pq;
    double* sq = asat_pq + m;
    for (int i = tid; i < m; i += 256) sp[i] = a.p[i];
    for (int j = tid; j < n; j += 256) sq[j] = a.q[j];
    if (tid == 0) s_zpred = 0;
    int run = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) run += __popcll(w[k]);
    const int len = run + (run > 0 ? 1 : 0);
    int x = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sgm = wsum[k];
        if (k < wv) woff += sgm;
        tot += sgm;
    }
    // publish this workgroup's count (bit 31: one of its rows has a zero square), then sum the predecessors'
    unsigned long long* agg = a.agg;
    {
        const double own0 = live ? (frow ? a.q[r] : a.p[r]) : 1.0;
        const int zs = __syncthreads_or(own0 * own0 == 0.0 ? 1 : 0);
        if (tid == 0) {
            s_flag = zs;
            __hip_atomic_store(&agg[g], ((unsigned long long)a.ticket << 32) | (unsigned)tot | (zs ? 0x80000000u : 0u),
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (wv == 0) {
        int before = 0;
        bool ok = false;
        for (int spin = 0; spin < (1 << 20) && !ok; ++spin) {
            const unsigned long long v = lane < g ? __hip_atomic_load(&agg[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                                  : ((unsigned long long)a.ticket << 32);
            const bool mine = (unsigned)(v >> 32) == a.ticket;
            ok = __all(mine);
            if (ok) {
                int c = lane < g ? (int)((unsigned)v & 0x7fffffffu) : 0;
                const int zb = __any(lane < g && ((unsigned)v >> 31)) ? 1 : 0;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
                before = c;
                if (lane == 0) s_zpred = zb;
            } else {
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (lane == 0) s_prefix = ok ? before : -1;
    }
    __syncthreads();
    const int carry = s_prefix;
    const bool gaveup = carry < 0;
    const int beg = (gaveup ? 0 : carry) + woff + x - len;
    if (live && !gaveup) a.rp[t] = beg;
    if (live && len > 0 && !gaveup) {
        // rows [0,n): diagonal first, then (j, n+i) ascending i;  rows [n,n+m): (n+i, j) ascending j,
        // diagonal last.  The diagonal value is summed in ascending index order (ASAt.m:19).
        const double own = frow ? sq[r] : sp[r];
        const double* other = frow ? sp : sq;
        int pos = frow ? beg + 1 : beg;
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            unsigned long long ww = w[k];
            while (ww) {
                const int o = k * 64 + __builtin_ctzll(ww);
                ww &= ww - 1ull;
                const double ov = other[o];
                sum = sum + ov * ov;                               // (p_i*Y_ij)*p_i  /  (Y_ij*q_j)*q_j
                if (pos < a.cap) {
                    a.ci[pos] = frow ? n + o : o;
                    a.va[pos] = own * ov;                          // q_j*(p_i*Y_ij)  /  p_i*(Y_ij*q_j)
                }
                ++pos;
            }
        }
        const int dpos = frow ? beg : pos;
        if (dpos < a.cap) {
            a.ci[dpos] = t;
            a.va[dpos] = sum;
        }
    }
    if (g == G - 1) {   // the last workgroup knows the total and every workgroup's zero-square bit
        __syncthreads();
        if (tid == 0) {
            const int total = gaveup ? 0x7fffffff : carry + tot;
            const int zs = (s_flag || s_zpred) ? 1 : 0;
            a.rp[M] = total;
            a.rp[M + 1] = zs;
            a.box[16] = (unsigned)total;
            a.box[17] = (unsigned)zs;
            __threadfence_system();
            a.box[0] = a.ticket;
        }
    }
}

void kkt_asat(ipd_ctx* ctx, Arena& dst, const uint8_t* s, const double* p, const double* q, int m,
              int n, Csr* H) {
    IPD_REQUIRE(m > 0 && n > 0, IPD_E_ARG, "ASAt: empty p or q");
    // H has at most 2*m*n + m + n entries and its row pointers, entry counts and tile counts are
    // 32-bit: refuse what could wrap (e.g. m = n = 40000 at rho ~ 1 fits in HBM but not in int32)
    IPD_REQUIRE(2LL * m * n + m + n < (1LL << 31) - 64, IPD_E_LIMIT,
                "ASAt: 2*m*n + m + n must stay below 2^31 (32-bit row pointers)");
    Arena& tmp = *ctx->scratch;
    AsatPlan pl;
    pl.m = m;
    pl.n = n;
    pl.nib = cdiv(m, 64);
    pl.njb = cdiv(n, 64);
    pl.colmask = tmp.alloc<unsigned long long>((size_t)pl.nib * n);
    pl.rowmask = tmp.alloc<unsigned long long>((size_t)pl.njb * m);
    pl.coloff = tmp.alloc<int>((size_t)pl.nib * n);
    pl.rowoff = tmp.alloc<int>((size_t)pl.njb * m);
    pl.cntc = tmp.alloc<int>((size_t)n);
    pl.cntr = tmp.alloc<int>((size_t)m);
    pl.rowlen = tmp.alloc<int>((size_t)m + n);
    const int M = m + n;
    int* rp = dst.alloc<int>((size_t)M + 2);  // rp[M] = nnz, rp[M+1] = zero-square flag
    const int ntiles = pl.nib * pl.njb;
    // small active sets (the previous call's count says so): masks + ONE more launch
    unsigned sticket = 0;
    const bool small = ctx->asat_nnz_hint > 0 && ctx->asat_nnz_hint <= 65536 && pl.nib <= 16 && pl.njb <= 16 &&
                       cdiv(M, 256) <= 64 &&
                       ctx->mailbox_begin(&sticket);
    if (!small) IPD_HIP(hipMemsetAsync(rp + M + 1, 0, sizeof(int), ctx->stream));
    if (m % 8 == 0 && (reinterpret_cast<uintptr_t>(s) & 7) == 0)
        hipLaunchKernelGGL(k_asat_masks<true>, dim3(cdiv(ntiles, 4)), dim3(256), 0, ctx->stream, s, pl);
    else
        hipLaunchKernelGGL(k_asat_masks<false>, dim3(cdiv(ntiles, 4)), dim3(256), 0, ctx->stream, s, pl);
    IPD_KERNEL_CHECK();
    if (small) {
        const long long cap = std::min<long long>(2LL * m * n + M, ctx->asat_nnz_hint + ctx->asat_nnz_hint / 4 + 64);
        Csr hs;
        hs.nr = hs.nc = M;
        hs.rp = rp;
        hs.ci = dst.alloc<int>((size_t)cap);
        hs.va = dst.alloc<double>((size_t)cap);
        AsatSmallArgs sa;
        sa.pl = pl;
        sa.p = p;
        sa.q = q;
        sa.rp = rp;
        sa.ci = hs.ci;
        sa.va = hs.va;
        sa.cap = (int)cap;
        sa.box = ctx->mailbox;
        sa.ticket = sticket;
        if (!ctx->asat_agg) {   // 64 tagged words, kept for the context's lifetime
            IPD_HIP(hipMalloc(&ctx->asat_agg, 64 * sizeof(unsigned long long)));
            IPD_HIP(hipMemsetAsync(ctx->asat_agg, 0, 64 * sizeof(unsigned long long), ctx->stream));
        }
        sa.agg = static_cast<unsigned long long*>(ctx->asat_agg);
        hipLaunchKernelGGL(k_asat_small, dim3(cdiv(M, 256)), dim3(256), sizeof(double) * (size_t)M, ctx->stream, sa);
        IPD_KERNEL_CHECK();
        unsigned w2[2] = {0, 0};
        ctx->mailbox_wait(sticket, w2, sizeof(w2));
        hs.nnz = (int)w2[0];
        if (hs.nnz != 0x7fffffff) ctx->asat_nnz_hint = hs.nnz;
        if ((long long)hs.nnz <= cap) {
            const bool zsq = w2[1] != 0;
            if (zsq && hs.nnz) {
                Csr clean;
                csr_drop_zeros(ctx, dst, hs, &clean);
                hs = clean;
            }
            *H = hs;
            return;
        }
        // the active set grew beyond the guess (or a spin gave up): the general path redoes the assembly
        IPD_HIP(hipMemsetAsync(rp + M + 1, 0, sizeof(int), ctx->stream));
    }
    hipLaunchKernelGGL(k_asat_offsets, dim3(cdiv(M, 256)), dim3(256), 0, ctx->stream, pl, p, q,
                       rp + M + 1);
    IPD_KERNEL_CHECK();
    exclusive_scan_i32(ctx, pl.rowlen, rp, M);
    // The entry count sizes ci/va, and reading it back is a host round trip in the middle of the
    // assembly.  Consecutive Newton steps have similar active sets, so the arrays are sized from
    // the previous call's count (+25 %) and the fill runs at once; the count is read afterwards,
    // and only when it exceeds the guess (the kernels never write past `cap`) the fill is redone.
    int meta[2] = {0, 0};
    Csr h;
    h.nr = h.nc = M;
    h.rp = rp;
    auto fill = [&](int cap) {
        hipLaunchKernelGGL(k_asat_fill, dim3(cdiv(ntiles, 4)), dim3(256), 0, ctx->stream, pl, p, q,
                           rp, h.ci, h.va, cap);
        IPD_KERNEL_CHECK();
        // 64-thread blocks: the sums are one dependent chain per row, so rows go to as many CUs
        // as there are (a 256-thread block per 256 rows left 248 CUs idle at m = n = 1024)
        // one wave per 64 rows (the sums are one dependent chain per row), both sides in one launch
        hipLaunchKernelGGL(k_asat_diag, dim3(cdiv(n, 64) + cdiv(m, 64)), dim3(64), 0, ctx->stream, pl, p, q,
                           cdiv(n, 64), rp, h.ci, h.va, cap);
        IPD_KERNEL_CHECK();
    };
    const long long worst = 2LL * m * n + M;
    long long guess = ctx->asat_nnz_hint > 0 ? std::min(worst, ctx->asat_nnz_hint + ctx->asat_nnz_hint / 4 + 64) : 0;
    if (guess > 0) {
        h.ci = dst.alloc<int>((size_t)guess);
        h.va = dst.alloc<double>((size_t)guess);
        fill((int)guess);
        ctx->fetch(rp + M, meta, 2);
        h.nnz = meta[0];
    }
    if (guess == 0 || meta[0] > guess) {
        if (guess == 0) ctx->fetch(rp + M, meta, 2);
        h.nnz = meta[0];
        h.ci = dst.alloc<int>((size_t)h.nnz);
        h.va = dst.alloc<double>((size_t)h.nnz);
        if (h.nnz) fill(h.nnz);
    }
    ctx->asat_nnz_hint = h.nnz;
    if (meta[1] && h.nnz) {
        Csr clean;
        csr_drop_zeros(ctx, dst, h, &clean);
        h = clean;
    }
    *H = h;
}

// ---------------------------------------------------------------------------
// invAAt / invHHt  (closed forms; O(m+n) plus one Ax for invHHt)
// ---------------------------------------------------------------------------
__device__ inline double block_sum_1024(double v, double* red) {
    // all 1024 threads participate; returns the sum in every thread
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k];
    return s;
}

// y = (diag(sg1 I_n, sg2 I_m) + A A')^{-1} x        invAAt.m:13-20
__global__ __launch_bounds__(1024) void k_inv_aat(const double* __restrict__ x,
                                                  const double* __restrict__ p,
                                                  const double* __restrict__ q, int m, int n,
                                                  double sg1, double sg2, double* __restrict__ y) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double a = 0, b = 0, c = 0, d = 0;
    for (int i = tid; i < m; i += 1024) {
        a += p[i] * p[i];
        c += p[i] * x[n + i];
    }
    for (int j = tid; j < n; j += 1024) {
        b += q[j] * q[j];
        d += q[j] * x[j];
    }
    const double np = block_sum_1024(a, red);
    const double nq = block_sum_1024(b, red);
    const double pvm = block_sum_1024(c, red);
    const double qvn = block_sum_1024(d, red);
    const double den = sg1 * sg2 + sg1 * nq + sg2 * np;
    const double cn = (np / (sg1 + np) * qvn - pvm);
    const double cm = (nq / (sg2 + nq) * pvm - qvn);
    for (int j = tid; j < n; j += 1024) y[j] = x[j] / (sg1 + np) + cn * q[j] / den;
    for (int i = tid; i < m; i += 1024) y[n + i] = x[n + i] / (sg2 + nq) + cm * p[i] / den;
}

void kkt_inv_aat(ipd_ctx* ctx, const double* x, const double* p, const double* q, int m, int n,
                 double sg1, double sg2, double* y) {
    hipLaunchKernelGGL(k_inv_aat, dim3(1), dim3(1024), 0, ctx->stream, x, p, q, m, n, sg1, sg2, y);
    IPD_KERNEL_CHECK();
}

__global__ __launch_bounds__(256) void k_sumsq_partial(const double* __restrict__ v, size_t len,
                                                       double* __restrict__ part) {
    __shared__ double red[4];
    double a = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < len; i += (size_t)gridDim.x * 256)
        a += v[i] * v[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) a += __shfl_xor(a, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Class2/invHHt.m:8-17 once l = Ax(phi), Vl = invAAt(l,sg+1), Vv1 = invAAt(v1,sg+1)
__global__ __launch_bounds__(1024) void k_inv_hht_combine(const double* __restrict__ v,
                                                          const double* __restrict__ l,
                                                          const double* __restrict__ Vl,
                                                          const double* __restrict__ Vv1,
                                                          const double* __restrict__ part,
                                                          int npart, int M, double sg,
                                                          double* __restrict__ y) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double a = 0, b = 0, c = 0;
    for (int k = tid; k < npart; k += 1024) a += part[k];
    for (int k = tid; k < M; k += 1024) {
        b += l[k] * Vl[k];
        c += l[k] * Vv1[k];
    }
    const double t = sg + block_sum_1024(a, red);
    const double lVl = block_sum_1024(b, red);
    const double lVv = block_sum_1024(c, red);
    const double s = t - lVl;
    const double v2 = v[M];
    for (int k = tid; k < M; k += 1024) y[k] = (s * Vv1[k] + lVv * Vl[k] - v2 * Vl[k]) / s;
    if (tid == 0) y[M] = (v2 - lVv) / s;
}

// invHHt with l = Ax(phi) and the partial sums of ||phi||^2 hoisted out: both depend on the
// problem only, while the warm start calls invHHt once per iteration (warmup_class2.m:72)
void kkt_inv_hht_pre(ipd_ctx* ctx, const double* v, const double* p, const double* q, int m, int n,
                     double sg, const double* l, const double* part, int npart, double* y) {
    Arena& tmp = *ctx->scratch;
    const int M = m + n;
    double* Vl = tmp.alloc<double>((size_t)M);
    double* Vv1 = tmp.alloc<double>((size_t)M);
    kkt_inv_aat(ctx, l, p, q, m, n, sg + 1, sg + 1, Vl);
    kkt_inv_aat(ctx, v, p, q, m, n, sg + 1, sg + 1, Vv1);
    hipLaunchKernelGGL(k_inv_hht_combine, dim3(1), dim3(1024), 0, ctx->stream, v, l, Vl, Vv1, part,
                       npart, M, sg, y);
    IPD_KERNEL_CHECK();
}

int kkt_phi_consts(ipd_ctx* ctx, const double* phi, const double* p, const double* q, int m, int n,
                   double* l, double* part /* 1024 doubles */) {
    const size_t mn = (size_t)m * n;
    const int npart = (int)std::max<size_t>(1, std::min<size_t>((mn + 255) / 256, 1024));
    hipLaunchKernelGGL(k_sumsq_partial, dim3(npart), dim3(256), 0, ctx->stream, phi, mn, part);
    IPD_KERNEL_CHECK();
    kkt_ax(ctx, phi, p, q, m, n, l);
    return npart;
}

void kkt_inv_hht(ipd_ctx* ctx, const double* v, const double* p, const double* q, int m, int n,
                 double sg, const double* phi, double* y) {
    Arena& tmp = *ctx->scratch;
    const int M = m + n;
    const size_t mn = (size_t)m * n;
    const int npart = (int)std::max<size_t>(1, std::min<size_t>((mn + 255) / 256, 1024));
    double* part = tmp.alloc<double>((size_t)npart);
    double* l = tmp.alloc<double>((size_t)M);
    double* Vl = tmp.alloc<double>((size_t)M);
    double* Vv1 = tmp.alloc<double>((size_t)M);
    hipLaunchKernelGGL(k_sumsq_partial, dim3(npart), dim3(256), 0, ctx->stream, phi, mn, part);
    IPD_KERNEL_CHECK();
    kkt_ax(ctx, phi, p, q, m, n, l);
    kkt_inv_aat(ctx, l, p, q, m, n, sg + 1, sg + 1, Vl);
    kkt_inv_aat(ctx, v, p, q, m, n, sg + 1, sg + 1, Vv1);
    hipLaunchKernelGGL(k_inv_hht_combine, dim3(1), dim3(1024), 0, ctx->stream, v, l, Vl, Vv1, part,
                       npart, M, sg, y);
    IPD_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static void check_mn(int64_t m, int64_t n) {
    IPD_REQUIRE(m > 0 && n > 0, IPD_E_ARG, "m and n must be positive");
    IPD_REQUIRE(m < (1 << 28) && n < (1 << 28) && m * n < (int64_t(1) << 40), IPD_E_LIMIT,
                "m*n too large");
}

extern "C" int ipd_ax_dev(ipd_ctx* ctx, const double* x, const double* p, const double* q,
                          int64_t m, int64_t n, double* y) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && x && p && q && y, IPD_E_ARG, "NULL argument");
        check_mn(m, n);
        CallScope scope(ctx);
        kkt_ax(ctx, x, p, q, (int)m, (int)n, y);
    });
}

extern "C" int ipd_aty_dev(ipd_ctx* ctx, const double* y, const double* p, const double* q,
                           int64_t m, int64_t n, double* z) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && z && p && q && y, IPD_E_ARG, "NULL argument");
        check_mn(m, n);
        CallScope scope(ctx);
        kkt_aty(ctx, y, p, q, (int)m, (int)n, z);
    });
}

extern "C" int ipd_asat_dev(ipd_ctx* ctx, const uint8_t* s, const double* p, const double* q,
                            int64_t m, int64_t n, ipd_dmat** H) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && s && p && q && H, IPD_E_ARG, "NULL argument");
        check_mn(m, n);
        CallScope scope(ctx);
        std::unique_ptr<ipd_dmat> d(new ipd_dmat());
        d->ctx = ctx;
        d->arena.reset(new Arena(&ctx->pool));
        kkt_asat(ctx, *d->arena, s, p, q, (int)m, (int)n, &d->m);
        *H = d.release();
    });
}

namespace {
struct HostVecs {
    ipd_ctx* ctx;
    Arena& a;
    explicit HostVecs(ipd_ctx* c) : ctx(c), a(*c->scratch) {}
    template <class T>
    T* up(const T* h, size_t n) {
        T* d = a.alloc<T>(n);
        ctx->upload(d, h, n);
        return d;
    }
};
}  // namespace

extern "C" int ipd_ax(ipd_ctx* ctx, const double* x, const double* p, const double* q, int64_t m,
                      int64_t n, double* y) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && x && p && q && y, IPD_E_ARG, "NULL argument");
        check_mn(m, n);
        CallScope scope(ctx);
        HostVecs hv(ctx);
        double* dx = hv.up(x, (size_t)(m * n));
        double* dp = hv.up(p, (size_t)m);
        double* dq = hv.up(q, (size_t)n);
        double* dy = ctx->scratch->alloc<double>((size_t)(m + n));
        kkt_ax(ctx, dx, dp, dq, (int)m, (int)n, dy);
        ctx->fetch(dy, y, (size_t)(m + n));
    });
}

extern "C" int ipd_aty(ipd_ctx* ctx, const double* y, const double* p, const double* q, int64_t m,
                       int64_t n, double* z) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && z && p && q && y, IPD_E_ARG, "NULL argument");
        check_mn(m, n);
        CallScope scope(ctx);
        HostVecs hv(ctx);
        double* dy = hv.up(y, (size_t)(m + n));
        double* dp = hv.up(p, (size_t)m);
        double* dq = hv.up(q, (size_t)n);
        double* dz = ctx->scratch->alloc<double>((size_t)(m * n));
        kkt_aty(ctx, dy, dp, dq, (int)m, (int)n, dz);
        ctx->fetch(dz, z, (size_t)(m * n));
    });
}

extern "C" int ipd_asat(ipd_ctx* ctx, const uint8_t* s, const double* p, const double* q,
                        int64_t m, int64_t n, ipd_csc_out* H) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && s && p && q && H, IPD_E_ARG, "NULL argument");
        check_mn(m, n);
        CallScope scope(ctx);
        HostVecs hv(ctx);
        uint8_t* ds = hv.up(s, (size_t)(m * n));
        double* dp = hv.up(p, (size_t)m);
        double* dq = hv.up(q, (size_t)n);
        Csr h;
        kkt_asat(ctx, *ctx->scratch, ds, dp, dq, (int)m, (int)n, &h);
        csr_download_as_csc(ctx, h, /*H symmetric: CSR == CSC*/ true, H);
    });
}

extern "C" int ipd_inv_aat(ipd_ctx* ctx, const double* x, const double* p, const double* q,
                           int64_t m, int64_t n, double sg1, double sg2, double* y) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && x && p && q && y, IPD_E_ARG, "NULL argument");
        check_mn(m, n);
        CallScope scope(ctx);
        HostVecs hv(ctx);
        double* dx = hv.up(x, (size_t)(m + n));
        double* dp = hv.up(p, (size_t)m);
        double* dq = hv.up(q, (size_t)n);
        double* dy = ctx->scratch->alloc<double>((size_t)(m + n));
        kkt_inv_aat(ctx, dx, dp, dq, (int)m, (int)n, sg1, sg2, dy);
        ctx->fetch(dy, y, (size_t)(m + n));
    });
}

extern "C" int ipd_inv_hht(ipd_ctx* ctx, const double* v, const double* p, const double* q,
                           int64_t m, int64_t n, double sg, const double* phi, double* y) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && v && p && q && phi && y, IPD_E_ARG, "NULL argument");
        check_mn(m, n);
        CallScope scope(ctx);
        HostVecs hv(ctx);
        double* dv = hv.up(v, (size_t)(m + n + 1));
        double* dp = hv.up(p, (size_t)m);
        double* dq = hv.up(q, (size_t)n);
        double* dphi = hv.up(phi, (size_t)(m * n));
        double* dy = ctx->scratch->alloc<double>((size_t)(m + n + 1));
        kkt_inv_hht(ctx, dv, dp, dq, (int)m, (int)n, sg, dphi, dy);
        ctx->fetch(dy, y, (size_t)(m + n + 1));
    });
}
