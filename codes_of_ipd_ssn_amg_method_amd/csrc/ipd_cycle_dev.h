// Device-side building blocks shared by ipd_cycle.hip and the micro-benchmarks in tools/:
// the level descriptor and the DPP-based reductions.
#pragma once

#include <hip/hip_runtime.h>

// threads per block of every phase kernel and of the single-workgroup kernels (their helpers
// share the block reductions, so it is one constant).  Measured, 1024 -> 512: regime-D V cycle
// 0.194 -> 0.190 ms, tree-mask W cycle 0.503 -> 0.431 ms, m=n=1024 Class 1 / Class 2 driver runs
// 1.53 / 0.74 -> 1.47 / 0.70 s, m=n=4096 Class 1 5.33 -> 4.97 s (only the bandwidth-bound
// m=n=2048 regime-D cycle loses: 0.328 -> 0.342 ms); 256: 0.234 ms and the sub-cycle kernel no
// longer takes 300-1000-row roots (Class 1 run 20 s).
static constexpr int BT = 512;

// ---------------------------------------------------------------------------
// device-side level descriptor
// ---------------------------------------------------------------------------
struct LevelDev {
    int N, nf, L, G;  // rows, F-block size (0 = Jacobi), lanes/row, blocks per launch
    // CSR (always present)
    const int* rp;
    const int* ci;
    const double* va;
    // padded copy of the off-diagonal part (S > 0): row r occupies [r*S, (r+1)*S),
    // 16-bit columns, padding entries have value 0; the diagonal lives in `diag`
    int S;
    const unsigned short* pci;
    const double* pva;
    const double* diag;
    const double* dinv;
    const double* Axi;
    const double* xx;
    double* r;
    double* rr;
};

// ---------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------
// Cross-lane sums use DPP (ALU-rate row operations) instead of __shfl_xor: hipcc lowers
// a double shuffle to two ds_bpermute round trips through the LDS pipe (~150 cycles a
// step, 6 dependent steps per wave sum, measured 0.4 us), which dominated the
// reduction phase of these few-microsecond kernels.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

// after this every lane holds the sum of its aligned 16-lane row
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_get<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_get<0x140, 0xf>(v);  // row_mirror
    return v;
}

// sum over the 64 lanes of the wave, result in every lane
__device__ __forceinline__ double wave_sum(double v) {
    v = row16_sum(v);
    v += dpp_get<0x142, 0xa>(v);  // row_bcast15 -> rows 1,3
    v += dpp_get<0x143, 0xc>(v);  // row_bcast31 -> rows 2,3 ; lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// sum over aligned groups of L lanes, L in {1,2,4,8,16,32,64}; result in every lane of the group
__device__ __forceinline__ double subwave_sum(double v, int L) {
    if (L >= 2) v += dpp_get<0xB1, 0xf>(v);
    if (L >= 4) v += dpp_get<0x4E, 0xf>(v);
    if (L >= 8) v += dpp_get<0x141, 0xf>(v);
    if (L >= 16) v += dpp_get<0x140, 0xf>(v);
    if (L >= 32) v += __shfl_xor(v, 16);
    if (L >= 64) v += __shfl_xor(v, 32);
    return v;
}

// sum over the whole 1024-thread block, result in every thread
__device__ __forceinline__ double block_sum(double v, double* red /*16 doubles of LDS*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < BT / 64; ++k) s += red[k];
    return s;
}

// sum within aligned groups of L threads (L = 4..1024, power of two); every
// thread of the block must call it.  Result valid in the group's first thread.
__device__ __forceinline__ double group_sum(double v, int L, double* red /*16 doubles*/) {
    if (L <= 64) return subwave_sum(v, L);
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    const int wpg = L >> 6;  // waves per group
    const int g0 = (threadIdx.x / L) * wpg;
    double s = 0.0;
    for (int k = 0; k < wpg; ++k) s += red[g0 + k];
    return s;
}

