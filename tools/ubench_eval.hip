// Micro-benchmark for the driver's evaluation pass (csrc/ipd_driver.hip, OpEval): which part of
// the per-entry work keeps the pass below the HBM roofline?
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../codes_of_ipd_ssn_amg_method_amd/csrc \
//         tools/ubench_eval.hip -o /tmp/ubench_eval && /tmp/ubench_eval 4096
// Variants (cumulative): 0 read + row sum, 1 + zk/prox arithmetic, 2 + byte mask store,
// 3 + per-column wave reductions (DPP), 4 = 3 with the butterfly column reduction.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ipd_cycle_dev.h"

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("%s: %s\n", #x, hipGetErrorString(e));                      \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

constexpr int TC = 16;

// 16 values per lane (one per column) -> per-column sums over the 64 lanes, returned spread
// over lanes: a butterfly reduce-scatter needs 8+4+2+1 exchanges instead of 16 full wave sums.
__device__ __forceinline__ double xor_get(double v, int mask) { return __shfl_xor(v, mask); }

template <int VAR>
__global__ __launch_bounds__(256) void k_var(const double* __restrict__ w,
                                             const double* __restrict__ p,
                                             const double* __restrict__ q,
                                             const double* __restrict__ lam, int m, int n, int reps,
                                             double itk, unsigned char* __restrict__ s,
                                             double* __restrict__ lpart, double* __restrict__ rpart) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int ib = blockIdx.x, jg = blockIdx.y;
    const int i = ib * 256 + tid;
    if (i >= m) return;
    const double pi = p[i], y2 = lam[n + i];
    double lacc = 0.0, acc = 0.0;
    double* rrow = rpart + ((size_t)ib * 4 + wv) * n;
    for (int rep = 0; rep < reps; ++rep) {
        const int j0 = (jg * reps + rep) * TC;
        if (j0 >= n) break;
        double raw[TC];
#pragma unroll
        for (int jj = 0; jj < TC; ++jj) raw[jj] = w[(size_t)(j0 + jj) * m + i];
        double xv[TC];
#pragma unroll
        for (int jj = 0; jj < TC; ++jj) {
            const int j = j0 + jj;
            double x = raw[jj];
            if (VAR >= 1) {
                const double aty = pi * lam[j] + y2 * q[j];
                const double z = itk * (raw[jj] - aty);
                x = z > 0.0 ? z : 0.0;
                acc += x * x;
                if (VAR >= 2) s[(size_t)j * m + i] = z >= 0.0 ? 1 : 0;
            }
            lacc += x * q[j];
            xv[jj] = x * pi;
        }
        if (VAR == 3) {
#pragma unroll
            for (int jj = 0; jj < TC; ++jj) {
                const double cs = wave_sum(xv[jj]);
                if (lane == 0) rrow[j0 + jj] = cs;
            }
        }
        if (VAR == 4) {
            // reduce-scatter over lane bits 0..3: after step b a lane keeps the half of its
            // columns selected by bit b of its lane id and adds its partner's copy of them
            double a8[8], a4[4], a2[2], a1;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool hi = lane & 1;
                const double keep = hi ? xv[k + 8] : xv[k];
                const double give = hi ? xv[k] : xv[k + 8];
                a8[k] = keep + xor_get(give, 1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool hi = lane & 2;
                const double keep = hi ? a8[k + 4] : a8[k];
                const double give = hi ? a8[k] : a8[k + 4];
                a4[k] = keep + xor_get(give, 2);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bool hi = lane & 4;
                const double keep = hi ? a4[k + 2] : a4[k];
                const double give = hi ? a4[k] : a4[k + 2];
                a2[k] = keep + xor_get(give, 4);
            }
            {
                const bool hi = lane & 8;
                const double keep = hi ? a2[1] : a2[0];
                const double give = hi ? a2[0] : a2[1];
                a1 = keep + xor_get(give, 8);
            }
            // lanes with equal (lane & 15) hold partial sums of the same column: add the 4 rows
            a1 += xor_get(a1, 16);
            a1 += xor_get(a1, 32);
            if (lane < 16) {
                const int col = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
                rrow[j0 + col] = a1;
            }
        }
    }
    lpart[(size_t)jg * m + i] = lacc + acc;
}

template <int VAR>
static void run(int N, const double* w, const double* p, const double* q, const double* lam,
                unsigned char* s, double* lpart, double* rpart) {
    const int nib = (N + 255) / 256, njb = (N + TC - 1) / TC;
    for (int reps : {1, 2, 4}) {
        const int njg = (njb + reps - 1) / reps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_var<VAR>, dim3(nib, njg), dim3(256), 0, 0, w, p, q, lam, N, N, reps,
                           39.2, s, lpart, rpart);
        CK(hipEventRecord(e0));
        const int R = 50;
        for (int r = 0; r < R; ++r)
            hipLaunchKernelGGL(k_var<VAR>, dim3(nib, njg), dim3(256), 0, 0, w, p, q, lam, N, N, reps,
                               39.2, s, lpart, rpart);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / R;
        const double bytes = 8.0 * N * (double)N + (VAR >= 2 ? (double)N * N : 0.0);
        printf("N=%d var=%d reps=%d grid=%dx%d  %.2f us  %.0f GB/s\n", N, VAR, reps, nib, njg, us,
               bytes / us * 1e-3);
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t mn = (size_t)N * N;
    double *w, *p, *q, *lam, *lpart, *rpart;
    unsigned char* s;
    CK(hipMalloc(&w, mn * 8));
    CK(hipMalloc(&s, mn));
    CK(hipMalloc(&p, N * 8));
    CK(hipMalloc(&q, N * 8));
    CK(hipMalloc(&lam, 2 * N * 8));
    CK(hipMalloc(&lpart, (size_t)N * ((N + 15) / 16) * 8));
    CK(hipMalloc(&rpart, (size_t)N * 4 * ((N + 255) / 256) * 8));
    std::vector<double> h(mn);
    for (size_t k = 0; k < mn; ++k) h[k] = (double)((k * 2654435761u) % 1000) / 1000.0 - 0.3;
    CK(hipMemcpy(w, h.data(), mn * 8, hipMemcpyHostToDevice));
    std::vector<double> one(2 * N, 1.0);
    CK(hipMemcpy(p, one.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, one.data(), N * 8, hipMemcpyHostToDevice));
    for (auto& v : one) v = 0.01;
    CK(hipMemcpy(lam, one.data(), 2 * N * 8, hipMemcpyHostToDevice));
    run<0>(N, w, p, q, lam, s, lpart, rpart);
    run<1>(N, w, p, q, lam, s, lpart, rpart);
    run<2>(N, w, p, q, lam, s, lpart, rpart);
    run<3>(N, w, p, q, lam, s, lpart, rpart);
    run<4>(N, w, p, q, lam, s, lpart, rpart);
    return 0;
}
