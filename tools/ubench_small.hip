// Micro-benchmark of the real phase_smooth code on a synthetic rho=1 level-1 half sweep.
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -Icodes_of_ipd_ssn_amg_method_amd/csrc -Iinclude tools/ubench_smooth.hip -o tools/ubench_smooth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <string>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
#include "ipd_cycle_dev.h"
#include "ipd_cycle_phases.h"
template <bool STAGED, bool PAD>
__global__ __launch_bounds__(BT) void k_smooth(SmoothArgs a) {
    __shared__ PhaseLds lds;
    extern __shared__ __attribute__((aligned(16))) double xs_dyn[];
    phase_smooth<STAGED, PAD>(a, blockIdx.x, gridDim.x, &lds, xs_dyn);
}
// timeline: per section, the LATEST wave-finish time over the whole grid (10 ns ticks)
__device__ __forceinline__ void mark(unsigned long long* tl, int slot) {
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { atomicMax(&tl[slot], t); atomicMin(&tl[16 + slot], t); }
}
template <bool PAD>
__global__ __launch_bounds__(BT) void k_smooth_tl(SmoothArgs a, unsigned long long* tl) {
    __shared__ PhaseLds lds;
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const LevelDev& lv = a.lv; const int tid = threadIdx.x; const int L = lv.L, gpb = BT / L; const int g = tid / L, gl = tid - g * L;
    mark(tl, 0);
    int row = __builtin_amdgcn_readfirstlane(a.row0 + blockIdx.x * gpb + g);
    bool owner = gl == 0;
    RowCursor rc; RowBatch bt; row_open<PAD>(lv, row, true, owner, gl, L, rc, bt);
    double rv = 0, dv = 0, axi = 0, eo = 0; if (owner) { rv = lv.r[row]; dv = lv.dinv[row]; axi = lv.Axi[row]; eo = a.eold[row]; }
    double xxv = lv.xx[0]; double cpart = 0;
    struct Q { double e, w, r, a; };
    vec_pass(lv.N, [&](int j) { Q q; q.e = a.eold[j]; q.w = (j >= a.u0 && j < a.u1) ? a.win[j] : 0.0; q.r = lv.r[j]; q.a = lv.Axi[j]; return q; },
             [&](int j, const Q& q) { xs[j] = (j >= a.u0 && j < a.u1) ? q.w : q.e; cpart += q.r - q.a * q.e; });
    mark(tl, 1);   // vector loads arrived (and, in order, the matrix loads before them)
    __syncthreads();
    mark(tl, 2);
    double s = row_finish<PAD>(lv, rc, bt, gl, L, [&](int j) { return xs[j]; });
    mark(tl, 3);
    double xig = 0; s = reduce_rows(s, L, true, cpart, &xig, &lds);
    double c = xig / xxv;
    mark(tl, 4);
    if (owner) { if (PAD) s += rc.dg * eo; double wv = eo + dv * (rv - s - axi * c); if (a.wout) a.wout[row] = wv; a.enew[row] = wv + c; }
    mark(tl, 5);
}

template <bool PAD>
__global__ __launch_bounds__(BT) void k_tl2(SmoothArgs a, unsigned long long* tl) {
    __shared__ PhaseLds lds;
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const LevelDev& lv = a.lv; const int tid = threadIdx.x; const int L = lv.L, gpb = BT / L; const int g = tid / L, gl = tid - g * L;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int row = a.row0 + blockIdx.x * gpb + g; bool valid = row < a.row1; bool owner = valid && gl == 0;
    RowCursor rc; RowBatch bt; row_open<PAD>(lv, row, valid, owner, gl, L, rc, bt);
    double keep = bt.a[0] + bt.a[7] + bt.j[3];
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime(); if (keep == 1.2345e-300) tl[30] = 1;
    double rv = 0, dv = 0, axi = 0, eo = 0; if (owner) { rv = lv.r[row]; dv = lv.dinv[row]; axi = lv.Axi[row]; eo = a.eold[row]; }
    double xxv = lv.xx[0]; double cpart = 0;
    struct Q { double e, w, r, a; };
    vec_pass(lv.N, [&](int j) { Q q; q.e = a.eold[j]; q.w = a.win[j]; q.r = lv.r[j]; q.a = lv.Axi[j]; return q; },
             [&](int j, const Q& q) { xs[j] = (j >= a.u0 && j < a.u1) ? q.w : q.e; cpart += q.r - q.a * q.e; });
    unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    double s = row_finish<PAD>(lv, rc, bt, gl, L, [&](int j) { return xs[j]; });
    unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
    double xig = 0; s = reduce_rows(s, L, true, cpart, &xig, &lds);
    double c = xig / xxv;
    unsigned long long t5 = __builtin_amdgcn_s_memrealtime();
    if (owner) { double wv = eo + dv * (rv - s - axi * c); if (a.wout) a.wout[row] = wv; a.enew[row] = wv + c; }
    unsigned long long t6 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { tl[0] = t1 - t0; tl[1] = t2 - t1; tl[2] = t3 - t2; tl[3] = t4 - t3; tl[4] = t5 - t4; tl[5] = t6 - t5; }
}
int main() {
    const int N = 2048, nf = 1024, per = 5; const int nnz = N * per;
    std::vector<int> rp(N + 1), ci(nnz); std::vector<double> va(nnz);
    srand(1);
    for (int r = 0; r <= N; ++r) rp[r] = r * per;
    for (int r = 0; r < N; ++r) { int o = r < nf ? nf : 0; int c[4]; for (int k = 0; k < 4; ++k) c[k] = o + rand() % 1024; std::sort(c, c + 4);
        if (r < nf) { ci[r * per] = r; for (int k = 0; k < 4; ++k) ci[r * per + 1 + k] = c[k]; } else { for (int k = 0; k < 4; ++k) ci[r * per + k] = c[k]; ci[r * per + 4] = r; } }
    for (int i = 0; i < nnz; ++i) va[i] = 1.0 / (1 + i % 7);
    int *drp, *dci; double *dva; CK(hipMalloc(&drp, 4 * (N + 1))); CK(hipMalloc(&dci, 4ull * nnz)); CK(hipMalloc(&dva, 8ull * nnz));
    CK(hipMemcpy(drp, rp.data(), 4 * (N + 1), hipMemcpyHostToDevice)); CK(hipMemcpy(dci, ci.data(), 4ull * nnz, hipMemcpyHostToDevice)); CK(hipMemcpy(dva, va.data(), 8ull * nnz, hipMemcpyHostToDevice));
    auto dvec = [&](size_t n, double v) { double* p; CK(hipMalloc(&p, 8 * n)); std::vector<double> h(n, v); CK(hipMemcpy(p, h.data(), 8 * n, hipMemcpyHostToDevice)); return p; };
    LevelDev lv; lv.N = N; lv.nf = nf; lv.L = 4; lv.G = 1; lv.rp = drp; lv.ci = dci; lv.va = dva; lv.S = 0; lv.pci = nullptr; lv.pva = nullptr; lv.diag = nullptr;
    lv.dinv = dvec(N, 1e-3); lv.Axi = dvec(N, 0.01); lv.xx = dvec(1, 20.0); lv.r = dvec(N, 1.0); lv.rr = dvec(N, 0);
    double *e1 = dvec(N, 0.5), *e2 = dvec(N, 0.5), *w = dvec(N, 0.5);
    unsigned long long* tl; CK(hipMalloc(&tl, 8 * 32));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    auto timeit = [&](const std::string& name, auto fn) {
        for (int i = 0; i < 20; ++i) fn();
        CK(hipEventRecord(ev0, st)); const int reps = 400; for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(ev1, st)); CK(hipEventSynchronize(ev1)); float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
        printf("%-52s %7.2f us/launch\n", name.c_str(), 1e3 * ms / reps);
    };
    SmoothArgs a; a.lv = lv; a.row0 = 0; a.row1 = nf; a.u0 = a.u1 = 0; a.eold = e1; a.win = w; a.enew = e2; a.wout = w; a.isnsp = 1; a.staged = 1; a.eold_zero = 0;
    for (int L : {1, 2, 4, 8}) for (int G : {1, 4, 16}) {
        a.lv.L = L; a.lv.G = G;
        for (int nsp : {1, 0}) { a.isnsp = nsp;
            timeit("tree-like csr staged L=" + std::to_string(L) + " G=" + std::to_string(G) + " nsp=" + std::to_string(nsp), [&] { hipLaunchKernelGGL((k_smooth<true, false>), dim3(G), dim3(BT), 8 * N, st, a); });
        }
    }
    for (int L : {1, 4}) {
        a.lv.L = L; a.lv.G = 1; a.row1 = BT / L; a.isnsp = 1;   // exactly one iteration
        hipLaunchKernelGGL((k_tl2<false>), dim3(1), dim3(BT), 8 * N, st, a, tl); CK(hipStreamSynchronize(st));
        hipLaunchKernelGGL((k_tl2<false>), dim3(1), dim3(BT), 8 * N, st, a, tl); CK(hipStreamSynchronize(st));
        unsigned long long h[8]; CK(hipMemcpy(h, tl, 64, hipMemcpyDeviceToHost));
        printf("L=%d one block, thread 0 sections (us): row_open %.2f  vec_pass %.2f  barrier %.2f  row_finish %.2f  reduce+c %.2f  update %.2f\n", L, h[0] / 100.0, h[1] / 100.0, h[2] / 100.0, h[3] / 100.0, h[4] / 100.0, h[5] / 100.0);
    }
    return 0;
}
