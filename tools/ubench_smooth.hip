// Micro-benchmark of the real phase_smooth code on a synthetic rho=1 level-1 half sweep.
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -Icodes_of_ipd_ssn_amg_method_amd/csrc -Iinclude tools/ubench_smooth.hip -o tools/ubench_smooth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <string>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
static constexpr int BT = 1024;
struct LevelDev { int N, nf, L, G; const int* rp; const int* ci; const double* va; int S; const unsigned short* pci; const double* pva; const double* diag; const double* dinv; const double* Axi; const double* xx; double* r; double* rr; };
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
#include "ipd_cycle_phases.h"
template <bool STAGED, bool PAD>
__global__ __launch_bounds__(BT) void k_smooth(SmoothArgs a) {
    __shared__ PhaseLds lds;
    extern __shared__ __attribute__((aligned(16))) double xs_dyn[];
    phase_smooth<STAGED, PAD>(a, blockIdx.x, &lds, xs_dyn);
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
int main() {
    const int N = 2048, nf = 1024, per = 1025; const int nnz = N * per;
    std::vector<int> rp(N + 1), ci(nnz); std::vector<double> va(nnz);
    for (int r = 0; r <= N; ++r) rp[r] = r * per;
    for (int r = 0; r < nf; ++r) { ci[r * per] = r; for (int k = 1; k < per; ++k) ci[r * per + k] = nf - 1 + k; }
    for (int r = nf; r < N; ++r) { for (int k = 0; k < per - 1; ++k) ci[r * per + k] = k; ci[r * per + per - 1] = r; }
    for (int i = 0; i < nnz; ++i) va[i] = 1.0 / (1 + i % 7);
    const int S = 1024; std::vector<unsigned short> pci((size_t)N * S); std::vector<double> pva((size_t)N * S), dg(N);
    for (int r = 0; r < N; ++r) { int k = 0; for (int t = rp[r]; t < rp[r + 1]; ++t) { if (ci[t] == r) dg[r] = va[t]; else { pci[(size_t)r * S + k] = ci[t]; pva[(size_t)r * S + k] = va[t]; ++k; } } }
    int *drp, *dci; double *dva; CK(hipMalloc(&drp, 4 * (N + 1))); CK(hipMalloc(&dci, 4ull * nnz)); CK(hipMalloc(&dva, 8ull * nnz));
    CK(hipMemcpy(drp, rp.data(), 4 * (N + 1), hipMemcpyHostToDevice)); CK(hipMemcpy(dci, ci.data(), 4ull * nnz, hipMemcpyHostToDevice)); CK(hipMemcpy(dva, va.data(), 8ull * nnz, hipMemcpyHostToDevice));
    unsigned short* dpci; double* dpva; CK(hipMalloc(&dpci, 2ull * N * S)); CK(hipMalloc(&dpva, 8ull * N * S));
    CK(hipMemcpy(dpci, pci.data(), 2ull * N * S, hipMemcpyHostToDevice)); CK(hipMemcpy(dpva, pva.data(), 8ull * N * S, hipMemcpyHostToDevice));
    auto dvec = [&](size_t n, double v) { double* p; CK(hipMalloc(&p, 8 * n)); std::vector<double> h(n, v); CK(hipMemcpy(p, h.data(), 8 * n, hipMemcpyHostToDevice)); return p; };
    LevelDev lv; lv.N = N; lv.nf = nf; lv.L = 256; lv.G = 256; lv.rp = drp; lv.ci = dci; lv.va = dva; lv.S = S; lv.pci = dpci; lv.pva = dpva; lv.diag = dvec(N, 2.0);
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
    for (int G : {256, 64, 8, 1}) {
        a.lv.G = G; a.row1 = G * 4;   // each block handles exactly its 4 rows (L=256)
        for (int nsp : {1, 0}) { a.isnsp = nsp;
            timeit("pad  staged G=" + std::to_string(G) + " nsp=" + std::to_string(nsp), [&] { hipLaunchKernelGGL((k_smooth<true, true>), dim3(G), dim3(BT), 8 * N, st, a); });
            timeit("csr  staged G=" + std::to_string(G) + " nsp=" + std::to_string(nsp), [&] { hipLaunchKernelGGL((k_smooth<true, false>), dim3(G), dim3(BT), 8 * N, st, a); });
            timeit("pad unstaged G=" + std::to_string(G) + " nsp=" + std::to_string(nsp), [&] { hipLaunchKernelGGL((k_smooth<false, true>), dim3(G), dim3(BT), 0, st, a); });
        }
    }
    for (int G : {256, 1}) {
        a.lv.G = G; a.row1 = G * 4; a.isnsp = 1;
        std::vector<unsigned long long> init(32, 0); for (int i = 16; i < 32; ++i) init[i] = ~0ull;
        CK(hipMemcpy(tl, init.data(), 8 * 32, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((k_smooth_tl<true>), dim3(G), dim3(BT), 8 * N, st, a, tl); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(tl, init.data(), 8 * 32, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((k_smooth_tl<true>), dim3(G), dim3(BT), 8 * N, st, a, tl); CK(hipStreamSynchronize(st));
        unsigned long long h[32]; CK(hipMemcpy(h, tl, 8 * 32, hipMemcpyDeviceToHost));
        unsigned long long t0 = h[16];
        printf("timeline G=%d (us after the first wave started; latest / earliest wave):\n", G);
        const char* nm[6] = {"start", "vec+matrix loads arrived", "after stage barrier", "row dot done", "reduce+c done", "update stored"};
        for (int i = 0; i < 6; ++i) printf("   %-28s %6.2f / %6.2f\n", nm[i], (h[i] - t0) / 100.0, (h[16 + i] - t0) / 100.0);
    }
    return 0;
}
