// Micro-benchmark: where does a 12.6 MB CSR half-sweep launch spend its time?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_rows.hip -o /tmp/ubench_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ __launch_bounds__(1024) void k_empty(int) {}

// pure stream: 16 B/lane
__global__ __launch_bounds__(1024) void k_stream(const double2* __restrict__ a, size_t n2, double* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n2; i += stride) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}

template <int U, int BTH>
__global__ __launch_bounds__(BTH) void k_rows(int nrows, int L, const int* __restrict__ rp, const int* __restrict__ ci,
                                               const double* __restrict__ va, const double* __restrict__ x, double* __restrict__ y) {
    __shared__ double red[BTH / 64];
    const int tid = threadIdx.x, gpb = BTH / L, g = tid / L, gl = tid - g * L;
    const int G = gridDim.x;
    const int niter = (nrows + G * gpb - 1) / (G * gpb);
    for (int it = 0; it < niter; ++it) {
        const int row = (it * G + blockIdx.x) * gpb + g;
        const bool valid = row < nrows;
        int e0 = 0, e1 = 0;
        if (valid) { e0 = rp[row]; e1 = rp[row + 1]; }
        double s = 0;
        for (int t = e0 + gl; t < e1; t += U * L) {
            int j[U]; double a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { int tt = t + u * L; bool ok = tt < e1; j[u] = ok ? ci[tt] : -1; a[u] = ok ? va[tt] : 0.0; }
#pragma unroll
            for (int u = 0; u < U; ++u) s += a[u] * (j[u] >= 0 ? x[j[u]] : 0.0);
        }
        if (L <= 64) { for (int d = L >> 1; d > 0; d >>= 1) s += __shfl_xor(s, d); }
        else {
            for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            const int wpg = L >> 6, g0 = (tid / L) * wpg; double t = 0; for (int k = 0; k < wpg; ++k) t += red[g0 + k]; s = t;
        }
        if (valid && gl == 0) y[row] = s;
    }
}

int main() {
    const int nrows = 1024, ncols = 2048, per = 1025;
    const int nnz = nrows * per;
    std::vector<int> rp(nrows + 1), ci(nnz); std::vector<double> va(nnz), x(ncols, 1.0);
    for (int r = 0; r <= nrows; ++r) rp[r] = r * per;
    for (int r = 0; r < nrows; ++r) { ci[r * per] = r; for (int k = 1; k < per; ++k) ci[r * per + k] = 1023 + k; }
    for (int i = 0; i < nnz; ++i) va[i] = 1.0 / (1 + i % 7);
    int *drp, *dci; double *dva, *dx, *dy; double2* dbig;
    CK(hipMalloc(&drp, 4 * (nrows + 1))); CK(hipMalloc(&dci, 4 * nnz)); CK(hipMalloc(&dva, 8 * nnz));
    CK(hipMalloc(&dx, 8 * ncols)); CK(hipMalloc(&dy, 8 * ncols)); CK(hipMalloc(&dbig, 12ull * nnz));
    CK(hipMemcpy(drp, rp.data(), 4 * (nrows + 1), hipMemcpyHostToDevice)); CK(hipMemcpy(dci, ci.data(), 4 * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(dva, va.data(), 8 * nnz, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, x.data(), 8 * ncols, hipMemcpyHostToDevice));
    CK(hipMemset(dbig, 0, 12ull * nnz));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn) {
        for (int i = 0; i < 20; ++i) fn();
        CK(hipEventRecord(e0, st));
        const int reps = 400;
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-40s %7.2f us/launch  %7.1f GB/s (12.66MB)\n", name, 1e3 * ms / reps, 12.66e6 / (1e-3 * ms / reps) / 1e9);
    };
    timeit("empty 256x1024", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, st, 0); });
    timeit("empty 1024x256", [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, 0); });
    for (int g : {256, 512, 1024, 2048})
        timeit(("stream 12.6MB grid " + std::to_string(g) + "x1024").c_str(), [&] { hipLaunchKernelGGL(k_stream, dim3(g), dim3(1024), 0, st, dbig, (size_t)(12ull * nnz / 16), dy); });
    timeit("stream 12.6MB grid 2048x256", [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, st, dbig, (size_t)(12ull * nnz / 16), dy); });
    for (int L : {64, 128, 256, 512, 1024}) {
        int G = std::min(256, (nrows * L + 1023) / 1024);
        timeit(("rows U4 BT1024 L=" + std::to_string(L) + " G=" + std::to_string(G)).c_str(), [&] { hipLaunchKernelGGL((k_rows<4, 1024>), dim3(G), dim3(1024), 0, st, nrows, L, drp, dci, dva, dx, dy); });
    }
    for (int L : {64, 128, 256}) {
        int G = (nrows * L + 255) / 256;
        timeit(("rows U4 BT256 L=" + std::to_string(L) + " G=" + std::to_string(G)).c_str(), [&] { hipLaunchKernelGGL((k_rows<4, 256>), dim3(G), dim3(256), 0, st, nrows, L, drp, dci, dva, dx, dy); });
        timeit(("rows U8 BT256 L=" + std::to_string(L) + " G=" + std::to_string(G)).c_str(), [&] { hipLaunchKernelGGL((k_rows<8, 256>), dim3(G), dim3(256), 0, st, nrows, L, drp, dci, dva, dx, dy); });
    }
    for (int L : {32, 64}) {
        int G = (nrows * L + 255) / 256;
        timeit(("rows U16 BT256 L=" + std::to_string(L) + " G=" + std::to_string(G)).c_str(), [&] { hipLaunchKernelGGL((k_rows<16, 256>), dim3(G), dim3(256), 0, st, nrows, L, drp, dci, dva, dx, dy); });
    }
    return 0;
}
