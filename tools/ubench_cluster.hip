// What a barrier among a handful of workgroups costs when they sit on ONE XCD (same L2) versus
// spread over the chip, with a small dependent gather sweep between barriers -- the price of a
// "cluster-resident" smoother for the 1024-8192-row levels of a realistic hierarchy.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_cluster.hip -o tools/bin/ubench_cluster
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                             \
    do {                                                                  \
        hipError_t e = (x);                                               \
        if (e != hipSuccess) {                                            \
            printf("%s: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

// monotonic counter barrier among `members` workgroups; bounded spin (gives up, sets *fail)
__device__ __forceinline__ bool cluster_barrier(unsigned* counter, unsigned target, int* fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);   // agent scope by default for global
        unsigned spins = 0;
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) {
            if (++spins > (1u << 22)) {
                *fail = 1;
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

template <bool SAME_XCD>
__global__ __launch_bounds__(256) void k_cluster(int members, int iters, int n, int nnz_per_row,
                                                 const int* __restrict__ ci,
                                                 const double* __restrict__ va, double* x, double* y,
                                                 unsigned* counter, int* fail, int do_sweep) {
    int me;
    if (SAME_XCD) {
        if (blockIdx.x % 8 != 0) return;   // round-robin placement: every 8th block is on XCD 0
        me = blockIdx.x / 8;
    } else {
        me = blockIdx.x;
    }
    if (me >= members) return;
    const int rows_per = (n + members - 1) / members;
    const int r0 = me * rows_per, r1 = min(n, r0 + rows_per);
    double* src = x;
    double* dst = y;
    for (int it = 0; it < iters; ++it) {
        if (do_sweep)
            for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
                double s = 0.0;
                for (int k = 0; k < nnz_per_row; ++k) {
                    const int t = r * nnz_per_row + k;
                    s += va[t] * __builtin_nontemporal_load(&src[ci[t]]);
                }
                __builtin_nontemporal_store(s, &dst[r]);
            }
        if (!cluster_barrier(counter, (unsigned)(members * (it + 1)), fail)) return;
        double* t = src;
        src = dst;
        dst = t;
    }
}

int main() {
    const int n = 2048, per = 6;
    std::vector<int> hci((size_t)n * per);
    std::vector<double> hva((size_t)n * per, 0.1), hx(n, 1.0);
    for (size_t t = 0; t < hci.size(); ++t) hci[t] = (int)((t * 2654435761u) % n);
    int *ci, *fail;
    double *va, *x, *y;
    unsigned* counter;
    CK(hipMalloc(&ci, hci.size() * 4));
    CK(hipMalloc(&va, hva.size() * 8));
    CK(hipMalloc(&x, n * 8));
    CK(hipMalloc(&y, n * 8));
    CK(hipMalloc(&counter, 4));
    CK(hipMalloc(&fail, 4));
    CK(hipMemcpy(ci, hci.data(), hci.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(va, hva.data(), hva.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int same = 1; same >= 0; --same)
        for (int sweep = 0; sweep <= 1; ++sweep)
            for (int members : {1, 2, 4, 8, 16, 32}) {
                CK(hipMemset(counter, 0, 4));
                CK(hipMemset(fail, 0, 4));
                const int grid = same ? members * 8 : members;
                CK(hipEventRecord(e0));
                if (same)
                    hipLaunchKernelGGL(k_cluster<true>, dim3(grid), dim3(256), 0, 0, members, iters, n,
                                       per, ci, va, x, y, counter, fail, sweep);
                else
                    hipLaunchKernelGGL(k_cluster<false>, dim3(grid), dim3(256), 0, 0, members, iters,
                                       n, per, ci, va, x, y, counter, fail, sweep);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                int hf = 0;
                CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
                printf("%s members=%2d sweep=%d  %.3f us per step%s\n", same ? "one XCD " : "spread  ",
                       members, sweep, 1e3 * ms / iters, hf ? "  (BARRIER GAVE UP)" : "");
                fflush(stdout);
            }
    return 0;
}
