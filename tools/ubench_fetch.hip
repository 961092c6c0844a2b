// What one device->host scalar round trip costs on this box, by mechanism (the setup makes ~60 of
// them per hierarchy):  hipcc -O2 --offload-arch=gfx950 tools/ubench_fetch.hip -o tools/bin/ubench_fetch
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                             \
    do {                                                                  \
        hipError_t e = (x);                                               \
        if (e != hipSuccess) {                                            \
            printf("%s: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

__global__ void k_produce(int* out, int v) { *out = v; }
__global__ void k_produce2(int* out, volatile int* host, int v) {
    *out = v;
    *host = v;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d, *pinned, *mapped;
    CK(hipMalloc(&d, 64));
    CK(hipHostMalloc(&pinned, 64, hipHostMallocDefault));
    CK(hipHostMalloc(&mapped, 64, hipHostMallocMapped | hipHostMallocCoherent));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int R = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int mode = 0; mode < 5; ++mode) {
        long long sum = 0;
        auto t0 = now();
        for (int r = 0; r < R; ++r) {
            if (mode == 0) {   // kernel, async copy to pinned, stream sync (today's fetch)
                hipLaunchKernelGGL(k_produce, dim3(1), dim3(1), 0, s, d, r);
                CK(hipMemcpyAsync(pinned, d, 4, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                sum += pinned[0];
            } else if (mode == 1) {   // kernel writes mapped host memory, stream sync
                hipLaunchKernelGGL(k_produce2, dim3(1), dim3(1), 0, s, d, mapped, r);
                CK(hipStreamSynchronize(s));
                sum += mapped[0];
            } else if (mode == 2) {   // kernel writes mapped memory, host spins on the value
                hipLaunchKernelGGL(k_produce2, dim3(1), dim3(1), 0, s, d, mapped, r + 1);
                while (*(volatile int*)mapped != r + 1) {
                }
                sum += mapped[0];
            } else if (mode == 3) {   // kernel only + stream sync (no data)
                hipLaunchKernelGGL(k_produce, dim3(1), dim3(1), 0, s, d, r);
                CK(hipStreamSynchronize(s));
            } else {                  // kernel, event record, event sync, then nothing
                hipLaunchKernelGGL(k_produce2, dim3(1), dim3(1), 0, s, d, mapped, r);
                CK(hipEventRecord(ev, s));
                CK(hipEventSynchronize(ev));
                sum += mapped[0];
            }
        }
        CK(hipStreamSynchronize(s));
        const char* names[] = {"kernel + memcpyAsync(pinned) + streamSync", "kernel->mapped + streamSync",
                               "kernel->mapped + host spin", "kernel + streamSync",
                               "kernel->mapped + event sync"};
        printf("%-44s %.2f us per round trip (check %lld)\n", names[mode], us(t0, now()) / R, sum);
    }
    return 0;
}
