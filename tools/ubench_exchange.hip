// What one all-to-all exchange of a half-iterate costs INSIDE a launch: the price of a
// level-resident smoother (matrix slice in registers, one launch per level visit) against the
// 4.6-5.0 us of a k_smooth launch.
//
// G workgroups of 512 threads, one per CU.  Every step each workgroup
//   1. forms its N/G rows of y = A x from matrix entries held in REGISTERS and x staged in LDS
//      (lanes-per-row DPP reduction + one LDS combine, as phase_smooth does),
//   2. publishes them as self-tagged granules (MI355X guide, Guideline 16 R2: the data IS the flag;
//      one fp64 value = two 8-byte {value32, tag} granules written by ONE 16-byte sc1 store),
//   3. sweeps all N granules of the step with 16-byte sc1 loads until every tag matches, stages the
//      values in LDS, barrier.
// Two buffers by step parity: a workgroup can only write step t+2 after it has seen all of t+1,
// which every workgroup publishes only after it has read all of t.
// A is a permutation (x_{t+1}[i] = x_t[(i+1) % N]) hidden among zero entries, so the result is
// known exactly.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_exchange.hip -o tools/bin/ubench_exchange
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

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
static constexpr int KMAX = 16;  // matrix entries per thread

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_get<0xB1, 0xf>(v);
    v += dpp_get<0x4E, 0xf>(v);
    v += dpp_get<0x141, 0xf>(v);
    v += dpp_get<0x140, 0xf>(v);
    v += dpp_get<0x142, 0xa>(v);
    v += dpp_get<0x143, 0xc>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// granule pair of one fp64 value: {lo, tag, hi, tag}
__device__ __forceinline__ v4u pack(double v, unsigned tag) {
    v4u g;
    g.x = (unsigned)__double2loint(v);
    g.y = tag;
    g.z = (unsigned)__double2hiint(v);
    g.w = tag;
    return g;
}

// PROTO 0: self-tagged granules swept by PW waves (the others wait at the barrier)
// PROTO 1: plain fp64 payload (sc1 stores by ONE wave, drained) + one flag word per producer;
//          wave 0 polls the G flags (4 per lane), then every thread loads 16 bytes of payload
// PROTO 2: no exchange at all (every workgroup keeps its stale LDS copy): the compute floor
template <int PROTO, int PW, int SLEEP, int BT, int LAUX = 16, int SAUX = 16, int PRESLEEP = 0>
__global__ __launch_bounds__(BT) void k_xchg(const unsigned short* __restrict__ pci,
                                             const double* __restrict__ pva, const double* x0,
                                             unsigned char* gran0, unsigned char* gran1,
                                             unsigned* flags, int N, int S, int steps,
                                             unsigned* tmo, double* xout, long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* xs = reinterpret_cast<double*>(smem);          // N
    double* red = xs + N;                                   // BT/64
    int* fail = reinterpret_cast<int*>(red + BT / 64);
    const int tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
    const int rpw = N / G;       // rows per workgroup
    const int L = BT / rpw;      // lanes per row (>= 64 here)
    const int g = tid / L, gl = tid - g * L;
    const int row = b * rpw + g;
    const int K = S / L;         // entries per lane
    unsigned short cj[KMAX];
    double av[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const bool ok = k < K;
        const size_t off = (size_t)row * S + (size_t)(ok ? k : 0) * L + gl;
        cj[k] = ok ? pci[off] : 0;
        av[k] = ok ? pva[off] : 0.0;
    }
    for (int j = tid; j < N; j += BT) xs[j] = x0[j];
    if (tid == 0) *fail = 0;
    __syncthreads();
    const int gbytes = PROTO == 0 ? N * 16 : N * 8;
    const auto r0 = __builtin_amdgcn_make_buffer_rsrc(gran0, 0, gbytes, 0x00020000);
    const auto r1 = __builtin_amdgcn_make_buffer_rsrc(gran1, 0, gbytes, 0x00020000);
    const auto rf = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 2 * G * 4, 0x00020000);
    long long t0 = 0, twait = 0;
    if (stamps && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    const int w = tid >> 6, lane = tid & 63;
    for (int step = 1; step <= steps; ++step) {
        // ---- rows from registers + LDS
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) s += av[k] * xs[cj[k]];
        s = wave_sum(s);
        __syncthreads();  // all reads of xs done (it is overwritten below)
        if (lane == 0) red[w] = s;
        __syncthreads();
        const auto rs = (step & 1) ? r1 : r0;
        bool bad = false;
        long long tw0 = 0;
        if (stamps && tid == 0) tw0 = __builtin_amdgcn_s_memtime();
        if (PROTO == 0) {
            if (tid < rpw) {
                double t = 0.0;
                const int wpg = L >> 6;
                for (int k = 0; k < wpg; ++k) t += red[tid * wpg + k];
                __builtin_amdgcn_raw_buffer_store_b128(pack(t, (unsigned)step), rs,
                                                       (b * rpw + tid) * 16, 0, SAUX);
            }
            if (PRESLEEP) __builtin_amdgcn_s_sleep(PRESLEEP);
            if (w < PW) {
                // granules of this wave: lane, lane+64*PW ... interleaved so that a pass is coalesced
                constexpr int NJMAX = 2048 / (64 * PW);
                const int nj = N / (64 * PW);
                unsigned spins = 0;
                for (;;) {
                    v4u gq[NJMAX];
#pragma unroll
                    for (int u = 0; u < NJMAX; ++u)
                        if (u < nj)
                            gq[u] = __builtin_amdgcn_raw_buffer_load_b128(
                                rs, ((u * PW + w) * 64 + lane) * 16, 0, LAUX);
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < NJMAX; ++u)
                        if (u < nj) ok &= (gq[u].y == (unsigned)step) & (gq[u].w == (unsigned)step);
                    if (__all(ok)) {
#pragma unroll
                        for (int u = 0; u < NJMAX; ++u)
                            if (u < nj)
                                xs[(u * PW + w) * 64 + lane] =
                                    __hiloint2double((int)gq[u].z, (int)gq[u].x);
                        break;
                    }
                    if (++spins > (1u << 18)) {
                        bad = true;
                        break;
                    }
                    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                    asm volatile("" ::: "memory");
                }
            }
        } else if (PROTO == 1) {
            if (w == 0) {
                if (lane < rpw) {
                    double t = 0.0;
                    const int wpg = L >> 6;
                    for (int k = 0; k < wpg; ++k) t += red[lane * wpg + k];
                    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
                    v2u q;
                    q.x = (unsigned)__double2loint(t);
                    q.y = (unsigned)__double2hiint(t);
                    __builtin_amdgcn_raw_buffer_store_b64(q, rs, (b * rpw + lane) * 8, 0, 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)step, rf, ((step & 1) * G + b) * 4, 0, 16);
                // poll the G flags of this parity: 4 per lane
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
                    if (lane * 4 < G) {
                        const v4u f = __builtin_amdgcn_raw_buffer_load_b128(
                            rf, ((step & 1) * G + lane * 4) * 4, 0, 16);
                        ok = (f.x == (unsigned)step) & (f.y == (unsigned)step) &
                             (f.z == (unsigned)step) & (f.w == (unsigned)step);
                    }
                    if (__all(ok)) break;
                    if (++spins > (1u << 18)) {
                        bad = true;
                        break;
                    }
                    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                    asm volatile("" ::: "memory");
                }
            }
            __syncthreads();
            // payload: 16 bytes per thread, sc1
            for (int j2 = tid; j2 < N / 2; j2 += BT) {
                const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, j2 * 16, 0, 16);
                xs[2 * j2] = __hiloint2double((int)q.y, (int)q.x);
                xs[2 * j2 + 1] = __hiloint2double((int)q.w, (int)q.z);
            }
        } else {
            if (tid < rpw) {
                double t = 0.0;
                const int wpg = L >> 6;
                for (int k = 0; k < wpg; ++k) t += red[tid * wpg + k];
                xs[b * rpw + tid] = t;
            }
        }
        if (bad) {
            *fail = 1;
            *tmo = (unsigned)step;
        }
        __syncthreads();
        if (stamps && tid == 0) twait += __builtin_amdgcn_s_memtime() - tw0;
        if (*fail) return;  // bounded spin gave up: every wave leaves
    }
    if (stamps && tid == 0) {
        stamps[2 * b] = __builtin_amdgcn_s_memrealtime() - t0;
        stamps[2 * b + 1] = twait;
    }
    if (b == 0)
        for (int j = tid; j < N; j += BT) xout[j] = xs[j];
}

typedef void (*kern_t)(const unsigned short*, const double*, const double*, unsigned char*,
                       unsigned char*, unsigned*, int, int, int, unsigned*, double*, long long*);
struct Variant {
    const char* name;
    kern_t fn;
    int proto;
    int bt;
};

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const Variant vars[] = {
        {"512thr compute only            ", k_xchg<2, 1, 0, 512>, 2, 512},
        {"512thr granules 8w sleep1      ", k_xchg<0, 8, 1, 512>, 0, 512},
        {"512thr 8w sleep1 loads sc0sc1  ", k_xchg<0, 8, 1, 512, 17, 16>, 0, 512},
        {"512thr 8w sleep1 stores sc0sc1 ", k_xchg<0, 8, 1, 512, 16, 17>, 0, 512},
        {"512thr 8w sleep1 both sc0sc1   ", k_xchg<0, 8, 1, 512, 17, 17>, 0, 512},
        {"512thr 8w sleep1 presleep 8    ", k_xchg<0, 8, 1, 512, 16, 16, 8>, 0, 512},
        {"512thr 8w sleep1 presleep 16   ", k_xchg<0, 8, 1, 512, 16, 16, 16>, 0, 512},
        {"512thr 8w sleep4               ", k_xchg<0, 8, 4, 512>, 0, 512},
        {"512thr 4w sleep1               ", k_xchg<0, 4, 1, 512>, 0, 512},
        {"512thr 8w sleep1 loads nt      ", k_xchg<0, 8, 1, 512, 18, 16>, 0, 512},
    };
    for (int N : {1024})
        for (int G : {128})
            for (const Variant& v : vars) {
                const int S = 1024;
                const int BT = v.bt;
                const int rpw = N / G, L = BT / rpw;
                if (L < 64 || S / L > KMAX) continue;
                std::vector<unsigned short> hci((size_t)N * S);
                std::vector<double> hva((size_t)N * S, 0.0), hx(N);
                for (int i = 0; i < N; ++i) {
                    for (int k = 0; k < S; ++k) hci[(size_t)i * S + k] = (unsigned short)((i * 7 + k * 13) % N);
                    const int k1 = (i * 31) % S;
                    hci[(size_t)i * S + k1] = (unsigned short)((i + 1) % N);
                    hva[(size_t)i * S + k1] = 1.0;
                    hx[i] = 0.5 + i;
                }
                unsigned short* ci;
                double *va, *x0, *xout;
                unsigned char *g0, *g1;
                unsigned *tmo, *flags;
                long long* stamps;
                CK(hipMalloc(&ci, hci.size() * 2));
                CK(hipMalloc(&va, hva.size() * 8));
                CK(hipMalloc(&x0, N * 8));
                CK(hipMalloc(&xout, N * 8));
                CK(hipMalloc(&g0, N * 16));
                CK(hipMalloc(&g1, N * 16));
                CK(hipMalloc(&tmo, 16));
                CK(hipMalloc(&flags, 2 * G * 4));
                CK(hipMalloc(&stamps, 2 * G * 8));
                CK(hipMemcpy(ci, hci.data(), hci.size() * 2, hipMemcpyHostToDevice));
                CK(hipMemcpy(va, hva.data(), hva.size() * 8, hipMemcpyHostToDevice));
                CK(hipMemcpy(x0, hx.data(), N * 8, hipMemcpyHostToDevice));
                const size_t lds = (size_t)N * 8 + (BT / 64) * 8 + 16;
                float best = 1e30f;
                unsigned htmo = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(g0, 0, N * 16));
                    CK(hipMemsetAsync(g1, 0, N * 16));
                    CK(hipMemsetAsync(flags, 0, 2 * G * 4));
                    CK(hipMemsetAsync(tmo, 0, 16));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(v.fn, dim3(G), dim3(BT), lds, 0, ci, va, x0, g0, g1, flags, N, S,
                                       steps, tmo, xout, stamps);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                    CK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
                    if (htmo) break;
                }
                std::vector<double> out(N);
                CK(hipMemcpy(out.data(), xout, N * 8, hipMemcpyDeviceToHost));
                int wrong = 0;
                if (v.proto != 2)
                    for (int i = 0; i < N; ++i)
                        if (out[i] != hx[(i + steps) % N]) ++wrong;
                std::vector<long long> hs(2 * G);
                CK(hipMemcpy(hs.data(), stamps, 2 * G * 8, hipMemcpyDeviceToHost));
                long long mx = 0, wsum = 0;
                for (int k = 0; k < G; ++k) {
                    mx = hs[2 * k] > mx ? hs[2 * k] : mx;
                    wsum += hs[2 * k + 1];
                }
                printf("N=%4d G=%3d rows/wg=%2d %2d entries/lane  %s %.3f us/step (in-kernel %.3f; publish+sweep %.0f clk avg)  wrong=%d%s\n",
                       N, G, rpw, S / L, v.name, 1e3 * best / steps, mx * 0.01 / steps,
                       (double)wsum / G / steps, wrong, htmo ? "  (SPIN GAVE UP)" : "");
                fflush(stdout);
                (void)hipFree(ci); (void)hipFree(va); (void)hipFree(x0); (void)hipFree(xout);
                (void)hipFree(g0); (void)hipFree(g1); (void)hipFree(tmo); (void)hipFree(stamps);
                (void)hipFree(flags);
            }
    return 0;
}
