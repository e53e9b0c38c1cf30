// Micro-benchmarks behind the design of potrf_diag2_kernel: what a LONE wave per SIMD pays for dependent v_mfma chains.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_latency mfma_latency.hip && ./mfma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k64(double* out, long long* cyc, int n) {
    const int lane = threadIdx.x & 63;
    f64x4 c = {1.0, 0.5, 0.25, 0.125}, e = {0.0, 0.0, 0.0, 0.0};
    double a = 1e-3 * lane, b = 1e-3;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {              // dependent chain, one accumulator
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        } else if (MODE == 1) {       // two independent chains
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e, 0, 0, 0);
        } else if (MODE == 2) {       // chain through a VALU op on the result (read accumulator -> mul -> operand)
            const double v = c[0] * 1e-3;
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, b, c, 0, 0, 0);
        } else if (MODE == 3) {       // as 2, with a second independent MFMA per step (the chol16 pattern)
            const double v = c[0] * 1e-3;
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, b, c, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e, 0, 0, 0);
        } else if (MODE == 4) {       // dependent v_fma_f64 chain (8 per step)
            for (int q = 0; q < 8; ++q) a = fma(a, b, a);
        } else if (MODE == 5) {       // v_rcp_f64 + two Newton steps, dependent
            double y = __builtin_amdgcn_rcp(a + 2.0);
            double en = fma(-(a + 2.0), y, 1.0);
            y = fma(y, en, y);
            en = fma(-(a + 2.0), y, 1.0);
            a = fma(y, en, y);
        } else if (MODE == 6) {       // readlane -> VALU use, dependent
            const int lo = __builtin_amdgcn_readlane(__double2loint(a), 5), hi = __builtin_amdgcn_readlane(__double2hiint(a), 5);
            a = a * __hiloint2double(hi, lo) + 1.0;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = c[0] + c[1] + e[0] + a;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
template <int NACC, bool ONEWAVE>
__global__ __launch_bounds__(256, 1) void kacc(double* out, long long* cyc, int n, int slot) {
    const int lane = threadIdx.x & 63;
    if (ONEWAVE && threadIdx.x >= 64) return;
    f64x4 c[NACC];
    for (int q = 0; q < NACC; ++q) c[q] = f64x4{1.0 + q, 0.5, 0.25, 0.125};
    const double a = 1e-3 * lane, b = 1e-3;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int q = 0; q < NACC; ++q) s += c[q][0] + c[q][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[slot] = t1 - t0;
}
// NW waves per SIMD (256 * NW threads), NACC independent accumulators each: aggregate issue rate of the matrix pipe
template <int NW, int NACC>
__global__ __launch_bounds__(256 * NW, NW) void kwaves(double* out, long long* cyc, int n, int slot) {
    const int lane = threadIdx.x & 63;
    f64x4 c[NACC];
    for (int q = 0; q < NACC; ++q) c[q] = f64x4{1.0 + q, 0.5, 0.25, 0.125};
    const double a = 1e-3 * lane, b = 1e-3;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int q = 0; q < NACC; ++q) s += c[q][0] + c[q][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[slot] = t1 - t0;
}
template <int MODE>
__global__ __launch_bounds__(256, 1) void k32(float* out, long long* cyc, int n) {
    const int lane = threadIdx.x & 63;
    f32x4 c = {1.0f, 0.5f, 0.25f, 0.125f}, e = {0.f, 0.f, 0.f, 0.f};
    float a = 1e-3f * lane, b = 1e-3f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        else if (MODE == 2) { const float v = c[0] * 1e-3f; c = __builtin_amdgcn_mfma_f32_16x16x4f32(v, b, c, 0, 0, 0); }
        else { const float v = c[0] * 1e-3f; c = __builtin_amdgcn_mfma_f32_16x16x4f32(v, b, c, 0, 0, 0); e = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, e, 0, 0, 0); }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = c[0] + c[1] + e[0] + a;
    if (threadIdx.x == 0) cyc[8 + MODE] = t1 - t0;
}
int main() {
    double* o; float* of; long long* c;
    hipMalloc(&o, 16384); hipMalloc(&of, 4096); hipMalloc(&c, 32 * 8); hipMemset(c, 0, 256);
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k64<0>, dim3(1), dim3(256), 0, 0, o, c, n);
        hipLaunchKernelGGL(k64<1>, dim3(1), dim3(256), 0, 0, o, c, n);
        hipLaunchKernelGGL(k64<2>, dim3(1), dim3(256), 0, 0, o, c, n);
        hipLaunchKernelGGL(k64<3>, dim3(1), dim3(256), 0, 0, o, c, n);
        hipLaunchKernelGGL(k64<4>, dim3(1), dim3(256), 0, 0, o, c, n);
        hipLaunchKernelGGL(k64<5>, dim3(1), dim3(256), 0, 0, o, c, n);
        hipLaunchKernelGGL(k64<6>, dim3(1), dim3(256), 0, 0, o, c, n);
        hipLaunchKernelGGL(k32<0>, dim3(1), dim3(256), 0, 0, of, c, n);
        hipLaunchKernelGGL(k32<2>, dim3(1), dim3(256), 0, 0, of, c, n);
        hipLaunchKernelGGL(k32<3>, dim3(1), dim3(256), 0, 0, of, c, n);
        hipLaunchKernelGGL((kacc<1, false>), dim3(1), dim3(256), 0, 0, o, c, n, 16);
        hipLaunchKernelGGL((kacc<2, false>), dim3(1), dim3(256), 0, 0, o, c, n, 17);
        hipLaunchKernelGGL((kacc<3, false>), dim3(1), dim3(256), 0, 0, o, c, n, 18);
        hipLaunchKernelGGL((kacc<4, false>), dim3(1), dim3(256), 0, 0, o, c, n, 19);
        hipLaunchKernelGGL((kacc<8, false>), dim3(1), dim3(256), 0, 0, o, c, n, 20);
        hipLaunchKernelGGL((kacc<16, false>), dim3(1), dim3(256), 0, 0, o, c, n, 21);
        hipLaunchKernelGGL((kacc<4, true>), dim3(1), dim3(256), 0, 0, o, c, n, 22);
        hipLaunchKernelGGL((kacc<16, true>), dim3(1), dim3(256), 0, 0, o, c, n, 23);
        hipLaunchKernelGGL((kwaves<1, 8>), dim3(1), dim3(256), 0, 0, o, c, n, 24);
        hipLaunchKernelGGL((kwaves<2, 8>), dim3(1), dim3(512), 0, 0, o, c, n, 25);
        hipLaunchKernelGGL((kwaves<3, 8>), dim3(1), dim3(768), 0, 0, o, c, n, 26);
        hipLaunchKernelGGL((kwaves<4, 8>), dim3(1), dim3(1024), 0, 0, o, c, n, 27);
        hipLaunchKernelGGL((kwaves<4, 4>), dim3(1), dim3(1024), 0, 0, o, c, n, 28);
        hipLaunchKernelGGL((kwaves<2, 16>), dim3(1), dim3(512), 0, 0, o, c, n, 29);
        hipDeviceSynchronize();
    }
    long long h[32];
    hipMemcpy(h, c, 256, hipMemcpyDeviceToHost);
    {
        const int nw[] = {1, 2, 3, 4, 4, 2}, na[] = {8, 8, 8, 8, 4, 16};
        for (int i = 0; i < 6; ++i)
            printf("LAT f64 %d waves per SIMD x %2d accumulators: %6.1f cycles per MFMA per SIMD (pipe: 64)\n", nw[i], na[i], (double)h[24 + i] / n / na[i] / nw[i]);
    }
    const int nacc[] = {1, 2, 3, 4, 8, 16, 4, 16};
    for (int i = 0; i < 8; ++i)
        printf("LAT f64 %2d independent accumulators, %s: %7.1f cycles per MFMA\n", nacc[i], i < 6 ? "4 waves (one per SIMD)" : "ONE wave on the CU     ", (double)h[16 + i] / n / nacc[i]);
    const char* nm[] = {"f64 dependent MFMA chain", "f64 two independent chains (per pair)", "f64 MFMA -> read acc -> mul -> MFMA", "f64 same + independent MFMA",
                        "f64 8 dependent v_fma", "f64 rcp + 2 Newton", "f64 readlane pair -> fma", "", "f32 dependent MFMA chain", "", "f32 MFMA -> read -> mul -> MFMA", "f32 same + independent MFMA"};
    for (int i = 0; i < 12; ++i)
        if (nm[i][0]) printf("LAT %-40s %7.1f cycles per step\n", nm[i], (double)h[i] / n);
    return 0;
}
