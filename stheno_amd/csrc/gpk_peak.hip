// gpk_peak.hip -- the chip's SUSTAINED matrix-pipe rate, measured: the number every `roofline.frac` of this library should be read
// against beside the nominal peak ("Peaks to divide by": re-derive on the box from a measured MFMA micro-benchmark
// and print the value used).
//
// One launch fills every SIMD of the device with WAVES register-resident waves (no LDS, no global loads inside the loop): each
// wave streams v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 -- the two instructions every GEMM of this library is made of --
// into ACC independent accumulators, operands rotating through eight registers of pseudo-random values (a dense MFMA stream on
// random data is the most power-hungry thing the chip can be asked to do; constant operands would flatter the clock).  The launch
// is repeated until `min_ms` of device time have passed, so the figure is the rate the power management GRANTS a sustained stream,
// not a burst.  Reported: flops / event time, and the shader clock the stream ran at (s_memtime cycles of one wave over the
// constant 100 MHz wall clock), from which the pipe's issue efficiency follows (rate / (CUs x clock x flops per CU and cycle)).
//
// A measurement hook like gpk_prof_*: it synchronises the stream; it is not on the product path.
#include "gpk_common.hpp"
#include "../../include/gpk.h"

namespace {

constexpr int PEAK_ACC = 8;        // independent accumulators per wave (>= the pipe's depth for back-to-back issue)
constexpr int PEAK_OPS = 8;        // operand registers the stream rotates through

__device__ __forceinline__ unsigned long long peak_wall_clock() { return wall_clock64(); }

template <typename T>
__global__ __launch_bounds__(512, 1) void mfma_peak_kernel(int iters, T* sink, unsigned long long* stamps) {
    typedef typename Traits<T>::acc_t acc_t;
    const int tid = threadIdx.x;
    // pseudo-random operands in (-1, 1): a different value in every lane and register
    unsigned s = (unsigned)(blockIdx.x * 512 + tid) * 2654435761u + 12345u;
    T a[PEAK_OPS], b[PEAK_OPS];
#pragma unroll
    for (int q = 0; q < PEAK_OPS; ++q) {
        s = s * 1664525u + 1013904223u;
        a[q] = (T)((int)(s >> 8) - (1 << 23)) * (T)(1.0 / (1 << 23));
        s = s * 1664525u + 1013904223u;
        b[q] = (T)((int)(s >> 8) - (1 << 23)) * (T)(1.0 / (1 << 23)) * (T)(1.0 / 64);
    }
    acc_t c[PEAK_ACC];
#pragma unroll
    for (int q = 0; q < PEAK_ACC; ++q) c[q] = acc_t{(T)0, (T)0, (T)0, (T)0};
    __syncthreads();
    const unsigned long long w0 = peak_wall_clock();
    const long long c0 = (long long)__builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < PEAK_OPS; ++r) {
#pragma unroll
            for (int q = 0; q < PEAK_ACC; ++q) c[q] = Traits<T>::mfma(a[(q + r) % PEAK_OPS], b[r], c[q]);
        }
    }
    const long long c1 = (long long)__builtin_readcyclecounter();
    const unsigned long long w1 = peak_wall_clock();
    T t = (T)0;
#pragma unroll
    for (int q = 0; q < PEAK_ACC; ++q) t += c[q][0] + c[q][1] + c[q][2] + c[q][3];
    if (t == (T)123.456) sink[0] = t;                   // (keeps the accumulators alive; never true)
    if (tid == 0 && stamps != nullptr) {
        stamps[2 * blockIdx.x] = (unsigned long long)(c1 - c0);
        stamps[2 * blockIdx.x + 1] = w1 - w0;
    }
}

template <typename T>
int mfma_peak(double min_ms, int waves_per_simd, double* tflops, double* ms_out, double* clock_mhz, double* issue_eff, hipStream_t stream) {
    if (tflops == nullptr) return GPK_ERR_ARG(4);
    if (waves_per_simd != 1 && waves_per_simd != 2) return GPK_ERR_ARG(3);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        return GPK_ERR_LAUNCH;
    const int threads = 256 * waves_per_simd;          // one workgroup per CU: `waves_per_simd` waves on each of its four SIMDs
    const int iters = 4096;                             // x 64 MFMAs per wave and launch: ~4-8 ms per launch
    T* sink = nullptr;
    unsigned long long* stamps = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int st = GPK_ERR_LAUNCH;
    double total_ms = 0.0, cyc = 0.0, wall = 0.0;
    long long launches = 0;
    unsigned long long* host = nullptr;
    do {
        if (hipMalloc(&sink, 64) != hipSuccess) break;
        if (hipMalloc(&stamps, sizeof(unsigned long long) * 2 * cus) != hipSuccess) break;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
        host = (unsigned long long*)malloc(sizeof(unsigned long long) * 2 * cus);
        if (host == nullptr) break;
        // warm-up (code object load, clocks ramp), then timed launches until min_ms of device time
        hipLaunchKernelGGL((mfma_peak_kernel<T>), dim3(cus), dim3(threads), 0, stream, iters, sink, stamps);
        if (hipStreamSynchronize(stream) != hipSuccess) break;
        bool ok = true;
        while (total_ms < min_ms && launches < 100000) {
            if (hipEventRecord(e0, stream) != hipSuccess) { ok = false; break; }
            for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((mfma_peak_kernel<T>), dim3(cus), dim3(threads), 0, stream, iters, sink, stamps);
            if (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { ok = false; break; }
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { ok = false; break; }
            total_ms += ms;
            launches += 4;
        }
        if (!ok || hipGetLastError() != hipSuccess) break;
        if (hipMemcpy(host, stamps, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost) != hipSuccess) break;
        for (int i = 0; i < cus; ++i) {
            cyc += (double)host[2 * i];
            wall += (double)host[2 * i + 1];
        }
        st = GPK_OK;
    } while (0);
    if (host) free(host);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (sink) (void)hipFree(sink);
    if (stamps) (void)hipFree(stamps);
    if (st != GPK_OK) return st;
    const double flops_per_mfma = 2.0 * 16 * 16 * 4;
    const double flops = (double)launches * cus * (threads / 64) * (double)iters * PEAK_OPS * PEAK_ACC * flops_per_mfma;
    *tflops = flops / (total_ms * 1e-3) / 1e12;
    if (ms_out) *ms_out = total_ms;
    // the wall clock ticks at 100 MHz: shader cycles per tick x 100 = MHz (of the LAST launch's k loops, averaged over the CUs)
    const double mhz = wall > 0 ? cyc / wall * 100.0 : 0.0;
    if (clock_mhz) *clock_mhz = mhz;
    // what the pipes deliver per cycle at that clock: fp64 128, fp32 256 flops per CU and cycle (16x16x4 MFMA: 64 / 32 cycles per SIMD)
    const double per_cu_cycle = sizeof(T) == 8 ? 128.0 : 256.0;
    if (issue_eff) *issue_eff = mhz > 0 ? (*tflops * 1e12) / ((double)cus * mhz * 1e6 * per_cu_cycle) : 0.0;
    return GPK_OK;
}

}  // namespace

extern "C" int gpk_mfma_peak(int dtype, double min_ms, int waves_per_simd, double* tflops, double* ms, double* clock_mhz, double* issue_eff,
                             void* stream) {
    if (dtype == GPK_F32) return mfma_peak<float>(min_ms, waves_per_simd, tflops, ms, clock_mhz, issue_eff, (hipStream_t)stream);
    if (dtype == GPK_F64) return mfma_peak<double>(min_ms, waves_per_simd, tflops, ms, clock_mhz, issue_eff, (hipStream_t)stream);
    return -1;
}
