// store_cost.hip -- what a global store costs the wave that issues it on gfx950 (development aid; results in
// profiles/r03_experiments.md).  One workgroup; each of NW waves issues NS back-to-back global_store_dwordx4 (64 lanes x 16 B =
// 8 full 128-byte rows per instruction) to distinct addresses and stamps the cycle counter before and after ISSUE (no wait for
// completion), then after s_waitcnt vmcnt(0).
//   hipcc --offload-arch=gfx950 -O3 -o store_cost store_cost.hip && ./store_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int NS>
__global__ __launch_bounds__(512, 2) void kstore(double* out, long long* cyc, int rowstride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f64x2 v; v[0] = (double)lane; v[1] = (double)wave;
    double* base = out + (size_t)wave * NS * 8 * rowstride + (size_t)(lane >> 3) * rowstride + (lane & 7) * 2;
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < NS; ++i) *reinterpret_cast<f64x2*>(base + (size_t)i * 8 * rowstride) = v;
    const long long t1 = (long long)__builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = (long long)__builtin_readcyclecounter();
    if (lane == 0) { cyc[wave * 2] = t1 - t0; cyc[wave * 2 + 1] = t2 - t0; }
}

int main() {
    double* o; long long* c;
    hipMalloc(&o, (size_t)64 << 20); hipMalloc(&c, 64 * sizeof(long long));
    hipMemset(o, 0, (size_t)64 << 20);
    for (int rep = 0; rep < 2; ++rep)
        for (int nw : {1, 2, 8})
            for (int stride : {16, 2048}) {
                hipLaunchKernelGGL(kstore<16>, dim3(1), dim3(64 * nw), 0, 0, o, c, stride);
                hipDeviceSynchronize();
                std::vector<long long> h(16);
                hipMemcpy(h.data(), c, 16 * sizeof(long long), hipMemcpyDeviceToHost);
                if (rep) printf("STORE %d waves x 16 stores of 1 KiB, row pitch %5d B: issue %5lld cycles (%4.0f per store), completed after %5lld\n", nw,
                                stride * 8, h[0], h[0] / 16.0, h[1]);
            }
    return 0;
}
