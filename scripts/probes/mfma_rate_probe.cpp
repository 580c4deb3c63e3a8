// fp32 MFMA issue-rate probe: cycles per v_mfma_f32_32x32x2_f32 for one wave per SIMD and for two, alone and interleaved with
// packed-fp32 VALU work of another / the same wave.   hipcc --offload-arch=gfx950 -O3 mfma_rate_probe.cpp -o mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VALU_PER_MFMA>
__global__ __launch_bounds__(256) void rate_kernel(float* out, unsigned long long* cyc, int iters, float x) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{x + i, x - i};
    const float a0 = x + threadIdx.x, b0 = x * 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[k & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < VALU_PER_MFMA; ++j) v[(k + j) & 7] = v[(k + j) & 7] * v[(k + j + 1) & 7] + v[(k + j + 2) & 7];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
void run(int blocks, float* out, unsigned long long* cyc) {
    const int iters = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<V><<<blocks, 256>>>(out, cyc, iters, 1.0f);
    hipEventRecord(e0);
    rate_kernel<V><<<blocks, 256>>>(out, cyc, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("blocks %4d  valu/mfma %d : %7.1f cycles per MFMA (wave 0 of block 0), kernel %.1f us -> %.1f ns per MFMA-slot\n", blocks, V, double(h[0]) / (iters * 16),
           ms * 1e3, ms * 1e6 / (iters * 16));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 4 * 8);
    for (int blocks : {256, 512, 1024}) {
        run<0>(blocks, out, cyc); run<1>(blocks, out, cyc); run<2>(blocks, out, cyc); run<4>(blocks, out, cyc); run<8>(blocks, out, cyc);
    }
    return 0;
}
