// fp32-accurate products on the bf16 matrix pipe ("bf16x6"): a = a0 + a1 + a2 exactly (three bf16 pieces of 8 mantissa bits), and
// a*b ~= a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0): six v_mfma_f32_32x32x16_bf16 per 32x32x16 block against eight
// v_mfma_f32_32x32x2_f32.  The probe measures (1) the issue rate of both, alone and with packed-fp32 VALU work between the MFMAs,
// (2) the error of a 32 x 32 x K product against float64 for native fp32 MFMAs, the six-term and the three-term (two-piece) split.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/bf16x6_probe.cpp -o build/bf16x6_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline float trunc_bf16(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }
__device__ inline __bf16 as_bf16(float x) { return __builtin_bit_cast(__bf16, (unsigned short)(__builtin_bit_cast(unsigned, x) >> 16)); }

// ---- rate: MODE 0 = 8 fp32 MFMAs per block, 1 = 6 bf16 MFMAs per block; V packed-fp32 VALU ops after every MFMA ----
template <int MODE, int V>
__global__ __launch_bounds__(256) void rate_kernel(float* out, unsigned long long* cyc, int iters, float x) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{x + i, x - i};
    const float a0 = x + threadIdx.x, b0 = x * 0.5f;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = as_bf16(a0 + i); bh[i] = as_bf16(b0 - i); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < (MODE ? 6 : 8) * 4; ++k) {                  // four 32x32x16 blocks (one per accumulator)
            if constexpr (MODE) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[k & 3], 0, 0, 0);
            else acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[k & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < V; ++j) v[(k + j) & 7] = v[(k + j) & 7] * v[(k + j + 1) & 7] + v[(k + j + 2) & 7];
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

template <int MODE, int V>
void run_rate(int blocks, float* out, unsigned long long* cyc) {
    const int iters = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE, V><<<blocks, 256>>>(out, cyc, iters, 1.0f);
    hipEventRecord(e0);
    rate_kernel<MODE, V><<<blocks, 256>>>(out, cyc, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-22s waves/SIMD %d  valu/mfma %d : %7.1f cycles per 32x32x16 block, kernel %6.1f us -> %6.1f ns per block per wave\n", MODE ? "6 x bf16 32x32x16" : "8 x fp32 32x32x2",
           blocks / 256, V, double(h[0]) / (iters * 4), ms * 1e3, ms * 1e6 / (iters * 4));
}

// ---- accuracy: one wave computes C[32][32] = A[32][K] * B[K][32]; modes 0 fp32 MFMA, 1 six-term split, 2 three-term (two pieces) ----
template <int MODE>
__global__ __launch_bounds__(64) void gemm_kernel(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if constexpr (MODE == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + hh], B[(k + hh) * 32 + r], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a[3], b[3];
            for (int i = 0; i < 8; ++i) {
                const float av = A[r * K + k + hh * 8 + i], bv = B[(k + hh * 8 + i) * 32 + r];
                const float a0 = trunc_bf16(av), a1 = trunc_bf16(av - a0), a2 = av - a0 - a1;
                const float b0 = trunc_bf16(bv), b1 = trunc_bf16(bv - b0), b2 = bv - b0 - b1;
                a[0][i] = as_bf16(a0); a[1][i] = as_bf16(a1); a[2][i] = as_bf16(a2);
                b[0][i] = as_bf16(b0); b[1][i] = as_bf16(b1); b[2][i] = as_bf16(b2);
            }
            // smallest terms first
            if constexpr (MODE == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * hh) * 32 + r] = acc[e];
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 4 * 8);
    for (int blocks : {256, 512}) {
        run_rate<0, 0>(blocks, out, cyc); run_rate<0, 2>(blocks, out, cyc);
        run_rate<1, 0>(blocks, out, cyc); run_rate<1, 2>(blocks, out, cyc); run_rate<1, 4>(blocks, out, cyc);
    }
    for (int K : {128, 1024}) {
        std::vector<float> A(32 * K), B(K * 32), C(1024);
        srand(7);
        for (auto& v : A) v = (rand() / float(RAND_MAX)) * 2.f - 0.5f;      // mostly positive, like post-ReLU activations
        for (auto& v : B) v = (rand() / float(RAND_MAX)) * 2.f - 1.f;
        std::vector<double> ref(1024, 0.0);
        double refmax = 0;
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += double(A[m * K + k]) * double(B[k * 32 + n]); ref[m * 32 + n] = s; refmax = std::max(refmax, std::fabs(s)); }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        const char* names[3] = {"native fp32 MFMA", "bf16 x 6 terms", "bf16 x 3 terms (2 pieces)"};
        for (int mode = 0; mode < 3; ++mode) {
            if (mode == 0) gemm_kernel<0><<<1, 64>>>(dA, dB, dC, K); else if (mode == 1) gemm_kernel<1><<<1, 64>>>(dA, dB, dC, K); else gemm_kernel<2><<<1, 64>>>(dA, dB, dC, K);
            hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
            double emax = 0;
            for (int i = 0; i < 1024; ++i) emax = std::max(emax, std::fabs(C[i] - ref[i]));
            printf("K=%4d  %-26s max |err| / max |ref| = %.3e\n", K, names[mode], emax / refmax);
        }
    }
    return 0;
}
