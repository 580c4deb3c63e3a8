// Standalone check + timing of the bf16x6 1x1 kernel against float64 on DenseNet block-1 / block-2 shapes.
//   hipcc --offload-arch=gfx950 -O3 -I gpu-ai-inference-server_amd/csrc scripts/probes/x6_probe.cpp -o build/x6_probe && build/x6_probe
#include "../../gpu-ai-inference-server_amd/csrc/kernels_x6.hip"

#include <cmath>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run(int B, int HW, int K, int pitch, int tile, bool pre) {
    const int M = B * HW * HW, N = 128;
    std::vector<float> in(size_t(M) * pitch), w(size_t(N) * K), sc(K), sf(K), bias(N);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return float(s >> 8) / 16777216.f; };
    for (auto& v : in) v = rnd() * 2.f - 0.7f;
    for (auto& v : w) v = (rnd() - 0.5f) * 0.2f;
    for (int k = 0; k < K; ++k) { sc[k] = 0.5f + rnd(); sf[k] = rnd() - 0.5f; }
    for (auto& v : bias) v = rnd() - 0.5f;
    float *din, *dw, *dsc, *dsf, *db, *dout; void* dw6;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dsc, K * 4)); CK(hipMalloc(&dsf, K * 4)); CK(hipMalloc(&db, N * 4));
    CK(hipMalloc(&dout, size_t(M) * N * 4)); CK(hipMalloc(&dw6, size_t(3) * N * K * 2));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, sc.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsf, sf.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bias.data(), N * 4, hipMemcpyHostToDevice));
    CK(ie::LaunchSplitWeightsX6(dw, dw6, N, K, nullptr));
    ie::ConvArgs a;
    a.in.p = din; a.in.n = B; a.in.h = HW; a.in.w = HW; a.in.c = K; a.in.sc = 1; a.in.sw = pitch; a.in.sh = int64_t(HW) * pitch; a.in.sn = int64_t(HW) * HW * pitch;
    a.out.p = dout; a.out.n = B; a.out.h = HW; a.out.w = HW; a.out.c = N; a.out.sc = 1; a.out.sw = N; a.out.sh = int64_t(HW) * N; a.out.sn = int64_t(HW) * HW * N;
    a.w16 = dw6; a.bias = db; a.relu = 1;
    if (pre) { a.pre_scale = dsc; a.pre_shift = dsf; a.pre_relu = 1; }
    CK(ie::LaunchConvX6(a, tile, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<float> out(size_t(M) * N);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, rmax = 0;
    for (int t = 0; t < 600; ++t) {
        const int m = (t < 200 ? t : (t < 400 ? M - 1 - ((t - 200) % M) : int((uint64_t(t) * 2654435761u) % M))) % M;
        for (int n = 0; n < N; n += 7) {
            double acc = bias[n];
            for (int k = 0; k < K; ++k) {
                float x = in[size_t(m) * pitch + k];
                if (pre) { x = x * sc[k] + sf[k]; x = x > 0 ? x : 0; }
                acc += double(x) * double(w[size_t(n) * K + k]);
            }
            if (acc < 0) acc = 0;
            emax = std::max(emax, std::fabs(acc - out[size_t(m) * N + n]));
            rmax = std::max(rmax, std::fabs(acc));
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(ie::LaunchConvX6(a, tile, nullptr));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 10; ++i) CK(ie::LaunchConvX6(a, tile, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 100.0, flops = 2.0 * M * N * K, bytes = double(M) * (K + N) * 4;
    printf("M=%6d K=%3d pitch=%4d tile %d pre=%d: %7.2f us  %6.1f TFLOP/s  %5.2f TB/s   max err %.2e of max |ref| %.2f  -> %.2e\n", M, K, pitch, tile, int(pre), us,
           flops / us / 1e6, bytes / us / 1e6, emax, rmax, emax / rmax);
    hipFree(din); hipFree(dw); hipFree(dsc); hipFree(dsf); hipFree(db); hipFree(dout); hipFree(dw6);
    return 0;
}

int main() {
    CK(ie::InitKernelsX6());
    if (run(1, 10, 96, 128, 0, true)) return 1;            // small, ragged (100 pixels), odd chunk count
    if (run(1, 10, 96, 128, 1, false)) return 1;
    if (run(1, 10, 64, 128, 1, false)) return 1;
    if (run(2, 8, 96, 128, 1, false)) return 1;
    if (run(2, 8, 64, 64, 1, false)) return 1;
    if (run(8, 8, 96, 96, 0, false)) return 1;
    for (int tile = 0; tile < 2; ++tile) {
        for (int K : {64, 128, 224}) if (run(32, 56, K, 256, tile, true)) return 1;
        for (int K : {128, 256, 480}) if (run(32, 28, K, 512, tile, true)) return 1;
    }
    return 0;
}
