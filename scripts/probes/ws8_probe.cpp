// Timing + bit-equality probe for conv1x1_ws_f8_kernel (kernels_ws8.hip) against conv_igemm_f8_kernel on ResNet-50 bottleneck shapes at a
// given batch:   build/ws8_probe <batch> <H=W> <K> <N> [K2]      (K2 > 0: also the DUAL launch with a second input of K2 channels)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I gpu-ai-inference-server_amd/csrc scripts/probes/ws8_probe.cpp -o build/ws8_probe
#include "../../gpu-ai-inference-server_amd/csrc/kernels_ws8.hip"
#include "../../gpu-ai-inference-server_amd/csrc/kernels_f8.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ie {      // (lives in kernels.hip in the library)
int ResidentPerCu(const void* kernel, int block, size_t lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, lds) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
    return n;
}
}  // namespace ie

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static ie::TensorArg nhwc8(void* p, int n, int h, int w, int c) {
    ie::TensorArg t;
    t.p = static_cast<float*>(p); t.n = n; t.h = h; t.w = w; t.c = c; t.sc = 1; t.sw = c; t.sh = int64_t(w) * c; t.sn = int64_t(h) * w * c; t.f8 = 1;
    return t;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 56, K = argc > 3 ? atoi(argv[3]) : 64, N = argc > 4 ? atoi(argv[4]) : 256;
    const int K2 = argc > 5 ? atoi(argv[5]) : 0;
    const size_t M = size_t(B) * H * H;
    unsigned char *in, *in2, *res, *out, *out2, *w8, *w8b;
    float *es, *bi;
    CK(hipMalloc(&in, M * K)); CK(hipMalloc(&in2, M * (K2 > 0 ? K2 : 32))); CK(hipMalloc(&res, M * N)); CK(hipMalloc(&out, M * N)); CK(hipMalloc(&out2, M * N));
    CK(hipMalloc(&w8, size_t(N) * K)); CK(hipMalloc(&w8b, size_t(N) * (K2 > 0 ? K2 : 32))); CK(hipMalloc(&es, N * 4)); CK(hipMalloc(&bi, N * 4));
    std::vector<unsigned char> h(M * size_t(std::max(std::max(N, K), std::max(K2, 32))) + 64);
    for (size_t i = 0; i < h.size(); ++i) { unsigned v = unsigned(i * 2654435761u) >> 24; h[i] = (v & 0x7f) >= 0x78 ? (v & 0x87) | 0x30 : v; }   // finite e4m3 codes, |x| < 240
    CK(hipMemcpy(in, h.data(), M * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(in2, h.data() + 17, M * (K2 > 0 ? K2 : 32), hipMemcpyHostToDevice));
    CK(hipMemcpy(res, h.data() + 5, M * N - 5, hipMemcpyHostToDevice));
    CK(hipMemcpy(w8, h.data() + 3, size_t(N) * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(w8b, h.data() + 9, size_t(N) * (K2 > 0 ? K2 : 32), hipMemcpyHostToDevice));
    std::vector<float> e(N, 1e-4f), b(N, 0.25f);
    CK(hipMemcpy(es, e.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bi, b.data(), N * 4, hipMemcpyHostToDevice));
    CK(ie::InitKernelsF8()); CK(ie::InitKernelsWs8());
    ie::ConvArgs a;
    a.in = nhwc8(in, B, H, H, K); a.out = nhwc8(out, B, H, H, N); a.res = nhwc8(res, B, H, H, N);
    a.w8 = w8; a.escale = es; a.bias = bi; a.relu = 1; a.res_scale = 0.01f; a.out_qscale = 3.0f;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto fn, const char* name) -> int {
        for (int i = 0; i < 2; ++i) if (fn() != hipSuccess) { printf("  %s: declined\n", name); return 0; }
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 10; ++i) fn();
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = double(M) * (K + 2.0 * N);
        printf("  %-28s %8.1f us  %6.2f TB/s algorithmic (in + residual + out)\n", name, ms * 100.0, bytes / (ms * 1e-4) / 1e12);
        return 0;
    };
    printf("B=%d %dx%d K=%d N=%d (M=%zu) + residual\n", B, H, H, K, N, M);
    for (int t = 0; t < 7; ++t) { char nm[64]; snprintf(nm, 64, "igemm_f8 tile %d", t); timeit([&] { return ie::LaunchConvIgemmF8(a, t, nullptr); }, nm); }
    std::vector<unsigned char> y0(M * N), y1(M * N);
    CK(ie::LaunchConvIgemmF8(a, 3, nullptr)); CK(hipDeviceSynchronize());
    CK(hipMemcpy(y0.data(), out, M * N, hipMemcpyDeviceToHost));
    a.out = nhwc8(out2, B, H, H, N);
    for (int t = 0; t < ie::kNumConvWs8Tiles; ++t) {
        char nm[64]; snprintf(nm, 64, "ws8 tile %d", t);
        CK(hipMemset(out2, 0x7f, M * N));
        if (ie::LaunchConvWs1x1F8(a, t, nullptr) != hipSuccess) { printf("  %s: declined\n", nm); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(y1.data(), out2, M * N, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < y0.size(); ++i) diff += y0[i] != y1[i];
        timeit([&] { return ie::LaunchConvWs1x1F8(a, t, nullptr); }, nm);
        printf("      bytes different from the igemm's output: %zu of %zu\n", diff, y0.size());
    }
    if (K2 > 0) {
        ie::ConvArgs d = a;
        d.res = ie::TensorArg();
        d.in2 = nhwc8(in2, B, H, H, K2); d.w8b = w8b; d.escale_b = es; d.bias_b = bi; d.sh2 = d.sw2 = 1;
        printf(" DUAL (second GEMM K2=%d instead of the residual tensor):\n", K2);
        for (int t = 1; t < ie::kNumConvWs8Tiles; ++t) { char nm[64]; snprintf(nm, 64, "ws8 dual tile %d", t); timeit([&] { return ie::LaunchConvWs1x1F8(d, t, nullptr); }, nm); }
        // the two-launch form it replaces: projection conv (writes the shortcut), then conv + residual
        ie::ConvArgs p = a;
        p.in = nhwc8(in2, B, H, H, K2); p.w8 = w8b; p.res = ie::TensorArg(); p.relu = 0; p.out = nhwc8(res, B, H, H, N);
        timeit([&] { return ie::LaunchConvWs1x1F8(p, 0, nullptr); }, "ws8 projection alone (t0)");
        timeit([&] { return ie::LaunchConvIgemmF8(p, 3, nullptr); }, "igemm projection alone (t3)");
    }
    return 0;
}
