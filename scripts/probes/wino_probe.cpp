// Standalone timing probe for the Winograd kernel: one DenseNet block-1 shaped 3x3 conv (56x56x128 -> 32) at a given batch, i.e. a given
// number of workgroups per CU (28 workgroups per image: batch 8 = one per CU, 16 = two, 32 = 3.5).  Tiles 0-3 four waves, 4-7 eight waves,
// 8-11 eight waves with bf16x6 products.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -I gpu-ai-inference-server_amd/csrc scripts/probes/wino_probe.cpp -o build/wino_probe && build/wino_probe <batch> <tile>
#include "../../gpu-ai-inference-server_amd/csrc/kernels_wino.hip"

#include <cmath>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, tile = argc > 2 ? atoi(argv[2]) : 3;
    const int H = 56, W = 56, C = 128;
    float *in, *out, *u, *w;
    const size_t nin = size_t(B) * H * W * C, nout = size_t(B) * H * W * 32;
    CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&u, 16 * 32 * C * 4)); CK(hipMalloc(&w, 32 * 9 * C * 4));
    std::vector<float> h(nin);
    for (size_t i = 0; i < nin; ++i) h[i] = float(unsigned(i * 2654435761u) ^ unsigned(i >> 7) * 40503u) * (1.0f / 4294967296.0f) - 0.37f;      // full 24-bit mantissas
    CK(hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), 32 * 9 * C * 4, hipMemcpyHostToDevice));
    CK(ie::InitKernelsWino());
    CK(ie::LaunchWinogradWeights(w, u, 32, C, nullptr));
    void* ux;
    CK(hipMalloc(&ux, size_t(3) * 16 * 32 * C * 2));
    CK(ie::LaunchWinogradWeightsX6(w, ux, 32, C, nullptr));
    ie::ConvArgs a;
    a.in.p = in; a.in.n = B; a.in.h = H; a.in.w = W; a.in.c = C; a.in.sc = 1; a.in.sw = C; a.in.sh = int64_t(W) * C; a.in.sn = int64_t(H) * W * C;
    a.out.p = out; a.out.n = B; a.out.h = H; a.out.w = W; a.out.c = 32; a.out.sc = 1; a.out.sw = 32; a.out.sh = int64_t(W) * 32; a.out.sn = int64_t(H) * W * 32;
    a.wfrag = u; a.w16 = ux; a.kh = a.kw = 3; a.sh = a.sw = 1; a.pt = a.pl = 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(ie::LaunchConvWino3x3(a, tile, nullptr));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 10; ++i) CK(ie::LaunchConvWino3x3(a, tile, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (tile >= 8) {                                   // bf16x6 variant: compare with the fp32 Winograd kernel of the same tile shape
        std::vector<float> y1(nout), y0(nout);
        CK(hipMemcpy(y1.data(), out, nout * 4, hipMemcpyDeviceToHost));
        CK(ie::LaunchConvWino3x3(a, tile - 4, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(y0.data(), out, nout * 4, hipMemcpyDeviceToHost));
        double emax = 0, rmax = 0;
        for (size_t i = 0; i < nout; ++i) { emax = std::max(emax, double(std::fabs(y1[i] - y0[i]))); rmax = std::max(rmax, double(std::fabs(y0[i]))); }
        printf("  bf16x6 vs fp32 Winograd kernel: max |diff| %.3e of max |y| %.3f -> %.2e\n", emax, rmax, emax / rmax);
    }
    const double flops = 2.0 * B * H * W * 32 * 9 * C;
    printf("batch %d tile %d: %.2f us per launch, %.1f TFLOP/s direct-equivalent (%.1f executed)\n", B, tile, ms * 100.f, flops / (ms * 1e-4) / 1e12, flops / 2.25 / (ms * 1e-4) / 1e12);
    return 0;
}
