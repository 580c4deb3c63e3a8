// Standalone timing of conv_stem_kernel (kernels_stem.hip) with parts of it switched off at compile time, to see what bounds it:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I gpu-ai-inference-server_amd/csrc [-DSTEM_ABLATE=n] scripts/probes/stem_probe.cpp -o build/stem_probe_n
//   build/stem_probe_n <batch> <1: half | 0: float | 2: half + fused max pool>
// STEM_ABLATE: 0 whole kernel, 1 no output stores, 2 no input loads, 3 no MFMA / LDS operand reads, 4 no epilogue arithmetic and stores.
#include "../../gpu-ai-inference-server_amd/csrc/kernels_stem.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, mode = argc > 2 ? atoi(argv[2]) : 1, half = mode != 0, pool = mode == 2;
    const int H = 224, W = 224, OH = pool ? 56 : 112, OW = pool ? 56 : 112, C = 64;
    float *x, *w, *bias;
    void* y;
    const size_t xin = size_t(B) * 3 * H * W, yout = size_t(B) * OH * OW * C;
    CK(hipMalloc(&x, xin * 4)); CK(hipMalloc(&w, 64 * 147 * 4)); CK(hipMalloc(&bias, 64 * 4)); CK(hipMalloc(&y, yout * 4));
    std::vector<float> hx(xin), hw(64 * 147), hb(64, 0.1f);
    for (size_t i = 0; i < xin; ++i) hx[i] = float((i * 2654435761u >> 8) & 0xFFFF) / 65536.f - 0.5f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = float((i * 40503u) & 0xFFF) / 4096.f - 0.5f;
    CK(hipMemcpy(x, hx.data(), xin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), 256, hipMemcpyHostToDevice));
    ie::ConvArgs a{};
    a.in.p = x; a.in.n = B; a.in.c = 3; a.in.h = H; a.in.w = W; a.in.sw = 1; a.in.sh = W; a.in.sc = int64_t(H) * W; a.in.sn = a.in.sc * 3;
    a.out.p = static_cast<float*>(y); a.out.n = B; a.out.c = C; a.out.h = OH; a.out.w = OW; a.out.sc = 1; a.out.sw = C; a.out.sh = int64_t(OW) * C; a.out.sn = a.out.sh * OH;
    a.out.f16 = half;
    a.w = w; a.bias = bias; a.kh = 7; a.kw = 7; a.sh = 2; a.sw = 2; a.pt = 3; a.pl = 3; a.relu = 1;
    CK(ie::InitKernelsStem());
    if (!(pool ? ie::ConvStemPoolEligible(a) : ie::ConvStemEligible(a))) { printf("not eligible\n"); return 2; }
    auto launch = [&]() { return pool ? ie::LaunchConvStemPool(a, nullptr) : ie::LaunchConvStem(a, nullptr); };
    for (int i = 0; i < 3; ++i) CK(launch());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 20; ++i) CK(launch());
    CK(hipEventRecord(e1, nullptr));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
#ifndef STEM_ABLATE
#define STEM_ABLATE 0
#endif
    const double bytes = xin * 4.0 + yout * (half ? 2.0 : 4.0);
    printf("ablate %d  B=%d %s: %.1f us per launch, %.2f TB/s of algorithmic bytes\n", STEM_ABLATE, B, pool ? "half + pool" : (half ? "half" : "float"), ms * 50.f, bytes / (ms / 20 * 1e-3) / 1e12);
    return 0;
}
