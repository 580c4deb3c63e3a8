// Standalone check + timing of dense_block_f16_kernel (kernels_block.hip): a random chain of dense layers on B images, against a CPU
// restatement that rounds to half exactly where the kernel does (prologue result, bottleneck tensor, new channels).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I gpu-ai-inference-server_amd/csrc scripts/probes/block_probe.cpp -o build/block_probe
//   build/block_probe <batch> <H=W> <K0> <layers>
#include "../../gpu-ai-inference-server_amd/csrc/kernels_block.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static unsigned long long g_s = 0x9E3779B97F4A7C15ull;
static float urand() {                       // U[0,1)
    g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17;
    return float((g_s >> 40) & 0xFFFFFF) / 16777216.0f;
}
static float hr(float v) { return float(_Float16(v)); }      // round to half and back

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, H = argc > 2 ? atoi(argv[2]) : 14, K0 = argc > 3 ? atoi(argv[3]) : 256, NL = argc > 4 ? atoi(argv[4]) : 3;
    const int W = H, P = K0 + 32 * NL;
    const size_t npix = size_t(B) * H * W;
    std::vector<_Float16> x(npix * P);
    for (size_t p = 0; p < npix; ++p)
        for (int c = 0; c < P; ++c) x[p * P + c] = c < K0 ? _Float16(urand() * 2.f - 1.f) : _Float16(1000.f);
    std::vector<_Float16> w16;
    std::vector<float> w32;
    ie::DenseBlockArgs a;
    a.pitch = P; a.in_coff = 0; a.n = B; a.h = H; a.w = W; a.nlayers = NL;
    for (int l = 0; l < NL; ++l) {
        ie::DenseBlockLayer& L = a.layer[l];
        L.K = K0 + 32 * l;
        L.out_coff = L.K;
        L.w1 = unsigned(w16.size());
        for (int i = 0; i < 128 * L.K; ++i) w16.push_back(_Float16((urand() * 2.f - 1.f) * 1.7f / std::sqrt(float(L.K))));
        L.w3 = unsigned(w16.size());
        for (int i = 0; i < 32 * 1152; ++i) w16.push_back(_Float16((urand() * 2.f - 1.f) * 1.7f / std::sqrt(1152.f)));
        L.ps = unsigned(w16.size());
        for (int i = 0; i < L.K; ++i) w16.push_back(_Float16(0.5f + urand()));
        L.pt = unsigned(w16.size());
        for (int i = 0; i < L.K; ++i) w16.push_back(_Float16(urand() * 0.6f - 0.3f));
        L.b1 = unsigned(w32.size());
        for (int i = 0; i < 128; ++i) w32.push_back(urand() * 0.2f - 0.1f);
        L.b3 = unsigned(w32.size());
        for (int i = 0; i < 32; ++i) w32.push_back(urand() * 0.2f - 0.1f);
        L.flags = 1 | 2;
        while (w16.size() % 8) w16.push_back(_Float16(0.f));
    }
    _Float16 *dx, *dw16, *dwf;
    float* dw32;
    CK(hipMalloc(&dx, x.size() * 2)); CK(hipMalloc(&dw16, w16.size() * 2)); CK(hipMalloc(&dwf, w16.size() * 2)); CK(hipMalloc(&dw32, std::max(w32.size() * 4, w16.size() * 4)));
    CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw32, w32.data(), w32.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dwf, 0, w16.size() * 2));
    for (int l = 0; l < NL; ++l) {
        CK(ie::LaunchPermuteWeightsFrag16(dw16 + a.layer[l].w1, dwf + a.layer[l].w1, 128, a.layer[l].K, nullptr));
        CK(ie::LaunchPermuteWeightsFrag16(dw16 + a.layer[l].w3, dwf + a.layer[l].w3, 32, 1152, nullptr));
    }
    a.x = dx; a.w16 = dw16; a.wfrag16 = dwf; a.w32 = dw32; a.w16_bytes = w16.size() * 2;
    CK(ie::InitKernelsBlock());
    if (!ie::DenseBlockEligible(a)) { printf("not eligible\n"); return 2; }
    CK(ie::LaunchDenseBlockF16(a, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<_Float16> y(x.size());
    CK(hipMemcpy(y.data(), dx, x.size() * 2, hipMemcpyDeviceToHost));

    // ---- CPU restatement on a few images ----
    double emax = 0, rmax = 0;
    size_t bad = 0;
    const int imgs[3] = {0, B / 2, B - 1};
    for (int ii = 0; ii < 3; ++ii) {
        const int b = imgs[ii];
        if (ii > 0 && b == imgs[ii - 1]) continue;
        std::vector<float> xi(size_t(H) * W * P);
        for (size_t i = 0; i < xi.size(); ++i) xi[i] = float(x[size_t(b) * H * W * P + i]);
        for (int l = 0; l < NL; ++l) {
            const ie::DenseBlockLayer& L = a.layer[l];
            std::vector<float> T(size_t(H) * W * 128);
            std::vector<float> arow(L.K);
            for (int p = 0; p < H * W; ++p) {
                for (int k = 0; k < L.K; ++k) {
                    float v = hr(std::fma(xi[size_t(p) * P + k], float(w16[L.ps + k]), float(w16[L.pt + k])));
                    arow[k] = v > 0.f ? v : 0.f;
                }
                for (int n = 0; n < 128; ++n) {
                    double s = 0;
                    for (int k = 0; k < L.K; ++k) s += double(arow[k]) * double(float(w16[L.w1 + size_t(n) * L.K + k]));
                    float v = float(s) + w32[L.b1 + n];
                    T[size_t(p) * 128 + n] = hr(v > 0.f ? v : 0.f);
                }
            }
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx)
                    for (int n = 0; n < 32; ++n) {
                        double s = 0;
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                const int iy = yy + ky - 1, ix = xx + kx - 1;
                                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                                const float* t = &T[size_t(iy * W + ix) * 128];
                                const _Float16* wp = &w16[L.w3 + (size_t(n) * 9 + ky * 3 + kx) * 128];
                                for (int k = 0; k < 128; ++k) s += double(t[k]) * double(float(wp[k]));
                            }
                        xi[size_t(yy * W + xx) * P + L.out_coff + n] = hr(float(s) + w32[L.b3 + n]);
                    }
        }
        for (size_t i = 0; i < xi.size(); ++i) {
            const double d = std::fabs(double(float(y[size_t(b) * H * W * P + i])) - double(xi[i]));
            emax = std::max(emax, d);
            rmax = std::max(rmax, std::fabs(double(xi[i])));
            if (d > 2e-2) ++bad;
        }
    }
    printf("B=%d %dx%d K0=%d layers=%d: max |diff| %.3e of max |ref| %.3f (%zu elements off by > 2e-2)\n", B, H, W, K0, NL, emax, rmax, bad);

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(ie::LaunchDenseBlockF16(a, nullptr));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 10; ++i) CK(ie::LaunchDenseBlockF16(a, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    {
        long long* dbg;
        CK(hipMalloc(&dbg, 128));
        CK(hipMemset(dbg, 0, 128));
        a.dbg = dbg;
        CK(ie::LaunchDenseBlockF16(a, nullptr));
        CK(hipDeviceSynchronize());
        long long h[16];
        CK(hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost));
        printf("  consumer wave 0, cycles per layer: 1x1 loop total %.0f (of it barrier waits %.0f, compute + issue %.0f), 1x1 epilogue -> barrier %.0f, 3x3 + stores %.0f, closing barrier %.0f\n",
               double(h[2]) / NL, double(h[0]) / NL, double(h[1]) / NL, double(h[3]) / NL, double(h[4]) / NL, double(h[5]) / NL);
        printf("  producer wave 4, cycles per layer: 1x1 loop total %.0f (commit %.0f, issue %.0f, barrier waits %.0f), 3x3 weights -> LDS -> barrier %.0f, prefetch -> closing barrier %.0f\n",
               double(h[11]) / NL, double(h[8]) / NL, double(h[9]) / NL, double(h[10]) / NL, double(h[12]) / NL, double(h[13]) / NL);
        a.dbg = nullptr;
    }
    double flops = 0;
    for (int l = 0; l < NL; ++l) flops += 2.0 * B * H * W * (128.0 * a.layer[l].K + 32.0 * 1152);
    printf("  %.1f us per launch (%.2f us per layer), %.1f TFLOP/s\n", ms * 100.f, ms * 100.f / NL, flops / (ms * 1e-4) / 1e12);
    return bad ? 3 : 0;
}
