// Standalone timing of conv1x1_ws_f16_kernel (kernels_ws.hip) on a dense-block-1 shaped layer, with parts switched off at compile time:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I gpu-ai-inference-server_amd/csrc [-DWS_ABLATE=n] scripts/probes/ws_probe.cpp -o build/ws_probe_n
//   build/ws_probe_n <K> <tile> [batch = 128] [H = W = 56]
// WS_ABLATE: 0 whole kernel, 1 no weight preamble, 2 no BN+ReLU prologue, 3 no stores (out-of-range offsets: dropped by the hardware), 5 no activation
// loads (out-of-range offsets: zeros, at once), 8 neither loads nor stores = the on-chip work alone.  (Skipping the MFMAs instead is not a valid variant:
// the compiler then removes the loads that fed them.)  -DWS_PER_CU=n overrides the workgroups per CU the persistent grid is sized for.
#include "../../gpu-ai-inference-server_amd/csrc/kernels_ws.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ie {      // (lives in kernels.hip in the library)
int ResidentPerCu(const void* kernel, int block, size_t lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, lds) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
    return n;
}
}  // namespace ie

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 128, tile = argc > 2 ? atoi(argv[2]) : 1, N = 128;
    const int B = argc > 3 ? atoi(argv[3]) : 128, H = argc > 4 ? atoi(argv[4]) : 56, W = H, P = K > 256 ? 512 : 256;      // P: the block buffer's pixel pitch
    const size_t npix = size_t(B) * H * W;
    _Float16 *x, *y, *w16, *ps, *pt;
    float* bias;
    CK(hipMalloc(&x, npix * P * 2)); CK(hipMalloc(&y, npix * N * 2)); CK(hipMalloc(&w16, size_t(N) * K * 2)); CK(hipMalloc(&ps, K * 2)); CK(hipMalloc(&pt, K * 2));
    CK(hipMalloc(&bias, N * 4));
    CK(hipMemset(x, 0x11, npix * P * 2)); CK(hipMemset(w16, 0x11, size_t(N) * K * 2)); CK(hipMemset(ps, 0x3c, K * 2)); CK(hipMemset(pt, 0, K * 2)); CK(hipMemset(bias, 0, N * 4));
    ie::ConvArgs a{};
    a.in.p = reinterpret_cast<float*>(x); a.in.n = B; a.in.c = K; a.in.h = H; a.in.w = W; a.in.sc = 1; a.in.sw = P; a.in.sh = int64_t(W) * P; a.in.sn = a.in.sh * H; a.in.f16 = 1;
    a.out.p = reinterpret_cast<float*>(y); a.out.n = B; a.out.c = N; a.out.h = H; a.out.w = W; a.out.sc = 1; a.out.sw = N; a.out.sh = int64_t(W) * N; a.out.sn = a.out.sh * H; a.out.f16 = 1;
    a.w16 = w16; a.w = reinterpret_cast<float*>(w16); a.bias = bias; a.kh = 1; a.kw = 1; a.sh = 1; a.sw = 1; a.relu = 1; a.pre_relu = 1;
    a.pre_scale = reinterpret_cast<float*>(ps); a.pre_shift = reinterpret_cast<float*>(pt); a.pre_scale16 = ps; a.pre_shift16 = pt;
    CK(ie::InitKernelsWs());
    if (!ie::ConvWsEligible(a, tile)) { printf("not eligible\n"); return 2; }
    for (int i = 0; i < 3; ++i) CK(ie::LaunchConvWs1x1F16(a, tile, nullptr));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 20; ++i) CK(ie::LaunchConvWs1x1F16(a, tile, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = double(npix) * (K + N) * 2;
    printf("ablate %d  K=%d tile %d B=%d %dx%d: %.1f us per launch, %.2f TB/s of algorithmic bytes\n", WS_ABLATE, K, tile, B, H, W, ms * 50.f, bytes / (ms / 20 * 1e-3) / 1e12);
    return 0;
}
