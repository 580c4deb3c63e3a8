// What bounds the streaming 1x1 kernels (3.3-4.2 TB/s of algorithmic bytes) -- the access SHAPE of an MFMA-fragment stream?  A wave of
// conv1x1_ws_*_kernel reads, per instruction, 32 B from each of 32 pixel rows (lane = (pixel, k half)) and writes the same shape.  This probe
// moves the same bytes (RB read + WB written per pixel, persistent waves, one 32-pixel row block at a time, no MFMA, no LDS) with the lanes of an
// instruction spread over 32, 16 or 8 rows (32 / 64 / 128 contiguous bytes per row), next to a plain 16-B-per-lane copy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/stream_probe.cpp -o build/stream_probe && build/stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy16(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) out[i] = in[i];
}

// LPR lanes per row (2, 4, 8): an instruction touches 64 / LPR rows with 16 * LPR contiguous bytes each.  RB, WB bytes per pixel (multiples of 16 * LPR).
template <int LPR, int RB, int WB, int NT_STORE>
__global__ __launch_bounds__(256) void pattern(const char* __restrict__ in, char* __restrict__ out, int nrb) {
    constexpr int ROWS = 64 / LPR, PASSES = 32 / ROWS;      // row groups per 32-pixel block
    constexpr int NL = RB / (16 * LPR), NS = WB / (16 * LPR);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane / LPR, q = lane % LPR;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in), 0, int(size_t(nrb) * 32 * RB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, int(size_t(nrb) * 32 * WB), 0x00020000);
    u32x4 v[PASSES * NL];
    const int stride = gridDim.x * 4;
    int rb = blockIdx.x * 4 + wave;
    auto issue = [&](int b) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int c = 0; c < NL; ++c)
                v[p * NL + c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, b < nrb ? unsigned((b * 32 + p * ROWS + row) * RB + c * 16 * LPR + q * 16) : 0x80000000u, 0, 0);
    };
    issue(rb);
    while (rb < nrb) {
        u32x4 w[PASSES * NL];
#pragma unroll
        for (int i = 0; i < PASSES * NL; ++i) w[i] = v[i];
        issue(rb + stride);                                      // the next block's loads fly while this one is stored
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                u32x4 x = w[(p * NL + c % NL)];
                x[0] ^= unsigned(c);
                __builtin_amdgcn_raw_buffer_store_b128(x, rs_out, unsigned((rb * 32 + p * ROWS + row) * WB + c * 16 * LPR + q * 16), 0, NT_STORE);
            }
        rb += stride;
    }
}


// The kernels' actual schedule: a ring of D chunks (64 B per pixel row each = two fragment loads) in flight ACROSS row-block boundaries, one chunk
// consumed (MFMAS dummy matrix instructions) and re-issued per step, the row block's stores behind its last chunk.  WAITFIX: wait for the chunk's two
// loads with an explicit count that allows the previous row block's stores to stay in flight (the compiler's own count, merged over the paths with
// and without an epilogue, drains them).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int RB, int WB, int MFMAS, int D>
__global__ __launch_bounds__(256) void ringed(const char* __restrict__ in, char* __restrict__ out, int nrb) {
    constexpr int CH = RB / 64, NS = WB / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in), 0, int(size_t(nrb) * 32 * RB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, int(size_t(nrb) * 32 * WB), 0x00020000);
    const int stride = gridDim.x * 4;
    int rb_l = blockIdx.x * 4 + wave, c_l = 0, rb_c = rb_l, c_c = 0;
    u32x4 ring[D][2];
    auto issue = [&](int s) {
        const unsigned off = rb_l < nrb ? unsigned((rb_l * 32 + r) * RB + c_l * 64 + hh * 16) : 0x80000000u;
        ring[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        ring[s][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + 32u, 0, 0);
        if (++c_l == CH) { c_l = 0; rb_l += stride; }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    u32x4 keep = {0, 0, 0, 0};
    while (rb_c < nrb) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const h8 a0 = __builtin_bit_cast(h8, ring[s][0]), a1 = __builtin_bit_cast(h8, ring[s][1]);
#pragma unroll
            for (int m = 0; m < MFMAS; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(m & 4 ? a1 : a0, m & 4 ? a0 : a1, acc[m & 3], 0, 0, 0);
            if (MFMAS == 0) { keep[0] ^= ring[s][0][0]; keep[1] ^= ring[s][1][1]; }
            issue(s);
            if (++c_c == CH) {
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    u32x4 x = keep;
                    if (MFMAS) { x[0] = __builtin_bit_cast(unsigned, acc[c & 3][c & 15]); x[1] = __builtin_bit_cast(unsigned, acc[(c + 1) & 3][(c + 5) & 15]); }
                    x[2] ^= unsigned(c);
                    __builtin_amdgcn_raw_buffer_store_b128(x, rs_out, unsigned((rb_c * 32 + r) * WB + c * 32 + hh * 16), 0, 0);
                }
                c_c = 0;
                rb_c += stride;
            }
        }
    }
}

template <int RB, int WB, int MFMAS, int D>
static int run_ring(const char* in, char* out, int npix, const char* what) {
    const int nrb = npix / 32;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wgs : {1024}) {
        ringed<RB, WB, MFMAS, D><<<wgs, 256>>>(in, out, nrb);
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 10; ++i) ringed<RB, WB, MFMAS, D><<<wgs, 256>>>(in, out, nrb);
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-34s read %3d B + write %3d B per pixel, ring of %d chunks, %d MFMAs per chunk, %4d workgroups: %6.1f us, %.2f TB/s\n", what, RB, WB, D, MFMAS, wgs,
               ms * 100.f, double(npix) * (RB + WB) / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}

template <int LPR, int RB, int WB, int NT_STORE>
static int run(const char* in, char* out, int npix, const char* what) {
    const int nrb = npix / 32;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wgs : {1024, 2048}) {
        pattern<LPR, RB, WB, NT_STORE><<<wgs, 256>>>(in, out, nrb);
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 10; ++i) pattern<LPR, RB, WB, NT_STORE><<<wgs, 256>>>(in, out, nrb);
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-34s read %3d B + write %3d B per pixel, %d lanes per row, %4d workgroups: %6.1f us, %.2f TB/s\n", what, RB, WB, LPR, wgs, ms * 100.f,
               double(npix) * (RB + WB) / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}

int main() {
    const int npix = 401408;                 // batch 128 x 56 x 56
    char *in, *out;
    const size_t bytes = size_t(npix) * 512;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {
        const size_t n = size_t(npix) * 256 / 16;
        copy16<<<4096, 256>>>(reinterpret_cast<const u32x4*>(in), reinterpret_cast<u32x4*>(out), n);
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 10; ++i) copy16<<<4096, 256>>>(reinterpret_cast<const u32x4*>(in), reinterpret_cast<u32x4*>(out), n);
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("plain copy, 16 B per lane, 103 MB each way: %.1f us, %.2f TB/s\n", ms * 100.f, double(n) * 32 / (ms / 10 * 1e-3) / 1e12);
    }
    printf("1x1 K=128 -> N=128 halfs (256 B in, 256 B out):\n");
    if (run<2, 256, 256, 0>(in, out, npix, "fragment shape (as the kernels)")) return 1;
    if (run<4, 256, 256, 0>(in, out, npix, "4 lanes per row")) return 1;
    if (run<8, 256, 256, 0>(in, out, npix, "8 lanes per row (whole lines)")) return 1;
    if (run<2, 256, 256, 2>(in, out, npix, "fragment shape, nt stores")) return 1;
    if (run_ring<256, 256, 0, 4>(in, out, npix, "the kernels' ring, no MFMA")) return 1;
    if (run_ring<256, 256, 8, 4>(in, out, npix, "the kernels' ring + MFMAs")) return 1;
    if (run_ring<256, 256, 8, 2>(in, out, npix, "ring of 2 + MFMAs")) return 1;
    printf("1x1 K=64 -> N=128 halfs (128 B in, 256 B out):\n");
    if (run_ring<128, 256, 0, 4>(in, out, npix, "the kernels' ring, no MFMA")) return 1;
    if (run_ring<128, 256, 8, 4>(in, out, npix, "the kernels' ring + MFMAs")) return 1;
    if (run<2, 128, 256, 0>(in, out, npix, "fragment shape")) return 1;
    if (run<8, 128, 256, 0>(in, out, npix, "8 lanes per row")) return 1;
    printf("1x1 K=256 -> N=128 halfs (512 B in, 256 B out):\n");
    if (run<2, 512, 256, 0>(in, out, npix, "fragment shape")) return 1;
    if (run<8, 512, 256, 0>(in, out, npix, "8 lanes per row")) return 1;
    if (run_ring<512, 256, 0, 4>(in, out, npix, "the kernels' ring, no MFMA")) return 1;
    if (run_ring<512, 256, 8, 4>(in, out, npix, "the kernels' ring + MFMAs")) return 1;
    printf("3x3 128 -> 32 halfs (256 B in, 64 B out):\n");
    if (run<2, 256, 64, 0>(in, out, npix, "fragment shape")) return 1;
    if (run<4, 256, 64, 0>(in, out, npix, "4 lanes per row")) return 1;
    return 0;
}
