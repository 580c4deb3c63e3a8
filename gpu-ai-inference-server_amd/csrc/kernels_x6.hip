// fp32 1x1 convolution on the bf16 matrix pipe with exactly split operands ("bf16x6"; fp32 precision mode, opt-in: IE_FP32_SPLIT=1).
//
// On gfx950 v_mfma_f32_32x32x2_f32 runs at the packed-fp32 VALU rate (157 TFLOP/s at 2.4 GHz, 129 at the ~2.0 GHz the part sustains) and
// does not overlap with VALU work; v_mfma_f32_32x32x16_bf16 is 16x faster and co-issues with it.  An fp32 number is the exact sum of three
// bf16 numbers (8 + 8 + 8 mantissa bits, same exponent range):  x = x0 + x1 + x2, x0 = trunc16(x), x1 = trunc16(x - x0), x2 = x - x0 - x1,
// and every product of two bf16 numbers is exact in fp32.  So
//     a * b  =  a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)  +  O(2^-24 |a b|)
// with the six partial products accumulated in fp32 by the MFMA (smallest first): the dropped terms are below one fp32 rounding of the
// product.  Measured (scripts/probes/bf16x6_probe.cpp): 198 cycles per 32x32x16 block against 517 for eight fp32 MFMAs (2.17x in time at
// the clocks each load sustains), max error of a 32x32x1024 product against float64 1.16e-6 of max|C| against 1.07e-6 for the fp32 MFMA.
//
// Kernel: a workgroup (4 waves) owns 32 * BMB pixels (BMB = 2 or 1) x 128 output channels; wave w owns the 32 channels of N block w and ALL the
// pixels (BMB accumulator tiles).  The activations go global -> registers (BN+ReLU prologue, split into three bf16 planes: VALU work
// that co-issues with the bf16 MFMAs) -> LDS in chunks of 32 input channels, double-buffered, one barrier per chunk; the pre-split
// weights (LaunchSplitWeightsX6, fragment-major: 1 KiB per load) stream from L2 into a register ring one chunk ahead.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// Three bf16 planes of W[cout][k] in the B-operand fragment order of v_mfma_f32_32x32x16_bf16:
//   element (((p * NB + nb) * KB + kb) * 64 + lane) * 8 + i  =  piece p of W[nb * 32 + (lane & 31)][kb * 16 + (lane >> 5) * 8 + i]
__global__ void split_weights_x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, const int Cout, const int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * K) return;
    const int co = idx / K, k = idx - co * K;
    const float x = w[idx];
    const unsigned h0 = f2u(x) & 0xffff0000u;
    const float r1 = x - u2f(h0);
    const unsigned h1 = f2u(r1) & 0xffff0000u;
    const float r2 = r1 - u2f(h1);
    const unsigned h2 = f2u(r2);
    const int NB = Cout >> 5, KB = K >> 4;
    const int nb = co >> 5, kb = k >> 4, lane = (co & 31) + 32 * ((k >> 3) & 1), i = k & 7;
    const size_t plane = size_t(Cout) * K;
    const size_t off = ((size_t(nb) * KB + kb) * 64 + lane) * 8 + i;
    dst[off] = static_cast<unsigned short>(h0 >> 16);
    dst[plane + off] = static_cast<unsigned short>(h1 >> 16);
    dst[2 * plane + off] = static_cast<unsigned short>(h2 >> 16);
    (void)NB;
}

hipError_t LaunchSplitWeightsX6(const float* w, void* dst, int Cout, int K, hipStream_t stream) {
    if ((Cout % 32) || (K % 16) || Cout <= 0 || K <= 0) return hipErrorInvalidValue;
    const int total = Cout * K;
    split_weights_x6_kernel<<<dim3((total + 255) / 256), dim3(256), 0, stream>>>(w, static_cast<unsigned short*>(dst), Cout, K);
    return hipGetLastError();
}

template <int BMB, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, BMB <= 2 ? 4 : 2))) void conv1x1_x6_kernel(const ConvArgs a) {
    constexpr int NT = 256, BM = 32 * BMB, KC = 32, PITCH = 40;      // LDS row pitch in bf16: 64 B of data + 16 B: the fragment reads are conflict-free
    constexpr int PLANE = BM * PITCH;                                  // bf16 elements per plane per buffer
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x6[];
    unsigned short* const sA = reinterpret_cast<unsigned short*>(smem_x6);      // [2 buffers][3 planes][BM][PITCH]

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.in.c, Cout = a.out.c, NCH = K / KC, KB = K >> 4, NB = Cout >> 5;
    const int M = a.out.n * a.out.h * a.out.w;
    const int m0 = blockIdx.x * BM;
    const int nb = blockIdx.y * 4 + wave;
    const int ipitch = int(a.in.sw), opitch = int(a.out.sw);

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w16), 0, int(int64_t(3) * Cout * K * 2), 0x00020000);
    const unsigned plane_bytes = unsigned(Cout) * unsigned(K) * 2u;

    // ---- activation staging: thread (row r0 + 32 u, 4-channel quad q of the chunk) ----
    const int q = tid & 7, r0 = tid >> 3;
    unsigned aoff[BMB];
#pragma unroll
    for (int u = 0; u < BMB; ++u) {
        const int p = m0 + r0 + 32 * u;
        aoff[u] = p < M ? unsigned(p * ipitch + q * 4) * 4u : OOB;
    }
    f32x4 rawA[BMB], rawB[BMB];
    auto issue_a = [&](f32x4 (&raw)[BMB], int c) {
#pragma unroll
        for (int u = 0; u < BMB; ++u)
            raw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, aoff[u], (c < NCH ? c : NCH - 1) * KC * 4, 0));      // past the end: the last chunk again, never consumed
    };
    auto stage_a = [&](const f32x4 (&raw)[BMB], int c, int buf) {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) {
            const int cc = c < NCH ? c : NCH - 1;
            sc = *reinterpret_cast<const f32x4*>(a.pre_scale + cc * KC + q * 4);
            sf = *reinterpret_cast<const f32x4*>(a.pre_shift + cc * KC + q * 4);
        }
        unsigned short* const base = sA + buf * 3 * PLANE + r0 * PITCH + q * 4;
#pragma unroll
        for (int u = 0; u < BMB; ++u) {
            unsigned h0[4], h1[4], h2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = raw[u][e];
                if constexpr (PRE) {
                    x = x * sc[e] + sf[e];
                    if (a.pre_relu) x = fmaxf(x, 0.f);
                }
                h0[e] = f2u(x);
                const float r1 = x - u2f(h0[e] & 0xffff0000u);
                h1[e] = f2u(r1);
                h2[e] = f2u(r1 - u2f(h1[e] & 0xffff0000u));
            }
            // pack the high halves of {e0, e1} and {e2, e3}: one v_perm_b32 per pair (bytes 2,3 of the first source low, of the second high)
            const u32x2 p0 = {__builtin_amdgcn_perm(h0[1], h0[0], 0x07060302u), __builtin_amdgcn_perm(h0[3], h0[2], 0x07060302u)};
            const u32x2 p1 = {__builtin_amdgcn_perm(h1[1], h1[0], 0x07060302u), __builtin_amdgcn_perm(h1[3], h1[2], 0x07060302u)};
            const u32x2 p2 = {__builtin_amdgcn_perm(h2[1], h2[0], 0x07060302u), __builtin_amdgcn_perm(h2[3], h2[2], 0x07060302u)};
            unsigned short* const o = base + u * 32 * PITCH;
            *reinterpret_cast<u32x2*>(o) = p0;
            *reinterpret_cast<u32x2*>(o + PLANE) = p1;
            *reinterpret_cast<u32x2*>(o + 2 * PLANE) = p2;
        }
    };
    // ---- weight ring: chunk c = k-blocks 2c, 2c+1, three planes ----
    u32x4 wA[2][3], wB[2][3];
    auto issue_w = [&](u32x4 (&wr)[2][3], int c) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                wr[kb][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, unsigned(lane) * 16u, int(unsigned(p) * plane_bytes + unsigned((nb * KB + 2 * (c < NCH ? c : NCH - 1) + kb) * 1024)), 0);
    };

    f32x16 acc[BMB];
#pragma unroll
    for (int mb = 0; mb < BMB; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;

    auto compute = [&](const u32x4 (&wr)[2][3], int buf) {
        const unsigned short* const A = sA + buf * 3 * PLANE + r * PITCH + hh * 8;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, wr[kb][0]), b1 = __builtin_bit_cast(bf16x8, wr[kb][1]), b2 = __builtin_bit_cast(bf16x8, wr[kb][2]);
#pragma unroll
            for (int mb = 0; mb < BMB; ++mb) {
                const unsigned short* const ap = A + mb * 32 * PITCH + kb * 16;
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap), a1 = *reinterpret_cast<const bf16x8*>(ap + PLANE), a2 = *reinterpret_cast<const bf16x8*>(ap + 2 * PLANE);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[mb], 0, 0, 0);      // smallest terms first
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[mb], 0, 0, 0);
            }
        }
    };

    // ---- prologue: chunk 0 staged, chunk 1 raw in flight ----
    issue_a(rawA, 0);
    issue_a(rawB, 1);
    issue_w(wA, 0);
    stage_a(rawA, 0, 0);
    issue_a(rawA, 2);
    __syncthreads();
    // Two chunks per trip (raw / ring parity).  Each half is ONE scheduling region: the MFMAs of the current chunk with the split of
    // the next one (VALU, co-issues with the bf16 pipe) threaded between them -- a wave issues in order, so without the interleave the
    // split would only start after the last MFMA had issued.
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 12 * BMB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // up to six VALU
        }
    };
    int c = 0;
    for (; c + 1 < NCH; c += 2) {
        issue_w(wB, c + 1);
        compute(wA, 0);
        stage_a(rawB, c + 1, 1);
        issue_a(rawB, c + 3);
        interleave();
        __syncthreads();
        issue_w(wA, c + 2);
        compute(wB, 1);
        stage_a(rawA, c + 2, 0);
        issue_a(rawA, c + 4);
        interleave();
        __syncthreads();
    }
    if (NCH & 1) compute(wA, 0);                        // odd chunk count: the last chunk sits in buffer 0, its weights in wA

    // ---- epilogue: lane (channel r of the N block, hh) holds 16 pixel rows of each accumulator tile ----
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * opitch + Cout) * 4), 0x00020000);
    const int n = nb * 32 + r;
    const float bq = a.bias != nullptr ? a.bias[n] : 0.f;
#pragma unroll
    for (int mb = 0; mb < BMB; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            float v = acc[mb][e] + bq;
            if (a.relu) v = fmaxf(v, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, m < M ? unsigned(m * opitch + n) * 4u : OOB, 0, 0);
        }
    (void)NB;
}

static size_t x6_lds_bytes(int bmb) { return size_t(2) * 3 * 32 * bmb * 40 * 2; }

// tile 0: 64 pixels per workgroup, tile 1: 32 (128 pixels: 255 VGPRs and no faster than 64 in scripts/probes/x6_probe.cpp)
bool ConvX6Eligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvX6Tiles) return false;
    if (a.in.f16 || a.out.f16 || a.in.f8 || a.out.f8 || a.w16 == nullptr || a.res.p != nullptr) return false;
    if (a.kh != 1 || a.kw != 1 || a.sh != 1 || a.sw != 1 || a.pt != 0 || a.pl != 0) return false;
    if (a.out.h != a.in.h || a.out.w != a.in.w || a.out.n != a.in.n) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || (a.in.c % 32) || a.in.c < 32 || (a.out.c % 128)) return false;
    if ((a.in.sw % 4) || a.in.sh != a.in.w * a.in.sw || a.in.sn != a.in.h * a.in.sh || (reinterpret_cast<uintptr_t>(a.in.p) & 15)) return false;
    if (a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh || (reinterpret_cast<uintptr_t>(a.out.p) & 3)) return false;
    if (reinterpret_cast<uintptr_t>(a.w16) & 15) return false;
    if (a.pre_scale && ((reinterpret_cast<uintptr_t>(a.pre_scale) & 15) || (reinterpret_cast<uintptr_t>(a.pre_shift) & 15))) return false;
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    if (M * a.in.sw * 4 >= (int64_t(1) << 31) || M * a.out.sw * 4 >= (int64_t(1) << 31) || int64_t(3) * a.out.c * a.in.c * 2 >= (int64_t(1) << 31)) return false;
    return true;
}

template <int BMB>
static hipError_t launch_x6(const ConvArgs& a, hipStream_t stream) {
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    const dim3 grid(unsigned((M + 32 * BMB - 1) / (32 * BMB)), unsigned(a.out.c / 128));
    if (a.pre_scale) conv1x1_x6_kernel<BMB, true><<<grid, dim3(256), x6_lds_bytes(BMB), stream>>>(a);
    else conv1x1_x6_kernel<BMB, false><<<grid, dim3(256), x6_lds_bytes(BMB), stream>>>(a);
    return hipGetLastError();
}

hipError_t LaunchConvX6(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvX6Eligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = 4 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + a.in.c);
    return tile == 0 ? launch_x6<2>(a, stream) : launch_x6<1>(a, stream);
}

hipError_t InitKernelsX6() {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_x6_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_x6_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_x6_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)) != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_x6_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}

}  // namespace ie
