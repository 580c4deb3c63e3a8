// Weights-stationary 1x1 convolution for the fp16 precision mode (DenseNet's BN -> ReLU -> 1x1 bottlenecks and transitions).
//
// In fp16 the 1x1 layers are HBM-bound, and what the tiled implicit GEMM spends its time on is not the MFMA but
// moving the ACTIVATION tile through LDS (ds_write_b128 runs at ~79 B/clk/CU, MI355X_MICROARCH.md §LDS) and a barrier per
// K-tile.  For a 1x1 / stride 1 conv over NHWC none of that is needed: the A operand of v_mfma_f32_32x32x16_f16 is "row m, 8
// consecutive k", which for pixel m is 16 contiguous bytes of HBM, so every lane loads its own fragments straight into
// registers (buffer_load_b128, out-of-range rows read zeros).  Only the WEIGHT slice [BN][K] (and the folded-BN scale/shift of
// the prologue) lives in LDS, loaded ONCE per workgroup; the workgroups are persistent and every wave then streams 32-pixel
// row blocks on its own - no barrier after the preamble, activations read from HBM exactly once per N-tile, weights re-read
// only from LDS.  A ring of register chunks (32 channels each) keeps D chunks per wave in flight across row-block boundaries.
//
// D = W x A^T (operands swapped like conv_igemm_f16_kernel): a lane owns one pixel and quads of consecutive channels; with
// v_permlane32_swap the two half-waves exchange quads so that each lane stores 8 consecutive halfs (16 bytes).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// Register chunks (32 channels x 32 pixels = 2 KiB each) a wave keeps in flight: 4 beside 64 accumulator registers, 6 otherwise.
constexpr int ws_ring_depth(int tn) { return tn >= 4 ? 4 : 6; }

template <int TN, int WAVES, bool PRE>
__global__ __launch_bounds__(64 * WAVES) void conv1x1_ws_f16_kernel(const ConvArgs a) {
    constexpr int NT = 64 * WAVES, BN = 32 * TN, D = ws_ring_depth(TN);
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_ws[];
    const int K = a.in.c, P = K + 8;                  // row pitch of the weight image: (2K+16)/16 odd for K % 16 == 0
    _Float16* const sB = smem_ws;                     // [BN][P]
    _Float16* const sS = sB + BN * P;                 // [K] prologue scale
    _Float16* const sT = sS + K;                      // [K] prologue shift
    float* const sBias = reinterpret_cast<float*>(sT + K);   // [BN] (zeros without a bias)
    const int Cout = a.out.c;
    const int M = a.out.n * a.out.h * a.out.w;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- preamble: weight slice (and BN scale/shift) -> LDS, once per workgroup ----
    {
        const _Float16* const w = static_cast<const _Float16*>(a.w16);
        const int k8 = K >> 3;
        for (int idx = tid; idx < BN * k8; idx += NT) {
            const int row = idx / k8, ck = idx - row * k8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n0 + row < Cout) v = *reinterpret_cast<const u32x4*>(w + int64_t(n0 + row) * K + ck * 8);
            *reinterpret_cast<u32x4*>(sB + row * P + ck * 8) = v;
        }
        for (int idx = tid; idx < BN; idx += NT) sBias[idx] = (a.bias != nullptr && n0 + idx < Cout) ? a.bias[n0 + idx] : 0.f;
        if constexpr (PRE) {
            const _Float16* const ps = static_cast<const _Float16*>(a.pre_scale16);
            const _Float16* const pt = static_cast<const _Float16*>(a.pre_shift16);
            for (int idx = tid; idx < k8; idx += NT) {
                *reinterpret_cast<u32x4*>(sS + idx * 8) = *reinterpret_cast<const u32x4*>(ps + idx * 8);
                *reinterpret_cast<u32x4*>(sT + idx * 8) = *reinterpret_cast<const u32x4*>(pt + idx * 8);
            }
        }
    }
    __syncthreads();

    const int nrb = (M + 31) >> 5;                    // 32-pixel row blocks
    const int stride = gridDim.x * WAVES;
    const int CH = K >> 5;                            // 32-channel chunks per row block
    const int ipitch = int(a.in.sw);
    constexpr unsigned OOB = 0x80000000u;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // load stream (rb_l, c_l) runs D chunks ahead of the compute stream (rb_c, c_c)
    int rb_l = blockIdx.x * WAVES + wave, c_l = 0;
    int rb_c = rb_l, c_c = 0;
    // Ring of register chunks with STATIC slots (the chunk loop below is unrolled by D, so the compiler's waitcnt pass sees the
    // loads in issue order and waits with exact counts, vmcnt(2(D-1)), instead of draining the ring).
    u32x4 ring[D][2];
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    auto issue = [&](int slot) {
        const int m = rb_l * 32 + r;
        const unsigned off = (rb_l < nrb && m < M) ? unsigned(m * ipitch + c_l * 32 + hh * 8) * 2u : OOB;
        ring[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        ring[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + 32u, 0, 0);
        if (++c_l == CH) { c_l = 0; rb_l += stride; }
    };
    auto compute = [&](const u32x4 c0, const u32x4 c1) {
        const int cbase = c_c * 32 + hh * 8;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8 av = __builtin_bit_cast(h8, kk == 0 ? c0 : c1);
            if constexpr (PRE) {
                const h8 s = *reinterpret_cast<const h8*>(sS + cbase + kk * 16);
                const h8 t = *reinterpret_cast<const h8*>(sT + cbase + kk * 16);
                av = av * s + t;
                if (a.pre_relu) av = __builtin_elementwise_max(av, h8{});
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const h8 b = *reinterpret_cast<const h8*>(sB + (j * 32 + r) * P + cbase + kk * 16);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, av, acc[j], 0, 0, 0);
            }
        }
    };
    // Stores go through a buffer descriptor with 32-bit offsets: rows past M and channels past Cout get an out-of-range
    // offset and are dropped by the hardware - no 64-bit address math, no divergent store branches.
    const bool store_half = a.out.f16 != 0;
    const int esz = store_half ? 2 : 4;
    const int opitch = int(a.out.sw);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * opitch + Cout) * esz), 0x00020000);
    auto epilogue = [&]() {
        const int m = rb_c * 32 + r;
        const unsigned rowoff = m < M ? unsigned(m * opitch * esz) : OOB;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = acc[j][e];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + j * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[4 * g + q] += bq[q];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (a.relu) v[e] = fmaxf(v[e], 0.f);
                acc[j][e] = 0.f;
            }
            if (store_half) {
                // quads (g, hh) -> after the half-wave exchange lane (r, hh) holds channels 8*(2gp+hh) .. +7 of pixel r
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const h4 qa = {_Float16(v[8 * gp + 0]), _Float16(v[8 * gp + 1]), _Float16(v[8 * gp + 2]), _Float16(v[8 * gp + 3])};
                    const h4 qb = {_Float16(v[8 * gp + 4]), _Float16(v[8 * gp + 5]), _Float16(v[8 * gp + 6]), _Float16(v[8 * gp + 7])};
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 xa = __builtin_bit_cast(u32x2, qa), xb = __builtin_bit_cast(u32x2, qb);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(xa[0], xb[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(xa[1], xb[1], false, false);
                    const int n = n0 + j * 32 + 8 * (2 * gp + hh);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rs_out, n < Cout ? rowoff + unsigned(n * 2) : OOB, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + j * 32 + 8 * g + 4 * hh;
                    const f32x4 q4 = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q4), rs_out, n < Cout ? rowoff + unsigned(n * 4) : OOB, 0, 0);
                }
            }
        }
    };

#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);
    // The ring slot is wave-uniform: a scalar jump copies the slot's registers to the operand registers (the load has to
    // have landed by then anyway) and another one re-issues into the slot, so the MFMA body and the epilogue exist once.
    // Row blocks end wherever the chunk count says (CH need not divide D), so the epilogue is inlined behind a wave-uniform
    // branch at each of the D positions.  Past the last row block the stream just runs on zeros: every load and store of a
    // row >= M carries an out-of-range offset, so the tail needs no guards.
    while (rb_c < nrb) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            compute(ring[s][0], ring[s][1]);
            issue(s);
            if (++c_c == CH) {
                epilogue();
                c_c = 0;
                rb_c += stride;
            }
        }
    }
}

struct WsTile { int tn, waves; };
constexpr WsTile kWsTiles[kNumConvWsTiles] = {{4, 8}, {4, 4}, {2, 8}, {2, 4}, {1, 8}, {1, 4}};

static size_t ws_lds_bytes(int tn, int K) { return size_t(32 * tn * (K + 8) + 2 * K) * sizeof(_Float16) + size_t(32 * tn) * sizeof(float); }

bool ConvWsEligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvWsTiles) return false;
    if (!a.in.f16 || a.w16 == nullptr || a.kh != 1 || a.kw != 1 || a.sh != 1 || a.sw != 1 || a.pt != 0 || a.pl != 0) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || a.in.h != a.out.h || a.in.w != a.out.w) return false;
    if ((a.in.c & 31) || (a.in.sw & 7) || (reinterpret_cast<uintptr_t>(a.in.p) & 15) || (reinterpret_cast<uintptr_t>(a.w16) & 15)) return false;
    if (a.in.sh != a.in.w * a.in.sw || a.in.sn != a.in.h * a.in.sh) return false;           // pixels at a constant pitch
    if (a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh) return false;
    if (a.pre_scale && (a.pre_scale16 == nullptr || a.pre_shift16 == nullptr || (reinterpret_cast<uintptr_t>(a.pre_scale16) & 15) ||
                        (reinterpret_cast<uintptr_t>(a.pre_shift16) & 15)))
        return false;
    const int64_t M = int64_t(a.in.n) * a.in.h * a.in.w;
    if (M * a.in.sw * 2 >= (int64_t(1) << 31) || M * a.out.sw * 4 >= (int64_t(1) << 31)) return false;
    // 16-byte stores: 8 halfs / 4 floats per lane
    if (a.out.f16 ? ((a.out.c % 8) || (a.out.sw % 8)) : ((a.out.c % 4) || (a.out.sw % 4))) return false;
    if (reinterpret_cast<uintptr_t>(a.out.p) % 16) return false;
    const WsTile t = kWsTiles[tile];
    if (ws_lds_bytes(t.tn, a.in.c) > size_t(160) * 1024) return false;
    if (t.tn > 1 && a.out.c <= 32 * (t.tn / 2)) return false;                               // do not waste MFMA rows on padding
    return true;
}

template <int TN, int WAVES, bool PRE>
static hipError_t launch_ws_t(const ConvArgs& a, hipStream_t stream) {
    const int64_t M = int64_t(a.in.n) * a.in.h * a.in.w;
    const int nrb = int((M + 31) / 32);
    const size_t lds = ws_lds_bytes(TN, a.in.c);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount;
    }
    // resident workgroups: LDS and 2048 threads per CU; then the smallest grid with the same number of row blocks per wave
    int per_cu = int((size_t(160) * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2048 / (64 * WAVES) ? 2048 / (64 * WAVES) : per_cu);
    if (per_cu > 4) per_cu = 4;
    const int slots = cus * per_cu;
    const int iters = (nrb + slots * WAVES - 1) / (slots * WAVES);
    int gx = (nrb + iters * WAVES - 1) / (iters * WAVES);
    gx = (gx + 7) & ~7;                      // same x -> same XCD for the N-tiles of one row range
    const int gy = (a.out.c + 32 * TN - 1) / (32 * TN);
    conv1x1_ws_f16_kernel<TN, WAVES, PRE><<<dim3(gx, gy), dim3(64 * WAVES), lds, stream>>>(a);
    return hipGetLastError();
}

hipError_t LaunchConvWs1x1F16(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvWsEligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = 2 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + int64_t(a.in.c - 1) + 1);
#define IE_WS(T, TN, W) \
    case T: return a.pre_scale ? launch_ws_t<TN, W, true>(a, stream) : launch_ws_t<TN, W, false>(a, stream);
    switch (tile) {
        IE_WS(0, 4, 8) IE_WS(1, 4, 4) IE_WS(2, 2, 8) IE_WS(3, 2, 4) IE_WS(4, 1, 8) IE_WS(5, 1, 4)
        default: return hipErrorInvalidValue;
    }
#undef IE_WS
}

hipError_t InitKernelsWs() {
    hipError_t e;
#define IE_WSI(TN, W)                                                                                                                       \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f16_kernel<TN, W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024)) != hipSuccess) return e;                                                                       \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f16_kernel<TN, W, false>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024)) != hipSuccess) return e;
    IE_WSI(4, 8) IE_WSI(4, 4) IE_WSI(2, 8) IE_WSI(2, 4) IE_WSI(1, 8) IE_WSI(1, 4)
#undef IE_WSI
    return hipSuccess;
}

}  // namespace ie
