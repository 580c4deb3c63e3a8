// "Direct split-K" convolution for layers whose output grid is too small to fill the chip (DenseNet dense blocks 3-4 at batch
// 32: M = 6272 / 1568 pixels, K up to 1152).
//
// Those layers are latency-bound in the tiled implicit GEMM: a workgroup walks its K-tiles one after the other, and every
// K-tile is a global -> register -> LDS -> barrier round trip (~2 us) that a one-deep prefetch cannot hide; 80 such launches
// were 40 % of the fp32 forward.  Here NOTHING is staged through LDS and NOTHING is sequential:
//   * a workgroup owns a 32-pixel x 32*TN-channel output tile and splits K (= taps x Cin) over its WAVES waves;
//   * both MFMA operands are "row, 4 consecutive k": 16 contiguous bytes of an NHWC pixel row (activations) or of a
//     [Cout][kh][kw][Cin] weight row - every lane loads its own A and B fragments straight from global memory, ALL of them up
//     front (up to MAXC chunks of 16 channels per wave, padding taps read an out-of-range offset = zeros), so the whole
//     operand fetch costs ONE memory latency;
//   * the waves' partial tiles are summed through LDS (deterministic order), then bias / ReLU / store.
// The BN+ReLU prologue of pre-activation convs is applied to the A fragments in registers (scale/shift staged in LDS while
// the operand loads are in flight); zero padding is applied after it.
// fp32 (v_mfma_f32_32x32x2_f32, one 16-byte fragment feeds four MFMAs) and fp16 (v_mfma_f32_32x32x16_f16, chunks of 32
// channels) variants share the structure.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

struct DirectGeom {
    int cpt;        // chunks per tap = Cin / CW
    int total;      // taps * cpt
};

// HALF = false: fp32 operands, chunk = 16 channels (two 16-byte loads of 4 floats per lane: k = 8q + 4hh + e)
// HALF = true : fp16 operands, chunk = 32 channels (two 16-byte loads of 8 halfs per lane:  k = 16q + 8hh + j)
template <bool HALF, int TN, int WAVES, int MAXC, bool PRE>
__global__ __launch_bounds__(64 * WAVES) void conv_direct_kernel(const ConvArgs a, const DirectGeom g) {
    constexpr int NT = 64 * WAVES, BN = 32 * TN;
    constexpr int CW = HALF ? 32 : 16;                 // channels per chunk
    constexpr int ESZ = HALF ? 2 : 4;
    constexpr int PP = BN + 4;                         // partial-tile row pitch (floats)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_direct[];
    float* const sPart = reinterpret_cast<float*>(smem_direct);                    // [WAVES][32][PP]
    unsigned char* const sPre = smem_direct + size_t(WAVES) * 32 * PP * 4;          // scale [Cin], shift [Cin] (float or half)

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.in.c, H = a.in.h, W = a.in.w, OH = a.out.h, OW = a.out.w, Cout = a.out.c;
    const int M = a.out.n * OH * OW;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * BN;
    const int KK = a.kh * a.kw;

    // this lane's output pixel
    const int m = m0 + r;
    const bool mok = m < M;
    const int mm = mok ? m : 0;
    const int b = mm / (OH * OW);
    const int rem = mm - b * (OH * OW);
    const int oy = rem / OW, ox = rem - oy * OW;
    const int iy0 = oy * a.sh - a.pt, ix0 = ox * a.sw - a.pl;

    const int cb = int(int64_t(g.total) * wave / WAVES), ce = int(int64_t(g.total) * (wave + 1) / WAVES);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const void* const wbase = HALF ? a.w16 : static_cast<const void*>(a.w);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wbase), 0, Cout * KK * Cin * ESZ, 0x00020000);

    // ---- every operand fragment of this wave's K slice, requested at once ----
    u32x4 A[MAXC][2], B[MAXC][TN][2];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = cb + i;
        const bool live = ch < ce;                     // wave-uniform
        const int chc = live ? ch : cb;
        const int tap = chc / g.cpt, c0 = (chc - tap * g.cpt) * CW;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        const int iy = iy0 + ky, ix = ix0 + kx;
        const bool ok = live && mok && unsigned(iy) < unsigned(H) && unsigned(ix) < unsigned(W);
        okmask |= ok ? (1u << i) : 0u;
        const unsigned offA = ok ? unsigned(int(b * a.in.sn + iy * a.in.sh + ix * a.in.sw) + c0 + hh * (CW / 4)) * unsigned(ESZ) : OOB;
        A[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, offA, 0, 0);
        A[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, offA + unsigned(CW / 2 * ESZ), 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + j * 32 + r;
            const unsigned offB = (live && n < Cout) ? unsigned((n * KK + tap) * Cin + c0 + hh * (CW / 4)) * unsigned(ESZ) : OOB;
            B[i][j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, offB, 0, 0);
            B[i][j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, offB + unsigned(CW / 2 * ESZ), 0, 0);
        }
    }

    // ---- BN scale/shift of the prologue -> LDS while the operands are in flight ----
    if constexpr (PRE) {
        if constexpr (HALF) {
            const u32x4* const ps = static_cast<const u32x4*>(a.pre_scale16);
            const u32x4* const pt = static_cast<const u32x4*>(a.pre_shift16);
            u32x4* const ds = reinterpret_cast<u32x4*>(sPre);
            u32x4* const dt = reinterpret_cast<u32x4*>(sPre + Cin * 2);
            for (int i = tid; i < Cin / 8; i += NT) { ds[i] = ps[i]; dt[i] = pt[i]; }
        } else {
            const u32x4* const ps = reinterpret_cast<const u32x4*>(a.pre_scale);
            const u32x4* const pt = reinterpret_cast<const u32x4*>(a.pre_shift);
            u32x4* const ds = reinterpret_cast<u32x4*>(sPre);
            u32x4* const dt = reinterpret_cast<u32x4*>(sPre + Cin * 4);
            for (int i = tid; i < Cin / 4; i += NT) { ds[i] = ps[i]; dt[i] = pt[i]; }
        }
        __syncthreads();
    }

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        if (cb + i < ce) {                             // wave-uniform: chunks past this wave's slice are skipped
            const int ch = cb + i;
            const int tap = ch / g.cpt, c0 = (ch - tap * g.cpt) * CW;
            const bool ok = (okmask >> i) & 1u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if constexpr (HALF) {
                    h8 av = __builtin_bit_cast(h8, A[i][q]);
                    if constexpr (PRE) {
                        const h8 s = *reinterpret_cast<const h8*>(sPre + (c0 + q * 16 + hh * 8) * 2);
                        const h8 t = *reinterpret_cast<const h8*>(sPre + Cin * 2 + (c0 + q * 16 + hh * 8) * 2);
                        av = av * s + t;
                        if (a.pre_relu) av = __builtin_elementwise_max(av, h8{});
                        if (!ok) av = h8{};                       // zero padding applies AFTER the activation
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, B[i][j][q]), av, acc[j], 0, 0, 0);
                } else {
                    f32x4 av = __builtin_bit_cast(f32x4, A[i][q]);
                    if constexpr (PRE) {
                        const f32x4 s = *reinterpret_cast<const f32x4*>(sPre + (c0 + q * 8 + hh * 4) * 4);
                        const f32x4 t = *reinterpret_cast<const f32x4*>(sPre + Cin * 4 + (c0 + q * 8 + hh * 4) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = av[e] * s[e] + t[e];
                            av[e] = ok ? (a.pre_relu ? fmaxf(x, 0.f) : x) : 0.f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const f32x4 bv = __builtin_bit_cast(f32x4, B[i][j][q]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[e], av[e], acc[j], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- cross-wave reduction through LDS (D = W x A^T: lane (r, hh) holds pixel r, channels 8g + 4hh + q of each n block) ----
    float* const mine = sPart + (wave * 32 + r) * PP;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<f32x4*>(mine + j * 32 + 8 * gq + 4 * hh) = f32x4{acc[j][4 * gq], acc[j][4 * gq + 1], acc[j][4 * gq + 2], acc[j][4 * gq + 3]};
    __syncthreads();
    // thread t sums channel pair (t % (BN/2))*2 of pixel t / (BN/2) over the waves, in wave order
    const int esz = a.out.f16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * a.out.sw + Cout) * esz), 0x00020000);
    for (int idx = tid; idx < 32 * (BN / 2); idx += NT) {
        const int p = idx / (BN / 2), c2 = (idx - p * (BN / 2)) * 2;
        f32x2 v = {0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const f32x2 x = *reinterpret_cast<const f32x2*>(sPart + (w * 32 + p) * PP + c2);
            v[0] += x[0];
            v[1] += x[1];
        }
        const int n = n0 + c2;
        if (a.bias != nullptr) {
            v[0] += a.bias[n < Cout ? n : Cout - 1];
            v[1] += a.bias[n + 1 < Cout ? n + 1 : Cout - 1];
        }
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
        const bool pok = m0 + p < M;
        const unsigned rowoff = pok ? unsigned((m0 + p) * int(a.out.sw) * esz) : OOB;      // Cout is even (eligibility)
        if (a.out.f16) {
            const h2 hv = {_Float16(v[0]), _Float16(v[1])};
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hv), rs_out, n < Cout ? rowoff + unsigned(n * 2) : OOB, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v), rs_out,
                                                  n < Cout ? rowoff + unsigned(n * 4) : OOB, 0, 0);
        }
    }
}

struct DirectTile { int tn, waves, maxc; };
constexpr DirectTile kDirectTiles[kNumConvDirectTiles] = {{1, 8, 8}, {1, 16, 4}, {1, 9, 8}, {1, 4, 8}, {1, 12, 6}, {2, 8, 4}};

static size_t direct_lds_bytes(int tile, int Cin, bool half, bool pre) {
    const DirectTile t = kDirectTiles[tile];
    return size_t(t.waves) * 32 * (32 * t.tn + 4) * 4 + (pre ? size_t(2) * Cin * (half ? 2 : 4) : 0);
}

bool ConvDirectEligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvDirectTiles) return false;
    const bool half = a.in.f16 != 0;
    const int cw = half ? 32 : 16;
    if (a.in.sc != 1 || a.out.sc != 1 || (a.in.c % cw) || a.kh * a.kw > 49) return false;
    if (half ? (a.w16 == nullptr) : (a.w == nullptr)) return false;
    const int align = half ? 8 : 4;                    // 16-byte fragments
    if ((a.in.sw % align) || (a.in.sh % align) || (a.in.sn % align) || (reinterpret_cast<uintptr_t>(a.in.p) & 15)) return false;
    if (reinterpret_cast<uintptr_t>(half ? a.w16 : static_cast<const void*>(a.w)) & 15) return false;
    if ((a.out.c & 1) || (a.out.sw & 1) || (reinterpret_cast<uintptr_t>(a.out.p) & 7)) return false;
    if (a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh) return false;    // output pixels at a constant pitch
    if (a.pre_scale) {
        if (half && (a.pre_scale16 == nullptr || a.pre_shift16 == nullptr)) return false;
        if ((reinterpret_cast<uintptr_t>(a.pre_scale) & 15) || (reinterpret_cast<uintptr_t>(a.pre_shift) & 15)) return false;
        if (half && ((reinterpret_cast<uintptr_t>(a.pre_scale16) & 15) || (reinterpret_cast<uintptr_t>(a.pre_shift16) & 15))) return false;
    }
    const int esz = half ? 2 : 4;
    const int64_t in_span = int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + a.in.c;
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    if (in_span * esz >= (int64_t(1) << 31) || M * a.out.sw * 4 >= (int64_t(1) << 31) ||
        int64_t(a.out.c) * a.kh * a.kw * a.in.c * esz >= (int64_t(1) << 31))
        return false;
    const DirectTile t = kDirectTiles[tile];
    const int total = a.kh * a.kw * (a.in.c / cw);
    if (total > t.waves * t.maxc || total < t.waves) return false;                         // every wave gets 1..MAXC chunks
    if (t.tn > 1 && a.out.c <= 32) return false;
    if (M > 65536) return false;                       // big layers belong to the tiled / weights-stationary kernels
    return direct_lds_bytes(tile, a.in.c, half, a.pre_scale != nullptr) <= size_t(160) * 1024;
}

template <bool HALF, int T>
static hipError_t launch_direct_t(const ConvArgs& a, hipStream_t stream) {
    constexpr DirectTile t = kDirectTiles[T];
    DirectGeom g;
    g.cpt = a.in.c / (HALF ? 32 : 16);
    g.total = a.kh * a.kw * g.cpt;
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    const dim3 grid(unsigned((M + 31) / 32), unsigned((a.out.c + 32 * t.tn - 1) / (32 * t.tn)));
    const size_t lds = direct_lds_bytes(T, a.in.c, HALF, a.pre_scale != nullptr);
    if (a.pre_scale) conv_direct_kernel<HALF, t.tn, t.waves, t.maxc, true><<<grid, dim3(64 * t.waves), lds, stream>>>(a, g);
    else conv_direct_kernel<HALF, t.tn, t.waves, t.maxc, false><<<grid, dim3(64 * t.waves), lds, stream>>>(a, g);
    return hipGetLastError();
}

hipError_t LaunchConvDirect(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvDirectEligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    const int esz = a.in.f16 ? 2 : 4;
    a.in_bytes = esz * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + int64_t(a.in.c - 1) + 1);
#define IE_DIR(T) \
    case T: return a.in.f16 ? launch_direct_t<true, T>(a, stream) : launch_direct_t<false, T>(a, stream);
    switch (tile) {
        IE_DIR(0) IE_DIR(1) IE_DIR(2) IE_DIR(3) IE_DIR(4) IE_DIR(5)
        default: return hipErrorInvalidValue;
    }
#undef IE_DIR
}

template <bool HALF, int T>
static hipError_t init_direct_t() {
    constexpr DirectTile t = kDirectTiles[T];
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<HALF, t.tn, t.waves, t.maxc, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<HALF, t.tn, t.waves, t.maxc, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t InitKernelsDirect() {
    hipError_t e;
#define IE_DIRI(T)                                                     \
    if ((e = init_direct_t<false, T>()) != hipSuccess) return e;       \
    if ((e = init_direct_t<true, T>()) != hipSuccess) return e;
    IE_DIRI(0) IE_DIRI(1) IE_DIRI(2) IE_DIRI(3) IE_DIRI(4) IE_DIRI(5)
#undef IE_DIRI
    return hipSuccess;
}

}  // namespace ie
