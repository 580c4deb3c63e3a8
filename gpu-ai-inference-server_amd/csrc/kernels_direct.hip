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

#include <algorithm>
#include <cstdlib>

#include "env.h"
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

// ------------------------------------------------------------------------------------------------------------------------
// "Window" variant (fp32, stride 1, output grid == input grid) on 16 x 16 x 4 MFMA tiles.
//
// In-kernel stamps on the kernel above (M = 1568, 3x3, K = 1152: 49 workgroups) showed two things: the 16-byte row-strided fragment
// loads of a workgroup (~300 KB) queue behind ONE CU's texture addresser (wave 0 done after 4 us, the last wave after 9 us), and
// even with every operand on chip the 576 32x32x2 MFMAs of a 32-pixel x 32-channel tile keep that CU busy for 4.4 us while 80 % of
// the chip idles.  So this variant
//   * works on 16-pixel x 16*TN-channel tiles (v_mfma_f32_16x16x4_f32): 4x as many workgroups, a quarter of the MFMA time each,
//   * reads weights from a fragment-major mirror of the blob (LaunchPermuteWeightsFrag: one fragment load = 1 KiB contiguous),
//   * copies the activations the 16 output pixels touch - ONE contiguous run of 16 + (kh-1)*W + (kw-1) NHWC pixel rows (taps are
//     shifts of the flattened pixel index; rows that belong to the neighbouring line / image are masked per lane, which is exactly
//     the padding rule) - to LDS with coalesced loads, BN+ReLU prologue applied on the way, and reads fragments from there.
// K is still split over the waves and reduced through LDS in wave order (the partial tiles reuse the window's storage).
// MFMA layout: D = W x A^T, lane (r, gk) supplies row r (an output channel resp. a pixel) and the 4 channels c0 + 4*gk + e, and ends
// up with channels 4*gk .. 4*gk+3 of pixel r.
// ------------------------------------------------------------------------------------------------------------------------
template <int TN, int WAVES, int MAXC, bool PRE>
__global__ __launch_bounds__(64 * WAVES) void conv_win_kernel(const ConvArgs a, const DirectGeom g) {
    constexpr int NT = 64 * WAVES, BN = 16 * TN, PP = BN + 4;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_direct[];
    float* const sWin = reinterpret_cast<float*>(smem_direct);                     // [npx][P]
    float* const sPart = reinterpret_cast<float*>(smem_direct);                    // [WAVES][16][PP], after the window is dead

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.in.c, H = a.in.h, W = a.in.w, Cout = a.out.c;
    const int P = Cin + 4;                             // rows 4 banks apart: the 16 rows of a quarter-wave cover all 64 banks
    const int M = a.out.n * H * W;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * BN;
    const int ipitch = int(a.in.sw);

    // ---- this wave's weight fragments: fragment-major, 1 KiB per load, all in flight at once ----
    const int cb = int(int64_t(g.total) * wave / WAVES), ce = int(int64_t(g.total) * (wave + 1) / WAVES);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, Cout * a.kh * a.kw * Cin * 4, 0x00020000);
    u32x4 B[MAXC][TN];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = cb + i;
#pragma unroll
        for (int j = 0; j < TN; ++j)
            B[i][j] = __builtin_amdgcn_raw_buffer_load_b128(
                rs_w, ch < ce ? unsigned(((int(blockIdx.y) * TN + j) * g.total + ch) * 64 + lane) * 16u : OOB, 0, 0);
    }

    // ---- activation window -> LDS (coalesced rows; prologue applied here, padding is masked at fragment time) ----
    {
        constexpr int U = 4;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
        const int c4n = Cin >> 2;
        const int p_lo = m0 - a.pt * W - a.pl;
        const int npx = 16 + (a.kh - 1) * W + (a.kw - 1);
        const int items = npx * c4n;
        for (int idx0 = tid; idx0 < items; idx0 += U * NT) {
            u32x4 v[U];
            int row[U], c4[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = idx0 + u * NT;
                row[u] = idx / c4n;
                c4[u] = idx - row[u] * c4n;
                const int p = p_lo + row[u];
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (idx < items && p >= 0 && p < M) ? unsigned(p * ipitch + c4[u] * 4) * 4u : OOB, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (idx0 + u * NT < items) {
                    f32x4 x = __builtin_bit_cast(f32x4, v[u]);
                    if constexpr (PRE) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.pre_scale + c4[u] * 4);
                        const f32x4 sf = *reinterpret_cast<const f32x4*>(a.pre_shift + c4[u] * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float y = x[e] * sc[e] + sf[e];
                            x[e] = a.pre_relu ? fmaxf(y, 0.f) : y;
                        }
                    }
                    *reinterpret_cast<f32x4*>(sWin + row[u] * P + c4[u] * 4) = x;
                }
            }
        }
    }
    __syncthreads();

    // this lane's output pixel (same grid as the input)
    const int m = m0 + r;
    const bool mok = m < M;
    const int rem = (mok ? m : 0) % (H * W);
    const int oy = rem / W, ox = rem - oy * W;

    f32x4 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        if (cb + i < ce) {                             // wave-uniform
            const int ch = cb + i;
            const int tap = ch / g.cpt, c0 = (ch - tap * g.cpt) * 16;
            const int ky = tap / a.kw, kx = tap - ky * a.kw;
            const bool ok = mok && unsigned(oy + ky - a.pt) < unsigned(H) && unsigned(ox + kx - a.pl) < unsigned(W);
            f32x4 av = *reinterpret_cast<const f32x4*>(sWin + (r + ky * W + kx) * P + c0 + gk * 4);
            if (!ok) av = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x4 bv = __builtin_bit_cast(f32x4, B[i][j]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[e], av[e], acc[j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                   // every wave is done with the window: its storage becomes the partial tiles

#pragma unroll
    for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(sPart + (wave * 16 + r) * PP + j * 16 + 4 * gk) = acc[j];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * a.out.sw + Cout) * 4), 0x00020000);
    for (int idx = tid; idx < 16 * (BN / 2); idx += NT) {
        const int p = idx / (BN / 2), c2 = (idx - p * (BN / 2)) * 2;
        f32x2 v = {0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const f32x2 x = *reinterpret_cast<const f32x2*>(sPart + (w * 16 + p) * PP + c2);
            v[0] += x[0];
            v[1] += x[1];
        }
        const int n = n0 + c2;                         // Cout % BN == 0 (eligibility): always in range
        if (a.bias != nullptr) { v[0] += a.bias[n]; v[1] += a.bias[n + 1]; }
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v), rs_out,
                                              m0 + p < M ? unsigned((m0 + p) * int(a.out.sw) + n) * 4u : OOB, 0, 0);
    }
}

// dst = fragment-major copy of one conv's weights [Cout][KK][Cin] (Cout % 16 == 0, Cin % 16 == 0): for 16-channel block nb, 16-channel
// chunk ch = tap * (Cin/16) + c0/16, lane (gk, r): the 4 floats W[nb*16 + r][tap][c0 + 4gk .. +4)
__global__ __launch_bounds__(256) void permute_weights_frag_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int KK, int Cin) {
    const int cpt = Cin >> 4, total = KK * cpt;
    const int64_t n4 = int64_t(Cout) * KK * Cin / 4;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n4; i += int64_t(gridDim.x) * 256) {
        const int lane = int(i & 63);
        const int64_t t = i >> 6;
        const int ch = int(t % total), nb = int(t / total);
        const int tap = ch / cpt, c0 = (ch - tap * cpt) * 16;
        const int r = lane & 15, gk = lane >> 4;
        reinterpret_cast<f32x4*>(dst)[i] = *reinterpret_cast<const f32x4*>(src + (int64_t(nb * 16 + r) * KK + tap) * Cin + c0 + gk * 4);
    }
}

hipError_t LaunchPermuteWeightsFrag(const float* src, float* dst, int Cout, int KK, int Cin, hipStream_t stream) {
    if ((Cout % 16) || (Cin % 16) || Cout <= 0 || KK <= 0) return hipErrorInvalidValue;
    const int64_t n4 = int64_t(Cout) * KK * Cin / 4;
    const unsigned blocks = unsigned(std::min<int64_t>((n4 + 255) / 256, 2048));
    permute_weights_frag_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(src, dst, Cout, KK, Cin);
    return hipGetLastError();
}

struct WinTile { int tn, waves, maxc; };
constexpr int kNumWinTiles = 4;
constexpr WinTile kWinTiles[kNumWinTiles] = {{1, 8, 9}, {2, 8, 9}, {1, 4, 18}, {2, 4, 18}};

static size_t win_lds_bytes(const ConvArgs& a, const WinTile& t) {
    const size_t win = size_t(16 + (a.kh - 1) * a.in.w + (a.kw - 1)) * (a.in.c + 4) * 4;
    const size_t part = size_t(t.waves) * 16 * (16 * t.tn + 4) * 4;
    return win > part ? win : part;
}

static bool win_eligible(const ConvArgs& a, int wt) {
    const WinTile t = kWinTiles[wt];
    if (a.in.f16 || a.out.f16 || a.wfrag == nullptr || a.res.p != nullptr) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || (a.in.c % 16) || (a.out.c % (16 * t.tn))) return false;
    if (a.sh != 1 || a.sw != 1 || a.out.h != a.in.h || a.out.w != a.in.w || a.out.n != a.in.n) return false;
    if (a.pt < 0 || a.pl < 0 || a.pt >= a.kh || a.pl >= a.kw) return false;
    if ((a.in.sw % 4) || a.in.sh != a.in.w * a.in.sw || a.in.sn != a.in.h * a.in.sh || (reinterpret_cast<uintptr_t>(a.in.p) & 15)) return false;
    if ((reinterpret_cast<uintptr_t>(a.wfrag) & 15) || (a.out.sw & 1) || (reinterpret_cast<uintptr_t>(a.out.p) & 7)) return false;
    if (a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh) return false;
    if (a.pre_scale && ((reinterpret_cast<uintptr_t>(a.pre_scale) & 15) || (reinterpret_cast<uintptr_t>(a.pre_shift) & 15))) return false;
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    if (M > 65536 || (M + 64) * a.in.sw * 4 >= (int64_t(1) << 31) || M * a.out.sw * 4 >= (int64_t(1) << 31) ||
        int64_t(a.out.c) * a.kh * a.kw * a.in.c * 4 >= (int64_t(1) << 31))
        return false;
    const int total = a.kh * a.kw * (a.in.c / 16);
    if (total > t.waves * t.maxc || total < t.waves) return false;
    return win_lds_bytes(a, t) <= size_t(160) * 1024;
}

template <int WT>
static hipError_t launch_win_t(const ConvArgs& a, hipStream_t stream) {
    constexpr WinTile t = kWinTiles[WT];
    DirectGeom g;
    g.cpt = a.in.c / 16;
    g.total = a.kh * a.kw * g.cpt;
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    const dim3 grid(unsigned((M + 15) / 16), unsigned(a.out.c / (16 * t.tn)));
    const size_t lds = win_lds_bytes(a, t);
    if (a.pre_scale) conv_win_kernel<t.tn, t.waves, t.maxc, true><<<grid, dim3(64 * t.waves), lds, stream>>>(a, g);
    else conv_win_kernel<t.tn, t.waves, t.maxc, false><<<grid, dim3(64 * t.waves), lds, stream>>>(a, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// "Activations-stationary" 1x1 conv (fp32; direct tiles 10..14).  Written for mid-size pixel counts with long K (DenseNet block 3
// at batch 32: M = 6272, N = 128, K = 256..992); it also beats the weights-stationary kernel on the big block-1 layers
// (M = 100352, K = 64..224: coalesced activation rows through LDS instead of 16-byte row-strided fragment loads).
// The tiled implicit GEMM has one 64 x 64 workgroup per CU on block 3 and pays an LDS round trip
// + barrier per K tile with nothing to overlap it (16-31 us against an MFMA floor of 4-15 us); the weight slice [128][K] does not fit
// in LDS, so the weights-stationary kernel does not apply.  Roles swapped: a workgroup copies ITS 32 (16) pixels' K channels into LDS once
// (coalesced rows, BN+ReLU prologue on the way; one barrier), every wave owns 16*TNW output channels and streams their weights
// from the fragment-major mirror (1 KiB contiguous per load) through a register ring with static slots - no barrier and no LDS
// write in the K loop.  16x16x4 MFMAs, two 16-pixel blocks per wave share each weight fragment; a lane ends up with 4 consecutive
// channels of one pixel and stores them as one 16-byte quad.  Loads past the last chunk use an out-of-range offset (no traffic); the
// last K % (16 * D) channels run from their ring slots behind wave-uniform branches.
// ------------------------------------------------------------------------------------------------------------------------
template <int WAVES, int TNW, int PB, bool PRE>
__global__ __launch_bounds__(64 * WAVES) void conv1x1_as_kernel(const ConvArgs a) {
    constexpr int NT = 64 * WAVES, D = PB == 1 ? 16 : 8, BNW = 16 * TNW, BN = BNW * WAVES;      // 16-pixel tiles: 128 MFMA cycles per chunk need a deeper ring to cover the weight latency
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_direct[];
    float* const sA = reinterpret_cast<float*>(smem_direct);                       // [16 * PB][P]

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.in.c, P = K + (a.debug >> 8), CH = K >> 4, Cout = a.out.c;      // row pad (floats) comes from the launcher
    const int M = a.out.n * a.out.h * a.out.w;
    const int m0 = blockIdx.x * (16 * PB), n0 = blockIdx.y * BN + wave * BNW;
    const int ipitch = int(a.in.sw), opitch = int(a.out.sw);

    // ---- weight ring: chunk c of 16-channel block nb is the KiB at ((nb * CH + c) * 64 + lane) * 16 ----
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, Cout * K * 4, 0x00020000);
    const int nb = n0 >> 4;
    // fp32 MFMAs and VALU instructions do not overlap on this part: the K loop carries no vector address arithmetic.
    // Weight loads: lane * 16 in the VGPR, everything else in the scalar offset; past the last chunk the last chunk is loaded again
    // (in range, never consumed).  Fragment reads: one base per pixel block, bumped once per ring trip, chunk = immediate offset.
    u32x4 ring[D][TNW];
    int c_l = 0;
    const unsigned wlane = unsigned(lane) * 16u;
    auto issue = [&](int slot) {
        const int c = c_l < CH ? c_l : CH - 1;
#pragma unroll
        for (int j = 0; j < TNW; ++j) ring[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, ((nb + j) * CH + c) * 1024, 0);
        ++c_l;
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);

    // ---- this workgroup's 16 * PB pixel rows -> LDS ----
    {
        constexpr int U = 4;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
        const int c4n = K >> 2;
        const int items = 16 * PB * c4n;
        for (int idx0 = tid; idx0 < items; idx0 += U * NT) {
            u32x4 v[U];
            int row[U], c4[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = idx0 + u * NT;
                row[u] = idx / c4n;
                c4[u] = idx - row[u] * c4n;
                const int p = m0 + row[u];
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (idx < items && p < M) ? unsigned(p * ipitch + c4[u] * 4) * 4u : OOB, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (idx0 + u * NT < items) {
                    f32x4 x = __builtin_bit_cast(f32x4, v[u]);
                    if constexpr (PRE) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.pre_scale + c4[u] * 4);
                        const f32x4 sf = *reinterpret_cast<const f32x4*>(a.pre_shift + c4[u] * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float y = x[e] * sc[e] + sf[e];
                            x[e] = a.pre_relu ? fmaxf(y, 0.f) : y;
                        }
                    }
                    *reinterpret_cast<f32x4*>(sA + row[u] * P + c4[u] * 4) = x;
                }
            }
        }
    }
    __syncthreads();

    f32x4 acc[PB][TNW];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[pb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* abase[PB];                            // this lane's fragment address in each pixel block, at the current ring trip
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) abase[pb] = sA + (pb * 16 + r) * P + gk * 4;
    // activation fragments are read one chunk ahead of the MFMAs that consume them (the LDS latency hides behind 4*PB*TNW MFMAs
    // instead of stalling every chunk; the scheduler otherwise sinks each read to its use).  The read ahead of the last chunk runs
    // 16 floats past K: the row's pad and the head of the next row (the launcher allocates 64 bytes behind the last row).
    f32x4 avn[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) avn[pb] = *reinterpret_cast<const f32x4*>(abase[pb]);
    auto compute = [&](int slot) {                     // slot = chunk index within the ring trip: the next chunk sits (slot + 1) * 16 floats on
        f32x4 av[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) av[pb] = avn[pb];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) avn[pb] = *reinterpret_cast<const f32x4*>(abase[pb] + (slot + 1) * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
                    acc[pb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, ring[slot][j])[e], av[pb][e], acc[pb][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, PB, 0);                 // next chunk's fragment reads first ...
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * PB * TNW, 0);       // ... then this chunk's MFMAs
    };
    const int full = CH / D, rem = CH - full * D;
    for (int it = 0; it < full; ++it) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            compute(s);
            issue(s);
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) abase[pb] += D * 16;
    }
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < rem) compute(s);                       // wave-uniform: the last CH % D chunks are already in their slots

    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * opitch + Cout) * 4), 0x00020000);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int m = m0 + pb * 16 + r;
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = n0 + j * 16 + 4 * gk;
            f32x4 v = acc[pb][j];
            if (a.bias != nullptr) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bq[e];
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, m < M ? unsigned(m * opitch + n) * 4u : OOB, 0, 0);
        }
    }
}

struct AsTile { int waves, tnw, pb; };
constexpr int kNumAsTiles = 5;
constexpr AsTile kAsTiles[kNumAsTiles] = {{8, 1, 2}, {4, 1, 2}, {8, 2, 2}, {4, 1, 1}, {2, 2, 1}};      // the 16-pixel tiles: the smallest grids

static bool as_eligible(const ConvArgs& a, int at) {
    const AsTile t = kAsTiles[at];
    if (a.in.f16 || a.out.f16 || a.wfrag == nullptr || a.res.p != nullptr) return false;
    if (a.kh != 1 || a.kw != 1 || a.sh != 1 || a.sw != 1 || a.pt != 0 || a.pl != 0) return false;
    if (a.out.h != a.in.h || a.out.w != a.in.w || a.out.n != a.in.n) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || (a.in.c % 16) || (a.out.c % (16 * t.tnw * t.waves))) return false;
    if ((a.in.sw % 4) || a.in.sh != a.in.w * a.in.sw || a.in.sn != a.in.h * a.in.sh || (reinterpret_cast<uintptr_t>(a.in.p) & 15)) return false;
    if ((a.out.sw % 4) || a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh || (reinterpret_cast<uintptr_t>(a.out.p) & 15)) return false;
    if ((reinterpret_cast<uintptr_t>(a.wfrag) & 15) || (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15))) return false;
    if (a.pre_scale && ((reinterpret_cast<uintptr_t>(a.pre_scale) & 15) || (reinterpret_cast<uintptr_t>(a.pre_shift) & 15))) return false;
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    if (M > (int64_t(1) << 22) || M * a.in.sw * 4 >= (int64_t(1) << 31) || M * a.out.sw * 4 >= (int64_t(1) << 31) || int64_t(a.out.c) * a.in.c * 4 >= (int64_t(1) << 31))
        return false;
    return size_t(16 * t.pb) * (a.in.c + 8) * 4 + 64 <= size_t(160) * 1024;
}

template <int AT>
static hipError_t launch_as_t(const ConvArgs& a_in, hipStream_t stream) {
    constexpr AsTile t = kAsTiles[AT];
    ConvArgs a = a_in;
    // Row pad of the LDS activation tile.  ds_read_b128 is served in four 16-lane groups ({0-3,12-15,20-27}, ...): with a pad of 4 floats
    // (row pitch = 4 mod 64 banks for K % 64 == 0) pixel row 11 of k-group 1 and row 12 of k-group 0 of one group share a 16-byte slot:
    // every fragment read 2-way conflicted (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.45 in profiles/r01 and r02).  A pad of 8 floats
    // (pitch = 8 or 40 mod 64 for K % 32 == 0) is conflict-free: the counter reads 0 and the LDS-active cycles halve -- at unchanged
    // kernel time (0.610 ms -> 0.610 ms per forward): the kernel was never LDS-bound.
    const int pad = Knobs().as_pad;
    a.debug = pad << 8;
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    const dim3 grid(unsigned((M + 16 * t.pb - 1) / (16 * t.pb)), unsigned(a.out.c / (16 * t.tnw * t.waves)));
    const size_t lds = size_t(16 * t.pb) * (a.in.c + pad) * 4 + 64;      // + the read-ahead past the last row
    if (lds > size_t(160) * 1024) return hipErrorInvalidValue;
    if (a.pre_scale) conv1x1_as_kernel<t.waves, t.tnw, t.pb, true><<<grid, dim3(64 * t.waves), lds, stream>>>(a);
    else conv1x1_as_kernel<t.waves, t.tnw, t.pb, false><<<grid, dim3(64 * t.waves), lds, stream>>>(a);
    return hipGetLastError();
}

template <int AT>
static hipError_t init_as_t() {
    constexpr AsTile t = kAsTiles[AT];
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_as_kernel<t.waves, t.tnw, t.pb, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_as_kernel<t.waves, t.tnw, t.pb, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

struct DirectTile { int tn, waves, maxc; };
constexpr DirectTile kDirectTiles[kNumDirectBaseTiles] = {{1, 8, 8}, {1, 16, 4}, {1, 9, 8}, {1, 4, 8}, {1, 12, 6}, {2, 8, 4}};

static size_t direct_lds_bytes(int tile, int Cin, bool half, bool pre) {
    const DirectTile t = kDirectTiles[tile];
    return size_t(t.waves) * 32 * (32 * t.tn + 4) * 4 + (pre ? size_t(2) * Cin * (half ? 2 : 4) : 0);
}

bool ConvDirectEligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvDirectTiles) return false;
    if (tile >= kNumDirectBaseTiles + kNumWinTiles) return as_eligible(a, tile - kNumDirectBaseTiles - kNumWinTiles);
    if (tile >= kNumDirectBaseTiles) return win_eligible(a, tile - kNumDirectBaseTiles);
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
    switch (tile - kNumDirectBaseTiles) {
        case 0: return launch_win_t<0>(a, stream);
        case 1: return launch_win_t<1>(a, stream);
        case 2: return launch_win_t<2>(a, stream);
        case 3: return launch_win_t<3>(a, stream);
        case 4: return launch_as_t<0>(a, stream);
        case 5: return launch_as_t<1>(a, stream);
        case 6: return launch_as_t<2>(a, stream);
        case 7: return launch_as_t<3>(a, stream);
        case 8: return launch_as_t<4>(a, stream);
        default: break;
    }
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

template <int WT>
static hipError_t init_win_t() {
    constexpr WinTile t = kWinTiles[WT];
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_win_kernel<t.tn, t.waves, t.maxc, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_win_kernel<t.tn, t.waves, t.maxc, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t InitKernelsDirect() {
    hipError_t e;
    if ((e = init_win_t<0>()) != hipSuccess) return e;
    if ((e = init_win_t<1>()) != hipSuccess) return e;
    if ((e = init_win_t<2>()) != hipSuccess) return e;
    if ((e = init_win_t<3>()) != hipSuccess) return e;
    if ((e = init_as_t<0>()) != hipSuccess) return e;
    if ((e = init_as_t<1>()) != hipSuccess) return e;
    if ((e = init_as_t<2>()) != hipSuccess) return e;
    if ((e = init_as_t<3>()) != hipSuccess) return e;
    if ((e = init_as_t<4>()) != hipSuccess) return e;
#define IE_DIRI(T)                                                     \
    if ((e = init_direct_t<false, T>()) != hipSuccess) return e;       \
    if ((e = init_direct_t<true, T>()) != hipSuccess) return e;
    IE_DIRI(0) IE_DIRI(1) IE_DIRI(2) IE_DIRI(3) IE_DIRI(4) IE_DIRI(5)
#undef IE_DIRI
    return hipSuccess;
}

}  // namespace ie
