// Stem convolution (DenseNet / ResNet conv1: 7x7, stride 2, pad 3, Cin = 3) straight from the graph's NCHW fp32 input.
//
// The generic implicit GEMM reads such an input with a scalar gather (one 4-byte element per lane per K step, an address
// computation each): on DenseNet-121 the stem alone was 5 % of the fp32 forward and 17 % of the fp16 one.  Here a workgroup owns
// an output tile of TH x 16 pixels, stages the (S*(TH-1)+KH) x (S*15+8) x Cin input window it needs into LDS ONCE (coalesced row
// segments of the NCHW planes, zero-filled outside the image, prefetched into registers while the previous tile is on the
// matrix cores) and keeps ALL weights in LDS for its whole (persistent) life.  K is ordered (c, ky, kx) with kx padded to 8
// (zero weights), so the 8 consecutive k of an MFMA operand are 8 consecutive window columns of one (c, ky) row.
//   fp16 mode: window and weights as halfs, v_mfma_f32_32x32x16_f16, half NHWC output.
//   fp32 mode: float window and weights, v_mfma_f32_32x32x2_f32 (4 MFMAs per 16-byte weight fragment), float NHWC output.
// D = W x A^T: a lane owns one output pixel and quads of channels (16-byte stores).
//
// POOL variant: the 3x3 / stride 2 / pad 1 max pool that follows the stem in DenseNet and ResNet runs in the same launch.  A
// workgroup of 8 waves owns 7 x 7 POOLED pixels = a 15 x 15 patch of conv outputs inside its 16 x 16 tile (origin 2 * 7 * t - 1: one halo row / column
// recomputed per side, 1.31x the conv work); the ReLU'd conv tile goes to LDS as halfs (zeros outside the image:
// after a ReLU they never win a max), a barrier, and 49 x Cout/8 threads each take the max of nine 16-byte reads and store 8 channels.  The
// 112 x 112 x 64 tensor between the two ops -- the largest of the whole network -- is never written or read.
#include <hip/hip_runtime.h>

#include "kernels.h"

#ifndef STEM_ABLATE
#define STEM_ABLATE 0     // scripts/probes/stem_probe.cpp builds variants with parts of the kernel switched off
#endif

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

struct StemGeom {
    int tiles_x, tiles_y, num_tiles;
    int oh, ow;                       // conv output size (== the output tensor's unless the max pool is fused)
};

// elements per conv pixel in the POOL variant's LDS tile: 16-byte aligned rows, consecutive pixels 4 banks apart (halfs: 144 B, floats: 272 B)
template <typename T> constexpr int stem_pool_pitch() { return sizeof(T) == 2 ? 72 : 68; }

// T = _Float16 (fp16 mode) or float (fp32 mode)
template <typename T, int CIN, int KH, int KW, int S, int WAVES, bool POOL, bool VEC>
__global__ __launch_bounds__(64 * WAVES) void conv_stem_kernel(const ConvArgs a, const StemGeom g) {
    static_assert(!POOL || (WAVES == 8 && S == 2), "the fused max pool: 16 x 16 conv tiles");
    constexpr int CP = stem_pool_pitch<T>();
    constexpr bool HALF = sizeof(T) == 2;
    constexpr int NT = 64 * WAVES, TW = 16, TH = 2 * WAVES;
    constexpr int G = CIN * KH;                          // (c, ky) groups of 8 k each
    constexpr int KP = HALF ? ((G + 1) / 2) * 16 : G * 8;   // padded K
    constexpr int WROWS = S * (TH - 1) + KH, WCOLS = S * (TW - 1) + 8;
    // window row pitch.  Lanes 16-31 of a row block sit one output row (S window rows) below lanes 0-15; with halfs the pitch
    // puts them 16 banks away (S*WP*2 == 64 mod 128: conflict-free 4-byte reads).  With floats the lanes are 8 bytes apart and a
    // 2-way conflict is unavoidable (and irrelevant beside 64-cycle MFMAs), so the pitch is just the padded width.
    constexpr int WP = HALF ? (S == 2 ? 48 : 96) : (POOL ? 44 : 40);
    static_assert(WP >= WCOLS, "window pitch too small");
    constexpr int BP = KP + (HALF ? 8 : 4);              // weight row pitch (16-byte units odd)
    constexpr int WIN = CIN * WROWS * WP;
    // VEC (image width % 4 == 0): the window is gathered as 16-byte groups of four columns.  The first column the tile needs, ix0 = S * cx0 - 3, is
    // 1 (3 with POOL) past a multiple of four, so XSH more columns are loaded on the left; in LDS the columns are shifted by PSH so that the 8-column
    // MFMA operands of the half variant still start on a 4-byte boundary.  3 loads per thread and tile instead of 10.
    static_assert(S == 2 && KW == 7, "column alignment of the vector gather");
    constexpr int XSH = POOL ? 3 : 1;
    constexpr int GR = (XSH + WCOLS + 3) / 4;             // 16-byte groups per window row
    constexpr int PSH = VEC ? (HALF ? 4 - XSH : 0) : 0;   // LDS column of the first loaded column
    constexpr int CB = VEC ? XSH + PSH : 0;               // LDS column of the first column the tile needs
    static_assert(!VEC || PSH + 4 * GR <= WP, "window pitch too small for the vector gather");
    constexpr int ELEMS = VEC ? CIN * WROWS * GR : CIN * WROWS * WCOLS, PIT = (ELEMS + NT - 1) / NT;
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_stem[];
    T* const sWt = reinterpret_cast<T*>(smem_stem);      // [64][BP]
    T* const sWin = sWt + 64 * BP;                       // [2][CIN][WROWS][WP]
    float* const sBias = reinterpret_cast<float*>(sWin + 2 * WIN);   // [64]
    T* const sC = reinterpret_cast<T*>(sBias + 64);                  // POOL: [TH * TW][CP] conv tile
    constexpr int PT = (TH - 2) / 2;                                 // POOL: pooled rows / columns per tile

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.in.h, W = a.in.w, OH = g.oh, OW = g.ow, Cout = a.out.c;
    const int opitch = int(a.out.sw);

    // ---- preamble: weights [Cout][KH][KW][CIN] (fp32) -> sWt[n][(c*KH + ky)*8 + kx], zero padded; eight independent loads in flight per
    //      thread (one dependent load per trip cost every workgroup ~40 L2 round trips before its first tile) ----
    {
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, Cout * KH * KW * CIN * 4, 0x00020000);
        constexpr int WTRIPS = ((64 * KP + NT - 1) / NT + 7) / 8 * 8;
        for (int t0 = 0; t0 < WTRIPS; t0 += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = tid + (t0 + u) * NT;
                const int n = q / KP, k = q - n * KP;
                const int gi = k >> 3, kx = k & 7;
                const int c = gi / KH, ky = gi - c * KH;
                const bool ok = q < 64 * KP && n < Cout && gi < G && kx < KW;
                wv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, ok ? unsigned(((n * KH + ky) * KW + kx) * CIN + c) * 4u : 0x80000000u, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = tid + (t0 + u) * NT;
                const int n = q / KP, k = q - n * KP;
                if (q < 64 * KP) sWt[n * BP + k] = T(wv[u]);
            }
        }
    }
    for (int q = tid; q < 64; q += NT) sBias[q] = (a.bias != nullptr && q < Cout) ? a.bias[q] : 0.f;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const int esz = a.out.f8 ? 1 : (a.out.f16 ? 2 : 4);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        a.out.p, 0, int((int64_t(a.out.n) * a.out.h * a.out.w - 1) * opitch * esz + Cout * esz), 0x00020000);

    // The window gather: element e = tid + i * NT of the window is (c, wy, wx) whatever the tile, so its offset relative to the window's
    // corner and its LDS slot are computed ONCE; per tile a load costs two compares and an add.  (Recomputing them per tile let the
    // register allocator place 64-bit address temporaries over registers of loads still in flight: an s_waitcnt vmcnt in the middle of
    // the gather, behind the previous tile's stores -- every tile paid a store round trip plus a load round trip.)
    int rel[PIT];
    unsigned slot[PIT];              // column | wy << 8 | LDS element offset << 16; wy = 255 for the padding elements e >= ELEMS (never in bounds)
    constexpr int EPR = VEC ? GR : WCOLS, EW = VEC ? 4 : 1;          // elements per window row, columns per element
#pragma unroll
    for (int i = 0; i < PIT; ++i) {
        const int e = tid + i * NT;
        const int c = e / (WROWS * EPR), rem2 = e - c * (WROWS * EPR);
        const int wy = rem2 / EPR, wx = (rem2 - wy * EPR) * EW;
        rel[i] = (c * H + wy) * W + wx;
        slot[i] = e < ELEMS ? unsigned(wx | (wy << 8) | (((c * WROWS + wy) * WP + wx + PSH) << 16)) : 0xFF00u;
    }
    static_assert(CIN * WROWS * WP < 65536 && WROWS < 255 && WCOLS + 8 < 256, "slot packing");
    float pv[VEC ? 1 : PIT];
    f32x4 pv4[VEC ? PIT : 1];
    auto issue = [&](int tile) {
        const int b = tile / (g.tiles_x * g.tiles_y);
        const int rem = tile - b * (g.tiles_x * g.tiles_y);
        const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
        const int cy0 = POOL ? ty * 2 * PT - 1 : ty * TH, cx0 = POOL ? tx * 2 * PT - 1 : tx * TW;     // first conv output of the tile
        const int iy0 = cy0 * S - a.pt, ix0 = cx0 * S - a.pl - (VEC ? XSH : 0);                      // VEC: ix0 % 4 == 0
        const int base = (b * CIN * H + iy0) * W + ix0;
        unsigned off[PIT];
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int iy = iy0 + int((slot[i] >> 8) & 0xFFu), ix = ix0 + int(slot[i] & 0xFFu);
            const bool ok = unsigned(iy) < unsigned(H) && unsigned(ix) < unsigned(W);          // VEC: W % 4 == 0, a group is inside or outside as a whole
            off[i] = (ok && STEM_ABLATE != 2) ? unsigned(base + rel[i]) * 4u : OOB;
        }
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            if constexpr (VEC) pv4[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off[i], 0, 0));
            else pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, off[i], 0, 0));
        }
    };
    auto commit = [&](int buf) {
        T* const win = sWin + buf * WIN;
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            if ((slot[i] & 0xFF00u) == 0xFF00u) continue;
            T* const d = win + (slot[i] >> 16);
            if constexpr (!VEC) d[0] = T(pv[i]);
            else if constexpr (HALF) {         // PSH is odd: columns 1-2 of the group share an aligned dword
                d[0] = T(pv4[i][0]);
                *reinterpret_cast<h2*>(d + 1) = h2{_Float16(pv4[i][1]), _Float16(pv4[i][2])};
                d[3] = T(pv4[i][3]);
            } else *reinterpret_cast<f32x4*>(d) = pv4[i];
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    const int oyl = 2 * wave + (r >> 4), oxl = r & 15;     // this lane's output pixel inside the tile
    const int lane_base = S * oyl * WP + S * oxl + CB;
    auto compute = [&](int buf) {
        const T* const win = sWin + buf * WIN + lane_base;
        if constexpr (HALF) {
            constexpr int KSTEPS = KP / 16;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                // half-wave hh takes group gi = 2*kk + hh; the pad group (gi == G) re-reads the last real one against zero weights
                const int g0 = 2 * kk, g1 = (2 * kk + 1 < G) ? 2 * kk + 1 : G - 1;
                const int off0 = ((g0 / KH) * WROWS + (g0 % KH)) * WP, off1 = ((g1 / KH) * WROWS + (g1 % KH)) * WP;
                const T* const ap = win + (hh ? off1 : off0);
                u32x4 araw;
#pragma unroll
                for (int q = 0; q < 4; ++q) araw[q] = *reinterpret_cast<const unsigned*>(ap + 2 * q);
                const h8 av = __builtin_bit_cast(h8, araw);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const h8 b = *reinterpret_cast<const h8*>(sWt + (j * 32 + r) * BP + kk * 16 + hh * 8);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, av, acc[j], 0, 0, 0);
                }
            }
        } else {
            // fp32: per (c, ky) group the half-waves take kx = 4*hh + e, e = 0..3 (one 16-byte weight fragment feeds four MFMAs)
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                const int off = ((gi / KH) * WROWS + (gi % KH)) * WP;
                const T* const ap = win + off + 4 * hh;
                float av[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = float(ap[e]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(sWt + (j * 32 + r) * BP + gi * 8 + hh * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[e], av[e], acc[j], 0, 0, 0);
                }
            }
        }
    };
    auto epilogue = [&](int tile) {
        const int b = tile / (g.tiles_x * g.tiles_y);
        const int rem = tile - b * (g.tiles_x * g.tiles_y);
        const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
        const int oy = ty * TH + oyl, ox = tx * TW + oxl;
        const unsigned rowoff = (oy < OH && ox < OW && STEM_ABLATE != 1) ? unsigned(((b * OH + oy) * OW + ox) * opitch * esz) : OOB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + j * 32 + 8 * gq + 4 * hh);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = acc[j][4 * gq + q] + bq[q];
                    v[4 * gq + q] = a.relu ? fmaxf(x, 0.f) : x;
                    acc[j][4 * gq + q] = 0.f;
                }
            }
            if (a.out.f8) {
                // fp8 mode: re-quantise with 1 / (output scale), one dword of four e4m3 per quad, then the half-waves trade quads so
                // each lane stores 16 consecutive channels (same exchange as conv_igemm_f8_kernel)
                unsigned d[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float q4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) q4[q] = __builtin_fminf(__builtin_fmaxf(v[4 * gq + q] * a.out_qscale, -448.f), 448.f);
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q4[0], q4[1], 0, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(q4[2], q4[3], pk, true);
                    d[gq] = unsigned(pk);
                }
                const auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                const int n = j * 32 + 16 * hh;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s0[1], s1[0], s1[1]}, rs_out, n < Cout ? rowoff + unsigned(n) : OOB, 0, 0);
            } else if (a.out.f16) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const h4 qa = {_Float16(v[8 * gp + 0]), _Float16(v[8 * gp + 1]), _Float16(v[8 * gp + 2]), _Float16(v[8 * gp + 3])};
                    const h4 qb = {_Float16(v[8 * gp + 4]), _Float16(v[8 * gp + 5]), _Float16(v[8 * gp + 6]), _Float16(v[8 * gp + 7])};
                    const u32x2 xa = __builtin_bit_cast(u32x2, qa), xb = __builtin_bit_cast(u32x2, qb);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(xa[0], xb[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(xa[1], xb[1], false, false);
                    const int n = j * 32 + 8 * (2 * gp + hh);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rs_out, n < Cout ? rowoff + unsigned(n * 2) : OOB, 0, 0);
                }
            } else {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int n = j * 32 + 8 * gq + 4 * hh;
                    const f32x4 q4 = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q4), rs_out, n < Cout ? rowoff + unsigned(n * 4) : OOB, 0, 0);
                }
            }
        }
    };

    // POOL: ReLU'd conv tile -> LDS (halfs, zeros outside the image) ...
    auto epilogue_to_lds = [&](int tile) {
        const int rem = tile % (g.tiles_x * g.tiles_y);
        const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
        const int cy = ty * 2 * PT - 1 + oyl, cx = tx * 2 * PT - 1 + oxl;
        const bool valid = unsigned(cy) < unsigned(OH) && unsigned(cx) < unsigned(OW);
        T* const dst = sC + (oyl * TW + oxl) * CP;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + j * 32 + 8 * gq + 4 * hh);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[4 * gq + q] = valid ? fmaxf(acc[j][4 * gq + q] + bq[q], 0.f) : 0.f;
                    acc[j][4 * gq + q] = 0.f;
                }
            }
            if constexpr (HALF) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const h4 qa = {_Float16(v[8 * gp + 0]), _Float16(v[8 * gp + 1]), _Float16(v[8 * gp + 2]), _Float16(v[8 * gp + 3])};
                    const h4 qb = {_Float16(v[8 * gp + 4]), _Float16(v[8 * gp + 5]), _Float16(v[8 * gp + 6]), _Float16(v[8 * gp + 7])};
                    const u32x2 xa = __builtin_bit_cast(u32x2, qa), xb = __builtin_bit_cast(u32x2, qb);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(xa[0], xb[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(xa[1], xb[1], false, false);
                    *reinterpret_cast<u32x4*>(dst + j * 32 + 8 * (2 * gp + hh)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            } else {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<f32x4*>(dst + j * 32 + 8 * gq + 4 * hh) = f32x4{v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
            }
        }
    };
    // ... and, after a barrier, one thread per (pooled pixel, 8 channels): max of the 3 x 3 window, one store
    auto pool_store = [&](int tile) {
        const int cgs = Cout >> 3;
        if (tid >= PT * PT * cgs) return;
        const int b = tile / (g.tiles_x * g.tiles_y);
        const int rem = tile - b * (g.tiles_x * g.tiles_y);
        const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
        const int pp = tid / cgs, cg = tid - pp * cgs;
        const int ply = pp / PT, plx = pp - ply * PT;
        const int py = ty * PT + ply, px = tx * PT + plx;
        const T* const src = sC + ((2 * ply) * TW + 2 * plx) * CP + cg * 8;
        const bool ok = py < a.out.h && px < a.out.w;
        const unsigned off = ok ? unsigned((((b * a.out.h + py) * a.out.w + px) * opitch + cg * 8) * esz) : OOB;
        if constexpr (HALF) {
            h8 m = *reinterpret_cast<const h8*>(src);
#pragma unroll
            for (int d = 1; d < 9; ++d) m = __builtin_elementwise_max(m, *reinterpret_cast<const h8*>(src + ((d / 3) * TW + (d % 3)) * CP));
            if (a.out.f8) {
                unsigned d[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float q4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) q4[e] = __builtin_fminf(float(m[4 * q + e]) * a.out_qscale, 448.f);
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q4[0], q4[1], 0, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(q4[2], q4[3], pk, true);
                    d[q] = unsigned(pk);
                }
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{d[0], d[1]}, rs_out, off, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), rs_out, off, 0, 0);
            }
        } else {
            f32x4 m0 = *reinterpret_cast<const f32x4*>(src), m1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int d = 1; d < 9; ++d) {
                const T* const q = src + ((d / 3) * TW + (d % 3)) * CP;
                m0 = __builtin_elementwise_max(m0, *reinterpret_cast<const f32x4*>(q));
                m1 = __builtin_elementwise_max(m1, *reinterpret_cast<const f32x4*>(q + 4));
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m0), rs_out, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m1), rs_out, ok ? off + 16u : OOB, 0, 0);
        }
    };

    int tile = blockIdx.x, buf = 0;
    if (tile < g.num_tiles) {
        issue(tile);
        commit(0);
    }
    __syncthreads();                         // weights + first window visible
    while (tile < g.num_tiles) {
        const int ntile = tile + gridDim.x;
        if (ntile < g.num_tiles) issue(ntile);
        __builtin_amdgcn_sched_barrier(0);
        if (STEM_ABLATE != 3) compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (POOL) {
            epilogue_to_lds(tile);
            __syncthreads();
            pool_store(tile);
        } else {
            if (STEM_ABLATE != 4) epilogue(tile);
        }
        if (ntile < g.num_tiles) commit(buf ^ 1);      // the other buffer: its last readers passed the previous barrier
        __syncthreads();
        tile = ntile;
        buf ^= 1;
    }
}

template <typename T, int CIN, int KH, int KW, int S, int WAVES, bool POOL = false>
static size_t stem_lds_bytes() {
    constexpr bool HALF = sizeof(T) == 2;
    constexpr int TH = 2 * WAVES, G = CIN * KH;
    constexpr int KP = HALF ? ((G + 1) / 2) * 16 : G * 8;
    constexpr int WROWS = S * (TH - 1) + KH;
    constexpr int WP = HALF ? (S == 2 ? 48 : 96) : (POOL ? 44 : 40);
    constexpr int BP = KP + (HALF ? 8 : 4);
    return size_t(64 * BP + 2 * CIN * WROWS * WP) * sizeof(T) + 64 * sizeof(float) + (POOL ? size_t(TH * 16 * stem_pool_pitch<T>()) * sizeof(T) : 0);
}

bool ConvStemEligible(const ConvArgs& a) {
    // the one shape this kernel is instantiated for: 7x7 / stride 2 / pad 3 over a 3-channel dense NCHW fp32 input
    if (a.in.f16 || a.in.c != 3 || a.kh != 7 || a.kw != 7 || a.sh != 2 || a.sw != 2 || a.pt != 3 || a.pl != 3) return false;
    if (a.pre_scale != nullptr || a.w == nullptr) return false;
    if (a.in.sw != 1 || a.in.sh != a.in.w || a.in.sc != int64_t(a.in.h) * a.in.w || a.in.sn != a.in.sc * a.in.c) return false;   // dense NCHW
    if (a.out.sc != 1 || a.out.c > 64 || (a.out.c & 7) || (a.out.sw & 7) || (reinterpret_cast<uintptr_t>(a.out.p) & 15)) return false;
    if (a.out.f8 && ((a.out.c & 15) || (a.out.sw & 15))) return false;
    if (a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh) return false;
    if (a.out.h != (a.in.h + 6 - 7) / 2 + 1 || a.out.w != (a.in.w + 6 - 7) / 2 + 1) return false;
    const int64_t in_elems = int64_t(a.in.n) * 3 * a.in.h * a.in.w, out_elems = int64_t(a.out.n) * a.out.h * a.out.w * a.out.sw;
    return in_elems * 4 < (int64_t(1) << 31) && out_elems * 4 < (int64_t(1) << 31);
}

// 16-byte gathers: every image row starts on a 16-byte boundary
static bool stem_vec_ok(const ConvArgs& a) { return (a.in.w & 3) == 0 && (reinterpret_cast<uintptr_t>(a.in.p) & 15) == 0; }

static int stem_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    return cus;
}

template <typename T>
static hipError_t launch_stem_t(const ConvArgs& a, hipStream_t stream) {
    constexpr int WAVES = 4;
    StemGeom g;
    g.tiles_x = (a.out.w + 15) / 16;
    g.tiles_y = (a.out.h + 2 * WAVES - 1) / (2 * WAVES);
    g.num_tiles = a.out.n * g.tiles_x * g.tiles_y;
    g.oh = a.out.h; g.ow = a.out.w;
    const size_t lds = stem_lds_bytes<T, 3, 7, 7, 2, WAVES>();
    const int cus = stem_cus();
    if (cus == 0) return hipErrorUnknown;
    int per_cu = int((size_t(160) * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const int slots = cus * per_cu;
    const int iters = (g.num_tiles + slots - 1) / slots;
    const int gx = (g.num_tiles + iters - 1) / iters;
    if (stem_vec_ok(a)) conv_stem_kernel<T, 3, 7, 7, 2, WAVES, false, true><<<dim3(gx), dim3(64 * WAVES), lds, stream>>>(a, g);
    else conv_stem_kernel<T, 3, 7, 7, 2, WAVES, false, false><<<dim3(gx), dim3(64 * WAVES), lds, stream>>>(a, g);
    return hipGetLastError();
}

hipError_t LaunchConvStem(const ConvArgs& a_in, hipStream_t stream) {
    if (!ConvStemEligible(a_in)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = int64_t(a.in.n) * a.in.c * a.in.h * a.in.w * 4;
    // half arithmetic only when the result is stored as half anyway (fp16 precision mode)
    return (a.out.f16 || a.out.f8) ? launch_stem_t<_Float16>(a, stream) : launch_stem_t<float>(a, stream);
}

// Stem + max pool in one launch: `a` is the stem conv's argument set with `out` = the POOLED tensor (float, half or e4m3, NHWC).
bool ConvStemPoolEligible(const ConvArgs& a) {
    if (a.in.f16 || a.in.f8 || a.in.c != 3 || a.kh != 7 || a.kw != 7 || a.sh != 2 || a.sw != 2 || a.pt != 3 || a.pl != 3) return false;
    if (a.pre_scale != nullptr || a.w == nullptr || !a.relu || a.res.p != nullptr) return false;      // the ReLU makes 0 the neutral element of the max
    if (a.in.sw != 1 || a.in.sh != a.in.w || a.in.sc != int64_t(a.in.h) * a.in.w || a.in.sn != a.in.sc * a.in.c) return false;   // dense NCHW
    if (a.out.sc != 1 || a.out.c > 64 || (a.out.c & 7) || (a.out.sw & 7) || (reinterpret_cast<uintptr_t>(a.out.p) & 15)) return false;
    if (a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh) return false;
    const int oh = (a.in.h + 6 - 7) / 2 + 1, ow = (a.in.w + 6 - 7) / 2 + 1;          // the conv's output
    if (oh < 1 || ow < 1 || a.out.h != (oh + 2 - 3) / 2 + 1 || a.out.w != (ow + 2 - 3) / 2 + 1) return false;   // 3x3 / s2 / p1 windows over it
    const int64_t in_elems = int64_t(a.in.n) * 3 * a.in.h * a.in.w, out_elems = int64_t(a.out.n) * a.out.h * a.out.w * a.out.sw;
    return in_elems * 4 < (int64_t(1) << 31) && out_elems * 4 < (int64_t(1) << 31);
}

template <typename T>
static hipError_t launch_stem_pool_t(const ConvArgs& a, hipStream_t stream) {
    constexpr int WAVES = 8, PT = 7;
    StemGeom g;
    g.tiles_x = (a.out.w + PT - 1) / PT;
    g.tiles_y = (a.out.h + PT - 1) / PT;
    g.num_tiles = a.out.n * g.tiles_x * g.tiles_y;
    g.oh = (a.in.h + 6 - 7) / 2 + 1; g.ow = (a.in.w + 6 - 7) / 2 + 1;
    const size_t lds = stem_lds_bytes<T, 3, 7, 7, 2, WAVES, true>();
    const int cus = stem_cus();
    if (cus == 0) return hipErrorUnknown;
    const int per_cu = int((size_t(160) * 1024) / lds) >= 2 ? 2 : 1;
    const int slots = cus * per_cu;
    const int iters = (g.num_tiles + slots - 1) / slots;
    const int gx = (g.num_tiles + iters - 1) / iters;
    if (stem_vec_ok(a)) conv_stem_kernel<T, 3, 7, 7, 2, WAVES, true, true><<<dim3(gx), dim3(64 * WAVES), lds, stream>>>(a, g);
    else conv_stem_kernel<T, 3, 7, 7, 2, WAVES, true, false><<<dim3(gx), dim3(64 * WAVES), lds, stream>>>(a, g);
    return hipGetLastError();
}

hipError_t LaunchConvStemPool(const ConvArgs& a_in, hipStream_t stream) {
    if (!ConvStemPoolEligible(a_in)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = int64_t(a.in.n) * a.in.c * a.in.h * a.in.w * 4;
    return (a.out.f16 || a.out.f8) ? launch_stem_pool_t<_Float16>(a, stream) : launch_stem_pool_t<float>(a, stream);
}

hipError_t InitKernelsStem() {
    const void* const kernels[] = {
        reinterpret_cast<const void*>(&conv_stem_kernel<_Float16, 3, 7, 7, 2, 4, false, false>), reinterpret_cast<const void*>(&conv_stem_kernel<_Float16, 3, 7, 7, 2, 4, false, true>),
        reinterpret_cast<const void*>(&conv_stem_kernel<_Float16, 3, 7, 7, 2, 8, true, false>),  reinterpret_cast<const void*>(&conv_stem_kernel<_Float16, 3, 7, 7, 2, 8, true, true>),
        reinterpret_cast<const void*>(&conv_stem_kernel<float, 3, 7, 7, 2, 4, false, false>),    reinterpret_cast<const void*>(&conv_stem_kernel<float, 3, 7, 7, 2, 4, false, true>),
        reinterpret_cast<const void*>(&conv_stem_kernel<float, 3, 7, 7, 2, 8, true, false>),     reinterpret_cast<const void*>(&conv_stem_kernel<float, 3, 7, 7, 2, 8, true, true>)};
    for (const void* k : kernels) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace ie
