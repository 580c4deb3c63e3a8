// Winograd F(2x2, 3x3) convolution for the DenseNet growth convs (fp32, 3x3 / stride 1 / pad 1, 32 output channels, even H and W).
//
// The 3x3 convs are 48 % of DenseNet-121's FLOPs and the fp32 matrix pipe (157 TFLOP/s) is what bounds them.  F(2x2, 3x3) computes
// a 2x2 output tile from a 4x4 input tile with 16 multiplies per (input channel, output channel) instead of 36: 2.25x fewer MACs, at
// the price of an input transform V = B^T d B, an output transform Y = A^T M A and transformed weights U = G g G^T (built once at
// load, LaunchWinogradWeights).  Exact real arithmetic, fp32 rounding of the same order as a direct fp32 conv (the transform
// matrices hold only 0, +-1, +-1/2); the parity tests hold it to the same 2e-4 bound as every other fp32 kernel.
//
// A workgroup (4 waves) owns TR x TC output tiles (<= 32, the M dimension of the MFMA) of one image and walks Cin in slices of 16
// channels; M_xi[tile][cout] += V_xi[tile][:] . U_xi[:][cout] on v_mfma_f32_32x32x2_f32, 16 positions xi split over the 4 waves; at the
// end the 16 accumulator tiles go through LDS in two halves, every (tile, cout) gets its 2x2 outputs (Y = A^T M A), bias / ReLU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// U[xi = 4i + j][cout][cin] = (G g G^T)[i][j], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], stored fragment-major for the MFMA B operand:
// element ((xi * (Cin/8) + ch8) * 64 + lane) * 4 + e  =  U[xi][cout = lane & 31][cin = ch8*8 + (lane >> 5)*4 + e]
__global__ void winograd_weights_kernel(const float* __restrict__ w, float* __restrict__ u, const int Cout, const int Cin) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin) return;
    const int co = idx / Cin, ci = idx - co * Cin;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = w[((co * 3 + a) * 3 + b) * Cin + ci];
    float t[4][3];      // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int ch8 = ci >> 3, hh = (ci >> 2) & 1, e = ci & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float uu[4] = {t[i][0], 0.5f * (t[i][0] + t[i][1] + t[i][2]), 0.5f * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xi = 4 * i + j;
            u[((int64_t(xi) * (Cin >> 3) + ch8) * 64 + (hh * 32 + co)) * 4 + e] = uu[j];
        }
    }
}

// The same U as three bf16 planes (U = U0 + U1 + U2 exactly; kernels_x6.hip has the arithmetic) in the B-operand fragment order of
// v_mfma_f32_32x32x16_bf16:  element ((((p * 16 + xi) * (Cin/16) + kb) * 64 + lane) * 8 + i  =  piece p of U[xi][lane & 31][kb*16 + (lane>>5)*8 + i]
__global__ void winograd_weights_x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, const int Cout, const int Cin) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin) return;
    const int co = idx / Cin, ci = idx - co * Cin;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = w[((co * 3 + a) * 3 + b) * Cin + ci];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int KB = Cin >> 4, kb = ci >> 4, lane = co + 32 * ((ci >> 3) & 1), e = ci & 7;
    const int64_t plane = int64_t(16) * 32 * Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float uu[4] = {t[i][0], 0.5f * (t[i][0] + t[i][1] + t[i][2]), 0.5f * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = uu[j];
            const unsigned h0 = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
            const float r1 = x - __builtin_bit_cast(float, h0);
            const unsigned h1 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
            const unsigned h2 = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, h1));
            const int64_t off = ((int64_t(4 * i + j) * KB + kb) * 64 + lane) * 8 + e;
            dst[off] = static_cast<unsigned short>(h0 >> 16);
            dst[plane + off] = static_cast<unsigned short>(h1 >> 16);
            dst[2 * plane + off] = static_cast<unsigned short>(h2 >> 16);
        }
    }
}

hipError_t LaunchWinogradWeightsX6(const float* w, void* dst, int Cout, int Cin, hipStream_t stream) {
    if (Cout != 32 || (Cin % 16) || Cin <= 0) return hipErrorInvalidValue;
    const int total = Cout * Cin;
    winograd_weights_x6_kernel<<<dim3((total + 255) / 256), dim3(256), 0, stream>>>(w, static_cast<unsigned short*>(dst), Cout, Cin);
    return hipGetLastError();
}

hipError_t LaunchWinogradWeights(const float* w, float* u, int Cout, int Cin, hipStream_t stream) {
    if (Cout != 32 || (Cin % 16) || Cin <= 0) return hipErrorInvalidValue;
    const int total = Cout * Cin;
    winograd_weights_kernel<<<dim3((total + 255) / 256), dim3(256), 0, stream>>>(w, u, Cout, Cin);
    return hipGetLastError();
}

struct WinoGeom {
    int TR, TC;           // output tiles per workgroup (rows x columns), TR * TC <= 32
    int TH, TW;           // tiles per image
    int bry, brx;         // workgroup blocks per image
};

// fp32 MFMAs and VALU work do not overlap on gfx950 (scripts/probes/mfma_rate_probe.cpp: v_mfma_f32_32x32x2_f32 issues every 65.5 cycles
// from one wave; each v_pk_fma_f32 placed between two of them adds 8 cycles, plus 12 per MFMA<->VALU alternation, from the same wave or
// from the other wave of the SIMD alike -- the fp32 matrix rate IS the packed-fp32 VALU rate, 256 FLOP/clk/CU).  Two earlier builds of
// this kernel (a shared sV[position][tile][channel] filled by per-thread transform items, then MFMAs; and the same software-pipelined
// with the transform interleaved into the MFMA issue shadow) measured exactly that: MFMAs alone 1365 cycles per 8-channel stage,
// transform alone 810, together 2000.  So the kernel minimises the COUNT of non-MFMA instructions instead of hiding them:
//   * wave w owns row i = w of the transformed 4x4 tile (positions 4w .. 4w+3) and builds its own A fragments in registers straight
//     from the window: lane (tile r, k-half hh) reads 2 window rows x 4 columns x its 4 channels (8 ds_read_b128), forms
//     m_c = d[ra][c] +- d[rb][c] and V[i][0..3] = m0-m2, m1+m2, m2-m1, m1-m3 as packed-fp32 ops: 16 v_pk per 16 MFMAs, in one block;
//   * no transformed-tile buffer in LDS, ONE barrier per 16-channel slice (the window is double-buffered and register-prefetched two
//     slices ahead), the LDS reads of the next stage are issued before the MFMAs of the current one;
//   * U fragments come straight from the fragment-major mirror in L2, two stages ahead.
// WAVES = 8 (tiles 4..7): two positions per wave instead of four -- half the MFMA chain per wave, four waves per SIMD instead of two
// (124 VGPRs), at 10 v_pk per 8 MFMAs instead of 16 per 16.  Measured (wino_probe, 56x56x128 -> 32): one workgroup per CU 15.7 us
// against 19.3, 3.5 per CU (batch 32) 50.0 us against 56.2; sixteen waves (one position each) gave nothing more (15.6 / 57.5).
// X6 (tiles 8..11, eight waves, opt-in IE_FP32_SPLIT=1): the Winograd-domain products M_xi += V_xi . U_xi on the bf16 matrix pipe with both
// operands split exactly into three bf16 pieces (kernels_x6.hip: six MFMAs per 32x32x16 block, fp32 accumulation, nothing above 2^-24 of
// a product dropped).  V is transformed in fp32 as before and split in registers (VALU work that co-issues with the bf16 MFMAs); U comes
// pre-split from LaunchWinogradWeightsX6 through `w16`.  One stage per 16-channel slice; a lane holds 8 channels.
template <int WAVES, bool X6 = false>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, (WAVES == 8 && !X6) ? 4 : 2))) void conv3x3_wino_kernel(const ConvArgs a, const WinoGeom g) {
    constexpr int NT = 64 * WAVES, CS = 16, LP = CS + 4, MP = 32 + 4, PITW = 1024 / NT;
    constexpr int NJ = 16 / WAVES;                      // positions (columns j of row i) per wave
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem_wino[];
    const int WR = 2 * g.TR + 2, WC = 2 * g.TC + 2, npx = WR * WC;
    const int win_floats = (npx * LP + 3) & ~3;
    float* const sWin = smem_wino;                      // [2][npx][LP]
    float* const sM = smem_wino;                        // [8][32][MP] in the epilogue (aliases)

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.in.c, H = a.in.h, W = a.in.w;
    const int ipitch = int(a.in.sw), opitch = int(a.out.sw);
    int bid = blockIdx.x;
    const int bx = bid % g.brx; bid /= g.brx;
    const int by = bid % g.bry;
    const int b = bid / g.bry;
    const int ty0 = by * g.TR, tx0 = bx * g.TC;
    const int ntr = min(g.TR, g.TH - ty0), ntc = min(g.TC, g.TW - tx0);
    const int y0 = 2 * ty0 - 1, x0 = 2 * tx0 - 1;
    const int ntiles = g.TR * g.TC;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, 16 * 32 * Cin * 4, 0x00020000);

    const int items = npx * 4;
    int woff[PITW];
#pragma unroll
    for (int i = 0; i < PITW; ++i) {
        const int it = tid + i * NT;
        int off = -1;
        if (it < items) {
            const int px = it >> 2, q = it & 3;
            const int wy = px / WC, wx = px - wy * WC;
            const int y = y0 + wy, x = x0 + wx;
            if (unsigned(y) < unsigned(H) && unsigned(x) < unsigned(W)) off = ((b * H + y) * W + x) * ipitch + q * 4;
        }
        woff[i] = off;
    }
    // fp32 MFMAs and VALU instructions do not overlap: the K loop carries no vector address arithmetic.  Loads: the lane's
    // part of the address in a VGPR fixed before the loop, slice / stage in the scalar offset (past the end: the last one again, in
    // range and never consumed; a lane outside the image keeps its out-of-range marker whatever the scalar offset says).
    unsigned wvo[PITW];
    float* wlds[2][PITW];
#pragma unroll
    for (int i = 0; i < PITW; ++i) {
        const int it = tid + i * NT;
        wvo[i] = woff[i] >= 0 ? unsigned(woff[i]) * 4u : OOB;
#pragma unroll
        for (int bf = 0; bf < 2; ++bf) wlds[bf][i] = sWin + bf * win_floats + (it >> 2) * LP + (it & 3) * 4;
    }
    f32x4 pvA[PITW], pvB[PITW];                         // windows in flight: even / odd slices
    const int nslices = Cin / CS, nstages = 2 * nslices;
    auto issue_window = [&](f32x4 (&pv)[PITW], int s) {
#pragma unroll
        for (int i = 0; i < PITW; ++i)
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, wvo[i], (s < nslices ? s : nslices - 1) * CS * 4, 0));
    };
    auto commit_window = [&](const f32x4 (&pv)[PITW], int buf) {
#pragma unroll
        for (int i = 0; i < PITW; ++i)
            if (tid + i * NT < items) *reinterpret_cast<f32x4*>(wlds[buf][i]) = pv[i];
    };
    // Row i of the transformed tile: m_c = d[ra][c] + sg * d[rb][c] with (ra, rb, sg) = (0,2,-), (1,2,+), (2,1,-), (1,3,-).
    const int wi = wave / (WAVES / 4), jh = wave % (WAVES / 4);       // this wave's row and (8 waves) its pair of columns
    const int ra = wi == 0 ? 0 : wi == 2 ? 2 : 1, rb = wi == 2 ? 1 : wi == 3 ? 3 : 2;
    const float sg = wi == 1 ? 1.f : -1.f;
    constexpr int NC = NJ == 4 ? 4 : 3;                 // window columns a wave needs: all four, or jh .. jh+2
    // Lanes beyond the block's tiles read in-allocation garbage; their accumulator rows are never read by the epilogue.
    const int ltr = r / g.TC, ltc = r - ltr * g.TC;
    const int col0 = NJ == 4 ? 0 : jh;
    const int dA = ((2 * ltr + ra) * WC + 2 * ltc + col0) * LP + hh * 4, dB = ((2 * ltr + rb) * WC + 2 * ltc + col0) * LP + hh * 4;
    f32x16 acc[NJ];
#pragma unroll
    for (int xl = 0; xl < NJ; ++xl)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xl][e] = 0.f;
    if constexpr (X6) {
        static_assert(NJ == 2, "the split variant is written for eight waves");
        const __amdgpu_buffer_rsrc_t rs_ux = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w16), 0, 3 * 16 * 32 * Cin * 2, 0x00020000);
        const int KB = Cin >> 4;
        const int plane_b = 16 * 32 * Cin * 2;          // bytes per plane
        u32x4 ux0[NJ][3], ux1[NJ][3];                   // U fragments: even / odd slices
        auto issue_ux = [&](u32x4 (&ux)[NJ][3], int sl) {
            const int kb = sl < nslices ? sl : nslices - 1;
#pragma unroll
            for (int xl = 0; xl < NJ; ++xl)
#pragma unroll
                for (int p = 0; p < 3; ++p) ux[xl][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_ux, unsigned(lane) * 16u, p * plane_b + ((NJ * wave + xl) * KB + kb) * 1024, 0);
        };
        // window rows ra / rb, columns jh .. jh+2, this lane's EIGHT channels (two quads) of the slice
        const float* const qdA[2] = {sWin + dA + hh * 4, sWin + win_floats + dA + hh * 4};      // dA carries hh * 4: + hh * 4 more = hh * 8
        const float* const qdB[2] = {sWin + dB + hh * 4, sWin + win_floats + dB + hh * 4};
        f32x4 xra[3][2], xrb[3][2];
        auto read_dx = [&](int buf) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    xra[c][q] = *reinterpret_cast<const f32x4*>(qdA[buf] + c * LP + q * 4);
                    xrb[c][q] = *reinterpret_cast<const f32x4*>(qdB[buf] + c * LP + q * 4);
                }
        };
        const float ka0 = jh == 0 ? 1.f : -1.f, ka1 = jh == 0 ? 0.f : 1.f, ka2 = jh == 0 ? -1.f : 0.f;
        const float kb0 = jh == 0 ? 0.f : 1.f, kb1 = jh == 0 ? 1.f : 0.f, kb2 = jh == 0 ? 1.f : -1.f;
        u32x4 axA[NJ][3], axB[NJ][3];                   // split A fragments: even / odd slices
        auto make_ax = [&](u32x4 (&ax)[NJ][3]) {
            f32x4 v[NJ][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 m[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) m[c] = xra[c][q] + sg * xrb[c][q];
                // branch-free (one scheduling region with the MFMAs): columns jh .. jh+2 with wave-uniform coefficients
                //   jh = 0: V0 = m0 - m2, V1 = m1 + m2;   jh = 1 (m = columns 1..3): V2 = m1 - m0, V3 = m0 - m2
                v[0][q] = ka0 * m[0] + ka1 * m[1] + ka2 * m[2];
                v[1][q] = kb0 * m[0] + kb1 * m[1] + kb2 * m[2];
            }
#pragma unroll
            for (int xl = 0; xl < NJ; ++xl) {
                unsigned h0[8], h1[8], h2[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = v[xl][e >> 2][e & 3];
                    h0[e] = __builtin_bit_cast(unsigned, x);
                    const float r1 = x - __builtin_bit_cast(float, h0[e] & 0xffff0000u);
                    h1[e] = __builtin_bit_cast(unsigned, r1);
                    h2[e] = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, h1[e] & 0xffff0000u));
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    ax[xl][0][d] = __builtin_amdgcn_perm(h0[2 * d + 1], h0[2 * d], 0x07060302u);
                    ax[xl][1][d] = __builtin_amdgcn_perm(h1[2 * d + 1], h1[2 * d], 0x07060302u);
                    ax[xl][2][d] = __builtin_amdgcn_perm(h2[2 * d + 1], h2[2 * d], 0x07060302u);
                }
            }
        };
        auto mfmas_x = [&](const u32x4 (&ax)[NJ][3], const u32x4 (&ux)[NJ][3]) {
#pragma unroll
            for (int xl = 0; xl < NJ; ++xl) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, ax[xl][0]), a1 = __builtin_bit_cast(bf16x8, ax[xl][1]), a2 = __builtin_bit_cast(bf16x8, ax[xl][2]);
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, ux[xl][0]), b1 = __builtin_bit_cast(bf16x8, ux[xl][1]), b2 = __builtin_bit_cast(bf16x8, ux[xl][2]);
                acc[xl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[xl], 0, 0, 0);      // smallest terms first
                acc[xl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[xl], 0, 0, 0);
                acc[xl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[xl], 0, 0, 0);
                acc[xl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[xl], 0, 0, 0);
                acc[xl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[xl], 0, 0, 0);
                acc[xl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[xl], 0, 0, 0);
            }
        };
        // ---- prologue: windows 0 and 1 in LDS, slice 0 transformed and split ----
        issue_window(pvA, 0);
        issue_window(pvB, 1);
        issue_ux(ux0, 0);
        commit_window(pvA, 0);
        issue_window(pvA, 2);
        __syncthreads();
        read_dx(0);
        make_ax(axA);
        commit_window(pvB, 1);
        issue_window(pvB, 3);
        __syncthreads();
        // slice s: MFMAs on the fragments made one trip earlier, while slice s+1 is read, transformed and split
        auto xslice = [&](auto parity, int sl, u32x4 (&a_cur)[NJ][3], u32x4 (&a_nxt)[NJ][3], u32x4 (&u_cur)[NJ][3], u32x4 (&u_nxt)[NJ][3], f32x4 (&pv_c)[PITW]) {
            constexpr int P = decltype(parity)::value;     // window buffer that held slice s
            read_dx(P ^ 1);
            issue_ux(u_nxt, sl + 1);
            mfmas_x(a_cur, u_cur);
            make_ax(a_nxt);
#pragma unroll
            for (int i = 0; i < 6 * NJ; ++i) {              // the split of slice s+1 threaded between the MFMAs of slice s (a wave issues in order)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
            }
            commit_window(pv_c, P);                         // window of slice s+2 (last reader of this buffer: the read of slice s, one trip back)
            issue_window(pv_c, sl + 4);
            __syncthreads();
        };
        for (int sl = 0; sl < nslices; sl += 2) {
            xslice(std::integral_constant<int, 0>{}, sl, axA, axB, ux0, ux1, pvA);
            xslice(std::integral_constant<int, 1>{}, sl + 1, axB, axA, ux1, ux0, pvB);
        }
    } else {
    const int c8n = Cin >> 3;
    u32x4 ub0[NJ], ub1[NJ];
    auto issue_u = [&](u32x4 (&ub)[NJ], int h) {
#pragma unroll
        for (int xl = 0; xl < NJ; ++xl)
            ub[xl] = __builtin_amdgcn_raw_buffer_load_b128(rs_u, unsigned(lane) * 16u, ((NJ * wave + xl) * c8n + (h < nstages ? h : nstages - 1)) * 1024, 0);
    };
    f32x4 dra[NC], drb[NC];                             // window rows ra / rb, the wave's columns, this lane's four channels
    const float* const pdA[2] = {sWin + dA, sWin + win_floats + dA};
    const float* const pdB[2] = {sWin + dB, sWin + win_floats + dB};
    auto read_d = [&](int buf, int half) {             // buf and half are compile-time at every call: immediate offsets only
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            dra[c] = *reinterpret_cast<const f32x4*>(pdA[buf] + half * 8 + c * LP);
            drb[c] = *reinterpret_cast<const f32x4*>(pdB[buf] + half * 8 + c * LP);
        }
    };
    f32x4 af[NJ];                                       // A fragments of the next stage: V[wi][j], four channels
    auto make_af = [&]() {
        f32x4 m[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) m[c] = dra[c] + sg * drb[c];
        if constexpr (NJ == 4) {
            af[0] = m[0] - m[2];
            af[1] = m[1] + m[2];
            af[2] = m[2] - m[1];
            af[3] = m[1] - m[3];
        } else if (jh == 0) {                           // columns 0..2: V[i][0], V[i][1]
            af[0] = m[0] - m[2];
            af[1] = m[1] + m[2];
        } else {                                        // columns 1..3: V[i][2], V[i][3]
            af[0] = m[1] - m[0];
            af[1] = m[0] - m[2];
        }
    };
    auto mfmas = [&](const u32x4 (&ub)[NJ]) {
        f32x4 ac[NJ];
#pragma unroll
        for (int xl = 0; xl < NJ; ++xl) ac[xl] = af[xl];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int xl = 0; xl < NJ; ++xl)
                acc[xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[xl][e], __builtin_bit_cast(f32x4, ub[xl])[e], acc[xl], 0, 0, 0);
    };

    // ---- prologue ----
    issue_window(pvA, 0);
    issue_window(pvB, 1);
    issue_u(ub0, 0);
    issue_u(ub1, 1);
    commit_window(pvA, 0);
    issue_window(pvA, 2);
    __syncthreads();
    read_d(0, 0);
    make_af();
    // One slice = two stages; the reads of the next stage are in flight while the MFMAs of the current one issue.
    auto slice = [&](auto parity, int s, f32x4 (&pv_next)[PITW]) {
        constexpr int P = decltype(parity)::value;     // window buffer of slice s
        read_d(P, 1);
        commit_window(pv_next, P ^ 1);                  // window of slice s+1; its buffer was last read before the previous barrier
        mfmas(ub0);                                     // stage 2s
        issue_u(ub0, 2 * s + 2);
        make_af();
        __syncthreads();
        read_d(P ^ 1, 0);
        issue_window(pv_next, s + 3);
        mfmas(ub1);                                     // stage 2s + 1
        issue_u(ub1, 2 * s + 3);
        make_af();
    };
    for (int s = 0; s < nslices; s += 2) {
        slice(std::integral_constant<int, 0>{}, s, pvB);
        slice(std::integral_constant<int, 1>{}, s + 1, pvA);
    }
    }
    __syncthreads();                                    // every wave is done with the windows: the storage becomes sM

    // ---- output transform (as above: M through LDS in two halves) ----
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(a.out.n) * H * W - 1) * opitch * 4 + 32 * 4), 0x00020000);
    constexpr int OIT = (32 * 32 + NT - 1) / NT;
    float s0[OIT][4], s1[OIT][4];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        if ((wi >> 1) == half) {
#pragma unroll
            for (int xl = 0; xl < NJ; ++xl)
#pragma unroll
                for (int e = 0; e < 16; ++e) sM[((4 * (wi & 1) + NJ * jh + xl) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * MP + r] = acc[xl][e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < OIT; ++k) {
            const int it = tid + k * NT;
            const int t = it >> 5, c = it & 31;
            if (t < ntiles) {
                const float* const mb = sM + t * MP + c;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ma = mb[(0 + j) * 32 * MP], mbv = mb[(4 + j) * 32 * MP];
                    if (half == 0) { s0[k][j] = ma + mbv; s1[k][j] = mbv; }
                    else { s0[k][j] += ma; s1[k][j] = s1[k][j] - ma - mbv; }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < OIT; ++k) {
        const int it = tid + k * NT;
        const int t = it >> 5, c = it & 31;
        if (t >= ntiles) continue;
        const int tr = t / g.TC, tc = t - tr * g.TC;
        float y[2][2];
        y[0][0] = s0[k][0] + s0[k][1] + s0[k][2];
        y[0][1] = s0[k][1] - s0[k][2] - s0[k][3];
        y[1][0] = s1[k][0] + s1[k][1] + s1[k][2];
        y[1][1] = s1[k][1] - s1[k][2] - s1[k][3];
        const float bq = a.bias != nullptr ? a.bias[c] : 0.f;
        const bool tok = tr < ntr && tc < ntc;
        const int oy = 2 * (ty0 + tr), ox = 2 * (tx0 + tc);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v = y[dy][dx] + bq;
                if (a.relu) v = fmaxf(v, 0.f);
                const unsigned off = tok ? unsigned(((b * H + oy + dy) * W + ox + dx) * opitch + c) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, off, 0, 0);
            }
    }
}

struct WinoTile { int tr, tc; };
constexpr WinoTile kWinoTiles[kNumConvWinoTiles] = {{4, 7}, {2, 14}, {4, 8}, {2, 16}, {4, 7}, {2, 14}, {4, 8}, {2, 16}, {4, 7}, {2, 14}, {4, 8}, {2, 16}};      // 4..7: eight waves; 8..11: eight waves, bf16x6 products

static size_t wino_lds_bytes(int tr, int tc) {
    const size_t npx = size_t(2 * tr + 2) * (2 * tc + 2);
    const size_t stage = 2 * ((npx * 20 + 3) & ~size_t(3)) * 4, mtx = size_t(8) * 32 * 36 * 4;      // two windows; the epilogue passes M in two halves
    return stage > mtx ? stage : mtx;
}

bool ConvWinoEligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvWinoTiles) return false;
    if (a.kh != 3 || a.kw != 3 || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1) return false;
    if (a.in.f16 || a.out.f16 || a.in.f8 || a.out.f8 || a.pre_scale != nullptr || a.res.p != nullptr || a.wfrag == nullptr) return false;
    if (a.out.c != 32 || (a.in.c % 32) || a.in.c < 32 || (a.in.h & 1) || (a.in.w & 1)) return false;      // two 16-channel slices per loop trip
    if (a.out.h != a.in.h || a.out.w != a.in.w || a.out.n != a.in.n) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || (a.in.sw & 3) || (a.out.sw & 3)) return false;
    if (a.in.sh != a.in.w * a.in.sw || a.in.sn != a.in.h * a.in.sh || a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh) return false;
    if ((reinterpret_cast<uintptr_t>(a.in.p) & 15) || (reinterpret_cast<uintptr_t>(a.out.p) & 15) || (reinterpret_cast<uintptr_t>(a.wfrag) & 15)) return false;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return false;
    const int64_t M = int64_t(a.in.n) * a.in.h * a.in.w;
    if (M * a.in.sw * 4 >= (int64_t(1) << 31) || M * a.out.sw * 4 >= (int64_t(1) << 31)) return false;
    const WinoTile t = kWinoTiles[tile];
    if ((2 * t.tr + 2) * (2 * t.tc + 2) * 4 > 4 * 256) return false;      // window prefetch slots
    if (tile >= 8 && (a.w16 == nullptr || (reinterpret_cast<uintptr_t>(a.w16) & 15))) return false;      // the split U mirror (IE_FP32_SPLIT=1)
    return t.tr * t.tc <= 32 && wino_lds_bytes(t.tr, t.tc) <= size_t(64) * 1024;
}

hipError_t LaunchConvWino3x3(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvWinoEligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = 4 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + a.in.c);
    const WinoTile t = kWinoTiles[tile];
    WinoGeom g;
    g.TR = t.tr; g.TC = t.tc;
    g.TH = a.in.h / 2; g.TW = a.in.w / 2;
    g.bry = (g.TH + g.TR - 1) / g.TR;
    g.brx = (g.TW + g.TC - 1) / g.TC;
    const int64_t blocks = int64_t(a.in.n) * g.bry * g.brx;
    if (blocks >= (int64_t(1) << 31)) return hipErrorInvalidValue;
    if (tile >= 8) conv3x3_wino_kernel<8, true><<<dim3(unsigned(blocks)), dim3(512), wino_lds_bytes(t.tr, t.tc), stream>>>(a, g);
    else if (tile >= 4) conv3x3_wino_kernel<8><<<dim3(unsigned(blocks)), dim3(512), wino_lds_bytes(t.tr, t.tc), stream>>>(a, g);
    else conv3x3_wino_kernel<4><<<dim3(unsigned(blocks)), dim3(256), wino_lds_bytes(t.tr, t.tc), stream>>>(a, g);
    return hipGetLastError();
}

hipError_t InitKernelsWino() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)) != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}

}  // namespace ie
