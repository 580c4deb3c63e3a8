// Winograd F(2x2, 3x3) convolution for the DenseNet growth convs (fp32, 3x3 / stride 1 / pad 1, 32 output channels, even H and W).
//
// The 3x3 convs are 48 % of DenseNet-121's FLOPs and the fp32 matrix pipe (157 TFLOP/s) is what bounds them.  F(2x2, 3x3) computes
// a 2x2 output tile from a 4x4 input tile with 16 multiplies per (input channel, output channel) instead of 36: 2.25x fewer MACs, at
// the price of an input transform V = B^T d B, an output transform Y = A^T M A and transformed weights U = G g G^T (built once at
// load, LaunchWinogradWeights).  Exact real arithmetic, fp32 rounding of the same order as a direct fp32 conv (the transform
// matrices hold only 0, +-1, +-1/2); the parity tests hold it to the same 2e-4 bound as every other fp32 kernel.
//
// A workgroup (4 waves) owns TR x TC output tiles (<= 32) of one image and walks Cin in slices of 16 channels:
//   * the (2TR+2) x (2TC+2) pixel window of the slice is register-prefetched one slice ahead and committed to LDS;
//   * every (tile, 4-channel quad) is transformed by one thread: 16 float4 reads of the window, 32 adds per channel, 16 float4
//     writes into sV[xi][tile][channel] (xi = the 16 positions of the transformed 4x4 tile);
//   * wave w multiplies the four positions xi = 4w .. 4w+3: M_xi[tile][cout] += V_xi[tile][:] . U_xi[:][cout] on v_mfma_f32_32x32x2_f32
//     (A fragments from sV, B fragments straight from the fragment-major U mirror in L2, prefetched one slice ahead);
//   * at the end the 16 accumulator tiles go through LDS once, every (tile, 4-cout quad) gets its 2x2 outputs (24 adds per channel),
//     bias / ReLU, and four 16-byte stores.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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

__global__ __launch_bounds__(256) void conv3x3_wino_kernel(const ConvArgs a, const WinoGeom g) {
    constexpr int NT = 256, CS = 16, LP = CS + 4, MP = 32 + 4, PITW = 4;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem_wino[];
    const int WR = 2 * g.TR + 2, WC = 2 * g.TC + 2, npx = WR * WC;
    float* const sWin = smem_wino;                      // [npx][LP]
    float* const sV = smem_wino + ((npx * LP + 3) & ~3);   // [16][32][LP]
    float* const sM = smem_wino;                        // [16][32][MP] after the K loop (aliases both)

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.in.c, H = a.in.h, W = a.in.w;
    const int ipitch = int(a.in.sw), opitch = int(a.out.sw);
    int bid = blockIdx.x;
    const int bx = bid % g.brx; bid /= g.brx;
    const int by = bid % g.bry;
    const int b = bid / g.bry;
    const int ty0 = by * g.TR, tx0 = bx * g.TC;
    const int ntr = min(g.TR, g.TH - ty0), ntc = min(g.TC, g.TW - tx0);      // valid tiles of this block
    const int y0 = 2 * ty0 - 1, x0 = 2 * tx0 - 1;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, 16 * 32 * Cin * 4, 0x00020000);

    // ---- window prefetch: item = (pixel, quad of the 16-channel slice) ----
    const int items = npx * 4;
    int woff[PITW];       // element offset of (pixel, quad) in the input view, or -1
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
    f32x4 pv[PITW];
    auto issue_window = [&](int s) {
#pragma unroll
        for (int i = 0; i < PITW; ++i)
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, woff[i] >= 0 ? unsigned(woff[i] + s * CS) * 4u : OOB, 0, 0));
    };
    auto commit_window = [&]() {
#pragma unroll
        for (int i = 0; i < PITW; ++i) {
            const int it = tid + i * NT;
            if (it < items) *reinterpret_cast<f32x4*>(sWin + (it >> 2) * LP + (it & 3) * 4) = pv[i];
        }
    };
    // ---- U fragments of this wave's four positions, one slice ahead: [slot][xi_local][kk] ----
    u32x4 ub[2][4][2];
    const int c8n = Cin >> 3;
    auto issue_u = [&](int s, int slot) {
#pragma unroll
        for (int xl = 0; xl < 4; ++xl)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                ub[slot][xl][kk] = __builtin_amdgcn_raw_buffer_load_b128(rs_u, unsigned(((4 * wave + xl) * c8n + 2 * s + kk) * 64 + lane) * 16u, 0, 0);
    };

    f32x16 acc[4];
#pragma unroll
    for (int xl = 0; xl < 4; ++xl)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xl][e] = 0.f;

    const int nslices = Cin / CS;
    issue_window(0);
    issue_u(0, 0);
    // this thread's transform item(s): (tile, quad); tiles are numbered row-major inside the block's TR x TC grid
    const int ntiles = g.TR * g.TC;
    for (int s = 0; s < nslices; ++s) {
        const int slot = s & 1;
        commit_window();
        __syncthreads();                                // window of slice s visible; sV free (MFMAs of slice s-1 passed the barrier below)
        if (s + 1 < nslices) {
            issue_window(s + 1);
            issue_u(s + 1, slot ^ 1);
        }
        // ---- input transform V = B^T d B ----
        for (int it = tid; it < ntiles * 4; it += NT) {
            const int t = it >> 2, q = it & 3;
            const int tr = t / g.TC, tc = t - tr * g.TC;
            const float* const base = sWin + ((2 * tr) * WC + 2 * tc) * LP + q * 4;
            f32x4 d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x4*>(base + (i * WC + j) * LP);
            f32x4 m[4][4];      // B^T d
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m[0][j] = d[0][j] - d[2][j];
                m[1][j] = d[1][j] + d[2][j];
                m[2][j] = d[2][j] - d[1][j];
                m[3][j] = d[1][j] - d[3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v0 = m[i][0] - m[i][2], v1 = m[i][1] + m[i][2], v2 = m[i][2] - m[i][1], v3 = m[i][1] - m[i][3];
                float* const o = sV + ((4 * i) * 32 + t) * LP + q * 4;
                *reinterpret_cast<f32x4*>(o) = v0;
                *reinterpret_cast<f32x4*>(o + 32 * LP) = v1;
                *reinterpret_cast<f32x4*>(o + 64 * LP) = v2;
                *reinterpret_cast<f32x4*>(o + 96 * LP) = v3;
            }
        }
        __syncthreads();                                // sV of slice s visible; the window may be overwritten
        // ---- 4 positions x 2 chunks of 8 channels: 32 MFMAs ----
#pragma unroll
        for (int xl = 0; xl < 4; ++xl) {
            const float* const A = sV + ((4 * wave + xl) * 32 + r) * LP + hh * 4;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(A + kk * 8);
                const f32x4 bf = __builtin_bit_cast(f32x4, ub[slot][xl][kk]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc[xl], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                    // every wave is done with sV / sWin: their storage becomes sM

    // ---- M -> LDS: acc[xl][e] is (tile row = (e&3) + 8*(e>>2) + 4*hh, cout = r) of position xi = 4*wave + xl ----
#pragma unroll
    for (int xl = 0; xl < 4; ++xl)
#pragma unroll
        for (int e = 0; e < 16; ++e) sM[((4 * wave + xl) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * MP + r] = acc[xl][e];
    __syncthreads();

    // ---- output transform Y = A^T M A, bias, ReLU, 2x2 pixels x 4 channels per item ----
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(a.out.n) * H * W - 1) * opitch * 4 + 32 * 4), 0x00020000);
    for (int it = tid; it < ntiles * 8; it += NT) {
        const int t = it >> 3, q = it & 7;
        const int tr = t / g.TC, tc = t - tr * g.TC;
        f32x4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const f32x4*>(sM + ((4 * i + j) * 32 + t) * MP + q * 4);
        f32x4 s0[4], s1[4];     // A^T m
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = m[0][j] + m[1][j] + m[2][j];
            s1[j] = m[1][j] - m[2][j] - m[3][j];
        }
        f32x4 y[2][2];
        y[0][0] = s0[0] + s0[1] + s0[2];
        y[0][1] = s0[1] - s0[2] - s0[3];
        y[1][0] = s1[0] + s1[1] + s1[2];
        y[1][1] = s1[1] - s1[2] - s1[3];
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) bq = *reinterpret_cast<const f32x4*>(a.bias + q * 4);
        const bool tok = tr < ntr && tc < ntc;
        const int oy = 2 * (ty0 + tr), ox = 2 * (tx0 + tc);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                f32x4 v = y[dy][dx];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] += bq[e];
                    if (a.relu) v[e] = fmaxf(v[e], 0.f);
                }
                const unsigned off = tok ? unsigned(((b * H + oy + dy) * W + ox + dx) * opitch + q * 4) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, off, 0, 0);
            }
    }
}

struct WinoTile { int tr, tc; };
constexpr WinoTile kWinoTiles[kNumConvWinoTiles] = {{4, 7}, {2, 14}, {4, 8}, {2, 16}};

static size_t wino_lds_bytes(int tr, int tc) {
    const size_t npx = size_t(2 * tr + 2) * (2 * tc + 2);
    const size_t stage = ((npx * 20 + 3) & ~size_t(3)) * 4 + size_t(16) * 32 * 20 * 4, mtx = size_t(16) * 32 * 36 * 4;
    return stage > mtx ? stage : mtx;
}

bool ConvWinoEligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvWinoTiles) return false;
    if (a.kh != 3 || a.kw != 3 || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1) return false;
    if (a.in.f16 || a.out.f16 || a.in.f8 || a.out.f8 || a.pre_scale != nullptr || a.res.p != nullptr || a.wfrag == nullptr) return false;
    if (a.out.c != 32 || (a.in.c % 16) || a.in.c < 16 || (a.in.h & 1) || (a.in.w & 1)) return false;
    if (a.out.h != a.in.h || a.out.w != a.in.w || a.out.n != a.in.n) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || (a.in.sw & 3) || (a.out.sw & 3)) return false;
    if (a.in.sh != a.in.w * a.in.sw || a.in.sn != a.in.h * a.in.sh || a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh) return false;
    if ((reinterpret_cast<uintptr_t>(a.in.p) & 15) || (reinterpret_cast<uintptr_t>(a.out.p) & 15) || (reinterpret_cast<uintptr_t>(a.wfrag) & 15)) return false;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return false;
    const int64_t M = int64_t(a.in.n) * a.in.h * a.in.w;
    if (M * a.in.sw * 4 >= (int64_t(1) << 31) || M * a.out.sw * 4 >= (int64_t(1) << 31)) return false;
    const WinoTile t = kWinoTiles[tile];
    if ((2 * t.tr + 2) * (2 * t.tc + 2) * 4 > 4 * 256) return false;      // window prefetch slots
    return wino_lds_bytes(t.tr, t.tc) <= size_t(160) * 1024;
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
    conv3x3_wino_kernel<<<dim3(unsigned(blocks)), dim3(256), wino_lds_bytes(t.tr, t.tc), stream>>>(a, g);
    return hipGetLastError();
}

hipError_t InitKernelsWino() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace ie
