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
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wino_kernel(const ConvArgs a, const WinoGeom g) {
    constexpr int NT = 256, CS = 16, LP = CS + 4, MP = 32 + 4, PITW = 4;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem_wino[];
    const int WR = 2 * g.TR + 2, WC = 2 * g.TC + 2, npx = WR * WC;
    float* const sWin = smem_wino;                      // [npx][LP]
    float* const sV = smem_wino + ((npx * LP + 3) & ~3);   // [16][32][LP]
    float* const sM = smem_wino;                        // [16][32][MP] after the K loop (aliases both)

    if (a.debug & 128) return;
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
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (woff[i] >= 0 && !(a.debug & 4)) ? unsigned(woff[i] + s * CS) * 4u : OOB, 0, 0));
    };
    auto commit_window = [&]() {
#pragma unroll
        for (int i = 0; i < PITW; ++i) {
            const int it = tid + i * NT;
            if (it < items) *reinterpret_cast<f32x4*>(sWin + (it >> 2) * LP + (it & 3) * 4) = pv[i];
        }
    };
    // ---- U fragments of this wave's four positions, one slice ahead: [slot][xi_local][kk] ----
    u32x4 ub[4][2];
    const int c8n = Cin >> 3;
    auto issue_u = [&](int s) {           // issued before the slice's transform: the latency hides behind it
#pragma unroll
        for (int xl = 0; xl < 4; ++xl)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                ub[xl][kk] = __builtin_amdgcn_raw_buffer_load_b128(rs_u, unsigned(((4 * wave + xl) * c8n + 2 * s + kk) * 64 + lane) * 16u, 0, 0);
    };

    f32x16 acc[4];
#pragma unroll
    for (int xl = 0; xl < 4; ++xl)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xl][e] = 0.f;

    const int nslices = (a.debug & 32) ? 0 : Cin / CS;
    issue_window(0);
    if (const int dly = (a.debug >> 8) & 0xff) {        // experiment: de-phase the two workgroups of a CU
        if (tid == 0) smem_wino[0] = __builtin_bit_cast(float, int(__builtin_amdgcn_s_getreg((3 << 11) | 4)));
        __syncthreads();
        const int slot = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, smem_wino[0]));
        __syncthreads();
        if (slot & 1)
            for (int i = 0; i < dly; ++i) __builtin_amdgcn_s_sleep(4);
    }
    // this thread's transform item(s): (tile, quad); tiles are numbered row-major inside the block's TR x TC grid
    const int ntiles = g.TR * g.TC;
    for (int s = 0; s < nslices; ++s) {
        commit_window();
        __syncthreads();                                // window of slice s visible; sV free (MFMAs of slice s-1 passed the barrier below)
        if (s + 1 < nslices) issue_window(s + 1);
        issue_u(s);
        // ---- input transform V = B^T d B: one (tile, channel pair) per thread-item: all four waves busy, 8-byte LDS accesses ----
        for (int it = tid; it < ((a.debug & 1) ? 0 : ntiles * (CS / 2)); it += NT) {      // debug bits: timing-only ablations
            const int t = it >> 3, c = (it & 7) * 2;
            const int tr = t / g.TC, tc = t - tr * g.TC;
            const float* const base = sWin + ((2 * tr) * WC + 2 * tc) * LP + c;
            f32x2 m[4][4];      // B^T d, column by column
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 d0 = *reinterpret_cast<const f32x2*>(base + j * LP), d1 = *reinterpret_cast<const f32x2*>(base + (WC + j) * LP),
                            d2 = *reinterpret_cast<const f32x2*>(base + (2 * WC + j) * LP), d3 = *reinterpret_cast<const f32x2*>(base + (3 * WC + j) * LP);
                m[0][j] = d0 - d2;
                m[1][j] = d1 + d2;
                m[2][j] = d2 - d1;
                m[3][j] = d1 - d3;
            }
            float* const o = sV + t * LP + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x2*>(o + (4 * i + 0) * 32 * LP) = m[i][0] - m[i][2];
                *reinterpret_cast<f32x2*>(o + (4 * i + 1) * 32 * LP) = m[i][1] + m[i][2];
                *reinterpret_cast<f32x2*>(o + (4 * i + 2) * 32 * LP) = m[i][2] - m[i][1];
                *reinterpret_cast<f32x2*>(o + (4 * i + 3) * 32 * LP) = m[i][1] - m[i][3];
            }
        }
        __syncthreads();                                // sV of slice s visible; the window may be overwritten
        // ---- 4 positions x 2 chunks of 8 channels: 32 MFMAs ----
#pragma unroll
        for (int xl = 0; xl < ((a.debug & 2) ? 0 : 4); ++xl) {
            const float* const A = sV + ((4 * wave + xl) * 32 + r) * LP + hh * 4;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(A + kk * 8);
                const f32x4 bf = __builtin_bit_cast(f32x4, ub[xl][kk]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc[xl], 0, 0, 0);
            }
        }
    }
    // ---- output transform Y = A^T M A.  The 16 accumulator tiles pass through LDS in two halves of 8 positions (rows i = 0,1 then
    //      i = 2,3 of the 4x4 position grid), so the epilogue needs no more LDS than the K loop (3 workgroups per CU); the partial
    //      sums over i are carried in registers between the halves: s0 = m0 + m1 + m2, s1 = m1 - m2 - m3 per column j ----
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(a.out.n) * H * W - 1) * opitch * 4 + 32 * 4), 0x00020000);
    constexpr int OIT = (32 * 32 + NT - 1) / NT;       // (tile, cout) items per thread
    float s0[OIT][4], s1[OIT][4];
#pragma unroll
    for (int half = (a.debug & 16) ? 2 : 0; half < 2; ++half) {
        __syncthreads();                                // previous users of the storage are done
        if ((wave >> 1) == half) {                      // waves 2*half, 2*half+1 hold positions 8*half .. 8*half+7
#pragma unroll
            for (int xl = 0; xl < 4; ++xl)
#pragma unroll
                for (int e = 0; e < 16; ++e) sM[((4 * (wave & 1) + xl) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * MP + r] = acc[xl][e];
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
                    const float ma = mb[(0 + j) * 32 * MP], mbv = mb[(4 + j) * 32 * MP];      // rows i = 2*half, 2*half + 1 of column j
                    if (half == 0) { s0[k][j] = ma + mbv; s1[k][j] = mbv; }                  // m0 + m1 ; m1
                    else { s0[k][j] += ma; s1[k][j] = s1[k][j] - ma - mbv; }                 // + m2 ; - m2 - m3
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
                const unsigned off = (tok && !(a.debug & 8)) ? unsigned(((b * H + oy + dy) * W + ox + dx) * opitch + c) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, off, 0, 0);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Pipelined variant (tiles 4..7).  Ablations of the kernel above on block 1 (64.7 us): MFMAs 27.5 us (the 2.25x-reduced work is
// MFMA-bound at ~24 us), input transform 9 us, and 27 us of barriers / LDS commits / epilogue -- and the three ADD UP: the phases
// of a workgroup do not overlap, and the two workgroups of a CU run in lockstep.  Here Cin is walked in stages of 8 channels with
// sV double-buffered: in stage h every wave issues the MFMAs of stage h (16 per wave) AND transforms stage h+1 into the other sV
// buffer, one barrier per stage; the window (16 channels, two stages) is double-buffered as well and committed every other stage.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wino_pipe_kernel(const ConvArgs a, const WinoGeom g) {
    constexpr int NT = 256, CS = 16, LP = CS + 4, VP = 8 + 4, MP = 32 + 4, PITW = 4;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem_wino[];
    const int WR = 2 * g.TR + 2, WC = 2 * g.TC + 2, npx = WR * WC;
    const int win_floats = (npx * LP + 3) & ~3;
    float* const sWin = smem_wino;                      // [2][npx][LP]
    float* const sV = smem_wino + 2 * win_floats;       // [2][16][32][VP]
    float* const sM = smem_wino;                        // [8][32][MP] in the epilogue (aliases)
    constexpr int VBUF = 16 * 32 * VP;

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
    f32x4 pvA[PITW], pvB[PITW];                         // windows in flight: even / odd slices, issued two slices (four stages) ahead
    const int nslices = Cin / CS, nstages = 2 * nslices;
    auto issue_window = [&](f32x4 (&pv)[PITW], int s) {
#pragma unroll
        for (int i = 0; i < PITW; ++i)
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (woff[i] >= 0 && s < nslices && !(a.debug & 32)) ? unsigned(woff[i] + s * CS) * 4u : OOB, 0, 0));
    };
    auto commit_window = [&](const f32x4 (&pv)[PITW], int buf) {
        float* const w = sWin + buf * win_floats;
#pragma unroll
        for (int i = 0; i < PITW; ++i) {
            const int it = tid + i * NT;
            if (it < items) *reinterpret_cast<f32x4*>(w + (it >> 2) * LP + (it & 3) * 4) = pv[i];
        }
    };
    // this thread's transform item: (tile, channel of the 8-channel stage).  Threads beyond the block's tiles transform in-allocation
    // garbage into sV rows >= ntiles, whose accumulator rows the epilogue never reads: no predicate, no branch in the stage body.
    const int tt = tid >> 3, tcn = tid & 7;
    const int ttr = tt / g.TC, ttc = tt - ttr * g.TC;
    const int tbase = ((2 * ttr) * WC + 2 * ttc) * LP + tcn;
    const int obase = tt * VP + tcn;
    const int abase = ((4 * wave) * 32 + r) * VP + hh * 4;
    const int c8n = Cin >> 3;
    u32x4 ub0[4], ub1[4];                               // U fragments of this wave's four positions: stage parity 0 / 1
    auto issue_u = [&](u32x4 (&ub)[4], int h) {
#pragma unroll
        for (int xl = 0; xl < 4; ++xl)
            ub[xl] = __builtin_amdgcn_raw_buffer_load_b128(rs_u, h < nstages ? unsigned(((4 * wave + xl) * c8n + ((a.debug & 4) ? 0 : h)) * 64 + lane) * 16u : OOB, 0, 0);
    };

    f32x16 acc[4];
#pragma unroll
    for (int xl = 0; xl < 4; ++xl)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xl][e] = 0.f;

    // One stage: the 16 MFMAs of stage h out of sV[P], and in their issue shadow (a wave issues in order, so the interleaving is
    // spelled out and pinned with sched_barrier) the transform of stage h+1 into sV[P^1], the U prefetch and the window commit.
    const bool dbg_nomfma = a.debug & 2, dbg_notr = a.debug & 1;
    auto stage = [&](auto parity, int h, u32x4 (&ub)[4], f32x4 (&pv)[PITW]) {
        constexpr int P = decltype(parity)::value;
        const float* const A = sV + P * VBUF + abase;
        f32x4 af[4];
#pragma unroll
        for (int xl = 0; xl < 4; ++xl) af[xl] = *reinterpret_cast<const f32x4*>(A + xl * 32 * VP);
        const float* const base = sWin + (((h + 1) >> 1) & 1) * win_floats + tbase + (P ^ 1) * 8;
        float* const o = sV + (P ^ 1) * VBUF + obase;
        float d[4][4];
#pragma unroll
        for (int step = 0; step < 16; ++step) {
            const int xl = step & 3, e = step >> 2;
            if (!dbg_nomfma) acc[xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[xl][e], __builtin_bit_cast(f32x4, ub[xl])[e], acc[xl], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (dbg_notr) {
            } else if (step < 4) {
                const int j = step;
                d[0][j] = base[j * LP];
                d[1][j] = base[(WC + j) * LP];
                d[2][j] = base[(2 * WC + j) * LP];
                d[3][j] = base[(3 * WC + j) * LP];
            } else if (step >= 8 && step < 12) {
                const int i = step - 8;
                float m[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    m[j] = i == 0 ? d[0][j] - d[2][j] : i == 1 ? d[1][j] + d[2][j] : i == 2 ? d[2][j] - d[1][j] : d[1][j] - d[3][j];
                o[(4 * i + 0) * 32 * VP] = m[0] - m[2];
                o[(4 * i + 1) * 32 * VP] = m[1] + m[2];
                o[(4 * i + 2) * 32 * VP] = m[2] - m[1];
                o[(4 * i + 3) * 32 * VP] = m[1] - m[3];
            } else if (step == 12) {
                if (P) commit_window(pv, ((h + 3) >> 1) & 1);  // window of slice (h+3)/2 into the buffer last read in iteration h-1
            } else if (step == 13) {
                if (P) issue_window(pv, (h + 7) >> 1);         // and the slice two further on into the registers just freed
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        issue_u(ub, h + 2);                             // after the last use of ub by this stage
        __syncthreads();
    };

    // ---- prologue: window 0 -> LDS, stage 0 transformed, window 1 committed ----
    issue_window(pvA, 0);
    issue_window(pvB, 1);
    issue_u(ub0, 0);
    issue_u(ub1, 1);
    commit_window(pvA, 0);
    issue_window(pvA, 2);
    __syncthreads();
    {
        const float* const base = sWin + tbase;
        float* const o = sV + obase;
        float m[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d0 = base[j * LP], d1 = base[(WC + j) * LP], d2 = base[(2 * WC + j) * LP], d3 = base[(3 * WC + j) * LP];
            m[0][j] = d0 - d2;
            m[1][j] = d1 + d2;
            m[2][j] = d2 - d1;
            m[3][j] = d1 - d3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[(4 * i + 0) * 32 * VP] = m[i][0] - m[i][2];
            o[(4 * i + 1) * 32 * VP] = m[i][1] + m[i][2];
            o[(4 * i + 2) * 32 * VP] = m[i][2] - m[i][1];
            o[(4 * i + 3) * 32 * VP] = m[i][1] - m[i][3];
        }
    }
    commit_window(pvB, 1);
    issue_window(pvB, 3);
    __syncthreads();
    for (int h = 0; h < nstages; h += 4) {              // two slices per trip (Cin % 32 == 0): stage 4k+1 commits slice 2k+2, stage 4k+3 slice 2k+3
        stage(std::integral_constant<int, 0>{}, h, ub0, pvA);
        stage(std::integral_constant<int, 1>{}, h + 1, ub1, pvA);
        stage(std::integral_constant<int, 0>{}, h + 2, ub0, pvB);
        stage(std::integral_constant<int, 1>{}, h + 3, ub1, pvB);
    }

    // ---- output transform (as above: M through LDS in two halves) ----
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(a.out.n) * H * W - 1) * opitch * 4 + 32 * 4), 0x00020000);
    constexpr int OIT = (32 * 32 + NT - 1) / NT;
    float s0[OIT][4], s1[OIT][4];
#pragma unroll
    for (int half = (a.debug & 16) ? 2 : 0; half < 2; ++half) {
        if (half) __syncthreads();
        if ((wave >> 1) == half) {
#pragma unroll
            for (int xl = 0; xl < 4; ++xl)
#pragma unroll
                for (int e = 0; e < 16; ++e) sM[((4 * (wave & 1) + xl) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * MP + r] = acc[xl][e];
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
                const unsigned off = (tok && !(a.debug & 8)) ? unsigned(((b * H + oy + dy) * W + ox + dx) * opitch + c) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, off, 0, 0);
            }
    }
}

struct WinoTile { int tr, tc; };
constexpr WinoTile kWinoTiles[kNumConvWinoTiles] = {{4, 7}, {2, 14}, {4, 8}, {2, 16}, {4, 7}, {2, 14}, {4, 8}, {2, 16}};      // 4..7: the pipelined kernel

static size_t wino_pipe_lds_bytes(int tr, int tc) {
    const size_t npx = size_t(2 * tr + 2) * (2 * tc + 2);
    const size_t stage = 2 * ((npx * 20 + 3) & ~size_t(3)) * 4 + size_t(2) * 16 * 32 * 12 * 4, mtx = size_t(8) * 32 * 36 * 4;
    return stage > mtx ? stage : mtx;
}

static size_t wino_lds_bytes(int tr, int tc) {
    const size_t npx = size_t(2 * tr + 2) * (2 * tc + 2);
    const size_t stage = ((npx * 20 + 3) & ~size_t(3)) * 4 + size_t(16) * 32 * 20 * 4, mtx = size_t(8) * 32 * 36 * 4;      // the epilogue passes M in two halves
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
    if (tile >= 4 && (t.tr * t.tc * 8 > 256 || (a.in.c % 32))) return false;     // one transform item per thread per stage; two slices per loop trip
    return (tile >= 4 ? wino_pipe_lds_bytes(t.tr, t.tc) : wino_lds_bytes(t.tr, t.tc)) <= size_t(160) * 1024;
}

hipError_t LaunchConvWino3x3(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvWinoEligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = 4 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + a.in.c);
    static const int dbg = [] { const char* e = std::getenv("IE_DEBUG_ABLATE"); return e ? std::atoi(e) : 0; }();
    a.debug = dbg;          // timing-only ablations (wrong results): 1 no input transform, 2 no MFMAs, 4 no window loads, 8 no stores
    const WinoTile t = kWinoTiles[tile];
    WinoGeom g;
    g.TR = t.tr; g.TC = t.tc;
    g.TH = a.in.h / 2; g.TW = a.in.w / 2;
    g.bry = (g.TH + g.TR - 1) / g.TR;
    g.brx = (g.TW + g.TC - 1) / g.TC;
    const int64_t blocks = int64_t(a.in.n) * g.bry * g.brx;
    if (blocks >= (int64_t(1) << 31)) return hipErrorInvalidValue;
    if (tile >= 4) conv3x3_wino_pipe_kernel<<<dim3(unsigned(blocks)), dim3(256), wino_pipe_lds_bytes(t.tr, t.tc), stream>>>(a, g);
    else conv3x3_wino_kernel<<<dim3(unsigned(blocks)), dim3(256), wino_lds_bytes(t.tr, t.tc), stream>>>(a, g);
    return hipGetLastError();
}

hipError_t InitKernelsWino() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace ie
