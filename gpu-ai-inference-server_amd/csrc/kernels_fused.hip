// Fused dense-layer step for DenseNet-style blocks on small output grids (fp32): the 3x3 growth conv of layer L and the 1x1
// bottleneck conv of layer L+1 in ONE launch.
//
// Dense blocks 3-4 at batch 32 are a strict chain of 80 latency-bound launches (M = 6272 / 1568 pixels): about a third of the fp32
// forward is per-launch fixed cost (launch gap, operand preamble, first-load latency, store drain).  The chain cannot be shortened
// by running layers side by side, but two of its links need no grid-wide dependency: the 1x1 conv of layer L+1 is PER PIXEL, and of
// its K = C + 32 input channels only the last 32 come from layer L's 3x3 conv -- for the same pixels.  So a workgroup that owns a
// tile of 16*PB pixels
//   1. computes the 3x3 conv of layer L for ITS pixels (conv_win_kernel's scheme: the bottleneck window it needs -- one contiguous
//      run of 16*PB + 2W + 2 NHWC pixel rows written by the PREVIOUS launch -- through LDS, weights from the fragment-major mirror,
//      K split over the 8 waves, partial tiles summed through LDS in wave order), stores those 32 channels to the block buffer and
//      keeps them in LDS;
//   2. runs conv1x1_as_kernel's loop for layer L+1 on its pixel rows: channels [0, C) copied from the block buffer (the loads are
//      issued BEFORE step 1 and land while it runs), the 32 fresh ones from step 1, BN+ReLU prologue on the way into LDS, weights
//      streamed through a register ring, each wave 16 of the 128 output channels.
// No barrier between workgroups, no recomputed halo: launch k reads only what launch k-1 wrote.  The planner emits these steps for a
// run of dense layers (plan.cpp, "dense fusion"); a block of n layers becomes n+1 launches instead of 2n, and the bottleneck tensor
// ping-pongs between two buffers (one being read as halo by neighbouring tiles while the other is written).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "env.h"
#include "kernels.h"

namespace ie {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Pad (floats) of the LDS pixel rows.  The fragment reads are ds_read_b128 at row * pitch + k-group * 4: with a pad of 4 two lanes of
// every 16-lane service group share a 16-byte slot (2-way conflict on every read, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.42 in
// profiles/r02); pitch = 8 mod 32 banks is conflict-free for channel counts that are multiples of 32 (kernels_direct.hip, launch_as_t).
constexpr int kRowPad = 8;

// PB: 16-pixel blocks per workgroup.  8 waves.  1x1: Cout == 128 (16 per wave).  3x3: stride 1, pad 1, Cout == 32, 9 * Cin3 / 16 <= 72 chunks.
// OCC2: a variant meant to run TWO workgroups per CU (16-pixel tiles, <= 128 VGPRs, <= 80 KB of LDS): the old-channel loads are issued only
// after the 3x3's MFMAs (their registers would otherwise overlap the 3x3's weight fragments); the latency this exposes is what the
// second workgroup on the CU fills.
template <int PB, bool PRE, bool OCC2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(OCC2 ? 4 : 2, OCC2 ? 4 : 2)))
void conv_dense_fused_kernel(const ConvArgs a, const FusedArgs f, const int part_off) {
    constexpr int WAVES = 8, NT = 64 * WAVES, PX = 16 * PB, D = 8, MAXC3 = 9, TN3 = 2, PP = 32 + 4;
    constexpr int MAXS = PB == 1 ? 8 : 16;             // staging slots per thread for the old channels: PX * (K - 32) / 4 <= MAXS * NT
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_fused[];
    const int K = a.in.c, Kold = K - 32, P = K + kRowPad, CH = K >> 4, Cin3 = f.in3.c, P3 = Cin3 + kRowPad;
    float* const sA = reinterpret_cast<float*>(smem_fused);                        // [PX][P]: the 1x1's activation rows
    float* const sWin = sA;                                                        // [npx][P3]: bottleneck window of the 3x3; sA's rows are only
                                                                                   // written once every wave is done with the window
    float* const sPart = sA + part_off;                                            // [WAVES/2][PX][PP], behind max(sA, sWin)

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.in.h, W = a.in.w;
    const int M = a.in.n * H * W;
    const int m0 = blockIdx.x * PX;
    const int ipitch = int(a.in.sw), opitch = int(a.out.sw), bpitch = int(f.in3.sw);

    // ---- (a) bottleneck window loads (issued first: they are waited for first) ----
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(f.in3.p, 0, int((int64_t(M - 1) * bpitch + Cin3) * 4), 0x00020000);
    const int c4n3 = Cin3 >> 2;
    const int p_lo = m0 - W - 1;
    const int npx = PX + 2 * W + 2;
    const int items3 = npx * c4n3;
    constexpr int MAXW = 8;                            // window slots per thread: items3 <= MAXW * NT (checked by the launcher)
    u32x4 wv[MAXW];
    // (row, 4-channel column) of slot u = (tid + u * NT) / c4n3: ONE division, then scalar strides with a carry -- an fp32 kernel pays
    // for VALU instructions in MFMA slots and this workgroup is alone on its CU
    const int wq = NT / c4n3, wr = NT - wq * c4n3;     // wave-uniform
    int wrow[MAXW], wc4[MAXW];
    wrow[0] = tid / c4n3;
    wc4[0] = tid - wrow[0] * c4n3;
#pragma unroll
    for (int u = 1; u < MAXW; ++u) {
        const int c = wc4[u - 1] + wr;
        const bool carry = c >= c4n3;
        wc4[u] = carry ? c - c4n3 : c;
        wrow[u] = wrow[u - 1] + wq + (carry ? 1 : 0);
    }
#pragma unroll
    for (int u = 0; u < MAXW; ++u) {
        const int p = p_lo + wrow[u];
        wv[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (wrow[u] < npx && p >= 0 && p < M) ? unsigned(p * bpitch + wc4[u] * 4) * 4u : OOB, 0, 0);
    }

    // ---- (b) the 3x3's weight fragments of this wave's K slice (fragment-major: 1 KiB per load) ----
    const int cpt3 = Cin3 >> 4, total3 = 9 * cpt3;
    const int cb = int(int64_t(total3) * wave / WAVES), ce = int(int64_t(total3) * (wave + 1) / WAVES);
    const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f.wfrag3), 0, 32 * 9 * Cin3 * 4, 0x00020000);
    u32x4 B3[MAXC3][TN3];
#pragma unroll
    for (int i = 0; i < MAXC3; ++i) {
        const int ch = cb + i;
#pragma unroll
        for (int j = 0; j < TN3; ++j)
            B3[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w3, (ch < ce && !(a.debug & 4)) ? unsigned((j * total3 + ch) * 64 + lane) * 16u : OOB, 0, 0);
    }

    // ---- (c) the 1x1's old channels [0, Kold) of this tile's pixel rows: loads only, consumed after the 3x3.  A thread keeps ONE
    //      4-channel column for all its rows (rows advance by rpp per slot), so the prologue's scale/shift for that column is one pair
    //      of 16-byte loads issued here, not a dependent global round trip per slot later ----
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int((int64_t(M - 1) * ipitch + K) * 4), 0x00020000);
    const int c4n = Kold >> 2;
    const int rpp = NT / c4n;                          // rows per pass (>= 1: Kold / 4 <= 512, checked by the launcher)
    const int xrow0 = tid / c4n, xc4 = tid - xrow0 * c4n;
    const bool xact = xrow0 < rpp;                     // NT - rpp * c4n threads sit this phase out
    u32x4 xv[MAXS];
    auto load_old_channels = [&]() {
#pragma unroll
        for (int u = 0; u < MAXS; ++u) {
            const int row = xrow0 + u * rpp;
            const int p = m0 + row;
            xv[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (xact && row < PX && p < M && !(a.debug & 8)) ? unsigned(p * ipitch + xc4 * 4) * 4u : OOB, 0, 0);
        }
    };
    if constexpr (!OCC2) load_old_channels();
    f32x4 xsc = {1.f, 1.f, 1.f, 1.f}, xsf = {0.f, 0.f, 0.f, 0.f};
    f32x2 fsc = {1.f, 1.f}, fsf = {0.f, 0.f}, fb3 = {0.f, 0.f};
    const int fc2 = (tid & 15) * 2;                    // this thread's channel pair of the fresh 32 (phase i)
    if constexpr (PRE) {
        if (xact) {
            xsc = *reinterpret_cast<const f32x4*>(a.pre_scale + xc4 * 4);
            xsf = *reinterpret_cast<const f32x4*>(a.pre_shift + xc4 * 4);
        }
        fsc = *reinterpret_cast<const f32x2*>(a.pre_scale + Kold + fc2);
        fsf = *reinterpret_cast<const f32x2*>(a.pre_shift + Kold + fc2);
    }
    if (f.bias3 != nullptr) fb3 = *reinterpret_cast<const f32x2*>(f.bias3 + fc2);
    f32x4 ebias = {0.f, 0.f, 0.f, 0.f};                // the 1x1's epilogue bias for this lane's 4 output channels
    if (a.bias != nullptr) ebias = *reinterpret_cast<const f32x4*>(a.bias + wave * 16 + 4 * gk);

    // ---- (d) window -> LDS ----
#pragma unroll
    for (int u = 0; u < MAXW; ++u)
        if (wrow[u] < npx) *reinterpret_cast<u32x4*>(sWin + wrow[u] * P3 + wc4[u] * 4) = wv[u];
    __syncthreads();

    // ---- (e) 3x3 on 16 x 16 x 4 tiles: lane (r, gk) owns pixel r of each pixel block ----
    f32x4 acc3[PB][TN3];
    bool mok[PB];
    int oy[PB], ox[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int m = m0 + pb * 16 + r;
        mok[pb] = m < M;
        const int rem = (mok[pb] ? m : 0) % (H * W);
        oy[pb] = rem / W;
        ox[pb] = rem - oy[pb] * W;
#pragma unroll
        for (int j = 0; j < TN3; ++j) acc3[pb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // which of the nine taps fall inside the image for this lane's pixel: one bit each, tested in the chunk loop with a shift
    unsigned tapmask[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (mok[pb] && unsigned(oy[pb] + t / 3 - 1) < unsigned(H) && unsigned(ox[pb] + t % 3 - 1) < unsigned(W)) mk |= 1u << t;
        tapmask[pb] = mk;
    }
    const float* const wlane = sWin + r * P3 + gk * 4;
#pragma unroll
    for (int i = 0; i < MAXC3; ++i) {
        if (cb + i < ce && !(a.debug & 1)) {           // wave-uniform (debug bits: timing-only ablations, wrong results)
            const int ch = cb + i;
            const int tap = ch / cpt3, c0 = (ch - tap * cpt3) * 16;
            const int ky = tap / 3, kx = tap - ky * 3;
            const float* const wtap = wlane + (ky * W + kx) * P3 + c0;      // scalar offset on a loop-invariant lane address
            f32x4 av[PB];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                av[pb] = *reinterpret_cast<const f32x4*>(wtap + pb * 16 * P3);
                if (!((tapmask[pb] >> tap) & 1u)) av[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < TN3; ++j) {
                const f32x4 bv = __builtin_bit_cast(f32x4, B3[i][j]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) acc3[pb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[e], av[pb][e], acc3[pb][j], 0, 0, 0);
            }
        }
    }

    if constexpr (OCC2) load_old_channels();

    // ---- (f) prime the 1x1's weight ring (its latency hides behind the reduction and the staging below) ----
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, 128 * K * 4, 0x00020000);
    u32x4 ring[D];
    int c_l = 0;
    auto issue = [&](int slot) {
        // lane * 16 in the VGPR, the rest scalar; past the last chunk the last one again (in range, never consumed): no VALU in the K loop
        ring[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (a.debug & 16) ? OOB : unsigned(lane) * 16u, (wave * CH + (c_l < CH ? c_l : CH - 1)) * 1024, 0);
        ++c_l;
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);

    // ---- (g) sum the eight partial 3x3 tiles in wave order: waves 4..7 publish, waves 0..3 add theirs and publish ----
    if (wave >= WAVES / 2) {                           // (the partial tiles have their own storage: no need to wait for the window's readers)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int j = 0; j < TN3; ++j) *reinterpret_cast<f32x4*>(sPart + ((wave - WAVES / 2) * PX + pb * 16 + r) * PP + j * 16 + 4 * gk) = acc3[pb][j];
    }
    __syncthreads();
    if (wave < WAVES / 2) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int j = 0; j < TN3; ++j) {
                float* const q = sPart + (wave * PX + pb * 16 + r) * PP + j * 16 + 4 * gk;
                const f32x4 o = *reinterpret_cast<const f32x4*>(q);
                *reinterpret_cast<f32x4*>(q) = f32x4{acc3[pb][j][0] + o[0], acc3[pb][j][1] + o[1], acc3[pb][j][2] + o[2], acc3[pb][j][3] + o[3]};
            }
    }

    // ---- (h) old channels: prologue, -> sA ----
    if (xact) {
#pragma unroll
        for (int u = 0; u < MAXS; ++u) {
            const int row = xrow0 + u * rpp;
            if (row < PX) {
                f32x4 x = __builtin_bit_cast(f32x4, xv[u]);
                if constexpr (PRE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = x[e] * xsc[e] + xsf[e];
                        x[e] = a.pre_relu ? fmaxf(y, 0.f) : y;
                    }
                }
                *reinterpret_cast<f32x4*>(sA + row * P + xc4 * 4) = x;
            }
        }
    }
    __syncthreads();

    // ---- (i) fresh channels: final sum, 3x3 epilogue, raw value to the block buffer, prologue'd value to sA ----
    {
        const __amdgpu_buffer_rsrc_t rs_o3 = __builtin_amdgcn_make_buffer_rsrc(f.out3.p, 0, int((int64_t(M - 1) * int(f.out3.sw) + 32) * 4), 0x00020000);
        for (int idx = tid; idx < PX * 16; idx += NT) {
            const int p = idx >> 4, c2 = (idx & 15) * 2;
            f32x2 v = {0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WAVES / 2; ++w) {
                const f32x2 x = *reinterpret_cast<const f32x2*>(sPart + (w * PX + p) * PP + c2);
                v[0] += x[0];
                v[1] += x[1];
            }
            v[0] += fb3[0];
            v[1] += fb3[1];
            if (f.relu3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs_o3, m0 + p < M ? unsigned((m0 + p) * int(f.out3.sw) + c2) * 4u : OOB, 0, 0);
            if constexpr (PRE) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float y = v[e] * fsc[e] + fsf[e];
                    v[e] = a.pre_relu ? fmaxf(y, 0.f) : y;
                }
            }
            *reinterpret_cast<f32x2*>(sA + p * P + Kold + c2) = v;
        }
    }
    __syncthreads();

    // ---- (k) the 1x1: conv1x1_as_kernel's loop, 16 output channels per wave ----
    f32x4 acc[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* abase[PB];                            // fragment address per pixel block at the current ring trip; chunk = immediate offset
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) abase[pb] = sA + (pb * 16 + r) * P + gk * 4;
    f32x4 avn[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) avn[pb] = *reinterpret_cast<const f32x4*>(abase[pb]);
    auto compute = [&](int slot) {                     // the read ahead of the last chunk runs 16 floats past K (pad + next row / the partial tiles)
        f32x4 av[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) av[pb] = avn[pb];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) avn[pb] = *reinterpret_cast<const f32x4*>(abase[pb] + (slot + 1) * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
                acc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, ring[slot])[e], av[pb][e], acc[pb], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, PB, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * PB, 0);
    };
    const int full = (a.debug & 2) ? 0 : CH / D, rem = (a.debug & 2) ? 0 : CH - (CH / D) * D;
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
        if (s < rem) compute(s);

    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * opitch + 128) * 4), 0x00020000);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int m = m0 + pb * 16 + r;
        const int n = wave * 16 + 4 * gk;
        f32x4 v = acc[pb];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += ebias[e];
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, m < M ? unsigned(m * opitch + n) * 4u : OOB, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant (tiles 4 / 5 = 16 / 32 pixels).  In the kernel above the two convs run one after the other inside a
// workgroup and ~9 us of loads, barriers and LDS traffic overlap nothing.  Here both operand sets are staged up front (window AND the
// old channels of the 1x1's rows: they no longer share storage), then the workgroup splits: waves 0-3 run the 3x3 (K split four
// ways, weights through a 6-chunk register ring, partial tiles summed by the same four waves behind an LDS arrival counter) while
// waves 4-7 run the 1x1 over the OLD channels (32 output channels per wave), which do not depend on the 3x3; only the last two
// 16-channel chunks wait (second arrival counter) for the fresh 32 channels.  A SIMD hosts one wave of each kind, so the two MFMA
// streams interleave on the matrix pipe and each hides the other's fragment reads, weight waits and reductions.
// The LDS counters are bumped by every 3x3 wave unconditionally (no early exit), so the waits always end.
// ------------------------------------------------------------------------------------------------------------------------
template <int PB, bool PRE>
__global__ __launch_bounds__(512) void conv_dense_fused_ws_kernel(const ConvArgs a, const FusedArgs f, const int win_off, const int part_off) {
    constexpr int NT = 512, PX = 16 * PB, D = 8, D3 = 6, TN3 = 2, PP = 32 + 4, CW = 4;      // CW: waves per role
    constexpr int MAXS = PB == 1 ? 8 : 16, MAXW = 8;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_fused[];
    const int K = a.in.c, Kold = K - 32, P = K + kRowPad, CH = K >> 4, Cin3 = f.in3.c, P3 = Cin3 + kRowPad;
    float* const sA = reinterpret_cast<float*>(smem_fused);                        // [PX][P]
    float* const sWin = sA + win_off;                                              // [npx][P3]
    float* const sPart = sA + part_off;                                            // [CW][PX][PP]
    int* const sCnt = reinterpret_cast<int*>(sPart + CW * PX * PP);                // [2] arrival counters

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // role by wave half (waves i and i + 4 share a SIMD: measured, the parity split -- a.debug bit 32 -- is 8 % slower), so every SIMD hosts
    // one wave of each role
    const bool by_half = (a.debug & 32) == 0;
    const bool conv3 = by_half ? wave < CW : (wave & 1) == 0;      // wave-uniform role
    const int wr = by_half ? (conv3 ? wave : wave - CW) : (wave >> 1);   // index inside the role
    const int H = a.in.h, W = a.in.w;
    const int M = a.in.n * H * W;
    const int m0 = blockIdx.x * PX;
    const int ipitch = int(a.in.sw), opitch = int(a.out.sw), bpitch = int(f.in3.sw);

    // ---- phase 0 (all waves): window + old channels -> LDS; each role primes its weight ring ----
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(f.in3.p, 0, int((int64_t(M - 1) * bpitch + Cin3) * 4), 0x00020000);
    const int c4n3 = Cin3 >> 2;
    const int p_lo = m0 - W - 1;
    const int items3 = (PX + 2 * W + 2) * c4n3;
    u32x4 wv[MAXW];
#pragma unroll
    for (int u = 0; u < MAXW; ++u) {
        const int idx = tid + u * NT;
        const int row = idx / c4n3, c4 = idx - row * c4n3;
        const int p = p_lo + row;
        wv[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (idx < items3 && p >= 0 && p < M) ? unsigned(p * bpitch + c4 * 4) * 4u : OOB, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int((int64_t(M - 1) * ipitch + K) * 4), 0x00020000);
    const int c4n = Kold >> 2;
    const int rpp = NT / c4n;
    const int xrow0 = tid / c4n, xc4 = tid - xrow0 * c4n;
    const bool xact = xrow0 < rpp;
    u32x4 xv[MAXS];
#pragma unroll
    for (int u = 0; u < MAXS; ++u) {
        const int row = xrow0 + u * rpp;
        const int p = m0 + row;
        xv[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (xact && row < PX && p < M) ? unsigned(p * ipitch + xc4 * 4) * 4u : OOB, 0, 0);
    }
    // weight rings: one array, the role decides what it holds (3x3: D3 chunks x 2 channel blocks; 1x1: D chunks x 2 channel blocks)
    const int cpt3 = Cin3 >> 4, total3 = 9 * cpt3;
    const int cb = int(int64_t(total3) * wr / CW), ce = int(int64_t(total3) * (wr + 1) / CW);
    const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f.wfrag3), 0, 32 * 9 * Cin3 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, 128 * K * 4, 0x00020000);
    u32x4 ring[D][2];
    int c_l = 0;                                       // next chunk to load (role-relative)
    auto issue = [&](int slot) {
        if (conv3) {
            const int ch = cb + c_l;
#pragma unroll
            for (int j = 0; j < TN3; ++j)
                ring[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w3, ch < ce ? unsigned((j * total3 + ch) * 64 + lane) * 16u : OOB, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                ring[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, c_l < CH ? unsigned(((2 * wr + j) * CH + c_l) * 64 + lane) * 16u : OOB, 0, 0);
        }
        ++c_l;
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (!conv3 || s < D3) issue(s);                // wave-uniform
    f32x4 xsc = {1.f, 1.f, 1.f, 1.f}, xsf = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PRE) {
        if (xact) {
            xsc = *reinterpret_cast<const f32x4*>(a.pre_scale + xc4 * 4);
            xsf = *reinterpret_cast<const f32x4*>(a.pre_shift + xc4 * 4);
        }
    }
    if (tid < 2) sCnt[tid] = 0;
#pragma unroll
    for (int u = 0; u < MAXW; ++u) {
        const int idx = tid + u * NT;
        if (idx < items3) {
            const int row = idx / c4n3, c4 = idx - row * c4n3;
            *reinterpret_cast<u32x4*>(sWin + row * P3 + c4 * 4) = wv[u];
        }
    }
    if (xact) {
#pragma unroll
        for (int u = 0; u < MAXS; ++u) {
            const int row = xrow0 + u * rpp;
            if (row < PX) {
                f32x4 x = __builtin_bit_cast(f32x4, xv[u]);
                if constexpr (PRE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = x[e] * xsc[e] + xsf[e];
                        x[e] = a.pre_relu ? fmaxf(y, 0.f) : y;
                    }
                }
                *reinterpret_cast<f32x4*>(sA + row * P + xc4 * 4) = x;
            }
        }
    }
    __syncthreads();                                   // the only workgroup-wide barrier

    if (conv3) {
        // ================= 3x3 role =================
        f32x4 acc3[PB][TN3];
        bool mok[PB];
        int oy[PB], ox[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int m = m0 + pb * 16 + r;
            mok[pb] = m < M;
            const int rem = (mok[pb] ? m : 0) % (H * W);
            oy[pb] = rem / W;
            ox[pb] = rem - oy[pb] * W;
#pragma unroll
            for (int j = 0; j < TN3; ++j) acc3[pb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        int c_c = 0;
        auto compute3 = [&](int slot) {
            const int ch = cb + c_c;
            const int tap = ch / cpt3, c0 = (ch - tap * cpt3) * 16;
            const int ky = tap / 3, kx = tap - ky * 3;
            f32x4 av[PB];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const bool ok = mok[pb] && unsigned(oy[pb] + ky - 1) < unsigned(H) && unsigned(ox[pb] + kx - 1) < unsigned(W);
                av[pb] = *reinterpret_cast<const f32x4*>(sWin + (pb * 16 + r + ky * W + kx) * P3 + c0 + gk * 4);
                if (!ok) av[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < TN3; ++j) {
                const f32x4 bv = __builtin_bit_cast(f32x4, ring[slot][j]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) acc3[pb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[e], av[pb][e], acc3[pb][j], 0, 0, 0);
            }
            ++c_c;
        };
        const int n3 = ce - cb, full3 = n3 / D3, rem3 = n3 - full3 * D3;
        for (int it = 0; it < full3; ++it) {
#pragma unroll
            for (int s = 0; s < D3; ++s) {
                compute3(s);
                issue(s);
            }
        }
#pragma unroll
        for (int s = 0; s < D3; ++s)
            if (s < rem3) compute3(s);
        // publish this wave's partial tile, wait for the other three
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int j = 0; j < TN3; ++j) *reinterpret_cast<f32x4*>(sPart + (wr * PX + pb * 16 + r) * PP + j * 16 + 4 * gk) = acc3[pb][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // ds writes retired before the arrival is visible (workgroup scope: LDS only, no cache write-back)
        if (lane == 0) __hip_atomic_fetch_add(&sCnt[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&sCnt[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < CW) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // final sum of a quarter of the tile: bias / ReLU of the 3x3, raw value to the block buffer, prologue'd value into the 1x1's rows
        {
            const __amdgpu_buffer_rsrc_t rs_o3 = __builtin_amdgcn_make_buffer_rsrc(f.out3.p, 0, int((int64_t(M - 1) * int(f.out3.sw) + 32) * 4), 0x00020000);
            constexpr int ROWS = PX / CW;              // rows per 3x3 wave
            for (int idx = lane; idx < ROWS * 16; idx += 64) {
                const int p = wr * ROWS + (idx >> 4), c2 = (idx & 15) * 2;
                f32x2 v = {0.f, 0.f};
#pragma unroll
                for (int w = 0; w < CW; ++w) {
                    const f32x2 x = *reinterpret_cast<const f32x2*>(sPart + (w * PX + p) * PP + c2);
                    v[0] += x[0];
                    v[1] += x[1];
                }
                if (f.bias3 != nullptr) { v[0] += f.bias3[c2]; v[1] += f.bias3[c2 + 1]; }
                if (f.relu3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs_o3, m0 + p < M ? unsigned((m0 + p) * int(f.out3.sw) + c2) * 4u : OOB, 0, 0);
                if constexpr (PRE) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float y = v[e] * a.pre_scale[Kold + c2 + e] + a.pre_shift[Kold + c2 + e];
                        v[e] = a.pre_relu ? fmaxf(y, 0.f) : y;
                    }
                }
                *reinterpret_cast<f32x2*>(sA + p * P + Kold + c2) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(&sCnt[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }

    // ================= 1x1 role: 32 output channels per wave =================
    f32x4 acc[PB][2];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[pb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ebias[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (a.bias != nullptr) {
        ebias[0] = *reinterpret_cast<const f32x4*>(a.bias + wr * 32 + 4 * gk);
        ebias[1] = *reinterpret_cast<const f32x4*>(a.bias + wr * 32 + 16 + 4 * gk);
    }
    int c_c = 0;
    const float* const arow = sA + r * P + gk * 4;
    f32x4 avn[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) avn[pb] = *reinterpret_cast<const f32x4*>(arow + pb * 16 * P);
    auto compute = [&](int slot) {
        f32x4 av[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) av[pb] = avn[pb];
        if (c_c + 1 == CH - 2) {                       // the next fragment read is the first of the fresh channels: they must have landed
            while (__hip_atomic_load(&sCnt[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < CW) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const float* const nxt = arow + (c_c + 1 < CH ? c_c + 1 : c_c) * 16;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) avn[pb] = *reinterpret_cast<const f32x4*>(nxt + pb * 16 * P);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
                    acc[pb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, ring[slot][j])[e], av[pb][e], acc[pb][j], 0, 0, 0);
        ++c_c;
    };
    const int full = CH / D, rem = CH - full * D;
    for (int it = 0; it < full; ++it) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            compute(s);
            issue(s);
        }
    }
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < rem) compute(s);

    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * opitch + 128) * 4), 0x00020000);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int m = m0 + pb * 16 + r;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = wr * 32 + j * 16 + 4 * gk;
            f32x4 v = acc[pb][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += ebias[j][e];
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, m < M ? unsigned(m * opitch + n) * 4u : OOB, 0, 0);
        }
    }
}

// LDS of the wave-specialised variant: rows, window, partial tiles (4 waves) and two counters, nothing shared
static size_t fused_ws_lds_bytes(const ConvArgs& a, const FusedArgs& f, int pb, int* win_off = nullptr, int* part_off = nullptr) {
    const size_t px = size_t(16) * pb;
    const size_t rows = px * (a.in.c + kRowPad) * 4, win = (px + 2 * a.in.w + 2) * (f.in3.c + kRowPad) * 4, part = size_t(4) * px * 36 * 4;
    if (win_off) *win_off = int(rows / 4);
    if (part_off) *part_off = int((rows + win) / 4);
    return rows + win + part + 16;
}

// LDS: the 1x1's rows and the 3x3's window share storage (the rows are staged in registers until the window is dead), the partial
// tiles of the 3x3 sit behind them.  Returns the partial tiles' offset in floats through part_off.
static size_t fused_lds_bytes(const ConvArgs& a, const FusedArgs& f, int pb, int* part_off = nullptr) {
    const size_t px = size_t(16) * pb;
    const size_t rows = px * (a.in.c + kRowPad) * 4, win = (px + 2 * a.in.w + 2) * (f.in3.c + kRowPad) * 4, part = size_t(4) * px * 36 * 4;
    const size_t first = (rows > win ? rows : win);
    if (part_off) *part_off = int(first / 4);
    return first + part;
}

static bool dense(const TensorArg& t) { return t.sc == 1 && t.sh == t.w * t.sw && t.sn == t.h * t.sh; }

// tile: 1 / 2 = 16-pixel blocks per workgroup (one workgroup per CU); 3 = 16-pixel tiles in the two-workgroups-per-CU variant
bool ConvDenseFusedEligible(const ConvArgs& a, const FusedArgs& f, int tile) {
    if (tile < 1 || tile > 5) return false;
    const int pb = tile == 3 ? 1 : (tile >= 4 ? tile - 3 : tile);
    if (tile == 3 && fused_lds_bytes(a, f, 1) > size_t(80) * 1024) return false;
    if (tile >= 4 && (fused_ws_lds_bytes(a, f, pb) > size_t(160) * 1024 || 9 * (f.in3.c / 16) < 4 * 1)) return false;
    if (a.in.f16 || a.out.f16 || a.in.f8 || a.out.f8 || f.in3.f16 || f.out3.f16 || f.in3.f8 || f.out3.f8) return false;
    if (a.wfrag == nullptr || f.wfrag3 == nullptr || a.res.p != nullptr) return false;
    if (a.kh != 1 || a.kw != 1 || a.sh != 1 || a.sw != 1 || a.pt != 0 || a.pl != 0) return false;
    if (a.out.c != 128 || f.out3.c != 32 || a.in.c < 48 || (a.in.c % 16) || (f.in3.c % 16)) return false;
    const int total3 = 9 * (f.in3.c / 16);
    if (total3 > 72 || total3 < 8) return false;
    // same pixel grid everywhere
    for (const TensorArg* t : {&a.out, &f.in3, &f.out3})
        if (t->n != a.in.n || t->h != a.in.h || t->w != a.in.w) return false;
    if (!dense(a.in) || !dense(a.out) || !dense(f.in3) || !dense(f.out3)) return false;
    // the fresh 32 channels are the tail of the 1x1's input view, in the same buffer rows
    if (f.out3.sw != a.in.sw || f.out3.p != a.in.p + (a.in.c - 32)) return false;
    if ((a.in.sw % 4) || (a.out.sw % 4) || (f.in3.sw % 4) || (f.out3.sw % 2)) return false;
    for (const void* p : {static_cast<const void*>(a.in.p), static_cast<const void*>(a.out.p), static_cast<const void*>(f.in3.p), static_cast<const void*>(a.wfrag),
                          static_cast<const void*>(f.wfrag3)})
        if (reinterpret_cast<uintptr_t>(p) & 15) return false;
    if (reinterpret_cast<uintptr_t>(f.out3.p) & 7) return false;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return false;
    if (a.pre_scale && ((reinterpret_cast<uintptr_t>(a.pre_scale) & 15) || (reinterpret_cast<uintptr_t>(a.pre_shift) & 15))) return false;
    const int64_t M = int64_t(a.in.n) * a.in.h * a.in.w;
    if (M > 65536 || (M + 64) * a.in.sw * 4 >= (int64_t(1) << 31) || M * a.out.sw * 4 >= (int64_t(1) << 31) || (M + 64) * f.in3.sw * 4 >= (int64_t(1) << 31)) return false;
    const int px = 16 * pb;
    {   // staging slots: a thread keeps one 4-channel column, rows advance by rpp = 512 / columns per slot
        const int c4n = (a.in.c - 32) / 4;
        if (c4n > 512 || (px + 512 / c4n - 1) / (512 / c4n) > (pb == 1 ? 8 : 16)) return false;
    }
    if ((a.in.c - 32) % 4 || (reinterpret_cast<uintptr_t>(a.pre_scale) & 15) || (f.bias3 && (reinterpret_cast<uintptr_t>(f.bias3) & 7))) return false;
    if (int64_t(px + 2 * a.in.w + 2) * (f.in3.c / 4) > int64_t(8) * 512) return false;            // window slots
    return fused_lds_bytes(a, f, pb) <= size_t(160) * 1024;
}

hipError_t LaunchConvDenseFused(const ConvArgs& a_in, const FusedArgs& f, int tile, hipStream_t stream) {
    if (!ConvDenseFusedEligible(a_in, f, tile)) return hipErrorInvalidValue;
    const bool occ2 = tile == 3;
    const int pb = occ2 ? 1 : (tile >= 4 ? tile - 3 : tile);
    if (tile >= 4) {
        const int64_t Mw = int64_t(a_in.in.n) * a_in.in.h * a_in.in.w;
        const dim3 gridw(unsigned((Mw + 16 * pb - 1) / (16 * pb)));
        int win_off = 0, poff = 0;
        const size_t ldsw = fused_ws_lds_bytes(a_in, f, pb, &win_off, &poff);
        ConvArgs aw = a_in;
        const int dbgw = Knobs().debug_ablate;
        aw.debug = dbgw;
        if (pb == 1) {
            if (aw.pre_scale) conv_dense_fused_ws_kernel<1, true><<<gridw, dim3(512), ldsw, stream>>>(aw, f, win_off, poff);
            else conv_dense_fused_ws_kernel<1, false><<<gridw, dim3(512), ldsw, stream>>>(aw, f, win_off, poff);
        } else {
            if (aw.pre_scale) conv_dense_fused_ws_kernel<2, true><<<gridw, dim3(512), ldsw, stream>>>(aw, f, win_off, poff);
            else conv_dense_fused_ws_kernel<2, false><<<gridw, dim3(512), ldsw, stream>>>(aw, f, win_off, poff);
        }
        return hipGetLastError();
    }
    ConvArgs a = a_in;
    const int dbg = Knobs().debug_ablate;
    a.debug = dbg;      // timing-only ablations (wrong results): 1 no 3x3 MFMAs, 2 no 1x1 loop, 4 no 3x3 weight loads, 8 no old-channel loads, 16 no 1x1 weight loads
    const int64_t M = int64_t(a.in.n) * a.in.h * a.in.w;
    const dim3 grid(unsigned((M + 16 * pb - 1) / (16 * pb)));
    int part_off = 0;
    const size_t lds = fused_lds_bytes(a, f, pb, &part_off);
    if (occ2) {
        if (a.pre_scale) conv_dense_fused_kernel<1, true, true><<<grid, dim3(512), lds, stream>>>(a, f, part_off);
        else conv_dense_fused_kernel<1, false, true><<<grid, dim3(512), lds, stream>>>(a, f, part_off);
    } else if (pb == 1) {
        if (a.pre_scale) conv_dense_fused_kernel<1, true, false><<<grid, dim3(512), lds, stream>>>(a, f, part_off);
        else conv_dense_fused_kernel<1, false, false><<<grid, dim3(512), lds, stream>>>(a, f, part_off);
    } else {
        if (a.pre_scale) conv_dense_fused_kernel<2, true, false><<<grid, dim3(512), lds, stream>>>(a, f, part_off);
        else conv_dense_fused_kernel<2, false, false><<<grid, dim3(512), lds, stream>>>(a, f, part_off);
    }
    return hipGetLastError();
}

hipError_t InitKernelsFused() {
    hipError_t e;
#define IE_FUSED_ATTR(PB, PRE, OCC)                                                                                                                                 \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dense_fused_kernel<PB, PRE, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    IE_FUSED_ATTR(1, true, false) IE_FUSED_ATTR(1, false, false) IE_FUSED_ATTR(2, true, false) IE_FUSED_ATTR(2, false, false) IE_FUSED_ATTR(1, true, true) IE_FUSED_ATTR(1, false, true)
#undef IE_FUSED_ATTR
#define IE_FUSED_WS_ATTR(PB, PRE)                                                                                                                                   \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dense_fused_ws_kernel<PB, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    IE_FUSED_WS_ATTR(1, true) IE_FUSED_WS_ATTR(1, false) IE_FUSED_WS_ATTR(2, true) IE_FUSED_WS_ATTR(2, false)
#undef IE_FUSED_WS_ATTR
    return hipSuccess;
}

}  // namespace ie
