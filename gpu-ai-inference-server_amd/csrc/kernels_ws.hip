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

#ifndef WS_ABLATE
#define WS_ABLATE 0       // scripts/probes/ws_probe.cpp builds variants with parts of conv1x1_ws_f16_kernel switched off (timing only, wrong results)
#endif

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// Register chunks (32 channels x 32 pixels = 2 KiB each) a wave keeps in flight: 4 beside 64 accumulator registers, 6 otherwise.
constexpr int ws_ring_depth(int tn) { return tn >= 4 ? 4 : 6; }

// FAST: half output, no residual -- the epilogue DenseNet's bottleneck and transition convs take.  The accumulators START at the bias (16-byte LDS
// reads straight into them at the top of a row block: no zeroing, no bias add), the results are rounded to half first and the ReLU is one packed max
// per two values: ~100 instructions per 32 x 128 outputs where the general epilogue (flags tested at run time, fp32 ReLU, separate zeroing) took ~350.
template <int TN, int WAVES, bool PRE, bool FAST>
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

    const int nrb = (M + 31) >> 5;                    // 32-pixel row blocks
    const int stride = gridDim.x * WAVES;
    const int CH = K >> 5;                            // 32-channel chunks per row block
    const int ipitch = int(a.in.sw);
    constexpr unsigned OOB = 0x80000000u;

    // load stream (rb_l, c_l) runs D chunks ahead of the compute stream (rb_c, c_c)
    int rb_l = blockIdx.x * WAVES + wave, c_l = 0;
    int rb_c = rb_l, c_c = 0;
    // Ring of register chunks with STATIC slots (the chunk loop below is unrolled by D, so the compiler's waitcnt pass sees the
    // loads in issue order and waits with exact counts, vmcnt(2(D-1)), instead of draining the ring).
    u32x4 ring[D][2];
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    auto issue = [&](int slot) {
        const int m = rb_l * 32 + r;
        const unsigned off = (rb_l < nrb && m < M && WS_ABLATE != 5 && WS_ABLATE != 8) ? unsigned(m * ipitch + c_l * 32 + hh * 8) * 2u : OOB;
        ring[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        ring[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + 32u, 0, 0);
        if (++c_l == CH) { c_l = 0; rb_l += stride; }
    };
    // the first D chunks are requested before the weight preamble: their HBM latency overlaps it
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);

    // ---- preamble: weight slice (and BN scale/shift) -> LDS, once per workgroup ----
    {
        const _Float16* const w = static_cast<const _Float16*>(a.w16);
        const int k8 = K >> 3;
        // U loads in flight per thread: a one-at-a-time loop would pay the L2 round trip BN*K/(8*NT) times in a row
        constexpr int U = 8;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w), 0, Cout * K * 2, 0x00020000);
        auto stage_trip = [&](int idx0) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = idx0 + u * NT;
                const int row = idx / k8, ck = idx - row * k8;
                const unsigned off = (idx < BN * k8 && n0 + row < Cout) ? unsigned((n0 + row) * K + ck * 8) * 2u : 0x80000000u;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = idx0 + u * NT;
                const int row = idx / k8, ck = idx - row * k8;
                if (idx < BN * k8) *reinterpret_cast<u32x4*>(sB + row * P + ck * 8) = v[u];
            }
        };
        // The first trip is peeled: behind a loop header the compiler drains every load in flight (s_waitcnt vmcnt(0)) -- the ring requested above
        // would complete before the first weight load is even issued, and the two HBM / L2 latencies would add up instead of overlapping.
        const int wtotal = WS_ABLATE == 1 ? 0 : BN * k8;
        if (tid < wtotal) stage_trip(tid);
        for (int idx0 = tid + U * NT; idx0 < wtotal; idx0 += U * NT) stage_trip(idx0);
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

    f32x16 acc[TN];
    auto start_at_bias = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + j * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][4 * g + q] = bq[q];
            }
    };
    start_at_bias();

    // weight fragments are read one (kk, j) step ahead of the MFMA that consumes them (LDS latency behind the previous MFMA)
    auto compute = [&](const u32x4 c0, const u32x4 c1) {
        const int cbase = c_c * 32 + hh * 8;
        const _Float16* const Bp = sB + r * P + cbase;
        h8 bfr[2];
        bfr[0] = *reinterpret_cast<const h8*>(Bp);
        h8 av[2] = {__builtin_bit_cast(h8, c0), __builtin_bit_cast(h8, c1)};
        if constexpr (PRE && WS_ABLATE != 2) {
            h8 s[2], t[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                s[kk] = *reinterpret_cast<const h8*>(sS + cbase + kk * 16);
                t[kk] = *reinterpret_cast<const h8*>(sT + cbase + kk * 16);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                av[kk] = av[kk] * s[kk] + t[kk];
                if (a.pre_relu) av[kk] = __builtin_elementwise_max(av[kk], h8{});
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, (PRE && WS_ABLATE != 2) ? 5 : 1, 0);      // the reads above go first
#pragma unroll
        for (int st = 0; st < 2 * TN; ++st) {
            const int kk = st / TN, j = st % TN;
            if (st + 1 < 2 * TN) {
                const int k1 = (st + 1) / TN, j1 = (st + 1) % TN;
                bfr[(st + 1) & 1] = *reinterpret_cast<const h8*>(Bp + j1 * 32 * P + k1 * 16);
            }
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bfr[st & 1], av[kk], acc[j], 0, 0, 0);
            // pin the order "next fragment read, then this step's MFMA": left alone the scheduler hoists every fragment read of the chunk to its
            // top (142 registers for the FAST variants instead of 132, 256+ and spills for the general ones)
            if (st + 1 < 2 * TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    };
    // Stores go through a buffer descriptor with 32-bit offsets: rows past M and channels past Cout get an out-of-range
    // offset and are dropped by the hardware - no 64-bit address math, no divergent store branches.
    const bool store_half = a.out.f16 != 0;
    const int esz = store_half ? 2 : 4;
    const int opitch = int(a.out.sw);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * opitch + Cout) * esz), 0x00020000);
    const bool has_res = a.res.p != nullptr;        // residual Add fused into the epilogue (same element type as the output)
    const int rpitch = int(a.res.sw);
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc(has_res ? a.res.p : a.out.p, 0, has_res ? int((int64_t(M - 1) * rpitch + Cout) * esz) : 0, 0x00020000);
    const h2 relu_floor = a.relu ? h2{_Float16(0.f), _Float16(0.f)} : h2{-__builtin_inff16(), -__builtin_inff16()};       // FAST: ReLU = packed max with this
    auto epilogue = [&]() {
        const int m = rb_c * 32 + r;
        const unsigned rowoff = (m < M && WS_ABLATE != 3 && WS_ABLATE != 8) ? unsigned(m * opitch * esz) : OOB;
        if constexpr (FAST) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    h2 p[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        p[t] = __builtin_elementwise_max(h2{_Float16(acc[j][8 * gp + 2 * t]), _Float16(acc[j][8 * gp + 2 * t + 1])}, relu_floor);
                    // quads (g, hh) -> after the half-wave exchange lane (r, hh) holds channels 8*(2gp+hh) .. +7 of pixel r
                    const auto s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, p[0]), __builtin_bit_cast(unsigned, p[2]), false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, p[1]), __builtin_bit_cast(unsigned, p[3]), false, false);
                    const int n = n0 + j * 32 + 8 * (2 * gp + hh);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rs_out, n < Cout ? rowoff + unsigned(n * 2) : OOB, 0, 0);
                }
            }
            start_at_bias();
            return;
        }
        const unsigned rrow = (has_res && m < M) ? unsigned(m * rpitch * esz) : OOB;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = acc[j][e];
            if (has_res) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + j * 32 + 8 * g + 4 * hh;
                    const unsigned off = n < Cout ? rrow + unsigned(n * esz) : OOB;
                    if (store_half) {
                        typedef unsigned u32x2r __attribute__((ext_vector_type(2)));
                        const h4 rq = __builtin_bit_cast(h4, __builtin_bit_cast(u32x2r, __builtin_amdgcn_raw_buffer_load_b64(rs_res, off, 0, 0)));
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[4 * g + q] += float(rq[q]);
                    } else {
                        const f32x4 rq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0));
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[4 * g + q] += rq[q];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (a.relu) v[e] = fmaxf(v[e], 0.f);
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
        start_at_bias();
    };

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
constexpr WsTile kWsTiles[6] = {{4, 8}, {4, 4}, {2, 8}, {2, 4}, {1, 8}, {1, 4}};

static size_t ws_lds_bytes(int tn, int K) { return size_t(32 * tn * (K + 8) + 2 * K) * sizeof(_Float16) + size_t(32 * tn) * sizeof(float); }

bool ConvWsEligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvWs16Tiles) return false;
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
    if (a.res.p != nullptr) {                          // fused residual: same element type and pixel-major layout as the output
        if (a.res.f16 != a.out.f16 || a.res.sc != 1 || (a.res.sw % 4) || (reinterpret_cast<uintptr_t>(a.res.p) % 16)) return false;
        if (a.res.sh != a.res.w * a.res.sw || a.res.sn != a.res.h * a.res.sh || M * a.res.sw * 4 >= (int64_t(1) << 31)) return false;
    }
    const WsTile t = kWsTiles[tile % 6];
    if (ws_lds_bytes(t.tn, a.in.c) > size_t(160) * 1024) return false;
    if (t.tn > 1 && a.out.c <= 32 * (t.tn / 2)) return false;                               // do not waste MFMA rows on padding
    return true;
}

template <int TN, int WAVES, bool PRE, bool FAST>
static hipError_t launch_ws_t(const ConvArgs& a, int grid_variant, hipStream_t stream) {      // 0: LDS / wave-slot heuristic, 1: one row block per wave, 2: one workgroup per CU
    const bool one_per_wave = grid_variant == 1;
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
    // (LDS and wave slots only: sizing the grid by the true residency -- registers included, ResidentPerCu() -- measured SLOWER: a few workgroups more
    //  than fit at once cost less than a fifth row block for every wave, e.g. K = 128 at batch 128: 39 us vs 55 us)
    int per_cu = int((size_t(160) * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2048 / (64 * WAVES) ? 2048 / (64 * WAVES) : per_cu);
    if (per_cu > 4) per_cu = 4;
    if (grid_variant == 2) per_cu = 1;
#ifdef WS_PER_CU
    per_cu = WS_PER_CU;                      // (probe builds)
#endif
    const int gy = (a.out.c + 32 * TN - 1) / (32 * TN);
    int slots = cus * per_cu / gy;           // the resident workgroups are shared by the gy N-tiles
    if (slots < 8) slots = 8;
    const int iters = one_per_wave ? 1 : (nrb + slots * WAVES - 1) / (slots * WAVES);
    int gx = (nrb + iters * WAVES - 1) / (iters * WAVES);
    gx = (gx + 7) & ~7;                      // same x -> same XCD for the N-tiles of one row range
    conv1x1_ws_f16_kernel<TN, WAVES, PRE, FAST><<<dim3(gx, gy), dim3(64 * WAVES), lds, stream>>>(a);
    return hipGetLastError();
}

hipError_t LaunchConvWs1x1F16(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvWsEligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = 2 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + int64_t(a.in.c - 1) + 1);
    const bool fast = a.out.f16 && a.res.p == nullptr;
#define IE_WS(T, TN, W)                                                                                                                                  \
    case T:                                                                                                                                              \
        return a.pre_scale ? (fast ? launch_ws_t<TN, W, true, true>(a, tile / 6, stream) : launch_ws_t<TN, W, true, false>(a, tile / 6, stream))       \
                           : (fast ? launch_ws_t<TN, W, false, true>(a, tile / 6, stream) : launch_ws_t<TN, W, false, false>(a, tile / 6, stream));
    switch (tile % 6) {
        IE_WS(0, 4, 8) IE_WS(1, 4, 4) IE_WS(2, 2, 8) IE_WS(3, 2, 4) IE_WS(4, 1, 8) IE_WS(5, 1, 4)
        default: return hipErrorInvalidValue;
    }
#undef IE_WS
}

hipError_t InitKernelsWs() {
    hipError_t e;
#define IE_WSI(TN, W)                                                                                                                                                                      \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f16_kernel<TN, W, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;   \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f16_kernel<TN, W, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;  \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f16_kernel<TN, W, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;  \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f16_kernel<TN, W, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    IE_WSI(4, 8) IE_WSI(4, 4) IE_WSI(2, 8) IE_WSI(2, 4) IE_WSI(1, 8) IE_WSI(1, 4)
#undef IE_WSI
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------------------------------
// Weights-stationary 3x3 / stride 1 / pad 1 convolution for the fp16 mode (DenseNet's growth convs: Cin = 128, Cout = 32; wider
// layers such as ResNet's 64->64 and 128->128 run one 32-channel N-tile per blockIdx.y, each with its own resident weights).
//
// Same 1-D raster of the zero-padded image stack as conv3x3_raster_kernel (kernels.hip): raster index
// p = (b*RH + y)*PW + x with PW = W + 1, RH = H + 1, and tap (ky, kx) of output position p reads raster position
// p + (ky-1)*PW + (kx-1), so the nine taps are nine constant row shifts of ONE LDS window.  What changes for fp16:
//   * the MFMA is 16x faster, so re-staging the 9 x 32 x Cin weights per output tile (more bytes than the tile's own
//     activations) would dominate: the workgroups are persistent and keep ALL weights of the layer in LDS (83 KB for
//     Cin = 128), loaded once;
//   * activations pass through LDS once per 64-channel slice: global -> registers (prefetched while the previous step is
//     on the matrix cores) -> one window of PR = BMp + 2*PW + 2 rows x 144 B; nine shifted fragment reads per row;
//   * D = W x A^T: a lane owns one raster position and quads of channels, v_permlane32_swap pairs the half-waves so every
//     lane stores 16 bytes; pad positions and positions past the raster get an out-of-range buffer offset.
// Raster positions are decoded to pixels with exact multiply-shift division (host-computed magic numbers).
// ------------------------------------------------------------------------------------------------------------------------
struct Ws3Geom {
    int PW, RH, PR, num_tiles, nslices;
    int sh_img, sh_pw;
    unsigned long long m_img, m_pw;
};

template <int WAVES, int TMW, int PIT>
__global__ __launch_bounds__(64 * WAVES) void conv3x3_ws_f16_kernel(const ConvArgs a, const Ws3Geom g) {
    constexpr int NT = 64 * WAVES, BMp = 32 * TMW * WAVES, LDP = 72, RPP = NT / 8;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_ws[];
    const int NS = g.nslices, PW = g.PW, PR = g.PR;
    _Float16* const sW = smem_ws;                                 // [9][NS][32][LDP]
    _Float16* const sP = sW + 9 * NS * 32 * LDP;                  // [PR][LDP]
    float* const sBias = reinterpret_cast<float*>(sP + PR * LDP); // [32]

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.in.c, H = a.in.h, W = a.in.w, Cout = a.out.c;
    const int n0 = blockIdx.y * 32;                   // output-channel tile (wider layers: one resident weight set per N-tile)
    const int img = g.RH * PW, Mr = a.in.n * img;
    const int isw = int(a.in.sw), opitch = int(a.out.sw);

    // raster position -> pixel index (b*H + y)*W + x, or -1 for pad rows / columns and positions outside the raster
    auto pix_of = [&](int j) -> int {
        if (j < 0 || j >= Mr) return -1;
        const int b = int((static_cast<unsigned long long>(unsigned(j)) * g.m_img) >> g.sh_img);
        const int rem = j - b * img;
        const int y = int((static_cast<unsigned long long>(unsigned(rem)) * g.m_pw) >> g.sh_pw);
        const int x = rem - y * PW;
        return (y < H && x < W) ? (b * H + y) * W + x : -1;
    };

    // ---- preamble: every weight of the layer -> LDS (zero-filled past Cin / Cout), bias -> LDS ----
    {
        const _Float16* const w = static_cast<const _Float16*>(a.w16);
        const int items = 9 * NS * 32 * 8;
        constexpr int U = 8;                             // loads in flight per thread
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w), 0, Cout * 9 * Cin * 2, 0x00020000);
        for (int q0 = tid; q0 < items; q0 += U * NT) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                const int ck = q & 7, row = q >> 3;      // row = (tap*NS + slice)*32 + n
                const int n = row & 31, ts = row >> 5;
                const int tap = ts / NS, sl = ts - tap * NS;
                const int c = sl * 64 + ck * 8;
                const unsigned off = (q < items && n0 + n < Cout && c < Cin) ? unsigned(((n0 + n) * 9 + tap) * Cin + c) * 2u : 0x80000000u;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                if (q < items) *reinterpret_cast<u32x4*>(sW + (q >> 3) * LDP + (q & 7) * 8) = v[u];
            }
        }
        for (int q = tid; q < 32; q += NT) sBias[q] = (a.bias != nullptr && n0 + q < Cout) ? a.bias[n0 + q] : 0.f;
    }

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(a.out.n) * a.out.h * a.out.w - 1) * opitch * 2 + Cout * 2), 0x00020000);

    f32x16 acc[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    const int c8 = (tid & 7) * 8;
    int poff[PIT];                                     // element offset of (pixel, c8) of this thread's window rows, or -1
    u32x4 pv[PIT];
    auto decode_rows = [&](int tile) {
        const int jbase = tile * BMp - PW - 1;
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int l = (tid >> 3) + i * RPP;
            const int pix = l < PR ? pix_of(jbase + l) : -1;
            poff[i] = pix >= 0 ? pix * isw + c8 : -1;
        }
    };
    auto issue = [&](int sl) {
        const int c0 = sl * 64;
        const bool cok = c0 + c8 < Cin;
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const unsigned off = (poff[i] >= 0 && cok) ? unsigned(poff[i] + c0) * 2u : OOB;
            pv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int l = (tid >> 3) + i * RPP;
            if (l < PR) *reinterpret_cast<u32x4*>(sP + l * LDP + c8) = pv[i];
        }
    };
    // nine shifted GEMMs out of LDS: 36 (tap, kk) steps, fragment reads one step ahead of the MFMAs
    auto compute_slice = [&](int sl) {
        const _Float16* const Abase = sP + (wave * 32 * TMW + r) * LDP + hh * 8;
        const _Float16* const Bbase = sW + (sl * 32 + r) * LDP + hh * 8;
        constexpr int STEPS = 36;
        h8 af[2][TMW], bf[2];
        auto read_step = [&](int st, int slot) {
            const int tap = st >> 2, kk = st & 3;
            const int shift = (tap / 3) * PW + (tap % 3);
            const _Float16* const A = Abase + shift * LDP + kk * 16;
#pragma unroll
            for (int i = 0; i < TMW; ++i) af[slot][i] = *reinterpret_cast<const h8*>(A + i * 32 * LDP);
            bf[slot] = *reinterpret_cast<const h8*>(Bbase + tap * NS * 32 * LDP + kk * 16);
        };
        read_step(0, 0);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int cur = st & 1;
            if (st + 1 < STEPS) read_step(st + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < TMW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[cur], af[cur][i], acc[i], 0, 0, 0);
        }
    };
    auto epilogue = [&](int tile) {
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int pix = pix_of(tile * BMp + (wave * TMW + i) * 32 + r);
            const unsigned rowoff = pix >= 0 ? unsigned(pix * opitch * 2) : OOB;
            float v[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + 8 * gq + 4 * hh);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = acc[i][4 * gq + q] + bq[q];
                    v[4 * gq + q] = a.relu ? fmaxf(x, 0.f) : x;
                    acc[i][4 * gq + q] = 0.f;
                }
            }
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const h4 qa = {_Float16(v[8 * gp + 0]), _Float16(v[8 * gp + 1]), _Float16(v[8 * gp + 2]), _Float16(v[8 * gp + 3])};
                const h4 qb = {_Float16(v[8 * gp + 4]), _Float16(v[8 * gp + 5]), _Float16(v[8 * gp + 6]), _Float16(v[8 * gp + 7])};
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 xa = __builtin_bit_cast(u32x2, qa), xb = __builtin_bit_cast(u32x2, qb);
                const auto s0 = __builtin_amdgcn_permlane32_swap(xa[0], xb[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(xa[1], xb[1], false, false);
                const int n = n0 + 8 * (2 * gp + hh);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rs_out, n < Cout ? rowoff + unsigned(n * 2) : OOB, 0, 0);
            }
        }
    };

    int tile = blockIdx.x, sl = 0;
    if (tile < g.num_tiles) {
        decode_rows(tile);
        issue(0);
    }
    while (tile < g.num_tiles) {
        __syncthreads();                 // every wave is done reading the previous window (first pass: nothing to wait for)
        commit();
        __syncthreads();                 // window (and, the first time, the weights) visible to every wave
        // next step of this workgroup: next slice of the tile, or slice 0 of its next tile
        int ntile = tile, nsl = sl + 1;
        if (nsl == NS) { nsl = 0; ntile = tile + gridDim.x; }
        if (ntile < g.num_tiles) {
            if (ntile != tile) decode_rows(ntile);
            issue(nsl);
        }
        __builtin_amdgcn_sched_barrier(0);
        compute_slice(sl);
        __builtin_amdgcn_sched_barrier(0);
        if (sl == NS - 1) epilogue(tile);
        tile = ntile;
        sl = nsl;
    }
}

struct Ws3Tile { int waves, tmw, pit; };
constexpr Ws3Tile kWs3Tiles[kNumConvWs3Tiles] = {{4, 2, 12}, {4, 1, 8}, {8, 1, 6}, {2, 1, 12}, {12, 1, 6}};      // {12, 1, 6}: three waves per SIMD beside the 83 KB of resident weights (384-pixel tiles: the window still fits)

static size_t ws3_lds_bytes(int tile, int Cin, int PW) {
    const Ws3Tile t = kWs3Tiles[tile];
    const int NS = (Cin + 63) / 64, PR = 32 * t.tmw * t.waves + 2 * PW + 2;
    return size_t(9 * NS * 32 + PR) * 72 * sizeof(_Float16) + 32 * sizeof(float);
}

bool ConvWs3Eligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvWs3Tiles) return false;
    if (!a.in.f16 || !a.out.f16 || a.w16 == nullptr || a.pre_scale != nullptr) return false;
    if (a.kh != 3 || a.kw != 3 || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1 || a.out.h != a.in.h || a.out.w != a.in.w) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || (a.in.c & 7) || (a.in.sw & 7) || (reinterpret_cast<uintptr_t>(a.in.p) & 15) ||
        (reinterpret_cast<uintptr_t>(a.w16) & 15))
        return false;
    if ((a.out.c & 7) || (a.out.sw & 7) || (reinterpret_cast<uintptr_t>(a.out.p) & 15)) return false;      // any Cout, 32 per N-tile
    if (a.in.sh != a.in.sw * a.in.w || a.in.sn != a.in.sh * a.in.h) return false;          // pixel-major NHWC views
    if (a.out.sh != a.out.sw * a.out.w || a.out.sn != a.out.sh * a.out.h) return false;
    const int64_t Mr = int64_t(a.in.n) * (a.in.h + 1) * (a.in.w + 1), Mpix = int64_t(a.in.n) * a.in.h * a.in.w;
    if (Mr + 4096 >= (int64_t(1) << 31) || Mpix * a.in.sw * 2 >= (int64_t(1) << 31) || Mpix * a.out.sw * 2 >= (int64_t(1) << 31)) return false;
    const Ws3Tile t = kWs3Tiles[tile];
    const int PR = 32 * t.tmw * t.waves + 2 * (a.in.w + 1) + 2;
    if (PR > t.pit * (64 * t.waves / 8)) return false;                                       // register prefetch covers the window
    return ws3_lds_bytes(tile, a.in.c, a.in.w + 1) <= size_t(160) * 1024;
}

static void magic_div(unsigned d, unsigned long long* m, int* sh) {      // floor(j / d) = (j * m) >> sh for 0 <= j < 2^31
    int L = 0;
    while ((1ull << L) < d) ++L;
    *sh = 31 + L;
    *m = ((1ull << (31 + L)) / d) + 1;
}

template <int T>
static hipError_t launch_ws3_t(const ConvArgs& a, hipStream_t stream) {
    constexpr Ws3Tile t = kWs3Tiles[T];
    constexpr int BMp = 32 * t.tmw * t.waves;
    Ws3Geom g;
    g.PW = a.in.w + 1;
    g.RH = a.in.h + 1;
    g.PR = BMp + 2 * g.PW + 2;
    g.nslices = (a.in.c + 63) / 64;
    const int64_t Mr = int64_t(a.in.n) * g.RH * g.PW;
    g.num_tiles = int((Mr + BMp - 1) / BMp);
    magic_div(unsigned(g.RH * g.PW), &g.m_img, &g.sh_img);
    magic_div(unsigned(g.PW), &g.m_pw, &g.sh_pw);
    const size_t lds = ws3_lds_bytes(T, a.in.c, g.PW);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount;
    }
    int per_cu = int((size_t(160) * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    const int gy = (a.out.c + 31) / 32;
    int slots = cus * per_cu / gy;                     // resident workgroups per N-tile
    if (slots < 1) slots = 1;
    const int iters = (g.num_tiles + slots - 1) / slots;
    const int gx = (g.num_tiles + iters - 1) / iters;
    conv3x3_ws_f16_kernel<t.waves, t.tmw, t.pit><<<dim3(gx, gy), dim3(64 * t.waves), lds, stream>>>(a, g);
    return hipGetLastError();
}

hipError_t LaunchConvWs3x3F16(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvWs3Eligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = 2 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + int64_t(a.in.c - 1) + 1);
    switch (tile) {
        case 0: return launch_ws3_t<0>(a, stream);
        case 1: return launch_ws3_t<1>(a, stream);
        case 2: return launch_ws3_t<2>(a, stream);
        case 3: return launch_ws3_t<3>(a, stream);
        case 4: return launch_ws3_t<4>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t InitKernelsWs3() {
    hipError_t e;
#define IE_WS3(T)                                                                                                             \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ws_f16_kernel<kWs3Tiles[T].waves, kWs3Tiles[T].tmw, kWs3Tiles[T].pit>), \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    IE_WS3(0) IE_WS3(1) IE_WS3(2) IE_WS3(3) IE_WS3(4)
#undef IE_WS3
    return hipSuccess;
}

}  // namespace ie
