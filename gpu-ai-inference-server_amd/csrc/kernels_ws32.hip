// Weights-stationary 1x1 convolution, fp32 (the headline precision): the fp32 twin of conv1x1_ws_f16_kernel (kernels_ws.hip).
//
// The tiled implicit GEMM reaches ~37 % of the fp32 MFMA peak on DenseNet's big 1x1 layers: every K-tile costs two barriers and
// a trip of the activation tile through LDS, and that time ADDS to the MFMA time.  For a 1x1 / stride 1 conv
// over NHWC the A operand needs no LDS at all: 16 contiguous bytes of a pixel row are the k = 4hh+e (e = 0..3) operands of four
// consecutive v_mfma_f32_32x32x2_f32, so every lane loads its own fragments straight from HBM into a register ring that runs
// several chunks ahead, and the weight slice [BN][K] (+ folded-BN scale/shift) sits in LDS for the whole life of the persistent
// workgroup.  After the preamble there is no barrier; each wave streams 32-pixel row blocks on its own, one ds_read_b128 per
// four MFMAs.  D = W x A^T: a lane owns one pixel and quads of channels -> 16-byte stores.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "env.h"
#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int TN, int WAVES, bool PRE>
__global__ __launch_bounds__(64 * WAVES) void conv1x1_ws_f32_kernel(const ConvArgs a) {
    // ring depth: a chunk (2 KiB per wave) feeds 8*TN MFMAs.  Deeper rings did not help the narrow tiles (measured: 12 slots
    // slower than 4-6): with TN = 1 the row-strided 16-byte fragment loads keep the CU's texture addresser as busy as the MFMAs.
    constexpr int NT = 64 * WAVES, BN = 32 * TN, D = TN >= 4 ? 4 : 6;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem_ws32[];
    const int K = a.in.c, P = K + 4;                   // weight row pitch: (K+4)/4 odd for K % 8 == 0 -> conflict-free b128 reads
    float* const sB = smem_ws32;                       // [BN][P]
    float* const sS = sB + BN * P;                     // [K] prologue scale
    float* const sT = sS + K;                          // [K] prologue shift
    float* const sBias = sT + K;                       // [BN]
    const int Cout = a.out.c;
    const int M = a.out.n * a.out.h * a.out.w;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nrb = (M + 31) >> 5;                     // 32-pixel row blocks
    const int stride = gridDim.x * WAVES;
    const int CH = K >> 4;                             // 16-channel chunks per row block (two 16-byte loads per lane)
    const int ipitch = int(a.in.sw), opitch = int(a.out.sw);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * opitch + Cout) * 4), 0x00020000);

    // ring of register chunks with static slots (loop unrolled by D): the waitcnt pass sees the loads in issue order
    int rb_l = blockIdx.x * WAVES + wave, c_l = 0;
    int rb_c = rb_l, c_c = 0;
    u32x4 ring[D][2];
    auto issue = [&](int slot) {
        const int m = rb_l * 32 + r;
        const unsigned off = (rb_l < nrb && m < M) ? unsigned(m * ipitch + c_l * 16 + hh * 4) * 4u : OOB;
        ring[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        ring[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + 32u, 0, 0);
        if (++c_l == CH) { c_l = 0; rb_l += stride; }
    };
    // the first D chunks are requested before the weight preamble: their HBM latency overlaps it
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);

    // ---- preamble: weight slice, BN scale/shift, bias -> LDS, once per workgroup (8 loads in flight per thread) ----
    {
        constexpr int U = 8;
        const int k4 = K >> 2;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, Cout * K * 4, 0x00020000);
        for (int idx0 = tid; idx0 < ((a.debug & 128) ? 0 : BN * k4); idx0 += U * NT) {      // debug 128: timing-only, no weight preamble
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = idx0 + u * NT;
                const int row = idx / k4, ck = idx - row * k4;
                const unsigned off = (idx < BN * k4 && n0 + row < Cout) ? unsigned((n0 + row) * K + ck * 4) * 4u : OOB;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = idx0 + u * NT;
                const int row = idx / k4, ck = idx - row * k4;
                if (idx < BN * k4) *reinterpret_cast<u32x4*>(sB + row * P + ck * 4) = v[u];
            }
        }
        for (int idx = tid; idx < BN; idx += NT) sBias[idx] = (a.bias != nullptr && n0 + idx < Cout) ? a.bias[n0 + idx] : 0.f;
        if constexpr (PRE) {
            for (int idx = tid; idx < k4; idx += NT) {
                *reinterpret_cast<f32x4*>(sS + idx * 4) = *reinterpret_cast<const f32x4*>(a.pre_scale + idx * 4);
                *reinterpret_cast<f32x4*>(sT + idx * 4) = *reinterpret_cast<const f32x4*>(a.pre_shift + idx * 4);
            }
        }
    }
    __syncthreads();

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // weight fragments are read one (q, j) step ahead of the four MFMAs that consume them, so the LDS latency hides behind the
    // previous step's 256 MFMA cycles instead of stalling every step
    auto compute = [&](const u32x4 c0, const u32x4 c1) {
        const int cbase = c_c * 16 + hh * 4;
        const float* const Bp = sB + r * P + cbase;
        f32x4 bfr[2];
        bfr[0] = *reinterpret_cast<const f32x4*>(Bp);
        f32x4 av[2] = {__builtin_bit_cast(f32x4, c0), __builtin_bit_cast(f32x4, c1)};
        if constexpr (PRE) {
            f32x4 s[2], t[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                s[q] = *reinterpret_cast<const f32x4*>(sS + cbase + q * 8);
                t[q] = *reinterpret_cast<const f32x4*>(sT + cbase + q * 8);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = av[q][e] * s[q][e] + t[q][e];
                    av[q][e] = a.pre_relu ? fmaxf(x, 0.f) : x;
                }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, PRE ? 5 : 1, 0);      // the reads above go first
#pragma unroll
        for (int st = 0; st < 2 * TN; ++st) {
            const int q = st / TN, j = st % TN;
            if (st + 1 < 2 * TN) {
                const int q1 = (st + 1) / TN, j1 = (st + 1) % TN;
                bfr[(st + 1) & 1] = *reinterpret_cast<const f32x4*>(Bp + j1 * 32 * P + q1 * 8);
            }
            if (!(a.debug & 64)) {                          // debug 64: timing-only, no MFMAs
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[st & 1][e], av[q][e], acc[j], 0, 0, 0);
            }
            // pin the order "next fragment read, then this step's four MFMAs" (the scheduler otherwise sinks the read to its use)
            if (st + 1 < 2 * TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };
    const bool has_res = a.res.p != nullptr;        // residual Add fused into the epilogue (ResNet shortcuts)
    const int rpitch = int(a.res.sw);
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc(has_res ? a.res.p : a.out.p, 0, has_res ? int((int64_t(M - 1) * rpitch + Cout) * 4) : 0, 0x00020000);
    auto epilogue = [&]() {
        const int m = rb_c * 32 + r;
        const unsigned rowoff = m < M ? unsigned(m * opitch * 4) : OOB;
        const unsigned rrow = (has_res && m < M) ? unsigned(m * rpitch * 4) : OOB;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = j * 32 + 8 * g + 4 * hh;
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + nl);
                f32x4 rq = {0.f, 0.f, 0.f, 0.f};
                if (has_res)
                    rq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, n0 + nl < Cout ? rrow + unsigned((n0 + nl) * 4) : OOB, 0, 0));
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = acc[j][4 * g + q] + bq[q] + rq[q];
                    v[q] = a.relu ? fmaxf(x, 0.f) : x;
                    acc[j][4 * g + q] = 0.f;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out,
                                                       (n0 + nl < Cout && !(a.debug & 32)) ? rowoff + unsigned((n0 + nl) * 4) : OOB, 0, 0);   // debug 32: no stores
            }
        }
    };

    // Row blocks end wherever the chunk count says, so the epilogue is inlined behind a wave-uniform branch at each of the D
    // positions; past the last row block the stream runs on zeros (every load / store of a row >= M is out of range).
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

// ------------------------------------------------------------------------------------------------------------------------
// K-split variant for mid-size pixel counts with long K (DenseNet block 3 at batch 32: M = 6272, K = 256..992, N = 128).
// There are too few 32-pixel row blocks to give every wave its own (196 x 4 N-tiles = 784 tasks for 2048 wave slots), and a
// wave that walks the whole K alone needs K/2 x 64 cycles.  Here the WAVES waves of a workgroup work on the SAME row block and
// each streams only its 1/WAVES slice of K through its own register ring; the weight slice [32][K] is still loaded once per
// persistent workgroup.  At the end of a row block the partial tiles are summed pairwise through one LDS tile buffer (log2(WAVES)
// steps, fixed order -> bitwise reproducible) and wave 0 applies bias / residual / ReLU and stores; the other waves are already
// streaming the next row block (their ring never drained).
// ------------------------------------------------------------------------------------------------------------------------
template <int WAVES, bool PRE>
__global__ __launch_bounds__(64 * WAVES) void conv1x1_wsk_f32_kernel(const ConvArgs a) {
    constexpr int NT = 64 * WAVES, BN = 32, D = 6, RP = 36;       // D: ring depth (see above); RP: row pitch of the reduction tiles (floats)
    constexpr unsigned OOB = 0x80000000u;
    static_assert((WAVES & (WAVES - 1)) == 0, "pairwise reduction needs a power of two");
    extern __shared__ __attribute__((aligned(16))) float smem_ws32[];
    const int K = a.in.c, P = K + 4;
    float* const sB = smem_ws32;                       // [32][P]
    float* const sS = sB + BN * P;                     // [K]
    float* const sT = sS + K;                          // [K]
    float* const sBias = sT + K;                       // [32]
    float* const sRed = sBias + BN;                    // [WAVES/2][32][RP]
    const int Cout = a.out.c;
    const int M = a.out.n * a.out.h * a.out.w;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nrb = (M + 31) >> 5;
    const int CHt = K >> 4;                            // 16-channel chunks of the whole K
    const int cb = CHt * wave / WAVES, ce = CHt * (wave + 1) / WAVES;      // this wave's slice (>= 1 chunk: eligibility)
    const int ipitch = int(a.in.sw), opitch = int(a.out.sw);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(M - 1) * opitch + Cout) * 4), 0x00020000);

    int rb_l = blockIdx.x, c_l = cb;
    int rb_c = rb_l, c_c = cb;
    u32x4 ring[D][2];
    auto issue = [&](int slot) {
        const int m = rb_l * 32 + r;
        const unsigned off = (rb_l < nrb && m < M) ? unsigned(m * ipitch + c_l * 16 + hh * 4) * 4u : OOB;
        ring[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        ring[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off + 32u, 0, 0);
        if (++c_l == ce) { c_l = cb; rb_l += gridDim.x; }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);

    {   // preamble: weight slice, BN scale/shift, bias -> LDS (8 loads in flight per thread)
        constexpr int U = 8;
        const int k4 = K >> 2;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, Cout * K * 4, 0x00020000);
        for (int idx0 = tid; idx0 < BN * k4; idx0 += U * NT) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = idx0 + u * NT;
                const int row = idx / k4, ck = idx - row * k4;
                const unsigned off = (idx < BN * k4 && n0 + row < Cout) ? unsigned((n0 + row) * K + ck * 4) * 4u : OOB;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = idx0 + u * NT;
                const int row = idx / k4, ck = idx - row * k4;
                if (idx < BN * k4) *reinterpret_cast<u32x4*>(sB + row * P + ck * 4) = v[u];
            }
        }
        for (int idx = tid; idx < BN; idx += NT) sBias[idx] = (a.bias != nullptr && n0 + idx < Cout) ? a.bias[n0 + idx] : 0.f;
        if constexpr (PRE) {
            for (int idx = tid; idx < k4; idx += NT) {
                *reinterpret_cast<f32x4*>(sS + idx * 4) = *reinterpret_cast<const f32x4*>(a.pre_scale + idx * 4);
                *reinterpret_cast<f32x4*>(sT + idx * 4) = *reinterpret_cast<const f32x4*>(a.pre_shift + idx * 4);
            }
        }
    }
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    auto compute = [&](const u32x4 c0, const u32x4 c1) {
        const int cbase = c_c * 16 + hh * 4;
        const float* const Bp = sB + r * P + cbase;
        f32x4 bfr[2];
        bfr[0] = *reinterpret_cast<const f32x4*>(Bp);
        bfr[1] = *reinterpret_cast<const f32x4*>(Bp + 8);
        f32x4 av[2] = {__builtin_bit_cast(f32x4, c0), __builtin_bit_cast(f32x4, c1)};
        if constexpr (PRE) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(sS + cbase + q * 8);
                const f32x4 t = *reinterpret_cast<const f32x4*>(sT + cbase + q * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = av[q][e] * s[e] + t[e];
                    av[q][e] = a.pre_relu ? fmaxf(x, 0.f) : x;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[q][e], av[q][e], acc, 0, 0, 0);
    };
    const bool has_res = a.res.p != nullptr;
    const int rpitch = int(a.res.sw);
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc(has_res ? a.res.p : a.out.p, 0, has_res ? int((int64_t(M - 1) * rpitch + Cout) * 4) : 0, 0x00020000);
    // lane (r, hh) holds pixel r, channels 8g + 4hh + q: a tile row is one pixel's 32 channels
    auto row_block_done = [&]() {
#pragma unroll
        for (int stride = WAVES / 2; stride >= 1; stride >>= 1) {
            if (wave >= stride && wave < 2 * stride) {
                float* const dst = sRed + ((wave - stride) * 32 + r) * RP + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(dst + 8 * g) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            }
            __syncthreads();
            if (wave < stride) {
                const float* const src = sRed + (wave * 32 + r) * RP + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(src + 8 * g);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[4 * g + q] += x[q];
                }
            }
            __syncthreads();
        }
        if (wave == 0) {
            const int m = rb_c * 32 + r;
            const unsigned rowoff = m < M ? unsigned(m * opitch * 4) : OOB;
            const unsigned rrow = (has_res && m < M) ? unsigned(m * rpitch * 4) : OOB;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = 8 * g + 4 * hh;
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + nl);
                f32x4 rq = {0.f, 0.f, 0.f, 0.f};
                if (has_res)
                    rq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, n0 + nl < Cout ? rrow + unsigned((n0 + nl) * 4) : OOB, 0, 0));
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = acc[4 * g + q] + bq[q] + rq[q];
                    v[q] = a.relu ? fmaxf(x, 0.f) : x;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, n0 + nl < Cout ? rowoff + unsigned((n0 + nl) * 4) : OOB, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    };

    // every wave sees the same number of row blocks, so the barriers inside row_block_done() pair up although the waves reach
    // them from different positions of their unrolled chunk loops
    while (rb_c < nrb) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            if (rb_c < nrb) {
                compute(ring[s][0], ring[s][1]);
                issue(s);
                if (++c_c == ce) {
                    row_block_done();
                    c_c = cb;
                    rb_c += gridDim.x;
                }
            }
        }
    }
}

struct Ws32Tile { int tn, waves; };
constexpr Ws32Tile kWs32Tiles[6] = {{4, 8}, {4, 4}, {2, 8}, {2, 4}, {1, 8}, {1, 4}};

static size_t ws32_lds_bytes(int tn, int K) { return size_t(32 * tn * (K + 4) + 2 * K + 32 * tn) * sizeof(float); }

static size_t wsk32_lds_bytes(int waves, int K) { return size_t(32 * (K + 4) + 2 * K + 32 + (waves / 2) * 32 * 36) * sizeof(float); }

bool ConvWs32Eligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvWs32Tiles) return false;
    if (a.in.f16 || a.out.f16 || a.w == nullptr || a.kh != 1 || a.kw != 1 || a.sh != 1 || a.sw != 1 || a.pt != 0 || a.pl != 0) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || a.in.h != a.out.h || a.in.w != a.out.w) return false;
    if ((a.in.c & 15) || (a.in.sw & 3) || (reinterpret_cast<uintptr_t>(a.in.p) & 15) || (reinterpret_cast<uintptr_t>(a.w) & 15)) return false;
    if (a.in.sh != a.in.w * a.in.sw || a.in.sn != a.in.h * a.in.sh) return false;           // pixels at a constant pitch
    if (a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh) return false;
    if (a.pre_scale && ((reinterpret_cast<uintptr_t>(a.pre_scale) & 15) || (reinterpret_cast<uintptr_t>(a.pre_shift) & 15))) return false;
    const int64_t M = int64_t(a.in.n) * a.in.h * a.in.w;
    if (M * a.in.sw * 4 >= (int64_t(1) << 31) || M * a.out.sw * 4 >= (int64_t(1) << 31)) return false;
    if ((a.out.c % 4) || (a.out.sw % 4) || (reinterpret_cast<uintptr_t>(a.out.p) % 16)) return false;      // 16-byte stores
    if (a.res.p != nullptr) {                          // fused residual: same pixel-major layout, 16-byte loads
        if (a.res.f16 || a.res.sc != 1 || (a.res.sw % 4) || (reinterpret_cast<uintptr_t>(a.res.p) % 16)) return false;
        if (a.res.sh != a.res.w * a.res.sw || a.res.sn != a.res.h * a.res.sh || M * a.res.sw * 4 >= (int64_t(1) << 31)) return false;
    }
    if (tile >= kNumConvWsTiles && tile < kNumConvWsTiles + 2) {      // K-split variants: every wave needs at least one 16-channel chunk
        const int waves = tile == kNumConvWsTiles ? 8 : 4;
        return a.in.c / 16 >= waves && wsk32_lds_bytes(waves, a.in.c) <= size_t(160) * 1024;
    }
    const Ws32Tile t = kWs32Tiles[(tile >= kNumConvWsTiles + 2 ? tile - (kNumConvWsTiles + 2) : tile) % 6];
    if (ws32_lds_bytes(t.tn, a.in.c) > size_t(160) * 1024) return false;
    if (t.tn > 1 && a.out.c <= 32 * (t.tn / 2)) return false;                               // do not waste MFMA rows on padding
    return true;
}

template <int WAVES, bool PRE>
static hipError_t launch_wsk32_t(const ConvArgs& a, hipStream_t stream) {
    const int64_t M = int64_t(a.in.n) * a.in.h * a.in.w;
    const int nrb = int((M + 31) / 32);
    const size_t lds = wsk32_lds_bytes(WAVES, a.in.c);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount;
    }
    const int gy = (a.out.c + 31) / 32;
    int per_cu = int((size_t(160) * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2048 / (64 * WAVES) ? 2048 / (64 * WAVES) : per_cu);
    if (per_cu > 4) per_cu = 4;
    int slots = cus * per_cu / gy;                     // resident workgroups per N-tile
    if (slots < 8) slots = 8;
    const int iters = (nrb + slots - 1) / slots;       // row blocks per workgroup
    int gx = (nrb + iters - 1) / iters;
    gx = (gx + 7) & ~7;
    conv1x1_wsk_f32_kernel<WAVES, PRE><<<dim3(gx, gy), dim3(64 * WAVES), lds, stream>>>(a);
    return hipGetLastError();
}

template <int TN, int WAVES, bool PRE>
static hipError_t launch_ws32_t(const ConvArgs& a, int grid_variant, hipStream_t stream) {      // 0: LDS / wave-slot heuristic, 1: one row block per wave, 2: one workgroup per CU
    const bool one_per_wave = grid_variant == 1;
    const int64_t M = int64_t(a.in.n) * a.in.h * a.in.w;
    const int nrb = int((M + 31) / 32);
    const size_t lds = ws32_lds_bytes(TN, a.in.c);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount;
    }
    const int gy = (a.out.c + 32 * TN - 1) / (32 * TN);
    // resident workgroups (LDS, 2048 threads per CU) shared by the gy N-tiles; then the smallest grid with the same number of
    // row blocks per wave, so that no wave runs a round more than the others have to
    int per_cu = int((size_t(160) * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2048 / (64 * WAVES) ? 2048 / (64 * WAVES) : per_cu);
    if (per_cu > 4) per_cu = 4;
    if (grid_variant == 2) per_cu = 1;
    int slots = cus * per_cu / gy;
    if (slots < 8) slots = 8;
    const int iters = one_per_wave ? 1 : (nrb + slots * WAVES - 1) / (slots * WAVES);
    int gx = (nrb + iters * WAVES - 1) / (iters * WAVES);
    gx = (gx + 7) & ~7;                      // same x -> same XCD for the N-tiles of one row range
    conv1x1_ws_f32_kernel<TN, WAVES, PRE><<<dim3(gx, gy), dim3(64 * WAVES), lds, stream>>>(a);
    return hipGetLastError();
}

hipError_t LaunchConvWs1x1F32(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvWs32Eligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    const int dbg = Knobs().debug_ablate;
    a.debug = dbg;                                     // timing-only ablations (wrong results): 32 no stores, 64 no MFMAs, 128 no weight preamble
    a.in_bytes = 4 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + int64_t(a.in.c - 1) + 1);
    if (tile == kNumConvWsTiles) return a.pre_scale ? launch_wsk32_t<8, true>(a, stream) : launch_wsk32_t<8, false>(a, stream);
    if (tile == kNumConvWsTiles + 1) return a.pre_scale ? launch_wsk32_t<4, true>(a, stream) : launch_wsk32_t<4, false>(a, stream);
    const bool pc1 = tile >= kNumConvWsTiles + 2;                     // 14-19: shapes 0-5, one workgroup per CU
    const int shape = (pc1 ? tile - (kNumConvWsTiles + 2) : tile) % 6, variant = pc1 ? 2 : (tile >= 6 ? 1 : 0);
#define IE_WS(T, TN, W) \
    case T: return a.pre_scale ? launch_ws32_t<TN, W, true>(a, variant, stream) : launch_ws32_t<TN, W, false>(a, variant, stream);
    switch (shape) {
        IE_WS(0, 4, 8) IE_WS(1, 4, 4) IE_WS(2, 2, 8) IE_WS(3, 2, 4) IE_WS(4, 1, 8) IE_WS(5, 1, 4)
        default: return hipErrorInvalidValue;
    }
#undef IE_WS
}

hipError_t InitKernelsWs32() {
    hipError_t e;
#define IE_WSI(TN, W)                                                                                                                       \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f32_kernel<TN, W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024)) != hipSuccess) return e;                                                                       \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f32_kernel<TN, W, false>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024)) != hipSuccess) return e;
    IE_WSI(4, 8) IE_WSI(4, 4) IE_WSI(2, 8) IE_WSI(2, 4) IE_WSI(1, 8) IE_WSI(1, 4)
#undef IE_WSI
#define IE_WSK(W, PRE)                                                                                                                     \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_wsk_f32_kernel<W, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024)) != hipSuccess) return e;
    IE_WSK(8, true) IE_WSK(8, false) IE_WSK(4, true) IE_WSK(4, false)
#undef IE_WSK
    return hipSuccess;
}

}  // namespace ie
