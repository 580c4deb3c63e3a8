// Weights-stationary 1x1 convolution for the fp8 precision mode (BASELINE configs[4], ResNet-50's bottleneck 1x1 convs).
//
// The tiled implicit GEMM (kernels_f8.hip) ran the K = 64 -> 256-channel convs of stage 1 at 1.4-1.9 TB/s: 128-deep K tiles half empty, the
// activation tile through LDS behind two barriers, four N-tile workgroups re-reading it.  Those layers are pure streaming -- 0.25 KB of
// output per 64 B of input -- so this is conv1x1_ws_f16_kernel (kernels_ws.hip) re-typed for e4m3: the weight slice [BN][K] (1 byte per
// element: every 1x1 of ResNet-50 fits) and the per-channel epilogue constants live in LDS for the life of a persistent workgroup, every
// wave streams 32-pixel row blocks on its own (no barrier after the preamble), a lane loads 16 bytes of its pixel row = the operands of
// TWO v_mfma_f32_32x32x16_fp8_fp8 steps (low / high 8 bytes; the weight rows in LDS are cut the same way) through a ring of register chunks.
//
// DUAL: a bottleneck block whose shortcut is a projection computes  relu(conv3(a) + convP(x))  as two GEMMs into two accumulator sets in the
// SAME launch (x read through the projection's stride): the shortcut tensor -- 205 MB at batch 256 in stage 1, written by one launch and
// read back by the next -- never exists.  The two products have different real-unit multipliers (input scale x weight-row scale), so they
// are combined in the epilogue:  acc1 * e1[n] + b1[n] + acc2 * e2[n] + b2[n].
//
// Epilogue as conv_igemm_f8_kernel: D = W x A^T, a lane owns one pixel and quads of channels; scale, bias, e4m3 shortcut (one 16-byte load
// per lane, redistributed by v_permlane32_swap), ReLU, re-quantisation, two more swaps, one 16-byte store of 16 consecutive channels.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef long i64x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr float kE4m3Max8 = 448.0f;
__device__ __forceinline__ unsigned pack4_e4m3_ws(float a, float b, float c, float d) {
    a = __builtin_fminf(__builtin_fmaxf(a, -kE4m3Max8), kE4m3Max8);
    b = __builtin_fminf(__builtin_fmaxf(b, -kE4m3Max8), kE4m3Max8);
    c = __builtin_fminf(__builtin_fmaxf(c, -kE4m3Max8), kE4m3Max8);
    d = __builtin_fminf(__builtin_fmaxf(d, -kE4m3Max8), kE4m3Max8);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return unsigned(p);
}
__device__ __forceinline__ void unpack4_e4m3_ws(unsigned p, float* v) {
    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(int(p), false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(int(p), true);
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
}
constexpr unsigned kOOB8 = 0x80000000u;
constexpr int kRing8 = 4;          // register chunks (32 channels x 32 pixels = 1 KiB) a wave keeps in flight
}  // namespace

struct Ws8Geom {
    int ohw, ow;                   // DUAL with a strided projection: output pixel -> (image, oy, ox)
    unsigned long long m_ohw, m_ow;
    int sh_ohw, sh_ow;
};

// STR: the (single) input is read through the conv's stride (ResNet's strided projection shortcuts): output pixel -> input pixel by the same
// multiply-shift division as DUAL's second input.
template <int TN, int WAVES, bool DUAL, bool STR = false>
__global__ __launch_bounds__(64 * WAVES) void conv1x1_ws_f8_kernel(const ConvArgs a, const Ws8Geom g) {
    static_assert(!(DUAL && STR), "the strided variant has one input");
    constexpr int NT = 64 * WAVES, BN = 32 * TN, D = kRing8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w8[];
    const int K1 = a.in.c, K2 = DUAL ? a.in2.c : 0;
    const int P1 = K1 + 16, P2 = K2 + 16;             // row pitches of the weight images: pitch / 16 odd for K % 32 == 0 (conflict-free ds_read_b128)
    unsigned char* const sB1 = smem_w8;               // [BN][P1]
    unsigned char* const sB2 = sB1 + BN * P1;         // [BN][P2]   (DUAL)
    float* const sE1 = reinterpret_cast<float*>(sB2 + (DUAL ? BN * P2 : 0));   // [BN] epilogue multipliers of GEMM 1
    float* const sBi = sE1 + BN;                      // [BN] bias (both GEMMs' biases summed)
    float* const sE2 = sBi + BN;                      // [BN] (DUAL)
    const int Cout = a.out.c;
    const int M = a.out.n * a.out.h * a.out.w;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nrb = (M + 31) >> 5;
    const int stride = gridDim.x * WAVES;
    const int CH1 = K1 >> 5, CH = CH1 + (K2 >> 5);    // 32-channel chunks per row block: GEMM 1's, then GEMM 2's
    const int ipitch = int(a.in.sw);

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(DUAL ? a.in2.p : a.in.p, 0, DUAL ? int(a.in2_bytes) : 0, 0x00020000);

    // load stream (rb_l, c_l) runs D chunks ahead of the compute stream (rb_c, c_c)
    int rb_l = blockIdx.x * WAVES + wave, c_l = 0;
    int rb_c = rb_l, c_c = 0;
    unsigned off1_l = 0, off2_l = 0;                  // byte offsets of the lane's pixel row in the two inputs for row block rb_l
    auto row_offsets = [&](int rb) {
        const int m = rb * 32 + r;
        const bool ok = rb < nrb && m < M;
        if constexpr (STR) {
            const unsigned um = ok ? unsigned(m) : 0u;
            const int b = int((static_cast<unsigned long long>(um) * g.m_ohw) >> g.sh_ohw);
            const int rem = int(um) - b * g.ohw;
            const int oy = int((static_cast<unsigned long long>(unsigned(rem)) * g.m_ow) >> g.sh_ow);
            const int ox = rem - oy * g.ow;
            off1_l = ok ? unsigned(b * int(a.in.sn) + oy * a.sh * int(a.in.sh) + ox * a.sw * ipitch + hh * 16) : kOOB8;
        } else {
            off1_l = ok ? unsigned(m * ipitch + hh * 16) : kOOB8;
        }
        if constexpr (DUAL) {
            // output pixel m -> (image b, oy, ox) -> the projection's input pixel (b, oy * sh, ox * sw)
            const unsigned um = ok ? unsigned(m) : 0u;
            const int b = int((static_cast<unsigned long long>(um) * g.m_ohw) >> g.sh_ohw);
            const int rem = int(um) - b * g.ohw;
            const int oy = int((static_cast<unsigned long long>(unsigned(rem)) * g.m_ow) >> g.sh_ow);
            const int ox = rem - oy * g.ow;
            off2_l = ok ? unsigned(b * int(a.in2.sn) + oy * a.sh2 * int(a.in2.sh) + ox * a.sw2 * int(a.in2.sw) + hh * 16) : kOOB8;
        }
    };
    row_offsets(rb_l);
    u32x4 ring[D];
    auto issue = [&](int slot) {
        if (!DUAL || c_l < CH1) ring[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off1_l == kOOB8 ? kOOB8 : off1_l + unsigned(c_l) * 32u, 0, 0);
        else ring[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_in2, off2_l == kOOB8 ? kOOB8 : off2_l + unsigned(c_l - CH1) * 32u, 0, 0);
        if (++c_l == CH) { c_l = 0; rb_l += stride; row_offsets(rb_l); }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);             // requested before the weight preamble: their HBM latency overlaps it

    // ---- preamble: weight slices and epilogue constants -> LDS, once per workgroup ----
    {
        auto stage_w = [&](const void* w8, int K, int P, unsigned char* dst) {
            const int k16 = K >> 4;
            constexpr int U = 8;
            const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w8), 0, Cout * K, 0x00020000);
            for (int idx0 = tid; idx0 < BN * k16; idx0 += U * NT) {
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = idx0 + u * NT;
                    const int row = idx / k16, ck = idx - row * k16;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (idx < BN * k16 && n0 + row < Cout) ? unsigned((n0 + row) * K + ck * 16) : kOOB8, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = idx0 + u * NT;
                    const int row = idx / k16, ck = idx - row * k16;
                    if (idx < BN * k16) *reinterpret_cast<u32x4*>(dst + row * P + ck * 16) = v[u];
                }
            }
        };
        stage_w(a.w8, K1, P1, sB1);
        if constexpr (DUAL) stage_w(a.w8b, K2, P2, sB2);
        // the output's 1 / scale is folded into the per-channel constants here, once per workgroup: the epilogue is one fma per product, one per
        // shortcut element and ONE v_med3_f32 for ReLU + e4m3 range clamp (scaling by a positive number commutes with both)
        for (int idx = tid; idx < BN; idx += NT) {
            const bool ok = n0 + idx < Cout;
            sE1[idx] = ok ? a.escale[n0 + idx] * a.out_qscale : 0.f;
            float b = (ok && a.bias) ? a.bias[n0 + idx] : 0.f;
            if constexpr (DUAL) {
                sE2[idx] = ok ? a.escale_b[n0 + idx] * a.out_qscale : 0.f;
                if (ok && a.bias_b) b += a.bias_b[n0 + idx];
            }
            sBi[idx] = b * a.out_qscale;
        }
    }
    __syncthreads();

    f32x16 acc[TN];
    f32x16 acc2[DUAL ? TN : 1];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[j][e] = 0.f;
            if constexpr (DUAL) acc2[j][e] = 0.f;
        }

    // one chunk = 32 channels = the 16 bytes this lane holds = two MFMA steps per N tile; weight fragments one step ahead
    auto compute = [&](const u32x4 c) {
        const i64x2 av = __builtin_bit_cast(i64x2, c);
        const bool second = DUAL && c_c >= CH1;
        const unsigned char* const Bp = (second ? sB2 + r * P2 + (c_c - CH1) * 32 : sB1 + r * P1 + c_c * 32) + hh * 16;
        const int P = second ? P2 : P1;
        i64x2 bfr[2];
        bfr[0] = *reinterpret_cast<const i64x2*>(Bp);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (j + 1 < TN) bfr[(j + 1) & 1] = *reinterpret_cast<const i64x2*>(Bp + (j + 1) * 32 * P);
            if (DUAL && second) {
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bfr[j & 1][0], av[0], acc2[j], 0, 0, 0);
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bfr[j & 1][1], av[1], acc2[j], 0, 0, 0);
            } else {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bfr[j & 1][0], av[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bfr[j & 1][1], av[1], acc[j], 0, 0, 0);
            }
        }
    };

    const int opitch = int(a.out.sw);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int(int64_t(M - 1) * opitch + Cout), 0x00020000);
    const bool has_res = a.res.p != nullptr;
    const int rpitch = int(a.res.sw);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(has_res ? a.res.p : a.out.p, 0, has_res ? int(int64_t(M - 1) * rpitch + Cout) : 0, 0x00020000);
    const float rsq = a.res_scale * a.out_qscale, lo = a.relu ? 0.f : -kE4m3Max8;
    auto epilogue = [&]() {
        const int m = rb_c * 32 + r;
        const unsigned rowoff = m < M ? unsigned(m * opitch) : kOOB8;
        const unsigned rrow = (has_res && m < M) ? unsigned(m * rpitch) : kOOB8;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + j * 32;
            unsigned rq[4] = {0u, 0u, 0u, 0u};
            if (has_res) {                                     // wave-uniform
                const int n16r = nb + 16 * hh;
                const u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(rs_res, (rrow != kOOB8 && n16r + 15 < Cout) ? rrow + unsigned(n16r) : kOOB8, 0, 0);
                const auto t0 = __builtin_amdgcn_permlane32_swap(rr[0], rr[1], false, false);
                const auto t1 = __builtin_amdgcn_permlane32_swap(rr[2], rr[3], false, false);
                rq[0] = t0[0]; rq[2] = t0[1]; rq[1] = t1[0]; rq[3] = t1[1];
            }
            unsigned d[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int nl = j * 32 + 8 * gq + 4 * hh;
                const f32x4 es = *reinterpret_cast<const f32x4*>(sE1 + nl);
                const f32x4 bs = *reinterpret_cast<const f32x4*>(sBi + nl);
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[j][4 * gq + q] * es[q] + bs[q];
                if constexpr (DUAL) {
                    const f32x4 e2 = *reinterpret_cast<const f32x4*>(sE2 + nl);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += acc2[j][4 * gq + q] * e2[q];
                }
                if (has_res) {
                    float rv[4];
                    unpack4_e4m3_ws(rq[gq], rv);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += rv[q] * rsq;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = __builtin_amdgcn_fmed3f(v[q], lo, kE4m3Max8);
                int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
                d[gq] = unsigned(pk);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[j][4 * gq + q] = 0.f;
                    if constexpr (DUAL) acc2[j][4 * gq + q] = 0.f;
                }
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            const int n16 = nb + 16 * hh;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s0[1], s1[0], s1[1]}, rs_out, (rowoff != kOOB8 && n16 + 15 < Cout) ? rowoff + unsigned(n16) : kOOB8, 0, 0);
        }
    };

    while (rb_c < nrb) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            compute(ring[s]);
            issue(s);
            if (++c_c == CH) {
                epilogue();
                c_c = 0;
                rb_c += stride;
            }
        }
    }
}

struct Ws8Tile { int tn, waves; };
constexpr int kNumWs8Shapes = 5;             // tile % 5 = shape; tile / 5 = 1: the persistent grid sized for ONE workgroup per CU
constexpr Ws8Tile kWs8Tiles[kNumWs8Shapes] = {{8, 8}, {4, 8}, {2, 8}, {4, 4}, {2, 4}};

static size_t ws8_lds_bytes(int tn, int K1, int K2) {
    const size_t bn = size_t(32) * tn;
    return bn * (K1 + 16) + (K2 > 0 ? bn * (K2 + 16) : 0) + 3 * bn * sizeof(float);
}

static bool ws8_tensor_ok(const TensorArg& t) {
    return t.f8 && t.sc == 1 && !(t.c & 31) && !(t.sw & 15) && !(t.sh & 15) && !(t.sn & 15) && !(reinterpret_cast<uintptr_t>(t.p) & 15);
}

bool ConvWs8Eligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvWs8Tiles) return false;
    const bool dual = a.in2.p != nullptr;
    if (!ws8_tensor_ok(a.in) || !a.out.f8 || a.w8 == nullptr || a.escale == nullptr || a.pre_scale != nullptr) return false;
    if (a.kh != 1 || a.kw != 1 || a.sh < 1 || a.sw < 1 || a.pt != 0 || a.pl != 0 || a.in.n != a.out.n) return false;
    if (a.out.h != (a.in.h - 1) / a.sh + 1 || a.out.w != (a.in.w - 1) / a.sw + 1) return false;
    if ((a.sh != 1 || a.sw != 1) && dual) return false;                                                       // a strided first input: the single-GEMM variant only
    if (a.in.sh != a.in.w * a.in.sw || a.in.sn != a.in.h * a.in.sh) return false;                          // pixels at a constant pitch
    if (a.out.sc != 1 || (a.out.c & 15) || (a.out.sw & 15) || (reinterpret_cast<uintptr_t>(a.out.p) & 15)) return false;
    if (a.out.sh != a.out.w * a.out.sw || a.out.sn != a.out.h * a.out.sh) return false;
    if ((reinterpret_cast<uintptr_t>(a.w8) & 15)) return false;
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    if (int64_t(a.in.n) * a.in.sn >= (int64_t(1) << 31) || M * a.out.sw >= (int64_t(1) << 31) || int64_t(a.out.c) * a.in.c >= (int64_t(1) << 31)) return false;
    if (a.res.p != nullptr) {
        if (!a.res.f8 || a.res.sc != 1 || (a.res.sw & 15) || (reinterpret_cast<uintptr_t>(a.res.p) & 15) || a.res.c != a.out.c) return false;
        if (a.res.sh != a.res.w * a.res.sw || a.res.sn != a.res.h * a.res.sh || a.res.n != a.out.n || a.res.h != a.out.h || a.res.w != a.out.w) return false;
        if (M * a.res.sw >= (int64_t(1) << 31)) return false;
    }
    if (dual) {
        if (!ws8_tensor_ok(a.in2) || a.w8b == nullptr || a.escale_b == nullptr || (reinterpret_cast<uintptr_t>(a.w8b) & 15) || a.sh2 < 1 || a.sw2 < 1) return false;
        if (a.in2.n != a.out.n || (a.out.h - 1) * a.sh2 >= a.in2.h || (a.out.w - 1) * a.sw2 >= a.in2.w) return false;
        const int64_t span2 = int64_t(a.in2.n - 1) * a.in2.sn + int64_t(a.in2.h - 1) * a.in2.sh + int64_t(a.in2.w - 1) * a.in2.sw + a.in2.c;
        if (span2 >= (int64_t(1) << 31) || int64_t(a.out.c) * a.in2.c >= (int64_t(1) << 31)) return false;
    }
    const Ws8Tile t = kWs8Tiles[tile % kNumWs8Shapes];
    if (dual && t.tn > 4) return false;                                                       // two accumulator sets: 2 x 4 x 16 registers
    if (ws8_lds_bytes(t.tn, a.in.c, dual ? a.in2.c : 0) > size_t(160) * 1024) return false;
    if (t.tn > 1 && a.out.c <= 32 * (t.tn / 2)) return false;                                 // do not waste MFMA rows on padding
    return true;
}

static void magic_div8(unsigned d, unsigned long long* m, int* sh) {      // floor(j / d) = (j * m) >> sh for 0 <= j < 2^31
    int L = 0;
    while ((1ull << L) < d) ++L;
    *sh = 31 + L;
    *m = ((1ull << (31 + L)) / d) + 1;
}

template <int TN, int WAVES, bool DUAL, bool STR = false>
static hipError_t launch_ws8_t(const ConvArgs& a, bool one_per_cu, hipStream_t stream) {
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    const int nrb = int((M + 31) / 32);
    const size_t lds = ws8_lds_bytes(TN, a.in.c, DUAL ? a.in2.c : 0);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount;
    }
    int per_cu = int((size_t(160) * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2048 / (64 * WAVES) ? 2048 / (64 * WAVES) : per_cu);
    if (per_cu > 2) per_cu = 2;                      // (the grid that measured best; see launch_ws_t in kernels_ws.hip on sizing by true residency)
    if (one_per_cu) per_cu = 1;                      // tiles 5-9: fewer, longer streams (scripts/probes/ws8_probe.cpp -DWS8_PER_CU); the search decides per shape
#ifdef WS8_PER_CU
    per_cu = WS8_PER_CU;                             // (probe builds)
#endif
    const int gy = (a.out.c + 32 * TN - 1) / (32 * TN);
    int slots = cus * per_cu / gy;
    if (slots < 8) slots = 8;
    const int iters = (nrb + slots * WAVES - 1) / (slots * WAVES);
    int gx = (nrb + iters * WAVES - 1) / (iters * WAVES);
    gx = (gx + 7) & ~7;
    Ws8Geom g{};
    g.ohw = a.out.h * a.out.w;
    g.ow = a.out.w;
    magic_div8(unsigned(g.ohw), &g.m_ohw, &g.sh_ohw);
    magic_div8(unsigned(g.ow), &g.m_ow, &g.sh_ow);
    conv1x1_ws_f8_kernel<TN, WAVES, DUAL, STR><<<dim3(gx, gy), dim3(64 * WAVES), lds, stream>>>(a, g);
    return hipGetLastError();
}

hipError_t LaunchConvWs1x1F8(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvWs8Eligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + a.in.c;
    const bool dual = a.in2.p != nullptr;
    const bool pc1 = tile >= kNumWs8Shapes;
    tile %= kNumWs8Shapes;
    if (dual) a.in2_bytes = int64_t(a.in2.n - 1) * a.in2.sn + int64_t(a.in2.h - 1) * a.in2.sh + int64_t(a.in2.w - 1) * a.in2.sw + a.in2.c;
    if (a.sh != 1 || a.sw != 1) {
        switch (tile) {
            case 0: return launch_ws8_t<8, 8, false, true>(a, pc1, stream);
            case 1: return launch_ws8_t<4, 8, false, true>(a, pc1, stream);
            case 2: return launch_ws8_t<2, 8, false, true>(a, pc1, stream);
            case 3: return launch_ws8_t<4, 4, false, true>(a, pc1, stream);
            case 4: return launch_ws8_t<2, 4, false, true>(a, pc1, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (tile) {
        case 0: return launch_ws8_t<8, 8, false>(a, pc1, stream);
        case 1: return dual ? launch_ws8_t<4, 8, true>(a, pc1, stream) : launch_ws8_t<4, 8, false>(a, pc1, stream);
        case 2: return dual ? launch_ws8_t<2, 8, true>(a, pc1, stream) : launch_ws8_t<2, 8, false>(a, pc1, stream);
        case 3: return dual ? launch_ws8_t<4, 4, true>(a, pc1, stream) : launch_ws8_t<4, 4, false>(a, pc1, stream);
        case 4: return dual ? launch_ws8_t<2, 4, true>(a, pc1, stream) : launch_ws8_t<2, 4, false>(a, pc1, stream);
        default: return hipErrorInvalidValue;
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Weights-stationary 3x3 / stride 1 / pad 1 convolution for the fp8 mode (ResNet-50's bottleneck 3x3s of stages 1-3): conv3x3_ws_f16_kernel
// (kernels_ws.hip) re-typed.  Same 1-D raster of the zero-padded image stack (pitch W + 1, one shared pad column / row), same LDS byte
// layout -- window rows and weight rows of 128 B data + 16 B pad -- but a row now holds 128 e4m3 channels, a 16-byte fragment read feeds
// TWO v_mfma_f32_32x32x16_fp8_fp8 steps, and the epilogue re-quantises (per-channel multiplier, bias, ReLU, 1 / output scale).  All weights
// of one 32-channel N tile stay in LDS for the life of a persistent workgroup (41 KB per 128 input channels); N tiles ride on blockIdx.y.
// The tiled implicit GEMM ran these layers at 430 (stage 1) ... 850 TFLOP/s with nine im2col copies of every window through LDS.
// ------------------------------------------------------------------------------------------------------------------------
struct Ws3Geom8 {
    int PW, RH, PR, num_tiles, nslices;
    int sh_img, sh_pw;
    unsigned long long m_img, m_pw;
};

template <int WAVES, int TMW, int PIT, int NKK>
__global__ __launch_bounds__(64 * WAVES) void conv3x3_ws_f8_kernel(const ConvArgs a, const Ws3Geom8 g) {
    constexpr int NT = 64 * WAVES, BMp = 32 * TMW * WAVES, LDP = 144, RPP = NT / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w8[];
    const int NS = g.nslices, PW = g.PW, PR = g.PR;
    unsigned char* const sW = smem_w8;                                 // [9][NS][32][LDP]
    unsigned char* const sP = sW + 9 * NS * 32 * LDP;                  // [PR][LDP]
    float* const sE = reinterpret_cast<float*>(sP + PR * LDP);         // [32] epilogue multipliers, [32] bias

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.in.c, H = a.in.h, W = a.in.w, Cout = a.out.c;
    const int n0 = blockIdx.y * 32;
    const int img = g.RH * PW, Mr = a.in.n * img;
    const int isw = int(a.in.sw), opitch = int(a.out.sw);

    auto pix_of = [&](int j) -> int {      // raster position -> pixel index (b * H + y) * W + x, or -1 for pad rows / columns and positions outside the raster
        if (j < 0 || j >= Mr) return -1;
        const int b = int((static_cast<unsigned long long>(unsigned(j)) * g.m_img) >> g.sh_img);
        const int rem = j - b * img;
        const int y = int((static_cast<unsigned long long>(unsigned(rem)) * g.m_pw) >> g.sh_pw);
        const int x = rem - y * PW;
        return (y < H && x < W) ? (b * H + y) * W + x : -1;
    };

    // ---- preamble: every weight of this N tile -> LDS (zero-filled past Cin / Cout), epilogue constants -> LDS ----
    {
        const int items = 9 * NS * 32 * 8;
        constexpr int U = 8;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w8), 0, Cout * 9 * Cin, 0x00020000);
        for (int q0 = tid; q0 < items; q0 += U * NT) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                const int ck = q & 7, row = q >> 3;      // row = (tap * NS + slice) * 32 + n
                const int n = row & 31, ts = row >> 5;
                const int tap = ts / NS, sl = ts - tap * NS;
                const int c = sl * 128 + ck * 16;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (q < items && n0 + n < Cout && c < Cin) ? unsigned(((n0 + n) * 9 + tap) * Cin + c) : kOOB8, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                if (q < items) *reinterpret_cast<u32x4*>(sW + (q >> 3) * LDP + (q & 7) * 16) = v[u];
            }
        }
        for (int q = tid; q < 32; q += NT) {
            sE[q] = n0 + q < Cout ? a.escale[n0 + q] : 0.f;
            sE[32 + q] = (a.bias != nullptr && n0 + q < Cout) ? a.bias[n0 + q] : 0.f;
        }
    }

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out.p, 0, int((int64_t(a.out.n) * a.out.h * a.out.w - 1) * opitch + Cout), 0x00020000);

    f32x16 acc[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    const int c16 = (tid & 7) * 16;
    int poff[PIT];                                     // byte offset of (pixel, c16) of this thread's window rows, or -1
    u32x4 pv[PIT];
    auto decode_rows = [&](int tile) {
        const int jbase = tile * BMp - PW - 1;
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int l = (tid >> 3) + i * RPP;
            const int pix = l < PR ? pix_of(jbase + l) : -1;
            poff[i] = pix >= 0 ? pix * isw + c16 : -1;
        }
    };
    auto issue = [&](int sl) {
        const int c0 = sl * 128;
        const bool cok = c0 + c16 < Cin;
#pragma unroll
        for (int i = 0; i < PIT; ++i) pv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (poff[i] >= 0 && cok) ? unsigned(poff[i] + c0) : kOOB8, 0, 0);
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int l = (tid >> 3) + i * RPP;
            if (l < PR) *reinterpret_cast<u32x4*>(sP + l * LDP + c16) = pv[i];
        }
    };
    // nine shifted GEMMs out of LDS: 9 x NKK fragment pairs (32 channels = two MFMA steps each), reads one pair ahead of the MFMAs
    auto compute_slice = [&](int sl) {
        const unsigned char* const Abase = sP + (wave * 32 * TMW + r) * LDP + hh * 16;
        const unsigned char* const Bbase = sW + (sl * 32 + r) * LDP + hh * 16;
        constexpr int STEPS = 9 * NKK;
        i64x2 af[2][TMW], bf[2];
        auto read_step = [&](int st, int slot) {
            const int tap = st / NKK, kk = st - tap * NKK;
            const int shift = (tap / 3) * PW + (tap % 3);
            const unsigned char* const A = Abase + shift * LDP + kk * 32;
#pragma unroll
            for (int i = 0; i < TMW; ++i) af[slot][i] = *reinterpret_cast<const i64x2*>(A + i * 32 * LDP);
            bf[slot] = *reinterpret_cast<const i64x2*>(Bbase + tap * NS * 32 * LDP + kk * 32);
        };
        read_step(0, 0);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int cur = st & 1;
            if (st + 1 < STEPS) read_step(st + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bf[cur][0], af[cur][i][0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bf[cur][1], af[cur][i][1], acc[i], 0, 0, 0);
            }
        }
    };
    const float qs = a.out_qscale;
    auto epilogue = [&](int tile) {
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int pix = pix_of(tile * BMp + (wave * TMW + i) * 32 + r);
            const unsigned rowoff = pix >= 0 ? unsigned(pix * opitch) : kOOB8;
            unsigned d[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 es = *reinterpret_cast<const f32x4*>(sE + 8 * gq + 4 * hh);
                const f32x4 bs = *reinterpret_cast<const f32x4*>(sE + 32 + 8 * gq + 4 * hh);
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = acc[i][4 * gq + q] * es[q] + bs[q];
                    if (a.relu) v[q] = fmaxf(v[q], 0.f);
                    acc[i][4 * gq + q] = 0.f;
                }
                d[gq] = pack4_e4m3_ws(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs);
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            const int n16 = n0 + 16 * hh;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s0[1], s1[0], s1[1]}, rs_out, (rowoff != kOOB8 && n16 + 15 < Cout) ? rowoff + unsigned(n16) : kOOB8, 0, 0);
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

struct Ws3Tile8 { int waves, tmw, pit; };
constexpr int kNumWs38Shapes = 4;
constexpr Ws3Tile8 kWs3Tiles8[kNumWs38Shapes] = {{4, 2, 12}, {4, 1, 8}, {8, 1, 6}, {8, 2, 10}};

static size_t ws38_lds_bytes(int tile, int Cin, int PW) {
    const Ws3Tile8 t = kWs3Tiles8[tile];
    const int NS = (Cin + 127) / 128, PR = 32 * t.tmw * t.waves + 2 * PW + 2;
    return size_t(9 * NS * 32 + PR) * 144 + 64 * sizeof(float);
}

bool ConvWs38Eligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvWs38Tiles) return false;
    if (!a.in.f8 || !a.out.f8 || a.w8 == nullptr || a.escale == nullptr || a.pre_scale != nullptr || a.res.p != nullptr || a.in2.p != nullptr) return false;
    if (a.kh != 3 || a.kw != 3 || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1 || a.out.h != a.in.h || a.out.w != a.in.w || a.out.n != a.in.n) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || (a.in.c & 31) || (a.in.sw & 15) || (reinterpret_cast<uintptr_t>(a.in.p) & 15) || (reinterpret_cast<uintptr_t>(a.w8) & 15)) return false;
    if ((a.out.c & 15) || (a.out.sw & 15) || (reinterpret_cast<uintptr_t>(a.out.p) & 15)) return false;
    if (a.in.sh != a.in.sw * a.in.w || a.in.sn != a.in.sh * a.in.h) return false;
    if (a.out.sh != a.out.sw * a.out.w || a.out.sn != a.out.sh * a.out.h) return false;
    const int64_t Mr = int64_t(a.in.n) * (a.in.h + 1) * (a.in.w + 1), Mpix = int64_t(a.in.n) * a.in.h * a.in.w;
    if (Mr + 4096 >= (int64_t(1) << 31) || Mpix * a.in.sw >= (int64_t(1) << 31) || Mpix * a.out.sw >= (int64_t(1) << 31) || int64_t(a.out.c) * 9 * a.in.c >= (int64_t(1) << 31)) return false;
    const Ws3Tile8 t = kWs3Tiles8[tile % kNumWs38Shapes];
    const int PR = 32 * t.tmw * t.waves + 2 * (a.in.w + 1) + 2;
    if (PR > t.pit * (64 * t.waves / 8)) return false;
    return ws38_lds_bytes(tile % kNumWs38Shapes, a.in.c, a.in.w + 1) <= size_t(160) * 1024;
}

template <int T, int NKK>
static hipError_t launch_ws38_t(const ConvArgs& a, bool one_per_cu, hipStream_t stream) {
    constexpr Ws3Tile8 t = kWs3Tiles8[T];
    constexpr int BMp = 32 * t.tmw * t.waves;
    Ws3Geom8 g;
    g.PW = a.in.w + 1;
    g.RH = a.in.h + 1;
    g.PR = BMp + 2 * g.PW + 2;
    g.nslices = (a.in.c + 127) / 128;
    const int64_t Mr = int64_t(a.in.n) * g.RH * g.PW;
    g.num_tiles = int((Mr + BMp - 1) / BMp);
    magic_div8(unsigned(g.RH * g.PW), &g.m_img, &g.sh_img);
    magic_div8(unsigned(g.PW), &g.m_pw, &g.sh_pw);
    const size_t lds = ws38_lds_bytes(T, a.in.c, g.PW);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount;
    }
    int per_cu = int((size_t(160) * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    if (one_per_cu) per_cu = 1;                      // tiles 4-7
    const int gy = (a.out.c + 31) / 32;
    int slots = cus * per_cu / gy;
    if (slots < 1) slots = 1;
    const int iters = (g.num_tiles + slots - 1) / slots;
    const int gx = (g.num_tiles + iters - 1) / iters;
    conv3x3_ws_f8_kernel<t.waves, t.tmw, t.pit, NKK><<<dim3(gx, gy), dim3(64 * t.waves), lds, stream>>>(a, g);
    return hipGetLastError();
}

hipError_t LaunchConvWs3x3F8(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvWs38Eligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + a.in.c;
    const bool half_slice = a.in.c <= 64;             // a 64-channel layer fills half of the 128-channel slice: two fragment pairs per tap instead of four
#define IE_WS38(T) case T: return half_slice ? launch_ws38_t<T, 2>(a, pc1, stream) : launch_ws38_t<T, 4>(a, pc1, stream);
    const bool pc1 = tile >= kNumWs38Shapes;
    switch (tile % kNumWs38Shapes) {
        IE_WS38(0) IE_WS38(1) IE_WS38(2) IE_WS38(3)
        default: return hipErrorInvalidValue;
    }
#undef IE_WS38
}

hipError_t InitKernelsWs8() {
    hipError_t e;
#define IE_WS8I(TN, W, DU)                                                                                                                       \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f8_kernel<TN, W, DU>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024)) != hipSuccess) return e;
    IE_WS8I(8, 8, false) IE_WS8I(4, 8, false) IE_WS8I(2, 8, false) IE_WS8I(4, 4, false) IE_WS8I(2, 4, false)
    IE_WS8I(4, 8, true) IE_WS8I(2, 8, true) IE_WS8I(4, 4, true) IE_WS8I(2, 4, true)
#undef IE_WS8I
#define IE_WS8S(TN, W)                                                                                                                                 \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_f8_kernel<TN, W, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024)) != hipSuccess) return e;
    IE_WS8S(8, 8) IE_WS8S(4, 8) IE_WS8S(2, 8) IE_WS8S(4, 4) IE_WS8S(2, 4)
#undef IE_WS8S
#define IE_WS38I(T, NKK)                                                                                                                   \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ws_f8_kernel<kWs3Tiles8[T].waves, kWs3Tiles8[T].tmw, kWs3Tiles8[T].pit, NKK>), \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    IE_WS38I(0, 2) IE_WS38I(0, 4) IE_WS38I(1, 2) IE_WS38I(1, 4) IE_WS38I(2, 2) IE_WS38I(2, 4) IE_WS38I(3, 2) IE_WS38I(3, 4)
#undef IE_WS38I
    return hipSuccess;
}

}  // namespace ie
