// fp8 precision mode (BASELINE configs[4]: "ResNet-50 ONNX fp8 batch=256 ... CDNA4 fp8 MFMA for 1x1/3x3 conv-as-GEMM").
//
// Activations between the stem and the global pool live in HBM as OCP e4m3 bytes (one per-tensor scale each, chosen at load from
// a calibration pass: real = q * s), conv weights as e4m3 with one scale per output channel, every accumulation in fp32 on
// v_mfma_f32_32x32x16_fp8_fp8.  A conv's epilogue turns the integer-free fp32 dot product of the two e4m3 operands back into real
// units with ONE per-channel multiplier (escale[o] = s_in * s_w[o]), adds the folded bias and the residual shortcut (its own e4m3
// tensor, dequantised with its scale), applies ReLU and re-quantises with the output tensor's 1/s.  The reference never computes
// in fp8 (ONNX Runtime runs the model's own fp32, model.cpp:1264-1270): parity of this mode is unpinned; tests state the tolerance.
//
// The conv kernel is the fp16 implicit GEMM (kernels_f16.hip) re-typed: the LDS image keeps the SAME byte layout ([rows][128 B data
// + 16 B pad], pitch/16 odd -> conflict-free ds_read_b128); one 16-byte fragment read now carries TWO K=16 MFMA steps (low and high
// 8 bytes) and a K-tile is 128 channels deep, so each byte fetched from HBM feeds 2x the MACs of the fp16 path.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "igemm_tiles.h"
#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef long i64x2 __attribute__((ext_vector_type(2)));

constexpr float kE4m3Max = 448.0f;

// four floats -> four e4m3 bytes (round to nearest even; the inputs are clamped to the finite range first, so nothing overflows
// into the NaN encoding)
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    a = __builtin_fminf(__builtin_fmaxf(a, -kE4m3Max), kE4m3Max);
    b = __builtin_fminf(__builtin_fmaxf(b, -kE4m3Max), kE4m3Max);
    c = __builtin_fminf(__builtin_fmaxf(c, -kE4m3Max), kE4m3Max);
    d = __builtin_fminf(__builtin_fmaxf(d, -kE4m3Max), kE4m3Max);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return unsigned(p);
}
__device__ __forceinline__ void unpack4_e4m3(unsigned p, float* v) {
    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(int(p), false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(int(p), true);
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_f8_kernel(const ConvArgs a, const int tiles_n, const int num_tiles) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BKE = 128;                     // K-tile depth in e4m3 elements = bytes (128 B per row, as in the fp32 / fp16 kernels)
    constexpr int LDP = BKE + 16;                // row pitch in bytes (144)
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "bad tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
    unsigned char* const sA = smem_q;                    // [2][BM][LDP]
    unsigned char* const sB = sA + 2 * BM * LDP;         // [2][BN][LDP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm_i = wave / WN, wn_i = wave % WN;
    const int r = lane & 31, hh = lane >> 5;

    int m0, n0;
    {
        const int lin = blockIdx.x;
        const int q = num_tiles >> 3, rem = num_tiles & 7, xcd = lin & 7;
        const int swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (lin >> 3);
        m0 = (swz / tiles_n) * BM;
        n0 = (swz % tiles_n) * BN;
    }

    const int Cin = a.in.c, H = a.in.h, W = a.in.w;
    const int OH = a.out.h, OW = a.out.w, Cout = a.out.c;
    const int M = a.out.n * OH * OW;
    const int Ktot = a.kh * a.kw * Cin;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // one 16-byte fragment per operand per TWO MFMAs (K = 16 each); reads of step kk+1 are issued before the MFMAs of step kk
    auto compute = [&](int buf) {
        const unsigned char* A = sA + buf * BM * LDP + (wm_i * TM * 32 + r) * LDP + hh * 16;
        const unsigned char* B = sB + buf * BN * LDP + (wn_i * TN * 32 + r) * LDP + hh * 16;
        i64x2 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const i64x2*>(A + i * 32 * LDP);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const i64x2*>(B + j * 32 * LDP);
#pragma unroll
        for (int kk = 0; kk < BKE / 32; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BKE / 32) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = *reinterpret_cast<const i64x2*>(A + i * 32 * LDP + (kk + 1) * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nxt][j] = *reinterpret_cast<const i64x2*>(B + j * 32 * LDP + (kk + 1) * 32);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bf[cur][j][0], af[cur][i][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bf[cur][j][1], af[cur][i][1], acc[i][j], 0, 0, 0);
                }
        }
    };

    // ---- staging: thread owns the 16-byte chunk `c16` (16 channels) of rows {rw + i*ROWS_PER_PASS} ----
    constexpr int ROWS_PER_PASS = NT / 8;
    constexpr int A_IT = BM / ROWS_PER_PASS, B_IT = BN / ROWS_PER_PASS;
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the thread count");
    const int c16 = (tid & 7) * 16;
    const int rw = tid >> 3;
    const int cblocks = (Cin + BKE - 1) / BKE;
    const int KT = a.kh * a.kw * cblocks;
    const int ish = int(a.in.sh), isw = int(a.in.sw);
    constexpr unsigned OOB = 0x80000000u;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w8), 0, Cout * Ktot, 0x00020000);

    int poff[A_IT];
    unsigned taps[A_IT];
    int boff[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + rw + i * ROWS_PER_PASS;
        const bool mok = m < M;
        const int mm = mok ? m : 0;
        const int b = mm / (OH * OW);
        const int rem = mm - b * (OH * OW);
        const int oy = rem / OW, ox = rem - oy * OW;
        const int iy0 = oy * a.sh - a.pt, ix0 = ox * a.sw - a.pl;
        poff[i] = b * int(a.in.sn) + iy0 * ish + ix0 * isw + c16;
        unsigned msk = 0;
        for (int ky = 0; ky < a.kh; ++ky)
            for (int kx = 0; kx < a.kw; ++kx)
                if (unsigned(iy0 + ky) < unsigned(H) && unsigned(ix0 + kx) < unsigned(W)) msk |= 1u << (ky * a.kw + kx);
        taps[i] = mok ? msk : 0u;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + rw + i * ROWS_PER_PASS;
        boff[i] = n < Cout ? n * Ktot + c16 : -1;
    }

    f32x4 ra[A_IT], rb[B_IT];
    auto issue_loads = [&](int kt) {
        const int tap = kt / cblocks;
        const int c0 = (kt - tap * cblocks) * BKE;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        const int tapoff = ky * ish + kx * isw + c0;
        const int woff = tap * Cin + c0;
        const bool cok = c0 + c16 < Cin;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bool ok = cok && ((taps[i] >> tap) & 1u);
            const unsigned off = ok ? unsigned(poff[i] + tapoff) : OOB;       // e4m3 zero == byte zero: out-of-range lanes are the padding
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const unsigned off = (cok && boff[i] >= 0) ? unsigned(boff[i] + woff) : OOB;
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
        }
    };
    auto finish_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<f32x4*>(sA + buf * BM * LDP + (rw + i * ROWS_PER_PASS) * LDP + c16) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<f32x4*>(sB + buf * BN * LDP + (rw + i * ROWS_PER_PASS) * LDP + c16) = rb[i];
    };

    if (KT > 0) {
        issue_loads(0);
        finish_store(0);
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT;
        if (more) issue_loads(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        if (more) finish_store(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  D = W x A^T: a lane owns ONE pixel (m = column r) and, per accumulator quad g, FOUR consecutive channels
    //      n = 8g + 4hh + q.  Quads are scaled / biased / shortcut-added / activated in fp32, re-quantised to one dword of four e4m3,
    //      then v_permlane32_swap pairs the half-waves so each lane stores the 16 consecutive channels 16*hh .. 16*hh+15 of its
    //      pixel in one 16-byte store. ----
    const int opitch = int(a.out.sw);
    const int rpitch = int(a.res.sw);
    const bool has_res = a.res.p != nullptr;
    const unsigned char* const resb = reinterpret_cast<const unsigned char*>(a.res.p);
    unsigned char* const outb = reinterpret_cast<unsigned char*>(a.out.p);
    const float qs = a.out_qscale, rs = a.res_scale;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm_i * TM + i) * 32 + r;
        const bool mok = m < M;
        const int64_t orow = int64_t(m) * opitch, rrow = int64_t(m) * rpitch;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + (wn_i * TN + j) * 32;
            // shortcut: ONE 16-byte load per lane (the 16 consecutive channels 16*hh .. +15 of its pixel, the layout it will store),
            // then the inverse of the store exchange hands every lane the four quads it accumulates (channels 8g + 4hh .. +3)
            unsigned rq[4] = {0u, 0u, 0u, 0u};
            if (has_res) {                                     // wave-uniform
                const int n16r = nb + 16 * hh;
                u32x4 rr = {0u, 0u, 0u, 0u};
                if (mok && n16r + 15 < Cout) rr = *reinterpret_cast<const u32x4*>(resb + rrow + n16r);
                const auto t0 = __builtin_amdgcn_permlane32_swap(rr[0], rr[1], false, false);     // lower: {own d0, upper's d0}; upper: {lower's d1, own d1}
                const auto t1 = __builtin_amdgcn_permlane32_swap(rr[2], rr[3], false, false);
                rq[0] = t0[0]; rq[2] = t0[1]; rq[1] = t1[0]; rq[3] = t1[1];
            }
            unsigned d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb + 8 * g + 4 * hh;
                const bool nok = n + 3 < Cout;
                const int nn = nok ? n : 0;
                const f32x4 es = *reinterpret_cast<const f32x4*>(a.escale + nn);
                f32x4 bs = {0.f, 0.f, 0.f, 0.f};
                if (a.bias) bs = *reinterpret_cast<const f32x4*>(a.bias + nn);
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] * es[q] + bs[q];
                if (has_res) {
                    float rv[4];
                    unpack4_e4m3(rq[g], rv);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += rv[q] * rs;
                }
                if (a.relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                d[g] = pack4_e4m3(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs);
            }
            // lower lanes keep quads g = 0, 1 and receive the partner's; upper lanes keep g = 2, 3 (see the header comment)
            const auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            const int n16 = nb + 16 * hh;
            if (mok && n16 + 15 < Cout) *reinterpret_cast<u32x4*>(outb + orow + n16) = u32x4{s0[0], s0[1], s1[0], s1[1]};
        }
    }
}

template <int T>
static size_t f8_lds_bytes() {
    constexpr IgemmTile t = kIgemmTiles[T];
    return size_t(2) * (t.bm + t.bn) * 144;
}

template <int T>
static hipError_t launch_f8_t(const ConvArgs& a, hipStream_t stream) {
    constexpr IgemmTile t = kIgemmTiles[T];
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    const int tiles_m = int((M + t.bm - 1) / t.bm), tiles_n = (a.out.c + t.bn - 1) / t.bn;
    const int num_tiles = tiles_m * tiles_n;
    conv_igemm_f8_kernel<t.bm, t.bn, t.wm, t.wn><<<dim3(num_tiles), dim3(64 * t.wm * t.wn), f8_lds_bytes<T>(), stream>>>(a, tiles_n, num_tiles);
    return hipGetLastError();
}

template <int T>
static hipError_t init_f8_t() {
    constexpr IgemmTile t = kIgemmTiles[T];
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f8_kernel<t.bm, t.bn, t.wm, t.wn>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024);
}

bool ConvF8Eligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumConvF8Tiles) return false;
    if (!a.in.f8 || !a.out.f8 || a.w8 == nullptr || a.escale == nullptr || a.pre_scale != nullptr) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || a.kh * a.kw > 32) return false;
    if ((a.in.c & 15) || (a.in.sw & 15) || (a.in.sh & 15) || (a.in.sn & 15) || (reinterpret_cast<uintptr_t>(a.in.p) & 15)) return false;
    if ((a.out.c & 15) || (a.out.sw & 15) || (reinterpret_cast<uintptr_t>(a.out.p) & 15)) return false;
    if ((reinterpret_cast<uintptr_t>(a.w8) & 15) || (reinterpret_cast<uintptr_t>(a.escale) & 15) || (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15))) return false;
    if (a.res.p && (!a.res.f8 || a.res.sc != 1 || (a.res.sw & 15) || (reinterpret_cast<uintptr_t>(a.res.p) & 15) || a.res.n != a.out.n || a.res.h != a.out.h ||
                    a.res.w != a.out.w || a.res.c != a.out.c))
        return false;
    const int64_t in_span = int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + a.in.c;
    if (in_span >= (int64_t(1) << 31) || int64_t(a.out.c) * a.kh * a.kw * a.in.c >= (int64_t(1) << 31)) return false;
    if (int64_t(a.out.n) * a.out.h * a.out.w * a.out.sw >= (int64_t(1) << 31)) return false;
    if (kIgemmTiles[tile].bn > 32 && a.out.c <= 32) return false;
    return true;
}

hipError_t LaunchConvIgemmF8(const ConvArgs& a_in, int tile, hipStream_t stream) {
    if (!ConvF8Eligible(a_in, tile)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.in_bytes = int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + a.in.c;
    switch (tile) {
        case 0: return launch_f8_t<0>(a, stream);
        case 1: return launch_f8_t<1>(a, stream);
        case 2: return launch_f8_t<2>(a, stream);
        case 3: return launch_f8_t<3>(a, stream);
        case 4: return launch_f8_t<4>(a, stream);
        case 5: return launch_f8_t<5>(a, stream);
        case 6: return launch_f8_t<6>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t InitKernelsF8() {
    hipError_t e;
    if ((e = init_f8_t<0>()) != hipSuccess) return e;
    if ((e = init_f8_t<1>()) != hipSuccess) return e;
    if ((e = init_f8_t<2>()) != hipSuccess) return e;
    if ((e = init_f8_t<3>()) != hipSuccess) return e;
    if ((e = init_f8_t<4>()) != hipSuccess) return e;
    if ((e = init_f8_t<5>()) != hipSuccess) return e;
    return init_f8_t<6>();
}

// ------------------------------------------------------------------------------------------------------------------------
// weight quantisation: one workgroup per output channel.  wscale[o] = max|w[o, :]| / 448 (1 for an all-zero row),
// w8[o, k] = e4m3(w[o, k] / wscale[o]).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_rows_e4m3_kernel(const float* __restrict__ w, unsigned char* __restrict__ w8, float* __restrict__ wscale,
                                                                  const int K) {
    __shared__ float red[4];
    const int o = blockIdx.x;
    const float* row = w + int64_t(o) * K;
    float m = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, fabsf(row[k]));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = m > 0.f ? m / kE4m3Max : 1.f;
    if (threadIdx.x == 0) wscale[o] = sc;
    const float inv = 1.f / sc;
    unsigned char* out = w8 + int64_t(o) * K;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        if (k + 3 < K && ((reinterpret_cast<uintptr_t>(out + k) & 3) == 0)) {
            *reinterpret_cast<unsigned*>(out + k) = pack4_e4m3(row[k] * inv, row[k + 1] * inv, row[k + 2] * inv, row[k + 3] * inv);
        } else {
            for (int q = 0; q < 4 && k + q < K; ++q) out[k + q] = (unsigned char)(pack4_e4m3(row[k + q] * inv, 0.f, 0.f, 0.f) & 0xffu);
        }
    }
}

hipError_t LaunchQuantizeRowsE4m3(const float* w, void* w8, float* wscale, int rows, int K, hipStream_t stream) {
    if (rows <= 0 || K <= 0) return hipSuccess;
    quantize_rows_e4m3_kernel<<<dim3(rows), dim3(256), 0, stream>>>(w, static_cast<unsigned char*>(w8), wscale, K);
    return hipGetLastError();
}

__global__ void scale_vector_kernel(const float* __restrict__ src, float* __restrict__ dst, const float s, const int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * s;
}

hipError_t LaunchScaleVector(const float* src, float* dst, float s, int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    scale_vector_kernel<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(src, dst, s, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// pooling on e4m3 NHWC tensors: 16 channels (16 bytes) per lane.  Max pooling keeps the input's scale (max commutes with a
// positive scale and returns one of its inputs: no rounding at all); average pooling dequantises, averages in fp32 and re-quantises
// with the output tensor's scale.
// ------------------------------------------------------------------------------------------------------------------------
__global__ void pool_f8_kernel(const PoolArgs a, const int64_t total) {
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int CV = a.out.c / 16;
    const int c = int(idx % CV) * 16;
    int64_t m = idx / CV;
    const int ox = int(m % a.out.w); m /= a.out.w;
    const int oy = int(m % a.out.h);
    const int b = int(m / a.out.h);
    float acc[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = a.is_max ? -INFINITY : 0.f;
    int cnt = 0;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in.p);
    for (int ky = 0; ky < a.kh; ++ky) {
        const int iy = oy * a.sh - a.pt + ky;
        if (iy >= a.in.h + a.pb) break;
        for (int kx = 0; kx < a.kw; ++kx) {
            const int ix = ox * a.sw - a.pl + kx;
            if (ix >= a.in.w + a.pr) break;
            const bool inside = unsigned(iy) < unsigned(a.in.h) && unsigned(ix) < unsigned(a.in.w);
            if (inside || a.count_include_pad) ++cnt;
            if (!inside) continue;
            const u32x4 t = *reinterpret_cast<const u32x4*>(inb + int64_t(b) * a.in.sn + int64_t(iy) * a.in.sh + int64_t(ix) * a.in.sw + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x[4];
                unpack4_e4m3(t[q], x);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * q + e] = a.is_max ? fmaxf(acc[4 * q + e], x[e]) : acc[4 * q + e] + x[e];
            }
        }
    }
    // real = q * in_scale; out_q = real * out_qscale.  For max pooling the planner gives the output the input's scale: factor == 1.
    const float f = a.is_max ? a.in_scale * a.out_qscale : a.in_scale * a.out_qscale / float(cnt > 0 ? cnt : 1);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = pack4_e4m3(acc[4 * q] * f, acc[4 * q + 1] * f, acc[4 * q + 2] * f, acc[4 * q + 3] * f);
    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(a.out.p) + int64_t(b) * a.out.sn + int64_t(oy) * a.out.sh + int64_t(ox) * a.out.sw + c) = o;
}

hipError_t LaunchPoolF8(const PoolArgs& a, hipStream_t stream) {
    if (!a.in.f8 || !a.out.f8 || a.in.sc != 1 || a.out.sc != 1 || a.pre_scale != nullptr) return hipErrorInvalidValue;
    if ((a.in.c & 15) || (a.in.sw & 15) || (a.out.sw & 15) || (reinterpret_cast<uintptr_t>(a.in.p) & 15) || (reinterpret_cast<uintptr_t>(a.out.p) & 15) ||
        a.in.c != a.out.c)
        return hipErrorInvalidValue;
    const int64_t total = int64_t(a.out.n) * a.out.h * a.out.w * (a.out.c / 16);
    if (total == 0) return hipSuccess;
    const int64_t blocks = (total + 255) / 256;
    if (blocks >= (int64_t(1) << 31)) return hipErrorInvalidValue;
    pool_f8_kernel<<<dim3(unsigned(blocks)), dim3(256), 0, stream>>>(a, total);
    return hipGetLastError();
}

// global average pool of an e4m3 NHWC tensor into a half (or float) [N, C] vector in real units.  Block = 64 channels x 4 pixel groups.
__global__ __launch_bounds__(256) void gap_f8_kernel(const TensorArg in, const TensorArg out, const float in_scale) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int b = blockIdx.y;
    const int HW = in.h * in.w;
    const bool cok = c < in.c;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(in.p);
    float acc = 0.f;
    if (cok)
        for (int p = g; p < HW; p += 4) {
            const int y = p / in.w, x = p - y * in.w;
            const unsigned q = inb[int64_t(b) * in.sn + int64_t(y) * in.sh + int64_t(x) * in.sw + c];
            acc += __builtin_amdgcn_cvt_f32_fp8(int(q), 0);
        }
    red[g][cl] = acc;
    __syncthreads();
    if (g == 0 && cok) {
        const float tot = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) * in_scale / float(HW);
        if (out.f16) reinterpret_cast<_Float16*>(out.p)[int64_t(b) * out.sn + c] = _Float16(tot);
        else out.p[int64_t(b) * out.sn + c] = tot;
    }
}

hipError_t LaunchGlobalAvgPoolF8(const TensorArg& in, const TensorArg& out, float in_scale, hipStream_t stream) {
    if (!in.f8 || out.f8 || in.sc != 1 || out.sc != 1) return hipErrorInvalidValue;
    if (in.n == 0 || in.c == 0) return hipSuccess;
    gap_f8_kernel<<<dim3((in.c + 63) / 64, in.n), dim3(256), 0, stream>>>(in, out, in_scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// calibration: max |x| over a planned view (fp32 or half NHWC / NCHW), folded into *result with an integer atomic max on the
// float's bit pattern (non-negative floats order like unsigned integers).
// ------------------------------------------------------------------------------------------------------------------------
__global__ void absmax_kernel(const TensorArg t, const int64_t total, unsigned* __restrict__ result) {
    float m = 0.f;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int c = int(idx % t.c);
        int64_t p = idx / t.c;
        const int x = int(p % t.w); p /= t.w;
        const int y = int(p % t.h);
        const int b = int(p / t.h);
        const int64_t off = int64_t(b) * t.sn + int64_t(y) * t.sh + int64_t(x) * t.sw + int64_t(c) * t.sc;
        const float v = t.f16 ? float(reinterpret_cast<const _Float16*>(t.p)[off]) : t.p[off];
        m = fmaxf(m, fabsf(v));
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0 && m > 0.f && m == m) atomicMax(result, __float_as_uint(m));
}

hipError_t LaunchAbsMax(const TensorArg& t, float* result, hipStream_t stream) {
    if (t.f8) return hipErrorInvalidValue;
    const int64_t total = int64_t(t.n) * t.h * t.w * t.c;
    if (total == 0) return hipSuccess;
    const int64_t blocks = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
    absmax_kernel<<<dim3(unsigned(blocks)), dim3(256), 0, stream>>>(t, total, reinterpret_cast<unsigned*>(result));
    return hipGetLastError();
}

// test support: dst[i] = e4m3 round trip of src[i] * inv_scale, back in real units
__global__ void e4m3_roundtrip_kernel(const float* __restrict__ src, float* __restrict__ dst, unsigned char* __restrict__ codes, const float scale, const int64_t n) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned q = pack4_e4m3(src[i] / scale, 0.f, 0.f, 0.f) & 0xffu;
    if (codes) codes[i] = (unsigned char)q;
    dst[i] = __builtin_amdgcn_cvt_f32_fp8(int(q), 0) * scale;
}

hipError_t LaunchE4m3RoundTrip(const float* src, float* dst, void* codes, float scale, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    e4m3_roundtrip_kernel<<<dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream>>>(src, dst, static_cast<unsigned char*>(codes), scale, n);
    return hipGetLastError();
}

}  // namespace ie
