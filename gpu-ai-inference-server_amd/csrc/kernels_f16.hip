// fp16 precision mode (BASELINE configs[2-3]: "DenseNet-121 fp16 ... MFMA fp16 path"): NHWC half activations and half
// weights on v_mfma_f32_32x32x16_f16 with fp32 accumulation; folded-BN scale/shift, bias and the split-K slabs stay fp32.
//
// The kernel is the fp32 implicit GEMM (kernels.hip, conv_igemm_kernel) re-typed: the LDS image has the SAME byte layout
// ([rows][128 B data + 16 B pad], pitch/16 odd -> conflict-free ds_read_b128), so one 16-byte fragment read now carries 8
// halfs = the whole K=16 slice of one MFMA (lane l: k = 8*(l>>5) + j, j = 0..7) and a K-tile is 64 channels deep.
// At 16x the fp32 MFMA rate the contraction is no longer the bound: this path is HBM / operand-traffic bound.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "igemm_tiles.h"
#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int BM, int BN, int WM, int WN, int KG, bool PRE>
__global__ __launch_bounds__(64 * WM * WN * KG) void conv_igemm_f16_kernel(const ConvArgs a, const int tiles_n, const int num_tiles,
                                                                                  const int vec_store) {
    constexpr int NT = 64 * WM * WN;             // threads of one K-group
    constexpr int BKE = 64;                      // K-tile depth in halfs (128 B per row, as in the fp32 kernel)
    constexpr int LDP = BKE + 8;                 // row pitch in halfs (144 B)
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int GROUP_LDS = 2 * (BM + BN) * LDP;   // halfs of LDS per K-group
    static_assert(TM >= 1 && TN >= 1, "bad tile");
    static_assert(KG == 1 || BM * BN * 2 <= GROUP_LDS, "K-groups need room for the fp32 partial tile");

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave_all / (WM * WN);
    _Float16* const sA = smem_h + grp * GROUP_LDS;   // [2][BM][LDP]
    _Float16* const sB = sA + 2 * BM * LDP;          // [2][BN][LDP]

    const int tid = threadIdx.x - grp * NT;
    const int lane = tid & 63;
    const int wave = wave_all - grp * (WM * WN);
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
    const int nsplit = gridDim.y, split = blockIdx.y;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // one 16-byte fragment per operand per MFMA (K = 16); reads of step kk+1 are issued before the MFMAs of step kk
    auto compute = [&](int buf) {
        const _Float16* A = sA + buf * BM * LDP + (wm_i * TM * 32 + r) * LDP + hh * 8;
        const _Float16* B = sB + buf * BN * LDP + (wn_i * TN * 32 + r) * LDP + hh * 8;
        h8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const h8*>(A + i * 32 * LDP);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const h8*>(B + j * 32 * LDP);
#pragma unroll
        for (int kk = 0; kk < BKE / 16; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BKE / 16) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = *reinterpret_cast<const h8*>(A + i * 32 * LDP + (kk + 1) * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nxt][j] = *reinterpret_cast<const h8*>(B + j * 32 * LDP + (kk + 1) * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
        }
    };

    // ---- staging: thread owns the 16-byte chunk `c8` (8 channels) of rows {rw + i*ROWS_PER_PASS} ----
    constexpr int ROWS_PER_PASS = NT / 8;
    constexpr int A_IT = BM / ROWS_PER_PASS, B_IT = BN / ROWS_PER_PASS;
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the thread count");
    const int c8 = (tid & 7) * 8;
    const int rw = tid >> 3;
    const int cblocks = (Cin + BKE - 1) / BKE;
    const int KT = a.kh * a.kw * cblocks;
    const int kt_begin = int(int64_t(KT) * split / nsplit), kt_end = int(int64_t(KT) * (split + 1) / nsplit);
    const int ish = int(a.in.sh), isw = int(a.in.sw);
    constexpr unsigned OOB = 0x80000000u;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(a.in.p, 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w16), 0, Cout * Ktot * 2, 0x00020000);

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
        poff[i] = b * int(a.in.sn) + iy0 * ish + ix0 * isw + c8;
        unsigned msk = 0;
        for (int ky = 0; ky < a.kh; ++ky)
            for (int kx = 0; kx < a.kw; ++kx)
                if (unsigned(iy0 + ky) < unsigned(H) && unsigned(ix0 + kx) < unsigned(W)) msk |= 1u << (ky * a.kw + kx);
        taps[i] = mok ? msk : 0u;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) boff[i] = (n0 + rw + i * ROWS_PER_PASS) * Ktot + c8;

    f32x4 ra[A_IT], rb[B_IT];
    // BN scale/shift of the prologue as halfs (the half mirror of the blob): y = max(fma(x, s, t), 0) runs as packed half math
    // (v_pk_fma_f16 / v_pk_max_f16, one rounding, the result is stored to LDS as half anyway) - 8 VALU per 8 channels instead
    // of ~40 for convert / fp32 fma / max / convert, which otherwise out-costs the MFMAs of a K-tile.
    h8 s16 = {}, t16 = {};
    unsigned okmask = 0;
    const _Float16* const ps16 = static_cast<const _Float16*>(a.pre_scale16);
    const _Float16* const pt16 = static_cast<const _Float16*>(a.pre_shift16);
    auto issue_loads = [&](int kt) {
        const int tap = kt / cblocks;
        const int c0 = (kt - tap * cblocks) * BKE;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        const int tapoff = ky * ish + kx * isw + c0;
        const int woff = tap * Cin + c0;
        const bool cok = c0 + c8 < Cin;
        if constexpr (PRE) {
            const int cc = cok ? c0 + c8 : 0;
            s16 = *reinterpret_cast<const h8*>(ps16 + cc);
            t16 = *reinterpret_cast<const h8*>(pt16 + cc);
            okmask = 0;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bool ok = cok && ((taps[i] >> tap) & 1u);
            const unsigned off = ok ? unsigned(poff[i] + tapoff) * 2u : OOB;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
            if constexpr (PRE) okmask |= ok ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const unsigned off = cok ? unsigned(boff[i] + woff) * 2u : OOB;
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
        }
    };
    auto finish_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            f32x4 raw = ra[i];
            if constexpr (PRE) {
                h8 v = __builtin_bit_cast(h8, raw);
                v = v * s16 + t16;
                if (a.pre_relu) v = __builtin_elementwise_max(v, h8{});
                if (!(okmask & (1u << i))) v = h8{};          // zero padding applies AFTER the activation
                raw = __builtin_bit_cast(f32x4, v);
            }
            *reinterpret_cast<f32x4*>(sA + buf * BM * LDP + (rw + i * ROWS_PER_PASS) * LDP + c8) = raw;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            *reinterpret_cast<f32x4*>(sB + buf * BN * LDP + (rw + i * ROWS_PER_PASS) * LDP + c8) = rb[i];
    };

    const int nkt = kt_end - kt_begin;
    const int gb = kt_begin + int(int64_t(nkt) * grp / KG), ge = kt_begin + int(int64_t(nkt) * (grp + 1) / KG);
    const int rounds = (nkt + KG - 1) / KG;
    if (gb < ge) {
        issue_loads(gb);
        finish_store(0);
    }
    __syncthreads();
    for (int it = 0; it < rounds; ++it) {
        const int kt = gb + it;
        const int buf = it & 1;
        const bool active = kt < ge, more = kt + 1 < ge;
        if (more) issue_loads(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (active) compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        if (more) finish_store(buf ^ 1);
        __syncthreads();
    }
    if constexpr (KG > 1) {
        float* const myf = reinterpret_cast<float*>(sA);
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) myf[((i * TN + j) * 16 + e) * NT + tid] = acc[i][j][e];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const float* p = reinterpret_cast<const float*>(smem_h + g * GROUP_LDS);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] += p[((i * TN + j) * 16 + e) * NT + tid];
        }
    }

    // ---- epilogue: fp32 slab (split-K) or bias + ReLU and a half / float store ----
    // The MFMA is issued with the operands swapped (D = W x A^T: rows = output channels, columns = pixels), so a lane owns ONE
    // pixel (m = column r) and, per accumulator quad, FOUR consecutive channels n = 8g + 4hh + q: one 8-byte (half) or
    // 16-byte (float) store per quad instead of four 2-/4-byte ones; lanes l and l+32 fill adjacent quads of the same pixel.
    const bool partial = nsplit > 1;
    const int opitch = partial ? Cout : int(a.out.sw);
    float* const outf = partial ? a.workspace + int64_t(split) * M * Cout : a.out.p;
    _Float16* const outh = reinterpret_cast<_Float16*>(a.out.p);
    const bool store_half = !partial && a.out.f16;
    const bool do_relu = a.relu && !partial;
    const bool has_bias = !partial && a.bias != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm_i * TM + i) * 32 + r;
        const bool mok = m < M;
        const int64_t row = int64_t(m) * opitch;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + (wn_i * TN + j) * 32 + 8 * g + 4 * hh;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = acc[i][j][4 * g + q] + (has_bias ? a.bias[n + q < Cout ? n + q : Cout - 1] : 0.f);
                    if (do_relu) v[q] = fmaxf(v[q], 0.f);
                }
                if (!mok) continue;
                if (vec_store && n + 3 < Cout) {
                    if (store_half) {
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        h4 hv = {_Float16(v[0]), _Float16(v[1]), _Float16(v[2]), _Float16(v[3])};
                        *reinterpret_cast<h4*>(outh + row + n) = hv;
                    } else {
                        *reinterpret_cast<f32x4*>(outf + row + n) = f32x4{v[0], v[1], v[2], v[3]};
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < Cout) {
                            if (store_half) outh[row + n + q] = _Float16(v[q]);
                            else outf[row + n + q] = v[q];
                        }
                }
            }
        }
    }
}

template <int T>
static size_t f16_lds_bytes() {
    constexpr IgemmTile t = kIgemmTiles[T];
    return size_t(2) * (t.bm + t.bn) * 72 * sizeof(_Float16) * t.kg;
}

template <int T, bool PRE>
static hipError_t launch_f16_t(const ConvArgs& a, int splitk, hipStream_t stream) {
    constexpr IgemmTile t = kIgemmTiles[T];
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    const int tiles_m = int((M + t.bm - 1) / t.bm), tiles_n = (a.out.c + t.bn - 1) / t.bn;
    const int num_tiles = tiles_m * tiles_n;
    if (splitk > 1 && (a.workspace == nullptr || int64_t(splitk) * M * a.out.c > a.workspace_floats)) return hipErrorInvalidValue;
    // quad stores need 4-channel granularity and 8-/16-byte aligned quads in whichever buffer this launch writes
    int vec_store;
    if (splitk > 1) vec_store = (a.out.c % 4 == 0) && (reinterpret_cast<uintptr_t>(a.workspace) % 16 == 0);
    else if (a.out.f16) vec_store = (a.out.c % 4 == 0) && (a.out.sw % 4 == 0) && (reinterpret_cast<uintptr_t>(a.out.p) % 8 == 0);
    else vec_store = (a.out.c % 4 == 0) && (a.out.sw % 4 == 0) && (reinterpret_cast<uintptr_t>(a.out.p) % 16 == 0);
    conv_igemm_f16_kernel<t.bm, t.bn, t.wm, t.wn, t.kg, PRE>
        <<<dim3(num_tiles, splitk), dim3(64 * t.wm * t.wn * t.kg), f16_lds_bytes<T>(), stream>>>(a, tiles_n, num_tiles, vec_store);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splitk == 1) return e;
    return LaunchSplitKReduce(a, splitk, stream);
}

template <int T, bool PRE>
static hipError_t init_f16_t() {
    constexpr IgemmTile t = kIgemmTiles[T];
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f16_kernel<t.bm, t.bn, t.wm, t.wn, t.kg, PRE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t LaunchConvIgemmF16(const ConvArgs& a_in, int tile, int splitk, hipStream_t stream) {
    ConvArgs a = a_in;
    if (!a.in.f16 || a.w16 == nullptr || a.out.sc != 1 || a.in.sc != 1) return hipErrorInvalidValue;
    if ((a.in.c & 7) || (a.in.sw & 7) || (a.in.sh & 7) || (a.in.sn & 7) || (reinterpret_cast<uintptr_t>(a.in.p) & 15) ||
        (reinterpret_cast<uintptr_t>(a.w16) & 15) || a.kh * a.kw > 32)
        return hipErrorInvalidValue;
    if (a.pre_scale && (a.pre_scale16 == nullptr || a.pre_shift16 == nullptr || (reinterpret_cast<uintptr_t>(a.pre_scale16) & 15) ||
                        (reinterpret_cast<uintptr_t>(a.pre_shift16) & 15)))
        return hipErrorInvalidValue;
    a.in_bytes = 2 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + int64_t(a.in.c - 1) + 1);
    if (a.in_bytes >= (int64_t(1) << 31) || int64_t(a.out.c) * a.kh * a.kw * a.in.c * 2 >= (int64_t(1) << 31)) return hipErrorInvalidValue;
    if (int64_t(a.out.n) * a.out.h * a.out.w * a.out.sw >= (int64_t(1) << 31) || splitk < 1 || splitk > 64) return hipErrorInvalidValue;
#define IE_CASE(T) \
    case T: return a.pre_scale ? launch_f16_t<T, true>(a, splitk, stream) : launch_f16_t<T, false>(a, splitk, stream);
    switch (tile) {
        IE_CASE(0) IE_CASE(1) IE_CASE(2) IE_CASE(3) IE_CASE(4) IE_CASE(5) IE_CASE(6) IE_CASE(7) IE_CASE(8) IE_CASE(9) IE_CASE(10)
        default: return hipErrorInvalidValue;
    }
#undef IE_CASE
}

hipError_t InitKernelsF16() {
    hipError_t e;
#define IE_INIT(T)                                              \
    if ((e = init_f16_t<T, true>()) != hipSuccess) return e;    \
    if ((e = init_f16_t<T, false>()) != hipSuccess) return e;
    IE_INIT(0) IE_INIT(1) IE_INIT(2) IE_INIT(3) IE_INIT(4) IE_INIT(5) IE_INIT(6) IE_INIT(7) IE_INIT(8) IE_INIT(9) IE_INIT(10)
#undef IE_INIT
    return hipSuccess;
}

__global__ void convert_f32_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, const int64_t n) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = _Float16(src[i]);
}

hipError_t LaunchConvertF32ToF16(const float* src, void* dst, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int64_t blocks = (n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096;
    convert_f32_f16_kernel<<<dim3(unsigned(blocks)), dim3(256), 0, stream>>>(src, static_cast<_Float16*>(dst), n);
    return hipGetLastError();
}

__global__ void convert_u8_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, const int64_t n, const float scale,
                                      const float bias) {
#pragma clang fp contract(off)      // keep x * scale + bias as two rounded operations: bit-identical to the host-side expression
    // 16 source bytes per lane when aligned, scalar tail otherwise
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t nv = ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) ? n / 16 : 0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 o;
            o.x = float(w[k] & 0xffu) * scale + bias;
            o.y = float((w[k] >> 8) & 0xffu) * scale + bias;
            o.z = float((w[k] >> 16) & 0xffu) * scale + bias;
            o.w = float(w[k] >> 24) * scale + bias;
            reinterpret_cast<float4*>(dst)[i * 4 + k] = o;
        }
    }
    for (int64_t i = nv * 16 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = float(src[i]) * scale + bias;
}

hipError_t LaunchConvertU8ToF32(const void* src, float* dst, int64_t n, float scale, float bias, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int64_t work = (n + 15) / 16;
    const int64_t blocks = (work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192;
    convert_u8_f32_kernel<<<dim3(unsigned(blocks)), dim3(256), 0, stream>>>(static_cast<const unsigned char*>(src), dst, n, scale, bias);
    return hipGetLastError();
}

}  // namespace ie
