// A whole chain of DenseNet dense layers per launch, fp16 precision mode (BASELINE configs[2]/[3]).
//
// Where a dense block's feature map is small (14x14 / 7x7: DenseNet-121 blocks 3-4) the per-layer kernels are latency-bound: at batch 128
// block 3 was 48 launches of 12-26 us for 1-4 us of HBM time each (profiles/r02/steps_f16_b128.txt).  But a dense layer needs nothing from
// another IMAGE: BN -> ReLU -> 1x1 conv (K -> 128) -> BN -> ReLU -> 3x3 conv (128 -> 32) reads the image's own K channels and writes its own 32
// new ones.  So ONE workgroup owns ONE image and walks the layers of the block in a loop -- no grid-wide dependency, no launch boundary:
//
//   per layer:  1x1:  the image's pixel rows stream HBM/L2 -> registers (BN+ReLU prologue as packed half math) -> LDS in 64-channel chunks
//                     (double-buffered, one barrier per chunk); the 1x1 weights stream L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane)
//                     from a FRAGMENT-MAJOR mirror of the half weights, so a weight fragment is 1 KiB contiguous both in memory and in LDS
//                     (lane-linear: conflict-free ds_read_b128, no VGPR round trip); accumulators stay in registers over the whole K loop.
//               the bottleneck tensor T[pixels][128] NEVER goes to memory: bias + ReLU + half conversion write it into an LDS raster of the
//                     zero-padded image (pitch W + 1 with one shared pad column, as conv3x3_ws_f16_kernel) whose pad entries stay zero.
//               3x3:  nine shifted GEMMs straight out of that raster; all 72 weight fragments of the layer are brought in by LDS-DMA while
//                     the 1x1's epilogue runs; the 32 new channels go to the block buffer (16-byte stores), and the same workgroup reads
//                     them back as the tail of the next layers' K (workgroup barrier: same CU, same L1).
//
// D = W x A^T for both convs (a lane owns one pixel and quads of consecutive channels), v_mfma_f32_32x32x16_f16, fp32 accumulation.
// Arithmetic replaced: the dense-block part of Ort::Session::Run (inference_engine/src/model.cpp:1264-1270).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kTPitch = 136;      // halfs per raster row of T: 128 channels + 8 (272 B = 17 x 16 B: conflict-free ds_read_b128 over consecutive rows)
constexpr int kAPitch = 72;       // halfs per staged activation row: 64 channels + 8 (144 B = 9 x 16 B)
constexpr int kBBytes = 16384;    // one 64-deep K chunk of the 1x1 weights: 4 k-steps x 4 N-tiles x 1 KiB
constexpr int kW3Bytes = 73728;   // all 3x3 weights of a layer: 72 k-steps x 1 KiB
constexpr unsigned kOOB = 0x80000000u;
}  // namespace

// CFG 0: up to 8 position tiles per image (14x14 maps): 1x1 waves = 4 (M) x 2 (N), each 2 x 2 tiles of 32 x 32; the 3x3 weights alias the 1x1 staging.
// CFG 1: up to 2 position tiles per image (7x7 maps):  1x1 waves = 2 (M) x 4 (N), each 1 x 1 tile; the 3x3 weights have LDS of their own.
template <int CFG>
__global__ __launch_bounds__(512) void dense_block_f16_kernel(const DenseBlockArgs a) {
    constexpr int WM1 = CFG == 0 ? 4 : 2, TM1 = CFG == 0 ? 2 : 1, WN1 = 8 / WM1, TN1 = 4 / WN1;
    constexpr int PIT = CFG == 0 ? 4 : 1;                  // staged rows per thread and chunk (64 rows per pass of the 512 threads)
    constexpr bool kAlias = CFG == 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_blk[];

    const int H = a.h, W = a.w, PW = W + 1, NP = H * PW;
    const int ntiles = (NP + 31) >> 5, M1p = ntiles * 32;
    const int trows = NP + 2 * PW + 2;
    _Float16* const sT = reinterpret_cast<_Float16*>(smem_blk);                              // [trows][kTPitch]
    unsigned char* const sS = smem_blk + size_t(trows) * kTPitch * 2;                        // staging
    _Float16* const sB[2] = {reinterpret_cast<_Float16*>(sS), reinterpret_cast<_Float16*>(sS + kBBytes)};
    _Float16* const sA[2] = {reinterpret_cast<_Float16*>(sS + 2 * kBBytes), reinterpret_cast<_Float16*>(sS + 2 * kBBytes + size_t(M1p) * kAPitch * 2)};
    unsigned char* const sEnd = sS + 2 * kBBytes + size_t(2) * M1p * kAPitch * 2;
    _Float16* const sW3 = reinterpret_cast<_Float16*>(kAlias ? sS + kBBytes : sEnd);        // aliases sB[1] + sA[*] (CFG 0) or follows the staging
    float* const sBias = reinterpret_cast<float*>(kAlias ? sEnd : sEnd + kW3Bytes);          // [128] 1x1 bias, [32] 3x3 bias

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.x;
    const int pitch = a.pitch;
    _Float16* const ximg = a.x + size_t(img) * H * W * pitch;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(ximg, 0, H * W * pitch * 2, 0x00020000);

    // ---- zero the raster once: pad rows / columns are never written again ----
    {
        const int n16 = trows * kTPitch / 8;
        for (int q = tid; q < n16; q += 512) reinterpret_cast<u32x4*>(sT)[q] = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- staging geometry: thread -> (row l = tid / 8 + 64 i, 8 channels at c8) ----
    const int c8 = (tid & 7) * 8;
    int poff[PIT];                     // element offset of (pixel, c8) inside the image, or -1 (pad column, position past the image)
#pragma unroll
    for (int i = 0; i < PIT; ++i) {
        const int p = (tid >> 3) + 64 * i;
        const int y = p / PW, x = p - y * PW;
        poff[i] = (p < NP && x < W) ? (y * W + x) * pitch + a.in_coff + c8 : -1;
    }
    // ---- 1x1 wave tile and per-lane validity of its positions ----
    const int mp = wave % WM1, np = wave / WM1;
    bool valid1[TM1];
    int trow1[TM1];                    // raster row of T the lane's pixel goes to
#pragma unroll
    for (int i = 0; i < TM1; ++i) {
        const int p = (mp * TM1 + i) * 32 + r;
        const int y = p / PW, x = p - y * PW;
        valid1[i] = p < NP && x < W;
        trow1[i] = p + PW + 1;
    }
    // ---- 3x3: wave w owns position tile w ----
    const int p3 = wave * 32 + r;
    const int y3 = p3 / PW, x3 = p3 - y3 * PW;
    const bool valid3 = wave < ntiles && p3 < NP && x3 < W;
    const unsigned orow3 = valid3 ? unsigned((y3 * W + x3) * pitch) * 2u : kOOB;

    u32x4 av[PIT];
    u32x4 sv, tv;                      // prologue scale / shift of this thread's 8 channels
    auto issue_a = [&](const DenseBlockLayer& L, int c) {
        const int cb = c * 64 + c8;
        const bool cok = cb < L.K;
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const unsigned off = (poff[i] >= 0 && cok) ? unsigned(poff[i] + c * 64) * 2u : kOOB;
            av[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
        }
        if (L.ps != 0xffffffffu) {
            const int cc = cok ? cb : 0;
            sv = *reinterpret_cast<const u32x4*>(a.w16 + L.ps + cc);
            tv = *reinterpret_cast<const u32x4*>(a.w16 + L.pt + cc);
        }
    };
    auto issue_b = [&](const DenseBlockLayer& L, int c, int buf) {
        const int nblk = (L.K - c * 64 >= 64 ? 4 : 2) * 4;                      // 1 KiB fragment blocks of this chunk
        const _Float16* const src = a.wfrag16 + L.w1 + size_t(c) * 16 * 512;
        for (int q = wave; q < nblk; q += 8)
            __builtin_amdgcn_global_load_lds(src + q * 512 + lane * 8, sB[buf] + q * 512, 16, 0, 0);
    };
    auto commit_a = [&](const DenseBlockLayer& L, int buf) {
        const bool pre = L.ps != 0xffffffffu;
        const h8 s8 = __builtin_bit_cast(h8, sv), t8 = __builtin_bit_cast(h8, tv);
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int l = (tid >> 3) + 64 * i;
            h8 v = __builtin_bit_cast(h8, av[i]);
            if (pre) {
                v = v * s8 + t8;
                if (L.flags & 1) v = __builtin_elementwise_max(v, h8{});
            }
            if (l < M1p) *reinterpret_cast<h8*>(sA[buf] + l * kAPitch + c8) = v;
        }
    };

    f32x16 acc[TM1][TN1];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM1; ++i)
#pragma unroll
            for (int j = 0; j < TN1; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    auto compute1 = [&](int buf, int nks) {
        const _Float16* const Ab = sA[buf] + (mp * TM1 * 32 + r) * kAPitch + hh * 8;
        const _Float16* const Bb = sB[buf] + (np * TN1) * 512 + lane * 8;
        for (int s = 0; s < nks; ++s) {
            h8 af[TM1], bf[TN1];
#pragma unroll
            for (int i = 0; i < TM1; ++i) {            // a tile past the image (7 tiles over 4 x 2) has no staged rows: feed zeros
                af[i] = h8{};
                if (mp * TM1 + i < ntiles) af[i] = *reinterpret_cast<const h8*>(Ab + i * 32 * kAPitch + s * 16);
            }
#pragma unroll
            for (int j = 0; j < TN1; ++j) bf[j] = *reinterpret_cast<const h8*>(Bb + (s * 4 + j) * 512);
#pragma unroll
            for (int i = 0; i < TM1; ++i)
#pragma unroll
                for (int j = 0; j < TN1; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    const DenseBlockLayer& L0 = a.layer[0];
    issue_a(L0, 0);
    issue_b(L0, 0, 0);
    __syncthreads();                                       // raster zeroed before anybody writes T

    for (int li = 0; li < a.nlayers; ++li) {
        const DenseBlockLayer& L = a.layer[li];
        const int K = L.K, NC = (K + 63) >> 6;
        if (tid < 128) sBias[tid] = L.b1 != 0xffffffffu ? a.w32[L.b1 + tid] : 0.f;
        else if (tid < 160) sBias[tid] = L.b3 != 0xffffffffu ? a.w32[L.b3 + tid - 128] : 0.f;
        zero_acc();
        // ---------------- 1x1: K -> 128 over all positions of the image ----------------
        for (int c = 0; c < NC; ++c) {
            const int buf = c & 1;
            commit_a(L, buf);
            __syncthreads();                               // chunk c staged (activations written, weight DMA landed: the barrier drains vmcnt)
            if (c + 1 < NC) {
                issue_a(L, c + 1);
                issue_b(L, c + 1, buf ^ 1);
            }
            if (mp * TM1 < ntiles) compute1(buf, K - c * 64 >= 64 ? 4 : 2);
        }
        __syncthreads();                                   // every wave is done with the staging buffers
        // ---------------- 3x3 weights by LDS-DMA while the 1x1 epilogue runs ----------------
        {
            const _Float16* const src = a.wfrag16 + L.w3;
            for (int q = wave; q < 72; q += 8) __builtin_amdgcn_global_load_lds(src + q * 512 + lane * 8, sW3 + q * 512, 16, 0, 0);
        }
        // ---------------- 1x1 epilogue: bias + ReLU -> half -> T raster (valid positions only) ----------------
#pragma unroll
        for (int i = 0; i < TM1; ++i)
#pragma unroll
            for (int j = 0; j < TN1; ++j) {
                const int nt = np * TN1 + j;
                float v[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + nt * 32 + 8 * g + 4 * hh);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = acc[i][j][4 * g + q] + bq[q];
                        v[4 * g + q] = (L.flags & 2) ? fmaxf(x, 0.f) : x;
                    }
                }
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const h4 qa = {_Float16(v[8 * gp + 0]), _Float16(v[8 * gp + 1]), _Float16(v[8 * gp + 2]), _Float16(v[8 * gp + 3])};
                    const h4 qb = {_Float16(v[8 * gp + 4]), _Float16(v[8 * gp + 5]), _Float16(v[8 * gp + 6]), _Float16(v[8 * gp + 7])};
                    const u32x2 xa = __builtin_bit_cast(u32x2, qa), xb = __builtin_bit_cast(u32x2, qb);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(xa[0], xb[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(xa[1], xb[1], false, false);
                    if (valid1[i]) *reinterpret_cast<u32x4*>(sT + trow1[i] * kTPitch + nt * 32 + 8 * (2 * gp + hh)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            }
        __syncthreads();                                   // T complete, 3x3 weights landed
        // ---------------- next layer's first chunk on its way during the 3x3 ----------------
        if (li + 1 < a.nlayers) {
            issue_a(a.layer[li + 1], 0);
            issue_b(a.layer[li + 1], 0, 0);
        }
        // ---------------- 3x3: nine shifted GEMMs out of the raster, position tile = wave ----------------
        if (wave < ntiles) {
            f32x16 acc3;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc3[e] = 0.f;
            const _Float16* const Tb = sT + (wave * 32 + r) * kTPitch + hh * 8;
            const _Float16* const Wb = sW3 + lane * 8;
            h8 af[2], bf[2];
            auto rd = [&](int st, int slot) {
                const int tap = st >> 3, kk = st & 7;
                const int shift = (tap / 3) * PW + (tap % 3);
                af[slot] = *reinterpret_cast<const h8*>(Tb + shift * kTPitch + kk * 16);
                bf[slot] = *reinterpret_cast<const h8*>(Wb + st * 512);
            };
            rd(0, 0);
#pragma unroll 8
            for (int st = 0; st < 72; ++st) {
                const int cur = st & 1;
                if (st + 1 < 72) rd(st + 1, cur ^ 1);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[cur], af[cur], acc3, 0, 0, 0);
            }
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + 128 + 8 * g + 4 * hh);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = acc3[4 * g + q] + bq[q];
                    v[4 * g + q] = (L.flags & 4) ? fmaxf(x, 0.f) : x;
                }
            }
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const h4 qa = {_Float16(v[8 * gp + 0]), _Float16(v[8 * gp + 1]), _Float16(v[8 * gp + 2]), _Float16(v[8 * gp + 3])};
                const h4 qb = {_Float16(v[8 * gp + 4]), _Float16(v[8 * gp + 5]), _Float16(v[8 * gp + 6]), _Float16(v[8 * gp + 7])};
                const u32x2 xa = __builtin_bit_cast(u32x2, qa), xb = __builtin_bit_cast(u32x2, qb);
                const auto s0 = __builtin_amdgcn_permlane32_swap(xa[0], xb[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(xa[1], xb[1], false, false);
                const unsigned off = valid3 ? orow3 + unsigned(L.out_coff + 8 * (2 * gp + hh)) * 2u : kOOB;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rs_x, off, 0, 0);
            }
        }
        __syncthreads();                                   // T and the 3x3 weights are free; the new channels are visible to the whole workgroup
    }
}

static size_t block_lds_bytes(int cfg, int H, int W) {
    const int PW = W + 1, NP = H * PW, M1p = (NP + 31) / 32 * 32, trows = NP + 2 * PW + 2;
    return size_t(trows) * kTPitch * 2 + 2 * kBBytes + size_t(2) * M1p * kAPitch * 2 + (cfg == 0 ? 0 : kW3Bytes) + 160 * sizeof(float);
}

static int block_cfg(int H, int W) {
    const int ntiles = (H * (W + 1) + 31) / 32;
    return ntiles <= 2 ? 1 : (ntiles <= 8 ? 0 : -1);
}

bool DenseBlockEligible(const DenseBlockArgs& a) {
    if (a.x == nullptr || a.wfrag16 == nullptr || a.w16 == nullptr || a.w32 == nullptr) return false;
    if (a.nlayers < 1 || a.nlayers > kMaxBlockLayers || a.n < 1 || a.h < 1 || a.w < 1) return false;
    const int cfg = block_cfg(a.h, a.w);
    if (cfg < 0 || block_lds_bytes(cfg, a.h, a.w) > size_t(160) * 1024) return false;
    if (cfg == 0) {        // the 3x3 weights alias the staging area behind the first weight buffer
        const int M1p = (a.h * (a.w + 1) + 31) / 32 * 32;
        if (size_t(kBBytes) + size_t(2) * M1p * kAPitch * 2 < size_t(kW3Bytes)) return false;
    }
    if ((a.pitch & 7) || (a.in_coff & 7) || (reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.wfrag16) & 15) ||
        (reinterpret_cast<uintptr_t>(a.w16) & 15))
        return false;
    if (int64_t(a.h) * a.w * a.pitch * 2 >= (int64_t(1) << 31)) return false;
    for (int l = 0; l < a.nlayers; ++l) {
        const DenseBlockLayer& L = a.layer[l];
        if (L.K < 64 || (L.K & 31) || a.in_coff + L.K > a.pitch) return false;
        if ((L.out_coff & 7) || L.out_coff + 32 > a.pitch) return false;
        if (L.out_coff < a.in_coff + L.K && L.out_coff + 32 > a.in_coff) return false;          // the new channels must not overlap what the layer reads
        if ((L.w1 & 7) || (L.w3 & 7)) return false;
        if (L.ps != 0xffffffffu && ((L.ps & 7) || (L.pt & 7) || L.pt == 0xffffffffu)) return false;
        // the first chunk of a layer is requested while the previous layer's 3x3 is still running: it must not contain that layer's output
        if (l > 0 && a.layer[l - 1].out_coff < a.in_coff + 64 && a.layer[l - 1].out_coff + 32 > a.in_coff) return false;
    }
    return true;
}

hipError_t LaunchDenseBlockF16(const DenseBlockArgs& a, hipStream_t stream) {
    if (!DenseBlockEligible(a)) return hipErrorInvalidValue;
    const int cfg = block_cfg(a.h, a.w);
    const size_t lds = block_lds_bytes(cfg, a.h, a.w);
    if (cfg == 0) dense_block_f16_kernel<0><<<dim3(a.n), dim3(512), lds, stream>>>(a);
    else dense_block_f16_kernel<1><<<dim3(a.n), dim3(512), lds, stream>>>(a);
    return hipGetLastError();
}

// dst = src ([rows][K] halfs, K % 16 == 0, rows % 32 == 0) in MFMA-fragment order: 1 KiB block (s, j) = k-step s of 16, row tile j of 32, at
// block index s * (rows / 32) + j; inside it lane l holds row 32 j + (l & 31), k = 16 s + 8 (l >> 5) .. + 7 (the operand map of
// v_mfma_f32_32x32x16_f16).  A K chunk of all row tiles, or a whole [32][K] matrix, is then one contiguous byte range.
__global__ void permute_frag16_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst, int rows, int K) {
    const int NJ = rows >> 5, nblk = NJ * (K >> 4);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int blk = gid >> 6, l = gid & 63;
    if (blk >= nblk) return;
    const int s = blk / NJ, j = blk - s * NJ;
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + size_t(j * 32 + (l & 31)) * K + 16 * s + 8 * (l >> 5));
    *reinterpret_cast<u32x4*>(dst + size_t(gid) * 8) = v;
}

hipError_t LaunchPermuteWeightsFrag16(const void* src, void* dst, int rows, int K, hipStream_t stream) {
    if ((rows & 31) || (K & 15) || rows <= 0 || K <= 0) return hipErrorInvalidValue;
    const int threads = (rows >> 5) * (K >> 4) * 64;
    permute_frag16_kernel<<<dim3((threads + 255) / 256), dim3(256), 0, stream>>>(static_cast<const _Float16*>(src), static_cast<_Float16*>(dst), rows, K);
    return hipGetLastError();
}

hipError_t InitKernelsBlock() {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_block_f16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_block_f16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    return hipSuccess;
}

}  // namespace ie
