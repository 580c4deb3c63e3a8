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

#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// s_setprio(1) for the producer waves, A/B'd in the probe: 355 vs 323 us per 24-layer chain at batch 128 (the raised priority also slows the
// consumers' epilogue and 3x3, where the producers have nothing to hide): off.
#ifndef IE_BLOCK_PRIO
#define IE_BLOCK_PRIO 0
#endif

namespace {
constexpr int kTPitch = 136;      // halfs per raster row of T: 128 channels + 8 (272 B = 17 x 16 B: conflict-free ds_read_b128 over consecutive rows)
constexpr int kAPitch = 72;       // halfs per staged activation row: 64 channels + 8 (144 B = 9 x 16 B)
constexpr int kW3Bytes = 73728;   // all 3x3 weights of a layer: 72 k-steps x 1 KiB
constexpr int kMaxK = 2048;       // most input channels of a 1x1: its prologue scale + shift (2 K halfs) are staged by 512 threads x 8 halfs
constexpr unsigned kOOB = 0x80000000u;
}  // namespace

// max(v, lo) per half without the canonicalising second v_pk_max_f16 that __builtin_elementwise_max emits (the inputs are results of an fma:
// already canonical)
__device__ __forceinline__ h8 pk_max8(h8 v, h8 lo) {
    u32x4 x = __builtin_bit_cast(u32x4, v);
    const u32x4 l = __builtin_bit_cast(u32x4, lo);
#pragma unroll
    for (int k = 0; k < 4; ++k) asm("v_pk_max_f16 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(l[k]));
    return __builtin_bit_cast(h8, x);
}

// CFG 0: up to 7 position tiles per image (14x14 maps: 210 raster positions); CFG 1: up to 2 (7x7 maps).
//
// Wave specialisation (the third build; what the first two measured is in DESIGN.md): with all eight waves doing everything -- wait for loads,
// prologue, ds_write, issue, barrier, MFMAs -- a 64-channel chunk cost 3.2 k cycles for 1.0 k cycles of MFMAs, because the barrier puts every wave
// in the same phase and the phases add up.  Now
//   * waves 4-7 (PRODUCERS) move data: chunk c + 1's pixel rows registers -> (BN+ReLU as packed half math) -> LDS and chunk c + 3's loads, the
//     layer's 3x3 weights (held in registers across the 1x1 loop, written to LDS behind it), the next layer's constants;
//   * waves 0-3 (CONSUMERS) only multiply: wave w owns the 32 bottleneck channels of N tile w for ALL position tiles (7 accumulators), takes its
//     weight fragments straight from L2 into a register ring (1 KiB contiguous each in the fragment-major mirror, fetched by exactly one wave)
//     and its activation fragments from the LDS chunk the producers finished one barrier earlier;
//   * one barrier per chunk; a producer wave shares its SIMD with a consumer wave, so its VALU / LDS-write / load-issue slots fill the gaps of the
//     consumer's MFMA stream instead of standing in front of them.
// Every load of a ring is a buffer load issued UNCONDITIONALLY (a chunk index past the layer's end becomes out-of-range offsets, which return
// zeros without touching memory): the queue looks the same on every path, so the compiler's vmcnt counts stay exact -- one conditional issue and
// every wait in the loop collapses to "all outstanding loads".
template <int CFG>
__global__ __launch_bounds__(512) void dense_block_f16_kernel(const DenseBlockArgs a) {
    constexpr bool BAND = CFG == 2;                        // a workgroup owns a BAND of rows of one image (one layer per launch, the bottleneck's halo rows recomputed)
    constexpr int TM = CFG == 0 ? 7 : (CFG == 1 ? 2 : 8);  // staged position tiles at most
    constexpr int PITP = TM;                               // 16-byte pieces per producer thread and chunk (rows tp / 8 + 32 i)
    constexpr int T3 = CFG == 1 ? 1 : 2;                   // 3x3: position tiles per consumer wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_blk[];

    // Raster geometry.  Whole-image mode: the H image rows are staged, T holds them between one zero row above and below (a staged position p
    // sits at raster row p + PW + 1).  Band mode: rows y0 - 1 .. y0 + R of the image are staged -- the halo rows too, their bottleneck values are
    // recomputed by this workgroup; rows outside the image stay zero -- and p sits at raster row p + 1.  Either way output position q (row q / PW
    // of the band, column q % PW) reads raster rows q + ky * PW + kx.
    const int H = a.h, W = a.w, PW = W + 1;
    const int R = BAND ? a.band_rows : H;
    const int nbands = BAND ? (H + R - 1) / R : 1;
    const int img = BAND ? int(blockIdx.x) / nbands : int(blockIdx.x);
    const int y0 = BAND ? (int(blockIdx.x) - img * nbands) * R : 0;          // first output row of this workgroup
    const int ys = BAND ? y0 - 1 : 0;                                        // image row of staged row 0
    const int NP = (BAND ? R + 2 : H) * PW;                                  // staged positions (1x1)
    const int Rb = H - y0 < R ? H - y0 : R;                                  // output rows that exist
    const int NP3 = Rb * PW;                                                 // output positions (3x3)
    const int ntiles = (NP + 31) >> 5, ntiles3 = (NP3 + 31) >> 5;
    const int toff = BAND ? 1 : PW + 1;
    const int trows = BAND ? NP + 2 : NP + 2 * PW + 2;
    _Float16* const sT = reinterpret_cast<_Float16*>(smem_blk);                              // [trows][kTPitch]
    unsigned char* const sS = smem_blk + size_t(trows) * kTPitch * 2;                        // staging: two activation chunks; then all 3x3 weights
    _Float16* const sA0 = reinterpret_cast<_Float16*>(sS);
    _Float16* const sA1 = reinterpret_cast<_Float16*>(sS + size_t(32 * TM) * kAPitch * 2);      // (always 32 x TM rows: no per-row guards)
    _Float16* const sW3 = reinterpret_cast<_Float16*>(sS);
    const size_t stage_bytes = size_t(kW3Bytes);                                          // >= 2 x 224 x 144 B
    // per-layer constants, double-buffered by layer parity (the next layer's are staged while this layer's 3x3 runs)
    float* const sBias0 = reinterpret_cast<float*>(sS + stage_bytes);                        // [2][160]: [128] 1x1 bias, [32] 3x3 bias
    _Float16* const sPre0 = reinterpret_cast<_Float16*>(sBias0 + 2 * 160);                   // [2][2][kMaxK]: prologue scale, shift

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pitch = a.pitch;
    _Float16* const ximg = a.x + size_t(img) * H * W * pitch;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(ximg, 0, H * W * pitch * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wf = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wfrag16), 0, int(a.w16_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w16), 0, int(a.w16_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w32 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w32), 0, int(a.w16_bytes * 2 < 0x7fffffffull ? a.w16_bytes * 2 : 0x7fffffffull), 0x00020000);

    // ---- zero the raster once: pad rows / columns are never written again ----
    {
        const int n16 = trows * kTPitch / 8;
        for (int q = tid; q < n16; q += 512) reinterpret_cast<u32x4*>(sT)[q] = u32x4{0u, 0u, 0u, 0u};
    }
    const int nl = a.nlayers;

    if (wave >= 4) {
        // =========================================== PRODUCERS ===========================================
        if (IE_BLOCK_PRIO) __builtin_amdgcn_s_setprio(1);  // data movers first: their VALU / LDS / load-issue slots fill the gaps of the consumers' MFMA stream
        const int tp = tid - 256;
        const int c8 = (tp & 7) * 8;
        int poff[PITP];                // element offset of (pixel, c8) inside the image, or -1 (pad column, position past the image)
#pragma unroll
        for (int i = 0; i < PITP; ++i) {
            const int p = (tp >> 3) + 32 * i;
            const int y = p / PW, x = p - y * PW, iy = ys + y;
            poff[i] = (p < NP && x < W && iy >= 0 && iy < H) ? (iy * W + x) * pitch + a.in_coff + c8 : -1;
        }
        struct ASlot { u32x4 v[PITP]; };
        ASlot ra[3];
        u32x4 w3r[18];                 // this thread's 288 bytes of the layer's 3x3 weights, held across the 1x1 loop
        u32x4 cpre[2];
        float cbias = 0.f;
        auto issue = [&](ASlot& A, const DenseBlockLayer& L, int c) {
            const bool cok = c * 64 + c8 < L.K;
#pragma unroll
            for (int i = 0; i < PITP; ++i)
                A.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (poff[i] >= 0 && cok) ? unsigned(poff[i] + c * 64) * 2u : kOOB, 0, 0);
        };
        auto issue_w3 = [&](const DenseBlockLayer& L) {
#pragma unroll
            for (int q = 0; q < 18; ++q) w3r[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_wf, (L.w3 + unsigned(q * 256 + tp) * 8u) * 2u, 0, 0);
        };
        // constants of a layer: two 16-byte pieces of (scale | shift) and one bias float per thread
        auto fetch_consts = [&](const DenseBlockLayer& L) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tp + 256 * j, half = q * 8 < L.K ? 0 : 1, off = q * 8 - half * L.K;
                const bool ok = L.ps != 0xffffffffu && off < L.K;
                cpre[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w16, ok ? ((half ? L.pt : L.ps) + unsigned(off)) * 2u : kOOB, 0, 0);
            }
            const unsigned bo = tp < 128 ? L.b1 : L.b3;
            cbias = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w32, (tp < 160 && bo != 0xffffffffu) ? (bo + unsigned(tp < 128 ? tp : tp - 128)) * 4u : kOOB, 0, 0));
        };
        auto store_consts = [&](const DenseBlockLayer& L, int par) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tp + 256 * j, half = q * 8 < L.K ? 0 : 1, off = q * 8 - half * L.K;
                if (off < L.K) {
                    // no prologue: scale 1 (0x3c00), shift 0 -- the commit loop is branch-free
                    const u32x4 ident = half ? u32x4{0u, 0u, 0u, 0u} : u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
                    *reinterpret_cast<u32x4*>(sPre0 + (par * 2 + half) * kMaxK + off) = L.ps != 0xffffffffu ? cpre[j] : ident;
                }
            }
            if (tp < 160) sBias0[par * 160 + tp] = cbias;
        };
        // Branch-free: a layer without a prologue has scale 1 / shift 0 staged (store_consts), ReLU is a select on a wave-uniform flag folded into
        // the clamp value, every thread's rows exist in the staging buffer (it always holds 32 x PITP rows).
        auto commit = [&](const ASlot& A, const DenseBlockLayer& L, const _Float16* sPre, int c) {
            _Float16* const dst = (c & 1) ? sA1 : sA0;
            const int cb = c * 64 + c8 < L.K ? c * 64 + c8 : 0;
            const h8 s8 = *reinterpret_cast<const h8*>(sPre + cb), t8 = *reinterpret_cast<const h8*>(sPre + kMaxK + cb);
            const _Float16 lo = (L.flags & 1) ? _Float16(0.f) : _Float16(-65504.f);
            const h8 lo8 = {lo, lo, lo, lo, lo, lo, lo, lo};
#pragma unroll
            for (int i = 0; i < PITP; ++i) {
                const int l = (tp >> 3) + 32 * i;
                h8 v = __builtin_bit_cast(h8, A.v[i]) * s8 + t8;
                v = pk_max8(v, lo8);
                *reinterpret_cast<h8*>(dst + l * kAPitch + c8) = v;
            }
        };
        const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && wave == 4;
        long long tp_[6] = {0, 0, 0, 0, 0, 0};
        fetch_consts(a.layer[0]);
        issue(ra[0], a.layer[0], 0);
        issue(ra[1], a.layer[0], 1);
        issue(ra[2], a.layer[0], 2);
        issue_w3(a.layer[0]);
        store_consts(a.layer[0], 0);
        __syncthreads();                                   // raster zeroed, first constants in place
        for (int li = 0; li < nl; ++li) {
            const DenseBlockLayer& L = a.layer[li];
            const int NC = (L.K + 63) >> 6;
            const _Float16* const sPre = sPre0 + (li & 1) * 2 * kMaxK;
            auto step = [&](ASlot& A, int c) {
                long long u0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
                commit(A, L, sPre, c);                     // waits for chunk c's loads only
                if (stamp) { const long long u1 = __builtin_amdgcn_s_memtime(); tp_[0] += u1 - u0; u0 = u1; }
                issue(A, L, c + 3);
                if (stamp) { const long long u1 = __builtin_amdgcn_s_memtime(); tp_[1] += u1 - u0; u0 = u1; }
                __syncthreads();                           // chunk c staged (the consumers are past compute(c - 2): this buffer was free)
                if (stamp) { const long long u1 = __builtin_amdgcn_s_memtime(); tp_[2] += u1 - u0; }
            };
            long long v0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
            int c0 = 0;
            for (; c0 + 3 <= NC; c0 += 3) {
                step(ra[0], c0);
                step(ra[1], c0 + 1);
                step(ra[2], c0 + 2);
            }
            if (c0 < NC) step(ra[0], c0);
            if (c0 + 1 < NC) step(ra[1], c0 + 1);
            __syncthreads();                               // the consumers are done with the last chunk: the staging area is free
            if (stamp) { const long long v1 = __builtin_amdgcn_s_memtime(); tp_[3] += v1 - v0; v0 = v1; }
#pragma unroll
            for (int q = 0; q < 18; ++q) *reinterpret_cast<u32x4*>(sW3 + (q * 256 + tp) * 8) = w3r[q];
            __syncthreads();                               // 3x3 weights in place (and the consumers' T raster)
            if (stamp) { const long long v1 = __builtin_amdgcn_s_memtime(); tp_[4] += v1 - v0; v0 = v1; }
            {       // the next layer's first chunks, 3x3 weights and constants on their way while the consumers run the 3x3
                    // (behind the last layer the first one's are requested again and never used: the queue must look the same on every path)
                const DenseBlockLayer& Ln = a.layer[li + 1 < nl ? li + 1 : 0];
                fetch_consts(Ln);
                issue(ra[0], Ln, 0);
                issue(ra[1], Ln, 1);
                issue(ra[2], Ln, 2);
                issue_w3(Ln);
                store_consts(Ln, (li + 1) & 1);
            }
            __syncthreads();                               // layer done: T and the 3x3 weights are free, the new channels are visible
            if (stamp) { const long long v1 = __builtin_amdgcn_s_memtime(); tp_[5] += v1 - v0; }
        }
        if (stamp && lane == 0)
            for (int q = 0; q < 6; ++q) a.dbg[8 + q] = tp_[q];
    } else {
        // =========================================== CONSUMERS ===========================================
        const int np = wave;                               // N tile of the 1x1: bottleneck channels 32 np .. 32 np + 31
        unsigned vmask = 0;                                // bit i: this lane's position of tile i is a pixel of the image
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int p = i * 32 + r;
            const int y = p / PW, x = p - y * PW, iy = ys + y;
            if (p < NP && x < W && iy >= 0 && iy < H) vmask |= 1u << i;
        }
        unsigned orow3[T3];                                // 3x3: byte offset of the lane's output pixel row, or out of range
#pragma unroll
        for (int j = 0; j < T3; ++j) {
            const int t = (CFG != 1 ? wave * T3 : wave) + j;
            const int p = t * 32 + r;
            const int y = p / PW, x = p - y * PW;
            orow3[j] = (t < ntiles3 && (CFG != 1 || wave < 2) && p < NP3 && x < W) ? unsigned(((y0 + y) * W + x) * pitch) * 2u : kOOB;
        }
        struct BSlot { u32x4 f[4]; };
        BSlot bs[3];
        auto issue_b = [&](BSlot& Bq, const DenseBlockLayer& L, int c) {
            const int nks = (L.K - c * 64) >> 4;           // <= 0 past the layer; 2 in the last chunk of a K % 64 == 32 layer
            const unsigned base = (L.w1 + (unsigned(c) * 16u + unsigned(np)) * 512u + unsigned(lane) * 8u) * 2u;
#pragma unroll
            for (int s = 0; s < 4; ++s) Bq.f[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_wf, s < nks ? base + unsigned(s) * 4096u : kOOB, 0, 0);
        };
        f32x16 acc[TM];
        // One chunk: TM position tiles x NKS k-steps as ONE straight line (no runtime tile count inside: a branch per tile made the compiler
        // serialise read -> wait -> MFMA, 2.2 k cycles per chunk for 0.9 k of MFMAs; tiles past the image multiply staging rows nobody reads
        // back).  Activation fragments come through a ring of four registers, read three MFMAs ahead of their use.
        auto compute_n = [&](const BSlot& Bq, int c, auto nks_c) {
            constexpr int NKS = decltype(nks_c)::value, NST = NKS * TM, AHEAD = 3;
            const _Float16* const Ab = ((c & 1) ? sA1 : sA0) + r * kAPitch + hh * 8;
            h8 af[4];
#pragma unroll
            for (int q = 0; q < AHEAD && q < NST; ++q) af[q & 3] = *reinterpret_cast<const h8*>(Ab + (q % TM) * 32 * kAPitch + (q / TM) * 16);
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD < NST ? AHEAD : NST, 0);
#pragma unroll
            for (int q = 0; q < NST; ++q) {
                if (q + AHEAD < NST) af[(q + AHEAD) & 3] = *reinterpret_cast<const h8*>(Ab + ((q + AHEAD) % TM) * 32 * kAPitch + ((q + AHEAD) / TM) * 16);
                acc[q % TM] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, Bq.f[q / TM]), af[q & 3], acc[q % TM], 0, 0, 0);
                if (q + AHEAD < NST) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        };
        // Always four k-steps: in the last chunk of a K % 64 == 32 layer the weight fragments of k-steps 2, 3 are out-of-range loads (zeros) and the
        // staged columns past K hold finite prologue values of zeros, so the two extra steps add exact zeros (two code paths here made the
        // register allocator spill 600 VGPRs).
        auto compute1 = [&](const BSlot& Bq, int c) { compute_n(Bq, c, std::integral_constant<int, 4>{}); };
        const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && wave == 0;
        long long tc_[6] = {0, 0, 0, 0, 0, 0};
        issue_b(bs[0], a.layer[0], 0);
        issue_b(bs[1], a.layer[0], 1);
        issue_b(bs[2], a.layer[0], 2);
        __syncthreads();                                   // raster zeroed, first constants in place
        for (int li = 0; li < nl; ++li) {
            const DenseBlockLayer& L = a.layer[li];
            const int K = L.K, NC = (K + 63) >> 6;
            long long v0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
            const float* const sBias = sBias0 + (li & 1) * 160;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            // ---------------- 1x1: K -> 128 (this wave: 32 of them) over all positions; chunk c lives in ring slot c % 3 and LDS buffer c & 1 ----------------
            auto step = [&](BSlot& Bq, int c) {
                long long u0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
                __syncthreads();                           // chunk c staged by the producers
                if (stamp) { const long long u1 = __builtin_amdgcn_s_memtime(); tc_[0] += u1 - u0; u0 = u1; }
                compute1(Bq, c);
                issue_b(Bq, L, c + 3);
                if (stamp) { const long long u1 = __builtin_amdgcn_s_memtime(); tc_[1] += u1 - u0; }
            };
            int c0 = 0;
            for (; c0 + 3 <= NC; c0 += 3) {
                step(bs[0], c0);
                step(bs[1], c0 + 1);
                step(bs[2], c0 + 2);
            }
            if (c0 < NC) step(bs[0], c0);
            if (c0 + 1 < NC) step(bs[1], c0 + 1);
            __syncthreads();                               // done with the staging area: the producers put the 3x3 weights there
            if (stamp) { const long long v1 = __builtin_amdgcn_s_memtime(); tc_[2] += v1 - v0; v0 = v1; }
            // ---------------- 1x1 epilogue: bias + ReLU -> half -> T raster (pixels only; pad entries stay zero) ----------------
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (i < ntiles) {
                    float v[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + np * 32 + 8 * g + 4 * hh);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x = acc[i][4 * g + q] + bq[q];
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
                        if ((vmask >> i) & 1u)
                            *reinterpret_cast<u32x4*>(sT + (i * 32 + r + toff) * kTPitch + np * 32 + 8 * (2 * gp + hh)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    }
                }
            }
            __syncthreads();                               // T complete, 3x3 weights in place
            if (stamp) { const long long v1 = __builtin_amdgcn_s_memtime(); tc_[3] += v1 - v0; v0 = v1; }
            {       // next layer's first weight chunks on their way during the 3x3 (same rule as the producers' prefetch)
                const DenseBlockLayer& Ln = a.layer[li + 1 < nl ? li + 1 : 0];
                issue_b(bs[0], Ln, 0);
                issue_b(bs[1], Ln, 1);
                issue_b(bs[2], Ln, 2);
            }
            // ---------------- 3x3: nine shifted GEMMs out of the raster; CFG 0: wave w owns tiles 2w, 2w + 1 (one weight fragment feeds both) ----------------
            const int t0 = CFG != 1 ? wave * T3 : wave;
            if (t0 < ntiles3 && (CFG != 1 || wave < 2)) {
                f32x16 acc3[T3];
#pragma unroll
                for (int j = 0; j < T3; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc3[j][e] = 0.f;
                const _Float16* const Tb = sT + (t0 * 32 + r) * kTPitch + hh * 8;
                const _Float16* const Wb = sW3 + lane * 8;
                // fragments two steps ahead of their MFMAs (rings of three): with one step the LDS latency showed (9.9 k cycles for 4.6 k of MFMAs)
                h8 af[3][T3], bf[3];
                auto rd = [&](int tap, int kk, int slot) {
                    const int shift = (tap / 3) * PW + (tap % 3);
#pragma unroll
                    for (int j = 0; j < T3; ++j) af[slot][j] = *reinterpret_cast<const h8*>(Tb + (j * 32 + shift) * kTPitch + kk * 16);
                    bf[slot] = *reinterpret_cast<const h8*>(Wb + (tap * 8 + kk) * 512);
                };
                rd(0, 0, 0);
                rd(0, 1, 1);
                for (int tap = 0; tap < 9; ++tap) {
                    // 24 steps per three taps would keep the ring index static; 8 steps per tap with a ring of 3 does not divide, so the
                    // slot is carried as 8 * tap mod 3 = (2 * tap) mod 3 through three unrolled variants
                    const int base = (2 * tap) % 3;
                    auto body = [&](auto base_c) {
                        constexpr int BS = decltype(base_c)::value;
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) {
                            const int nk = kk + 2, ntap = tap + (nk >> 3);
                            if (ntap < 9) rd(ntap, nk & 7, (BS + kk + 2) % 3);
#pragma unroll
                            for (int j = 0; j < T3; ++j)
                                acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[(BS + kk) % 3], af[(BS + kk) % 3][j], acc3[j], 0, 0, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, T3 + 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, T3, 0);
                        }
                    };
                    if (base == 0) body(std::integral_constant<int, 0>{});
                    else if (base == 1) body(std::integral_constant<int, 1>{});
                    else body(std::integral_constant<int, 2>{});
                }
#pragma unroll
                for (int j = 0; j < T3; ++j) {
                    float v[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + 128 + 8 * g + 4 * hh);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x = acc3[j][4 * g + q] + bq[q];
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
                        const unsigned off = orow3[j] != kOOB ? orow3[j] + unsigned(L.out_coff + 8 * (2 * gp + hh)) * 2u : kOOB;
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rs_x, off, 0, 0);
                    }
                }
            }
            if (stamp) { const long long v1 = __builtin_amdgcn_s_memtime(); tc_[4] += v1 - v0; v0 = v1; }
            __syncthreads();                               // layer done: T and the 3x3 weights are free, the new channels are visible
            if (stamp) { const long long v1 = __builtin_amdgcn_s_memtime(); tc_[5] += v1 - v0; }
        }
        if (stamp && lane == 0)
            for (int q = 0; q < 6; ++q) a.dbg[q] = tc_[q];
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Strip mode: ONE dense layer on maps too large for a workgroup per image (DenseNet-121 blocks 1-2: 56x56, 28x28), without the bottleneck
// tensor's round trip through memory (61 % of those layers' bytes) and without the per-band fixed cost that sank band mode.  A workgroup owns a
// STRIP of image rows and slides down it: every step stages R new rows of the image (R x (W + 1) raster positions = TMS tiles), runs the 1x1 over
// their K channels (chunks through LDS, producer / consumer waves as in the chain kernel), writes the new bottleneck rows into a RING of R + 2
// raster rows in LDS, and runs the 3x3 for the R rows whose three input rows are now complete.  The 72 KB of 3x3 weights are loaded ONCE per
// workgroup and stay in LDS (their own region, beside the ring and the chunk staging); the 1x1 weights are re-streamed from L2 every step straight
// into the consumers' register rings.  The producers run one chunk ahead across step boundaries (chunk 0 of step s + 1 is staged while the
// consumers finish step s), so the phases form one stream with one barrier each: NC chunks + one 3x3 phase per step.
// A ring row's pad column and the guard row in front of the ring stay zero for the whole launch; a ring row that stands for a row outside the
// image is written with zeros (the ring slots are reused, unlike the chain kernel's raster).
// ---------------------------------------------------------------------------------------------------------------------------------------------
template <int TMS>
__global__ __launch_bounds__(512) void dense_strip_f16_kernel(const DenseBlockArgs a) {
    constexpr int PITS = TMS;                              // 16-byte pieces per producer thread and chunk (rows tp / 8 + 32 i)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_blk[];
    const int H = a.h, W = a.w, PW = W + 1;
    const int R = a.band_rows, NR = R + 2, SRows = a.strip_rows;
    const int nstrips = (H + SRows - 1) / SRows;
    const int img = int(blockIdx.x) / nstrips;
    const int ya = (int(blockIdx.x) - img * nstrips) * SRows;
    const int yb = ya + SRows < H ? ya + SRows : H;
    const int ys0 = ya - 1;                                // image row of the first staged row
    const int nsteps = (yb - ya + 2 + R - 1) / R;          // rows ya - 1 .. yb are staged
    const int NPs = R * PW;                                // raster positions per step
    _Float16* const sW3 = reinterpret_cast<_Float16*>(smem_blk);                             // all 3x3 weights, fragment-major (72 KB)
    _Float16* const sT = sW3 + kW3Bytes / 2;                                                 // [1 + NR * PW + 1][kTPitch]: guard row, ring, guard row
    const int trows = NR * PW + 2;
    _Float16* const sA0 = sT + size_t(trows) * kTPitch;
    _Float16* const sA1 = sA0 + size_t(32 * TMS) * kAPitch;
    float* const sBias = reinterpret_cast<float*>(sA1 + size_t(32 * TMS) * kAPitch);         // [128] 1x1 bias, [32] 3x3 bias
    _Float16* const sPre = reinterpret_cast<_Float16*>(sBias + 160);                         // [2][kMaxK] prologue scale, shift

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pitch = a.pitch;
    _Float16* const ximg = a.x + size_t(img) * H * W * pitch;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(ximg, 0, H * W * pitch * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wf = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wfrag16), 0, int(a.w16_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w16), 0, int(a.w16_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w32 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w32), 0, int(a.w16_bytes * 2 < 0x7fffffffull ? a.w16_bytes * 2 : 0x7fffffffull), 0x00020000);
    const DenseBlockLayer& L = a.layer[0];
    const int K = L.K, NC = (K + 63) >> 6;
    const int G = nsteps * NC;                             // chunks of the whole strip, as one stream

    {       // ring, pad columns and guard rows start at zero
        const int n16 = trows * kTPitch / 8;
        for (int q = tid; q < n16; q += 512) reinterpret_cast<u32x4*>(sT)[q] = u32x4{0u, 0u, 0u, 0u};
    }

    if (wave >= 4) {
        // =========================================== PRODUCERS ===========================================
        const int tp = tid - 256;
        const int c8 = (tp & 7) * 8;
        int pry[PITS], px[PITS];       // the thread's positions inside a step: row of the step, column (pad column / past the step: px = -1)
#pragma unroll
        for (int i = 0; i < PITS; ++i) {
            const int p = (tp >> 3) + 32 * i;
            const int y = p / PW, x = p - y * PW;
            pry[i] = y;
            px[i] = (p < NPs && x < W) ? x : -1;
        }
        struct ASlot { u32x4 v[PITS]; };
        ASlot ra[3];
        auto issue = [&](ASlot& A, int g) {                // chunk g of the stream = (step g / NC, chunk g % NC); past the strip: zeros, no traffic
            const int st = g / NC, c = g - st * NC;
            const bool cok = g < G && c * 64 + c8 < K;
            const int ybase = ys0 + st * R;
#pragma unroll
            for (int i = 0; i < PITS; ++i) {
                const int y = ybase + pry[i];
                const bool ok = cok && px[i] >= 0 && y >= 0 && y < H;
                A.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? unsigned((y * W + px[i]) * pitch + a.in_coff + c * 64 + c8) * 2u : kOOB, 0, 0);
            }
        };
        auto commit = [&](const ASlot& A, int g) {
            _Float16* const dst = (g & 1) ? sA1 : sA0;
            const int c = g % NC;
            const int cb = c * 64 + c8 < K ? c * 64 + c8 : 0;
            const h8 s8 = *reinterpret_cast<const h8*>(sPre + cb), t8 = *reinterpret_cast<const h8*>(sPre + kMaxK + cb);
            const _Float16 lo = (L.flags & 1) ? _Float16(0.f) : _Float16(-65504.f);
            const h8 lo8 = {lo, lo, lo, lo, lo, lo, lo, lo};
#pragma unroll
            for (int i = 0; i < PITS; ++i) {
                const int l = (tp >> 3) + 32 * i;
                h8 v = __builtin_bit_cast(h8, A.v[i]) * s8 + t8;
                v = pk_max8(v, lo8);
                *reinterpret_cast<h8*>(dst + l * kAPitch + c8) = v;
            }
        };
        // ---- once per workgroup: constants and the 3x3 weights -> LDS ----
        {
            u32x4 cpre[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tp + 256 * j, half = q * 8 < K ? 0 : 1, off = q * 8 - half * K;
                const bool ok = L.ps != 0xffffffffu && off < K;
                cpre[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w16, ok ? ((half ? L.pt : L.ps) + unsigned(off)) * 2u : kOOB, 0, 0);
            }
            const unsigned bo = tp < 128 ? L.b1 : L.b3;
            const float cbias = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w32, (tp < 160 && bo != 0xffffffffu) ? (bo + unsigned(tp < 128 ? tp : tp - 128)) * 4u : kOOB, 0, 0));
            u32x4 w3r[18];
#pragma unroll
            for (int q = 0; q < 18; ++q) w3r[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_wf, (L.w3 + unsigned(q * 256 + tp) * 8u) * 2u, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tp + 256 * j, half = q * 8 < K ? 0 : 1, off = q * 8 - half * K;
                if (off < K) {
                    const u32x4 ident = half ? u32x4{0u, 0u, 0u, 0u} : u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
                    *reinterpret_cast<u32x4*>(sPre + half * kMaxK + off) = L.ps != 0xffffffffu ? cpre[j] : ident;
                }
            }
            if (tp < 160) sBias[tp] = cbias;
#pragma unroll
            for (int q = 0; q < 18; ++q) *reinterpret_cast<u32x4*>(sW3 + (q * 256 + tp) * 8) = w3r[q];
        }
        issue(ra[0], 0);
        issue(ra[1], 1);
        issue(ra[2], 2);
        __syncthreads();                                   // constants, 3x3 weights and the zeroed ring in place
        commit(ra[0], 0);
        issue(ra[0], 3);
        __syncthreads();                                   // chunk 0 staged
        // chunk g (slot g % 3) is committed while the consumers compute chunk g - 1; behind the last chunk of a step comes their 3x3 phase
        auto step = [&](ASlot& A, int g) {
            commit(A, g);                                  // (g == G: zeros into a buffer nobody reads)
            issue(A, g + 3);
            __syncthreads();
            if ((g - 1) % NC == NC - 1) __syncthreads();   // the consumers' 3x3 phase
        };
        int g = 1;
        for (; g + 3 <= G + 1; g += 3) {
            step(ra[1], g);
            step(ra[2], g + 1);
            step(ra[0], g + 2);
        }
        if (g <= G) { step(ra[1], g); ++g; }
        if (g <= G) { step(ra[2], g); ++g; }
    } else {
        // =========================================== CONSUMERS ===========================================
        const int np = wave;
        int pry[TMS], px[TMS];         // per tile: the lane's row inside a step and column (-1: pad column / past the step)
#pragma unroll
        for (int i = 0; i < TMS; ++i) {
            const int p = i * 32 + r;
            const int y = p / PW, x = p - y * PW;
            pry[i] = y;
            px[i] = (p < NPs && x < W) ? x : -1;
        }
        // the 3x3's tile of this wave (tile = wave): static selects, a runtime index into the per-tile arrays would send them to scratch
        int my_ry = pry[0], my_x = px[0];
#pragma unroll
        for (int i = 1; i < TMS; ++i)
            if (wave == i) { my_ry = pry[i]; my_x = px[i]; }
        struct BSlot { u32x4 f[4]; };
        BSlot bs[3];
        auto issue_b = [&](BSlot& Bq, int g) {             // the 1x1 weights of chunk g % NC (the same bytes every step: L2)
            const int c = g % NC;
            const int nks = g < G ? (K - c * 64) >> 4 : 0;
            const unsigned base = (L.w1 + (unsigned(c) * 16u + unsigned(np)) * 512u + unsigned(lane) * 8u) * 2u;
#pragma unroll
            for (int s = 0; s < 4; ++s) Bq.f[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_wf, s < nks ? base + unsigned(s) * 4096u : kOOB, 0, 0);
        };
        f32x16 acc[TMS];
        auto compute1 = [&](const BSlot& Bq, int g) {
            constexpr int NST = 4 * TMS, AHEAD = 3;
            const _Float16* const Ab = ((g & 1) ? sA1 : sA0) + r * kAPitch + hh * 8;
            h8 af[4];
#pragma unroll
            for (int q = 0; q < AHEAD; ++q) af[q & 3] = *reinterpret_cast<const h8*>(Ab + (q % TMS) * 32 * kAPitch + (q / TMS) * 16);
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
            for (int q = 0; q < NST; ++q) {
                if (q + AHEAD < NST) af[(q + AHEAD) & 3] = *reinterpret_cast<const h8*>(Ab + ((q + AHEAD) % TMS) * 32 * kAPitch + ((q + AHEAD) / TMS) * 16);
                acc[q % TMS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, Bq.f[q / TMS]), af[q & 3], acc[q % TMS], 0, 0, 0);
                if (q + AHEAD < NST) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        };
        // end of a step: bottleneck rows -> ring (pixels: value; rows outside the image: zero; pad columns untouched), then the 3x3 of the R rows
        // whose three input rows are complete
        auto finish_step = [&](int st) {
            const int sbase = st * R;                      // staged-row index of the step's first row
#pragma unroll
            for (int i = 0; i < TMS; ++i) {
                float v[16];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + np * 32 + 8 * gq + 4 * hh);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = acc[i][4 * gq + q] + bq[q];
                        v[4 * gq + q] = (L.flags & 2) ? fmaxf(x, 0.f) : x;
                        acc[i][4 * gq + q] = 0.f;
                    }
                }
                const int yimg = ys0 + sbase + pry[i];
                const bool inimg = yimg >= 0 && yimg < H;
                const int slot = (sbase + pry[i]) % NR;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const h4 qa = {_Float16(v[8 * gp + 0]), _Float16(v[8 * gp + 1]), _Float16(v[8 * gp + 2]), _Float16(v[8 * gp + 3])};
                    const h4 qb = {_Float16(v[8 * gp + 4]), _Float16(v[8 * gp + 5]), _Float16(v[8 * gp + 6]), _Float16(v[8 * gp + 7])};
                    const u32x2 xa = __builtin_bit_cast(u32x2, qa), xb = __builtin_bit_cast(u32x2, qb);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(xa[0], xb[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(xa[1], xb[1], false, false);
                    if (px[i] >= 0)
                        *reinterpret_cast<u32x4*>(sT + (1 + slot * PW + px[i]) * kTPitch + np * 32 + 8 * (2 * gp + hh)) =
                            inimg ? u32x4{s0[0], s1[0], s0[1], s1[1]} : u32x4{0u, 0u, 0u, 0u};
                }
            }
            __syncthreads();                               // the step's bottleneck rows are in the ring
            if (wave < TMS) {                              // 3x3: tile = wave; output row o = (first staged row of the step) - 1 + row of the position
                const int o = ys0 + sbase - 1 + my_ry;
                const int xo = my_x;
                const bool valid = xo >= 0 && o >= ya && o < yb;
                const int xs = xo >= 0 ? xo : 0;
                // raster rows of the three taps' input rows o - 1, o, o + 1: ring slot ((row - ys0) mod NR); tap kx adds kx - 1 positions
                const _Float16* tb[3];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    int sl = (o + ky - 1 - ys0) % NR;
                    if (sl < 0) sl += NR;
                    tb[ky] = sT + (sl * PW + xs) * kTPitch + hh * 8;          // = raster index (1 + sl * PW + xs) - 1: the kx = 0 tap
                }
                f32x16 acc3;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc3[e] = 0.f;
                const _Float16* const Wb = sW3 + lane * 8;
                h8 af[3], bf[3];
                auto rd = [&](int tap, int kk, int slot) {
                    af[slot] = *reinterpret_cast<const h8*>(tb[tap / 3] + (tap % 3) * kTPitch + kk * 16);
                    bf[slot] = *reinterpret_cast<const h8*>(Wb + (tap * 8 + kk) * 512);
                };
                rd(0, 0, 0);
                rd(0, 1, 1);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        const int st3 = tap * 8 + kk, nx = st3 + 2;
                        if (nx < 72) rd(nx >> 3, nx & 7, nx % 3);
                        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[st3 % 3], af[st3 % 3], acc3, 0, 0, 0);
                    }
                }
                float v[16];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + 128 + 8 * gq + 4 * hh);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = acc3[4 * gq + q] + bq[q];
                        v[4 * gq + q] = (L.flags & 4) ? fmaxf(x, 0.f) : x;
                    }
                }
                const unsigned orow = valid ? unsigned((o * W + xs) * pitch) * 2u : kOOB;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const h4 qa = {_Float16(v[8 * gp + 0]), _Float16(v[8 * gp + 1]), _Float16(v[8 * gp + 2]), _Float16(v[8 * gp + 3])};
                    const h4 qb = {_Float16(v[8 * gp + 4]), _Float16(v[8 * gp + 5]), _Float16(v[8 * gp + 6]), _Float16(v[8 * gp + 7])};
                    const u32x2 xa = __builtin_bit_cast(u32x2, qa), xb = __builtin_bit_cast(u32x2, qb);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(xa[0], xb[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(xa[1], xb[1], false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rs_x, valid ? orow + unsigned(L.out_coff + 8 * (2 * gp + hh)) * 2u : kOOB, 0, 0);
                }
            }
            __syncthreads();                               // 3x3 phase over: the ring slots of the oldest rows may be rewritten
        };
#pragma unroll
        for (int i = 0; i < TMS; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        issue_b(bs[0], 0);
        issue_b(bs[1], 1);
        issue_b(bs[2], 2);
        __syncthreads();                                   // constants, 3x3 weights and the zeroed ring in place
        __syncthreads();                                   // chunk 0 staged
        auto step = [&](BSlot& Bq, int g) {
            compute1(Bq, g);
            issue_b(Bq, g + 3);
            if (g % NC == NC - 1) finish_step(g / NC);     // its first barrier closes the compute phase, its second the 3x3 phase
            else __syncthreads();
        };
        int g = 0;
        for (; g + 3 <= G; g += 3) {
            step(bs[0], g);
            step(bs[1], g + 1);
            step(bs[2], g + 2);
        }
        if (g < G) { step(bs[0], g); ++g; }
        if (g < G) { step(bs[1], g); ++g; }
    }
}

static size_t block_lds_bytes(int cfg, int H, int W, int R) {
    const int PW = W + 1;
    if (cfg >= 3) {            // strip mode: 3x3 weights + ring of R + 2 raster rows (and two guard rows) + two chunk buffers + constants
        const int tms = cfg == 3 ? 2 : 3;
        return size_t(kW3Bytes) + size_t((R + 2) * PW + 2) * kTPitch * 2 + size_t(2) * 32 * tms * kAPitch * 2 + 160 * sizeof(float) + size_t(2) * kMaxK * 2;
    }
    const int trows = cfg == 2 ? (R + 2) * PW + 2 : H * PW + 2 * PW + 2;
    const size_t stage = std::max(size_t(kW3Bytes), size_t(2) * 32 * (cfg == 0 ? 7 : (cfg == 1 ? 2 : 8)) * kAPitch * 2);
    return size_t(trows) * kTPitch * 2 + stage + 2 * 160 * sizeof(float) + size_t(4) * kMaxK * 2;
}

// 0 / 1: one workgroup per image (7 / 2 position tiles at most, chains of layers); one layer on larger maps: 3 / 4 = strip mode (a workgroup slides
// down a strip of rows, `*rows` rows = 2 / 3 tiles per step), else 2 = band mode (`*rows` image rows per workgroup, halo recomputed)
static int block_cfg(int H, int W, int nlayers, int* rows) {
    const int PW = W + 1, ntiles = (H * PW + 31) / 32;
    *rows = 0;
    if (ntiles <= 2) return 1;
    if (ntiles <= 7) return 0;
    if (nlayers != 1) return -1;
    if (PW <= 64) {
        int R = 96 / PW;                       // R * PW raster positions per step in three tiles (two when one row takes more than 32)
        if (R > H) R = H;
        *rows = R;
        return (R * PW + 31) / 32 <= 2 ? 3 : 4;
    }
    int R = 256 / PW - 2;                      // (R + 2) * PW staged positions in 8 tiles
    if (R > H) R = H;
    if (R < 1) return -1;
    *rows = R;
    return 2;
}

static int strip_rows_for(int n, int H, int R) {          // image rows per strip: about one workgroup per CU, never fewer than 4 steps per strip
    int nstrips = (256 + n - 1) / n;
    const int most = H / (4 * R) > 0 ? H / (4 * R) : 1;
    if (nstrips > most) nstrips = most;
    if (nstrips < 1) nstrips = 1;
    return (H + nstrips - 1) / nstrips;
}

bool DenseBlockEligible(const DenseBlockArgs& a) {
    if (a.x == nullptr || a.wfrag16 == nullptr || a.w16 == nullptr || a.w32 == nullptr) return false;
    if (a.nlayers < 1 || a.nlayers > kMaxBlockLayers || a.n < 1 || a.h < 1 || a.w < 1) return false;
    int rows = 0;
    const int cfg = block_cfg(a.h, a.w, a.nlayers, &rows);
    if (cfg < 0 || block_lds_bytes(cfg, a.h, a.w, rows) > size_t(160) * 1024) return false;
    if (int64_t(a.n) * ((cfg == 2 ? (a.h + rows - 1) / rows : 1)) >= (int64_t(1) << 31)) return false;
    if (cfg >= 3 && a.layer[0].K > kMaxK) return false;
    if ((a.pitch & 7) || (a.in_coff & 7) || (reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.wfrag16) & 15) ||
        (reinterpret_cast<uintptr_t>(a.w16) & 15))
        return false;
    if (int64_t(a.h) * a.w * a.pitch * 2 >= (int64_t(1) << 31) || a.w16_bytes == 0 || a.w16_bytes >= (uint64_t(1) << 31)) return false;
    for (int l = 0; l < a.nlayers; ++l) {
        const DenseBlockLayer& L = a.layer[l];
        if (L.K < 64 || (L.K & 31) || L.K > kMaxK || a.in_coff + L.K > a.pitch) return false;
        if ((L.out_coff & 7) || L.out_coff + 32 > a.pitch) return false;
        if (L.out_coff < a.in_coff + L.K && L.out_coff + 32 > a.in_coff) return false;          // the new channels must not overlap what the layer reads
        if (uint64_t(L.w1) + uint64_t(128) * L.K > a.w16_bytes / 2 || uint64_t(L.w3) + 32 * 1152 > a.w16_bytes / 2) return false;
        if ((L.w1 & 7) || (L.w3 & 7)) return false;
        if (L.ps != 0xffffffffu && ((L.ps & 7) || (L.pt & 7) || L.pt == 0xffffffffu)) return false;
        // the first three chunks of a layer are requested while the previous layer's 3x3 is still running: they must not contain that layer's output
        if (l > 0 && a.layer[l - 1].out_coff < a.in_coff + 192 && a.layer[l - 1].out_coff + 32 > a.in_coff) return false;
    }
    return true;
}

hipError_t LaunchDenseBlockF16(const DenseBlockArgs& a_in, hipStream_t stream) {
    if (!DenseBlockEligible(a_in)) return hipErrorInvalidValue;
    DenseBlockArgs b = a_in;
    int rows = 0;
    const int cfg = block_cfg(b.h, b.w, b.nlayers, &rows);
    const size_t lds = block_lds_bytes(cfg, b.h, b.w, rows);
    b.band_rows = rows;
    if (cfg == 0) dense_block_f16_kernel<0><<<dim3(b.n), dim3(512), lds, stream>>>(b);
    else if (cfg == 1) dense_block_f16_kernel<1><<<dim3(b.n), dim3(512), lds, stream>>>(b);
    else if (cfg == 2) dense_block_f16_kernel<2><<<dim3(b.n * ((b.h + rows - 1) / rows)), dim3(512), lds, stream>>>(b);
    else {
        b.strip_rows = strip_rows_for(b.n, b.h, rows);
        const int nstrips = (b.h + b.strip_rows - 1) / b.strip_rows;
        if (cfg == 3) dense_strip_f16_kernel<2><<<dim3(b.n * nstrips), dim3(512), lds, stream>>>(b);
        else dense_strip_f16_kernel<3><<<dim3(b.n * nstrips), dim3(512), lds, stream>>>(b);
    }
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
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_block_f16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_strip_f16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_strip_f16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    return hipSuccess;
}

}  // namespace ie
