// Hand-written gfx950 (CDNA4, wave64) kernels for the ONNX operator set the in-scope graphs execute.
// They replace the device work ONNX Runtime's CUDA EP performs inside `Ort::Session::Run`
// (reference call site: inference_engine/src/model.cpp:1264-1270).
//
//   conv_igemm_kernel   Conv / Gemm / MatMul as an implicit GEMM on the f32 matrix cores
//                       (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 64 FLOP/clk/SIMD).
//                       M = N*OH*OW output pixels, N = Cout, K = kh*kw*Cin.
//                       * NHWC activations: channels are the contiguous GEMM-K axis, so operand staging is
//                         coalesced 16 B/lane global loads -> registers -> LDS
//                       * folded BatchNorm + ReLU of the *input* (DenseNet pre-activation) is applied while
//                         staging the A operand, bias + ReLU on the accumulators in the epilogue
//                       * the epilogue writes Cout channels at a channel offset of a wider NHWC row, which is
//                         what makes Concat free
//                       * LDS tiles are [rows][BK+4] floats: with pitch/4 odd every ds_read_b128 of a
//                         16-lane group hits 16 distinct 16-B slots (conflict-free), one b128 read feeds 4 MFMAs
//                       * register-staged double buffering: global loads of K-tile t+1 are in flight while the
//                         MFMAs of K-tile t run; one barrier per K-tile
//                       * blockIdx is remapped so each XCD (private 4 MiB L2) owns a contiguous run of M-tiles:
//                         the 3x3 halo rows shared by neighbouring tiles are then L2 hits
//   pool / gap / eltwise / copy: bandwidth-bound NHWC kernels, 16 B per lane where alignment allows.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "igemm_tiles.h"
#include "env.h"
#include "kernels.h"

namespace ie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// element access that honours TensorArg::f16 (scalar kernels; the MFMA kernels have dedicated half paths)
__device__ __forceinline__ float ld_elem(const float* p, int f16, int64_t i) {
    return f16 ? float(reinterpret_cast<const _Float16*>(p)[i]) : p[i];
}
__device__ __forceinline__ void st_elem(float* p, int f16, int64_t i, float v) {
    if (f16) reinterpret_cast<_Float16*>(p)[i] = _Float16(v);
    else p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// In-launch split-K combine ("the last arriver reduces").  Every K-slice workgroup of an output tile stores its raw
// accumulators to a slab, publishes them with ONE agent-scope release, and draws a ticket from the tile's counter; the
// workgroup that draws the last ticket acquires (agent scope), sums ALL slabs in slice order (its own included, from
// memory, so the result does not depend on arrival order: bitwise reproducible) and runs the normal bias/ReLU epilogue.
// Placement independent (no assumption on dispatch order / XCD), nobody spins, so it cannot deadlock.
// Counters are zero at allocation and reset by the last arriver.
// ------------------------------------------------------------------------------------------------
template <int TM, int TN, int NT>
__device__ __forceinline__ bool splitk_combine(f32x16 (&acc)[TM][TN], float* __restrict__ ws, int* counters, const int tile,
                                               const int num_tiles, const int split, const int nsplit, const int tid, int* lds_word) {
    constexpr int PER_WG = TM * TN * 16 * NT;
    float* slab = ws + (int64_t(split) * num_tiles + tile) * PER_WG;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) slab[((i * TN + j) * 16 + e) * NT + tid] = acc[i][j][e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores ...
    __syncthreads();                                            // ... before one lane publishes for the workgroup
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the write-back must finish before the ticket is visible
        *lds_word = __hip_atomic_fetch_add(counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int ticket = *lds_word;
    __syncthreads();                                            // the word lives in reusable LDS: everyone reads it before anyone moves on
    if (ticket != nsplit - 1) return false;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // drop this CU's stale L1 lines of the other slabs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {
        const float* p = ws + (int64_t(sp) * num_tiles + tile) * PER_WG;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += p[((i * TN + j) * 16 + e) * NT + tid];
    }
    if (tid == 0) __hip_atomic_store(counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// ------------------------------------------------------------------------------------------------
// implicit-GEMM convolution on v_mfma_f32_32x32x2_f32
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int KG, bool VEC, bool PRE, bool DEEP>
__global__ __launch_bounds__(64 * WM * WN * KG) void conv_igemm_kernel(const ConvArgs a, const int tiles_n, const int num_tiles) {
    constexpr int NT = 64 * WM * WN;            // threads of one K-group (they stage and compute one K-slice together)
    constexpr int BK = kIgemmBK;
    constexpr int LDP = BK + kIgemmLdsPad;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int GROUP_LDS = 2 * (BM + BN) * LDP;   // floats of LDS per K-group (double-buffered A and B tiles)
    static_assert(TM >= 1 && TN >= 1 && BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "bad tile");
    static_assert(KG == 1 || (VEC && BM * BN <= GROUP_LDS), "K-groups need the vector path and room for the partial tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave_all / (WM * WN);       // K-group of this wave
    float* const sA = smem + grp * GROUP_LDS;   // [2][BM][LDP]
    float* const sB = sA + 2 * BM * LDP;        // [2][BN][LDP]

    const int tid = threadIdx.x - grp * NT;     // thread index inside the K-group
    const int lane = tid & 63;
    const int wave = wave_all - grp * (WM * WN);
    const int wm_i = wave / WN, wn_i = wave % WN;
    const int r = lane & 31, hh = lane >> 5;

    // XCD-aware bijective remap of a linear tile index: indices i, i+8, i+16.. share an XCD (workgroups are dealt
    // round-robin over the 8 XCDs and the persistent stride is a multiple of 8), so each XCD owns a contiguous tile range.
    auto tile_origin = [&](int lin, int& m0_, int& n0_, int& id_) {
        const int q = num_tiles >> 3, rem = num_tiles & 7, xcd = lin & 7;
        const int swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (lin >> 3);
        const int tn = swz % tiles_n, tm = swz / tiles_n;
        m0_ = tm * BM;
        n0_ = tn * BN;
        id_ = tm * tiles_n + tn;
    };

    const int Cin = a.in.c, H = a.in.h, W = a.in.w;
    const int OH = a.out.h, OW = a.out.w, Cout = a.out.c;
    const int M = a.out.n * OH * OW;
    const int Ktot = a.kh * a.kw * Cin;
    const float* __restrict__ in = a.in.p;
    const float* __restrict__ wgt = a.w;
    const bool has_pre = a.pre_scale != nullptr;
    const int nsplit = gridDim.y, split = blockIdx.y;   // split-K over workgroups: grid.y slices the K-tiles

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();

    // Fragment reads are software-pipelined by hand: the ds_read_b128s of step kk+1 are issued BEFORE the MFMAs of step kk
    // (two register sets), otherwise every group of MFMAs starts with an exposed LDS round trip.
    auto compute = [&](int buf) {
        const float* A = sA + buf * BM * LDP + (wm_i * TM * 32 + r) * LDP + hh * 4;
        const float* B = sB + buf * BN * LDP + (wn_i * TN * 32 + r) * LDP + hh * 4;
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(A + i * 32 * LDP);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(B + j * 32 * LDP);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = *reinterpret_cast<const f32x4*>(A + i * 32 * LDP + (kk + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nxt][j] = *reinterpret_cast<const f32x4*>(B + j * 32 * LDP + (kk + 1) * 8);
            }
            // lanes 0-31 carry k = kk*8+e, lanes 32-63 carry k = kk*8+4+e, identically for A and B,
            // so the four K=2 MFMAs together cover the 8 k-values of this chunk.
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][e], bf[cur][j][e], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);      // next step's DS reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);  // ... then this step's MFMAs
        }
    };

    // ---- epilogue of one finished tile ----------------------------------------------------------------------
    // nsplit > 1: either the in-launch combine (only the last-arriving K-slice of a tile stores) or, with the two-pass
    // fallback, every slice stores its raw slab [split][M][Cout] for splitk_reduce_kernel.
    auto epilogue = [&](const int m0, const int n0, const int tile_id) {
        if (nsplit > 1 && a.counters != nullptr) {
            if (!splitk_combine<TM, TN, NT>(acc, a.workspace, a.counters, tile_id, num_tiles, split, nsplit, tid, reinterpret_cast<int*>(smem)))
                return;
        }
        const bool partial = nsplit > 1 && a.counters == nullptr;
        float* __restrict__ out = partial ? a.workspace + int64_t(split) * M * Cout : a.out.p;
        const int opitch = partial ? Cout : int(a.out.sw);
        // finish the values in place first, so the stores below issue back-to-back from distinct registers
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn_i * TN + j) * 32 + r;
            const float bv = (!partial && a.bias != nullptr && n < Cout) ? a.bias[n] : 0.f;
            const bool do_relu = a.relu && !partial;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][j][e] + bv;
                    acc[i][j][e] = do_relu ? fmaxf(v, 0.f) : v;
                }
        }
        const bool full = (m0 + BM <= M) && (n0 + BN <= Cout);     // workgroup-uniform: interior tiles skip the guards
        if (!partial && a.out.f16) {
            // fp16 precision mode: this fp32-compute kernel (the stem) feeds half activations
            _Float16* oh = reinterpret_cast<_Float16*>(a.out.p);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn_i * TN + j) * 32 + r;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int mb = m0 + (wm_i * TM + i) * 32 + 4 * hh;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mb + (e & 3) + 8 * (e >> 2);
                        if (n < Cout && m < M) oh[int64_t(m) * opitch + n] = _Float16(acc[i][j][e]);
                    }
                }
            }
        } else if (full) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn_i * TN + j) * 32 + r;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float* o = out + int64_t(m0 + (wm_i * TM + i) * 32 + 4 * hh) * opitch + n;
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[((e & 3) + 8 * (e >> 2)) * opitch] = acc[i][j][e];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn_i * TN + j) * 32 + r;
                const bool nok = n < Cout;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int mb = m0 + (wm_i * TM + i) * 32 + 4 * hh;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mb + (e & 3) + 8 * (e >> 2);
                        if (nok && m < M) out[int64_t(m) * opitch + n] = acc[i][j][e];
                    }
                }
            }
        }
    };

    int m0, n0, tile_id;
    tile_origin(blockIdx.x, m0, n0, tile_id);

    if constexpr (VEC) {
        // ---- float4 staging: thread owns column-quad `c4` of rows {rw + i*ROWS_PER_PASS} ----------------
        // The per-K-tile VALU work is kept to a few instructions per row (it competes with the MFMAs of the
        // co-resident waves for issue slots): all per-row geometry is folded ONCE into a 32-bit element offset of
        // the window's top-left tap plus a bit mask of the taps that fall inside the image; per K-tile a row costs one
        // add, one bit test and one select.  Loads are buffer loads: a masked-off lane gets an out-of-range offset and
        // the hardware returns zeros, which IS the conv zero padding when there is no activation prologue.
        constexpr int ROWS_PER_PASS = NT / 8;
        constexpr int A_IT = BM / ROWS_PER_PASS, B_IT = BN / ROWS_PER_PASS;
        static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the thread count");
        const int c4 = (tid & 7) * 4;
        const int rw = tid >> 3;
        const int cblocks = (Cin + BK - 1) / BK;
        const int KT = a.kh * a.kw * cblocks;
        const int kt_begin = int(int64_t(KT) * split / nsplit), kt_end = int(int64_t(KT) * (split + 1) / nsplit);
        const int ish = int(a.in.sh), isw = int(a.in.sw);
        constexpr unsigned OOB = 0x80000000u;    // >= num_records of every descriptor (views are < 2^31 bytes)

        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(in), 0, int(a.in_bytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(wgt), 0, Cout * Ktot * 4, 0x00020000);

        int poff[A_IT];                  // element offset of tap (0,0), channel c4 of this row's window
        unsigned taps[A_IT];             // bit t set: tap t of this row lies inside the image (kh*kw <= 32)
        int boff[B_IT];                  // element offset of (row n, k = c4) in the packed weights
        auto setup_rows = [&](const int tm0, const int tn0) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int m = tm0 + rw + i * ROWS_PER_PASS;
                const bool mok = m < M;
                const int mm = mok ? m : 0;
                const int b = mm / (OH * OW);
                const int rem = mm - b * (OH * OW);
                const int oy = rem / OW, ox = rem - oy * OW;
                const int iy0 = oy * a.sh - a.pt, ix0 = ox * a.sw - a.pl;
                poff[i] = b * int(a.in.sn) + iy0 * ish + ix0 * isw + c4;
                unsigned msk = 0;
                for (int ky = 0; ky < a.kh; ++ky)
                    for (int kx = 0; kx < a.kw; ++kx)
                        if (unsigned(iy0 + ky) < unsigned(H) && unsigned(ix0 + kx) < unsigned(W)) msk |= 1u << (ky * a.kw + kx);
                taps[i] = mok ? msk : 0u;
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) boff[i] = (tn0 + rw + i * ROWS_PER_PASS) * Ktot + c4;   // rows >= Cout land out of range
        };
        setup_rows(m0, n0);

        struct Stage {                   // one K-tile of operands in registers, between its loads and its LDS store
            f32x4 ra[A_IT], rb[B_IT];
            f32x4 s4, t4;
            unsigned okmask;
        };
        Stage st0, st1;                  // st1 is only live in the DEEP variant
        // Phase 1: issue the loads of K-tile kt (raw values; nothing here depends on their arrival).
        auto issue_loads = [&](Stage& st, int kt) {
            const int tap = kt / cblocks;
            const int c0 = (kt - tap * cblocks) * BK;
            const int ky = tap / a.kw, kx = tap - ky * a.kw;
            const int tapoff = ky * ish + kx * isw + c0;          // scalar
            const int woff = tap * Cin + c0;                      // scalar
            const bool cok = c0 + c4 < Cin;
            unsigned okmask = 0;
            if constexpr (PRE) {
                const int cc = cok ? c0 + c4 : 0;
                st.s4 = *reinterpret_cast<const f32x4*>(a.pre_scale + cc);
                st.t4 = *reinterpret_cast<const f32x4*>(a.pre_shift + cc);
            }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const bool ok = cok && ((taps[i] >> tap) & 1u);
                const unsigned off = ok ? unsigned(poff[i] + tapoff) * 4u : OOB;
                st.ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
                if constexpr (PRE) okmask |= ok ? (1u << i) : 0u;
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const unsigned off = cok ? unsigned(boff[i] + woff) * 4u : OOB;
                st.rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
            }
            st.okmask = okmask;
        };
        // Phase 3 (after the MFMAs of the previous tile): activation prologue (+ re-zeroing of padded lanes), LDS store.
        auto finish_store = [&](const Stage& st, int buf) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                f32x4 v = st.ra[i];
                if constexpr (PRE) {
                    v = v * st.s4 + st.t4;
                    if (a.pre_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    if (!(st.okmask & (1u << i))) v = f32x4{0.f, 0.f, 0.f, 0.f};   // zero padding applies AFTER the activation
                }
                *reinterpret_cast<f32x4*>(sA + buf * BM * LDP + (rw + i * ROWS_PER_PASS) * LDP + c4) = v;
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                *reinterpret_cast<f32x4*>(sB + buf * BN * LDP + (rw + i * ROWS_PER_PASS) * LDP + c4) = st.rb[i];
        };

        // This K-group's contiguous share of the workgroup's K-tiles; every group runs the same number of barrier rounds.
        const int nkt = kt_end - kt_begin;
        const int gb = kt_begin + int(int64_t(nkt) * grp / KG), ge = kt_begin + int(int64_t(nkt) * (grp + 1) / KG);
        const int rounds = (nkt + KG - 1) / KG;
        // K-tile rotation: workgroup t starts its K loop at K-tile (t mod n) and wraps around.  Row pitches of the NHWC
        // buffers and of the packed weights are multiples of 1 KiB, so workgroups marching through K in lockstep would all
        // hit the same few L2 / memory channels at any instant; rotating the start spreads them over all channels.
        const int ng = ge - gb;
        auto rot_of = [&](int id) { return ng > 1 && !(a.debug & 16) ? id % ng : 0; };
        int rot = rot_of(tile_id);
        auto kt_of = [&](int j, int rt) { const int x = j + rt; return gb + (x >= ng ? x - ng : x); };
        if (ng > 0) {
            issue_loads(st0, kt_of(0, rot));
            if constexpr (DEEP) { if (ng > 1) issue_loads(st1, kt_of(1, rot)); }
            finish_store(st0, 0);
        }
        __syncthreads();
        // Persistent workgroups (KG == 1, !DEEP): a workgroup walks tiles lin, lin + gridDim.x, ...  The operand loads of the
        // NEXT tile's first K-tile are issued before the last MFMA block of the current tile, so neither the next tile's first
        // HBM/L2 round trip nor the current tile's store epilogue leaves the matrix cores idle.
        int lin = blockIdx.x;
        int parity = 0;
        for (;;) {
            const int nlin = lin + int(gridDim.x);
            const bool has_next = KG == 1 && !DEEP && nlin < num_tiles;
            int nm0 = 0, nn0 = 0, nid = 0;
            if (has_next) tile_origin(nlin, nm0, nn0, nid);
            const int nrot = has_next ? rot_of(nid) : 0;
            if constexpr (DEEP) {
                // K-tile j lives in register stage j&1; tile j+2 is issued at the top of step j, tile j+1 is committed after it.
                for (int it = 0; it < rounds; ++it) {
                    const int buf = it & 1;
                    const bool active = it < ng, more = it + 1 < ng, more2 = it + 2 < ng;
                    if (it & 1) { if (more2) issue_loads(st1, kt_of(it + 2, rot)); }
                    else        { if (more2) issue_loads(st0, kt_of(it + 2, rot)); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (active) compute(buf);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) { if (it & 1) finish_store(st0, buf ^ 1); else finish_store(st1, buf ^ 1); }
                    __syncthreads();
                }
            } else {
            for (int it = 0; it < rounds; ++it) {
                const int buf = parity;
                const bool active = it < ng, more = it + 1 < ng;
                const bool cross = has_next && !more && active;      // last K-tile of this output tile: prefetch across the seam
                if (!(a.debug & 1)) {
                    if (more) issue_loads(st0, kt_of(it + 1, rot));
                    else if (cross) { setup_rows(nm0, nn0); issue_loads(st0, kt_of(0, nrot)); }
                }
                __builtin_amdgcn_sched_barrier(0);       // loads stay ahead of the MFMA block ...
                if (active && !(a.debug & 2)) compute(buf);
                __builtin_amdgcn_sched_barrier(0);       // ... and their consumers stay behind it
                if ((more || cross) && !(a.debug & 4)) finish_store(st0, buf ^ 1);
                if (!(a.debug & 8)) __syncthreads();
                parity ^= 1;
            }
            }
            if constexpr (KG > 1) {
                // Sum the K-groups' partial tiles through LDS (each group's own staging area is free now), in group order.
                if (grp > 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) sA[((i * TN + j) * 16 + e) * NT + tid] = acc[i][j][e];
                }
                __syncthreads();
                if (grp > 0) return;
#pragma unroll
                for (int g = 1; g < KG; ++g) {
                    const float* p = smem + g * GROUP_LDS;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[i][j][e] += p[((i * TN + j) * 16 + e) * NT + tid];
                }
            }
            epilogue(m0, n0, tile_id);
            if (!has_next) break;
            lin = nlin; m0 = nm0; n0 = nn0; tile_id = nid; rot = nrot;
            zero_acc();
        }
    } else {
        // ---- scalar gather staging: any Cin, any input strides (NCHW stem); one tile per workgroup ----------
        constexpr int ROWS_PER_PASS = NT / BK;
        constexpr int A_IT = BM / ROWS_PER_PASS, B_IT = BN / ROWS_PER_PASS;
        int64_t* const s_rbase = reinterpret_cast<int64_t*>(smem + 2 * (BM + BN) * LDP);
        int* const s_iy0 = reinterpret_cast<int*>(s_rbase + BM);
        int* const s_ix0 = s_iy0 + BM;
        for (int row = tid; row < BM; row += NT) {
            const int m = m0 + row;
            if (m < M) {
                const int b = m / (OH * OW);
                const int rem = m - b * (OH * OW);
                const int oy = rem / OW, ox = rem - oy * OW;
                s_iy0[row] = oy * a.sh - a.pt;
                s_ix0[row] = ox * a.sw - a.pl;
                s_rbase[row] = int64_t(b) * a.in.sn;
            } else {
                s_iy0[row] = -(1 << 28);
                s_ix0[row] = 0;
                s_rbase[row] = 0;
            }
        }
        __syncthreads();
        const int col = tid & (BK - 1);
        const int rw = tid / BK;
        const int KT = (Ktot + BK - 1) / BK;
        const int kt_begin = int(int64_t(KT) * split / nsplit), kt_end = int(int64_t(KT) * (split + 1) / nsplit);
        float ra[A_IT], rb[B_IT];
        float ps = 1.f, pt = 0.f;
        unsigned okA = 0, okB = 0;
        auto issue_loads = [&](int kt) {
            const int k = kt * BK + col;
            const bool kok = k < Ktot;
            const int kk = kok ? k : 0;
            const int tap = kk / Cin;
            const int c = kk - tap * Cin;
            const int ky = tap / a.kw, kx = tap - ky * a.kw;
            if (has_pre) { ps = a.pre_scale[c]; pt = a.pre_shift[c]; }
            okA = 0; okB = 0;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int row = rw + i * ROWS_PER_PASS;
                const int iy = s_iy0[row] + ky, ix = s_ix0[row] + kx;
                const bool ok = kok && unsigned(iy) < unsigned(H) && unsigned(ix) < unsigned(W);
                okA |= ok ? (1u << i) : 0u;
                const int64_t off = ok ? s_rbase[row] + int64_t(iy) * a.in.sh + int64_t(ix) * a.in.sw + int64_t(c) * a.in.sc : 0;
                ra[i] = in[off];
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int n = n0 + rw + i * ROWS_PER_PASS;
                const bool bok = kok && n < Cout;
                okB |= bok ? (1u << i) : 0u;
                rb[i] = wgt[bok ? int64_t(n) * Ktot + k : 0];
            }
        };
        auto finish_store = [&](int buf) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                float v = ra[i];
                if (has_pre) { v = v * ps + pt; if (a.pre_relu) v = fmaxf(v, 0.f); }
                if (!(okA & (1u << i))) v = 0.f;
                sA[buf * BM * LDP + (rw + i * ROWS_PER_PASS) * LDP + col] = v;
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                sB[buf * BN * LDP + (rw + i * ROWS_PER_PASS) * LDP + col] = (okB & (1u << i)) ? rb[i] : 0.f;
        };
        if (kt_begin < kt_end) {
            issue_loads(kt_begin);
            finish_store(0);
            __syncthreads();
            for (int kt = kt_begin; kt < kt_end; ++kt) {
                const int buf = (kt - kt_begin) & 1;
                const bool more = kt + 1 < kt_end;
                if (more) issue_loads(kt + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(buf);
                __builtin_amdgcn_sched_barrier(0);
                if (more) finish_store(buf ^ 1);
                __syncthreads();
            }
        }
        epilogue(m0, n0, tile_id);
    }
}

// Sum the split-K slabs, add bias, apply ReLU, write the NHWC view.  One thread per output element (n fastest).
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, const int nsplit, const int64_t M, const int Cout,
                                     const float* __restrict__ bias, const int relu, float* __restrict__ out, const int64_t opitch,
                                     const int out_f16) {
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t total = M * Cout;
    if (idx >= total) return;
    const int n = int(idx % Cout);
    const int64_t m = idx / Cout;
    float v = bias ? bias[n] : 0.f;
    for (int s = 0; s < nsplit; ++s) v += ws[int64_t(s) * total + idx];
    if (relu) v = fmaxf(v, 0.f);
    st_elem(out, out_f16, m * opitch + n, v);
}

hipError_t LaunchSplitKReduce(const ConvArgs& a, int splitk, hipStream_t stream) {
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    const int64_t total = M * a.out.c;
    splitk_reduce_kernel<<<dim3(unsigned((total + 255) / 256)), dim3(256), 0, stream>>>(a.workspace, splitk, M, a.out.c, a.bias,
                                                                                      a.relu, a.out.p, a.out.sw, a.out.f16);
    return hipGetLastError();
}

// Resident workgroups the chip can hold for this kernel (occupancy API x CU count), and the persistent grid derived from
// it: never more workgroups than tiles; a multiple of 8 so the XCD-aware tile remap keeps lin % 8 == blockIdx.x % 8.
// Workgroups of `kernel` one CU holds (registers, LDS, wave slots: the occupancy API), cached per (kernel, LDS bytes).
int ResidentPerCu(const void* kernel, int block, size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, int> cache;
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find({kernel, lds});
    if (it != cache.end()) return it->second;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
    return cache[{kernel, lds}] = per_cu;
}

static int PersistentSlots(const void* kernel, int block, size_t lds) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return ResidentPerCu(kernel, block, lds) * cus;
}
static int PersistentGrid(int num_tiles, int slots, int splitk) {
    const bool off = Knobs().no_persistent;
    if (off) return num_tiles;
    int g = slots / (splitk > 0 ? splitk : 1);
    if (g < 8) g = 8;
    g &= ~7;
    return num_tiles < g ? num_tiles : g;
}

// Does the split-K scratch of this launch fit the workspace / counter arrays handed in by the executor?
static bool SplitKFits(const ConvArgs& a, int splitk, int num_tiles, int tile_elems, int64_t out_elems) {
    if (a.workspace == nullptr) return false;
    if (a.counters != nullptr) return num_tiles <= a.num_counters && int64_t(splitk) * num_tiles * tile_elems <= a.workspace_floats;
    return int64_t(splitk) * out_elems <= a.workspace_floats;
}

bool SplitKWorkspaceOk(int64_t workspace_floats, int num_counters, int splitk, int64_t num_tiles, int tile_elems) {
    return num_tiles <= num_counters && int64_t(splitk) * num_tiles * tile_elems <= workspace_floats;
}

template <int T, bool VEC>
static size_t igemm_lds_bytes() {
    constexpr IgemmTile t = kIgemmTiles[T];
    size_t b = size_t(2) * (t.bm + t.bn) * (kIgemmBK + kIgemmLdsPad) * sizeof(float) * t.kg;
    if (!VEC) b += size_t(t.bm) * (sizeof(int64_t) + 2 * sizeof(int));
    return b;
}

template <int T, bool VEC, bool PRE>
static hipError_t launch_igemm_t(const ConvArgs& a, int splitk, hipStream_t stream) {
    constexpr IgemmTile t = kIgemmTiles[T];
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w;
    const int tiles_m = int((M + t.bm - 1) / t.bm), tiles_n = (a.out.c + t.bn - 1) / t.bn;
    const int num_tiles = tiles_m * tiles_n;
    if (splitk > 1 && !SplitKFits(a, splitk, num_tiles, t.bm * t.bn, M * a.out.c)) return hipErrorInvalidValue;
    if (t.kg > 1 && splitk > 1 && a.counters != nullptr) return hipErrorInvalidValue;   // in-launch combine assumes one K-group
    int grid = num_tiles;
    if (VEC && t.kg == 1 && !t.deep && !(splitk > 1 && a.counters != nullptr)) {
        const int slots = PersistentSlots(reinterpret_cast<const void*>(&conv_igemm_kernel<t.bm, t.bn, t.wm, t.wn, t.kg, VEC, PRE, (t.deep != 0)>),
                                          64 * t.wm * t.wn * t.kg, igemm_lds_bytes<T, VEC>());
        grid = PersistentGrid(num_tiles, slots, splitk);
    }
    conv_igemm_kernel<t.bm, t.bn, t.wm, t.wn, t.kg, VEC, PRE, (t.deep != 0)>
        <<<dim3(grid, splitk), dim3(64 * t.wm * t.wn * t.kg), igemm_lds_bytes<T, VEC>(), stream>>>(a, tiles_n, num_tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splitk == 1 || a.counters != nullptr) return e;
    const int64_t total = M * a.out.c;
    splitk_reduce_kernel<<<dim3(unsigned((total + 255) / 256)), dim3(256), 0, stream>>>(a.workspace, splitk, M, a.out.c, a.bias,
                                                                                      a.relu, a.out.p, a.out.sw, a.out.f16);
    return hipGetLastError();
}

template <int T, bool VEC, bool PRE>
static hipError_t init_igemm_t() {
    constexpr IgemmTile t = kIgemmTiles[T];
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<t.bm, t.bn, t.wm, t.wn, t.kg, VEC, PRE, (t.deep != 0)>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(igemm_lds_bytes<T, VEC>()));
}

hipError_t LaunchConvIgemm(const ConvArgs& a_in, int tile, int vec, int splitk, hipStream_t stream) {
    ConvArgs a = a_in;
    if (a.in.f16) return hipErrorInvalidValue;      // half inputs go through LaunchConvIgemmF16
    const int dbg = Knobs().debug_ablate;
    a.debug = dbg;   // timing-only ablations (wrong results): 1 no loads, 2 no MFMA, 4 no LDS stores, 8 no barrier
    if (a.out.sc != 1) return hipErrorInvalidValue;
    // bytes from in.p to one past the last element of the view: the range of the kernel's buffer descriptor
    a.in_bytes = 4 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + int64_t(a.in.c - 1) * a.in.sc + 1);
    if (vec && (a.in_bytes >= (int64_t(1) << 31) || int64_t(a.out.c) * a.kh * a.kw * a.in.c * 4 >= (int64_t(1) << 31) || a.kh * a.kw > 32))
        return hipErrorInvalidValue;
    if (splitk < 1 || splitk > 64 || (splitk > 1 && a.workspace == nullptr)) return hipErrorInvalidValue;
    if (int64_t(a.out.n) * a.out.h * a.out.w >= (int64_t(1) << 31)) return hipErrorInvalidValue;
    // 32-bit element offsets inside the kernels
    if (int64_t(a.in.n) * a.in.sn >= (int64_t(1) << 31) || int64_t(a.out.c) * a.kh * a.kw * a.in.c >= (int64_t(1) << 31) ||
        int64_t(a.out.n) * a.out.h * a.out.w * a.out.sw >= (int64_t(1) << 31))
        return hipErrorInvalidValue;
    if (vec) {
        if (a.in.sc != 1 || (a.in.c & 3) || (a.in.sw & 3) || (a.in.sh & 3) || (a.in.sn & 3) ||
            (reinterpret_cast<uintptr_t>(a.in.p) & 15) || (reinterpret_cast<uintptr_t>(a.w) & 15))
            return hipErrorInvalidValue;
        if (a.pre_scale && ((reinterpret_cast<uintptr_t>(a.pre_scale) & 15) || (reinterpret_cast<uintptr_t>(a.pre_shift) & 15)))
            return hipErrorInvalidValue;
    }
#define IE_CASE(T)                                                                                  \
    case T:                                                                                          \
        if (!vec) return launch_igemm_t<T, false, false>(a, splitk, stream);                        \
        return a.pre_scale ? launch_igemm_t<T, true, true>(a, splitk, stream) : launch_igemm_t<T, true, false>(a, splitk, stream);
#define IE_CASE_VEC(T)                                                                              \
    case T:                                                                                          \
        if (!vec) return hipErrorInvalidValue;                                                       \
        return a.pre_scale ? launch_igemm_t<T, true, true>(a, splitk, stream) : launch_igemm_t<T, true, false>(a, splitk, stream);
    switch (tile) {
        IE_CASE(0) IE_CASE(1) IE_CASE(2) IE_CASE(3) IE_CASE(4) IE_CASE(5) IE_CASE(6)
        IE_CASE_VEC(7) IE_CASE_VEC(8) IE_CASE_VEC(9) IE_CASE_VEC(10) IE_CASE_VEC(11) IE_CASE_VEC(12) IE_CASE_VEC(13) IE_CASE_VEC(14)
        IE_CASE_VEC(15)
        default: return hipErrorInvalidValue;
    }
#undef IE_CASE
#undef IE_CASE_VEC
}

hipError_t InitRasterKernels();

hipError_t InitKernels() {
    hipError_t e;
#define IE_INIT(T)                                                     \
    if ((e = init_igemm_t<T, true, true>()) != hipSuccess) return e;   \
    if ((e = init_igemm_t<T, true, false>()) != hipSuccess) return e;  \
    if ((e = init_igemm_t<T, false, false>()) != hipSuccess) return e;
    IE_INIT(0) IE_INIT(1) IE_INIT(2) IE_INIT(3) IE_INIT(4) IE_INIT(5) IE_INIT(6)
#undef IE_INIT
#define IE_INIT_VEC(T)                                                 \
    if ((e = init_igemm_t<T, true, true>()) != hipSuccess) return e;   \
    if ((e = init_igemm_t<T, true, false>()) != hipSuccess) return e;
    IE_INIT_VEC(7) IE_INIT_VEC(8) IE_INIT_VEC(9) IE_INIT_VEC(10) IE_INIT_VEC(11) IE_INIT_VEC(12) IE_INIT_VEC(13) IE_INIT_VEC(14) IE_INIT_VEC(15)
#undef IE_INIT_VEC
    return InitRasterKernels();
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with an LDS-resident input window ("raster" kernel)
//
// The batch is viewed as ONE 1-D raster of a zero-padded image stack: every image row gets one shared pad column
// (row pitch PW = W + 1) and every image one shared pad row (RH = H + 1 rows), so raster index p = (b*RH + y)*PW + x.
// For output position p, tap (ky, kx) reads input raster index p + (ky-1)*PW + (kx-1): the nine taps are nine
// constant shifts of the same raster.  A workgroup owns BMp consecutive output positions and, per 32-channel slice of
// Cin, stages the BMp + 2*PW + 2 raster rows it needs into LDS ONCE (instead of nine im2col copies) together with
// the slice's 9 x BN x 32 weights; the nine taps then run as MFMA GEMMs whose A operand is read straight from the
// LDS window at row offset (lane + shift).  Because the 32 lanes of an MFMA row block read 32 CONSECUTIVE raster
// rows, every ds_read_b128 is conflict-free for every shift with the same [rows][32+4] layout as the igemm kernel.
// Pad positions compute garbage that is never stored (1/(W+1) of the rows).
// ------------------------------------------------------------------------------------------------
struct RasterTile { int waves, tmw, tn; };
constexpr int kNumRasterTiles = 8;
// 96- and 192-position tiles exist because the tile count of a layer is fixed by its raster length: different tile sizes land on
// different fractions of the chip's resident-workgroup slots in the last round
constexpr RasterTile kRasterTiles[kNumRasterTiles] = {{4, 1, 1}, {4, 2, 1}, {2, 1, 1}, {1, 1, 1}, {4, 1, 2}, {4, 2, 2}, {3, 1, 1}, {6, 1, 1}};
// window float4s per thread that the register-prefetch variant holds (0 = synchronous staging only)
constexpr int kRasterPit[kNumRasterTiles] = {8, 12, 12, 0, 8, 12, 12, 8};

template <int WAVES, int TMW, int TN, int PIT>
__global__ __launch_bounds__(64 * WAVES) void conv3x3_raster_kernel(const ConvArgs a, const int PW, const int RH, const int PR,
                                                                    const int tiles_n, const int num_tiles) {
    constexpr int NT = 64 * WAVES;
    constexpr int BMp = 32 * TMW * WAVES, BN = 32 * TN;
    constexpr int CK = kIgemmBK, LDP = CK + kIgemmLdsPad;
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sP = smem;                                  // [PR][LDP]   input raster window
    float* const sW = smem + PR * LDP;                       // [9][BN][LDP] weights of the current channel slice
    int* const sPix = reinterpret_cast<int*>(sW + 9 * BN * LDP);   // [PR] pixel index (b*H + y)*W + x, or -1 for pad rows

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;

    int tile_m, tile_n;
    {
        const int bid = blockIdx.x;
        const int q = num_tiles >> 3, rem = num_tiles & 7, xcd = bid & 7;
        const int swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
        tile_n = swz % tiles_n;
        tile_m = swz / tiles_n;
    }
    const int P0 = tile_m * BMp, n0 = tile_n * BN;
    const int Cin = a.in.c, H = a.in.h, W = a.in.w, Cout = a.out.c;
    const int img = RH * PW;
    const int Mr = a.in.n * img;
    const int jbase = P0 - PW - 1;
    const int isw = int(a.in.sw);
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int chunks = (Cin + CK - 1) / CK;
    const int ch_begin = int(int64_t(chunks) * split / nsplit), ch_end = int(int64_t(chunks) * (split + 1) / nsplit);

    for (int l = tid; l < PR; l += NT) {
        const int j = jbase + l;
        int pix = -1;
        if (j >= 0 && j < Mr) {
            const int b = j / img;
            const int rem = j - b * img;
            const int y = rem / PW, x = rem - y * PW;
            if (y < H && x < W) pix = (b * H + y) * W + x;
        }
        sPix[l] = pix;
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in.p), 0, int(a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, Cout * 9 * Cin * 4, 0x00020000);

    f32x16 acc[TMW][TN];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- nine shifted GEMMs out of LDS: 36 (tap, kk) steps, fragment reads one step ahead of the MFMAs ----
    auto compute_slice = [&]() {
        const float* Abase = sP + (wave * 32 * TMW + r) * LDP + hh * 4;
        const float* Bbase = sW + r * LDP + hh * 4;
        constexpr int KSTEPS = CK / 8, STEPS = 9 * KSTEPS;
        f32x4 af[2][TMW], bf[2][TN];
        auto read_step = [&](int st, int slot) {
            const int tap = st / KSTEPS, kk = st - tap * KSTEPS;
            const int shift = (tap / 3) * PW + (tap % 3);
            const float* A = Abase + shift * LDP + kk * 8;
            const float* B = Bbase + tap * BN * LDP + kk * 8;
#pragma unroll
            for (int i = 0; i < TMW; ++i) af[slot][i] = *reinterpret_cast<const f32x4*>(A + i * 32 * LDP);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const f32x4*>(B + j * 32 * LDP);
        };
        read_step(0, 0);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int cur = st & 1;
            if (st + 1 < STEPS) read_step(st + 1, cur ^ 1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][e], bf[cur][j][e], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TMW + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TMW * TN, 0);
        }
    };

    if constexpr (PIT > 0) {
        // ---- register-prefetched staging: the next channel slice's window + weights are loaded (all loads in flight at
        //      once) while the current slice is on the matrix cores, then committed to LDS between two barriers ----
        constexpr int RPP = NT / 8;                 // window rows covered per pass of the workgroup
        constexpr int WIT = 9 * BN * 8 / NT;        // weight float4s per thread per slice
        static_assert(9 * BN * 8 % NT == 0, "weight items must divide evenly");
        const int c4 = (tid & 7) * 4;
        int poff[PIT];                               // element offset of (pixel, c4) or -1
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int l = (tid >> 3) + i * RPP;
            const int pix = l < PR ? sPix[l] : -1;
            poff[i] = pix >= 0 ? pix * isw + c4 : -1;
        }
        int woff[WIT];
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            const int q = tid + i * NT;
            const int tap = q / (BN * 8);
            const int rem = q - tap * (BN * 8);
            const int n = n0 + (rem >> 3);
            woff[i] = n < Cout ? (n * 9 + tap) * Cin + (rem & 7) * 4 : -1;
        }
        f32x4 pv[PIT], wv[WIT];
        auto issue = [&](int ch) {
            const int c0 = ch * CK;
            const bool cok = c0 + c4 < Cin;
#pragma unroll
            for (int i = 0; i < PIT; ++i) {
                const unsigned off = (poff[i] >= 0 && cok) ? unsigned(poff[i] + c0) * 4u : OOB;
                pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < WIT; ++i) {
                const int cw = c0 + ((tid + i * NT) & 7) * 4;
                const unsigned off = (woff[i] >= 0 && cw < Cin) ? unsigned(woff[i] + c0) * 4u : OOB;
                wv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
            }
        };
        auto commit = [&]() {
#pragma unroll
            for (int i = 0; i < PIT; ++i) {
                const int l = (tid >> 3) + i * RPP;
                if (l < PR) *reinterpret_cast<f32x4*>(sP + l * LDP + c4) = pv[i];
            }
#pragma unroll
            for (int i = 0; i < WIT; ++i) {
                const int q = tid + i * NT;
                *reinterpret_cast<f32x4*>(sW + (q >> 3) * LDP + (q & 7) * 4) = wv[i];
            }
        };
        if (ch_begin < ch_end) {
            issue(ch_begin);
            commit();
            __syncthreads();
            for (int ch = ch_begin; ch < ch_end; ++ch) {
                const bool more = ch + 1 < ch_end;
                if (more) issue(ch + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute_slice();
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    __syncthreads();      // every wave is done reading the current slice
                    commit();
                    __syncthreads();
                }
            }
        }
    } else {
    constexpr int U = 4;     // loads kept in flight per thread while staging
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int c0 = ch * CK;
        if (ch != ch_begin) __syncthreads();          // every wave is done reading the previous slice
        // ---- stage the raster window: PR rows x 8 float4 ----
        const int c4 = (tid & 7) * 4;
        const bool cok = c0 + c4 < Cin;
        for (int l0 = tid >> 3; l0 < PR; l0 += U * (NT / 8)) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int l = l0 + u * (NT / 8);
                const int pix = l < PR ? sPix[l] : -1;
                const unsigned off = (pix >= 0 && cok) ? unsigned(pix * isw + c0 + c4) * 4u : OOB;
                v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int l = l0 + u * (NT / 8);
                if (l < PR) *reinterpret_cast<f32x4*>(sP + l * LDP + c4) = v[u];
            }
        }
        // ---- stage the slice's weights: 9 taps x BN rows x 8 float4 ----
        constexpr int WITEMS = 9 * BN * 8;
        for (int q0 = tid; q0 < WITEMS; q0 += U * NT) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                const int tap = q / (BN * 8);
                const int rem = q - tap * (BN * 8);
                const int n = n0 + (rem >> 3);
                const int c = c0 + (rem & 7) * 4;
                const unsigned off = (q < WITEMS && n < Cout && c < Cin) ? unsigned((n * 9 + tap) * Cin + c) * 4u : OOB;
                v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                if (q < WITEMS) *reinterpret_cast<f32x4*>(sW + (q >> 3) * LDP + (q & 7) * 4) = v[u];
            }
        }
        __syncthreads();
        compute_slice();
    }
    }

    // ---- epilogue: only real pixels are stored (pad positions of the raster are dropped) ----
    const int tile_id = tile_m * tiles_n + tile_n;
    if (nsplit > 1 && a.counters != nullptr) {
        if (!splitk_combine<TMW, TN, NT>(acc, a.workspace, a.counters, tile_id, num_tiles, split, nsplit, tid, reinterpret_cast<int*>(sP)))
            return;
    }
    const bool partial = nsplit > 1 && a.counters == nullptr;
    const int Mpix = a.in.n * H * W;
    float* __restrict__ out = partial ? a.workspace + int64_t(split) * Mpix * Cout : a.out.p;
    const int opitch = partial ? Cout : int(a.out.sw);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + r;
        const bool nok = n < Cout;
        const float bv = (!partial && a.bias != nullptr && nok) ? a.bias[n] : 0.f;
        const bool do_relu = a.relu && !partial;
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int lb = wave * 32 * TMW + i * 32 + 4 * hh + PW + 1;     // window row of this lane's first output position
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pix = sPix[lb + (e & 3) + 8 * (e >> 2)];
                float v = acc[i][j][e] + bv;
                if (do_relu) v = fmaxf(v, 0.f);
                if (nok && pix >= 0) out[int64_t(pix) * opitch + n] = v;
            }
        }
    }
}

static size_t raster_lds_bytes(int tile, int PW) {
    const RasterTile t = kRasterTiles[tile];
    const int BMp = 32 * t.tmw * t.waves, BN = 32 * t.tn, LDP = kIgemmBK + kIgemmLdsPad;
    const int PR = BMp + 2 * PW + 2;
    return size_t(PR) * LDP * 4 + size_t(9) * BN * LDP * 4 + size_t(PR) * 4;
}
constexpr size_t kRasterMaxLds = 150 * 1024;

template <int T>
static hipError_t launch_raster_t(const ConvArgs& a, int splitk, hipStream_t stream) {
    constexpr RasterTile t = kRasterTiles[T];
    constexpr int BMp = 32 * t.tmw * t.waves, BN = 32 * t.tn;
    const int PW = a.in.w + 1, RH = a.in.h + 1;
    const int PR = BMp + 2 * PW + 2;
    const int64_t Mr = int64_t(a.in.n) * RH * PW;
    const int tiles_m = int((Mr + BMp - 1) / BMp), tiles_n = (a.out.c + BN - 1) / BN;
    const int num_tiles = tiles_m * tiles_n;
    if (splitk > 1 && !SplitKFits(a, splitk, num_tiles, BMp * BN, int64_t(a.out.n) * a.out.h * a.out.w * a.out.c)) return hipErrorInvalidValue;
    constexpr int PIT = kRasterPit[T];
    if (PIT > 0 && PR <= PIT * (64 * t.waves / 8))
        conv3x3_raster_kernel<t.waves, t.tmw, t.tn, PIT>
            <<<dim3(num_tiles, splitk), dim3(64 * t.waves), raster_lds_bytes(T, PW), stream>>>(a, PW, RH, PR, tiles_n, num_tiles);
    else
        conv3x3_raster_kernel<t.waves, t.tmw, t.tn, 0>
            <<<dim3(num_tiles, splitk), dim3(64 * t.waves), raster_lds_bytes(T, PW), stream>>>(a, PW, RH, PR, tiles_n, num_tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splitk == 1 || a.counters != nullptr) return e;
    const int64_t M = int64_t(a.out.n) * a.out.h * a.out.w, total = M * a.out.c;
    splitk_reduce_kernel<<<dim3(unsigned((total + 255) / 256)), dim3(256), 0, stream>>>(a.workspace, splitk, M, a.out.c, a.bias,
                                                                                      a.relu, a.out.p, a.out.sw, a.out.f16);
    return hipGetLastError();
}

int ConvRasterTileBn(int tile) { return (tile >= 0 && tile < kNumRasterTiles) ? 32 * kRasterTiles[tile].tn : 0; }

bool ConvRasterEligible(const ConvArgs& a, int tile) {
    if (tile < 0 || tile >= kNumRasterTiles) return false;
    if (a.kh != 3 || a.kw != 3 || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1) return false;
    if (a.out.h != a.in.h || a.out.w != a.in.w || a.pre_scale != nullptr || a.in.f16 || a.out.f16) return false;
    if (a.in.sc != 1 || a.out.sc != 1 || (a.in.c & 3) || (a.in.sw & 3) || (reinterpret_cast<uintptr_t>(a.in.p) & 15) ||
        (reinterpret_cast<uintptr_t>(a.w) & 15))
        return false;
    if (a.in.sh != a.in.sw * a.in.w || a.in.sn != a.in.sh * a.in.h) return false;      // pixel-major NHWC view
    if (int64_t(a.in.n) * (a.in.h + 1) * (a.in.w + 1) + 4096 >= (int64_t(1) << 31)) return false;
    return raster_lds_bytes(tile, a.in.w + 1) <= kRasterMaxLds;
}

hipError_t LaunchConvRaster3x3(const ConvArgs& a_in, int tile, int splitk, hipStream_t stream) {
    ConvArgs a = a_in;
    a.in_bytes = 4 * (int64_t(a.in.n - 1) * a.in.sn + int64_t(a.in.h - 1) * a.in.sh + int64_t(a.in.w - 1) * a.in.sw + int64_t(a.in.c - 1) * a.in.sc + 1);
    if (!ConvRasterEligible(a, tile) || a.in_bytes >= (int64_t(1) << 31) || int64_t(a.out.c) * 9 * a.in.c * 4 >= (int64_t(1) << 31))
        return hipErrorInvalidValue;
    if (splitk < 1 || splitk > 64 || (splitk > 1 && a.workspace == nullptr)) return hipErrorInvalidValue;
    switch (tile) {
        case 0: return launch_raster_t<0>(a, splitk, stream);
        case 1: return launch_raster_t<1>(a, splitk, stream);
        case 2: return launch_raster_t<2>(a, splitk, stream);
        case 3: return launch_raster_t<3>(a, splitk, stream);
        case 4: return launch_raster_t<4>(a, splitk, stream);
        case 5: return launch_raster_t<5>(a, splitk, stream);
        case 6: return launch_raster_t<6>(a, splitk, stream);
        case 7: return launch_raster_t<7>(a, splitk, stream);
        default: return hipErrorInvalidValue;
    }
}

template <int T>
static hipError_t init_raster_t() {
    constexpr RasterTile t = kRasterTiles[T];
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_raster_kernel<t.waves, t.tmw, t.tn, 0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(kRasterMaxLds));
    if (e != hipSuccess || kRasterPit[T] == 0) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_raster_kernel<t.waves, t.tmw, t.tn, kRasterPit[T]>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(kRasterMaxLds));
}

hipError_t InitRasterKernels() {
    hipError_t e;
    if ((e = init_raster_t<0>()) != hipSuccess) return e;
    if ((e = init_raster_t<1>()) != hipSuccess) return e;
    if ((e = init_raster_t<2>()) != hipSuccess) return e;
    if ((e = init_raster_t<3>()) != hipSuccess) return e;
    if ((e = init_raster_t<4>()) != hipSuccess) return e;
    if ((e = init_raster_t<5>()) != hipSuccess) return e;
    if ((e = init_raster_t<6>()) != hipSuccess) return e;
    return init_raster_t<7>();
}

// ------------------------------------------------------------------------------------------------
// naive convolution: one thread per output element (tiny / odd shapes, on-device cross-check)
// ------------------------------------------------------------------------------------------------
__global__ void conv_naive_kernel(const ConvArgs a, const int64_t total) {
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int Cout = a.out.c, OW = a.out.w, OH = a.out.h, Cin = a.in.c;
    const int n = int(idx % Cout);
    int64_t m = idx / Cout;
    const int ox = int(m % OW); m /= OW;
    const int oy = int(m % OH);
    const int b = int(m / OH);
    float acc = 0.f;
    const float* wrow = a.w + int64_t(n) * a.kh * a.kw * Cin;
    for (int ky = 0; ky < a.kh; ++ky) {
        const int iy = oy * a.sh - a.pt + ky;
        if (unsigned(iy) >= unsigned(a.in.h)) continue;
        for (int kx = 0; kx < a.kw; ++kx) {
            const int ix = ox * a.sw - a.pl + kx;
            if (unsigned(ix) >= unsigned(a.in.w)) continue;
            const int64_t px = int64_t(b) * a.in.sn + int64_t(iy) * a.in.sh + int64_t(ix) * a.in.sw;
            const float* wp = wrow + (ky * a.kw + kx) * Cin;
            for (int c = 0; c < Cin; ++c) {
                float v = ld_elem(a.in.p, a.in.f16, px + int64_t(c) * a.in.sc);
                if (a.pre_scale) { v = v * a.pre_scale[c] + a.pre_shift[c]; if (a.pre_relu) v = fmaxf(v, 0.f); }
                acc = fmaf(v, wp[c], acc);
            }
        }
    }
    if (a.bias) acc += a.bias[n];
    if (a.relu) acc = fmaxf(acc, 0.f);
    st_elem(a.out.p, a.out.f16, int64_t(b) * a.out.sn + int64_t(oy) * a.out.sh + int64_t(ox) * a.out.sw + n, acc);
}

hipError_t LaunchConvNaive(const ConvArgs& a, hipStream_t stream) {
    const int64_t total = int64_t(a.out.n) * a.out.h * a.out.w * a.out.c;
    if (total == 0) return hipSuccess;
    const int64_t blocks = (total + 255) / 256;
    if (blocks >= (int64_t(1) << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv_naive_kernel, dim3(unsigned(blocks)), dim3(256), 0, stream, a, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// pooling (NHWC).  V = channels per thread (4 -> 16 B/lane).
// ------------------------------------------------------------------------------------------------
template <int V>      // V = 4: float4 lanes (fp32), V = 8: 8 halfs per lane (fp16), V = 1: scalar, either element type
__global__ void pool_kernel(const PoolArgs a, const int64_t total) {
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int CV = a.out.c / V;
    const int c = int(idx % CV) * V;
    int64_t m = idx / CV;
    const int ox = int(m % a.out.w); m /= a.out.w;
    const int oy = int(m % a.out.h);
    const int b = int(m / a.out.h);
    float acc[V], ps[V], pt[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        acc[v] = a.is_max ? -INFINITY : 0.f;
        ps[v] = 1.f;
        pt[v] = 0.f;
    }
    if (a.pre_scale) {
        if constexpr (V >= 4) {        // c is a multiple of V and the blob's sub-arrays are 32-byte aligned: 16-byte loads
#pragma unroll
            for (int q = 0; q < V / 4; ++q) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(a.pre_scale + c + 4 * q);
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.pre_shift + c + 4 * q);
                ps[4 * q] = s4.x; ps[4 * q + 1] = s4.y; ps[4 * q + 2] = s4.z; ps[4 * q + 3] = s4.w;
                pt[4 * q] = t4.x; pt[4 * q + 1] = t4.y; pt[4 * q + 2] = t4.z; pt[4 * q + 3] = t4.w;
            }
        } else {
            ps[0] = a.pre_scale[c];
            pt[0] = a.pre_shift[c];
        }
    }
    int cnt = 0;
    for (int ky = 0; ky < a.kh; ++ky) {
        const int iy = oy * a.sh - a.pt + ky;
        if (iy >= a.in.h + a.pb) break;                      // beyond the padded extent (ceil_mode)
        for (int kx = 0; kx < a.kw; ++kx) {
            const int ix = ox * a.sw - a.pl + kx;
            if (ix >= a.in.w + a.pr) break;
            const bool inside = unsigned(iy) < unsigned(a.in.h) && unsigned(ix) < unsigned(a.in.w);
            if (inside || a.count_include_pad) ++cnt;
            if (!inside) continue;
            const int64_t off = int64_t(b) * a.in.sn + int64_t(iy) * a.in.sh + int64_t(ix) * a.in.sw + c;
            float x[V];
            if constexpr (V == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(a.in.p + off);
                x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
            } else if constexpr (V == 8) {
                const h8 t = *reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(a.in.p) + off);
#pragma unroll
                for (int v = 0; v < 8; ++v) x[v] = float(t[v]);
            } else x[0] = ld_elem(a.in.p, a.in.f16, off);
            if (a.pre_scale) {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float y = x[v] * ps[v] + pt[v];
                    x[v] = a.pre_relu ? fmaxf(y, 0.f) : y;
                }
            }
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = a.is_max ? fmaxf(acc[v], x[v]) : acc[v] + x[v];
        }
    }
    const int64_t ooff = int64_t(b) * a.out.sn + int64_t(oy) * a.out.sh + int64_t(ox) * a.out.sw + c;
    const float inv = (!a.is_max && cnt > 0) ? 1.f / float(cnt) : 1.f;
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = a.is_max ? acc[v] : acc[v] * inv;
    if constexpr (V == 4) {
        f32x4 t;
        t.x = acc[0]; t.y = acc[1]; t.z = acc[2]; t.w = acc[3];
        *reinterpret_cast<f32x4*>(a.out.p + ooff) = t;
    } else if constexpr (V == 8) {
        h8 t;
#pragma unroll
        for (int v = 0; v < 8; ++v) t[v] = _Float16(acc[v]);
        *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(a.out.p) + ooff) = t;
    } else st_elem(a.out.p, a.out.f16, ooff, acc[0]);
}

static bool aligned4(const TensorArg& t) {
    return !t.f16 && t.sc == 1 && !(t.c & 3) && !(t.sw & 3) && !(t.sh & 3) && !(t.sn & 3) && !(reinterpret_cast<uintptr_t>(t.p) & 15);
}
static bool aligned8h(const TensorArg& t) {
    return t.f16 && t.sc == 1 && !(t.c & 7) && !(t.sw & 7) && !(t.sh & 7) && !(t.sn & 7) && !(reinterpret_cast<uintptr_t>(t.p) & 15);
}

hipError_t LaunchPool(const PoolArgs& a, hipStream_t stream) {
    if (a.in.sc != 1 || a.out.sc != 1) return hipErrorInvalidValue;
    const bool v4 = aligned4(a.in) && aligned4(a.out), v8 = aligned8h(a.in) && aligned8h(a.out);
    const int64_t total = int64_t(a.out.n) * a.out.h * a.out.w * (a.out.c / (v8 ? 8 : (v4 ? 4 : 1)));
    if (total == 0) return hipSuccess;
    const int64_t blocks = (total + 255) / 256;
    if (blocks >= (int64_t(1) << 31)) return hipErrorInvalidValue;
    if (v8) hipLaunchKernelGGL(pool_kernel<8>, dim3(unsigned(blocks)), dim3(256), 0, stream, a, total);
    else if (v4) hipLaunchKernelGGL(pool_kernel<4>, dim3(unsigned(blocks)), dim3(256), 0, stream, a, total);
    else hipLaunchKernelGGL(pool_kernel<1>, dim3(unsigned(blocks)), dim3(256), 0, stream, a, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// global average pool with fused scale/shift/ReLU prologue.  Block = 64 channels x 4 pixel groups.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gap_kernel(const TensorArg in, const TensorArg out, const float* __restrict__ ps,
                                                   const float* __restrict__ pt, const int pre_relu) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int b = blockIdx.y;
    const int HW = in.h * in.w;
    float s = 1.f, t = 0.f;
    const bool cok = c < in.c;
    if (ps && cok) { s = ps[c]; t = pt[c]; }
    float acc = 0.f;
    if (cok)
        for (int p = g; p < HW; p += 4) {
            const int y = p / in.w, x = p - y * in.w;
            float v = ld_elem(in.p, in.f16, int64_t(b) * in.sn + int64_t(y) * in.sh + int64_t(x) * in.sw + c);
            if (ps) { v = v * s + t; }
            if (pre_relu) v = fmaxf(v, 0.f);
            acc += v;
        }
    red[g][cl] = acc;
    __syncthreads();
    if (g == 0 && cok) {
        const float tot = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        st_elem(out.p, out.f16, int64_t(b) * out.sn + c, tot / float(HW));
    }
}

hipError_t LaunchGlobalAvgPool(const TensorArg& in, const TensorArg& out, const float* pre_scale, const float* pre_shift,
                               int pre_relu, hipStream_t stream) {
    if (in.sc != 1 || out.sc != 1) return hipErrorInvalidValue;
    if (in.n == 0 || in.c == 0) return hipSuccess;
    hipLaunchKernelGGL(gap_kernel, dim3((in.c + 63) / 64, in.n), dim3(256), 0, stream, in, out, pre_scale, pre_shift, pre_relu);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// elementwise: out = relu?( scale[c]*a + shift[c] (+ b) )   (stand-alone BN / ReLU / residual Add)
// ------------------------------------------------------------------------------------------------
__global__ void eltwise_kernel(const EltArgs a, const int64_t total) {
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = int(idx % a.out.c);
    int64_t m = idx / a.out.c;
    const int x = int(m % a.out.w); m /= a.out.w;
    const int y = int(m % a.out.h);
    const int b = int(m / a.out.h);
    float v = ld_elem(a.a.p, a.a.f16, int64_t(b) * a.a.sn + int64_t(y) * a.a.sh + int64_t(x) * a.a.sw + int64_t(c) * a.a.sc);
    if (a.scale) v = v * a.scale[c] + a.shift[c];
    if (a.b.p) v += ld_elem(a.b.p, a.b.f16, int64_t(b) * a.b.sn + int64_t(y) * a.b.sh + int64_t(x) * a.b.sw + int64_t(c) * a.b.sc);
    if (a.relu) v = fmaxf(v, 0.f);
    st_elem(a.out.p, a.out.f16, int64_t(b) * a.out.sn + int64_t(y) * a.out.sh + int64_t(x) * a.out.sw + int64_t(c) * a.out.sc, v);
}

// 16 bytes per lane (4 floats / 8 halfs of one pixel's channels) when every operand is pixel-major NHWC of one element type
template <bool HALF>
__global__ void eltwise_vec_kernel(const EltArgs a, const int64_t total_vec, const int cv) {
    constexpr int V = HALF ? 8 : 4;
    typedef _Float16 h8v __attribute__((ext_vector_type(8)));
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total_vec) return;
    const int64_t p = idx / cv;
    const int c = int(idx - p * cv) * V;
    float v[V], w[V];
    auto load = [&](const TensorArg& t, float* dst) {
        if constexpr (HALF) {
            const h8v x = *reinterpret_cast<const h8v*>(reinterpret_cast<const _Float16*>(t.p) + p * t.sw + c);
#pragma unroll
            for (int i = 0; i < V; ++i) dst[i] = float(x[i]);
        } else {
            const float4 x = *reinterpret_cast<const float4*>(t.p + p * t.sw + c);
            dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
        }
    };
    load(a.a, v);
    if (a.scale) {
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = v[i] * a.scale[c + i] + a.shift[c + i];
    }
    if (a.b.p) {
        load(a.b, w);
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] += w[i];
    }
    if (a.relu) {
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if constexpr (HALF) {
        h8v o;
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = _Float16(v[i]);
        *reinterpret_cast<h8v*>(reinterpret_cast<_Float16*>(a.out.p) + p * a.out.sw + c) = o;
    } else {
        *reinterpret_cast<float4*>(a.out.p + p * a.out.sw + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

static bool elt_vec_ok(const TensorArg& t, int f16, int V) {
    return t.f16 == f16 && t.sc == 1 && (t.c % V) == 0 && (t.sw % V) == 0 && t.sh == t.w * t.sw && t.sn == t.h * t.sh &&
           (reinterpret_cast<uintptr_t>(t.p) % 16) == 0;
}

hipError_t LaunchEltwise(const EltArgs& a, hipStream_t stream) {
    const int64_t total = int64_t(a.out.n) * a.out.h * a.out.w * a.out.c;
    if (total == 0) return hipSuccess;
    {
        const int f16 = a.out.f16, V = f16 ? 8 : 4;
        if (elt_vec_ok(a.out, f16, V) && elt_vec_ok(a.a, f16, V) && (a.b.p == nullptr || elt_vec_ok(a.b, f16, V))) {
            const int64_t tv = total / V;
            const int64_t vblocks = (tv + 255) / 256;
            if (vblocks < (int64_t(1) << 31)) {
                if (f16) eltwise_vec_kernel<true><<<dim3(unsigned(vblocks)), dim3(256), 0, stream>>>(a, tv, a.out.c / V);
                else eltwise_vec_kernel<false><<<dim3(unsigned(vblocks)), dim3(256), 0, stream>>>(a, tv, a.out.c / V);
                return hipGetLastError();
            }
        }
    }
    const int64_t blocks = (total + 255) / 256;
    if (blocks >= (int64_t(1) << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(eltwise_kernel, dim3(unsigned(blocks)), dim3(256), 0, stream, a, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// strided copy (layout transforms NCHW<->NHWC, channel-slice copies).  Threads walk the OUTPUT in its
// memory order so the stores are coalesced.
// ------------------------------------------------------------------------------------------------
__global__ void copy_kernel(const TensorArg in, const TensorArg out, const int out_c_fastest, const int64_t total) {
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int b, c, y, x;
    int64_t m = idx;
    if (out_c_fastest) {
        c = int(m % out.c); m /= out.c;
        x = int(m % out.w); m /= out.w;
        y = int(m % out.h); b = int(m / out.h);
    } else {
        x = int(m % out.w); m /= out.w;
        y = int(m % out.h); m /= out.h;
        c = int(m % out.c); b = int(m / out.c);
    }
    st_elem(out.p, out.f16, int64_t(b) * out.sn + int64_t(y) * out.sh + int64_t(x) * out.sw + int64_t(c) * out.sc,
            ld_elem(in.p, in.f16, int64_t(b) * in.sn + int64_t(y) * in.sh + int64_t(x) * in.sw + int64_t(c) * in.sc));
}

hipError_t LaunchCopy(const TensorArg& in, const TensorArg& out, hipStream_t stream) {
    const int64_t total = int64_t(out.n) * out.h * out.w * out.c;
    if (total == 0) return hipSuccess;
    const int64_t blocks = (total + 255) / 256;
    if (blocks >= (int64_t(1) << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(copy_kernel, dim3(unsigned(blocks)), dim3(256), 0, stream, in, out, out.sc == 1 ? 1 : 0, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// vector add (parity with the reference's smoke-test kernel; grid-stride, 16 B per lane when aligned)
// ------------------------------------------------------------------------------------------------
__global__ void vector_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ r, const int64_t n) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) r[i] = a[i] + b[i];
}

// ------------------------------------------------------------------------------------------------
// calibration microbenchmark: register-resident v_mfma_f32_32x32x2_f32 loop (no memory traffic) with NACC
// independent accumulator chains per wave; gives the device's achievable fp32 MFMA rate for roofline fractions.
// ------------------------------------------------------------------------------------------------
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = float(threadIdx.x + i + e) * 1e-3f;
    float av = 1.0f + 1e-6f * float(threadIdx.x), bv = 0.999f - 1e-6f * float(threadIdx.x & 31);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// returns TFLOP/s (<= 0 on error); blocks_per_cu x 4 waves per CU
double MfmaPeakTflops(int nacc, int blocks_per_cu, int iters) {
    float* d = nullptr;
    const int blocks = 256 * blocks_per_cu;
    if (hipMalloc(reinterpret_cast<void**>(&d), size_t(blocks) * 256 * 4) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto launch = [&] {
        switch (nacc) {
            case 1: mfma_peak_kernel<1><<<blocks, 256>>>(d, iters); break;
            case 2: mfma_peak_kernel<2><<<blocks, 256>>>(d, iters); break;
            default: mfma_peak_kernel<4><<<blocks, 256>>>(d, iters); break;
        }
    };
    launch();
    (void)hipEventRecord(e0, nullptr);
    launch();
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    const int na = nacc == 1 ? 1 : (nacc == 2 ? 2 : 4);
    const double flops = double(blocks) * 4.0 * double(iters) * 8.0 * na * (2.0 * 32 * 32 * 2);
    return flops / (double(ms) * 1e-3) / 1e12;
}

hipError_t LaunchVectorAdd(const float* a, const float* b, float* result, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int64_t blocks = (n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048;
    hipLaunchKernelGGL(vector_add_kernel, dim3(unsigned(blocks)), dim3(256), 0, stream, a, b, result, n);
    return hipGetLastError();
}

}  // namespace ie
