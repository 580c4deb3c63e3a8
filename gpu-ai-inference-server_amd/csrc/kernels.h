// Launchers of the hand-written gfx950 kernels (definitions in kernels.hip).
// Every launcher only enqueues work on `stream` (no allocation, no synchronisation), so a whole forward
// pass can be captured into a hipGraph (executor.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ie {

// Strided activation operand: element (n, y, x, c) lives at  base + n*sn + y*sh + x*sw + c*sc.
// f16 != 0: the buffer holds IEEE half elements (fp16 precision mode); `p` is then only a typed alias of the address and
// all strides stay in ELEMENTS.
struct TensorArg {
    float* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
    int64_t sn = 0, sh = 0, sw = 0, sc = 0;
    int f16 = 0;
    int f8 = 0;                        // the buffer holds OCP e4m3 bytes (fp8 precision mode); strides in ELEMENTS = bytes; never set together with f16
};

struct ConvArgs {
    TensorArg in, out;                 // out is always NHWC (sc == 1)
    TensorArg res;                     // res.p != null: out = act(conv + bias + res), res has the output's shape (residual Add fused)
    const float* w = nullptr;          // [Cout][kh][kw][Cin]
    const void* w16 = nullptr;         // the same weights as halfs at the same element offset (fp16 precision mode)
    const float* wfrag = nullptr;      // fragment-major fp32 mirror of w (LaunchPermuteWeightsFrag), or null
    const float* bias = nullptr;       // [Cout] or null
    const float* pre_scale = nullptr;  // [Cin] or null: x <- x*scale + shift (then ReLU if pre_relu) before the conv
    const float* pre_shift = nullptr;
    const void* pre_scale16 = nullptr; // half copies of pre_scale / pre_shift (fp16 precision mode, packed-half prologue)
    const void* pre_shift16 = nullptr;
    int kh = 1, kw = 1, sh = 1, sw = 1, pt = 0, pl = 0;
    int pre_relu = 0, relu = 0;
    float* workspace = nullptr;        // split-K partial slabs (only when splitk > 1)
    int64_t workspace_floats = 0;      // capacity of `workspace`
    int* counters = nullptr;           // per-tile arrival counters for the in-launch combine; null = two-pass reduce kernel
    int num_counters = 0;
    // fp8 precision mode (kernels_f8.hip): e4m3 weights at the same element offsets as `w`, the per-output-channel multiplier that
    // turns the e4m3 x e4m3 dot product into real units (input scale x weight-row scale), the shortcut's scale, 1 / output scale
    const void* w8 = nullptr;
    const float* escale = nullptr;
    float res_scale = 1.f, out_qscale = 1.f;
    // fp8 mode, DUAL 1x1 (kernels_ws8.hip): a second GEMM accumulated into the same output -- the projection shortcut of a bottleneck block,
    // out = act(conv(in) + conv_b(in2)) with in2 read at pixel (oy * sh2, ox * sw2); in2.p == null: none
    TensorArg in2;
    const void* w8b = nullptr;
    const float* escale_b = nullptr;
    const float* bias_b = nullptr;
    int sh2 = 1, sw2 = 1;
    int64_t in2_bytes = 0;
    int debug = 0;                     // timing-only ablation bits (IE_DEBUG_ABLATE), 0 in production
    int64_t in_bytes = 0;              // filled by LaunchConvIgemm: byte span of the input view (buffer descriptor range)
};

// The 3x3 half of a fused dense-layer step (kernels_fused.hip): bottleneck tensor in, 32 fresh channels out (the tail of the 1x1's
// input view), fragment-major 3x3 weights.
struct FusedArgs {
    TensorArg in3, out3;
    const float* wfrag3 = nullptr;
    const float* bias3 = nullptr;
    int relu3 = 0;
};

struct PoolArgs {
    TensorArg in, out;                 // NHWC
    int kh = 1, kw = 1, sh = 1, sw = 1, pt = 0, pl = 0, pb = 0, pr = 0;
    int is_max = 0, count_include_pad = 0;
    const float* pre_scale = nullptr;  // per channel: x <- x*scale + shift (then ReLU if pre_relu) before pooling
    const float* pre_shift = nullptr;  // (a transition's BN -> ReLU -> 1x1 conv -> AvgPool runs as BN -> ReLU -> AvgPool -> 1x1 conv)
    int pre_relu = 0;
    float in_scale = 1.f, out_qscale = 1.f;   // e4m3 tensors: real = q * in_scale, q_out = real * out_qscale
};

struct EltArgs {
    TensorArg a, b, out;               // b.p == null: no second operand
    const float* scale = nullptr;      // per channel, or null
    const float* shift = nullptr;
    int relu = 0;
};

// vec: 1 = float4 NHWC operand staging, 0 = scalar gather staging.  tile: index into kIgemmTiles.
// splitk > 1: the K-tiles are divided over grid.y workgroups that write partial slabs to a.workspace; the slabs are
// combined inside the same launch by the last-arriving workgroup of each tile (a.counters != null) or by a second
// kernel (a.counters == null).  Both sum in slice order: deterministic, no float atomics.
bool SplitKWorkspaceOk(int64_t workspace_floats, int num_counters, int splitk, int64_t num_tiles, int tile_elems);
hipError_t LaunchConvIgemm(const ConvArgs& a, int tile, int vec, int splitk, hipStream_t stream);
// 3x3 / stride 1 / pad 1 with an LDS-resident input window (see kernels.hip).  tile: 0..kNumConvRasterTiles-1.
constexpr int kNumConvRasterTiles = 8;
bool ConvRasterEligible(const ConvArgs& a, int tile);
int ConvRasterTileBn(int tile);
hipError_t LaunchConvRaster3x3(const ConvArgs& a, int tile, int splitk, hipStream_t stream);
hipError_t LaunchConvNaive(const ConvArgs& a, hipStream_t stream);
// fp16 precision mode (kernels_f16.hip): NHWC half activations + half weights on v_mfma_f32_32x32x16_f16, fp32 accumulate,
// fp32 scale/shift/bias; output half or float.  Same tile table as the fp32 igemm (entries with deep == 0).
// second pass of a two-pass split-K conv: out = act(sum of a.workspace slabs + bias)
hipError_t LaunchSplitKReduce(const ConvArgs& a, int splitk, hipStream_t stream);
hipError_t LaunchConvIgemmF16(const ConvArgs& a, int tile, int splitk, hipStream_t stream);
hipError_t InitKernelsF16();
// fp16 weights-stationary 1x1 conv (kernels_ws.hip): weights in LDS once per persistent workgroup, activations streamed
// from HBM straight into MFMA fragments.  tile % 6 = {output channels per workgroup, waves}; tile / 6 = 0: persistent workgroups
// with an even number of row blocks per wave, 1: one row block per wave (the hardware's workgroup dispatch balances the load),
// 2 (fp16 only, tiles 12-17): ONE persistent workgroup per CU -- fewer, longer streams: 10-15 % faster than the fuller grid on DenseNet's
// block-1 / block-2 shapes at batch 128 (scripts/probes/ws_probe.cpp -DWS_PER_CU), the search decides per shape.
constexpr int kNumConvWsTiles = 12;
constexpr int kNumConvWs16Tiles = 18;
bool ConvWsEligible(const ConvArgs& a, int tile);
hipError_t LaunchConvWs1x1F16(const ConvArgs& a, int tile, hipStream_t stream);
hipError_t InitKernelsWs();
// fp32 twin (kernels_ws32.hip): same tile table, v_mfma_f32_32x32x2_f32, float in / float out; two more tiles (12, 13) are the
// K-split variants (8 / 4 waves of a workgroup share one row block and split K, partial tiles summed through LDS); 14-19: shapes 0-5 on a grid
// of ONE persistent workgroup per CU (as the fp16 tiles 12-17)
constexpr int kNumConvWs32Tiles = kNumConvWsTiles + 2 + 6;
bool ConvWs32Eligible(const ConvArgs& a, int tile);
hipError_t LaunchConvWs1x1F32(const ConvArgs& a, int tile, hipStream_t stream);
hipError_t InitKernelsWs32();
// fp16 weights-stationary 3x3/s1/p1 conv (Cout <= 32, all weights of the layer resident in LDS, raster window per 64-channel slice)
constexpr int kNumConvWs3Tiles = 5;
bool ConvWs3Eligible(const ConvArgs& a, int tile);
hipError_t LaunchConvWs3x3F16(const ConvArgs& a, int tile, hipStream_t stream);
hipError_t InitKernelsWs3();
// "Direct split-K" conv for small output grids (kernels_direct.hip): K split over the waves of a workgroup, both operands loaded
// straight from global memory into MFMA fragments (all at once), partial tiles summed through LDS.  fp32 and fp16.
constexpr int kNumDirectBaseTiles = 6;     // tiles 0..5: both operands straight from global memory
constexpr int kNumConvDirectTiles = 15;    // tiles 6..9: "window" variants (fp32, 16x16x4 MFMA tiles): activations through LDS, fragment-major weights
                                           // tiles 10..14: activations-stationary 1x1 (fp32): 32 / 16 pixel rows in LDS, weights streamed from the mirror
hipError_t LaunchPermuteWeightsFrag(const float* src, float* dst, int Cout, int KK, int Cin, hipStream_t stream);
bool ConvDirectEligible(const ConvArgs& a, int tile);
hipError_t LaunchConvDirect(const ConvArgs& a, int tile, hipStream_t stream);
hipError_t InitKernelsDirect();
// Stem conv (7x7 / stride 2 / pad 3, Cin = 3, Cout <= 64) straight from the dense NCHW fp32 graph input (kernels_stem.hip);
// half arithmetic + half output in fp16 mode, fp32 otherwise.
// Workgroups of `kernel` (block threads, lds dynamic LDS bytes) one CU holds, from the occupancy API (registers included), cached.
int ResidentPerCu(const void* kernel, int block, size_t lds);

bool ConvStemEligible(const ConvArgs& a);
hipError_t LaunchConvStem(const ConvArgs& a, hipStream_t stream);
// The stem AND the 3x3 / stride 2 / pad 1 max pool behind it in one launch: `a` is the stem's argument set with out = the POOLED tensor.
bool ConvStemPoolEligible(const ConvArgs& a);
hipError_t LaunchConvStemPool(const ConvArgs& a, hipStream_t stream);
hipError_t InitKernelsStem();
// Winograd F(2x2, 3x3) conv (kernels_wino.hip): fp32, 3x3 / stride 1 / pad 1, 32 output channels, even H and W; a.wfrag = the transformed
// weights U (16 x Cout x Cin floats, fragment-major) built by LaunchWinogradWeights.  tile: 0..3 = output tiles per workgroup.
constexpr int kNumConvWinoTiles = 12;       // output tiles per workgroup: 4x7, 2x14, 4x8, 2x16; 0..3 four waves, 4..7 eight waves, 8..11 eight waves with bf16x6 products (need `w16` = LaunchWinogradWeightsX6's mirror)
bool ConvWinoEligible(const ConvArgs& a, int tile);
hipError_t LaunchConvWino3x3(const ConvArgs& a, int tile, hipStream_t stream);
hipError_t LaunchWinogradWeights(const float* w, float* u, int Cout, int Cin, hipStream_t stream);
hipError_t LaunchWinogradWeightsX6(const float* w, void* dst, int Cout, int Cin, hipStream_t stream);      // 3 x 16 x Cout x Cin bf16
hipError_t InitKernelsWino();
// fp32 1x1 conv on the bf16 matrix pipe with exactly split operands (kernels_x6.hip; opt-in, IE_FP32_SPLIT=1): `w16` points at the three
// bf16 planes LaunchSplitWeightsX6 built from the conv's fp32 weights.  tile 0: 64 pixels per workgroup, 1: 32.
constexpr int kNumConvX6Tiles = 2;
bool ConvX6Eligible(const ConvArgs& a, int tile);
hipError_t LaunchConvX6(const ConvArgs& a, int tile, hipStream_t stream);
hipError_t LaunchSplitWeightsX6(const float* w, void* dst, int Cout, int K, hipStream_t stream);
hipError_t InitKernelsX6();
// Fused dense-layer step (fp32): 3x3 growth conv of layer L + 1x1 bottleneck conv of layer L+1 per 16*pb-pixel tile, one launch.
// tile: 1 / 2 = 16-pixel blocks per workgroup; 3 = 16-pixel tiles, two workgroups per CU (<= 128 VGPRs, <= 80 KB LDS);
// 4 / 5 = wave-specialised variant (3x3 and 1x1 run concurrently on different waves), 16 / 32 pixels
bool ConvDenseFusedEligible(const ConvArgs& a, const FusedArgs& f, int tile);
hipError_t LaunchConvDenseFused(const ConvArgs& a, const FusedArgs& f, int tile, hipStream_t stream);
hipError_t InitKernelsFused();
// fp16 mode: a chain of dense layers (BN -> ReLU -> 1x1 conv K -> 128 -> BN -> ReLU -> 3x3 conv 128 -> 32) per launch, one workgroup per image, the
// bottleneck tensor kept in LDS (kernels_block.hip).  Weights come from a fragment-major mirror of the half blob (LaunchPermuteWeightsFrag16).
struct DenseBlockLayer {
    int K = 0;                         // input channels of the 1x1 (multiple of 32, >= 64): channels [in_coff, in_coff + K) of the pixel row
    int out_coff = 0;                  // where the 32 new channels go in the pixel row
    unsigned w1 = 0, w3 = 0;           // element offsets into wfrag16: fragment-major [128][K] and [32][9 * 128] half weights
    unsigned ps = 0xffffffffu, pt = 0xffffffffu;   // element offsets into w16 of the 1x1's prologue scale / shift (halfs), 0xffffffff = none
    unsigned b1 = 0xffffffffu, b3 = 0xffffffffu;   // float offsets into w32 of the two biases, 0xffffffff = none
    int flags = 0;                     // 1: ReLU in the prologue, 2: ReLU after the 1x1, 4: ReLU after the 3x3
};
constexpr int kMaxBlockLayers = 24;
struct DenseBlockArgs {
    _Float16* x = nullptr;             // block buffer, NHWC halfs
    int pitch = 0, in_coff = 0;        // halfs per pixel row; first channel the 1x1 convs read
    int n = 0, h = 0, w = 0;
    const _Float16* wfrag16 = nullptr;
    const _Float16* w16 = nullptr;
    uint64_t w16_bytes = 0;            // size of each of the two half blobs (buffer descriptors: < 2 GiB)
    const float* w32 = nullptr;
    int nlayers = 0;
    int band_rows = 0;                 // set by the launcher: 0 = a workgroup per image, else image rows per workgroup (band mode) / per step (strip mode)
    int strip_rows = 0;                // set by the launcher: > 0 = strip mode: image rows per workgroup, walked in steps of band_rows
    long long* dbg = nullptr;          // probes only: per-phase cycle sums of workgroup 0 / wave 0 (1x1 loop, weight DMA + 1x1 epilogue, 3x3, closing barrier)
    DenseBlockLayer layer[kMaxBlockLayers];
};
bool DenseBlockEligible(const DenseBlockArgs& a);
hipError_t LaunchDenseBlockF16(const DenseBlockArgs& a, hipStream_t stream);
hipError_t LaunchPermuteWeightsFrag16(const void* src, void* dst, int rows, int K, hipStream_t stream);
hipError_t InitKernelsBlock();
// fp8 precision mode (kernels_f8.hip): implicit GEMM on v_mfma_f32_32x32x16_fp8_fp8 over e4m3 NHWC activations and e4m3 weights,
// fp32 accumulate, per-channel rescale + bias + e4m3 shortcut + ReLU + re-quantisation in the epilogue.  Tiles 0..6 of kIgemmTiles.
constexpr int kNumConvF8Tiles = 7;
bool ConvF8Eligible(const ConvArgs& a, int tile);
hipError_t LaunchConvIgemmF8(const ConvArgs& a, int tile, hipStream_t stream);
hipError_t InitKernelsF8();
// fp8 weights-stationary 1x1 conv (kernels_ws8.hip): weights + epilogue constants in LDS, activations streamed from HBM into MFMA fragments;
// with a.in2 set also the projection shortcut's GEMM in the same launch.  tile: {N tiles of 32 per workgroup, waves} = {8,8} {4,8} {2,8} {4,4} {2,4}
constexpr int kNumConvWs8Tiles = 10;       // 5 tile shapes x {grid sized for two workgroups per CU, for one (tiles 5-9)}
bool ConvWs8Eligible(const ConvArgs& a, int tile);
hipError_t LaunchConvWs1x1F8(const ConvArgs& a, int tile, hipStream_t stream);
hipError_t InitKernelsWs8();
// fp8 weights-stationary 3x3/s1/p1 conv (kernels_ws8.hip): all weights of a 32-channel N tile resident in LDS, raster window per 128-channel slice
constexpr int kNumConvWs38Tiles = 8;      // 4 tile shapes x {grid sized by LDS (<= two workgroups per CU), one workgroup per CU (tiles 4-7)}
bool ConvWs38Eligible(const ConvArgs& a, int tile);
hipError_t LaunchConvWs3x3F8(const ConvArgs& a, int tile, hipStream_t stream);
// w8[o, :] = e4m3(w[o, :] / wscale[o]) with wscale[o] = max|w[o, :]| / 448, one workgroup per row
hipError_t LaunchQuantizeRowsE4m3(const float* w, void* w8, float* wscale, int rows, int K, hipStream_t stream);
hipError_t LaunchScaleVector(const float* src, float* dst, float s, int n, hipStream_t stream);
hipError_t LaunchPoolF8(const PoolArgs& a, hipStream_t stream);
hipError_t LaunchGlobalAvgPoolF8(const TensorArg& in, const TensorArg& out, float in_scale, hipStream_t stream);
// calibration: *result = max(*result, max |x| over the view) (fp32 / half views; *result must start at 0)
hipError_t LaunchAbsMax(const TensorArg& t, float* result, hipStream_t stream);
hipError_t LaunchE4m3RoundTrip(const float* src, float* dst, void* codes, float scale, int64_t n, hipStream_t stream);
hipError_t LaunchConvertF32ToF16(const float* src, void* dst, int64_t n, hipStream_t stream);
// UINT8 ingest: dst[i] = float(src[i]) * scale + bias (images travel over PCIe as bytes, 4x fewer than fp32)
hipError_t LaunchConvertU8ToF32(const void* src, float* dst, int64_t n, float scale, float bias, hipStream_t stream);
hipError_t LaunchPool(const PoolArgs& a, hipStream_t stream);
// out[n, c] = mean over (y, x) of f(in[n, y, x, c]),  f = optional scale/shift/ReLU prologue
hipError_t LaunchGlobalAvgPool(const TensorArg& in, const TensorArg& out, const float* pre_scale, const float* pre_shift,
                               int pre_relu, hipStream_t stream);
hipError_t LaunchEltwise(const EltArgs& a, hipStream_t stream);
hipError_t LaunchCopy(const TensorArg& in, const TensorArg& out, hipStream_t stream);
// result[i] = a[i] + b[i]  (the reference's only authored kernel: cuda_utils.cu:10-15)
hipError_t LaunchVectorAdd(const float* a, const float* b, float* result, int64_t n, hipStream_t stream);

// Calibration: achievable fp32 MFMA rate of this device (register-resident loop, no memory traffic).
double MfmaPeakTflops(int nacc, int blocks_per_cu, int iters);

// One-time per-process setup (raises the dynamic-LDS limit of the igemm kernels).
hipError_t InitKernels();

}  // namespace ie
