// Graph planner: ONNX graph -> fused NHWC execution plan (host-only code, no HIP).
//
// This is the part of `Ort::Session` construction the reference relies on at model.cpp:843-847
// (parse + ORT_ENABLE_ALL graph optimisation, model.cpp:896) re-designed for MI355X:
//   * activations live in NHWC buffers (channels contiguous -> coalesced 16 B/lane HBM access)
//   * BatchNormalization is folded to per-channel scale/shift at load and fused either into the
//     producing conv's weights/bias (Conv->BN->ReLU) or into the consuming conv's operand staging
//     (DenseNet's pre-activation BN->ReLU->Conv)
//   * Concat(axis=1) is free: producers write into channel slices of one planned buffer
//   * dead activation buffers are recycled so the working set stays inside the 256 MiB Infinity Cache
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "onnx_reader.h"

namespace ie {

// An activation tensor inside a device buffer.  NHWC unless `nchw` (graph inputs/outputs as the ABI hands them).
struct View {
    int buf = -1;
    int64_t n = 0, c = 0, h = 1, w = 1;
    int64_t c_off = 0;    // first channel inside the buffer's pixel row
    int64_t pitch = 0;    // floats per pixel row of the buffer (>= c_off + c)
    bool nchw = false;    // dense NCHW (pitch/c_off unused)
    bool f16 = false;     // elements are IEEE halfs (fp16 precision mode: every buffer that is not a graph input/output)
    bool f8 = false;      // elements are OCP e4m3 bytes with one per-tensor scale (fp8 precision mode: spatial tensors between stem and global pool)
    int64_t numel() const { return n * c * h * w; }
    int esize() const { return f8 ? 1 : (f16 ? 2 : 4); }
};

// Storage/compute precision of a plan.  F16: activations between the graph's fp32 inputs and outputs are stored as halfs and
// the convolutions run on the fp16 MFMA path with fp32 accumulation (BASELINE.json configs[2-3]); folded BN scale/shift,
// biases and split-K partial sums stay fp32.
// F8 (BASELINE.json configs[4]): spatial activations are stored as OCP e4m3 with a per-tensor scale calibrated at load, conv weights
// as e4m3 with a per-output-channel scale, the convolutions run on the fp8 MFMA path with fp32 accumulation; [N, C] vectors (after
// the global pool) are halfs, graph inputs / outputs stay fp32.
enum class Precision : int { F32 = 0, F16 = 1, F8 = 2 };

enum class StepKind : int { Conv = 0, Pool = 1, GlobalAvgPool = 2, Eltwise = 3, Copy = 4 };

// Conv algorithm chosen at plan time.
enum class ConvAlgo : int {
    IgemmVec = 0,     // MFMA implicit GEMM, NHWC float4 operand staging (Cin % 4 == 0)
    IgemmScalar = 1,  // MFMA implicit GEMM, scalar gather staging (any Cin / NCHW input, e.g. the 7x7 stem)
    Naive = 2,        // one thread per output element (tiny or odd shapes, and on-device cross-check)
    Raster3x3 = 3,    // 3x3/s1/p1 MFMA conv with an LDS-resident input window (nine shifted GEMMs over a padded raster)
    Ws1x1 = 4,        // weights-stationary 1x1/s1 conv (fp32 and fp16), activations streamed from HBM into MFMA fragments
    Ws3x3 = 5,        // fp16 mode: weights-stationary 3x3/s1/p1 conv (Cout <= 32), raster window in LDS
    Stem = 6,         // 7x7/s2/p3 conv over the 3-channel NCHW fp32 graph input: LDS window per output tile, weights resident
    Direct = 7,       // small output grids: K split over the waves of a workgroup, operands loaded straight into MFMA fragments
    IgemmF8 = 8,      // fp8 mode: implicit GEMM over e4m3 activations / weights (v_mfma_f32_32x32x16_fp8_fp8)
    DenseFused = 9,   // fp32: 3x3 growth conv of dense layer L + 1x1 bottleneck conv of layer L+1 in one launch (Step::parts holds the two convs)
    Wino3x3 = 10,     // fp32: Winograd F(2x2, 3x3) for 3x3/s1/p1 convs with 32 output channels on even-sized images (2.25x fewer MACs)
    X6 = 11,          // fp32 1x1 conv on the bf16 matrix pipe with exactly split operands (kernels_x6.hip; opt-in IE_FP32_SPLIT=1)
    DenseBlock = 12,  // fp16: a chain of dense layers (1x1 K -> 128, 3x3 128 -> 32) in ONE launch, one workgroup per image, the bottleneck tensor
                      // kept in LDS (kernels_block.hip).  Step::parts = the 2n conv steps as the planner emitted them; tile 1 = fused, 0 = the parts
    StemPool = 14,    // the stem conv AND the 3x3/s2/p1 max pool behind it in one launch (conv_stem_kernel<POOL>, kernels_stem.hip): the conv
                      // tile is pooled in LDS, the tensor between the two ops never exists.  Step::parts = {conv, pool}; tile 1 = fused, 0 = the parts
    DualF8 = 13,      // fp8: a bottleneck block's last 1x1 conv AND the projection conv of its shortcut as two GEMMs of one launch (kernels_ws8.hip):
                      // the shortcut tensor never exists.  Step::parts = {projection conv, last conv}; this step's own fields repeat the last
                      // conv's with has_in2 cleared.  The fp16 plan built for the fp8 calibration carries the same step and runs its parts.
};

struct Step {
    StepKind kind = StepKind::Conv;
    std::string name;          // ONNX node name(s) this step came from (profiling / debugging)
    View in, in2, out;         // in2: second operand of a residual Add
    bool has_in2 = false;
    // conv / pool geometry
    int kh = 1, kw = 1, sh = 1, sw = 1, pt = 0, pl = 0, pb = 0, pr = 0;
    bool pool_max = false;
    bool count_include_pad = false;
    // offsets (in floats) into the weight blob; -1 = absent
    int64_t w_off = -1;        // conv weights packed [Cout][kh][kw][Cin]
    int64_t bias_off = -1;     // [Cout]
    int64_t pre_scale_off = -1, pre_shift_off = -1;   // per input channel, applied before the op (then pre_relu)
    bool pre_relu = false;
    bool relu = false;         // applied to the result
    ConvAlgo algo = ConvAlgo::Naive;
    int tile = 0;              // igemm tile configuration index (see igemm_tiles.h)
    int base_tile = 0;         // the tiled implicit GEMM's heuristic tile (what the executor falls back to when a specialised launcher declines)
    int splitk = 1;            // >1: K-tiles split over this many workgroups per output tile (+ reduce kernel)
    int idx = -1;              // position in Plan::steps
    int in_src = -1, in2_src = -1;   // index of the step that produced `in` / `in2` (-1: a graph input); fp8 mode looks the tensors' scales up by it
    // DenseFused: parts = {the 3x3 conv step, the 1x1 conv step} exactly as the planner emitted them; this step's own fields repeat
    // the 1x1's (in, out, weights, prologue, epilogue); tile = 16-pixel blocks per workgroup (1 or 2).  The executor launches the
    // parts one after the other when the fused launcher declines.
    std::vector<Step> parts;
    double flops = 0;          // algorithmic FLOPs (2*MACs) of this step for the planned shape
    double bytes = 0;          // algorithmic bytes: operands read once + result written once
};

struct IoDesc {
    std::string name;
    int elem_type = ONNX_FLOAT;
    std::vector<int64_t> model_dims;   // as declared in the ONNX file (-1 = symbolic)
    std::vector<int64_t> dims;         // resolved for this plan
    View view;                         // device staging view (dense NCHW order as the ABI expects)
};

struct Plan {
    std::vector<IoDesc> inputs, outputs;
    Precision precision = Precision::F32;
    std::vector<int64_t> buffer_floats;   // size of each device activation buffer in ELEMENTS
    std::vector<char> buffer_f16;         // element type of each buffer: 0 = float, 1 = half, 2 = e4m3 byte
    std::vector<Step> steps;
    std::vector<float> weights;           // packed blob (batch independent)
    int64_t workspace_floats = 0;         // split-K partial-sum slabs (max over steps of splitk*M*Cout)
    double total_flops = 0, total_bytes = 0;
    int64_t activation_floats() const { int64_t s = 0; for (auto b : buffer_floats) s += b; return s; }
    int64_t activation_bytes() const {
        int64_t s = 0;
        for (size_t i = 0; i < buffer_floats.size(); ++i) s += buffer_floats[i] * (buffer_f16[i] == 2 ? 1 : (buffer_f16[i] ? 2 : 4));
        return s;
    }
};

// Static (shape independent) facts, available right after parsing: what ExtractModelMetadata
// (model.cpp:910-972) and EstimateModelMemoryUsage (model.cpp:979-1035) report.
struct ModelInfo {
    std::vector<OnnxValueInfo> inputs, outputs;
    size_t memory_usage_bytes = 0;
};
ModelInfo DescribeModel(const OnnxModel& m);

// Build the plan for concrete input shapes (one entry per graph input, in graph order).
// Throws std::runtime_error with an ORT-like message on unsupported ops or shape mismatches.
// f8_fusions: apply the step fusions of the fp8 mode (DualF8) whatever the precision -- the fp8 calibration runs the graph in fp16 and needs
// the SAME step list as the fp8 plan it calibrates.
Plan BuildPlan(const OnnxModel& m, const std::vector<std::vector<int64_t>>& input_shapes, Precision precision = Precision::F32, bool f8_fusions = false);

std::string PlanToJson(const Plan& p);

}  // namespace ie
