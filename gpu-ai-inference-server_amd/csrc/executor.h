// Device-side model: weights resident in HBM, planned activation buffers, hipGraphs per input shape.
//
// MI355X-native replacement for the Ort::Session the reference owns per model
// (inference_engine/src/model.cpp:706-714, created at :847, run at :1264-1270).
//
// Ownership: one DeviceWeights per (model, device) holds everything derived from the ONNX initializers (the packed fp32 blob
// and its half / fragment-major / fp8 mirrors) plus the kernel-choice cache.  One DeviceModel is one EXECUTION LANE on that
// device: its own streams, events, pinned staging, plan instances and hipGraphs.  Lanes on the same device share a
// DeviceWeights (config.json "instance_count"), replicas on other devices own theirs and receive the blob by RCCL broadcast
// (bridge.cpp).  A lane is used by one host thread at a time; the bridge's lane pool guarantees that.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "kernels.h"
#include "onnx_reader.h"
#include "env.h"
#include "plan.h"

namespace ie {

struct StepTiming {
    std::string name, kernel;
    double ms = 0, flops = 0, bytes = 0;
};

struct DeviceWeights {
    int device = 0;
    float* d_weights = nullptr;        // packed fp32 blob (what the RCCL broadcast moves)
    void* d_weights16 = nullptr;       // fp16 / fp8 modes: the same blob as halfs, same element offsets
    void* d_weights16_frag = nullptr;  // fp16 mode: the half conv weights of the dense-block convs again in MFMA-fragment order at the same element offsets (kernels_block.hip)
    struct Frag16Region { int64_t w_off; int rows, k; };
    std::vector<Frag16Region> frag16_regions;
    float* d_weights_frag = nullptr;   // fp32 mode: conv weights again in MFMA-fragment order at the same offsets
    void* d_weights8 = nullptr;        // fp8 mode: conv weights as OCP e4m3 bytes at the same element offsets (per-output-channel scaled)
    // fp8 mode: per fp8 conv (keyed by its weight offset) 2 x Cout floats in d_f8_aux: the weight rows' scales, then the epilogue
    // multipliers escale[o] = (input tensor scale) x (weight row scale); act_scale[i] = scale of the tensor step i writes (real = q x
    // scale), found by a calibration pass at load
    float* d_f8_aux = nullptr;
    struct F8Conv { int step; int64_t w_off; int cout, k; int in_src; int64_t aux_off; };
    std::vector<F8Conv> f8_convs;
    std::vector<float> act_scale;
    bool f8_ready = false;
    std::atomic<uint64_t> f8_gen{0};   // bumped whenever act_scale / escale change: graphs captured under an older value are stale (scales are by-value kernel arguments)
    // fp32 mode: Winograd-transformed weights U (16 x Cout x Cin floats per eligible 3x3 conv, fragment-major), own offsets
    float* d_weights_wino = nullptr;
    struct WinoRegion { int64_t w_off, u_off; int cout, cin; };
    std::vector<WinoRegion> wino_regions;
    // fp32 mode with IE_FP32_SPLIT=1: every eligible 1x1 conv's weights as three bf16 planes in MFMA-fragment order (kernels_x6.hip)
    void* d_weights_x6 = nullptr;
    void* d_weights_wino_x6 = nullptr;   // and the Winograd U of every region of wino_regions as three bf16 planes (byte offset = 6 x its u_off)
    struct X6Region { int64_t w_off, byte_off; int cout, k; };
    std::vector<X6Region> x6_regions;
    bool fp32_split = false;
    struct FragRegion { int64_t w_off; int cout, kk, cin; };
    std::vector<FragRegion> frag_regions;
    size_t weight_floats = 0;
    size_t wino_floats = 0, x6_bytes = 0, f8_aux_floats = 0;   // sizes of the derived allocations (checksums / observability)
    bool uploaded = false;             // the fp32 blob holds the model's weights (false until the upload or the broadcast happened)
    std::atomic<size_t> device_bytes{0};
    // kernel choices found by the search, shared by every lane of the device: conv signature -> (encoded tile, split-K)
    std::mutex tune_mu;
    std::map<std::vector<int64_t>, std::pair<int, int>> tune_cache;
    std::string tune_cache_path;       // "" = not persisted
    bool tune_dirty = false;
    ~DeviceWeights();
};

struct PlanInstance {
    Plan plan;
    std::vector<float*> buffers;       // device activation buffers (plan.buffer_floats); aliased entries are not owned
    std::vector<char> owned;           // buffers[i] was hipMalloc'ed by this instance
    float* workspace = nullptr;        // split-K slabs
    bool owns_workspace = false;
    int64_t workspace_floats = 0;
    int* counters = nullptr;           // split-K arrival counters (zero between launches)
    hipGraphExec_t graph_exec = nullptr;   // the whole forward (device-resident replay); for a chunk instance: its head steps
    bool graph_ready = false;
    uint64_t captured_gen = 0;         // DeviceWeights::f8_gen the graphs of this instance were captured under
    std::vector<void*> u8_stage;       // per graph input: device staging for UINT8 payloads (allocated on first use)
    // ---- pipelined host path (ModelInfer): the batch is cut into `chunks.size()` image ranges; the first `head_steps` steps run
    // per range as soon as that range's H2D has landed (every step is batch-parallel, a range is a pointer offset), the rest runs
    // once on the whole batch.  Chunk instances hold the plan for the range's batch size (own kernel choices) and alias the
    // parent's buffers.
    std::vector<std::unique_ptr<PlanInstance>> chunks;
    int64_t batch_off = 0;             // chunk instance: first image of its range inside the parent's tensors
    int head_steps = 0;
    hipGraphExec_t tail_exec = nullptr;
    bool pipeline_tried = false;
};

struct DeviceModelOptions {
    Precision precision = Precision::F32;
    std::shared_ptr<DeviceWeights> share;   // lane on a device that already holds the weights
    bool upload_weights = true;             // false: allocate the blob only (an RCCL broadcast fills it, then WeightsArrived())
    std::string tune_cache_path;            // where the kernel-choice cache is persisted ("" = nowhere)
    bool fp32_split = false;                // fp32 mode: allow the bf16x6 kernels (config.json "fp32_split"; IE_FP32_SPLIT overrides)
};

class DeviceModel {
public:
    // Parses nothing itself: takes the decoded graph. Throws std::runtime_error (never falls back to a CPU path).
    DeviceModel(std::shared_ptr<const OnnxModel> model, int device_id, const DeviceModelOptions& opt);
    ~DeviceModel();
    DeviceModel(const DeviceModel&) = delete;
    DeviceModel& operator=(const DeviceModel&) = delete;

    // Get (or build: plan + allocate + capture) the instance for these input shapes.  allow_tune: run the exhaustive kernel
    // search for conv signatures not in the cache (load time / EnginePrepare); without it (the request path) unseen signatures
    // take the cached choice of the nearest pixel count or the planner's default -- a request never waits for a search.
    PlanInstance& Prepare(const std::vector<std::vector<int64_t>>& shapes, bool allow_tune);
    // Enqueue one forward of `pi` on the model stream (graph replay when available).
    void Enqueue(PlanInstance& pi);
    // Eager forward with hipEvents around every launch; per-step times in ms (averaged over iters).
    std::vector<StepTiming> Profile(PlanInstance& pi, int iters);
    void Synchronize();

    // Host-buffer inference (the ModelInfer path): inputs[i] has in_bytes[i] valid bytes (shorter payloads are
    // zero-extended like the reference's zero-initialised Tensor), outputs[i] receives min(out_bytes[i], produced)
    // bytes and the rest of the caller buffer is zero-filled.
    // in_u8[i] != 0: inputs[i] is a UINT8 payload (one byte per element); it is uploaded as bytes and converted on the device
    // (x * u8_scale + u8_bias), cutting the PCIe traffic of an image batch 4x.
    void InferHost(PlanInstance& pi, const std::vector<const void*>& inputs, const std::vector<size_t>& in_bytes,
                   const std::vector<void*>& outputs, const std::vector<size_t>& out_bytes, const std::vector<char>& in_u8 = {});
    void SetU8Transform(float scale, float bias) { u8_scale_ = scale; u8_bias_ = bias; }
    // Gather/scatter form used by the dynamic batcher and the sharder: several callers' row blocks land at byte offsets of one
    // device batch.  Input segment: `have` valid bytes at `host`, zero-extended to `need` bytes, placed at `dev_off` of input k.
    // Output segment: min(cap, need) bytes from `dev_off` of output j copied to `host`, the rest of `cap` zero-filled.
    struct InSeg { const void* host; size_t have, need, dev_off; bool u8 = false; };   // u8: have/need/dev_off count bytes = elements
    struct OutSeg { void* host; size_t cap, need, dev_off; };
    void InferHostSegments(PlanInstance& pi, const std::vector<std::vector<InSeg>>& in, const std::vector<std::vector<OutSeg>>& out);

    hipStream_t stream() const { return stream_; }
    int device() const { return device_; }
    float* weights() const { return w_->d_weights; }
    const std::shared_ptr<DeviceWeights>& shared_weights() const { return w_; }
    Precision precision() const { return precision_; }
    bool fp32_split() const { return fp32_split_; }           // fp32 mode with the bf16x6 kernels in the search
    // The fp32 blob was (re)written in place (RCCL broadcast, EngineWeightsUpdated): rebuild the half / fragment-major mirrors.
    // adopt_act_scales (fp8 mode): take these per-step activation scales (the weight owner that calibrated them) instead of running
    // the calibration pass again -- receivers of the weight broadcast hold the same weights on the same hardware.
    void WeightsArrived(const std::vector<float>* adopt_act_scales = nullptr);
    std::vector<float> f8_act_scales() const { return w_->act_scale; }
    // FNV-1a of every mirror derived from the fp32 blob as it sits in HBM (half / fragment-major / Winograd U / bf16x6 / e4m3 + scales):
    // a replica filled by the weight broadcast must end up with the primary's derived data too.  Call with the lane held.
    std::vector<std::pair<std::string, uint64_t>> MirrorChecksums();
    // Synchronous copies / fills on THIS model's (non-blocking) stream.  The legacy-stream forms (hipMemcpy, hipMemset) are refused by the runtime
    // while any other thread captures a graph ("would make the legacy stream depend on a capturing stream"): replicas of one model load, tune and
    // capture concurrently, so nothing in the engine may touch the legacy stream.
    void CopySync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, const char* what);
    void ZeroSync(void* dst, size_t bytes, const char* what);
    size_t weight_bytes() const { return w_->weight_floats * sizeof(float); }
    size_t device_bytes() const { return device_bytes_ + w_->device_bytes.load(); }
    PlanInstance* current() { return current_; }
    // counters of the pipelined host path (observability / tests)
    int64_t pipelined_calls() const { return pipelined_calls_; }
    int last_chunks() const { return last_chunks_; }
    int last_head_steps() const { return last_head_steps_; }
    double last_forward_ms() const { return last_forward_ms_; }
    int max_inflight_replays() const { return max_inflight_replays_; }

private:
    void RunSteps(PlanInstance& pi, size_t first, size_t last, std::vector<hipEvent_t>* events);
    void BuildInstance(PlanInstance& pi, const std::vector<std::vector<int64_t>>& shapes);
    void FreeInstance(PlanInstance& pi);
    void Capture(PlanInstance& pi, size_t first, size_t last, hipGraphExec_t* exec);
    // Exhaustive (tile, split-K) search per distinct conv shape, timed with HIP events on the model's stream; the
    // MI355X counterpart of the reference's cudnn_conv_algo_search = Exhaustive (model.cpp:886).  Only steps [0, nsteps).
    void Autotune(PlanInstance& pi, size_t nsteps, bool allow_search);
    void SaveTuneCache();
    void EnsurePipeline(PlanInstance& pi, bool allow_tune);
    void AllocInstance(PlanInstance& pi);
    void RefreshGraphs(PlanInstance& pi);   // (re)capture the graphs of `pi` when they are missing or older than the current fp8 scales
    void PrepareF8(const std::vector<float>* adopt_act_scales = nullptr);       // quantise the conv weights, calibrate the activation scales, derive the epilogue multipliers (once per DeviceWeights)
    void LaunchStep(const PlanInstance& pi, const Step& s, hipStream_t stream);
    ConvArgs MakeConvArgs(const PlanInstance& pi, const Step& s) const;
    bool MakeBlockArgs(const PlanInstance& pi, const Step& s, DenseBlockArgs* out) const;   // false: the step is not a well-formed dense-block chain

    Env env_;                            // the lane's switches, read once in the constructor
    std::shared_ptr<const OnnxModel> model_;
    int device_ = 0;
    hipStream_t stream_ = nullptr;       // compute
    hipStream_t copy_stream_ = nullptr;  // H2D of the pipelined host path
    std::vector<hipEvent_t> h2d_events_;
    hipEvent_t t0_event_ = nullptr, t1_event_ = nullptr;   // device time of the last host-path forward
    std::shared_ptr<DeviceWeights> w_;
    bool owns_weights_ = false;
    bool upload_weights_ = true;
    bool fp32_split_ = false;
    Precision precision_ = Precision::F32;
    size_t device_bytes_ = 0;
    float u8_scale_ = 1.0f / 255.0f, u8_bias_ = 0.0f;
    bool use_graph_ = true;
    bool autotune_ = true;
    bool tune_on_demand_ = false;      // IE_TUNE_ON_DEMAND=1: search on the request path too (round-1 behaviour)
    int pipeline_chunks_ = 2;          // IE_PIPELINE_CHUNKS (0/1 = off); measured: 2 ranges beat 4 and 8 (per-range launches cost more than they hide)
    int pipeline_head_ = -1;           // IE_PIPELINE_HEAD: steps run per chunk (-1 = modelled)
    bool two_pass_splitk_ = true;      // IE_SPLITK_IN_LAUNCH=1 selects the in-launch combine instead of the reduce kernel
    std::map<std::vector<int64_t>, std::unique_ptr<PlanInstance>> plans_;
    // Least-recently-used order of the plan keys (front = oldest).  A server that sees many distinct batch sizes would otherwise keep
    // one set of activation buffers + one hipGraph per size forever; beyond IE_MAX_PLANS (default 8) the oldest instance is freed
    // (its kernel choices stay in the tune cache, so re-creating it is cheap).
    std::vector<std::vector<int64_t>> lru_;
    size_t max_plans_ = 8;
    PlanInstance* current_ = nullptr;
    void* pinned_ = nullptr;           // pinned host staging for D2H results
    size_t pinned_bytes_ = 0;
    int64_t pipelined_calls_ = 0;
    int last_chunks_ = 1, last_head_steps_ = 0;
    double last_forward_ms_ = 0;
    int max_inflight_replays_ = 0;      // IE_MAX_INFLIGHT_REPLAYS (0 = unlimited)
};

// Device queries behind IsCudaAvailable / GetDeviceCount / GetDeviceInfo / GetMemoryInfo
// (reference: inference_engine/src/cuda_utils.cu:17-57, 152-176).
int HipDeviceCount();
std::string HipDeviceInfo(int device_id);
bool HipMemoryInfo(int device_id, size_t* total, size_t* free_b);

}  // namespace ie
