// Device-side model: weights resident in HBM, planned activation buffers, one hipGraph per input shape.
//
// MI355X-native replacement for the Ort::Session the reference owns per model
// (inference_engine/src/model.cpp:706-714, created at :847, run at :1264-1270).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "onnx_reader.h"
#include "plan.h"

namespace ie {

struct StepTiming {
    std::string name, kernel;
    double ms = 0, flops = 0, bytes = 0;
};

struct PlanInstance {
    Plan plan;
    std::vector<float*> buffers;       // device activation buffers (plan.buffer_floats); aliased I/O entries are not owned
    std::vector<char> owned;           // buffers[i] was hipMalloc'ed by this instance
    // Sub-batch streams: a batch of B images can run as S independent sub-batches of B/S on S streams inside ONE
    // hipGraph (fork/join).  Layers whose output grid cannot fill 256 CUs (dense blocks 3-4 at batch 32) are latency
    // bound, so two half-batches in flight together use the idle CUs.  The parent instance then owns only the
    // full-size input/output buffers; each sub instance owns its activations and aliases slices of the parent's I/O.
    std::vector<std::unique_ptr<PlanInstance>> subs;
    hipStream_t stream = nullptr;      // stream this instance's kernels are enqueued on
    hipEvent_t done = nullptr;         // join event (sub instances on side streams)
    hipEvent_t fork = nullptr;         // parent only
    float* workspace = nullptr;        // split-K slabs
    int64_t workspace_floats = 0;
    int* counters = nullptr;           // split-K arrival counters (zero between launches)
    hipGraphExec_t graph_exec = nullptr;
    bool graph_ready = false;
    std::vector<void*> u8_stage;       // per graph input: device staging for UINT8 payloads (allocated on first use)
};

// A few helper threads that split big host memcpys (caller buffer -> pinned staging).  One thread moves ~10 GB/s, the
// PCIe Gen5 link ~55 GB/s: without help the CPU copy, not the DMA, bounds ModelInfer's H2D stage.
class CopyPool {
public:
    explicit CopyPool(int helpers);
    ~CopyPool();
    void Copy(void* dst, const void* src, size_t n);
private:
    void Worker(int idx);
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t n_ = 0;
    uint64_t generation_ = 0;
    int pending_ = 0;
    bool stop_ = false;
};

class DeviceModel {
public:
    // Parses nothing itself: takes the decoded graph. Throws std::runtime_error (never falls back to a CPU path).
    DeviceModel(std::shared_ptr<const OnnxModel> model, int device_id, Precision precision = Precision::F32);
    ~DeviceModel();
    DeviceModel(const DeviceModel&) = delete;
    DeviceModel& operator=(const DeviceModel&) = delete;

    // Get (or build: plan + allocate + capture) the instance for these input shapes.
    PlanInstance& Prepare(const std::vector<std::vector<int64_t>>& shapes);
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
    // Gather/scatter form used by the dynamic batcher: several callers' row blocks land at byte offsets of one device batch.
    // Input segment: `have` valid bytes at `host`, zero-extended to `need` bytes, placed at `dev_off` of input k.
    // Output segment: min(cap, need) bytes from `dev_off` of output j copied to `host`, the rest of `cap` zero-filled.
    struct InSeg { const void* host; size_t have, need, dev_off; bool u8 = false; };   // u8: have/need/dev_off count bytes = elements
    struct OutSeg { void* host; size_t cap, need, dev_off; };
    void InferHostSegments(PlanInstance& pi, const std::vector<std::vector<InSeg>>& in, const std::vector<std::vector<OutSeg>>& out);

    hipStream_t stream() const { return stream_; }
    int device() const { return device_; }
    float* weights() const { return d_weights_; }
    Precision precision() const { return precision_; }
    // fp16 mode keeps a half mirror of the fp32 weight blob; call after the blob was rewritten in place (RCCL broadcast).
    void RefreshHalfWeights();
    size_t weight_bytes() const { return weight_floats_ * sizeof(float); }
    size_t device_bytes() const { return device_bytes_; }
    std::mutex& mutex() { return mu_; }
    PlanInstance* current() { return current_; }

private:
    void RunSteps(PlanInstance& pi, std::vector<StepTiming>* timings, std::vector<hipEvent_t>* events);
    void BuildInstance(PlanInstance& pi, const std::vector<std::vector<int64_t>>& shapes, bool io_only);
    void FreeInstance(PlanInstance& pi);
    // Exhaustive (tile, split-K) search per distinct conv shape, timed with HIP events on the model's stream; the
    // MI355X counterpart of the reference's cudnn_conv_algo_search = Exhaustive (model.cpp:886).
    void Autotune(PlanInstance& pi);
    void LaunchStep(const PlanInstance& pi, const Step& s, hipStream_t stream);
    ConvArgs MakeConvArgs(const PlanInstance& pi, const Step& s) const;

    std::shared_ptr<const OnnxModel> model_;
    int device_ = 0;
    hipStream_t stream_ = nullptr;
    float* d_weights_ = nullptr;
    void* d_weights16_ = nullptr;      // fp16 mode: the same blob as halfs, same element offsets
    float* d_weights_frag_ = nullptr;  // fp32 mode: conv weights again in MFMA-fragment order at the same offsets (window kernels)
    struct FragRegion { int64_t w_off; int cout, kk, cin; };
    std::vector<FragRegion> frag_regions_;
    Precision precision_ = Precision::F32;
    size_t weight_floats_ = 0;
    size_t device_bytes_ = 0;
    float u8_scale_ = 1.0f / 255.0f, u8_bias_ = 0.0f;
    bool use_graph_ = true;
    bool autotune_ = true;
    int sub_streams_ = 1;              // IE_STREAMS: sub-batches run concurrently per forward
    std::vector<hipStream_t> side_streams_;
    bool two_pass_splitk_ = true;      // IE_SPLITK_IN_LAUNCH=1 selects the in-launch combine instead of the reduce kernel
    std::map<std::vector<int64_t>, std::pair<int, int>> tune_cache_;   // conv signature -> (tile, splitk)
    std::map<std::vector<int64_t>, std::unique_ptr<PlanInstance>> plans_;
    // Least-recently-used order of the plan keys (front = oldest).  A server that sees many distinct batch sizes would otherwise keep
    // one set of activation buffers + one hipGraph per size forever; beyond IE_MAX_PLANS (default 8) the oldest instance is freed
    // (its autotune results stay in tune_cache_, so re-creating it is cheap).
    std::vector<std::vector<int64_t>> lru_;
    size_t max_plans_ = 8;
    PlanInstance* current_ = nullptr;
    std::unique_ptr<CopyPool> copy_pool_;
    void* pinned_ = nullptr;           // pinned host staging ring for H2D/D2H
    size_t pinned_bytes_ = 0;
    std::mutex mu_;
};

// Device queries behind IsCudaAvailable / GetDeviceCount / GetDeviceInfo / GetMemoryInfo
// (reference: inference_engine/src/cuda_utils.cu:17-57, 152-176).
int HipDeviceCount();
std::string HipDeviceInfo(int device_id);
bool HipMemoryInfo(int device_id, size_t* total, size_t* free_b);

}  // namespace ie
