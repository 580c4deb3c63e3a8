// C ABI of libinference_engine.so (declared in include/inference_bridge.h, include/inference_engine_ext.h).
//
// Behavioural mirror of the reference bridge (inference_engine/src/inference_bridge.cpp) and of the
// Model/ModelImpl plumbing it drives (inference_engine/src/model.cpp:503-613, 734-794, 1158-1328), written
// fresh for the MI355X engine.  Deliberate differences from the reference, all listed in DESIGN.md:
//   * thread-safe (registry mutex, per-model mutex, shared ownership so an unload cannot free a model under
//     an in-flight ModelInfer; the reference has no locks at all)
//   * graph input/output names come from the ONNX graph, not the hard-coded {"input"}/{"output"}
//     (model_repository.cpp:143-144) that makes densenet_onnx unservable in the reference
//   * ModelInfer never writes more dims than the caller's array holds and zero-fills the unused tail of an
//     output buffer (the reference overflows / leaves it uninitialised, inference_bridge.cpp:794-812)
//   * no CPU execution provider: without a HIP device Load fails loudly
#include <atomic>
#include <cctype>
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <shared_mutex>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/inference_bridge.h"
#include "../../include/inference_engine_ext.h"
#include <rccl/rccl.h>

#include "config.h"
#include "env.h"
#include "executor.h"
#include "kernels.h"
#include "onnx_reader.h"
#include "plan.h"
#include "repository.h"

static_assert(sizeof(Shape) == 16, "Shape layout");
static_assert(sizeof(TensorData) == 48 && offsetof(TensorData, data) == 32 && offsetof(TensorData, data_size) == 40, "TensorData layout");
static_assert(sizeof(ModelConfig) == 64 && offsetof(ModelConfig, dynamic_batching) == 56, "ModelConfig layout");
static_assert(sizeof(ModelMetadata) == 72 && offsetof(ModelMetadata, load_time_ns) == 64, "ModelMetadata layout");
static_assert(sizeof(ModelStats) == 32, "ModelStats layout");
static_assert(sizeof(CudaMemoryInfo) == 24, "CudaMemoryInfo layout");

namespace {

// Optional ROCTX ranges: one range per ModelInfer call, named after the model, when IE_ROCTX=1 and the ROCm
// marker library is present (`rocprofv3 --marker-trace` then shows requests next to the kernels).  Loaded lazily with dlopen so
// the engine keeps libamdhip64 as its only link-time dependency.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const ie::Env env = ie::Env::Read();
        const char* e = env.get("IE_ROCTX");
        if (!e || e[0] != '1') return;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx& roctx() { static Roctx r; return r; }
struct RoctxRange {
    bool on;
    explicit RoctxRange(const std::string& name) : on(roctx().push != nullptr) { if (on) roctx().push(name.c_str()); }
    ~RoctxRange() { if (on) roctx().pop(); }
};


char* dup_cstr(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}
void set_error(ErrorMessage* error, const std::string& msg) {
    if (error) *error = dup_cstr(msg);
}

// A model's execution lanes (ie::DeviceModel objects) are used by one host thread at a time.  Requests that run on ONE lane take any
// free one (config.json "instance_count" lanes on the primary device + the shard replicas); a sharded request takes the first
// `n` lanes together and waits for them to drain first.
struct LanePool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> busy;
    int exclusive_waiters = 0;
    int in_flight = 0, max_in_flight = 0;
    void Reset(size_t n) { std::lock_guard<std::mutex> g(mu); busy.assign(n, 0); exclusive_waiters = 0; in_flight = 0; }
    // prefer the highest-numbered free lane: extra lanes first, so lane 0 (EnginePrepare / EngineRunPrepared) stays free longest
    int AcquireAny() {
        std::unique_lock<std::mutex> lk(mu);
        int k = -1;
        cv.wait(lk, [&] {
            if (exclusive_waiters > 0 || busy.empty()) return busy.empty();
            for (int i = int(busy.size()) - 1; i >= 0; --i) if (!busy[size_t(i)]) { k = i; return true; }
            return false;
        });
        if (k < 0) return -1;
        busy[size_t(k)] = 1;
        max_in_flight = std::max(max_in_flight, ++in_flight);
        return k;
    }
    void AcquireOne(int k) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return size_t(k) < busy.size() && !busy[size_t(k)]; });
        busy[size_t(k)] = 1;
        max_in_flight = std::max(max_in_flight, ++in_flight);
    }
    void AcquireRange(int n) {
        std::unique_lock<std::mutex> lk(mu);
        ++exclusive_waiters;
        cv.wait(lk, [&] { for (int i = 0; i < n; ++i) if (busy[size_t(i)]) return false; return true; });
        --exclusive_waiters;
        for (int i = 0; i < n; ++i) busy[size_t(i)] = 1;
        max_in_flight = std::max(max_in_flight, ++in_flight);
    }
    void Release(int first, int n) {
        { std::lock_guard<std::mutex> g(mu); for (int i = first; i < first + n; ++i) busy[size_t(i)] = 0; --in_flight; }
        cv.notify_all();
    }
};

// Persistent helper threads, one per shard replica: a sharded ModelInfer hands slice k to thread k-1 and runs slice 0 itself.
// (Round 1 spawned and joined std::threads per call; at 4 images per GPU that churn was a first-order cost.)
class WorkerPool {
public:
    ~WorkerPool() { Stop(); }
    void Start(int n) {
        Stop();
        for (int i = 0; i < n; ++i) {
            ws_.push_back(std::make_unique<W>());
            W* w = ws_.back().get();
            w->th = std::thread([this, w] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(w->mu);
                        w->cv.wait(lk, [&] { return w->stop || w->has; });
                        if (w->stop) return;
                        job = std::move(w->job);
                        w->has = false;
                    }
                    job();
                    { std::lock_guard<std::mutex> g(dmu_); --pending_; }
                    dcv_.notify_all();
                }
            });
        }
    }
    void Stop() {
        for (auto& w : ws_) { { std::lock_guard<std::mutex> g(w->mu); w->stop = true; } w->cv.notify_all(); }
        for (auto& w : ws_) if (w->th.joinable()) w->th.join();
        ws_.clear();
    }
    size_t size() const { return ws_.size(); }
    void Submit(size_t k, std::function<void()> fn) {
        { std::lock_guard<std::mutex> g(dmu_); ++pending_; }
        W* w = ws_.at(k).get();
        { std::lock_guard<std::mutex> g(w->mu); w->job = std::move(fn); w->has = true; }
        w->cv.notify_one();
    }
    void Wait() { std::unique_lock<std::mutex> lk(dmu_); dcv_.wait(lk, [&] { return pending_ == 0; }); }
private:
    struct W { std::thread th; std::mutex mu; std::condition_variable cv; std::function<void()> job; bool has = false, stop = false; };
    std::vector<std::unique_ptr<W>> ws_;
    std::mutex dmu_;
    std::condition_variable dcv_;
    int pending_ = 0;
};

uint64_t fnv1a64(const void* data, size_t n) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

struct ModelObj {
    std::string path;
    ModelType type = MODEL_UNKNOWN;
    DeviceType device = DEVICE_GPU;
    int device_id = 0;
    std::string name, version;
    std::vector<std::string> input_names, output_names;   // config names until Load replaces them with the graph's

    std::shared_mutex life;        // Load / Unload exclusive; everything that touches the lanes holds it shared
    std::atomic<bool> loaded{false};
    std::mutex err_mu;
    std::string last_error;
    void SetError(const std::string& m) { std::lock_guard<std::mutex> g(err_mu); last_error = m; }
    std::string GetError() { std::lock_guard<std::mutex> g(err_mu); return last_error; }
    std::shared_ptr<const ie::OnnxModel> onnx;
    ie::ModelInfo info;
    ie::EngineConfig cfg;          // config.json, parsed once at load
    // Execution lanes.  lanes[0] is the primary.  lanes[1 .. num_shards) are the shard replicas of the in-process batch sharding
    // (single process, all GPUs of the node; IE_GPUS=<n> / config.json "gpus": n -> devices device_id .. device_id+n-1, or
    // IE_SHARD_DEVICES=<id,id,...>, ids may repeat, which is how the single-GPU tests exercise the logic).  lanes[num_shards ..) are
    // extra lanes on the primary device (config.json / ModelConfig "instance_count"): concurrent requests run side by side.
    // A replica on another device owns its weights and receives the packed blob by ONE ncclBroadcast at load; lanes on a device that
    // already holds the blob share it.
    std::vector<std::unique_ptr<ie::DeviceModel>> lanes;
    int num_shards = 1;
    LanePool pool;
    WorkerPool workers;
    struct RcclInfo { bool used = false; int ranks = 0; size_t bytes = 0; double init_ms = 0, bcast_ms = 0; int owners = 1; } rccl;
    float u8_scale = 1.0f / 255.0f, u8_bias = 0.0f;
    int64_t load_time_ns = 0;
    std::atomic<int64_t> inference_count{0}, total_ns{0}, last_ns{0};
    std::atomic<size_t> memory_usage_bytes{0};
    // device-side accounting for the observability string (ModelGetMetadata.description): HIP-event time of the forwards and the
    // planner's algorithmic FLOPs / bytes of what ran
    std::mutex acct_mu;
    double acct_ms = 0, acct_flops = 0, acct_bytes = 0;
    int64_t acct_forwards = 0, acct_images = 0;

    // ---- dynamic request batcher: honours the reference's inert max_batch_size / dynamic_batching fields
    // (model.h:63,70-71).  Concurrent ModelInfer calls (one per gin goroutine) are coalesced into ONE device batch and the
    // results are scattered back per caller.  Enabled by IE_DYNAMIC_BATCH=<max rows> or config.json
    // {"dynamic_batching": true, "max_batch_size": N}; only for graphs whose inputs/outputs have a symbolic batch axis.
    struct Pending {
        std::vector<const void*> in_ptr;
        std::vector<size_t> in_bytes;
        std::vector<char> in_u8;          // 1 = UINT8 payload for a FLOAT32 graph input (converted on the device)
        std::vector<std::vector<int64_t>> shapes;
        TensorData* outputs = nullptr;
        int num_outputs = 0;
        int64_t rows = 0;
        bool done = false, ok = false;
        std::string err;
    };
    int cfg_max_batch = 0;        // from ModelCreate's ModelConfig {dynamic_batching, max_batch_size}
    int cfg_instances = 0;        // from ModelCreate's ModelConfig.instance_count
    int max_batch = 0;            // 0/1 = batching off
    int batch_window_us = 200;
    bool batchable = false;       // set at Load: symbolic batch axis on every graph input and output
    std::mutex bmu;
    std::condition_variable bcv;
    std::deque<Pending*> queue;
    bool leader_active = false;
    std::atomic<int64_t> device_batches{0}, coalesced_requests{0}, shard_calls{0};

    bool Load();      // model.cpp:503-548 + 825-871
    void Unload();    // model.cpp:618-648
    void Execute(std::vector<Pending*>& batch);   // runs one device batch for these callers
    void RunBatched(Pending& req);                // leader/follower coalescing
    void BroadcastWeights();                      // RCCL: primary's packed blob -> every other weight owner
    void Account(ie::DeviceModel& d, const ie::PlanInstance& pi);
    using Segs = std::pair<std::vector<std::vector<ie::DeviceModel::InSeg>>, std::vector<std::vector<ie::DeviceModel::OutSeg>>>;
    std::vector<ie::IoDesc> RunOnLanes(const std::vector<std::vector<int64_t>>& shapes, int64_t rows, bool allow_shard, const Segs& segs, bool* sharded);
};

#define NCCL_OK(call)                                                                                             \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess) throw std::runtime_error(std::string("RCCL error in " #call ": ") + ncclGetErrorString(r_)); \
    } while (0)

// One communicator per distinct device (ncclCommInitAll, single process), one in-place ncclBroadcast of the packed fp32 blob from
// the primary to every other device's owner inside a group call, then each receiver rebuilds its derived mirrors.  This is the
// only collective of the whole path; nothing is exchanged per inference.  The reference has no counterpart: it
// hard-codes device 0 (inference_bridge.cpp:346-347).
void ModelObj::BroadcastWeights() {
    std::vector<ie::DeviceModel*> owners;          // lanes that own a weight allocation of their own, primary first
    for (auto& l : lanes) {
        bool first = true;
        for (auto* o : owners) if (o->shared_weights() == l->shared_weights()) first = false;
        if (first) owners.push_back(l.get());
    }
    rccl.owners = int(owners.size());
    if (owners.size() < 2) return;
    std::vector<int> devs;                         // distinct devices, the primary's first = rank 0 = root
    std::vector<ie::DeviceModel*> rank_owner;
    for (auto* o : owners)
        if (std::find(devs.begin(), devs.end(), o->device()) == devs.end()) { devs.push_back(o->device()); rank_owner.push_back(o); }
    const size_t count = owners[0]->weight_bytes() / sizeof(float);
    std::vector<ncclComm_t> comms(devs.size(), nullptr);
    auto t0 = std::chrono::steady_clock::now();
    NCCL_OK(ncclCommInitAll(comms.data(), int(devs.size()), devs.data()));
    auto t1 = std::chrono::steady_clock::now();
    try {
        if (devs.size() > 1) {
            NCCL_OK(ncclGroupStart());
            for (size_t r = 0; r < devs.size(); ++r) {
                if (hipSetDevice(devs[r]) != hipSuccess) throw std::runtime_error("hipSetDevice failed during the weight broadcast");
                NCCL_OK(ncclBroadcast(rank_owner[r]->weights(), rank_owner[r]->weights(), count, ncclFloat, 0, comms[r], rank_owner[r]->stream()));
            }
            NCCL_OK(ncclGroupEnd());
            for (auto* o : rank_owner) o->Synchronize();
        }
        // further owners on a device that already holds the blob (IE_SHARD_PRIVATE_WEIGHTS=1, how the one-GPU box moves real bytes
        // through RCCL): with one rank an out-of-place broadcast copies send -> recv; with more ranks a device-to-device copy does
        for (auto* o : owners) {
            if (std::find(rank_owner.begin(), rank_owner.end(), o) != rank_owner.end()) continue;
            const size_t r = size_t(std::find(devs.begin(), devs.end(), o->device()) - devs.begin());
            if (hipSetDevice(devs[r]) != hipSuccess) throw std::runtime_error("hipSetDevice failed during the weight broadcast");
            if (devs.size() == 1) NCCL_OK(ncclBroadcast(rank_owner[r]->weights(), o->weights(), count, ncclFloat, 0, comms[r], o->stream()));
            else if (hipMemcpyAsync(o->weights(), rank_owner[r]->weights(), count * sizeof(float), hipMemcpyDeviceToDevice, o->stream()) != hipSuccess)
                throw std::runtime_error("device-to-device weight copy failed");
            o->Synchronize();
        }
    } catch (...) {
        for (auto c : comms) if (c) (void)ncclCommDestroy(c);
        throw;
    }
    auto t2 = std::chrono::steady_clock::now();
    for (auto c : comms) if (c) (void)ncclCommDestroy(c);
    // receivers rebuild their derived mirrors; in fp8 mode they adopt the primary's calibrated activation scales (same weights, same
    // hardware: re-running the calibration pass on every receiver would be 8x redundant work at load)
    const std::vector<float> scales = owners[0]->f8_act_scales();
    for (size_t i = 1; i < owners.size(); ++i) owners[i]->WeightsArrived(scales.empty() ? nullptr : &scales);
    (void)hipSetDevice(device_id);
    rccl.used = true;
    rccl.ranks = int(devs.size());
    rccl.bytes = count * sizeof(float);
    rccl.init_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    rccl.bcast_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
}

bool ModelObj::Load() {
    std::unique_lock<std::shared_mutex> g(life);
    auto t0 = std::chrono::steady_clock::now();
    std::error_code ec;
    if (!std::filesystem::exists(path, ec)) {
        SetError("Model file not found: " + path);
        return false;
    }
    bool ok = false;
    switch (type) {
        case MODEL_TENSORFLOW: SetError("TensorFlow model loading not implemented"); break;
        case MODEL_TENSORRT: SetError("TensorRT model loading not implemented"); break;
        case MODEL_PYTORCH: SetError("PyTorch model loading not implemented"); break;
        case MODEL_CUSTOM: SetError("Custom model loading not implemented"); break;
        case MODEL_ONNX: {
            try {
                const ie::Env env = ie::Env::Read();       // this load's switches, read once
                const std::string file = path + "/model.onnx";
                if (!std::filesystem::exists(file, ec)) {
                    SetError("ONNX model file not found: " + file);
                    break;
                }
                if (device != DEVICE_GPU) {
                    SetError("DEVICE_CPU execution is not provided by the MI355X engine (a HIP device is required)");
                    break;
                }
                auto parsed = std::make_shared<ie::OnnxModel>(ie::LoadOnnxFile(file));
                ie::ModelInfo inf = ie::DescribeModel(*parsed);
                ie::EngineConfig conf = ie::LoadEngineConfig(path);       // read ONCE; a malformed file is a load error
                // Precision: IE_PRECISION=fp16|fp32, else config.json {"precision": "fp16"}; default fp32 (the reference's
                // ONNX Runtime session computes in the model's own fp32).
                ie::Precision prec = ie::Precision::F32;
                {
                    std::string want = conf.precision;
                    if (const char* e = env.get("IE_PRECISION")) want = e;
                    for (auto& ch : want) ch = char(std::tolower(static_cast<unsigned char>(ch)));
                    if (want == "fp16" || want == "f16" || want == "half" || want == "float16") prec = ie::Precision::F16;
                    else if (want == "fp8" || want == "f8" || want == "e4m3" || want == "float8") prec = ie::Precision::F8;
                    else if (!want.empty() && want != "fp32" && want != "f32" && want != "float32" && want != "float") {
                        SetError("ONNX model loading error: unsupported precision '" + want + "' (fp32, fp16 or fp8)");
                        break;
                    }
                }
                // UINT8 ingest transform x * scale + bias: config.json "uint8_scale" / "uint8_bias" (default 1/255, 0: the reference
                // client's /255 convention, client/test_client.py:189)
                u8_scale = conf.uint8_scale;
                u8_bias = conf.uint8_bias;
                ie::DeviceModelOptions opt;
                opt.precision = prec;
                opt.fp32_split = conf.fp32_split;
                opt.tune_cache_path = path + "/.ie_tune." + (prec == ie::Precision::F16 ? "fp16" : (prec == ie::Precision::F8 ? "fp8" : "fp32")) + ".txt";
                auto primary = std::make_unique<ie::DeviceModel>(parsed, device_id, opt);
                primary->SetU8Transform(u8_scale, u8_bias);
                // Plan + tune at load, off the request path (like Ort::Session's constructor, which also rejects unsupported graphs
                // here): the config's declared shape (symbolic dims -> 1) and every batch size of "tune_batches" / IE_TUNE_BATCHES.
                std::vector<std::vector<int64_t>> shapes;
                for (size_t i = 0; i < inf.inputs.size(); ++i) {
                    std::vector<int64_t> s = inf.inputs[i].dims;
                    int64_t cfg_batch = 1;
                    for (const auto& ic : conf.inputs)
                        if (ic.name == inf.inputs[i].name && ic.shape.size() == s.size() && !ic.shape.empty() && ic.shape[0] > 0) cfg_batch = ic.shape[0];
                    for (size_t k = 0; k < s.size(); ++k) if (s[k] <= 0) s[k] = (k == 0 ? cfg_batch : 1);
                    shapes.push_back(s);
                }
                primary->Prepare(shapes, true);
                bool symbolic_batch = !inf.inputs.empty();
                for (auto& vi : inf.inputs) if (vi.dims.empty() || vi.dims[0] > 0) symbolic_batch = false;
                for (auto& vi : inf.outputs) if (vi.dims.empty() || vi.dims[0] > 0) symbolic_batch = false;
                {
                    std::vector<int64_t> tb = conf.tune_batches;
                    if (const char* e = env.get("IE_TUNE_BATCHES")) {
                        tb.clear();
                        std::stringstream ss(e);
                        std::string tok;
                        while (std::getline(ss, tok, ',')) if (!tok.empty()) tb.push_back(std::atoll(tok.c_str()));
                    }
                    for (int64_t b : tb) {
                        if (!symbolic_batch || b <= 0 || b > 65536) continue;
                        std::vector<std::vector<int64_t>> sh = shapes;
                        for (auto& s : sh) s[0] = b;
                        primary->Prepare(sh, true);
                    }
                }
                // ---- shard replicas and extra lanes ----
                std::vector<int> shard_ids;                       // devices of lanes[1 .. num_shards)
                if (const char* e = env.get("IE_SHARD_DEVICES")) {
                    std::stringstream ss(e);
                    std::string tok;
                    while (std::getline(ss, tok, ',')) if (!tok.empty()) shard_ids.push_back(std::atoi(tok.c_str()));
                    if (!shard_ids.empty()) shard_ids.erase(shard_ids.begin());           // the first id is the primary's slice
                } else {
                    int n = conf.gpus > 0 ? conf.gpus : 1;
                    if (const char* e = env.get("IE_GPUS")) n = std::atoi(e);
                    const int have = ie::HipDeviceCount();
                    for (int k = 1; k < n && device_id + k < have; ++k) shard_ids.push_back(device_id + k);
                }
                if (!symbolic_batch) shard_ids.clear();           // a fixed-batch graph cannot be sliced
                int instances = conf.instance_count > 0 ? conf.instance_count : (cfg_instances > 0 ? cfg_instances : 1);
                if (const char* e = env.get("IE_INSTANCES")) instances = std::atoi(e);
                instances = std::max(1, std::min(instances, 16));
                const bool private_weights = [&] { const char* e = env.get("IE_SHARD_PRIVATE_WEIGHTS"); return e && e[0] == '1'; }();
                std::vector<std::unique_ptr<ie::DeviceModel>> built;
                built.push_back(std::move(primary));
                struct Spec { int dev; bool shard; };
                std::vector<Spec> specs;
                for (int id : shard_ids) specs.push_back({id, true});
                for (int k = 1; k < instances; ++k) specs.push_back({device_id, false});
                // constructors run here (they decide who shares whose weights); planning, allocation and graph capture of the
                // replicas then run in parallel on the shard worker threads
                for (const Spec& sp : specs) {
                    ie::DeviceModelOptions o = opt;
                    o.tune_cache_path.clear();
                    ie::DeviceModel* holder = nullptr;
                    if (!(sp.shard && private_weights))
                        for (auto& l : built) if (l->device() == sp.dev) { holder = l.get(); break; }
                    if (holder) o.share = holder->shared_weights();
                    else o.upload_weights = false;                                    // filled by the RCCL broadcast below
                    auto r = std::make_unique<ie::DeviceModel>(parsed, sp.dev, o);
                    r->SetU8Transform(u8_scale, u8_bias);
                    if (!holder) {                                                    // same hardware: adopt the primary's kernel choices
                        auto& src = *built[0]->shared_weights();
                        auto& dst = *r->shared_weights();
                        std::lock_guard<std::mutex> g1(src.tune_mu);
                        dst.tune_cache = src.tune_cache;
                    }
                    built.push_back(std::move(r));
                }
                workers.Start(int(shard_ids.size()));
                if (built.size() > 1) {
                    std::vector<std::string> errs(built.size());
                    WorkerPool builders;
                    builders.Start(int(built.size()) - 1);
                    for (size_t k = 1; k < built.size(); ++k)
                        builders.Submit(k - 1, [&, k] {
                            try { built[k]->Prepare(shapes, false); } catch (const std::exception& e) { errs[k] = e.what(); }
                        });
                    builders.Wait();
                    builders.Stop();
                    for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
                }
                lanes = std::move(built);
                num_shards = int(shard_ids.size()) + 1;
                rccl = RcclInfo();
                BroadcastWeights();
                pool.Reset(lanes.size());
                input_names.clear();
                output_names.clear();
                for (auto& vi : inf.inputs) input_names.push_back(vi.name);
                for (auto& vi : inf.outputs) output_names.push_back(vi.name);
                memory_usage_bytes = inf.memory_usage_bytes;   // reference's estimate formula, model.cpp:979-1035
                batchable = symbolic_batch;
                {   // batching knobs: environment first, then config.json, then ModelCreate's ModelConfig
                    max_batch = cfg_max_batch;
                    if (conf.dynamic_batching && conf.max_batch_size > 1) max_batch = conf.max_batch_size;
                    if (const char* e = env.get("IE_DYNAMIC_BATCH")) max_batch = std::atoi(e);
                    if (conf.batch_window_us >= 0) batch_window_us = conf.batch_window_us;
                    if (const char* e = env.get("IE_BATCH_WINDOW_US")) batch_window_us = std::max(0, std::atoi(e));
                    if (max_batch > 4096) max_batch = 4096;
                }
                onnx = parsed;
                info = std::move(inf);
                cfg = std::move(conf);
                ok = true;
            } catch (const std::exception& e) {
                workers.Stop();
                lanes.clear();
                SetError(std::string("ONNX model loading error: ") + e.what());
            }
            break;
        }
        default: SetError("Unsupported model type"); return false;
    }
    load_time_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    loaded = ok;
    return ok;
}

void ModelObj::Unload() {
    std::unique_lock<std::shared_mutex> g(life);      // waits for every in-flight ModelInfer (they hold it shared)
    workers.Stop();
    lanes.clear();
    pool.Reset(0);
    onnx.reset();
    loaded = false;
}

}  // namespace

struct Model_t {
    std::shared_ptr<ModelObj> model;
};

struct InferenceManager_t {
    std::string repo_path;
    std::unique_ptr<ie::Repository> repo;
    std::mutex mu;
    std::unordered_map<std::string, std::shared_ptr<ModelObj>> models;   // keyed by name only (bridge:320)
};

extern "C" {

// ---- device queries ---------------------------------------------------------------------------------
bool IsCudaAvailable(void) { return ie::HipDeviceCount() > 0; }
int GetDeviceCount(void) { return ie::HipDeviceCount(); }
const char* GetDeviceInfo(int device_id) {
    try { return dup_cstr(ie::HipDeviceInfo(device_id)); } catch (...) { return dup_cstr("Unknown device"); }
}
CudaMemoryInfo GetMemoryInfo(int device_id) {
    CudaMemoryInfo m{0, 0, 0};
    size_t total = 0, fr = 0;
    if (ie::HipMemoryInfo(device_id, &total, &fr)) { m.total = total; m.free = fr; m.used = total - fr; }
    return m;
}

// ---- manager ------------------------------------------------------------------------------------------
InferenceManagerHandle InferenceInitialize(const char* model_repository_path) {
    try {
        auto* mgr = new InferenceManager_t();
        mgr->repo_path = model_repository_path ? model_repository_path : "";
        mgr->repo = std::make_unique<ie::Repository>(mgr->repo_path);
        mgr->repo->Scan();
        return mgr;
    } catch (const std::exception& e) {
        std::cerr << "Exception in InferenceInitialize: " << e.what() << std::endl;
        return nullptr;
    } catch (...) { return nullptr; }
}

void InferenceShutdown(InferenceManagerHandle handle) {
    try { delete handle; } catch (...) {}
}

bool InferenceLoadModel(InferenceManagerHandle handle, const char* model_name, const char* version, ErrorMessage* error) {
    if (!handle || !model_name) { set_error(error, "Invalid handle or model name"); return false; }
    try {
        const std::string name = model_name;
        handle->repo->Scan();   // pick up models added after InferenceInitialize (InferenceListModels rescans too)
        std::string ver = version ? version : handle->repo->LatestVersion(name);
        std::string path = handle->repo->ModelPath(name, ver);
        std::error_code ec;
        if (path.empty() || !std::filesystem::exists(path, ec)) { set_error(error, "Model path not found: " + path); return false; }
        std::shared_ptr<ModelObj> obj;
        {
            std::lock_guard<std::mutex> g(handle->mu);
            if (handle->models.count(name)) { set_error(error, "Model already loaded"); return false; }
            const std::string onnx_file = path + "/model.onnx";
            if (!std::filesystem::exists(onnx_file, ec)) { set_error(error, "ONNX file not found at: " + onnx_file); return false; }
            if (ie::Repository::DetectType(path) == ie::RepoModelType::Unknown) { set_error(error, "Unable to determine model type"); return false; }
            obj = std::make_shared<ModelObj>();
            obj->path = path;
            obj->type = MODEL_ONNX;
            obj->device = DEVICE_GPU;      // bridge:346-347: GPU, device 0
            obj->device_id = 0;
            if (const char* d = ie::Env::Read().get("IE_DEVICE_ID")) obj->device_id = std::atoi(d);
            obj->name = name;
            obj->version = ver.empty() ? handle->repo->LatestVersion(name) : ver;
            obj->input_names = {"input"};
            obj->output_names = {"output"};
            handle->models[name] = obj;    // reserve the name; concurrent loaders of the same name now fail fast
        }
        if (!obj->Load()) {
            const std::string msg = obj->GetError();
            std::lock_guard<std::mutex> g(handle->mu);
            handle->models.erase(name);
            set_error(error, msg);
            return false;
        }
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

bool InferenceUnloadModel(InferenceManagerHandle handle, const char* model_name, const char* /*version*/, ErrorMessage* error) {
    if (!handle || !model_name) { set_error(error, "Invalid handle or model name"); return false; }
    try {
        std::shared_ptr<ModelObj> obj;
        {
            std::lock_guard<std::mutex> g(handle->mu);
            auto it = handle->models.find(model_name);
            if (it == handle->models.end()) { set_error(error, "Model not found"); return false; }
            obj = it->second;
            handle->models.erase(it);
        }
        obj->Unload();   // waits for an in-flight ModelInfer; wrappers that still exist see "Model not loaded"
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

bool InferenceIsModelLoaded(InferenceManagerHandle handle, const char* model_name, const char* /*version*/) {
    if (!handle || !model_name) return false;
    try {
        std::lock_guard<std::mutex> g(handle->mu);
        auto it = handle->models.find(model_name);
        return it != handle->models.end() && it->second->loaded.load();
    } catch (...) { return false; }
}

char** InferenceListModels(InferenceManagerHandle handle, int* num_models) {
    if (!handle || !num_models) return nullptr;
    try {
        handle->repo->Scan();
        std::vector<std::string> names = handle->repo->Models();
        *num_models = int(names.size());
        if (names.empty()) return nullptr;
        char** out = static_cast<char**>(std::malloc(sizeof(char*) * names.size()));
        for (size_t i = 0; i < names.size(); ++i) out[i] = dup_cstr(names[i]);
        return out;
    } catch (...) { *num_models = 0; return nullptr; }
}

void InferenceFreeModelList(char** models, int num_models) {
    if (!models) return;
    for (int i = 0; i < num_models; ++i) std::free(models[i]);
    std::free(models);
}

// ---- model ----------------------------------------------------------------------------------------------
ModelHandle ModelCreate(const char* model_path, ModelType type, const ModelConfig* config, DeviceType device, int device_id,
                        ErrorMessage* error) {
    if (!model_path || !config) { set_error(error, "Invalid model path or configuration"); return nullptr; }
    try {
        auto obj = std::make_shared<ModelObj>();
        obj->path = model_path;
        obj->type = type;
        obj->device = device;
        obj->device_id = device_id;
        obj->name = config->name ? config->name : "";
        obj->version = config->version ? config->version : "1";
        if (config->dynamic_batching && config->max_batch_size > 1) obj->cfg_max_batch = config->max_batch_size;
        if (config->instance_count > 1) obj->cfg_instances = config->instance_count;      // model.h:63: carried, never read, by the reference
        for (int i = 0; i < config->num_inputs; ++i)
            if (config->input_names && config->input_names[i]) obj->input_names.push_back(config->input_names[i]);
        for (int i = 0; i < config->num_outputs; ++i)
            if (config->output_names && config->output_names[i]) obj->output_names.push_back(config->output_names[i]);
        return new Model_t{obj};
    } catch (const std::exception& e) { set_error(error, e.what()); return nullptr; }
    catch (...) { set_error(error, "unknown error"); return nullptr; }
}

void ModelDestroy(ModelHandle handle) {
    try { delete handle; } catch (...) {}
}

bool ModelLoad(ModelHandle handle, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return false; }
    try {
        bool ok = handle->model->Load();
        if (!ok) set_error(error, handle->model->GetError());
        return ok;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

bool ModelUnload(ModelHandle handle, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return false; }
    try { handle->model->Unload(); return true; }
    catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

bool ModelIsLoaded(ModelHandle handle) {
    if (!handle) return false;
    try { return handle->model->loaded.load(); } catch (...) { return false; }
}

bool ModelInfer(ModelHandle handle, const TensorData* inputs, int num_inputs, TensorData* outputs, int num_outputs,
                ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return false; }
    ModelObj& M = *handle->model;
    if (!M.loaded.load()) { set_error(error, "Model not loaded"); return false; }
    if (!inputs || num_inputs <= 0 || !outputs || num_outputs <= 0) { set_error(error, "Invalid parameters"); return false; }
    try {
        ModelObj::Pending req;
        bool batched = false;
        {
            std::shared_lock<std::shared_mutex> g(M.life);
            if (!M.loaded.load() || M.lanes.empty()) { set_error(error, "Model not loaded"); return false; }
            auto failv = [&](const std::string& msg) { M.SetError(msg); set_error(error, msg); return false; };
            // ---- ValidateInputs (model.cpp:734-794): count, then names ----
            const auto& gin = M.info.inputs;
            if (size_t(num_inputs) != gin.size())
                return failv("Expected " + std::to_string(gin.size()) + " inputs, got " + std::to_string(num_inputs));
            for (int i = 0; i < num_inputs; ++i) {
                const std::string nm = inputs[i].name ? inputs[i].name : "";
                bool known = false;
                for (auto& vi : gin) if (vi.name == nm) known = true;
                if (!known) return failv("Unexpected input name: " + nm);
            }
            // ---- InferONNX (model.cpp:1158-1328): order inputs by graph index ----
            req.in_ptr.assign(gin.size(), nullptr);
            req.in_bytes.assign(gin.size(), 0);
            req.in_u8.assign(gin.size(), 0);
            req.shapes.assign(gin.size(), {});
            req.outputs = outputs;
            req.num_outputs = num_outputs;
            std::vector<char> provided(gin.size(), 0);
            for (int i = 0; i < num_inputs && req.err.empty(); ++i) {
                const TensorData& t = inputs[i];
                const std::string nm = t.name ? t.name : "";
                for (size_t k = 0; k < gin.size(); ++k) {
                    if (gin[k].name != nm) continue;
                    // FLOAT32 as in the reference (bridge:744), plus UINT8 image bytes that the engine converts on the device
                    if (t.data_type != DATATYPE_FLOAT32 && t.data_type != DATATYPE_UINT8) { req.err = "Unsupported data type for input: " + nm; break; }
                    req.in_u8[k] = t.data_type == DATATYPE_UINT8 ? 1 : 0;
                    provided[k] = 1;
                    req.shapes[k].clear();
                    if (t.shape.dims && t.shape.num_dims > 0) req.shapes[k].assign(t.shape.dims, t.shape.dims + t.shape.num_dims);
                    req.in_ptr[k] = (t.data && t.data_size > 0) ? t.data : nullptr;
                    req.in_bytes[k] = req.in_ptr[k] ? t.data_size : 0;
                }
            }
            for (size_t k = 0; k < gin.size() && req.err.empty(); ++k)
                if (!provided[k]) req.err = "Required input tensor not provided: " + gin[k].name;
            // coalescing needs one common leading (batch) dimension and the declared ranks
            batched = req.err.empty() && M.batchable && M.max_batch > 1;
            if (batched) {
                req.rows = req.shapes[0].empty() ? 0 : req.shapes[0][0];
                for (size_t k = 0; k < gin.size(); ++k)
                    if (req.shapes[k].size() != gin[k].dims.size() || req.shapes[k][0] != req.rows) batched = false;
                if (req.rows <= 0 || req.rows >= M.max_batch) batched = false;
                for (char u : req.in_u8) if (u) batched = false;          // byte payloads are not coalesced
            }
            if (!batched) req.rows = 0;     // rows > 0 marks a request that may be padded / coalesced
        }
        // From here on the reference counts the call in its statistics even when it fails (model.cpp:572-612).
        const auto t0 = std::chrono::steady_clock::now();
        const RoctxRange range("ModelInfer:" + M.name);
        if (req.err.empty()) {
            if (batched) M.RunBatched(req);
            else { std::vector<ModelObj::Pending*> one{&req}; M.Execute(one); }
        }
        const int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        M.inference_count.fetch_add(1);
        M.total_ns.fetch_add(ns);
        M.last_ns.store(ns);
        if (!req.ok) {
            M.SetError(req.err);
            set_error(error, req.err);
            return false;
        }
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

}  // extern "C"

namespace {

// outputs by index, in graph-output order (bridge:787-813); never writes past the caller's dims array
void write_out_dims(TensorData* outputs, int num_outputs, const std::vector<ie::IoDesc>& odesc, int64_t rows) {
    for (int i = 0; i < num_outputs && size_t(i) < odesc.size(); ++i) {
        std::vector<int64_t> dims = odesc[size_t(i)].dims;
        if (rows > 0 && !dims.empty()) dims[0] = rows;
        const int cap = outputs[i].shape.dims ? outputs[i].shape.num_dims : 0;
        const int nd = int(dims.size());
        if (outputs[i].shape.dims) {
            const int nw = nd < cap ? nd : cap;
            for (int j = 0; j < nw; ++j) outputs[i].shape.dims[j] = dims[size_t(j)];
            outputs[i].shape.num_dims = nw;
        }
    }
}

// Device-side accounting behind ModelGetMetadata.description: what ran (planner FLOPs / bytes) and how long the device took.
void ModelObj::Account(ie::DeviceModel& d, const ie::PlanInstance& pi) {
    const double ms = d.last_forward_ms();
    if (ms <= 0) return;
    std::lock_guard<std::mutex> g(acct_mu);
    acct_ms += ms;
    acct_flops += pi.plan.total_flops;
    acct_bytes += pi.plan.total_bytes;
    acct_forwards += 1;
    acct_images += pi.plan.inputs.empty() || pi.plan.inputs[0].dims.empty() ? 0 : pi.plan.inputs[0].dims[0];
}

// Run one device batch described by gather/scatter segments over `rows` rows (rows == 0: the shapes are used as they are and the
// batch cannot be cut).  One lane, or -- allow_shard, num_shards > 1 and at least one row per replica -- contiguous row slices on
// all shard lanes at once (slice k on worker thread k-1, slice 0 on the calling thread).  Returns a COPY of the output descriptors
// of the plan that ran, taken while the lane is still held: once a lane is released any other request may take it and `Prepare` a
// new shape there, which can evict (free) this plan instance from the lane's bounded cache.
std::vector<ie::IoDesc> ModelObj::RunOnLanes(const std::vector<std::vector<int64_t>>& shapes, int64_t rows, bool allow_shard, const Segs& segs,
                                             bool* sharded) {
    const int S = num_shards;
    *sharded = false;
    if (!(allow_shard && S > 1 && rows >= S)) {
        const int k = pool.AcquireAny();
        if (k < 0) throw std::runtime_error("Model not loaded");
        try {
            ie::DeviceModel& D = *lanes[size_t(k)];
            ie::PlanInstance& pi = D.Prepare(shapes, false);
            D.InferHostSegments(pi, segs.first, segs.second);
            Account(D, pi);
            std::vector<ie::IoDesc> outs = pi.plan.outputs;
            pool.Release(k, 1);
            return outs;
        } catch (...) {
            pool.Release(k, 1);
            throw;
        }
    }
    pool.AcquireRange(S);
    std::vector<std::string> errs(static_cast<size_t>(S));
    std::vector<ie::PlanInstance*> pis(static_cast<size_t>(S), nullptr);
    auto run_slice = [&](int k) {
        try {
            ie::DeviceModel& D = *lanes[size_t(k)];
            const int64_t r0 = rows * k / S, r1 = rows * (k + 1) / S, nr = r1 - r0;
            std::vector<std::vector<int64_t>> sh = shapes;
            for (auto& x : sh) x[0] = nr;
            ie::PlanInstance& pi = D.Prepare(sh, false);
            pis[size_t(k)] = &pi;
            Segs mine;
            mine.first.resize(segs.first.size());
            mine.second.resize(segs.second.size());
            for (size_t i = 0; i < segs.first.size() && i < pi.plan.inputs.size(); ++i) {
                if (segs.first[i].empty()) continue;
                const size_t row_bytes = size_t(pi.plan.inputs[i].view.numel() / nr) * (segs.first[i][0].u8 ? 1 : sizeof(float));
                const size_t a = size_t(r0) * row_bytes, b = size_t(r1) * row_bytes;
                for (const auto& sg : segs.first[i]) {
                    const size_t lo = std::max(sg.dev_off, a), hi = std::min(sg.dev_off + sg.need, b);
                    if (hi <= lo) continue;
                    const size_t delta = lo - sg.dev_off;
                    ie::DeviceModel::InSeg n2{sg.host ? static_cast<const char*>(sg.host) + delta : nullptr,
                                              sg.have > delta ? std::min(sg.have - delta, hi - lo) : 0, hi - lo, lo - a, sg.u8};
                    mine.first[i].push_back(n2);
                }
            }
            for (size_t j = 0; j < segs.second.size() && j < pi.plan.outputs.size(); ++j) {
                const size_t row_bytes = size_t(pi.plan.outputs[j].view.numel() / nr) * sizeof(float);
                const size_t a = size_t(r0) * row_bytes, b = size_t(r1) * row_bytes;
                for (const auto& sg : segs.second[j]) {
                    const size_t nb = std::min(sg.cap, sg.need);
                    const size_t lo = std::max(sg.dev_off, a), hi = std::min(sg.dev_off + nb, b);
                    if (hi <= lo) continue;
                    mine.second[j].push_back({static_cast<char*>(sg.host) + (lo - sg.dev_off), hi - lo, hi - lo, lo - a});
                }
            }
            D.InferHostSegments(pi, mine.first, mine.second);
            Account(D, pi);
        } catch (const std::exception& e) {
            errs[size_t(k)] = e.what();
        } catch (...) {
            errs[size_t(k)] = "unknown error";
        }
    };
    for (int k = 1; k < S; ++k) workers.Submit(size_t(k - 1), [&run_slice, k] { run_slice(k); });
    run_slice(0);
    workers.Wait();
    std::vector<ie::IoDesc> outs0;
    if (pis[0]) outs0 = pis[0]->plan.outputs;
    pool.Release(0, S);
    for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
    // the slices wrote only the bytes they produced: zero-fill whatever a caller buffer has beyond its result
    for (const auto& outs : segs.second)
        for (const auto& sg : outs) {
            const size_t nb = std::min(sg.cap, sg.need);
            if (sg.cap > nb) std::memset(static_cast<char*>(sg.host) + nb, 0, sg.cap - nb);
        }
    *sharded = true;
    return outs0;
}

void ModelObj::Execute(std::vector<Pending*>& batch) {
    std::shared_lock<std::shared_mutex> g(life);
    auto fail_all = [&](const std::string& msg) { for (auto* r : batch) { r->ok = false; r->err = msg; } };
    if (!loaded.load() || lanes.empty()) { fail_all("Model not loaded"); return; }
    try {
        const bool coalesced = !(batch.size() == 1 && !(batchable && max_batch > 1 && batch[0]->rows > 0));
        const size_t nin = info.inputs.size(), nout = info.outputs.size();
        Segs segs;
        segs.first.resize(nin);
        segs.second.resize(nout);
        // elements per row of every graph input / output come from the request's own shapes (inputs) and the model (outputs are
        // sized by the plan: the segment's `need` is clipped by InferHostSegments against the planned tensor)
        auto row_elems = [](const std::vector<int64_t>& sh) { size_t n = 1; for (size_t k = 1; k < sh.size(); ++k) n *= size_t(sh[k]); return n; };
        if (!coalesced) {
            Pending& r = *batch[0];
            const int64_t rows = r.shapes.empty() || r.shapes[0].empty() ? 0 : r.shapes[0][0];
            bool same_rows = rows > 0 && batchable;
            for (auto& sh : r.shapes) if (sh.empty() || sh[0] != rows) same_rows = false;
            bool sharded = false;
            if (same_rows && num_shards > 1 && rows >= num_shards) {
                // per-row sizes need the output row size: take it from the primary's plan for one row per shard ... the plan for the
                // slice is only known inside the slice, so describe outputs by the model's declared dims instead
                for (size_t i = 0; i < nin; ++i) {
                    const size_t rb = row_elems(r.shapes[i]) * (r.in_u8[i] ? 1 : sizeof(float));
                    segs.first[i].push_back({r.in_ptr[i], r.in_ptr[i] ? r.in_bytes[i] : 0, size_t(rows) * rb, 0, r.in_u8[i] != 0});
                }
                for (int j = 0; j < r.num_outputs && size_t(j) < nout; ++j) {
                    const TensorData& o = r.outputs[j];
                    if (o.data_type != DATATYPE_FLOAT32 || !o.data || o.data_size == 0) continue;
                    size_t re = 1;
                    bool known = !info.outputs[size_t(j)].dims.empty();
                    for (size_t k = 1; k < info.outputs[size_t(j)].dims.size(); ++k) {
                        if (info.outputs[size_t(j)].dims[k] <= 0) known = false;
                        else re *= size_t(info.outputs[size_t(j)].dims[k]);
                    }
                    if (!known) { same_rows = false; break; }
                    segs.second[size_t(j)].push_back({o.data, o.data_size, size_t(rows) * re * sizeof(float), 0});
                }
            }
            if (same_rows && num_shards > 1 && rows >= num_shards) {
                const std::vector<ie::IoDesc> outs = RunOnLanes(r.shapes, rows, true, segs, &sharded);
                write_out_dims(r.outputs, r.num_outputs, outs, rows);
                if (sharded) shard_calls.fetch_add(1);
                r.ok = true;
                return;
            }
            // ---- one request on one lane ----
            const int k = pool.AcquireAny();
            if (k < 0) { fail_all("Model not loaded"); return; }
            try {
                ie::DeviceModel& D = *lanes[size_t(k)];
                ie::PlanInstance& pi = D.Prepare(r.shapes, false);
                std::vector<void*> out_ptr;
                std::vector<size_t> out_bytes;
                for (int i = 0; i < r.num_outputs; ++i) {
                    const bool copy = r.outputs[i].data_type == DATATYPE_FLOAT32 && r.outputs[i].data && r.outputs[i].data_size > 0;
                    out_ptr.push_back(copy ? r.outputs[i].data : nullptr);
                    out_bytes.push_back(copy ? r.outputs[i].data_size : 0);
                }
                D.InferHost(pi, r.in_ptr, r.in_bytes, out_ptr, out_bytes, r.in_u8);
                Account(D, pi);
                write_out_dims(r.outputs, r.num_outputs, pi.plan.outputs, 0);
                pool.Release(k, 1);
            } catch (...) {
                pool.Release(k, 1);
                throw;
            }
            r.ok = true;
            return;
        }
        // ---- coalesced batch: rows of all callers back to back, padded up to a power-of-two bucket so only a handful of
        //      plans / hipGraphs ever exist; with shard replicas the bucket is cut over them like a single large request ----
        int64_t total = 0;
        for (auto* r : batch) total += r->rows;
        int64_t bucket = 1;
        while (bucket < total) bucket <<= 1;
        if (bucket > max_batch && total <= max_batch) bucket = max_batch;
        std::vector<std::vector<int64_t>> shapes = batch[0]->shapes;
        for (auto& sh : shapes) sh[0] = bucket;
        std::vector<size_t> out_row_bytes(nout, 0);
        bool out_known = true;
        for (size_t j = 0; j < nout; ++j) {
            size_t re = 1;
            if (info.outputs[j].dims.empty()) out_known = false;
            for (size_t k = 1; k < info.outputs[j].dims.size(); ++k) {
                if (info.outputs[j].dims[k] <= 0) out_known = false;
                else re *= size_t(info.outputs[j].dims[k]);
            }
            out_row_bytes[j] = re * sizeof(float);
        }
        if (!out_known) {      // output row size only known from a plan: take it from a one-lane plan of the bucket
            const int k = pool.AcquireAny();
            if (k < 0) { fail_all("Model not loaded"); return; }
            try {
                ie::PlanInstance& pi = lanes[size_t(k)]->Prepare(shapes, false);
                for (size_t j = 0; j < nout && j < pi.plan.outputs.size(); ++j) out_row_bytes[j] = size_t(pi.plan.outputs[j].view.numel() / bucket) * sizeof(float);
                pool.Release(k, 1);
            } catch (...) { pool.Release(k, 1); throw; }
        }
        int64_t row0 = 0;
        for (auto* r : batch) {
            for (size_t k = 0; k < nin; ++k) {
                const size_t rb = row_elems(shapes[k]) * sizeof(float);
                segs.first[k].push_back({r->in_ptr[k], r->in_bytes[k], size_t(r->rows) * rb, size_t(row0) * rb});
            }
            for (int j = 0; j < r->num_outputs && size_t(j) < nout; ++j) {
                const TensorData& o = r->outputs[j];
                if (o.data_type != DATATYPE_FLOAT32 || !o.data || o.data_size == 0) continue;
                segs.second[size_t(j)].push_back({o.data, o.data_size, size_t(r->rows) * out_row_bytes[size_t(j)], size_t(row0) * out_row_bytes[size_t(j)]});
            }
            row0 += r->rows;
        }
        // rows of the bucket beyond `total` stay whatever the input buffer held: they are padding whose results nobody reads
        bool sharded = false;
        const std::vector<ie::IoDesc> outs = RunOnLanes(shapes, bucket, out_known, segs, &sharded);
        device_batches.fetch_add(1);
        coalesced_requests.fetch_add(int64_t(batch.size()));
        if (sharded) shard_calls.fetch_add(1);
        for (auto* r : batch) {
            write_out_dims(r->outputs, r->num_outputs, outs, r->rows);
            r->ok = true;
        }
    } catch (const std::exception& e) {
        fail_all(std::string("ONNX inference error: ") + e.what());
    }
}

void ModelObj::RunBatched(Pending& req) {
    std::unique_lock<std::mutex> lk(bmu);
    queue.push_back(&req);
    bcv.notify_all();                                  // a waiting leader re-checks whether its batch is full
    while (!req.done) {
        if (leader_active) { bcv.wait(lk); continue; }
        leader_active = true;                          // this caller drives the next device batch
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(batch_window_us);
        auto queued_rows = [&] { int64_t n = 0; for (auto* r : queue) n += r->rows; return n; };
        while (queued_rows() < max_batch && bcv.wait_until(lk, deadline) != std::cv_status::timeout) {}
        std::vector<Pending*> batch;
        int64_t rows = 0;
        for (auto it = queue.begin(); it != queue.end();) {
            Pending* r = *it;
            bool compatible = batch.empty();
            if (!compatible) {
                compatible = rows + r->rows <= max_batch;
                for (size_t k = 0; k < r->shapes.size() && compatible; ++k)
                    compatible = std::equal(r->shapes[k].begin() + 1, r->shapes[k].end(), batch[0]->shapes[k].begin() + 1,
                                            batch[0]->shapes[k].end());
            }
            if (compatible) { batch.push_back(r); rows += r->rows; it = queue.erase(it); }
            else ++it;
        }
        lk.unlock();
        Execute(batch);
        lk.lock();
        for (auto* r : batch) r->done = true;
        leader_active = false;
        bcv.notify_all();
    }
}

}  // namespace


namespace {
// Runtime facts of a loaded model.  json == false: one line of key=value pairs for ModelMetadata.description; json == true: the
// document EngineGetRuntimeInfo returns.  The caller holds M.life (shared).
std::string describe_runtime(ModelObj& M, bool json, bool checksums = false) {
    double ms, fl, by;
    int64_t fw, im;
    { std::lock_guard<std::mutex> g(M.acct_mu); ms = M.acct_ms; fl = M.acct_flops; by = M.acct_bytes; fw = M.acct_forwards; im = M.acct_images; }
    const bool up = M.loaded.load() && !M.lanes.empty();
    const ie::Precision prec = up ? M.lanes[0]->precision() : ie::Precision::F32;
    const char* pname = prec == ie::Precision::F16 ? "fp16" : (prec == ie::Precision::F8 ? "fp8" : "fp32");
    const double mfma_peak = prec == ie::Precision::F16 ? 2500.0 : (prec == ie::Precision::F8 ? 5000.0 : 157.3);     // TFLOP/s dense, gfx950
    const double tflops = ms > 0 ? fl / (ms * 1e-3) / 1e12 : 0, gbs = ms > 0 ? by / (ms * 1e-3) / 1e9 : 0;
    int max_in_flight;
    { std::lock_guard<std::mutex> g(M.pool.mu); max_in_flight = M.pool.max_in_flight; }
    std::ostringstream o;
    o.precision(6);
    if (!json) {
        o << "mi355x-engine precision=" << pname << (up && !M.lanes.empty() && M.lanes[0]->fp32_split() ? "+bf16x6" : "") << " lanes=" << M.lanes.size() << " shards=" << (up ? M.num_shards : 0) << " forwards=" << fw
          << " images=" << im << " device_ms_avg=" << (fw ? ms / double(fw) : 0.0) << " achieved_tflops=" << tflops << " frac_mfma_peak=" << tflops / mfma_peak
          << " algorithmic_gbs=" << gbs << " frac_hbm_peak=" << gbs / 8000.0;
        return o.str();
    }
    o << "{\"loaded\":" << (up ? "true" : "false") << ",\"precision\":\"" << pname << "\",\"fp32_split\":" << (up && !M.lanes.empty() && M.lanes[0]->fp32_split() ? "true" : "false")
      << ",\"lanes\":" << M.lanes.size() << ",\"shards\":" << (up ? M.num_shards : 0)
      << ",\"lane_devices\":[";
    for (size_t i = 0; i < M.lanes.size(); ++i) o << (i ? "," : "") << M.lanes[i]->device();
    o << "],\"lane_shares_weights_with\":[";
    for (size_t i = 0; i < M.lanes.size(); ++i) {
        size_t first = i;
        for (size_t k = 0; k < i; ++k) if (M.lanes[k]->shared_weights() == M.lanes[i]->shared_weights()) { first = k; break; }
        o << (i ? "," : "") << first;
    }
    o << "],\"max_in_flight\":" << max_in_flight << ",\"rccl\":{\"used\":" << (M.rccl.used ? "true" : "false") << ",\"ranks\":" << M.rccl.ranks
      << ",\"weight_owners\":" << M.rccl.owners << ",\"bytes\":" << M.rccl.bytes << ",\"init_ms\":" << M.rccl.init_ms << ",\"broadcast_ms\":" << M.rccl.bcast_ms
      << "},\"forwards\":" << fw << ",\"images\":" << im << ",\"device_ms_total\":" << ms << ",\"achieved_tflops\":" << tflops
      << ",\"mfma_peak_tflops\":" << mfma_peak << ",\"algorithmic_gbs\":" << gbs << ",\"hbm_peak_gbs\":8000";
    if (up) {
        o << ",\"pipelined_calls\":[";
        for (size_t i = 0; i < M.lanes.size(); ++i) o << (i ? "," : "") << M.lanes[i]->pipelined_calls();
        o << "],\"last_chunks\":" << M.lanes[0]->last_chunks() << ",\"last_head_steps\":" << M.lanes[0]->last_head_steps();
    }
    if (up && prec == ie::Precision::F8) {
        // fp8 mode: the calibrated per-tensor scales (index = plan step that writes the tensor; real value = e4m3 code x scale)
        const auto& W = *M.lanes[0]->shared_weights();
        o << ",\"f8_ready\":" << (W.f8_ready ? "true" : "false") << ",\"f8_act_scales\":[";
        o.precision(9);
        for (size_t i = 0; i < W.act_scale.size(); ++i) o << (i ? "," : "") << W.act_scale[i];
        o << "]";
        o.precision(6);
    }
    if (up && checksums) {
        // FNV-1a of every lane's packed fp32 blob as it sits in HBM (a replica filled by the RCCL broadcast must equal the primary)
        o << ",\"weight_checksums\":[";
        for (size_t i = 0; i < M.lanes.size(); ++i) {
            std::vector<char> host(M.lanes[i]->weight_bytes());
            try { M.lanes[i]->CopySync(host.data(), M.lanes[i]->weights(), host.size(), hipMemcpyDeviceToHost, "hipMemcpy(weight checksum)"); }
            catch (const std::exception&) { (void)hipGetLastError(); host.clear(); }
            o << (i ? "," : "") << "\"" << std::hex << fnv1a64(host.data(), host.size()) << std::dec << "\"";
        }
        // and of everything derived from it per lane (half / fragment-major / Winograd U / bf16x6 / e4m3 + scales)
        o << "],\"mirror_checksums\":[";
        for (size_t i = 0; i < M.lanes.size(); ++i) {
            o << (i ? "," : "") << "{";
            bool first = true;
            for (const auto& kv : M.lanes[i]->MirrorChecksums()) {
                o << (first ? "" : ",") << "\"" << kv.first << "\":\"" << std::hex << kv.second << std::dec << "\"";
                first = false;
            }
            o << "}";
        }
        o << "]";
    }
    o << "}";
    return o.str();
}

// shared hold on the model's lifetime + exclusive use of the primary lane (the ext API works on lane 0)
struct Lane0 {
    ModelObj& M;
    std::shared_lock<std::shared_mutex> lk;
    bool ok = false;
    explicit Lane0(ModelObj& m) : M(m), lk(m.life) {
        if (M.loaded.load() && !M.lanes.empty()) { M.pool.AcquireOne(0); ok = true; }
    }
    ~Lane0() { if (ok) M.pool.Release(0, 1); }
    ie::DeviceModel* dev() const { return ok ? M.lanes[0].get() : nullptr; }
};
}  // namespace

extern "C" {

ModelMetadata* ModelGetMetadata(ModelHandle handle) {
    if (!handle) return nullptr;
    try {
        ModelObj& M = *handle->model;
        std::shared_lock<std::shared_mutex> g(M.life);
        auto* md = static_cast<ModelMetadata*>(std::calloc(1, sizeof(ModelMetadata)));
        md->name = dup_cstr(M.name);
        md->version = dup_cstr(M.version);
        md->model_type = M.type;
        // The reference leaves `description` empty (inference_bridge.cpp:836-926).  The engine uses the free-form string to surface
        // what ModelStats' four fixed counters cannot: device time per forward from HIP events, the planner's
        // algorithmic FLOP/s and bytes/s of what ran against the gfx950 peaks, lanes / shards.  Go reads it through GetMetadata().
        md->description = dup_cstr(describe_runtime(M, false));
        md->load_time_ns = M.load_time_ns;
        auto fill = [](const std::vector<std::string>& v, const char*** arr, int* n) {
            *n = int(v.size());
            *arr = nullptr;
            if (v.empty()) return;
            *arr = static_cast<const char**>(std::malloc(sizeof(char*) * v.size()));
            for (size_t i = 0; i < v.size(); ++i) (*arr)[i] = dup_cstr(v[i]);
        };
        fill(M.input_names, &md->inputs, &md->num_inputs);
        fill(M.output_names, &md->outputs, &md->num_outputs);
        return md;
    } catch (...) { return nullptr; }
}

void ModelFreeMetadata(ModelMetadata* md) {
    if (!md) return;
    std::free(const_cast<char*>(md->name));
    std::free(const_cast<char*>(md->version));
    std::free(const_cast<char*>(md->description));
    if (md->inputs) { for (int i = 0; i < md->num_inputs; ++i) std::free(const_cast<char*>(md->inputs[i])); std::free(md->inputs); }
    if (md->outputs) { for (int i = 0; i < md->num_outputs; ++i) std::free(const_cast<char*>(md->outputs[i])); std::free(md->outputs); }
    std::free(md);
}

ModelStats* ModelGetStats(ModelHandle handle) {
    if (!handle) return nullptr;
    try {
        ModelObj& M = *handle->model;
        auto* s = static_cast<ModelStats*>(std::calloc(1, sizeof(ModelStats)));
        s->inference_count = M.inference_count.load();
        s->total_inference_time_ns = M.total_ns.load();
        s->last_inference_time_ns = M.last_ns.load();
        s->memory_usage_bytes = M.memory_usage_bytes.load();
        return s;
    } catch (...) { return nullptr; }
}

void ModelFreeStats(ModelStats* stats) { std::free(stats); }

void FreeErrorMessage(ErrorMessage error) { std::free(error); }

ModelHandle GetModelHandle(InferenceManagerHandle handle, const char* model_name, const char* /*version*/, ErrorMessage* error) {
    if (!handle || !model_name) { set_error(error, "Invalid handle or model name"); return nullptr; }
    try {
        std::lock_guard<std::mutex> g(handle->mu);
        auto it = handle->models.find(model_name);
        if (it == handle->models.end()) { set_error(error, "Model not found in loaded models"); return nullptr; }
        return new Model_t{it->second};   // wrapper shares ownership; ModelDestroy frees only the wrapper
    } catch (const std::exception& e) { set_error(error, e.what()); return nullptr; }
    catch (...) { set_error(error, "unknown error"); return nullptr; }
}

// ---- extensions (include/inference_engine_ext.h) --------------------------------------------------------------
static void json_vi(std::ostringstream& o, const std::vector<ie::OnnxValueInfo>& v) {
    o << "[";
    for (size_t i = 0; i < v.size(); ++i) {
        o << (i ? "," : "") << "{\"name\":\"" << v[i].name << "\",\"elem_type\":" << v[i].elem_type << ",\"dims\":[";
        for (size_t k = 0; k < v[i].dims.size(); ++k) o << (k ? "," : "") << v[i].dims[k];
        o << "]}";
    }
    o << "]";
}

char* EngineDescribeModel(const char* path, int batch, ErrorMessage* error) {
    if (!path) { set_error(error, "Invalid parameters"); return nullptr; }
    try {
        std::string file = path;
        std::error_code ec;
        if (std::filesystem::is_directory(file, ec)) file += "/model.onnx";
        if (!std::filesystem::exists(file, ec)) { set_error(error, "ONNX model file not found: " + file); return nullptr; }
        ie::OnnxModel m = ie::LoadOnnxFile(file);
        ie::ModelInfo info = ie::DescribeModel(m);
        std::ostringstream o;
        o << "{\"ir_version\":" << m.ir_version << ",\"opset\":" << m.opset << ",\"producer\":\"" << m.producer
          << "\",\"num_nodes\":" << m.nodes.size() << ",\"num_initializers\":" << m.initializers.size()
          << ",\"memory_usage_bytes\":" << info.memory_usage_bytes << ",\"inputs\":";
        json_vi(o, info.inputs);
        o << ",\"outputs\":";
        json_vi(o, info.outputs);
        if (std::filesystem::is_directory(path, ec)) {
            // what the engine reads from <dir>/config.json at load (top-level keys only); a malformed file is an error here as at load
            const ie::EngineConfig c = ie::LoadEngineConfig(path);
            auto esc = [](const std::string& t) { std::string r; for (char ch : t) { if (ch == '"' || ch == '\\') r += '\\'; if (static_cast<unsigned char>(ch) >= 0x20) r += ch; } return r; };
            auto io = [&](const std::vector<ie::IoConfig>& v) {
                o << "[";
                for (size_t i = 0; i < v.size(); ++i) {
                    o << (i ? "," : "") << "{\"name\":\"" << esc(v[i].name) << "\",\"data_type\":\"" << esc(v[i].data_type) << "\",\"label_filename\":\""
                      << esc(v[i].label_filename) << "\",\"dims\":[";
                    for (size_t k = 0; k < v[i].dims.size(); ++k) o << (k ? "," : "") << v[i].dims[k];
                    o << "],\"shape\":[";
                    for (size_t k = 0; k < v[i].shape.size(); ++k) o << (k ? "," : "") << v[i].shape[k];
                    o << "]}";
                }
                o << "]";
            };
            o << ",\"config\":{\"present\":" << (c.present ? "true" : "false") << ",\"name\":\"" << esc(c.name) << "\",\"version\":\"" << esc(c.version)
              << "\",\"platform\":\"" << esc(c.platform) << "\",\"precision\":\"" << esc(c.precision) << "\",\"gpus\":" << c.gpus
              << ",\"uint8_scale\":" << c.uint8_scale << ",\"uint8_bias\":" << c.uint8_bias << ",\"dynamic_batching\":" << (c.dynamic_batching ? "true" : "false")
              << ",\"max_batch_size\":" << c.max_batch_size << ",\"batch_window_us\":" << c.batch_window_us << ",\"instance_count\":" << c.instance_count
              << ",\"fp32_split\":" << (c.fp32_split ? "true" : "false") << ",\"tune_batches\":[";
            for (size_t k = 0; k < c.tune_batches.size(); ++k) o << (k ? "," : "") << c.tune_batches[k];
            o << "],\"inputs\":";
            io(c.inputs);
            o << ",\"outputs\":";
            io(c.outputs);
            o << "}";
        }
        if (batch > 0) {
            std::vector<std::vector<int64_t>> shapes;
            for (auto& vi : info.inputs) {
                std::vector<int64_t> s = vi.dims;
                for (size_t k = 0; k < s.size(); ++k) if (s[k] <= 0) s[k] = (k == 0 ? batch : 1);
                shapes.push_back(s);
            }
            ie::Precision prec = ie::Precision::F32;       // the planner is host code: IE_PRECISION selects what to describe
            const ie::Env env = ie::Env::Read();
            if (const char* e = env.get("IE_PRECISION")) {
                std::string w = e;
                if (w == "fp16" || w == "f16" || w == "half") prec = ie::Precision::F16;
                else if (w == "fp8" || w == "f8" || w == "e4m3") prec = ie::Precision::F8;
            }
            o << ",\"plan\":" << ie::PlanToJson(ie::BuildPlan(m, shapes, prec));
        }
        o << "}";
        return dup_cstr(o.str());
    } catch (const std::exception& e) { set_error(error, e.what()); return nullptr; }
    catch (...) { set_error(error, "unknown error"); return nullptr; }
}

float* EnginePlanWeights(const char* path, int batch, size_t* count, ErrorMessage* error) {
    if (!path || !count || batch <= 0) { set_error(error, "Invalid parameters"); return nullptr; }
    try {
        std::string file = path;
        std::error_code ec;
        if (std::filesystem::is_directory(file, ec)) file += "/model.onnx";
        if (!std::filesystem::exists(file, ec)) { set_error(error, "ONNX model file not found: " + file); return nullptr; }
        ie::OnnxModel m = ie::LoadOnnxFile(file);
        std::vector<std::vector<int64_t>> shapes;
        for (auto& vi : m.inputs) {
            std::vector<int64_t> s = vi.dims;
            for (size_t k = 0; k < s.size(); ++k) if (s[k] <= 0) s[k] = (k == 0 ? batch : 1);
            shapes.push_back(s);
        }
        ie::Plan p = ie::BuildPlan(m, shapes, ie::Precision::F32);
        float* out = static_cast<float*>(std::malloc(std::max<size_t>(p.weights.size(), 1) * sizeof(float)));
        if (!out) { set_error(error, "out of memory"); return nullptr; }
        std::memcpy(out, p.weights.data(), p.weights.size() * sizeof(float));
        *count = p.weights.size();
        return out;
    } catch (const std::exception& e) { set_error(error, e.what()); return nullptr; }
    catch (...) { set_error(error, "unknown error"); return nullptr; }
}

bool EnginePrepare(ModelHandle handle, const Shape* input_shapes, int num_inputs, void** d_inputs, void** d_outputs,
                   int num_outputs, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return false; }
    try {
        ModelObj& M = *handle->model;
        Lane0 L0(M);
        if (!L0.dev()) { set_error(error, "Model not loaded"); return false; }
        if (!input_shapes || num_inputs <= 0) { set_error(error, "Invalid parameters"); return false; }
        std::vector<std::vector<int64_t>> shapes;
        for (int i = 0; i < num_inputs; ++i) {
            if (!input_shapes[i].dims || input_shapes[i].num_dims <= 0) { set_error(error, "Invalid parameters"); return false; }
            shapes.emplace_back(input_shapes[i].dims, input_shapes[i].dims + input_shapes[i].num_dims);
        }
        ie::PlanInstance& pi = L0.dev()->Prepare(shapes, true);
        for (int i = 0; d_inputs && i < num_inputs && size_t(i) < pi.plan.inputs.size(); ++i)
            d_inputs[i] = pi.buffers[size_t(pi.plan.inputs[size_t(i)].view.buf)];
        for (int i = 0; d_outputs && i < num_outputs && size_t(i) < pi.plan.outputs.size(); ++i)
            d_outputs[i] = pi.buffers[size_t(pi.plan.outputs[size_t(i)].view.buf)];
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

bool EngineRunPrepared(ModelHandle handle, int iters, int sync, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return false; }
    try {
        ModelObj& M = *handle->model;
        Lane0 L0(M);
        if (!L0.dev() || !L0.dev()->current()) { set_error(error, "Model not prepared"); return false; }
        // IE_MAX_INFLIGHT_REPLAYS=n: synchronise every n replays (a profiler's per-dispatch bookkeeping under deep un-synchronised graph
        // queues crashed rocprofv3 --kernel-trace: profiles/r02/graph_burst_under_kernel_trace_sigsegv.log); default: no cap
        const int cap = L0.dev()->max_inflight_replays();
        for (int i = 0; i < iters; ++i) {
            L0.dev()->Enqueue(*L0.dev()->current());
            if (cap > 0 && (i + 1) % cap == 0) L0.dev()->Synchronize();
        }
        if (sync) L0.dev()->Synchronize();
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

bool EngineSynchronize(ModelHandle handle, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return false; }
    try {
        ModelObj& M = *handle->model;
        Lane0 L0(M);
        if (!L0.dev()) { set_error(error, "Model not loaded"); return false; }
        L0.dev()->Synchronize();
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

void* EngineGetStream(ModelHandle handle) {
    if (!handle) return nullptr;
    try {
        ModelObj& M = *handle->model;
        std::shared_lock<std::shared_mutex> g(M.life);
        return M.lanes.empty() ? nullptr : static_cast<void*>(M.lanes[0]->stream());
    } catch (...) { return nullptr; }
}

char* EngineProfile(ModelHandle handle, int iters, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return nullptr; }
    try {
        ModelObj& M = *handle->model;
        Lane0 L0(M);
        if (!L0.dev() || !L0.dev()->current()) { set_error(error, "Model not prepared"); return nullptr; }
        auto t = L0.dev()->Profile(*L0.dev()->current(), iters > 0 ? iters : 1);
        std::ostringstream o;
        o.precision(9);
        o << "[";
        for (size_t i = 0; i < t.size(); ++i) {
            std::string nm;
            for (char c : t[i].name) if (c != '"' && c != '\\') nm += c;
            o << (i ? "," : "") << "{\"name\":\"" << nm << "\",\"kernel\":\"" << t[i].kernel << "\",\"ms\":" << t[i].ms
              << ",\"flops\":" << t[i].flops << ",\"bytes\":" << t[i].bytes << "}";
        }
        o << "]";
        return dup_cstr(o.str());
    } catch (const std::exception& e) { set_error(error, e.what()); return nullptr; }
    catch (...) { set_error(error, "unknown error"); return nullptr; }
}

bool EngineGetWeightBlob(ModelHandle handle, void** d_ptr, size_t* bytes, ErrorMessage* error) {
    if (!handle || !d_ptr || !bytes) { set_error(error, "Invalid parameters"); return false; }
    try {
        ModelObj& M = *handle->model;
        Lane0 L0(M);
        if (!L0.dev()) { set_error(error, "Model not loaded"); return false; }
        *d_ptr = L0.dev()->weights();
        *bytes = L0.dev()->weight_bytes();
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

bool EngineWeightsUpdated(ModelHandle handle, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid parameters"); return false; }
    try {
        // The primary's fp32 blob was rewritten in place.  Every lane may be running a forward on the derived mirrors (extra
        // instance_count lanes and same-device shards share them), so ALL lanes are held for the update; replicas that own a blob of
        // their own (other devices, IE_SHARD_PRIVATE_WEIGHTS) get the new weights through the same broadcast as at load.
        ModelObj& M = *handle->model;
        std::shared_lock<std::shared_mutex> g(M.life);
        if (!M.loaded.load() || M.lanes.empty()) { set_error(error, "Model not loaded"); return false; }
        const int n = int(M.lanes.size());
        M.pool.AcquireRange(n);
        try {
            for (auto& l : M.lanes) l->Synchronize();
            M.lanes[0]->WeightsArrived();
            M.BroadcastWeights();
        } catch (...) { M.pool.Release(0, n); throw; }
        M.pool.Release(0, n);
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

int EngineGetPrecision(ModelHandle handle) {
    if (!handle) return -1;
    ModelObj& M = *handle->model;
    std::shared_lock<std::shared_mutex> g(M.life);
    if (!M.loaded.load() || M.lanes.empty()) return -1;
    return int(M.lanes[0]->precision());
}

bool EngineMemcpy(ModelHandle handle, void* dst, const void* src, size_t bytes, int kind, ErrorMessage* error) {
    if (!handle || !dst || !src || kind < 1 || kind > 3) { set_error(error, "Invalid parameters"); return false; }
    try {
        ModelObj& M = *handle->model;
        Lane0 L0(M);
        if (!L0.dev()) { set_error(error, "Model not loaded"); return false; }
        L0.dev()->Synchronize();
        hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : (kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
        hipError_t e = hipMemcpy(dst, src, bytes, k);
        if (e != hipSuccess) { set_error(error, std::string("HIP error in hipMemcpy: ") + hipGetErrorString(e)); return false; }
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

double EngineMfmaPeak(int nacc, int blocks_per_cu, int iters) {
    if (ie::HipDeviceCount() <= 0) return -1.0;
    try { return ie::MfmaPeakTflops(nacc, blocks_per_cu, iters); } catch (...) { return -1.0; }
}

bool EngineGetBatcherStats(ModelHandle handle, int64_t* device_batches, int64_t* coalesced_requests, int* max_batch) {
    if (!handle) return false;
    ModelObj& M = *handle->model;
    if (device_batches) *device_batches = M.device_batches.load();
    if (coalesced_requests) *coalesced_requests = M.coalesced_requests.load();
    if (max_batch) *max_batch = (M.batchable && M.max_batch > 1) ? M.max_batch : 0;
    return true;
}

bool EngineGetShardStats(ModelHandle handle, int* num_shards, int64_t* sharded_calls) {
    if (!handle) return false;
    ModelObj& M = *handle->model;
    std::shared_lock<std::shared_mutex> g(M.life);
    if (num_shards) *num_shards = M.lanes.empty() ? 0 : M.num_shards;
    if (sharded_calls) *sharded_calls = M.shard_calls.load();
    return true;
}

char* EngineGetRuntimeInfo(ModelHandle handle, int with_checksums, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return nullptr; }
    try {
        ModelObj& M = *handle->model;
        std::shared_lock<std::shared_mutex> g(M.life);
        if (with_checksums && M.loaded.load() && !M.lanes.empty()) {
            // reading the blobs must not race a forward: hold every lane
            M.pool.AcquireRange(int(M.lanes.size()));
            std::string r;
            try { r = describe_runtime(M, true, true); } catch (...) { M.pool.Release(0, int(M.lanes.size())); throw; }
            M.pool.Release(0, int(M.lanes.size()));
            return dup_cstr(r);
        }
        return dup_cstr(describe_runtime(M, true, false));
    } catch (const std::exception& e) { set_error(error, e.what()); return nullptr; }
    catch (...) { set_error(error, "unknown error"); return nullptr; }
}

bool EngineE4m3RoundTrip(const float* src, float* dst, unsigned char* codes, size_t n, float scale, ErrorMessage* error) {
    if (!src || !dst || !(scale > 0.f)) { set_error(error, "Invalid parameters"); return false; }
    if (ie::HipDeviceCount() <= 0) { set_error(error, "No HIP device available"); return false; }
    if (n == 0) return true;
    float *ds = nullptr, *dd = nullptr;
    unsigned char* dc = nullptr;
    bool ok = hipMalloc(reinterpret_cast<void**>(&ds), n * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&dd), n * 4) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&dc), n) == hipSuccess && hipMemcpy(ds, src, n * 4, hipMemcpyHostToDevice) == hipSuccess &&
              ie::LaunchE4m3RoundTrip(ds, dd, dc, scale, int64_t(n), nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
              hipMemcpy(dst, dd, n * 4, hipMemcpyDeviceToHost) == hipSuccess && (!codes || hipMemcpy(codes, dc, n, hipMemcpyDeviceToHost) == hipSuccess);
    if (ds) (void)hipFree(ds);
    if (dd) (void)hipFree(dd);
    if (dc) (void)hipFree(dc);
    if (!ok) { (void)hipGetLastError(); set_error(error, "HIP error in EngineE4m3RoundTrip"); }
    return ok;
}

bool EngineVectorAdd(const float* a, const float* b, float* result, size_t n, ErrorMessage* error) {
    if (!a || !b || !result) { set_error(error, "Invalid parameters"); return false; }
    if (ie::HipDeviceCount() <= 0) { set_error(error, "No HIP device available"); return false; }
    float *da = nullptr, *db = nullptr, *dr = nullptr;
    bool ok = false;
    std::string msg;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess) { msg = std::string("HIP error in ") + what + ": " + hipGetErrorString(e); return false; }
        return true;
    };
    const size_t bytes = n * sizeof(float);
    if (n == 0) return true;
    if (chk(hipMalloc(reinterpret_cast<void**>(&da), bytes), "hipMalloc") && chk(hipMalloc(reinterpret_cast<void**>(&db), bytes), "hipMalloc") &&
        chk(hipMalloc(reinterpret_cast<void**>(&dr), bytes), "hipMalloc") && chk(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice), "hipMemcpy") &&
        chk(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice), "hipMemcpy") && chk(ie::LaunchVectorAdd(da, db, dr, int64_t(n), nullptr), "vector_add") &&
        chk(hipDeviceSynchronize(), "hipDeviceSynchronize") && chk(hipMemcpy(result, dr, bytes, hipMemcpyDeviceToHost), "hipMemcpy"))
        ok = true;
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dr) (void)hipFree(dr);
    if (!ok) set_error(error, msg);
    return ok;
}

}  // extern "C"
