// Internals shared by the translation units behind the C ABI (bridge.cpp: the extern "C" shims; bridge_load.cpp: load / unload, shard replicas and the
// RCCL weight broadcast; bridge_run.cpp: lane dispatch, batch sharding and the request batcher).  Nothing here is exported.
#pragma once
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

namespace ie_bridge {


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
inline const Roctx& roctx() { static Roctx r; return r; }
struct RoctxRange {
    bool on;
    explicit RoctxRange(const std::string& name) : on(roctx().push != nullptr) { if (on) roctx().push(name.c_str()); }
    ~RoctxRange() { if (on) roctx().pop(); }
};


inline char* dup_cstr(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}
inline void set_error(ErrorMessage* error, const std::string& msg) {
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

inline uint64_t fnv1a64(const void* data, size_t n) {
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

// outputs by index, in graph-output order (bridge:787-813); never writes past the caller's dims array  (bridge_run.cpp)
void write_out_dims(TensorData* outputs, int num_outputs, const std::vector<ie::IoDesc>& odesc, int64_t rows);

}  // namespace ie_bridge

#define NCCL_OK(call)                                                                                             \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess) throw std::runtime_error(std::string("RCCL error in " #call ": ") + ncclGetErrorString(r_)); \
    } while (0)

struct Model_t {
    std::shared_ptr<ie_bridge::ModelObj> model;
};

struct InferenceManager_t {
    std::string repo_path;
    std::unique_ptr<ie::Repository> repo;
    std::mutex mu;
    std::unordered_map<std::string, std::shared_ptr<ie_bridge::ModelObj>> models;   // keyed by name only (bridge:320)
};
