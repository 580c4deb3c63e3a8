#include "executor.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "igemm_tiles.h"
#include "kernels.h"

namespace ie {
namespace {

void check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

TensorArg make_arg(const PlanInstance& pi, const View& v) {
    TensorArg t;
    float* base = pi.buffers.at(size_t(v.buf));
    t.n = int(v.n); t.h = int(v.h); t.w = int(v.w); t.c = int(v.c);
    t.f16 = v.f16 ? 1 : 0;
    if (v.nchw) {
        t.p = base;
        t.sw = 1; t.sh = v.w; t.sc = v.h * v.w; t.sn = v.c * v.h * v.w;
    } else {
        t.p = v.f16 ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + v.c_off * 2) : base + v.c_off;
        t.sc = 1; t.sw = v.pitch; t.sh = v.w * v.pitch; t.sn = v.h * v.w * v.pitch;
    }
    return t;
}

constexpr size_t kChunk = size_t(4) << 20;   // pinned staging chunk
constexpr int64_t kTuneWorkspaceFloats = int64_t(16) << 20;   // 64 MiB of split-K slabs available to the autotuner
constexpr int kNumCounters = 1 << 16;
constexpr int kSlots = 4;

std::once_flag g_kernels_once;
hipError_t g_kernels_err = hipSuccess;

}  // namespace

CopyPool::CopyPool(int helpers) {
    for (int i = 0; i < helpers; ++i) threads_.emplace_back([this, i] { Worker(i); });
}
CopyPool::~CopyPool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
}
void CopyPool::Worker(int idx) {
    uint64_t seen = 0;
    for (;;) {
        char* d; const char* s; size_t n; size_t parts;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            d = dst_; s = src_; n = n_; parts = threads_.size() + 1;
        }
        const size_t per = (n / parts + 63) & ~size_t(63);
        const size_t b = std::min(n, per * size_t(idx + 1)), e = std::min(n, per * size_t(idx + 2));
        if (e > b) std::memcpy(d + b, s + b, e - b);
        {
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
}
void CopyPool::Copy(void* dst, const void* src, size_t n) {
    if (threads_.empty() || n < (size_t(1) << 20)) { std::memcpy(dst, src, n); return; }
    const size_t parts = threads_.size() + 1;
    {
        std::lock_guard<std::mutex> g(mu_);
        dst_ = static_cast<char*>(dst); src_ = static_cast<const char*>(src); n_ = n;
        pending_ = int(threads_.size());
        ++generation_;
    }
    cv_.notify_all();
    const size_t per = (n / parts + 63) & ~size_t(63);
    std::memcpy(dst, src, std::min(n, per));      // the caller copies slice 0
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
}

int HipDeviceCount() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

std::string HipDeviceInfo(int device_id) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device_id) != hipSuccess) { (void)hipGetLastError(); return "Unknown device"; }
    // Same shape as the reference's string (cuda_utils.cu:51-54); on AMD the "compute capability" pair is the
    // gfx major.minor HIP reports (9.5 for gfx950).
    return "Device " + std::to_string(device_id) + ": " + p.name + " (Compute Capability " + std::to_string(p.major) + "." +
           std::to_string(p.minor) + ")";
}

bool HipMemoryInfo(int device_id, size_t* total, size_t* free_b) {
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipSetDevice(device_id) != hipSuccess) { (void)hipGetLastError(); return false; }
    bool ok = hipMemGetInfo(free_b, total) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    (void)hipSetDevice(prev);
    return ok;
}

DeviceModel::DeviceModel(std::shared_ptr<const OnnxModel> model, int device_id, Precision precision)
    : model_(std::move(model)), device_(device_id), precision_(precision) {
    int n = HipDeviceCount();
    if (n <= 0) throw std::runtime_error("No HIP device available: the MI355X engine has no CPU fallback");
    if (device_id < 0 || device_id >= n) throw std::runtime_error("Invalid device id " + std::to_string(device_id));
    check(hipSetDevice(device_), "hipSetDevice");
    std::call_once(g_kernels_once, [] {
        g_kernels_err = InitKernels();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsF16();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsWs();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsWs32();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsWs3();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsStem();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsDirect();
    });
    check(g_kernels_err, "InitKernels");
    check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    const char* ng = std::getenv("IE_DISABLE_GRAPH");
    use_graph_ = !(ng && ng[0] == '1');
    const char* at = std::getenv("IE_AUTOTUNE");
    // Measured on MI355X (DenseNet-121 B=32): the in-launch combine (agent-scope release/acquire per tile) costs more than
    // the kernel boundary it removes: 3.63 ms/step vs 3.42 ms/step with the separate reduce kernel.  Two-pass is the default.
    const char* tp = std::getenv("IE_SPLITK_IN_LAUNCH");
    two_pass_splitk_ = !(tp && tp[0] == '1');
    autotune_ = !(at && at[0] == '0') && !std::getenv("IE_FORCE_TILE") && !std::getenv("IE_FORCE_SPLITK") && !std::getenv("IE_FORCE_ALGO");
    // Optional persistent tuning cache (IE_TUNE_CACHE=<file>): "<17 signature ints> : <tile> <splitk>" per line.
    if (const char* tc = std::getenv("IE_TUNE_CACHE")) {
        std::ifstream f(tc);
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream is(line);
            std::vector<int64_t> key;
            std::string tok;
            while (is >> tok && tok != ":") key.push_back(std::stoll(tok));
            int t = -1, sp = 0;
            if ((is >> t >> sp) && ((t >= 0 && t < kNumIgemmTiles) || (t >= 100 && t < 100 + kNumConvRasterTiles) || (t >= 200 && t < 200 + kNumConvWs32Tiles) ||
                                     (t >= 300 && t < 300 + kNumConvWs3Tiles) || (t >= 400 && t < 400 + kNumConvDirectTiles)) &&
                sp >= 1 && sp <= 64)
                tune_cache_[key] = {t, sp};
        }
    }
    if (const char* ns = std::getenv("IE_STREAMS")) sub_streams_ = std::max(1, std::min(8, std::atoi(ns)));
    if (const char* mp = std::getenv("IE_MAX_PLANS")) max_plans_ = size_t(std::max(1, std::atoi(mp)));
    for (int i = 1; i < sub_streams_; ++i) {
        hipStream_t st = nullptr;
        check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
        side_streams_.push_back(st);
    }
    {
        int helpers = 3;
        if (const char* h = std::getenv("IE_COPY_THREADS")) helpers = std::max(0, std::min(15, std::atoi(h) - 1));
        copy_pool_ = std::make_unique<CopyPool>(helpers);
    }
    pinned_bytes_ = kChunk * kSlots;
    check(hipHostMalloc(&pinned_, pinned_bytes_, hipHostMallocDefault), "hipHostMalloc");
}

DeviceModel::~DeviceModel() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& kv : plans_) FreeInstance(*kv.second);
    for (auto st : side_streams_) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (d_weights_) (void)hipFree(d_weights_);
    if (d_weights16_) (void)hipFree(d_weights16_);
    if (d_weights_frag_) (void)hipFree(d_weights_frag_);
    if (pinned_) (void)hipHostFree(pinned_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void DeviceModel::FreeInstance(PlanInstance& pi) {
    for (auto& sub : pi.subs) FreeInstance(*sub);
    if (pi.graph_exec) (void)hipGraphExecDestroy(pi.graph_exec);
    for (size_t i = 0; i < pi.buffers.size(); ++i)
        if (pi.buffers[i] && pi.owned[i]) (void)hipFree(pi.buffers[i]);
    if (pi.workspace) (void)hipFree(pi.workspace);
    if (pi.counters) (void)hipFree(pi.counters);
    for (void* p : pi.u8_stage) if (p) (void)hipFree(p);
    if (pi.done) (void)hipEventDestroy(pi.done);
    if (pi.fork) (void)hipEventDestroy(pi.fork);
}

// Plan + allocate one instance.  io_only: allocate just the graph input/output buffers (parent of sub-batch instances).
void DeviceModel::BuildInstance(PlanInstance& pi, const std::vector<std::vector<int64_t>>& shapes, bool io_only) {
    pi.plan = BuildPlan(*model_, shapes, precision_);
    pi.stream = stream_;
    if (!d_weights_) {
        weight_floats_ = pi.plan.weights.size();
        check(hipMalloc(reinterpret_cast<void**>(&d_weights_), std::max<size_t>(weight_floats_, 4) * sizeof(float)), "hipMalloc(weights)");
        device_bytes_ += weight_floats_ * sizeof(float);
        check(hipMemcpy(d_weights_, pi.plan.weights.data(), weight_floats_ * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(weights)");
        if (precision_ == Precision::F16) {
            check(hipMalloc(&d_weights16_, std::max<size_t>(weight_floats_, 8) * 2), "hipMalloc(weights16)");
            device_bytes_ += weight_floats_ * 2;
        } else if (const char* nf = std::getenv("IE_NO_FRAG_WEIGHTS"); !(nf && std::atoi(nf) != 0)) {
            for (const Step& st : pi.plan.steps)
                if (st.kind == StepKind::Conv && st.w_off >= 0 && st.out.c % 16 == 0 && st.in.c % 16 == 0 && st.kh * st.kw <= 49)
                    frag_regions_.push_back({st.w_off, int(st.out.c), st.kh * st.kw, int(st.in.c)});
            if (!frag_regions_.empty()) {
                check(hipMalloc(reinterpret_cast<void**>(&d_weights_frag_), weight_floats_ * sizeof(float)), "hipMalloc(weights_frag)");
                device_bytes_ += weight_floats_ * sizeof(float);
            }
        }
        RefreshHalfWeights();
    } else if (pi.plan.weights.size() != weight_floats_) {
        throw std::runtime_error("internal error: weight blob layout depends on the input shape");
    }
    std::vector<float>().swap(pi.plan.weights);   // the host copy of the blob is only needed for the first upload
    std::vector<char> is_io(pi.plan.buffer_floats.size(), 0);
    for (auto& d : pi.plan.inputs) is_io[size_t(d.view.buf)] = 1;
    for (auto& d : pi.plan.outputs) is_io[size_t(d.view.buf)] = 1;
    pi.buffers.assign(pi.plan.buffer_floats.size(), nullptr);
    pi.owned.assign(pi.plan.buffer_floats.size(), 0);
    for (size_t i = 0; i < pi.plan.buffer_floats.size(); ++i) {
        if (io_only && !is_io[i]) continue;
        float* p = nullptr;
        size_t bytes = size_t(std::max<int64_t>(pi.plan.buffer_floats[i], 8)) * (pi.plan.buffer_f16[i] ? 2 : 4);
        check(hipMalloc(reinterpret_cast<void**>(&p), bytes), "hipMalloc(activations)");
        check(hipMemsetAsync(p, 0, bytes, stream_), "hipMemset(activations)");
        device_bytes_ += bytes;
        pi.buffers[i] = p;
        pi.owned[i] = 1;
    }
    if (!io_only) {   // split-K scratch: slabs + per-tile arrival counters (tile-padded slabs need up to 2x the exact S*M*N)
        pi.workspace_floats = std::max<int64_t>(2 * pi.plan.workspace_floats, kTuneWorkspaceFloats);
        size_t bytes = size_t(pi.workspace_floats) * sizeof(float);
        check(hipMalloc(reinterpret_cast<void**>(&pi.workspace), bytes), "hipMalloc(workspace)");
        device_bytes_ += bytes;
        if (!two_pass_splitk_) {
            check(hipMalloc(reinterpret_cast<void**>(&pi.counters), kNumCounters * sizeof(int)), "hipMalloc(counters)");
            check(hipMemsetAsync(pi.counters, 0, kNumCounters * sizeof(int), stream_), "hipMemset(counters)");
        }
    }
}

PlanInstance& DeviceModel::Prepare(const std::vector<std::vector<int64_t>>& shapes) {
    std::vector<int64_t> key;
    for (auto& s : shapes) { key.push_back(int64_t(s.size())); key.insert(key.end(), s.begin(), s.end()); }
    auto touch = [&](const std::vector<int64_t>& k) {
        for (size_t i = 0; i < lru_.size(); ++i) if (lru_[i] == k) { lru_.erase(lru_.begin() + long(i)); break; }
        lru_.push_back(k);
    };
    auto it = plans_.find(key);
    if (it != plans_.end()) { touch(key); current_ = it->second.get(); return *current_; }
    while (plans_.size() >= max_plans_ && !lru_.empty()) {        // make room: drop the least recently used instance
        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        auto old = plans_.find(lru_.front());
        if (old != plans_.end()) {
            if (current_ == old->second.get()) current_ = nullptr;
            FreeInstance(*old->second);
            plans_.erase(old);
        }
        lru_.erase(lru_.begin());
    }

    check(hipSetDevice(device_), "hipSetDevice");
    // Sub-batch split: every graph input shares the leading batch dimension and it divides evenly.
    int nsub = sub_streams_;
    int64_t batch = shapes.empty() || shapes[0].empty() ? 0 : shapes[0][0];
    // Measured on MI355X (DenseNet-121): in fp16 mode the per-launch fixed costs (weight preamble, first loads, store drain) are half
    // of the forward, and two half-batches on two streams (IE_STREAMS=2) overlap them in back-to-back replays: batch 128 52.6k ->
    // 56.2k images/s (4 streams: 43k; fp32 batch 32: 10.77k -> 10.57k).
    // It is NOT the default: a single call's latency gets worse (p50 2.39 -> 2.65 ms per 128-image step, ModelInfer with UINT8
    // payloads 3.50 -> 3.65 ms) and the ABI path serves one request per model at a time.  IE_STREAMS=2 opts in.
    for (auto& s : shapes) if (s.empty() || s[0] != batch) nsub = 1;
    if (batch < 2 * nsub || batch % nsub != 0) nsub = 1;

    auto pi = std::make_unique<PlanInstance>();
    try {
        BuildInstance(*pi, shapes, nsub > 1);
        if (nsub > 1) {
            for (auto& d : pi->plan.outputs) if (d.dims.empty() || d.dims[0] != batch) nsub = 1;
        }
        if (nsub > 1) {
            std::vector<std::vector<int64_t>> sub_shapes = shapes;
            for (auto& s : sub_shapes) s[0] = batch / nsub;
            check(hipEventCreateWithFlags(&pi->fork, hipEventDisableTiming), "hipEventCreate");
            for (int k = 0; k < nsub; ++k) {
                auto sub = std::make_unique<PlanInstance>();
                try {
                    BuildInstance(*sub, sub_shapes, false);
                } catch (const std::exception&) {
                    // e.g. the model fixes its batch dimension: run the whole batch as one instance
                    FreeInstance(*sub);
                    for (auto& s2 : pi->subs) FreeInstance(*s2);
                    pi->subs.clear();
                    nsub = 1;
                    break;
                }
                sub->stream = k == 0 ? stream_ : side_streams_[size_t(k - 1)];
                check(hipEventCreateWithFlags(&sub->done, hipEventDisableTiming), "hipEventCreate");
                // alias the sub instance's I/O buffers to slices of the parent's full-size buffers
                auto alias = [&](const std::vector<IoDesc>& sub_io, const std::vector<IoDesc>& full_io) {
                    for (size_t i = 0; i < sub_io.size(); ++i) {
                        const size_t sb = size_t(sub_io[i].view.buf), fb = size_t(full_io[i].view.buf);
                        if (sub->buffers[sb] && sub->owned[sb]) (void)hipFree(sub->buffers[sb]);
                        sub->buffers[sb] = pi->buffers[fb] + int64_t(k) * sub_io[i].view.numel();
                        sub->owned[sb] = 0;
                    }
                };
                alias(sub->plan.inputs, pi->plan.inputs);
                alias(sub->plan.outputs, pi->plan.outputs);
                pi->subs.push_back(std::move(sub));
            }
        }
        if (nsub == 1 && pi->workspace == nullptr) {
            // nsub fell back to 1 after an io_only build: rebuild fully
            FreeInstance(*pi);
            pi = std::make_unique<PlanInstance>();
            BuildInstance(*pi, shapes, false);
        }
        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        if (autotune_) {
            if (pi->subs.empty()) Autotune(*pi);
            for (auto& sub : pi->subs) Autotune(*sub);
        }
        if (use_graph_) {
            hipGraph_t graph = nullptr;
            check(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
            try {
                RunSteps(*pi, nullptr, nullptr);
            } catch (...) {
                (void)hipStreamEndCapture(stream_, &graph);
                if (graph) (void)hipGraphDestroy(graph);
                throw;
            }
            check(hipStreamEndCapture(stream_, &graph), "hipStreamEndCapture");
            hipError_t e = hipGraphInstantiate(&pi->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            check(e, "hipGraphInstantiate");
            pi->graph_ready = true;
        }
    } catch (...) {
        FreeInstance(*pi);
        throw;
    }
    current_ = pi.get();
    plans_[key] = std::move(pi);
    touch(key);
    return *current_;
}

// Rebuilds what is derived from the fp32 weight blob: the half mirror (fp16 mode) or the fragment-major conv weights (fp32 mode).
void DeviceModel::RefreshHalfWeights() {
    if (!d_weights16_ && !d_weights_frag_) return;
    check(hipSetDevice(device_), "hipSetDevice");
    if (d_weights16_) check(LaunchConvertF32ToF16(d_weights_, d_weights16_, int64_t(weight_floats_), stream_), "convert_f32_f16");
    if (d_weights_frag_)
        for (const FragRegion& fr : frag_regions_)
            check(LaunchPermuteWeightsFrag(d_weights_ + fr.w_off, d_weights_frag_ + fr.w_off, fr.cout, fr.kk, fr.cin, stream_), "permute_weights_frag");
    check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

void DeviceModel::Autotune(PlanInstance& pi) {
    hipEvent_t e0, e1;
    check(hipEventCreate(&e0), "hipEventCreate");
    check(hipEventCreate(&e1), "hipEventCreate");
    static const int kSplits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
    constexpr size_t kScrubBytes = size_t(64) << 20;       // > 8 x 4 MiB of L2
    void* scrub = nullptr;
    if (const char* e = std::getenv("IE_TUNE_HOT"); !(e && std::atoi(e) != 0))
        if (hipMalloc(&scrub, kScrubBytes) != hipSuccess) { scrub = nullptr; (void)hipGetLastError(); }
    try {
        for (Step& s : pi.plan.steps) {
            if (s.kind != StepKind::Conv || s.algo == ConvAlgo::Naive || s.algo == ConvAlgo::Stem) continue;
            // the planner's default may already name a specialised kernel: the search starts from the tiled implicit GEMM either way
            if (s.algo == ConvAlgo::Ws1x1 || s.algo == ConvAlgo::Ws3x3 || s.algo == ConvAlgo::Direct || s.algo == ConvAlgo::Raster3x3) {
                s.algo = ConvAlgo::IgemmVec;
                s.tile = s.base_tile;
                s.splitk = 1;
            }
            const int64_t M = s.out.n * s.out.h * s.out.w, N = s.out.c;
            const int64_t bk = s.in.f16 ? 2 * kIgemmBK : kIgemmBK;
            const int64_t KT = s.algo == ConvAlgo::IgemmVec ? int64_t(s.kh) * s.kw * ((s.in.c + bk - 1) / bk)
                                                           : (int64_t(s.kh) * s.kw * s.in.c + kIgemmBK - 1) / kIgemmBK;
            std::vector<int64_t> key = {M, N, s.in.c, s.kh, s.kw, s.sh, s.sw, s.pt, s.pl, s.in.h, s.in.w, s.in.pitch, s.out.pitch,
                                        s.in.nchw, int64_t(s.algo), s.pre_scale_off >= 0, s.bias_off >= 0};
            if (s.in.f16 || s.out.f16) { key.push_back(s.in.f16); key.push_back(s.out.f16); }   // fp32 signatures keep 17 entries
            if (s.has_in2) key.push_back(1);              // a fused residual changes which kernels apply (18 / 20 entries)
            auto hit = tune_cache_.find(key);
            auto apply = [&](int enc_tile, int sp) {     // tile >= 100 encodes the raster kernel, >= 200 the weights-stationary 1x1
                if (enc_tile >= 400) { s.algo = ConvAlgo::Direct; s.tile = enc_tile - 400; }
                else if (enc_tile >= 300) { s.algo = ConvAlgo::Ws3x3; s.tile = enc_tile - 300; }
                else if (enc_tile >= 200) { s.algo = ConvAlgo::Ws1x1; s.tile = enc_tile - 200; }
                else if (enc_tile >= 100) { s.algo = ConvAlgo::Raster3x3; s.tile = enc_tile - 100; }
                else s.tile = enc_tile;
                s.splitk = sp;
            };
            if (hit != tune_cache_.end()) { apply(hit->second.first, hit->second.second); continue; }
            float best = 1e30f;
            int best_tile = s.tile, best_split = s.splitk;
            auto time_trial = [&](const Step& trial) {
                LaunchStep(pi, trial, stream_);              // warm
                float best_ms = 1e30f;
                if (scrub) {
                    // cold-cache protocol: in the real forward a layer finds neither its weights nor its input in L2 (the other 125
                    // layers ran in between); back-to-back repeats would flatter every kernel that re-reads operands from L2
                    for (int rep = 0; rep < 3; ++rep) {
                        check(hipMemsetAsync(scrub, 0, kScrubBytes, stream_), "hipMemsetAsync(scrub)");
                        check(hipEventRecord(e0, stream_), "hipEventRecord");
                        LaunchStep(pi, trial, stream_);
                        check(hipEventRecord(e1, stream_), "hipEventRecord");
                        check(hipEventSynchronize(e1), "hipEventSynchronize");
                        float ms = 0;
                        check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
                        best_ms = std::min(best_ms, ms);
                    }
                    return best_ms;
                }
                for (int rep = 0; rep < 2; ++rep) {          // best of two timed triples: robust against one-off hiccups
                    check(hipEventRecord(e0, stream_), "hipEventRecord");
                    for (int r = 0; r < 3; ++r) LaunchStep(pi, trial, stream_);
                    check(hipEventRecord(e1, stream_), "hipEventRecord");
                    check(hipEventSynchronize(e1), "hipEventSynchronize");
                    float ms = 0;
                    check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
                    best_ms = std::min(best_ms, ms);
                }
                return best_ms;
            };
            // LDS-window kernel for 3x3/s1/p1 convs without an activation prologue
            if (!s.in.f16 && !s.out.f16 && s.algo == ConvAlgo::IgemmVec && s.kh == 3 && s.kw == 3 && s.sh == 1 && s.sw == 1 && s.pt == 1 && s.pl == 1 && s.pb == 1 &&
                s.pr == 1 && s.pre_scale_off < 0) {
                ConvArgs probe;
                probe.in = make_arg(pi, s.in);
                probe.out = make_arg(pi, s.out);
                probe.w = d_weights_ + s.w_off;
                probe.kh = 3; probe.kw = 3; probe.pt = 1; probe.pl = 1;
                const int64_t chunks = (s.in.c + kIgemmBK - 1) / kIgemmBK;
                const int64_t Mr = s.in.n * (s.in.h + 1) * (s.in.w + 1);
                for (int t = 0; t < kNumConvRasterTiles; ++t) {
                    if (!ConvRasterEligible(probe, t)) continue;
                    const int bn = ConvRasterTileBn(t);
                    if ((bn > 32 && N <= 32)) continue;
                    for (int sp : {1, 2, 4, 8}) {
                        if (sp > chunks) continue;
                        if (sp > 1 && Mr / 64 * sp > 16384) continue;
                        if (sp > 1 && int64_t(sp) * (M + 256) * (N + 64) * 2 > pi.workspace_floats) continue;
                        Step trial = s;
                        trial.algo = ConvAlgo::Raster3x3;
                        trial.tile = t;
                        trial.splitk = sp;
                        float ms = time_trial(trial);
                        if (ms < best) { best = ms; best_tile = 100 + t; best_split = sp; }
                    }
                }
            }
            // the kernels_direct.hip family (every variant checks its own pixel-count / shape limits): K split over the waves with
            // operands straight from global memory, LDS-window tiles, activations-stationary 1x1, window + streamed weights
            if (s.algo == ConvAlgo::IgemmVec) {
                ConvArgs probe = MakeConvArgs(pi, s);
                probe.res = TensorArg();                   // LaunchStep adds the shortcut in a second kernel for this family
                for (int t = 0; t < kNumConvDirectTiles; ++t) {
                    if (!ConvDirectEligible(probe, t)) continue;
                    Step trial = s;
                    trial.algo = ConvAlgo::Direct;
                    trial.tile = t;
                    trial.splitk = 1;
                    float ms = time_trial(trial);
                    if (ms < best) { best = ms; best_tile = 400 + t; best_split = 1; }
                }
            }
            // 1x1/s1: weights-stationary streaming kernel (either precision)
            if (s.algo == ConvAlgo::IgemmVec && s.kh == 1 && s.kw == 1) {
                ConvArgs probe = MakeConvArgs(pi, s);
                for (int t = 0; t < (s.in.f16 ? kNumConvWsTiles : kNumConvWs32Tiles); ++t) {
                    if (!(s.in.f16 ? ConvWsEligible(probe, t) : ConvWs32Eligible(probe, t))) continue;
                    Step trial = s;
                    trial.algo = ConvAlgo::Ws1x1;
                    trial.tile = t;
                    trial.splitk = 1;
                    float ms = time_trial(trial);
                    if (ms < best) { best = ms; best_tile = 200 + t; best_split = 1; }
                }
            }
            if (s.in.f16 && s.algo == ConvAlgo::IgemmVec && s.kh == 3 && s.kw == 3) {
                ConvArgs probe = MakeConvArgs(pi, s);
                for (int t = 0; t < kNumConvWs3Tiles; ++t) {
                    if (!ConvWs3Eligible(probe, t)) continue;
                    Step trial = s;
                    trial.algo = ConvAlgo::Ws3x3;
                    trial.tile = t;
                    trial.splitk = 1;
                    float ms = time_trial(trial);
                    if (ms < best) { best = ms; best_tile = 300 + t; best_split = 1; }
                }
            }
            for (int t = 0; t < kNumIgemmTiles; ++t) {
                const IgemmTile& T = kIgemmTiles[t];
                if ((T.bn > 32 && N <= 32) || (T.bn > 64 && N <= 64)) continue;
                if ((T.kg > 1 || T.deep) && (s.algo != ConvAlgo::IgemmVec || KT < 2 * T.kg)) continue;
                if (T.deep && s.in.f16) continue;         // the fp16 kernel has no deep-prefetch variants
                const int64_t wgs = ((M + T.bm - 1) / T.bm) * ((N + T.bn - 1) / T.bn);
                if (T.deep && wgs > 1024) continue;       // the deep-prefetch variants target grids that cannot fill the chip
                for (int sp : kSplits) {
                    if (sp > 1 && T.kg > 1 && ((!two_pass_splitk_ && !s.in.f16) || KT / (sp * T.kg) < 2)) continue;
                    if (sp > 1 && (KT / sp < 2 || int64_t(sp) * wgs * T.bm * T.bn > pi.workspace_floats || wgs > kNumCounters ||
                                   wgs * sp > 8192 || wgs >= 1024))
                        continue;
                    Step trial = s;
                    trial.tile = t;
                    trial.splitk = sp;
                    float ms = time_trial(trial);
                    if (ms < best) { best = ms; best_tile = t; best_split = sp; }
                }
            }
            apply(best_tile, best_split);
            tune_cache_[key] = {best_tile, best_split};
        }
    } catch (...) {
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (scrub) (void)hipFree(scrub);
        throw;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (scrub) (void)hipFree(scrub);
    if (const char* tc = std::getenv("IE_TUNE_CACHE")) {
        std::ofstream f(tc, std::ios::trunc);
        for (auto& kv : tune_cache_) {
            for (auto v : kv.first) f << v << ' ';
            f << ": " << kv.second.first << ' ' << kv.second.second << '\n';
        }
    }
}

ConvArgs DeviceModel::MakeConvArgs(const PlanInstance& pi, const Step& s) const {
    const float* wb = d_weights_;
    auto wp = [&](int64_t off) -> const float* { return off >= 0 ? wb + off : nullptr; };
    ConvArgs a;
    a.in = make_arg(pi, s.in);
    a.out = make_arg(pi, s.out);
    if (s.has_in2) a.res = make_arg(pi, s.in2);
    a.w = wp(s.w_off);
    a.w16 = d_weights16_ && s.w_off >= 0 ? static_cast<const char*>(d_weights16_) + s.w_off * 2 : nullptr;
    a.wfrag = d_weights_frag_ && s.w_off >= 0 && s.out.c % 16 == 0 && s.in.c % 16 == 0 && s.kh * s.kw <= 49 ? d_weights_frag_ + s.w_off : nullptr;
    a.bias = wp(s.bias_off);
    a.pre_scale = wp(s.pre_scale_off);
    a.pre_shift = wp(s.pre_shift_off);
    if (d_weights16_ && s.pre_scale_off >= 0) {
        a.pre_scale16 = static_cast<const char*>(d_weights16_) + s.pre_scale_off * 2;
        a.pre_shift16 = static_cast<const char*>(d_weights16_) + s.pre_shift_off * 2;
    }
    a.kh = s.kh; a.kw = s.kw; a.sh = s.sh; a.sw = s.sw; a.pt = s.pt; a.pl = s.pl;
    a.pre_relu = s.pre_relu; a.relu = s.relu;
    a.workspace = pi.workspace;
    a.workspace_floats = pi.workspace_floats;
    a.counters = pi.counters;
    a.num_counters = pi.counters ? kNumCounters : 0;
    return a;
}

void DeviceModel::LaunchStep(const PlanInstance& pi, const Step& s_in, hipStream_t stream_) {
    const Step& s = s_in;
    const float* wb = d_weights_;
    auto wp = [&](int64_t off) -> const float* { return off >= 0 ? wb + off : nullptr; };
    switch (s.kind) {
        case StepKind::Conv: {
            ConvArgs a = MakeConvArgs(pi, s);
            // A specialised launcher that declines this operand set (alignment, LDS budget ...) hands the step to the tiled kernel.
            Step fb;
            const Step* sp = &s_in;
            {
                bool ok = true;
                ConvArgs plain = a;                        // what a kernel without a residual epilogue is asked to do (see split_res below)
                plain.res = TensorArg();
                switch (s_in.algo) {
                    case ConvAlgo::Ws1x1: ok = s_in.in.f16 ? ConvWsEligible(a, s_in.tile) : ConvWs32Eligible(a, s_in.tile); break;
                    case ConvAlgo::Ws3x3: ok = ConvWs3Eligible(plain, s_in.tile); break;
                    case ConvAlgo::Direct: ok = ConvDirectEligible(plain, s_in.tile); break;
                    case ConvAlgo::Raster3x3: ok = ConvRasterEligible(plain, s_in.tile); break;
                    case ConvAlgo::Stem: ok = ConvStemEligible(a); break;
                    default: break;
                }
                if (!ok) {
                    fb = s_in;
                    fb.algo = s_in.algo == ConvAlgo::Stem ? ConvAlgo::IgemmScalar : ConvAlgo::IgemmVec;
                    fb.tile = s_in.base_tile;
                    fb.splitk = 1;
                    sp = &fb;
                }
            }
            const Step& s = *sp;
            // A fused residual Add lives in the weights-stationary 1x1 epilogues; any other kernel runs the conv without its ReLU
            // and adds the shortcut in place afterwards.
            const TensorArg res = a.res;
            const int relu = a.relu;
            const bool split_res = res.p != nullptr && s.algo != ConvAlgo::Ws1x1;
            if (split_res) { a.res = TensorArg(); a.relu = 0; }
            if (s.algo == ConvAlgo::Naive) check(LaunchConvNaive(a, stream_), "conv_naive");
            else if (s.algo == ConvAlgo::Raster3x3) check(LaunchConvRaster3x3(a, s.tile, s.splitk, stream_), "conv3x3_raster");
            else if (s.algo == ConvAlgo::Ws1x1 && s.in.f16) check(LaunchConvWs1x1F16(a, s.tile, stream_), "conv1x1_ws_f16");
            else if (s.algo == ConvAlgo::Ws1x1) check(LaunchConvWs1x1F32(a, s.tile, stream_), "conv1x1_ws_f32");
            else if (s.algo == ConvAlgo::Ws3x3) check(LaunchConvWs3x3F16(a, s.tile, stream_), "conv3x3_ws_f16");
            else if (s.algo == ConvAlgo::Stem) check(LaunchConvStem(a, stream_), "conv_stem");
            else if (s.algo == ConvAlgo::Direct) check(LaunchConvDirect(a, s.tile, stream_), "conv_direct");
            else if (s.in.f16) check(LaunchConvIgemmF16(a, s.tile, s.splitk, stream_), "conv_igemm_f16");
            else check(LaunchConvIgemm(a, s.tile, s.algo == ConvAlgo::IgemmVec ? 1 : 0, s.splitk, stream_), "conv_igemm");
            if (split_res) {
                EltArgs e;
                e.a = a.out; e.b = res; e.out = a.out; e.relu = relu;
                check(LaunchEltwise(e, stream_), "eltwise(residual)");
            }
            break;
        }
        case StepKind::Pool: {
            PoolArgs a;
            a.in = make_arg(pi, s.in);
            a.out = make_arg(pi, s.out);
            a.kh = s.kh; a.kw = s.kw; a.sh = s.sh; a.sw = s.sw; a.pt = s.pt; a.pl = s.pl; a.pb = s.pb; a.pr = s.pr;
            a.is_max = s.pool_max; a.count_include_pad = s.count_include_pad;
            a.pre_scale = wp(s.pre_scale_off); a.pre_shift = wp(s.pre_shift_off); a.pre_relu = s.pre_relu;
            check(LaunchPool(a, stream_), "pool");
            break;
        }
        case StepKind::GlobalAvgPool:
            check(LaunchGlobalAvgPool(make_arg(pi, s.in), make_arg(pi, s.out), wp(s.pre_scale_off), wp(s.pre_shift_off),
                                      s.pre_relu, stream_), "global_avg_pool");
            break;
        case StepKind::Eltwise: {
            EltArgs a;
            a.a = make_arg(pi, s.in);
            if (s.has_in2) a.b = make_arg(pi, s.in2);
            a.out = make_arg(pi, s.out);
            a.scale = wp(s.pre_scale_off);
            a.shift = wp(s.pre_shift_off);
            a.relu = s.relu;
            check(LaunchEltwise(a, stream_), "eltwise");
            break;
        }
        case StepKind::Copy:
            check(LaunchCopy(make_arg(pi, s.in), make_arg(pi, s.out), stream_), "copy");
            break;
    }
}

static std::string kernel_label(const Step& s) {
    switch (s.kind) {
        case StepKind::Conv:
            if (s.algo == ConvAlgo::Naive) return "conv_naive_kernel";
            if (s.algo == ConvAlgo::Direct) {       // one launcher family, three kernels (kernels_direct.hip): report the one that runs
                const char* k = s.tile >= 10 ? "conv1x1_as_kernel<f32,t" : s.tile >= kNumDirectBaseTiles ? "conv_win_kernel<f32,t"
                                             : s.in.f16 ? "conv_direct_kernel<f16,t" : "conv_direct_kernel<f32,t";
                return std::string(k) + std::to_string(s.tile) + ">";
            }
            if (s.algo == ConvAlgo::Stem) return s.out.f16 ? "conv_stem_kernel<f16>" : "conv_stem_kernel<f32>";
            if (s.algo == ConvAlgo::Ws1x1) return std::string(s.in.f16 ? "conv1x1_ws_f16_kernel<t" : "conv1x1_ws_f32_kernel<t") + std::to_string(s.tile) + ">";
            if (s.algo == ConvAlgo::Ws3x3) return "conv3x3_ws_f16_kernel<t" + std::to_string(s.tile) + ">";
            if (s.algo == ConvAlgo::Raster3x3)
                return "conv3x3_raster_kernel<t" + std::to_string(s.tile) + (s.splitk > 1 ? ",splitk" + std::to_string(s.splitk) : std::string()) + ">";
            return std::string(s.in.f16 ? "conv_igemm_f16_kernel<" : "conv_igemm_kernel<") + std::to_string(kIgemmTiles[s.tile].bm) + "x" +
                   std::to_string(kIgemmTiles[s.tile].bn) + (kIgemmTiles[s.tile].kg > 1 ? "x" + std::to_string(kIgemmTiles[s.tile].kg) + "kg" : std::string()) +
                   (kIgemmTiles[s.tile].deep ? ",deep" : "") +
                   (s.algo == ConvAlgo::IgemmVec ? ",vec" : ",scalar") +
                   (s.splitk > 1 ? ",splitk" + std::to_string(s.splitk) : std::string()) + ">";
        case StepKind::Pool: return "pool_kernel";
        case StepKind::GlobalAvgPool: return "gap_kernel";
        case StepKind::Eltwise: return "eltwise_kernel";
        case StepKind::Copy: return "copy_kernel";
    }
    return "?";
}

void DeviceModel::RunSteps(PlanInstance& pi, std::vector<StepTiming>* timings, std::vector<hipEvent_t>* events) {
    (void)timings;
    if (!pi.subs.empty() && !events) {
        // fork: side streams wait for everything enqueued on the main stream so far; join: main waits for every sub-batch
        check(hipEventRecord(pi.fork, stream_), "hipEventRecord");
        for (auto& sub : pi.subs) {
            if (sub->stream != stream_) check(hipStreamWaitEvent(sub->stream, pi.fork, 0), "hipStreamWaitEvent");
            for (const Step& s : sub->plan.steps) LaunchStep(*sub, s, sub->stream);
            if (sub->stream != stream_) check(hipEventRecord(sub->done, sub->stream), "hipEventRecord");
        }
        for (auto& sub : pi.subs)
            if (sub->stream != stream_) check(hipStreamWaitEvent(stream_, sub->done, 0), "hipStreamWaitEvent");
        return;
    }
    // single instance, or instrumented pass (sub-batches one after the other on the main stream, an event after every launch)
    size_t k = 0;
    if (events) check(hipEventRecord((*events)[k++], stream_), "hipEventRecord");
    if (pi.subs.empty()) {
        for (const Step& s : pi.plan.steps) {
            LaunchStep(pi, s, stream_);
            if (events) check(hipEventRecord((*events)[k++], stream_), "hipEventRecord");
        }
    } else {
        for (auto& sub : pi.subs)
            for (const Step& s : sub->plan.steps) {
                LaunchStep(*sub, s, stream_);
                if (events) check(hipEventRecord((*events)[k++], stream_), "hipEventRecord");
            }
    }
}

void DeviceModel::Enqueue(PlanInstance& pi) {
    check(hipSetDevice(device_), "hipSetDevice");
    if (pi.graph_ready) check(hipGraphLaunch(pi.graph_exec, stream_), "hipGraphLaunch");
    else RunSteps(pi, nullptr, nullptr);
}

void DeviceModel::Synchronize() {
    check(hipSetDevice(device_), "hipSetDevice");
    check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

std::vector<StepTiming> DeviceModel::Profile(PlanInstance& pi, int iters) {
    check(hipSetDevice(device_), "hipSetDevice");
    std::vector<const Step*> steps;
    if (pi.subs.empty()) for (const Step& s : pi.plan.steps) steps.push_back(&s);
    else for (auto& sub : pi.subs) for (const Step& s : sub->plan.steps) steps.push_back(&s);
    const size_t ns = steps.size();
    std::vector<hipEvent_t> ev(ns + 1);
    for (auto& e : ev) check(hipEventCreate(&e), "hipEventCreate");
    std::vector<StepTiming> out(ns);
    for (size_t i = 0; i < ns; ++i) {
        out[i].name = steps[i]->name;
        out[i].kernel = kernel_label(*steps[i]);
        out[i].flops = steps[i]->flops;
        out[i].bytes = steps[i]->bytes;
    }
    try {
        RunSteps(pi, nullptr, nullptr);   // warm
        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        for (int it = 0; it < iters; ++it) {
            RunSteps(pi, nullptr, &ev);
            check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
            for (size_t i = 0; i < ns; ++i) {
                float ms = 0;
                check(hipEventElapsedTime(&ms, ev[i], ev[i + 1]), "hipEventElapsedTime");
                out[i].ms += ms / float(iters);
            }
        }
    } catch (...) {
        for (auto& e : ev) (void)hipEventDestroy(e);
        throw;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return out;
}

void DeviceModel::InferHost(PlanInstance& pi, const std::vector<const void*>& inputs, const std::vector<size_t>& in_bytes,
                            const std::vector<void*>& outputs, const std::vector<size_t>& out_bytes, const std::vector<char>& in_u8) {
    std::vector<std::vector<InSeg>> in(pi.plan.inputs.size());
    for (size_t i = 0; i < pi.plan.inputs.size(); ++i) {
        const bool u8 = i < in_u8.size() && in_u8[i];
        const size_t need = size_t(pi.plan.inputs[i].view.numel()) * (u8 ? 1 : sizeof(float));
        in[i].push_back({inputs[i], inputs[i] ? std::min(in_bytes[i], need) : 0, need, 0, u8});
    }
    std::vector<std::vector<OutSeg>> out(pi.plan.outputs.size());
    for (size_t i = 0; i < outputs.size() && i < pi.plan.outputs.size(); ++i) {
        if (!outputs[i] || out_bytes[i] == 0) continue;
        out[i].push_back({outputs[i], out_bytes[i], size_t(pi.plan.outputs[i].view.numel()) * sizeof(float), 0});
    }
    InferHostSegments(pi, in, out);
}

void DeviceModel::InferHostSegments(PlanInstance& pi, const std::vector<std::vector<InSeg>>& in, const std::vector<std::vector<OutSeg>>& out) {
    check(hipSetDevice(device_), "hipSetDevice");
    hipEvent_t slot_ev[kSlots];
    for (auto& e : slot_ev) check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    bool slot_used[kSlots] = {false, false, false, false};
    int slot = 0;
    auto fail_cleanup = [&] { for (auto& e : slot_ev) (void)hipEventDestroy(e); };
    try {
        // ---- H2D: caller memory -> pinned ring -> device, CPU copy of chunk k+1 overlaps the DMA of chunk k ----
        for (size_t i = 0; i < pi.plan.inputs.size() && i < in.size(); ++i) {
            const View& v = pi.plan.inputs[i].view;
            char* base = reinterpret_cast<char*>(pi.buffers[size_t(v.buf)]);
            const bool u8 = !in[i].empty() && in[i][0].u8;
            if (u8) {       // bytes go to a device staging buffer; one kernel converts them into the fp32 input buffer afterwards
                if (pi.u8_stage.size() < pi.plan.inputs.size()) pi.u8_stage.resize(pi.plan.inputs.size(), nullptr);
                if (!pi.u8_stage[i]) {
                    check(hipMalloc(&pi.u8_stage[i], std::max<size_t>(size_t(v.numel()), 16)), "hipMalloc(u8 staging)");
                    device_bytes_ += size_t(v.numel());
                }
                base = static_cast<char*>(pi.u8_stage[i]);
            }
            for (const InSeg& sg : in[i]) {
                if (sg.u8 != u8) throw std::runtime_error("internal error: mixed UINT8 / FLOAT32 segments for one input");
                char* dst = base + sg.dev_off;
                const size_t have = sg.host ? std::min(sg.have, sg.need) : 0;
                for (size_t off = 0; off < have; off += kChunk) {
                    const size_t nb = std::min(kChunk, have - off);
                    if (slot_used[slot]) check(hipEventSynchronize(slot_ev[slot]), "hipEventSynchronize");
                    char* stage = static_cast<char*>(pinned_) + size_t(slot) * kChunk;
                    copy_pool_->Copy(stage, static_cast<const char*>(sg.host) + off, nb);
                    check(hipMemcpyAsync(dst + off, stage, nb, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(H2D)");
                    check(hipEventRecord(slot_ev[slot], stream_), "hipEventRecord");
                    slot_used[slot] = true;
                    slot = (slot + 1) % kSlots;
                }
                if (have < sg.need) check(hipMemsetAsync(dst + have, 0, sg.need - have, stream_), "hipMemsetAsync");
            }
            if (u8)
                check(LaunchConvertU8ToF32(pi.u8_stage[i], pi.buffers[size_t(v.buf)], v.numel(), u8_scale_, u8_bias_, stream_), "convert_u8_f32");
        }
        Enqueue(pi);
        // ---- D2H: one transfer per output when it fits the ring, then per-caller scatter on the host ----------
        for (size_t j = 0; j < pi.plan.outputs.size() && j < out.size(); ++j) {
            if (out[j].empty()) continue;
            const View& v = pi.plan.outputs[j].view;
            const char* src = reinterpret_cast<const char*>(pi.buffers[size_t(v.buf)]);
            size_t lo = SIZE_MAX, hi = 0;
            for (const OutSeg& sg : out[j]) {
                const size_t nb = std::min(sg.cap, sg.need);
                if (!nb) continue;
                lo = std::min(lo, sg.dev_off);
                hi = std::max(hi, sg.dev_off + nb);
            }
            if (hi > lo && hi - lo <= kChunk * kSlots) {
                check(hipStreamSynchronize(stream_), "hipStreamSynchronize");     // ring is free again
                check(hipMemcpyAsync(pinned_, src + lo, hi - lo, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H)");
                check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
                for (const OutSeg& sg : out[j]) {
                    const size_t nb = std::min(sg.cap, sg.need);
                    if (nb) std::memcpy(sg.host, static_cast<char*>(pinned_) + (sg.dev_off - lo), nb);
                }
            } else {
                for (const OutSeg& sg : out[j]) {
                    const size_t nbytes = std::min(sg.cap, sg.need);
                    for (size_t off = 0; off < nbytes; off += kChunk * kSlots) {
                        const size_t nb = std::min(kChunk * kSlots, nbytes - off);
                        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
                        check(hipMemcpyAsync(pinned_, src + sg.dev_off + off, nb, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H)");
                        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
                        std::memcpy(static_cast<char*>(sg.host) + off, pinned_, nb);
                    }
                }
            }
            for (const OutSeg& sg : out[j]) {
                const size_t nb = std::min(sg.cap, sg.need);
                if (sg.cap > nb) std::memset(static_cast<char*>(sg.host) + nb, 0, sg.cap - nb);
            }
        }
        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    } catch (...) {
        (void)hipStreamSynchronize(stream_);
        fail_cleanup();
        throw;
    }
    fail_cleanup();
}

}  // namespace ie
