#include "executor.h"
#include "env.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <unistd.h>
#include <sstream>
#include <stdexcept>

#include "igemm_tiles.h"
#include "kernels.h"

namespace ie {
namespace {

void check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

// Kernel operand for a planned view.  A chunk instance of the pipelined host path (PlanInstance::batch_off > 0) addresses its
// image range inside the parent's buffers: same layout, pointer advanced by batch_off images.
TensorArg make_arg(const PlanInstance& pi, const View& v) {
    TensorArg t;
    char* base = reinterpret_cast<char*>(pi.buffers.at(size_t(v.buf)));
    t.n = int(v.n); t.h = int(v.h); t.w = int(v.w); t.c = int(v.c);
    t.f16 = v.f16 ? 1 : 0;
    t.f8 = v.f8 ? 1 : 0;
    const int64_t esize = v.esize();
    if (v.nchw) {
        t.sw = 1; t.sh = v.w; t.sc = v.h * v.w; t.sn = v.c * v.h * v.w;
        t.p = reinterpret_cast<float*>(base + pi.batch_off * t.sn * esize);
    } else {
        t.sc = 1; t.sw = v.pitch; t.sh = v.w * v.pitch; t.sn = v.h * v.w * v.pitch;
        t.p = reinterpret_cast<float*>(base + (v.c_off + pi.batch_off * t.sn) * esize);
    }
    return t;
}

constexpr size_t kPinnedBytes = size_t(4) << 20;              // pinned staging for results (logits are KBs; larger outputs go direct)
constexpr int64_t kTuneWorkspaceFloats = int64_t(16) << 20;   // 64 MiB of split-K slabs available to the autotuner
constexpr int kNumCounters = 1 << 16;
constexpr int kMaxChunks = 8;
constexpr const char* kTuneFileTag = "# ie-tune-v2";

std::once_flag g_kernels_once;
hipError_t g_kernels_err = hipSuccess;

}  // namespace

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

DeviceWeights::~DeviceWeights() {
    (void)hipSetDevice(device);
    if (d_weights) (void)hipFree(d_weights);
    if (d_weights16) (void)hipFree(d_weights16);
    if (d_weights16_frag) (void)hipFree(d_weights16_frag);
    if (d_weights_frag) (void)hipFree(d_weights_frag);
    if (d_weights8) (void)hipFree(d_weights8);
    if (d_f8_aux) (void)hipFree(d_f8_aux);
    if (d_weights_wino) (void)hipFree(d_weights_wino);
    if (d_weights_x6) (void)hipFree(d_weights_x6);
    if (d_weights_wino_x6) (void)hipFree(d_weights_wino_x6);
}

namespace {

std::string tune_file_header() {
    std::ostringstream o;
    o << kTuneFileTag << ' ' << kNumIgemmTiles << ' ' << kNumConvRasterTiles << ' ' << kNumConvWs32Tiles << ' ' << kNumConvWs16Tiles << ' ' << kNumConvWs3Tiles << ' '
      << kNumConvDirectTiles << ' ' << kNumConvWinoTiles << ' ' << kNumConvX6Tiles << ' ' << kNumConvWs8Tiles << ' ' << kNumConvWs38Tiles;
    return o.str();
}

// "<signature ints> : <encoded tile> <splitk>" per line behind a header naming the tile tables the numbers index into; a file
// written by an engine with other tables is ignored as a whole.
void load_tune_file(const std::string& path, std::map<std::vector<int64_t>, std::pair<int, int>>& cache) {
    std::ifstream f(path);
    std::string line;
    if (!f || !std::getline(f, line) || line != tune_file_header()) return;
    while (std::getline(f, line)) {
        std::istringstream is(line);
        std::vector<int64_t> key;
        std::string tok;
        bool ok = true;
        while (is >> tok && tok != ":") {
            char* endp = nullptr;
            const long long v = std::strtoll(tok.c_str(), &endp, 10);
            if (!endp || *endp) { ok = false; break; }
            key.push_back(v);
        }
        int t = -1, sp = 0;
        if (ok && (is >> t >> sp) && ((t >= 0 && t < kNumIgemmTiles) || (t >= 100 && t < 100 + (kNumConvRasterTiles > kNumConvWs8Tiles ? kNumConvRasterTiles : kNumConvWs8Tiles)) ||      // (fp32: raster tiles; fp8 files: conv1x1_ws_f8 tiles)
                                        (t >= 200 && t < 200 + (kNumConvWs16Tiles > kNumConvWs32Tiles ? kNumConvWs16Tiles : kNumConvWs32Tiles)) ||
                                       (t >= 300 && t < 300 + kNumConvWs3Tiles) || (t >= 400 && t < 400 + kNumConvDirectTiles) || (t >= 500 && t < 500 + kNumConvWinoTiles) || (t >= 600 && t < 600 + kNumConvX6Tiles)) &&
            sp >= 1 && sp <= 64 && key.size() >= 6)          // (the fused steps have short signatures: stem + pool 7 numbers, dense block 10, dual 9)
            cache[key] = {t, sp};
    }
}

}  // namespace

DeviceModel::DeviceModel(std::shared_ptr<const OnnxModel> model, int device_id, const DeviceModelOptions& opt)
    : model_(std::move(model)), device_(device_id), precision_(opt.precision) {
    env_ = Env::Read();
    SetLaunchKnobs(env_);
    max_inflight_replays_ = std::max(0, env_.integer("IE_MAX_INFLIGHT_REPLAYS", 0));
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
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsF8();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsFused();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsWino();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsX6();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsBlock();
        if (g_kernels_err == hipSuccess) g_kernels_err = InitKernelsWs8();
    });
    check(g_kernels_err, "InitKernels");
    fp32_split_ = opt.fp32_split && opt.precision == Precision::F32;
    if (const char* e = env_.get("IE_FP32_SPLIT")) fp32_split_ = std::atoi(e) != 0 && opt.precision == Precision::F32;
    if (opt.share) {
        if (opt.share->device != device_) throw std::runtime_error("internal error: shared weights live on another device");
        w_ = opt.share;
    } else {
        w_ = std::make_shared<DeviceWeights>();
        w_->device = device_;
        w_->uploaded = false;
        owns_weights_ = true;
        // Persistent kernel-choice cache: IE_TUNE_CACHE=<file> (""/"0" = none), else the path the bridge derived from the model directory.
        std::string path = opt.tune_cache_path;
        if (const char* tc = env_.get("IE_TUNE_CACHE")) path = (tc[0] == 0 || (tc[0] == '0' && tc[1] == 0)) ? std::string() : std::string(tc);
        if (fp32_split_ && !path.empty()) path += ".x6";       // choices made with the bf16x6 kernels in the search are their own file
        w_->tune_cache_path = path;
        if (!path.empty()) load_tune_file(path, w_->tune_cache);
    }
    upload_weights_ = opt.upload_weights;
    check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    check(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking), "hipStreamCreate");
    h2d_events_.resize(kMaxChunks, nullptr);
    for (auto& e : h2d_events_) check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    check(hipEventCreate(&t0_event_), "hipEventCreate");
    check(hipEventCreate(&t1_event_), "hipEventCreate");
    const char* ng = env_.get("IE_DISABLE_GRAPH");
    use_graph_ = !(ng && ng[0] == '1');
    const char* at = env_.get("IE_AUTOTUNE");
    // Measured on MI355X (DenseNet-121 B=32): the in-launch combine (agent-scope release/acquire per tile) costs more than
    // the kernel boundary it removes: 3.63 ms/step vs 3.42 ms/step with the separate reduce kernel.  Two-pass is the default.
    const char* tp = env_.get("IE_SPLITK_IN_LAUNCH");
    two_pass_splitk_ = !(tp && tp[0] == '1');
    autotune_ = !(at && at[0] == '0') && !env_.get("IE_FORCE_TILE") && !env_.get("IE_FORCE_SPLITK") && !env_.get("IE_FORCE_ALGO");
    if (const char* od = env_.get("IE_TUNE_ON_DEMAND")) tune_on_demand_ = od[0] == '1';
    if (const char* pc = env_.get("IE_PIPELINE_CHUNKS")) pipeline_chunks_ = std::max(0, std::min(kMaxChunks, std::atoi(pc)));
    if (const char* ph = env_.get("IE_PIPELINE_HEAD")) pipeline_head_ = std::max(0, std::atoi(ph));
    if (const char* mp = env_.get("IE_MAX_PLANS")) max_plans_ = size_t(std::max(1, std::atoi(mp)));
    pinned_bytes_ = kPinnedBytes;
    check(hipHostMalloc(&pinned_, pinned_bytes_, hipHostMallocDefault), "hipHostMalloc");
}

DeviceModel::~DeviceModel() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
    for (auto& kv : plans_) FreeInstance(*kv.second);
    for (auto e : h2d_events_) if (e) (void)hipEventDestroy(e);
    if (t0_event_) (void)hipEventDestroy(t0_event_);
    if (t1_event_) (void)hipEventDestroy(t1_event_);
    if (pinned_) (void)hipHostFree(pinned_);
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void DeviceModel::FreeInstance(PlanInstance& pi) {
    for (auto& ch : pi.chunks) FreeInstance(*ch);
    pi.chunks.clear();
    if (pi.graph_exec) (void)hipGraphExecDestroy(pi.graph_exec);
    if (pi.tail_exec) (void)hipGraphExecDestroy(pi.tail_exec);
    pi.graph_exec = nullptr;
    pi.tail_exec = nullptr;
    for (size_t i = 0; i < pi.buffers.size(); ++i)
        if (pi.buffers[i] && pi.owned[i]) (void)hipFree(pi.buffers[i]);
    pi.buffers.clear();
    if (pi.workspace && pi.owns_workspace) (void)hipFree(pi.workspace);
    if (pi.counters && pi.owns_workspace) (void)hipFree(pi.counters);
    pi.workspace = nullptr;
    pi.counters = nullptr;
    for (void* p : pi.u8_stage) if (p) (void)hipFree(p);
    pi.u8_stage.clear();
}

void DeviceModel::CopySync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, const char* what) {
    if (bytes == 0) return;
    check(hipMemcpyAsync(dst, src, bytes, kind, stream_), what);
    check(hipStreamSynchronize(stream_), what);
}

void DeviceModel::ZeroSync(void* dst, size_t bytes, const char* what) {
    if (bytes == 0) return;
    check(hipMemsetAsync(dst, 0, bytes, stream_), what);
    check(hipStreamSynchronize(stream_), what);
}

// Plan + allocate one instance (and, the first time on this device, put the packed weights into HBM).
void DeviceModel::BuildInstance(PlanInstance& pi, const std::vector<std::vector<int64_t>>& shapes) {
    pi.plan = BuildPlan(*model_, shapes, precision_);
    {
        std::lock_guard<std::mutex> g(w_->tune_mu);     // lanes of one device may be built concurrently
        if (!w_->d_weights) {
            w_->weight_floats = pi.plan.weights.size();
            check(hipMalloc(reinterpret_cast<void**>(&w_->d_weights), std::max<size_t>(w_->weight_floats, 4) * sizeof(float)), "hipMalloc(weights)");
            w_->device_bytes += w_->weight_floats * sizeof(float);
            if (upload_weights_) {
                CopySync(w_->d_weights, pi.plan.weights.data(), w_->weight_floats * sizeof(float), hipMemcpyHostToDevice, "hipMemcpy(weights)");
                w_->uploaded = true;
            } else {
                ZeroSync(w_->d_weights, w_->weight_floats * sizeof(float), "hipMemset(weights)");
            }
            if (precision_ == Precision::F16 || precision_ == Precision::F8) {
                check(hipMalloc(&w_->d_weights16, std::max<size_t>(w_->weight_floats, 8) * 2), "hipMalloc(weights16)");
                w_->device_bytes += w_->weight_floats * 2;
                if (precision_ == Precision::F16) {
                    // fragment-major mirror of the dense-layer convs (1x1 -> 128 and 3x3 128 -> 32 channels), whatever step they are part of in
                    // this plan instance: the weights are shared by every plan instance, another batch size may fuse other layers
                    auto add16 = [&](const Step& st) {
                        if (st.kind != StepKind::Conv || st.w_off < 0 || st.in.nchw || st.sh != 1 || st.sw != 1) return;
                        const bool one = st.kh == 1 && st.kw == 1 && st.out.c == 128 && st.in.c % 32 == 0;
                        const bool three = st.kh == 3 && st.kw == 3 && st.out.c == 32 && st.in.c == 128;
                        if (one || three) w_->frag16_regions.push_back({st.w_off, int(st.out.c), int(st.kh * st.kw * st.in.c)});
                    };
                    for (const Step& st : pi.plan.steps) {
                        if (st.parts.empty()) add16(st);
                        else for (const Step& q : st.parts) add16(q);
                    }
                    if (!w_->frag16_regions.empty()) {
                        check(hipMalloc(&w_->d_weights16_frag, std::max<size_t>(w_->weight_floats, 8) * 2), "hipMalloc(weights16_frag)");
                        w_->device_bytes += w_->weight_floats * 2;
                    }
                }
                if (precision_ == Precision::F8) {
                    check(hipMalloc(&w_->d_weights8, std::max<size_t>(w_->weight_floats, 16)), "hipMalloc(weights8)");
                    ZeroSync(w_->d_weights8, std::max<size_t>(w_->weight_floats, 16), "hipMemset(weights8)");
                    w_->device_bytes += w_->weight_floats;
                    int64_t aux = 0;
                    auto add8 = [&](const Step& st, int i) {
                        if (st.kind != StepKind::Conv || st.algo != ConvAlgo::IgemmF8) return;
                        w_->f8_convs.push_back({i, st.w_off, int(st.out.c), int(st.kh * st.kw * st.in.c), st.in_src, aux});
                        aux += 2 * ((st.out.c + 3) / 4 * 4);
                    };
                    for (size_t i = 0; i < pi.plan.steps.size(); ++i) {
                        const Step& st = pi.plan.steps[i];
                        if (st.parts.empty()) add8(st, int(i));
                        else for (const Step& q : st.parts) add8(q, int(i));       // a projection-shortcut step: both of its convs
                    }
                    check(hipMalloc(reinterpret_cast<void**>(&w_->d_f8_aux), size_t(std::max<int64_t>(aux, 4)) * sizeof(float)), "hipMalloc(f8 aux)");
                    w_->f8_aux_floats = size_t(aux);
                    w_->device_bytes += size_t(aux) * sizeof(float);
                    w_->act_scale.assign(pi.plan.steps.size(), 0.f);
                }
            } else if (const char* nf = env_.get("IE_NO_FRAG_WEIGHTS"); !(nf && std::atoi(nf) != 0)) {
                auto add_region = [&](const Step& st) {
                    if (st.kind == StepKind::Conv && st.w_off >= 0 && st.out.c % 16 == 0 && st.in.c % 16 == 0 && st.kh * st.kw <= 49)
                        w_->frag_regions.push_back({st.w_off, int(st.out.c), st.kh * st.kw, int(st.in.c)});
                };
                for (const Step& st : pi.plan.steps) {
                    if (st.parts.empty()) add_region(st);
                    else for (const Step& q : st.parts) add_region(q);       // a fused dense-layer step: both of its convs
                }
                if (!w_->frag_regions.empty()) {
                    check(hipMalloc(reinterpret_cast<void**>(&w_->d_weights_frag), w_->weight_floats * sizeof(float)), "hipMalloc(weights_frag)");
                    w_->device_bytes += w_->weight_floats * sizeof(float);
                }
                // Winograd-transformed weights for the 3x3 / s1 / p1 convs with 32 output channels (kernels_wino.hip)
                // (every such conv of the graph, also the ones this first plan instance runs inside a fused dense-layer step: the weights are
                // shared by all plan instances, and another batch size fuses other layers)
                int64_t utot = 0;
                auto add_wino = [&](const Step& st) {
                    if (st.kind == StepKind::Conv && st.w_off >= 0 && st.kh == 3 && st.kw == 3 && st.sh == 1 && st.sw == 1 && st.pt == 1 && st.pl == 1 &&
                        st.out.c == 32 && st.in.c % 32 == 0 && !st.in.nchw) {
                        w_->wino_regions.push_back({st.w_off, utot, 32, int(st.in.c)});
                        utot += int64_t(16) * 32 * st.in.c;
                    }
                };
                for (const Step& st : pi.plan.steps) {
                    if (st.parts.empty()) add_wino(st);
                    else for (const Step& q : st.parts) add_wino(q);
                }
                if (utot > 0) {
                    check(hipMalloc(reinterpret_cast<void**>(&w_->d_weights_wino), size_t(utot) * sizeof(float)), "hipMalloc(weights_wino)");
                    w_->wino_floats = size_t(utot);
                    w_->device_bytes += size_t(utot) * sizeof(float);
                }
                // bf16x6 mirrors (opt-in): the 1x1 convs the split kernel can run, whatever step they are part of in this plan instance
                const bool split = fp32_split_;
                w_->fp32_split = split;
                if (split) {
                    int64_t xtot = 0;
                    auto add_x6 = [&](const Step& st) {
                        if (st.kind == StepKind::Conv && st.w_off >= 0 && st.kh == 1 && st.kw == 1 && st.sh == 1 && st.sw == 1 && st.out.c % 128 == 0 && st.in.c % 32 == 0 &&
                            !st.in.nchw) {
                            w_->x6_regions.push_back({st.w_off, xtot, int(st.out.c), int(st.in.c)});
                            xtot += int64_t(3) * st.out.c * st.in.c * 2;
                        }
                    };
                    for (const Step& st : pi.plan.steps) {
                        if (st.parts.empty()) add_x6(st);
                        else for (const Step& q : st.parts) add_x6(q);
                    }
                    if (xtot > 0) {
                        check(hipMalloc(&w_->d_weights_x6, size_t(xtot)), "hipMalloc(weights_x6)");
                        w_->x6_bytes = size_t(xtot);
                        w_->device_bytes += size_t(xtot);
                    }
                    if (utot > 0) {                     // the split Winograd U: 3 planes x 2 bytes against 4 bytes per element
                        check(hipMalloc(&w_->d_weights_wino_x6, size_t(utot) * 6), "hipMalloc(weights_wino_x6)");
                        w_->device_bytes += size_t(utot) * 6;
                    }
                }
            }
            if (w_->uploaded) WeightsArrived();
        } else if (pi.plan.weights.size() != w_->weight_floats) {
            throw std::runtime_error("internal error: weight blob layout depends on the input shape");
        }
    }
    std::vector<float>().swap(pi.plan.weights);   // the host copy of the blob is only needed for the first upload
    AllocInstance(pi);
}

// Device memory of an instance whose plan is set: activation buffers (element size by buffer type) and split-K scratch.
void DeviceModel::AllocInstance(PlanInstance& pi) {
    pi.buffers.assign(pi.plan.buffer_floats.size(), nullptr);
    pi.owned.assign(pi.plan.buffer_floats.size(), 0);
    for (size_t i = 0; i < pi.plan.buffer_floats.size(); ++i) {
        float* p = nullptr;
        const int dt = pi.plan.buffer_f16[i];
        size_t bytes = size_t(std::max<int64_t>(pi.plan.buffer_floats[i], 16)) * (dt == 2 ? 1 : (dt == 1 ? 2 : 4));
        check(hipMalloc(reinterpret_cast<void**>(&p), bytes), "hipMalloc(activations)");
        check(hipMemsetAsync(p, 0, bytes, stream_), "hipMemset(activations)");
        device_bytes_ += bytes;
        pi.buffers[i] = p;
        pi.owned[i] = 1;
    }
    // split-K scratch: slabs + per-tile arrival counters (tile-padded slabs need up to 2x the exact S*M*N)
    pi.workspace_floats = std::max<int64_t>(2 * pi.plan.workspace_floats, kTuneWorkspaceFloats);
    size_t bytes = size_t(pi.workspace_floats) * sizeof(float);
    check(hipMalloc(reinterpret_cast<void**>(&pi.workspace), bytes), "hipMalloc(workspace)");
    pi.owns_workspace = true;
    device_bytes_ += bytes;
    if (!two_pass_splitk_) {
        check(hipMalloc(reinterpret_cast<void**>(&pi.counters), kNumCounters * sizeof(int)), "hipMalloc(counters)");
        check(hipMemsetAsync(pi.counters, 0, kNumCounters * sizeof(int), stream_), "hipMemset(counters)");
    }
}

// Capture steps [first, last) of `pi` into an executable graph.
void DeviceModel::Capture(PlanInstance& pi, size_t first, size_t last, hipGraphExec_t* exec) {
    hipGraph_t graph = nullptr;
    check(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
    try {
        RunSteps(pi, first, last, nullptr);
    } catch (...) {
        (void)hipStreamEndCapture(stream_, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    check(hipStreamEndCapture(stream_, &graph), "hipStreamEndCapture");
    hipError_t e = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    check(e, "hipGraphInstantiate");
}

PlanInstance& DeviceModel::Prepare(const std::vector<std::vector<int64_t>>& shapes, bool allow_tune) {
    std::vector<int64_t> key;
    for (auto& s : shapes) { key.push_back(int64_t(s.size())); key.insert(key.end(), s.begin(), s.end()); }
    auto touch = [&](const std::vector<int64_t>& k) {
        for (size_t i = 0; i < lru_.size(); ++i) if (lru_[i] == k) { lru_.erase(lru_.begin() + long(i)); break; }
        lru_.push_back(k);
    };
    check(hipSetDevice(device_), "hipSetDevice");
    auto it = plans_.find(key);
    if (it != plans_.end()) {
        touch(key);
        current_ = it->second.get();
        if (allow_tune) EnsurePipeline(*current_, true);
        return *current_;
    }
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

    auto pi = std::make_unique<PlanInstance>();
    try {
        BuildInstance(*pi, shapes);
        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        if (autotune_) Autotune(*pi, pi->plan.steps.size(), allow_tune || tune_on_demand_);
        // fp8 mode bakes the activation scales into the launches: nothing is captured before they exist (a replica that owns a
        // never-uploaded blob is built before the weight broadcast); RefreshGraphs captures on first use
        if (use_graph_ && !(precision_ == Precision::F8 && !w_->f8_ready)) {
            pi->captured_gen = w_->f8_gen.load();
            Capture(*pi, 0, pi->plan.steps.size(), &pi->graph_exec);
            pi->graph_ready = true;
        }
        if (allow_tune) EnsurePipeline(*pi, true);
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
void DeviceModel::WeightsArrived(const std::vector<float>* adopt_act_scales) {
    check(hipSetDevice(device_), "hipSetDevice");
    w_->uploaded = true;
    if (w_->d_weights_x6)
        for (const auto& xr : w_->x6_regions)
            check(LaunchSplitWeightsX6(w_->d_weights + xr.w_off, static_cast<char*>(w_->d_weights_x6) + xr.byte_off, xr.cout, xr.k, stream_), "split_weights_x6");
    if (w_->d_weights_wino) {
        check(hipSetDevice(device_), "hipSetDevice");
        for (const auto& wr : w_->wino_regions) {
            check(LaunchWinogradWeights(w_->d_weights + wr.w_off, w_->d_weights_wino + wr.u_off, wr.cout, wr.cin, stream_), "winograd_weights");
            if (w_->d_weights_wino_x6)
                check(LaunchWinogradWeightsX6(w_->d_weights + wr.w_off, static_cast<char*>(w_->d_weights_wino_x6) + wr.u_off * 6, wr.cout, wr.cin, stream_), "winograd_weights_x6");
        }
        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    }
    if (!w_->d_weights16 && !w_->d_weights_frag) return;      // (the fp16 fragment-major mirror only exists beside the half mirror)
    check(hipSetDevice(device_), "hipSetDevice");
    w_->f8_ready = false;
    if (w_->d_weights16) check(LaunchConvertF32ToF16(w_->d_weights, w_->d_weights16, int64_t(w_->weight_floats), stream_), "convert_f32_f16");
    if (w_->d_weights16_frag)
        for (const auto& fr : w_->frag16_regions)
            check(LaunchPermuteWeightsFrag16(static_cast<const char*>(w_->d_weights16) + fr.w_off * 2, static_cast<char*>(w_->d_weights16_frag) + fr.w_off * 2, fr.rows, fr.k,
                                             stream_), "permute_weights_frag16");
    if (w_->d_weights_frag)
        for (const DeviceWeights::FragRegion& fr : w_->frag_regions)
            check(LaunchPermuteWeightsFrag(w_->d_weights + fr.w_off, w_->d_weights_frag + fr.w_off, fr.cout, fr.kk, fr.cin, stream_), "permute_weights_frag");
    check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    if (precision_ == Precision::F8) PrepareF8(adopt_act_scales);
}

std::vector<std::pair<std::string, uint64_t>> DeviceModel::MirrorChecksums() {
    check(hipSetDevice(device_), "hipSetDevice");
    check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    const DeviceWeights& W = *w_;
    std::vector<std::pair<std::string, uint64_t>> out;
    std::vector<char> host;
    auto fnv = [](uint64_t h, const char* p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= uint8_t(p[i]); h *= 0x100000001B3ull; } return h; };
    auto pull = [&](const void* dev, size_t bytes) {
        host.resize(bytes);
        if (bytes) CopySync(host.data(), dev, bytes, hipMemcpyDeviceToHost, "hipMemcpy(mirror)");
    };
    constexpr uint64_t kBasis = 0xCBF29CE484222325ull;
    if (W.d_weights16) { pull(W.d_weights16, W.weight_floats * 2); out.push_back({"half", fnv(kBasis, host.data(), host.size())}); }
    if (W.d_weights16_frag) {
        uint64_t h = kBasis;
        for (const auto& fr : W.frag16_regions) { pull(static_cast<const char*>(W.d_weights16_frag) + fr.w_off * 2, size_t(fr.rows) * size_t(fr.k) * 2); h = fnv(h, host.data(), host.size()); }
        out.push_back({"half_fragment_major", h});
    }
    if (W.d_weights_frag) {        // only the conv regions of the fragment-major blob are ever written
        uint64_t h = kBasis;
        for (const auto& fr : W.frag_regions) {
            pull(W.d_weights_frag + fr.w_off, size_t(fr.cout) * size_t(fr.kk) * size_t(fr.cin) * sizeof(float));
            h = fnv(h, host.data(), host.size());
        }
        out.push_back({"fragment_major", h});
    }
    if (W.d_weights_wino) { pull(W.d_weights_wino, W.wino_floats * sizeof(float)); out.push_back({"winograd_u", fnv(kBasis, host.data(), host.size())}); }
    if (W.d_weights_x6) { pull(W.d_weights_x6, W.x6_bytes); out.push_back({"bf16x6", fnv(kBasis, host.data(), host.size())}); }
    if (W.d_weights8) {
        uint64_t h = kBasis;
        for (const auto& fc : W.f8_convs) { pull(static_cast<const char*>(W.d_weights8) + fc.w_off, size_t(fc.cout) * size_t(fc.k)); h = fnv(h, host.data(), host.size()); }
        out.push_back({"e4m3", h});
        pull(W.d_f8_aux, W.f8_aux_floats * sizeof(float));
        out.push_back({"e4m3_scales", fnv(kBasis, host.data(), host.size())});
        out.push_back({"act_scales", fnv(kBasis, reinterpret_cast<const char*>(W.act_scale.data()), W.act_scale.size() * sizeof(float))});
    }
    return out;
}

namespace {

// The counter-based generator of modelgen/rng.py, so the calibration images have the distribution of
// modelgen.models.synthetic_input: 0.6 x a coarse per-image grid of random levels (nearest-upsampled) + 0.4 x U[0,1) noise.
uint64_t rng_mix(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
uint64_t rng_key(uint64_t seed, const std::string& stream) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (unsigned char b : stream) { h ^= b; h *= 0x100000001B3ull; }
    return rng_mix(seed * 0x9E3779B97F4A7C15ull + h);
}
float rng_uniform(uint64_t key, uint64_t i) {
    return float(double(rng_mix(key + (i + 1) * 0x9E3779B97F4A7C15ull) >> 40) * (1.0 / 16777216.0));
}
std::vector<float> synthetic_images(int64_t b, int64_t c, int64_t h, int64_t w, const std::string& stream) {
    std::vector<float> x(size_t(b * c * h * w));
    const uint64_t kf = rng_key(20250704, stream), kc = rng_key(20250704, stream + "/coarse");
    if (h < 8 || w < 8) {
        for (size_t i = 0; i < x.size(); ++i) x[i] = rng_uniform(kf, i);
        return x;
    }
    const int64_t cell = std::max<int64_t>(h / 7, 1), gh = (h + cell - 1) / cell, gw = (w + cell - 1) / cell;
    for (int64_t n = 0; n < b; ++n)
        for (int64_t ch = 0; ch < c; ++ch)
            for (int64_t y = 0; y < h; ++y)
                for (int64_t xx = 0; xx < w; ++xx) {
                    const uint64_t i = uint64_t(((n * c + ch) * h + y) * w + xx);
                    const uint64_t j = uint64_t(((n * c + ch) * gh + y / cell) * gw + xx / cell);
                    x[i] = 0.6f * rng_uniform(kc, j) + 0.4f * rng_uniform(kf, i);
                }
    return x;
}

}  // namespace

// fp8 mode, once per DeviceWeights (and again after the blob was rewritten):
//   1. every fp8 conv's weight rows -> e4m3 with a per-row scale (max |w| / 448);
//   2. calibration: the SAME graph planned in fp16 mode runs eagerly on IE_F8_CALIB_BATCH (default 8) synthetic images and the
//      max |x| of every step's output is taken; a tensor's scale is margin x max / 448 (margin 2: inputs of another
//      draw saturate only beyond twice the calibrated range; e4m3 precision is relative, the headroom costs no mantissa bits);
//      max pools keep their input's scale (their output IS one of their inputs);
//   3. epilogue multipliers escale[o] = (input tensor scale) x (weight row scale).
void DeviceModel::PrepareF8(const std::vector<float>* adopt) {
    DeviceWeights& W = *w_;
    if (W.f8_ready || !W.uploaded) return;
    if (adopt && adopt->size() != W.act_scale.size()) adopt = nullptr;
    check(hipSetDevice(device_), "hipSetDevice");
    for (const auto& fc : W.f8_convs)
        check(LaunchQuantizeRowsE4m3(W.d_weights + fc.w_off, static_cast<char*>(W.d_weights8) + fc.w_off, W.d_f8_aux + fc.aux_off, fc.cout, fc.k, stream_),
              "quantize_rows_e4m3");
    // ---- calibration pass in fp16 (skipped when the scales of the weight owner that already calibrated are adopted) ----
    if (adopt) W.act_scale = *adopt;
    else {
    int64_t nc = 8;
    if (const char* e = env_.get("IE_F8_CALIB_BATCH")) nc = std::max(1, std::min(64, std::atoi(e)));
    float margin = 2.0f;
    if (const char* e = env_.get("IE_F8_MARGIN")) margin = std::max(1.0f, float(std::atof(e)));
    std::vector<std::vector<int64_t>> shapes;
    for (const auto& vi : model_->inputs) {
        std::vector<int64_t> sh = vi.dims;
        for (size_t k = 0; k < sh.size(); ++k) if (sh[k] <= 0) sh[k] = (k == 0 ? nc : 1);
        shapes.push_back(sh);
    }
    PlanInstance cal;
    cal.plan = BuildPlan(*model_, shapes, Precision::F16, true);       // the fp8 plan's step fusions, so the step lists line up
    if (cal.plan.steps.size() != W.act_scale.size()) throw std::runtime_error("fp8 calibration: the fp16 plan of the graph has a different step list");
    std::vector<float>().swap(cal.plan.weights);
    float* d_amax = nullptr;
    try {
        AllocInstance(cal);
        check(hipMalloc(reinterpret_cast<void**>(&d_amax), cal.plan.steps.size() * sizeof(float)), "hipMalloc(amax)");
        check(hipMemsetAsync(d_amax, 0, cal.plan.steps.size() * sizeof(float), stream_), "hipMemset(amax)");
        for (size_t i = 0; i < cal.plan.inputs.size(); ++i) {
            const View& v = cal.plan.inputs[i].view;
            const std::vector<float> x = synthetic_images(v.n, v.c, v.h, v.w, "calibration/" + cal.plan.inputs[i].name);
            check(hipMemcpyAsync(cal.buffers[size_t(v.buf)], x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice, stream_), "hipMemcpy(calibration input)");
            check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        }
        for (size_t i = 0; i < cal.plan.steps.size(); ++i) {
            LaunchStep(cal, cal.plan.steps[i], stream_);
            check(LaunchAbsMax(make_arg(cal, cal.plan.steps[i].out), d_amax + i, stream_), "absmax");
        }
        std::vector<float> amax(cal.plan.steps.size());
        check(hipMemcpyAsync(amax.data(), d_amax, amax.size() * sizeof(float), hipMemcpyDeviceToHost, stream_), "hipMemcpy(amax)");
        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        for (size_t i = 0; i < amax.size(); ++i) {
            const Step& st = cal.plan.steps[i];
            float sc = (amax[i] > 0.f ? amax[i] : 1.f) * margin / 448.0f;
            if (st.kind == StepKind::Pool && st.pool_max && st.in_src >= 0 && W.act_scale[size_t(st.in_src)] > 0.f) sc = W.act_scale[size_t(st.in_src)];
            W.act_scale[i] = sc;
        }
    } catch (...) {
        if (d_amax) (void)hipFree(d_amax);
        FreeInstance(cal);
        throw;
    }
    (void)hipFree(d_amax);
    FreeInstance(cal);
    }
    for (const auto& fc : W.f8_convs) {
        if (fc.in_src < 0) throw std::runtime_error("fp8 precision: a conv reads a tensor whose producer is unknown");
        const int64_t half = (fc.cout + 3) / 4 * 4;
        check(LaunchScaleVector(W.d_f8_aux + fc.aux_off, W.d_f8_aux + fc.aux_off + half, W.act_scale[size_t(fc.in_src)], fc.cout, stream_), "scale_vector");
    }
    check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    W.f8_gen.fetch_add(1);
    W.f8_ready = true;
}

// ---- pipelined host path --------------------------------------------------------------------------------------------------
// Cut the batch of `pi` into image ranges and decide how many leading steps run per range (PlanInstance::chunks / head_steps).
// The range plans come from the planner for the range's batch size (so they carry their own kernel choices); they are accepted
// only when they are the parent's plan with a smaller n: same steps, same views, same weight offsets.
void DeviceModel::EnsurePipeline(PlanInstance& pi, bool allow_tune) {
    if (pi.pipeline_tried) return;
    pi.pipeline_tried = true;
    if (pipeline_chunks_ < 2 || pi.plan.inputs.empty() || pi.plan.steps.size() < 4) return;
    const int64_t B = pi.plan.inputs[0].dims.empty() ? 0 : pi.plan.inputs[0].dims[0];
    for (auto& d : pi.plan.inputs) if (d.dims.empty() || d.dims[0] != B || d.view.n != B) return;
    for (auto& d : pi.plan.outputs) if (d.dims.empty() || d.dims[0] != B) return;
    int C = pipeline_chunks_;
    while (C > 1 && (B % C != 0 || B / C < 4)) C /= 2;
    if (C < 2) return;
    const int64_t Bc = B / C;
    std::vector<std::vector<int64_t>> sub_shapes;
    for (auto& d : pi.plan.inputs) { sub_shapes.push_back(d.dims); sub_shapes.back()[0] = Bc; }
    Plan sub;
    try {
        sub = BuildPlan(*model_, sub_shapes, precision_);
    } catch (const std::exception&) {
        return;          // e.g. the model fixes its batch dimension
    }
    const Plan& full = pi.plan;
    if (sub.steps.size() != full.steps.size() || sub.weights.size() != w_->weight_floats || 2 * sub.workspace_floats > pi.workspace_floats) return;
    auto same_view = [&](const View& a, const View& b) {
        return a.buf == b.buf && a.c == b.c && a.h == b.h && a.w == b.w && a.c_off == b.c_off && a.pitch == b.pitch && a.nchw == b.nchw && a.f16 == b.f16 &&
               a.n * C == b.n;
    };
    auto image_stride = [](const View& v) { return v.nchw ? v.c * v.h * v.w : v.h * v.w * v.pitch; };
    // ---- how many steps per range: enough modelled compute to cover the upload of the remaining ranges ----
    size_t head = 0;
    {
        double in_bytes = 0;
        for (auto& d : full.inputs) in_bytes += double(d.view.numel()) * 4.0;
        const double t_h2d = in_bytes * double(C - 1) / double(C) / 40e9;                // pageable H2D: ~40 GB/s measured
        const double peak = precision_ == Precision::F32 ? 0.45 * 157.3e12 : 0.3 * 2.5e15;
        double total = 0;
        std::vector<double> est(full.steps.size());
        for (size_t i = 0; i < full.steps.size(); ++i) { est[i] = std::max(full.steps[i].flops / peak, full.steps[i].bytes / 2.5e12) + 4e-6; total += est[i]; }
        const double want = std::min(t_h2d, 0.6 * total);
        double acc = 0;
        while (head < full.steps.size() - 1 && acc < want) acc += est[head++];
        if (pipeline_head_ >= 0) head = std::min<size_t>(size_t(pipeline_head_), full.steps.size() - 1);
    }
    // ---- validity: range plan == parent plan with n / C, and no buffer that is read by the tail holds tensors of two different
    //      image strides during the head (a later range would overwrite an earlier range's live rows) ----
    auto valid = [&](size_t h) {
        for (size_t i = 0; i < h; ++i) {
            const Step &a = sub.steps[i], &b = full.steps[i];
            if (a.kind != b.kind || !same_view(a.in, b.in) || !same_view(a.out, b.out) || a.has_in2 != b.has_in2 || (a.has_in2 && !same_view(a.in2, b.in2)) ||
                a.w_off != b.w_off || a.bias_off != b.bias_off || a.pre_scale_off != b.pre_scale_off || a.pre_shift_off != b.pre_shift_off ||
                a.parts.size() != b.parts.size())
                return false;
            for (size_t q = 0; q < a.parts.size(); ++q)
                if (!same_view(a.parts[q].in, b.parts[q].in) || !same_view(a.parts[q].out, b.parts[q].out) || a.parts[q].w_off != b.parts[q].w_off) return false;
        }
        // Tensors that cross from the head into the tail: a tail step reads a view that no earlier TAIL step has written.  Range c+1
        // runs its head after range c finished its own, so inside the head a range may overwrite rows of earlier ranges only in
        // buffers whose content is dead at the boundary.  Rows of different ranges are disjoint when every tensor the head puts into
        // a buffer has the same image stride; so: every buffer holding a crossing tensor must see ONE image stride during the head.
        std::map<int, int64_t> head_stride;       // buffer -> image stride of the head's tensors in it (-1 = mixed)
        auto note = [&](const View& v) {
            auto f = head_stride.find(v.buf);
            if (f == head_stride.end()) head_stride[v.buf] = image_stride(v);
            else if (f->second != image_stride(v)) f->second = -1;
        };
        for (size_t i = 0; i < h; ++i) {
            note(full.steps[i].in); note(full.steps[i].out);
            if (full.steps[i].has_in2) note(full.steps[i].in2);
            for (const Step& q : full.steps[i].parts) { note(q.in); note(q.out); }
        }
        std::vector<const View*> tail_written;
        auto covered = [&](const View& v) {      // the union of the tail's earlier writes (concat slices) spans the view's channels
            int64_t pos = v.c_off;
            const int64_t end = v.c_off + v.c;
            bool progress = true;
            while (pos < end && progress) {
                progress = false;
                for (const View* w : tail_written)
                    if (w->buf == v.buf && w->nchw == v.nchw && image_stride(*w) == image_stride(v) && w->c_off <= pos && w->c_off + w->c > pos) {
                        pos = w->c_off + w->c;
                        progress = true;
                    }
            }
            return pos >= end;
        };
        for (size_t i = h; i < full.steps.size(); ++i) {
            for (const View* v : {&full.steps[i].in, full.steps[i].has_in2 ? &full.steps[i].in2 : nullptr}) {
                if (!v || covered(*v)) continue;
                auto f = head_stride.find(v->buf);
                if (f != head_stride.end() && (f->second == -1 || f->second != image_stride(*v))) return false;
            }
            for (const Step& q : full.steps[i].parts) {      // a fused step also reads its 3x3's input and writes its 3x3's slice
                if (!covered(q.in)) {
                    auto f = head_stride.find(q.in.buf);
                    if (f != head_stride.end() && (f->second == -1 || f->second != image_stride(q.in))) return false;
                }
                tail_written.push_back(&q.out);
            }
            tail_written.push_back(&full.steps[i].out);
        }
        return true;
    };
    while (head > 0 && !valid(head)) --head;
    if (head == 0) return;
    std::vector<float>().swap(sub.weights);
    for (int c = 0; c < C; ++c) {
        auto ch = std::make_unique<PlanInstance>();
        ch->plan = sub;
        ch->buffers = pi.buffers;
        ch->owned.assign(pi.buffers.size(), 0);
        ch->workspace = pi.workspace;
        ch->workspace_floats = pi.workspace_floats;
        ch->counters = pi.counters;
        ch->batch_off = int64_t(c) * Bc;
        pi.chunks.push_back(std::move(ch));
    }
    try {
        if (autotune_) {
            Autotune(*pi.chunks[0], head, allow_tune || tune_on_demand_);
            for (int c = 1; c < C; ++c)
                for (size_t i = 0; i < head; ++i) {
                    Step& d = pi.chunks[size_t(c)]->plan.steps[i];
                    const Step& s0 = pi.chunks[0]->plan.steps[i];
                    d.algo = s0.algo; d.tile = s0.tile; d.splitk = s0.splitk;
                }
        }
        if (use_graph_ && !(precision_ == Precision::F8 && !w_->f8_ready) && pi.graph_ready) {
            for (auto& ch : pi.chunks) { Capture(*ch, 0, head, &ch->graph_exec); ch->graph_ready = true; }
            Capture(pi, head, pi.plan.steps.size(), &pi.tail_exec);
        }
    } catch (...) {
        for (auto& ch : pi.chunks) FreeInstance(*ch);
        pi.chunks.clear();
        if (pi.tail_exec) { (void)hipGraphExecDestroy(pi.tail_exec); pi.tail_exec = nullptr; }
        throw;
    }
    pi.head_steps = int(head);
}

// Kernel choice per conv step of pi.plan.steps[0, nsteps): an exact hit in the device's cache, else (allow_search) the exhaustive
// timed search, else the cached choice of the same conv at the nearest pixel count (within 2x), else the planner's default.
void DeviceModel::Autotune(PlanInstance& pi, size_t nsteps, bool allow_search) {
    hipEvent_t e0, e1;
    check(hipEventCreate(&e0), "hipEventCreate");
    check(hipEventCreate(&e1), "hipEventCreate");
    static const int kSplits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
    constexpr size_t kScrubBytes = size_t(64) << 20;       // > 8 x 4 MiB of L2
    void* scrub = nullptr;
    if (const char* e = env_.get("IE_TUNE_HOT"); allow_search && !(e && std::atoi(e) != 0))
        if (hipMalloc(&scrub, kScrubBytes) != hipSuccess) { scrub = nullptr; (void)hipGetLastError(); }
    bool searched = false;
    try {
        for (size_t si = 0; si < nsteps && si < pi.plan.steps.size(); ++si) {
            Step& s = pi.plan.steps[si];
            if (s.kind != StepKind::Conv || s.algo == ConvAlgo::Naive || s.algo == ConvAlgo::Stem) continue;
            if (s.algo == ConvAlgo::DualF8) {
                if (!s.in.f8) continue;                // (a calibration plan: never tuned)
                std::vector<int64_t> keyd = {s.out.n * s.out.h * s.out.w, s.out.c, s.in.c, s.parts.size() == 2 ? s.parts[0].in.c : 0, s.parts.size() == 2 ? s.parts[0].sh : 0,
                                             s.in.pitch, s.out.pitch, int64_t(ConvAlgo::DualF8), s.bias_off >= 0};
                {
                    std::lock_guard<std::mutex> g(w_->tune_mu);
                    auto hit = w_->tune_cache.find(keyd);
                    if (hit != w_->tune_cache.end()) { s.tile = hit->second.first; continue; }
                }
                if (!allow_search || !w_->f8_ready) continue;
                searched = true;
                float bestd = 1e30f;
                int best_t = 101;
                for (int t = 1; t < kNumConvWs8Tiles; ++t) {
                    Step trial = s;
                    trial.tile = 100 + t;
                    if (!ConvWs8Eligible(MakeConvArgs(pi, trial), t)) continue;
                    LaunchStep(pi, trial, stream_);
                    float ms_best = 1e30f;
                    for (int rep = 0; rep < 3; ++rep) {
                        if (scrub) check(hipMemsetAsync(scrub, 0, kScrubBytes, stream_), "hipMemsetAsync(scrub)");
                        check(hipEventRecord(e0, stream_), "hipEventRecord");
                        LaunchStep(pi, trial, stream_);
                        check(hipEventRecord(e1, stream_), "hipEventRecord");
                        check(hipEventSynchronize(e1), "hipEventSynchronize");
                        float ms = 0;
                        check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
                        ms_best = std::min(ms_best, ms);
                    }
                    if (ms_best < bestd) { bestd = ms_best; best_t = 100 + t; }
                }
                s.tile = best_t;
                std::lock_guard<std::mutex> g(w_->tune_mu);
                w_->tune_cache[keyd] = {best_t, 1};
                w_->tune_dirty = true;
                continue;
            }
            if (s.algo == ConvAlgo::StemPool) {
                // fused vs the two launches: one timed choice (tile 1 / 0), cached like the others
                std::vector<int64_t> keys = {s.in.n, s.in.h, s.in.w, s.out.c, s.out.pitch, s.out.f8 ? 2 : (s.out.f16 ? 1 : 0), int64_t(ConvAlgo::StemPool)};
                int choice = -1;
                {
                    std::lock_guard<std::mutex> g(w_->tune_mu);
                    auto hit = w_->tune_cache.find(keys);
                    if (hit != w_->tune_cache.end()) choice = hit->second.first;
                }
                const bool can = ConvStemPoolEligible(MakeConvArgs(pi, s));
                if (choice < 0 && allow_search && can && !(s.out.f8 && !w_->f8_ready)) {
                    searched = true;
                    float best[2] = {1e30f, 1e30f};
                    for (int t = 0; t < 2; ++t) {
                        Step trial = s;
                        trial.tile = t;
                        LaunchStep(pi, trial, stream_);
                        for (int rep = 0; rep < 3; ++rep) {
                            if (scrub) check(hipMemsetAsync(scrub, 0, kScrubBytes, stream_), "hipMemsetAsync(scrub)");
                            check(hipEventRecord(e0, stream_), "hipEventRecord");
                            LaunchStep(pi, trial, stream_);
                            check(hipEventRecord(e1, stream_), "hipEventRecord");
                            check(hipEventSynchronize(e1), "hipEventSynchronize");
                            float ms = 0;
                            check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
                            best[t] = std::min(best[t], ms);
                        }
                    }
                    choice = best[1] <= best[0] * 1.05f ? 1 : 0;      // within the timer's noise the single launch wins: it moves a third of the bytes
                    if (env_.get("IE_TUNE_LOG")) std::fprintf(stderr, "[tune] stem + pool %s: one launch %.1f us, two launches %.1f us\n", s.name.c_str(), best[1] * 1e3, best[0] * 1e3);
                    std::lock_guard<std::mutex> g(w_->tune_mu);
                    w_->tune_cache[keys] = {choice, 1};
                    w_->tune_dirty = true;
                }
                if (!can) choice = 0;
                if (choice >= 0) s.tile = choice;
                continue;
            }
            if (s.algo == ConvAlgo::DenseBlock) {
                // the parts get their own kernel choices (what runs when the chain kernel declines, and the yardstick); chain vs parts is
                // one more timed choice (tile 1 = one launch, 0 = the 2n plain launches)
                PlanInstance tmp;
                tmp.plan.steps = s.parts;
                tmp.buffers = pi.buffers;
                tmp.owned.assign(pi.buffers.size(), 0);
                tmp.workspace = pi.workspace;
                tmp.workspace_floats = pi.workspace_floats;
                tmp.counters = pi.counters;
                tmp.batch_off = pi.batch_off;
                for (size_t q = 0; q < tmp.plan.steps.size(); ++q) tmp.plan.steps[q].idx = int(q);
                Autotune(tmp, tmp.plan.steps.size(), allow_search);
                for (size_t q = 0; q < s.parts.size(); ++q) { s.parts[q].algo = tmp.plan.steps[q].algo; s.parts[q].tile = tmp.plan.steps[q].tile; s.parts[q].splitk = tmp.plan.steps[q].splitk; }
                std::vector<int64_t> keyb = {s.in.n * s.in.h * s.in.w, int64_t(s.parts.size()), s.in.c, s.in.h, s.in.w, s.in.pitch, s.in.c_off, int64_t(ConvAlgo::DenseBlock),
                                             s.pre_scale_off >= 0, s.bias_off >= 0};
                int choice = -1;
                {
                    std::lock_guard<std::mutex> g(w_->tune_mu);
                    auto hit = w_->tune_cache.find(keyb);
                    if (hit != w_->tune_cache.end()) choice = hit->second.first;
                }
                DenseBlockArgs b;
                const bool can = MakeBlockArgs(pi, s, &b) && DenseBlockEligible(b);
                if (choice < 0 && allow_search && can) {
                    searched = true;
                    float best[2] = {1e30f, 1e30f};
                    for (int t = 0; t < 2; ++t) {
                        Step trial = s;
                        trial.tile = t;
                        LaunchStep(pi, trial, stream_);
                        for (int rep = 0; rep < 3; ++rep) {
                            if (scrub) check(hipMemsetAsync(scrub, 0, kScrubBytes, stream_), "hipMemsetAsync(scrub)");
                            check(hipEventRecord(e0, stream_), "hipEventRecord");
                            LaunchStep(pi, trial, stream_);
                            check(hipEventRecord(e1, stream_), "hipEventRecord");
                            check(hipEventSynchronize(e1), "hipEventSynchronize");
                            float ms = 0;
                            check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
                            best[t] = std::min(best[t], ms);
                        }
                    }
                    choice = best[1] <= best[0] ? 1 : 0;
                    if (env_.get("IE_TUNE_LOG")) std::fprintf(stderr, "[tune] dense block %s: chain %.1f us, %zu launches %.1f us\n", s.name.c_str(), best[1] * 1e3, s.parts.size(), best[0] * 1e3);
                    std::lock_guard<std::mutex> g(w_->tune_mu);
                    w_->tune_cache[keyb] = {choice, 1};
                    w_->tune_dirty = true;
                }
                if (!can) choice = 0;
                if (choice >= 0) s.tile = choice;
                continue;
            }
            if (s.algo == ConvAlgo::DenseFused) {
                // the parts get their own kernel choices (they are what runs when the fused launcher declines, and the yardstick);
                // then fused vs split is one more timed choice, cached like the others (tile 1 = fused, 0 = split)
                PlanInstance tmp;
                tmp.plan.steps = s.parts;
                tmp.buffers = pi.buffers;
                tmp.owned.assign(pi.buffers.size(), 0);
                tmp.workspace = pi.workspace;
                tmp.workspace_floats = pi.workspace_floats;
                tmp.counters = pi.counters;
                tmp.batch_off = pi.batch_off;
                for (size_t q = 0; q < tmp.plan.steps.size(); ++q) tmp.plan.steps[q].idx = int(q);
                Autotune(tmp, tmp.plan.steps.size(), allow_search);
                for (size_t q = 0; q < s.parts.size(); ++q) { s.parts[q].algo = tmp.plan.steps[q].algo; s.parts[q].tile = tmp.plan.steps[q].tile; s.parts[q].splitk = tmp.plan.steps[q].splitk; }
                const Step& p3 = s.parts[0];
                std::vector<int64_t> keyf = {s.out.n * s.out.h * s.out.w, s.out.c, s.in.c, 1, 1, 1, 1, 0, 0, s.in.h, s.in.w, s.in.pitch, s.out.pitch, 0, int64_t(ConvAlgo::DenseFused),
                                             s.pre_scale_off >= 0, s.bias_off >= 0, p3.in.c, p3.in.pitch};
                int choice = -1;
                {
                    std::lock_guard<std::mutex> g(w_->tune_mu);
                    auto hit = w_->tune_cache.find(keyf);
                    if (hit != w_->tune_cache.end()) choice = hit->second.first;
                }
                if (choice < 0 && allow_search) {
                    searched = true;
                    auto time_it = [&](const Step& trial) {
                        LaunchStep(pi, trial, stream_);
                        float best_ms = 1e30f;
                        for (int rep = 0; rep < 3; ++rep) {
                            if (scrub) check(hipMemsetAsync(scrub, 0, kScrubBytes, stream_), "hipMemsetAsync(scrub)");
                            check(hipEventRecord(e0, stream_), "hipEventRecord");
                            LaunchStep(pi, trial, stream_);
                            check(hipEventRecord(e1, stream_), "hipEventRecord");
                            check(hipEventSynchronize(e1), "hipEventSynchronize");
                            float ms = 0;
                            check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
                            best_ms = std::min(best_ms, ms);
                        }
                        return best_ms;
                    };
                    float best_t = 1e30f;
                    choice = 0;
                    for (int t = 0; t <= 5; ++t) {         // 0 = the two plain launches; 1 .. 5 = fused tile variants (kernels_fused.hip)
                        Step trial = s;
                        trial.tile = t;
                        if (t > 0) {
                            ConvArgs a1 = MakeConvArgs(pi, trial);
                            FusedArgs f;
                            f.in3 = make_arg(pi, p3.in);
                            f.out3 = make_arg(pi, p3.out);
                            f.wfrag3 = w_->d_weights_frag && p3.w_off >= 0 ? w_->d_weights_frag + p3.w_off : nullptr;
                            if (!ConvDenseFusedEligible(a1, f, t)) continue;
                        }
                        const float ms = time_it(trial);
                        if (ms < best_t) { best_t = ms; choice = t; }
                    }
                    std::lock_guard<std::mutex> g(w_->tune_mu);
                    w_->tune_cache[keyf] = {choice, 1};
                    w_->tune_dirty = true;
                }
                if (choice >= 0) s.tile = choice;          // tile 0: ConvDenseFusedEligible declines, the parts run
                continue;
            }
            const Step planned = s;                    // the planner's default, kept when nothing better is known
            // the planner's default may already name a specialised kernel: the search starts from the tiled implicit GEMM either way
            if (s.algo == ConvAlgo::Ws1x1 || s.algo == ConvAlgo::Ws3x3 || s.algo == ConvAlgo::Direct || s.algo == ConvAlgo::Raster3x3 || s.algo == ConvAlgo::Wino3x3 || s.algo == ConvAlgo::X6) {
                s.algo = ConvAlgo::IgemmVec;
                s.tile = s.base_tile;
                s.splitk = 1;
            }
            const int64_t M = s.out.n * s.out.h * s.out.w, N = s.out.c;
            if (s.algo == ConvAlgo::IgemmF8) {
                // fp8 convs: one kernel family, the search is over its tile shapes
                std::vector<int64_t> key8 = {M, N, s.in.c, s.kh, s.kw, s.sh, s.sw, s.pt, s.pl, s.in.h, s.in.w, s.in.pitch, s.out.pitch, 0, int64_t(s.algo), 0,
                                             s.bias_off >= 0, 8, s.has_in2};
                {
                    std::lock_guard<std::mutex> g(w_->tune_mu);
                    auto hit = w_->tune_cache.find(key8);
                    if (hit != w_->tune_cache.end()) { s.tile = hit->second.first; continue; }
                }
                if (!allow_search || !w_->f8_ready) continue;
                searched = true;
                float best8 = 1e30f;
                int best_t = s.tile;
                for (int tc = 0; tc < kNumConvF8Tiles + kNumConvWs8Tiles + kNumConvWs38Tiles; ++tc) {
                    // candidates: the tiled implicit GEMM's tiles, then (tile >= 100) the weights-stationary 1x1 kernel's, then (>= 200) the 3x3's
                    const int t = tc < kNumConvF8Tiles ? tc : (tc < kNumConvF8Tiles + kNumConvWs8Tiles ? 100 + (tc - kNumConvF8Tiles) : 200 + (tc - kNumConvF8Tiles - kNumConvWs8Tiles));
                    if (t < 100 && kIgemmTiles[t].bn > 32 && N <= 32) continue;
                    Step trial = s;
                    trial.tile = t;
                    if (t >= 200 ? !ConvWs38Eligible(MakeConvArgs(pi, trial), t - 200) : (t >= 100 && !ConvWs8Eligible(MakeConvArgs(pi, trial), t - 100))) continue;
                    LaunchStep(pi, trial, stream_);
                    float ms_best = 1e30f;
                    for (int rep = 0; rep < 3; ++rep) {
                        if (scrub) check(hipMemsetAsync(scrub, 0, kScrubBytes, stream_), "hipMemsetAsync(scrub)");
                        check(hipEventRecord(e0, stream_), "hipEventRecord");
                        LaunchStep(pi, trial, stream_);
                        check(hipEventRecord(e1, stream_), "hipEventRecord");
                        check(hipEventSynchronize(e1), "hipEventSynchronize");
                        float ms = 0;
                        check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
                        ms_best = std::min(ms_best, ms);
                    }
                    if (ms_best < best8) { best8 = ms_best; best_t = t; }
                }
                s.tile = best_t;
                std::lock_guard<std::mutex> g(w_->tune_mu);
                w_->tune_cache[key8] = {best_t, 1};
                w_->tune_dirty = true;
                continue;
            }
            const int64_t bk = s.in.f16 ? 2 * kIgemmBK : kIgemmBK;
            const int64_t KT = s.algo == ConvAlgo::IgemmVec ? int64_t(s.kh) * s.kw * ((s.in.c + bk - 1) / bk)
                                                           : (int64_t(s.kh) * s.kw * s.in.c + kIgemmBK - 1) / kIgemmBK;
            std::vector<int64_t> key = {M, N, s.in.c, s.kh, s.kw, s.sh, s.sw, s.pt, s.pl, s.in.h, s.in.w, s.in.pitch, s.out.pitch,
                                        s.in.nchw, int64_t(s.algo), s.pre_scale_off >= 0, s.bias_off >= 0};
            if (s.in.f16 || s.out.f16) { key.push_back(s.in.f16); key.push_back(s.out.f16); }   // fp32 signatures keep 17 entries
            if (s.has_in2) key.push_back(1);              // a fused residual changes which kernels apply (18 / 20 entries)
            auto apply = [&](int enc_tile, int sp) {     // tile >= 100 encodes the raster kernel, >= 200 the weights-stationary 1x1
                if (enc_tile >= 600) { s.algo = ConvAlgo::X6; s.tile = enc_tile - 600; }
                else if (enc_tile >= 500) { s.algo = ConvAlgo::Wino3x3; s.tile = enc_tile - 500; }
                else if (enc_tile >= 400) { s.algo = ConvAlgo::Direct; s.tile = enc_tile - 400; }
                else if (enc_tile >= 300) { s.algo = ConvAlgo::Ws3x3; s.tile = enc_tile - 300; }
                else if (enc_tile >= 200) { s.algo = ConvAlgo::Ws1x1; s.tile = enc_tile - 200; }
                else if (enc_tile >= 100) { s.algo = ConvAlgo::Raster3x3; s.tile = enc_tile - 100; }
                else s.tile = enc_tile;
                s.splitk = sp;
            };
            {
                std::pair<int, int> choice{-1, 0};
                bool exact = false;
                {
                    std::lock_guard<std::mutex> g(w_->tune_mu);
                    auto hit = w_->tune_cache.find(key);
                    if (hit != w_->tune_cache.end()) { choice = hit->second; exact = true; }
                    else if (!allow_search) {
                        // nearest pixel count of the same conv (every other field of the signature equal), at most 2x away
                        double best_d = 1.0;        // |log2(M' / M)| <= 1
                        for (const auto& kv : w_->tune_cache) {
                            if (kv.first.size() != key.size() || !std::equal(kv.first.begin() + 1, kv.first.end(), key.begin() + 1)) continue;
                            const double d = std::fabs(std::log2(double(kv.first[0]) / double(M)));
                            if (d <= best_d) { best_d = d; choice = kv.second; }
                        }
                    }
                }
                if (choice.first >= 0) {
                    int sp = choice.second;
                    if (!exact && sp > 1) {     // split-K slabs must fit this instance's workspace at this pixel count
                        if (choice.first < 100) {
                            const IgemmTile& T = kIgemmTiles[choice.first];
                            const int64_t wgs = ((M + T.bm - 1) / T.bm) * ((N + T.bn - 1) / T.bn);
                            if (KT / sp < 2 || int64_t(sp) * wgs * T.bm * T.bn > pi.workspace_floats || wgs > kNumCounters || wgs * sp > 8192 || wgs >= 1024) sp = 1;
                        } else if (choice.first >= 500) {
                            sp = 1;
                        } else if (choice.first < 200) {
                            if (int64_t(sp) * (M + 256) * (N + 64) * 2 > pi.workspace_floats || (s.in.n * (s.in.h + 1) * (s.in.w + 1)) / 64 * sp > 16384) sp = 1;
                        } else sp = 1;
                    }
                    apply(choice.first, sp);
                    continue;
                }
                if (!allow_search) { s = planned; continue; }
            }
            searched = true;
            float best = 1e30f;
            int best_tile = s.tile, best_split = s.splitk;
            const bool tune_log = env_.flag("IE_TUNE_LOG");
            auto time_trial_raw = [&](const Step& trial) {
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
            auto time_trial = [&](const Step& trial) {
                const float ms = time_trial_raw(trial);
                if (tune_log)
                    std::fprintf(stderr, "[ie-tune] M=%lld N=%lld K=%lld algo=%d tile=%d splitk=%d: %.4f ms\n", static_cast<long long>(M), static_cast<long long>(N),
                                 static_cast<long long>(s.kh * s.kw * s.in.c), int(trial.algo), trial.tile, trial.splitk, ms);
                return ms;
            };
            // LDS-window kernel for 3x3/s1/p1 convs without an activation prologue
            if (!s.in.f16 && !s.out.f16 && s.algo == ConvAlgo::IgemmVec && s.kh == 3 && s.kw == 3 && s.sh == 1 && s.sw == 1 && s.pt == 1 && s.pl == 1 && s.pb == 1 &&
                s.pr == 1 && s.pre_scale_off < 0) {
                ConvArgs probe;
                probe.in = make_arg(pi, s.in);
                probe.out = make_arg(pi, s.out);
                probe.w = w_->d_weights + s.w_off;
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
            // Winograd F(2x2, 3x3): 2.25x fewer MACs for the 32-channel 3x3 convs on even-sized images
            if (!s.in.f16 && !s.out.f16 && s.algo == ConvAlgo::IgemmVec && s.kh == 3 && s.kw == 3 && s.out.c == 32 && !s.has_in2) {
                Step probe_step = s;
                probe_step.algo = ConvAlgo::Wino3x3;
                ConvArgs probe = MakeConvArgs(pi, probe_step);
                for (int t = 0; t < kNumConvWinoTiles; ++t) {
                    if (!ConvWinoEligible(probe, t)) {
                        if (tune_log)
                            std::fprintf(stderr, "[ie-tune] M=%lld wino tile %d ineligible: wfrag=%p in(c=%d h=%d w=%d sw=%lld sh=%lld sn=%lld p=%p) out(c=%d sw=%lld p=%p) bias=%p pre=%p\n",
                                         static_cast<long long>(M), t, (const void*)probe.wfrag, probe.in.c, probe.in.h, probe.in.w, (long long)probe.in.sw, (long long)probe.in.sh,
                                         (long long)probe.in.sn, (void*)probe.in.p, probe.out.c, (long long)probe.out.sw, (void*)probe.out.p, (const void*)probe.bias, (const void*)probe.pre_scale);
                        continue;
                    }
                    Step trial = s;
                    trial.algo = ConvAlgo::Wino3x3;
                    trial.tile = t;
                    trial.splitk = 1;
                    float ms = time_trial(trial);
                    if (ms < best) { best = ms; best_tile = 500 + t; best_split = 1; }
                }
            }
            // bf16x6 (opt-in): the 1x1 convs on the bf16 matrix pipe with exactly split operands
            if (w_->d_weights_x6 && !s.in.f16 && !s.out.f16 && s.algo == ConvAlgo::IgemmVec && s.kh == 1 && s.kw == 1 && !s.has_in2) {
                Step probe_step = s;
                probe_step.algo = ConvAlgo::X6;
                ConvArgs probe = MakeConvArgs(pi, probe_step);
                for (int t = 0; t < kNumConvX6Tiles; ++t) {
                    if (!ConvX6Eligible(probe, t)) continue;
                    Step trial = s;
                    trial.algo = ConvAlgo::X6;
                    trial.tile = t;
                    trial.splitk = 1;
                    float ms = time_trial(trial);
                    if (ms < best) { best = ms; best_tile = 600 + t; best_split = 1; }
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
                for (int t = 0; t < (s.in.f16 ? kNumConvWs16Tiles : kNumConvWs32Tiles); ++t) {
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
            std::lock_guard<std::mutex> g(w_->tune_mu);
            w_->tune_cache[key] = {best_tile, best_split};
            w_->tune_dirty = true;
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
    if (searched) SaveTuneCache();
}

// Whole-file rewrite through a temporary + rename, so a reader (another process loading the same model) never sees a torn file
// and concurrent writers cannot interleave lines.
void DeviceModel::SaveTuneCache() {
    std::lock_guard<std::mutex> g(w_->tune_mu);
    if (w_->tune_cache_path.empty() || !w_->tune_dirty) return;
    const std::string tmp = w_->tune_cache_path + ".tmp" + std::to_string(long(getpid())) + "." + std::to_string(device_);
    {
        std::ofstream f(tmp, std::ios::trunc);
        if (!f) return;                                   // read-only model directory: keep the choices in memory only
        f << tune_file_header() << '\n';
        for (auto& kv : w_->tune_cache) {
            for (auto v : kv.first) f << v << ' ';
            f << ": " << kv.second.first << ' ' << kv.second.second << '\n';
        }
        if (!f.good()) { f.close(); std::remove(tmp.c_str()); return; }
    }
    if (std::rename(tmp.c_str(), w_->tune_cache_path.c_str()) != 0) std::remove(tmp.c_str());
    else w_->tune_dirty = false;
}

// Kernel arguments of a dense-block chain step (kernels_block.hip) from its parts: (1x1, 3x3) pairs on one concat buffer.
bool DeviceModel::MakeBlockArgs(const PlanInstance& pi, const Step& s, DenseBlockArgs* out) const {
    if (s.parts.size() < 2 || s.parts.size() % 2 || s.parts.size() / 2 > size_t(kMaxBlockLayers) || !w_->d_weights16 || !w_->d_weights16_frag) return false;
    DenseBlockArgs b;
    const Step& f1 = s.parts[0];
    const TensorArg xin = make_arg(pi, f1.in);
    b.x = reinterpret_cast<_Float16*>(xin.p) - f1.in.c_off;           // pixel row start of the concat buffer (this plan instance's image range)
    b.pitch = int(f1.in.pitch);
    b.in_coff = int(f1.in.c_off);
    b.n = int(f1.in.n); b.h = int(f1.in.h); b.w = int(f1.in.w);
    b.wfrag16 = static_cast<const _Float16*>(w_->d_weights16_frag);
    b.w16 = static_cast<const _Float16*>(w_->d_weights16);
    b.w32 = w_->d_weights;
    b.w16_bytes = uint64_t(w_->weight_floats) * 2;
    b.nlayers = int(s.parts.size() / 2);
    auto u = [](int64_t off) { return off >= 0 ? unsigned(off) : 0xffffffffu; };
    for (int l = 0; l < b.nlayers; ++l) {
        const Step& c1 = s.parts[size_t(2 * l)];
        const Step& c3 = s.parts[size_t(2 * l + 1)];
        bool have1 = false, have3 = false;
        for (const auto& fr : w_->frag16_regions) { have1 = have1 || fr.w_off == c1.w_off; have3 = have3 || fr.w_off == c3.w_off; }
        if (!have1 || !have3 || c1.w_off >= (int64_t(1) << 31) || c3.w_off >= (int64_t(1) << 31)) return false;
        DenseBlockLayer& L = b.layer[l];
        L.K = int(c1.in.c);
        L.out_coff = int(c3.out.c_off);
        L.w1 = unsigned(c1.w_off);
        L.w3 = unsigned(c3.w_off);
        L.ps = u(c1.pre_scale_off);
        L.pt = u(c1.pre_shift_off);
        L.b1 = u(c1.bias_off);
        L.b3 = u(c3.bias_off);
        L.flags = (c1.pre_relu ? 1 : 0) | (c1.relu ? 2 : 0) | (c3.relu ? 4 : 0);
    }
    *out = b;
    return true;
}

ConvArgs DeviceModel::MakeConvArgs(const PlanInstance& pi, const Step& s) const {
    const float* wb = w_->d_weights;
    auto wp = [&](int64_t off) -> const float* { return off >= 0 ? wb + off : nullptr; };
    ConvArgs a;
    a.in = make_arg(pi, s.in);
    a.out = make_arg(pi, s.out);
    if (s.has_in2) a.res = make_arg(pi, s.in2);
    a.w = wp(s.w_off);
    a.w16 = w_->d_weights16 && s.w_off >= 0 ? static_cast<const char*>(w_->d_weights16) + s.w_off * 2 : nullptr;
    a.wfrag = w_->d_weights_frag && s.w_off >= 0 && s.out.c % 16 == 0 && s.in.c % 16 == 0 && s.kh * s.kw <= 49 ? w_->d_weights_frag + s.w_off : nullptr;
    a.bias = wp(s.bias_off);
    a.pre_scale = wp(s.pre_scale_off);
    a.pre_shift = wp(s.pre_shift_off);
    if (w_->d_weights16 && s.pre_scale_off >= 0) {
        a.pre_scale16 = static_cast<const char*>(w_->d_weights16) + s.pre_scale_off * 2;
        a.pre_shift16 = static_cast<const char*>(w_->d_weights16) + s.pre_shift_off * 2;
    }
    a.kh = s.kh; a.kw = s.kw; a.sh = s.sh; a.sw = s.sw; a.pt = s.pt; a.pl = s.pl;
    a.pre_relu = s.pre_relu; a.relu = s.relu;
    a.workspace = pi.workspace;
    a.workspace_floats = pi.workspace_floats;
    a.counters = pi.counters;
    a.num_counters = pi.counters ? kNumCounters : 0;
    if (s.algo == ConvAlgo::X6) {                    // the split kernel reads its three bf16 planes through `w16`
        a.w16 = nullptr;
        for (const auto& xr : w_->x6_regions)
            if (xr.w_off == s.w_off && w_->d_weights_x6) a.w16 = static_cast<const char*>(w_->d_weights_x6) + xr.byte_off;
    }
    if (s.algo == ConvAlgo::Wino3x3) {               // the Winograd kernel reads the transformed weights through `wfrag`
        a.wfrag = nullptr;
        a.w16 = nullptr;
        for (const auto& wr : w_->wino_regions)
            if (wr.w_off == s.w_off && w_->d_weights_wino) {
                a.wfrag = w_->d_weights_wino + wr.u_off;
                a.w16 = w_->d_weights_wino_x6 ? static_cast<const char*>(w_->d_weights_wino_x6) + wr.u_off * 6 : nullptr;
            }
    }
    if (s.out.f8 || s.in.f8) {       // fp8 mode: e4m3 weights, per-channel epilogue multipliers, tensor scales
        const DeviceWeights& W = *w_;
        auto scale_of = [&](int step) { return step >= 0 && size_t(step) < W.act_scale.size() && W.act_scale[size_t(step)] > 0.f ? W.act_scale[size_t(step)] : 1.f; };
        a.out_qscale = 1.0f / scale_of(s.idx);
        a.res_scale = scale_of(s.in2_src);
        for (const auto& fc : W.f8_convs)
            if (fc.w_off == s.w_off) {
                a.w8 = static_cast<const char*>(W.d_weights8) + fc.w_off;
                a.escale = W.d_f8_aux + fc.aux_off + (fc.cout + 3) / 4 * 4;
            }
        if (s.algo == ConvAlgo::DualF8 && s.parts.size() == 2) {       // the projection conv rides along as a second GEMM
            const Step& pj = s.parts[0];
            a.in2 = make_arg(pi, pj.in);
            a.sh2 = pj.sh; a.sw2 = pj.sw;
            a.bias_b = wp(pj.bias_off);
            for (const auto& fc : W.f8_convs)
                if (fc.w_off == pj.w_off) {
                    a.w8b = static_cast<const char*>(W.d_weights8) + fc.w_off;
                    a.escale_b = W.d_f8_aux + fc.aux_off + (fc.cout + 3) / 4 * 4;
                }
        }
    }
    return a;
}

void DeviceModel::LaunchStep(const PlanInstance& pi, const Step& s_in, hipStream_t stream_) {
    const Step& s = s_in;
    const float* wb = w_->d_weights;
    auto wp = [&](int64_t off) -> const float* { return off >= 0 ? wb + off : nullptr; };
    switch (s.kind) {
        case StepKind::Conv: {
            if (s.algo == ConvAlgo::DualF8) {
                if (s.in.f8) {          // fp8 plan: the two GEMMs of one launch; e4m3 tensors never reach another kernel
                    if (!w_->f8_ready) throw std::runtime_error("fp8 precision: scales are not calibrated yet");
                    const ConvArgs a = MakeConvArgs(pi, s);
                    const int t = s.tile >= 100 ? s.tile - 100 : 1;
                    if (a.in2.p == nullptr || !ConvWs8Eligible(a, t)) throw std::runtime_error("fp8 precision: the projection-shortcut step " + s.name + " cannot run on the dual 1x1 kernel");
                    check(LaunchConvWs1x1F8(a, t, stream_), "conv1x1_ws_f8(dual)");
                } else {                // the fp16 plan of the fp8 calibration: the same step list, the two plain convs
                    for (const Step& q : s.parts) LaunchStep(pi, q, stream_);
                }
                break;
            }
            if (s.algo == ConvAlgo::StemPool) {
                // the stem conv and the max pool behind it in one launch (out = the pooled tensor); the two plain steps when the launcher declines
                const ConvArgs a = MakeConvArgs(pi, s);
                if (s.tile != 0 && ConvStemPoolEligible(a)) check(LaunchConvStemPool(a, stream_), "conv_stem_pool");
                else for (const Step& q : s.parts) LaunchStep(pi, q, stream_);
                break;
            }
            if (s.algo == ConvAlgo::DenseBlock) {
                DenseBlockArgs b;
                if (s.tile != 0 && MakeBlockArgs(pi, s, &b) && DenseBlockEligible(b)) check(LaunchDenseBlockF16(b, stream_), "dense_block_f16");
                else for (const Step& q : s.parts) LaunchStep(pi, q, stream_);
                break;
            }
            if (s.algo == ConvAlgo::DenseFused) {
                // one launch for the 3x3 of dense layer L and the 1x1 of layer L+1; the two plain steps when the fused kernel declines
                const Step& s3 = s.parts.at(0);
                ConvArgs a1 = MakeConvArgs(pi, s);
                FusedArgs f;
                f.in3 = make_arg(pi, s3.in);
                f.out3 = make_arg(pi, s3.out);
                f.wfrag3 = w_->d_weights_frag && s3.w_off >= 0 ? w_->d_weights_frag + s3.w_off : nullptr;
                f.bias3 = wp(s3.bias_off);
                f.relu3 = s3.relu;
                if (ConvDenseFusedEligible(a1, f, s.tile)) check(LaunchConvDenseFused(a1, f, s.tile, stream_), "conv_dense_fused");
                else {
                    LaunchStep(pi, s.parts[0], stream_);
                    LaunchStep(pi, s.parts[1], stream_);
                }
                break;
            }
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
                    case ConvAlgo::Wino3x3: ok = ConvWinoEligible(plain, s_in.tile); break;
                    case ConvAlgo::X6: ok = ConvX6Eligible(plain, s_in.tile); break;
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
            if (s_in.algo == ConvAlgo::IgemmF8) {
                // e4m3 tensors: only the fp8 kernel may touch them; a declined launch is an error, never a hand-over to a kernel that
                // would read the bytes as floats
                if (!w_->f8_ready) throw std::runtime_error("fp8 precision: scales are not calibrated yet");
                int t8 = s_in.tile;         // a weights-stationary launcher that declines these operands hands the step to the tiled fp8 kernel
                if (t8 >= 200 ? !ConvWs38Eligible(a, t8 - 200) : (t8 >= 100 && !ConvWs8Eligible(a, t8 - 100))) t8 = s_in.base_tile < kNumConvF8Tiles ? s_in.base_tile : 3;
                if (t8 >= 200) check(LaunchConvWs3x3F8(a, t8 - 200, stream_), "conv3x3_ws_f8");        // weights-stationary 3x3 (kernels_ws8.hip)
                else if (t8 >= 100) check(LaunchConvWs1x1F8(a, t8 - 100, stream_), "conv1x1_ws_f8");   // weights-stationary 1x1
                else check(LaunchConvIgemmF8(a, t8, stream_), "conv_igemm_f8");
                break;
            }
            if (a.in.f8 || a.res.f8 || (a.out.f8 && s_in.algo != ConvAlgo::Stem)) throw std::runtime_error("internal error: fp8 tensor reached a non-fp8 conv kernel");
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
            else if (s.algo == ConvAlgo::Wino3x3) check(LaunchConvWino3x3(a, s.tile, stream_), "conv3x3_wino");
            else if (s.algo == ConvAlgo::X6) check(LaunchConvX6(a, s.tile, stream_), "conv1x1_x6");
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
            if (a.in.f8 || a.out.f8) {
                const DeviceWeights& W = *w_;
                auto scale_of = [&](int step) { return step >= 0 && size_t(step) < W.act_scale.size() && W.act_scale[size_t(step)] > 0.f ? W.act_scale[size_t(step)] : 1.f; };
                a.in_scale = scale_of(s.in_src);
                a.out_qscale = 1.0f / scale_of(s.idx);
                check(LaunchPoolF8(a, stream_), "pool_f8");
                break;
            }
            check(LaunchPool(a, stream_), "pool");
            break;
        }
        case StepKind::GlobalAvgPool:
            if (s.in.f8) {
                const DeviceWeights& W = *w_;
                const float sc = s.in_src >= 0 && size_t(s.in_src) < W.act_scale.size() && W.act_scale[size_t(s.in_src)] > 0.f ? W.act_scale[size_t(s.in_src)] : 1.f;
                check(LaunchGlobalAvgPoolF8(make_arg(pi, s.in), make_arg(pi, s.out), sc, stream_), "global_avg_pool_f8");
                break;
            }
            check(LaunchGlobalAvgPool(make_arg(pi, s.in), make_arg(pi, s.out), wp(s.pre_scale_off), wp(s.pre_shift_off),
                                      s.pre_relu, stream_), "global_avg_pool");
            break;
        case StepKind::Eltwise: {
            if (s.in.f8 || s.out.f8) throw std::runtime_error("internal error: fp8 tensor reached the eltwise kernel");
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
            if (s.in.f8 || s.out.f8) throw std::runtime_error("internal error: fp8 tensor reached the copy kernel");
            check(LaunchCopy(make_arg(pi, s.in), make_arg(pi, s.out), stream_), "copy");
            break;
    }
}

static std::string kernel_label(const Step& s) {
    switch (s.kind) {
        case StepKind::Conv:
            if (s.algo == ConvAlgo::Naive) return "conv_naive_kernel";
            if (s.algo == ConvAlgo::DenseBlock) return s.tile != 0 ? "dense_block_f16_kernel<" + std::to_string(s.parts.size() / 2) + " layers>" : "dense_block_parts<" + std::to_string(s.parts.size()) + " launches>";
            if (s.algo == ConvAlgo::DenseFused) return (s.tile >= 4 ? "conv_dense_fused_ws_kernel<t" : "conv_dense_fused_kernel<t") + std::to_string(s.tile) + ">";
            if (s.algo == ConvAlgo::DualF8) return s.in.f8 ? "conv1x1_ws_f8_kernel<dual,t" + std::to_string(s.tile >= 100 ? s.tile - 100 : 1) + ">" : "dual_f8_parts<2 launches>";
            if (s.algo == ConvAlgo::IgemmF8 && s.tile >= 200) return "conv3x3_ws_f8_kernel<t" + std::to_string(s.tile - 200) + ">";
            if (s.algo == ConvAlgo::IgemmF8 && s.tile >= 100) return "conv1x1_ws_f8_kernel<t" + std::to_string(s.tile - 100) + ">";
            if (s.algo == ConvAlgo::IgemmF8)
                return "conv_igemm_f8_kernel<" + std::to_string(kIgemmTiles[s.tile].bm) + "x" + std::to_string(kIgemmTiles[s.tile].bn) + ">";
            if (s.algo == ConvAlgo::Direct) {       // one launcher family, three kernels (kernels_direct.hip): report the one that runs
                const char* k = s.tile >= 10 ? "conv1x1_as_kernel<f32,t" : s.tile >= kNumDirectBaseTiles ? "conv_win_kernel<f32,t"
                                             : s.in.f16 ? "conv_direct_kernel<f16,t" : "conv_direct_kernel<f32,t";
                return std::string(k) + std::to_string(s.tile) + ">";
            }
            if (s.algo == ConvAlgo::StemPool) return s.tile != 0 ? (s.out.f8 ? "conv_stem_kernel<f16,pool,e4m3 out>" : (s.out.f16 ? "conv_stem_kernel<f16,pool>" : "conv_stem_kernel<f32,pool>")) : "stem + pool (2 launches)";
            if (s.algo == ConvAlgo::Stem) return s.out.f8 ? "conv_stem_kernel<f16,e4m3 out>" : (s.out.f16 ? "conv_stem_kernel<f16>" : "conv_stem_kernel<f32>");
            if (s.algo == ConvAlgo::Ws1x1) return std::string(s.in.f16 ? "conv1x1_ws_f16_kernel<t" : "conv1x1_ws_f32_kernel<t") + std::to_string(s.tile) + ">";
            if (s.algo == ConvAlgo::Ws3x3) return "conv3x3_ws_f16_kernel<t" + std::to_string(s.tile) + ">";
            if (s.algo == ConvAlgo::Wino3x3) return std::string(s.tile >= 8 ? "conv3x3_wino_x6_kernel<t" : "conv3x3_wino_kernel<t") + std::to_string(s.tile) + ">";
            if (s.algo == ConvAlgo::X6) return "conv1x1_x6_kernel<t" + std::to_string(s.tile) + ">";
            if (s.algo == ConvAlgo::Raster3x3)
                return "conv3x3_raster_kernel<t" + std::to_string(s.tile) + (s.splitk > 1 ? ",splitk" + std::to_string(s.splitk) : std::string()) + ">";
            return std::string(s.in.f16 ? "conv_igemm_f16_kernel<" : "conv_igemm_kernel<") + std::to_string(kIgemmTiles[s.tile].bm) + "x" +
                   std::to_string(kIgemmTiles[s.tile].bn) + (kIgemmTiles[s.tile].kg > 1 ? "x" + std::to_string(kIgemmTiles[s.tile].kg) + "kg" : std::string()) +
                   (kIgemmTiles[s.tile].deep ? ",deep" : "") +
                   (s.algo == ConvAlgo::IgemmVec ? ",vec" : ",scalar") +
                   (s.splitk > 1 ? ",splitk" + std::to_string(s.splitk) : std::string()) + ">";
        case StepKind::Pool: return s.in.f8 ? "pool_f8_kernel" : "pool_kernel";
        case StepKind::GlobalAvgPool: return s.in.f8 ? "gap_f8_kernel" : "gap_kernel";
        case StepKind::Eltwise: return "eltwise_kernel";
        case StepKind::Copy: return "copy_kernel";
    }
    return "?";
}

void DeviceModel::RunSteps(PlanInstance& pi, size_t first, size_t last, std::vector<hipEvent_t>* events) {
    size_t k = 0;
    if (events) check(hipEventRecord((*events)[k++], stream_), "hipEventRecord");
    for (size_t i = first; i < last && i < pi.plan.steps.size(); ++i) {
        LaunchStep(pi, pi.plan.steps[i], stream_);
        if (events) check(hipEventRecord((*events)[k++], stream_), "hipEventRecord");
    }
}

// Graphs hold the fp8 activation scales as by-value kernel arguments: an instance captured before the scales existed (deferred) or
// under scales that have since been recalibrated (EngineWeightsUpdated, the weight broadcast) is captured again here, on the lane
// that owns it, before its next launch.
void DeviceModel::RefreshGraphs(PlanInstance& pi) {
    if (!use_graph_) return;
    if (precision_ == Precision::F8 && !w_->f8_ready) throw std::runtime_error("fp8 precision: scales are not calibrated yet");
    const uint64_t gen = w_->f8_gen.load();
    if (pi.graph_ready && pi.captured_gen == gen) return;
    check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    auto drop = [](hipGraphExec_t& g) { if (g) (void)hipGraphExecDestroy(g); g = nullptr; };
    drop(pi.graph_exec);
    pi.graph_ready = false;
    Capture(pi, 0, pi.plan.steps.size(), &pi.graph_exec);
    pi.graph_ready = true;
    if (!pi.chunks.empty() && pi.head_steps > 0) {
        for (auto& ch : pi.chunks) {
            drop(ch->graph_exec);
            ch->graph_ready = false;
            Capture(*ch, 0, size_t(pi.head_steps), &ch->graph_exec);
            ch->graph_ready = true;
        }
        drop(pi.tail_exec);
        Capture(pi, size_t(pi.head_steps), pi.plan.steps.size(), &pi.tail_exec);
    }
    pi.captured_gen = gen;
}

void DeviceModel::Enqueue(PlanInstance& pi) {
    check(hipSetDevice(device_), "hipSetDevice");
    RefreshGraphs(pi);
    if (pi.graph_ready) check(hipGraphLaunch(pi.graph_exec, stream_), "hipGraphLaunch");
    else RunSteps(pi, 0, pi.plan.steps.size(), nullptr);
}

void DeviceModel::Synchronize() {
    check(hipSetDevice(device_), "hipSetDevice");
    check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

std::vector<StepTiming> DeviceModel::Profile(PlanInstance& pi, int iters) {
    check(hipSetDevice(device_), "hipSetDevice");
    const size_t ns = pi.plan.steps.size();
    std::vector<hipEvent_t> ev(ns + 1);
    for (auto& e : ev) check(hipEventCreate(&e), "hipEventCreate");
    std::vector<StepTiming> out(ns);
    for (size_t i = 0; i < ns; ++i) {
        out[i].name = pi.plan.steps[i].name;
        out[i].kernel = kernel_label(pi.plan.steps[i]);
        out[i].flops = pi.plan.steps[i].flops;
        out[i].bytes = pi.plan.steps[i].bytes;
    }
    try {
        RunSteps(pi, 0, ns, nullptr);   // warm
        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        // per step the MEDIAN over the passes: one disturbed pass (clock ramp, a neighbour's burst) must not colour a family's figure
        std::vector<std::vector<float>> samples(ns);
        for (int it = 0; it < iters; ++it) {
            RunSteps(pi, 0, ns, &ev);
            check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
            for (size_t i = 0; i < ns; ++i) {
                float ms = 0;
                check(hipEventElapsedTime(&ms, ev[i], ev[i + 1]), "hipEventElapsedTime");
                samples[i].push_back(ms);
            }
        }
        for (size_t i = 0; i < ns; ++i) {
            std::sort(samples[i].begin(), samples[i].end());
            const size_t n = samples[i].size();
            out[i].ms = n == 0 ? 0.0 : (n % 2 ? samples[i][n / 2] : 0.5 * (samples[i][n / 2 - 1] + samples[i][n / 2]));
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

// The ModelInfer data path: caller memory -> HBM, forward, HBM -> caller memory.  This replaces the five host copies each way of
// the reference (inference_bridge.cpp:738-749, 806-812 and model.cpp:1229-1238, 1290-1311) with ONE DMA per direction straight
// from / to the caller's buffers (hipMemcpyAsync takes pageable memory at ~40-55 GB/s on this platform, measured by
// scripts/probes/h2d_probe.cpp; a caller that hands over pinned memory gets a fully asynchronous copy), and overlaps the upload
// with compute: the batch is cut into image ranges, range c's upload is followed by the head steps of range c on the compute
// stream while range c+1 uploads on the copy stream; the remaining steps run once on the whole batch.
void DeviceModel::InferHostSegments(PlanInstance& pi, const std::vector<std::vector<InSeg>>& in, const std::vector<std::vector<OutSeg>>& out) {
    check(hipSetDevice(device_), "hipSetDevice");
    EnsurePipeline(pi, false);
    RefreshGraphs(pi);
    const int C = pi.chunks.empty() ? 1 : int(pi.chunks.size());
    struct Job { int chunk; char* dst; const char* src; size_t copy, zero; };      // copy `copy` bytes src -> dst, then zero `zero` bytes behind them
    std::vector<Job> jobs;
    struct U8 { size_t input; };
    std::vector<char> is_u8(pi.plan.inputs.size(), 0);
    try {
        for (size_t i = 0; i < pi.plan.inputs.size() && i < in.size(); ++i) {
            const View& v = pi.plan.inputs[i].view;
            char* base = reinterpret_cast<char*>(pi.buffers[size_t(v.buf)]);
            const bool u8 = !in[i].empty() && in[i][0].u8;
            is_u8[i] = u8 ? 1 : 0;
            if (u8) {       // bytes go to a device staging buffer; a kernel per range converts them into the fp32 input buffer
                if (pi.u8_stage.size() < pi.plan.inputs.size()) pi.u8_stage.resize(pi.plan.inputs.size(), nullptr);
                if (!pi.u8_stage[i]) {
                    check(hipMalloc(&pi.u8_stage[i], std::max<size_t>(size_t(v.numel()), 16)), "hipMalloc(u8 staging)");
                    device_bytes_ += size_t(v.numel());
                }
                base = static_cast<char*>(pi.u8_stage[i]);
            }
            const size_t total = size_t(v.numel()) * (u8 ? 1 : sizeof(float));
            const size_t per_chunk = total / size_t(C);          // C divides the batch, so ranges are whole images
            for (const InSeg& sg : in[i]) {
                if (sg.u8 != u8) throw std::runtime_error("internal error: mixed UINT8 / FLOAT32 segments for one input");
                if (sg.dev_off + sg.need > total) throw std::runtime_error("internal error: input segment exceeds the planned tensor");
                const size_t have = sg.host ? std::min(sg.have, sg.need) : 0;
                size_t off = 0;
                while (off < sg.need) {                          // split at range boundaries
                    const size_t dev = sg.dev_off + off;
                    const int c = int(std::min<size_t>(dev / per_chunk, size_t(C - 1)));
                    const size_t end = std::min(sg.need, (size_t(c) + 1) * per_chunk - sg.dev_off);
                    const size_t nb = end - off;
                    const size_t cp = off < have ? std::min(nb, have - off) : 0;
                    jobs.push_back({c, base + dev, cp ? static_cast<const char*>(sg.host) + off : nullptr, cp, nb - cp});
                    off = end;
                }
            }
        }
        // ---- upload + head, range by range ----
        for (int c = 0; c < C; ++c) {
            for (const Job& j : jobs) {
                if (j.chunk != c) continue;
                if (j.copy) check(hipMemcpyAsync(j.dst, j.src, j.copy, hipMemcpyHostToDevice, copy_stream_), "hipMemcpyAsync(H2D)");
                if (j.zero) check(hipMemsetAsync(j.dst + j.copy, 0, j.zero, copy_stream_), "hipMemsetAsync");
            }
            for (size_t i = 0; i < pi.plan.inputs.size(); ++i) {
                if (!is_u8[i]) continue;
                const View& v = pi.plan.inputs[i].view;
                const int64_t n = v.numel() / C;
                check(LaunchConvertU8ToF32(static_cast<char*>(pi.u8_stage[i]) + int64_t(c) * n, pi.buffers[size_t(v.buf)] + int64_t(c) * n, n, u8_scale_, u8_bias_,
                                           copy_stream_), "convert_u8_f32");
            }
            check(hipEventRecord(h2d_events_[size_t(c)], copy_stream_), "hipEventRecord");
            check(hipStreamWaitEvent(stream_, h2d_events_[size_t(c)], 0), "hipStreamWaitEvent");
            if (c == 0) check(hipEventRecord(t0_event_, stream_), "hipEventRecord");
            if (C > 1) {
                PlanInstance& ch = *pi.chunks[size_t(c)];
                if (ch.graph_ready) check(hipGraphLaunch(ch.graph_exec, stream_), "hipGraphLaunch(head)");
                else RunSteps(ch, 0, size_t(pi.head_steps), nullptr);
            }
        }
        if (C > 1) {
            if (pi.tail_exec) check(hipGraphLaunch(pi.tail_exec, stream_), "hipGraphLaunch(tail)");
            else RunSteps(pi, size_t(pi.head_steps), pi.plan.steps.size(), nullptr);
            ++pipelined_calls_;
        } else {
            Enqueue(pi);
        }
        check(hipEventRecord(t1_event_, stream_), "hipEventRecord");
        last_chunks_ = C;
        last_head_steps_ = C > 1 ? pi.head_steps : 0;
        // ---- D2H: results are small (logits): one transfer per output into pinned staging, ONE synchronisation, then the
        //      per-caller scatter on the host; an output too large for the staging goes straight to the caller's memory ----
        struct Scatter { size_t j, lo, stage_off; };
        std::vector<Scatter> staged;
        size_t stage_used = 0;
        for (size_t j = 0; j < pi.plan.outputs.size() && j < out.size(); ++j) {
            if (out[j].empty()) continue;
            const View& v = pi.plan.outputs[j].view;
            const char* src = reinterpret_cast<const char*>(pi.buffers[size_t(v.buf)]);
            const size_t total = size_t(v.numel()) * sizeof(float);
            size_t lo = SIZE_MAX, hi = 0;
            for (const OutSeg& sg : out[j]) {
                const size_t nb = std::min(sg.cap, sg.need);
                if (!nb) continue;
                if (sg.dev_off + nb > total) throw std::runtime_error("internal error: output segment exceeds the planned tensor");
                lo = std::min(lo, sg.dev_off);
                hi = std::max(hi, sg.dev_off + nb);
            }
            if (hi <= lo) continue;
            if (stage_used + (hi - lo) <= pinned_bytes_) {
                check(hipMemcpyAsync(static_cast<char*>(pinned_) + stage_used, src + lo, hi - lo, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H)");
                staged.push_back({j, lo, stage_used});
                stage_used += (hi - lo + 63) & ~size_t(63);
            } else {
                for (const OutSeg& sg : out[j]) {
                    const size_t nb = std::min(sg.cap, sg.need);
                    if (nb) check(hipMemcpyAsync(sg.host, src + sg.dev_off, nb, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H)");
                }
            }
        }
        check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
        {
            float ms = 0;
            last_forward_ms_ = hipEventElapsedTime(&ms, t0_event_, t1_event_) == hipSuccess ? double(ms) : 0.0;
        }
        for (const Scatter& sc : staged)
            for (const OutSeg& sg : out[sc.j]) {
                const size_t nb = std::min(sg.cap, sg.need);
                if (nb) std::memcpy(sg.host, static_cast<char*>(pinned_) + sc.stage_off + (sg.dev_off - sc.lo), nb);
            }
        for (size_t j = 0; j < out.size(); ++j)
            for (const OutSeg& sg : out[j]) {
                const size_t nb = std::min(sg.cap, sg.need);
                if (sg.cap > nb) std::memset(static_cast<char*>(sg.host) + nb, 0, sg.cap - nb);
            }
    } catch (...) {
        (void)hipStreamSynchronize(copy_stream_);
        (void)hipStreamSynchronize(stream_);
        throw;
    }
}

}  // namespace ie
