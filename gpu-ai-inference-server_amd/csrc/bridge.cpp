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
#include <regex>
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
        const char* e = std::getenv("IE_ROCTX");
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

struct ModelObj {
    std::string path;
    ModelType type = MODEL_UNKNOWN;
    DeviceType device = DEVICE_GPU;
    int device_id = 0;
    std::string name, version;
    std::vector<std::string> input_names, output_names;   // config names until Load replaces them with the graph's

    std::mutex mu;                 // serialises Load / Unload / Infer on this model
    std::atomic<bool> loaded{false};
    std::string last_error;
    std::shared_ptr<const ie::OnnxModel> onnx;
    ie::ModelInfo info;
    std::unique_ptr<ie::DeviceModel> dev;
    // In-process batch sharding (single process, all GPUs of the node): extra replicas of the model on other devices.
    // A request's rows are cut into contiguous slices, one per replica, run concurrently (one host thread per replica), and the
    // result rows land at their offsets of the caller's output buffers.  IE_GPUS=<n> / config.json "gpus": n (devices
    // device_id .. device_id+n-1) or IE_SHARD_DEVICES=<id,id,...> (explicit list; ids may repeat, which is how the single-GPU
    // tests exercise the sharding logic).  Every replica uploads the packed weights itself (32 MB over PCIe, once, at load).
    std::vector<std::unique_ptr<ie::DeviceModel>> replicas;     // replicas[k] serves slice k+1 (slice 0 is `dev`)
    float u8_scale = 1.0f / 255.0f, u8_bias = 0.0f;
    int64_t load_time_ns = 0;
    std::atomic<int64_t> inference_count{0}, total_ns{0}, last_ns{0};
    std::atomic<size_t> memory_usage_bytes{0};

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
    void Execute(std::vector<Pending*>& batch);   // runs one device batch for these callers (takes `mu`)
    void RunBatched(Pending& req);                // leader/follower coalescing
};

bool ModelObj::Load() {
    std::lock_guard<std::mutex> g(mu);
    auto t0 = std::chrono::steady_clock::now();
    std::error_code ec;
    if (!std::filesystem::exists(path, ec)) {
        last_error = "Model file not found: " + path;
        return false;
    }
    bool ok = false;
    switch (type) {
        case MODEL_TENSORFLOW: last_error = "TensorFlow model loading not implemented"; break;
        case MODEL_TENSORRT: last_error = "TensorRT model loading not implemented"; break;
        case MODEL_PYTORCH: last_error = "PyTorch model loading not implemented"; break;
        case MODEL_CUSTOM: last_error = "Custom model loading not implemented"; break;
        case MODEL_ONNX: {
            try {
                const std::string file = path + "/model.onnx";
                if (!std::filesystem::exists(file, ec)) {
                    last_error = "ONNX model file not found: " + file;
                    break;
                }
                if (device != DEVICE_GPU) {
                    last_error = "DEVICE_CPU execution is not provided by the MI355X engine (a HIP device is required)";
                    break;
                }
                auto parsed = std::make_shared<ie::OnnxModel>(ie::LoadOnnxFile(file));
                ie::ModelInfo inf = ie::DescribeModel(*parsed);
                // Precision: IE_PRECISION=fp16|fp32, else config.json {"precision": "fp16"}; default fp32 (the reference's
                // ONNX Runtime session computes in the model's own fp32).
                ie::Precision prec = ie::Precision::F32;
                {
                    std::string want;
                    if (const char* e = std::getenv("IE_PRECISION")) want = e;
                    else {
                        std::ifstream cf(path + "/config.json");
                        if (cf) {
                            std::stringstream ss; ss << cf.rdbuf();
                            const std::string txt = ss.str();
                            std::smatch mm;
                            if (std::regex_search(txt, mm, std::regex("\"precision\"\\s*:\\s*\"([A-Za-z0-9]+)\""))) want = mm[1];
                        }
                    }
                    for (auto& ch : want) ch = char(std::tolower(static_cast<unsigned char>(ch)));
                    if (want == "fp16" || want == "f16" || want == "half" || want == "float16") prec = ie::Precision::F16;
                    else if (!want.empty() && want != "fp32" && want != "f32" && want != "float32" && want != "float") {
                        last_error = "ONNX model loading error: unsupported precision '" + want + "' (fp32 or fp16)";
                        break;
                    }
                }
                auto dm = std::make_unique<ie::DeviceModel>(parsed, device_id, prec);
                {   // UINT8 ingest transform x * scale + bias: config.json "uint8_scale" / "uint8_bias" (default 1/255, 0: the reference
                    // client's /255 convention, client/test_client.py:189)
                    std::ifstream cf(path + "/config.json");
                    if (cf) {
                        std::stringstream ss; ss << cf.rdbuf();
                        const std::string txt = ss.str();
                        std::smatch mm;
                        float sc = 1.0f / 255.0f, bi = 0.0f;
                        const std::string num = "(-?[0-9]*\\.?[0-9]+(?:[eE][-+]?[0-9]+)?)";
                        if (std::regex_search(txt, mm, std::regex("\"uint8_scale\"\\s*:\\s*" + num))) sc = std::stof(mm[1]);
                        if (std::regex_search(txt, mm, std::regex("\"uint8_bias\"\\s*:\\s*" + num))) bi = std::stof(mm[1]);
                        dm->SetU8Transform(sc, bi);
                        u8_scale = sc; u8_bias = bi;
                    }
                }
                // Plan once at load (symbolic dims -> 1): rejects unsupported graphs here, like Ort::Session's
                // constructor does, and puts the packed weights into HBM.
                std::vector<std::vector<int64_t>> shapes;
                for (auto& vi : inf.inputs) {
                    std::vector<int64_t> s = vi.dims;
                    for (auto& d : s) if (d <= 0) d = 1;
                    shapes.push_back(s);
                }
                dm->Prepare(shapes);
                std::vector<std::unique_ptr<ie::DeviceModel>> reps;
                {
                    std::vector<int> ids;
                    if (const char* e = std::getenv("IE_SHARD_DEVICES")) {
                        std::stringstream ss(e);
                        std::string tok;
                        while (std::getline(ss, tok, ',')) if (!tok.empty()) ids.push_back(std::atoi(tok.c_str()));
                        if (!ids.empty()) ids.erase(ids.begin());           // the first id is the primary device's slice
                    } else {
                        int n = 1;
                        if (const char* e = std::getenv("IE_GPUS")) n = std::atoi(e);
                        else {
                            std::ifstream cf(path + "/config.json");
                            if (cf) {
                                std::stringstream ss; ss << cf.rdbuf();
                                const std::string txt = ss.str();
                                std::smatch mm;
                                if (std::regex_search(txt, mm, std::regex("\"gpus\"\\s*:\\s*(\\d+)"))) n = std::stoi(mm[1]);
                            }
                        }
                        const int have = ie::HipDeviceCount();
                        for (int k = 1; k < n && device_id + k < have; ++k) ids.push_back(device_id + k);
                    }
                    for (int id : ids) {
                        auto r = std::make_unique<ie::DeviceModel>(parsed, id, prec);
                        r->SetU8Transform(u8_scale, u8_bias);
                        r->Prepare(shapes);
                        reps.push_back(std::move(r));
                    }
                }
                input_names.clear();
                output_names.clear();
                for (auto& vi : inf.inputs) input_names.push_back(vi.name);
                for (auto& vi : inf.outputs) output_names.push_back(vi.name);
                memory_usage_bytes = inf.memory_usage_bytes;   // reference's estimate formula, model.cpp:979-1035
                batchable = !inf.inputs.empty();
                for (auto& vi : inf.inputs) if (vi.dims.empty() || vi.dims[0] > 0) batchable = false;
                for (auto& vi : inf.outputs) if (vi.dims.empty() || vi.dims[0] > 0) batchable = false;
                {   // batching knobs: environment first, then the two config.json keys the reference carries but never reads
                    max_batch = cfg_max_batch;
                    if (const char* e = std::getenv("IE_DYNAMIC_BATCH")) max_batch = std::atoi(e);
                    else {
                        std::ifstream cf(path + "/config.json");
                        if (cf) {
                            std::stringstream ss; ss << cf.rdbuf();
                            const std::string txt = ss.str();
                            std::smatch mm;
                            const bool dyn = std::regex_search(txt, std::regex("\"dynamic_batching\"\\s*:\\s*true"));
                            if (dyn && std::regex_search(txt, mm, std::regex("\"max_batch_size\"\\s*:\\s*(\\d+)"))) max_batch = std::stoi(mm[1]);
                        }
                    }
                    if (const char* e = std::getenv("IE_BATCH_WINDOW_US")) batch_window_us = std::max(0, std::atoi(e));
                    if (max_batch > 4096) max_batch = 4096;
                }
                onnx = parsed;
                info = std::move(inf);
                dev = std::move(dm);
                replicas = std::move(reps);
                ok = true;
            } catch (const std::exception& e) {
                last_error = std::string("ONNX model loading error: ") + e.what();
            }
            break;
        }
        default: last_error = "Unsupported model type"; return false;
    }
    load_time_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    loaded = ok;
    return ok;
}

void ModelObj::Unload() {
    std::lock_guard<std::mutex> g(mu);
    replicas.clear();
    dev.reset();
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
            if (const char* d = std::getenv("IE_DEVICE_ID")) obj->device_id = std::atoi(d);
            obj->name = name;
            obj->version = ver.empty() ? handle->repo->LatestVersion(name) : ver;
            obj->input_names = {"input"};
            obj->output_names = {"output"};
            handle->models[name] = obj;    // reserve the name; concurrent loaders of the same name now fail fast
        }
        if (!obj->Load()) {
            std::string msg;
            { std::lock_guard<std::mutex> g(obj->mu); msg = obj->last_error; }
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
        if (!ok) { std::lock_guard<std::mutex> g(handle->model->mu); set_error(error, handle->model->last_error); }
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
            std::lock_guard<std::mutex> g(M.mu);
            if (!M.loaded.load() || !M.dev) { set_error(error, "Model not loaded"); return false; }
            auto failv = [&](const std::string& msg) { M.last_error = msg; set_error(error, msg); return false; };
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
            { std::lock_guard<std::mutex> g(M.mu); M.last_error = req.err; }
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

void ModelObj::Execute(std::vector<Pending*>& batch) {
    std::lock_guard<std::mutex> g(mu);
    auto fail_all = [&](const std::string& msg) { for (auto* r : batch) { r->ok = false; r->err = msg; } };
    if (!loaded.load() || !dev) { fail_all("Model not loaded"); return; }
    try {
        if (batch.size() == 1 && !(batchable && max_batch > 1 && batch[0]->rows > 0) && !replicas.empty() && batchable) {
            // ---- one request, rows sharded over the replicas (contiguous slices, like sharding.shard_batch) ----
            Pending& r = *batch[0];
            const int64_t rows = r.shapes.empty() || r.shapes[0].empty() ? 0 : r.shapes[0][0];
            bool same_rows = rows > 0;
            for (auto& sh : r.shapes) if (sh.empty() || sh[0] != rows) same_rows = false;
            const int S = int(replicas.size()) + 1;
            if (same_rows && rows >= S) {
                std::vector<std::string> errs;
                errs.resize(size_t(S));
                std::vector<ie::PlanInstance*> pis;
                pis.resize(size_t(S), nullptr);
                auto run_slice = [&](int k) {
                    try {
                        ie::DeviceModel& D = k == 0 ? *dev : *replicas[size_t(k - 1)];
                        const int64_t r0 = rows * k / S, r1 = rows * (k + 1) / S, nr = r1 - r0;
                        std::vector<std::vector<int64_t>> shapes = r.shapes;
                        for (auto& sh : shapes) sh[0] = nr;
                        ie::PlanInstance& pi = D.Prepare(shapes);
                        pis[size_t(k)] = &pi;
                        std::vector<const void*> in_ptr(r.in_ptr.size(), nullptr);
                        std::vector<size_t> in_bytes(r.in_ptr.size(), 0);
                        for (size_t i = 0; i < r.in_ptr.size(); ++i) {
                            const size_t row_bytes = size_t(pi.plan.inputs[i].view.numel() / nr) * (r.in_u8[i] ? 1 : sizeof(float));
                            const size_t off = size_t(r0) * row_bytes;
                            if (r.in_ptr[i] && r.in_bytes[i] > off) {
                                in_ptr[i] = static_cast<const char*>(r.in_ptr[i]) + off;
                                in_bytes[i] = std::min(r.in_bytes[i] - off, size_t(nr) * row_bytes);
                            }
                        }
                        std::vector<void*> out_ptr;
                        std::vector<size_t> out_bytes;
                        for (int j = 0; j < r.num_outputs; ++j) {
                            void* p = nullptr;
                            size_t nb = 0;
                            const TensorData& o = r.outputs[j];
                            if (size_t(j) < pi.plan.outputs.size() && o.data_type == DATATYPE_FLOAT32 && o.data && o.data_size > 0) {
                                const size_t row_bytes = size_t(pi.plan.outputs[size_t(j)].view.numel() / nr) * sizeof(float);
                                const size_t off = size_t(r0) * row_bytes;
                                if (o.data_size > off) {
                                    p = static_cast<char*>(o.data) + off;
                                    // the last slice also owns (zero-fills) whatever the caller's buffer has beyond the result
                                    nb = k == S - 1 ? o.data_size - off : std::min(o.data_size - off, size_t(nr) * row_bytes);
                                }
                            }
                            out_ptr.push_back(p);
                            out_bytes.push_back(nb);
                        }
                        D.InferHost(pi, in_ptr, in_bytes, out_ptr, out_bytes, r.in_u8);
                    } catch (const std::exception& e) {
                        errs[size_t(k)] = e.what();
                    }
                };
                std::vector<std::thread> workers;
                for (int k = 1; k < S; ++k) workers.emplace_back(run_slice, k);
                run_slice(0);
                for (auto& w : workers) w.join();
                for (auto& e : errs) if (!e.empty()) { fail_all("ONNX inference error: " + e); return; }
                write_out_dims(r.outputs, r.num_outputs, pis[0]->plan.outputs, rows);
                shard_calls.fetch_add(1);
                r.ok = true;
                return;
            }
        }
        if (batch.size() == 1 && !(batchable && max_batch > 1 && batch[0]->rows > 0)) {
            Pending& r = *batch[0];
            ie::PlanInstance& pi = dev->Prepare(r.shapes);
            std::vector<void*> out_ptr;
            std::vector<size_t> out_bytes;
            for (int i = 0; i < r.num_outputs; ++i) {
                const bool copy = r.outputs[i].data_type == DATATYPE_FLOAT32 && r.outputs[i].data && r.outputs[i].data_size > 0;
                out_ptr.push_back(copy ? r.outputs[i].data : nullptr);
                out_bytes.push_back(copy ? r.outputs[i].data_size : 0);
            }
            dev->InferHost(pi, r.in_ptr, r.in_bytes, out_ptr, out_bytes, r.in_u8);
            write_out_dims(r.outputs, r.num_outputs, pi.plan.outputs, 0);
            r.ok = true;
            return;
        }
        // ---- coalesced batch: rows of all callers back to back, padded up to a power-of-two bucket so only a handful of
        //      plans / hipGraphs ever exist ----
        int64_t total = 0;
        for (auto* r : batch) total += r->rows;
        int64_t bucket = 1;
        while (bucket < total) bucket <<= 1;
        if (bucket > max_batch && total <= max_batch) bucket = max_batch;
        std::vector<std::vector<int64_t>> shapes = batch[0]->shapes;
        for (auto& sh : shapes) sh[0] = bucket;
        ie::PlanInstance& pi = dev->Prepare(shapes);
        std::vector<std::vector<ie::DeviceModel::InSeg>> in(pi.plan.inputs.size());
        std::vector<std::vector<ie::DeviceModel::OutSeg>> out(pi.plan.outputs.size());
        int64_t row0 = 0;
        for (auto* r : batch) {
            for (size_t k = 0; k < pi.plan.inputs.size(); ++k) {
                const size_t row_bytes = size_t(pi.plan.inputs[k].view.numel() / bucket) * sizeof(float);
                in[k].push_back({r->in_ptr[k], r->in_bytes[k], size_t(r->rows) * row_bytes, size_t(row0) * row_bytes});
            }
            for (int j = 0; j < r->num_outputs && size_t(j) < pi.plan.outputs.size(); ++j) {
                const TensorData& o = r->outputs[j];
                if (o.data_type != DATATYPE_FLOAT32 || !o.data || o.data_size == 0) continue;
                const size_t row_bytes = size_t(pi.plan.outputs[size_t(j)].view.numel() / bucket) * sizeof(float);
                out[size_t(j)].push_back({o.data, o.data_size, size_t(r->rows) * row_bytes, size_t(row0) * row_bytes});
            }
            row0 += r->rows;
        }
        dev->InferHostSegments(pi, in, out);
        device_batches.fetch_add(1);
        coalesced_requests.fetch_add(int64_t(batch.size()));
        for (auto* r : batch) {
            write_out_dims(r->outputs, r->num_outputs, pi.plan.outputs, r->rows);
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

extern "C" {

ModelMetadata* ModelGetMetadata(ModelHandle handle) {
    if (!handle) return nullptr;
    try {
        ModelObj& M = *handle->model;
        std::lock_guard<std::mutex> g(M.mu);
        auto* md = static_cast<ModelMetadata*>(std::calloc(1, sizeof(ModelMetadata)));
        md->name = dup_cstr(M.name);
        md->version = dup_cstr(M.version);
        md->model_type = M.type;
        md->description = dup_cstr("");
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
        if (batch > 0) {
            std::vector<std::vector<int64_t>> shapes;
            for (auto& vi : info.inputs) {
                std::vector<int64_t> s = vi.dims;
                for (size_t k = 0; k < s.size(); ++k) if (s[k] <= 0) s[k] = (k == 0 ? batch : 1);
                shapes.push_back(s);
            }
            ie::Precision prec = ie::Precision::F32;       // the planner is host code: IE_PRECISION selects what to describe
            if (const char* e = std::getenv("IE_PRECISION")) {
                std::string w = e;
                if (w == "fp16" || w == "f16" || w == "half") prec = ie::Precision::F16;
            }
            o << ",\"plan\":" << ie::PlanToJson(ie::BuildPlan(m, shapes, prec));
        }
        o << "}";
        return dup_cstr(o.str());
    } catch (const std::exception& e) { set_error(error, e.what()); return nullptr; }
    catch (...) { set_error(error, "unknown error"); return nullptr; }
}

bool EnginePrepare(ModelHandle handle, const Shape* input_shapes, int num_inputs, void** d_inputs, void** d_outputs,
                   int num_outputs, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return false; }
    try {
        ModelObj& M = *handle->model;
        std::lock_guard<std::mutex> g(M.mu);
        if (!M.loaded.load() || !M.dev) { set_error(error, "Model not loaded"); return false; }
        if (!input_shapes || num_inputs <= 0) { set_error(error, "Invalid parameters"); return false; }
        std::vector<std::vector<int64_t>> shapes;
        for (int i = 0; i < num_inputs; ++i) {
            if (!input_shapes[i].dims || input_shapes[i].num_dims <= 0) { set_error(error, "Invalid parameters"); return false; }
            shapes.emplace_back(input_shapes[i].dims, input_shapes[i].dims + input_shapes[i].num_dims);
        }
        ie::PlanInstance& pi = M.dev->Prepare(shapes);
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
        std::lock_guard<std::mutex> g(M.mu);
        if (!M.loaded.load() || !M.dev || !M.dev->current()) { set_error(error, "Model not prepared"); return false; }
        for (int i = 0; i < iters; ++i) M.dev->Enqueue(*M.dev->current());
        if (sync) M.dev->Synchronize();
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

bool EngineSynchronize(ModelHandle handle, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return false; }
    try {
        ModelObj& M = *handle->model;
        std::lock_guard<std::mutex> g(M.mu);
        if (!M.dev) { set_error(error, "Model not loaded"); return false; }
        M.dev->Synchronize();
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

void* EngineGetStream(ModelHandle handle) {
    if (!handle) return nullptr;
    try {
        ModelObj& M = *handle->model;
        std::lock_guard<std::mutex> g(M.mu);
        return M.dev ? static_cast<void*>(M.dev->stream()) : nullptr;
    } catch (...) { return nullptr; }
}

char* EngineProfile(ModelHandle handle, int iters, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid model handle"); return nullptr; }
    try {
        ModelObj& M = *handle->model;
        std::lock_guard<std::mutex> g(M.mu);
        if (!M.loaded.load() || !M.dev || !M.dev->current()) { set_error(error, "Model not prepared"); return nullptr; }
        auto t = M.dev->Profile(*M.dev->current(), iters > 0 ? iters : 1);
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
        std::lock_guard<std::mutex> g(M.mu);
        if (!M.loaded.load() || !M.dev) { set_error(error, "Model not loaded"); return false; }
        *d_ptr = M.dev->weights();
        *bytes = M.dev->weight_bytes();
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

bool EngineWeightsUpdated(ModelHandle handle, ErrorMessage* error) {
    if (!handle) { set_error(error, "Invalid parameters"); return false; }
    try {
        ModelObj& M = *handle->model;
        std::lock_guard<std::mutex> g(M.mu);
        if (!M.loaded.load() || !M.dev) { set_error(error, "Model not loaded"); return false; }
        M.dev->Synchronize();
        M.dev->RefreshHalfWeights();
        return true;
    } catch (const std::exception& e) { set_error(error, e.what()); return false; }
    catch (...) { set_error(error, "unknown error"); return false; }
}

int EngineGetPrecision(ModelHandle handle) {
    if (!handle) return -1;
    ModelObj& M = *handle->model;
    std::lock_guard<std::mutex> g(M.mu);
    if (!M.loaded.load() || !M.dev) return -1;
    return int(M.dev->precision());
}

bool EngineMemcpy(ModelHandle handle, void* dst, const void* src, size_t bytes, int kind, ErrorMessage* error) {
    if (!handle || !dst || !src || kind < 1 || kind > 3) { set_error(error, "Invalid parameters"); return false; }
    try {
        ModelObj& M = *handle->model;
        std::lock_guard<std::mutex> g(M.mu);
        if (!M.loaded.load() || !M.dev) { set_error(error, "Model not loaded"); return false; }
        M.dev->Synchronize();
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
    std::lock_guard<std::mutex> g(M.mu);
    if (num_shards) *num_shards = M.dev ? int(M.replicas.size()) + 1 : 0;
    if (sharded_calls) *sharded_calls = M.shard_calls.load();
    return true;
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
