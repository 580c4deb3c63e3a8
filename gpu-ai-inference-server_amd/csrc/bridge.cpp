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
#include "bridge_internal.h"

static_assert(sizeof(Shape) == 16, "Shape layout");
static_assert(sizeof(TensorData) == 48 && offsetof(TensorData, data) == 32 && offsetof(TensorData, data_size) == 40, "TensorData layout");
static_assert(sizeof(ModelConfig) == 64 && offsetof(ModelConfig, dynamic_batching) == 56, "ModelConfig layout");
static_assert(sizeof(ModelMetadata) == 72 && offsetof(ModelMetadata, load_time_ns) == 64, "ModelMetadata layout");
static_assert(sizeof(ModelStats) == 32, "ModelStats layout");
static_assert(sizeof(CudaMemoryInfo) == 24, "CudaMemoryInfo layout");

using namespace ie_bridge;

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
