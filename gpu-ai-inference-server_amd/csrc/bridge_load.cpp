// Model load / unload behind InferenceLoadModel / ModelLoad (reference: inference_engine/src/model.cpp:503-548, 618-648, 825-871): plan + weights on the
// primary lane, shard replicas on the other devices, ONE RCCL broadcast of the packed weight blob at load, extra lanes per device.
#include "bridge_internal.h"

namespace ie_bridge {


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


}  // namespace ie_bridge
