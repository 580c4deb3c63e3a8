// config.json reader for a model version directory.
//
// The reference's C++ side never parses config.json (model_repository.cpp:131-156 fills a hard-coded ModelConfig); the Go server
// does (server/main.go:605-674: name, version, inputs/outputs with name / dims / shape / data_type / label_filename).  The engine
// reads the same file ONCE at load with a real JSON parser -- top-level keys only, so a "gpus" inside a nested object or a string
// cannot switch anything on -- for the Go-side keys plus its own extension keys:
//   "precision": "fp32" | "fp16" | "fp8"        storage/compute precision of the plan (default fp32)
//   "gpus": n                                    in-process batch sharding over devices device_id .. device_id+n-1
//   "uint8_scale", "uint8_bias"                  on-device transform of DATATYPE_UINT8 payloads (default 1/255, 0)
//   "dynamic_batching": bool, "max_batch_size": n, "batch_window_us": n     request coalescing (model.h:63,70-71 carries the first two)
//   "instance_count": n                          concurrent execution lanes per model (model.h:63)
//   "tune_batches": [b, ...]                     batch sizes to plan + autotune at load (besides inputs[0].shape[0])
//   "fp32_split": true                           fp32 mode: allow the bf16x6 kernels (fp32 products from exactly split bf16 operands)
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace ie {

// Minimal JSON document model (RFC 8259): enough to walk a config file; numbers are kept as double.
struct JsonValue {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;   // insertion order; duplicate keys: the last one wins in find()
    const JsonValue* find(const std::string& key) const;
};
// Throws std::runtime_error("config.json parse error at byte N: ...") on malformed input or trailing garbage.
JsonValue ParseJson(const std::string& text);

struct IoConfig {
    std::string name, data_type, label_filename;
    std::vector<int64_t> dims, shape;
};

struct EngineConfig {
    bool present = false;              // a config.json existed
    std::string name, version, platform;
    std::vector<IoConfig> inputs, outputs;
    std::string precision;             // lower-cased; empty = unset
    int gpus = 0;                      // 0 = unset
    bool has_u8 = false;
    float uint8_scale = 1.0f / 255.0f, uint8_bias = 0.0f;
    bool dynamic_batching = false;
    int max_batch_size = 0;
    int batch_window_us = -1;          // -1 = unset
    int instance_count = 0;            // 0 = unset
    std::vector<int64_t> tune_batches;
    bool fp32_split = false;           // fp32 mode: let the search use the bf16x6 kernels (kernels_x6.hip)
};

EngineConfig ParseEngineConfig(const std::string& json_text);
// <dir>/config.json; a missing file gives present == false, a malformed one throws.
EngineConfig LoadEngineConfig(const std::string& model_dir);

}  // namespace ie
