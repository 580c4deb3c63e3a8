// ONNX ModelProto wire-format reader (subset needed for inference graphs).
//
// Replaces what the reference gets from `Ort::Session(env, "<dir>/model.onnx", opts)`
// (inference_engine/src/model.cpp:843-847) and the introspection calls of ExtractModelMetadata
// (model.cpp:910-972): graph nodes, initializers, graph input/output names, shapes and element types.
// No protobuf library: the container has none for C++, and the subset is small.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace ie {

enum OnnxDType : int { ONNX_FLOAT = 1, ONNX_UINT8 = 2, ONNX_INT8 = 3, ONNX_INT32 = 6, ONNX_INT64 = 7,
                       ONNX_BOOL = 9, ONNX_FLOAT16 = 10, ONNX_DOUBLE = 11 };

struct OnnxTensor {
    std::string name;
    int dtype = ONNX_FLOAT;
    std::vector<int64_t> dims;
    std::vector<float> f;      // FLOAT / DOUBLE / FLOAT16 payloads converted to float
    std::vector<int64_t> i;    // INT32 / INT64 payloads
    int64_t numel() const { int64_t n = 1; for (auto d : dims) n *= d; return n; }
};

struct OnnxAttr {
    std::string name;
    int type = 0;              // AttributeProto.AttributeType
    float f = 0.f;
    int64_t i = 0;
    std::string s;
    std::vector<int64_t> ints;
    std::vector<float> floats;
    bool has_t = false;
    OnnxTensor t;
};

struct OnnxNode {
    std::string op, name;
    std::vector<std::string> inputs, outputs;
    std::map<std::string, OnnxAttr> attrs;
    int64_t attr_i(const std::string& k, int64_t def) const {
        auto it = attrs.find(k); return it == attrs.end() ? def : it->second.i;
    }
    float attr_f(const std::string& k, float def) const {
        auto it = attrs.find(k); return it == attrs.end() ? def : it->second.f;
    }
    std::vector<int64_t> attr_ints(const std::string& k, std::vector<int64_t> def) const {
        auto it = attrs.find(k); return it == attrs.end() ? def : it->second.ints;
    }
};

struct OnnxValueInfo {
    std::string name;
    int elem_type = ONNX_FLOAT;
    std::vector<int64_t> dims;   // -1 for symbolic / unknown dims (as ORT's GetShape reports them)
};

struct OnnxModel {
    int64_t ir_version = 0;
    int64_t opset = 0;
    std::string producer;
    std::string graph_name;
    std::vector<OnnxNode> nodes;
    std::map<std::string, OnnxTensor> initializers;
    std::vector<OnnxValueInfo> inputs;    // graph inputs that are not initializers
    std::vector<OnnxValueInfo> outputs;
};

// Throws std::runtime_error on malformed input.
OnnxModel ParseOnnx(const uint8_t* data, size_t size);
OnnxModel LoadOnnxFile(const std::string& path);

}  // namespace ie
