/*
 * C ABI of libinference_engine.so — MI355X-native engine, drop-in for the reference's bridge.
 *
 * Every identifier, enumerator value, field order and function signature below is an ABI fact the
 * reference's cgo binding compiles against (inference_engine/binding/inference_binding.go:3-9 includes a
 * header of this name and calls these 21 symbols); the declarations correspond one-to-one to
 * inference_engine/include/inference_bridge.h:12-133 of the reference.  Struct sizes on LP64:
 * Shape 16, TensorData 48, ModelConfig 64, ModelMetadata 72, ModelStats 32, CudaMemoryInfo 24
 * (checked by static asserts in csrc/bridge.cpp and by tests/test_abi.py).
 *
 * Ownership: strings returned through ErrorMessage*, GetDeviceInfo and the model list are malloc-family
 * memory released by the caller (FreeErrorMessage / free / InferenceFreeModelList).  Tensor payloads and
 * dims arrays stay caller-owned; the engine never keeps a pointer past the call.  All entry points are
 * thread-safe and never let a C++ exception escape.
 */
#ifndef INFERENCE_BRIDGE_H
#define INFERENCE_BRIDGE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* opaque handles — reference inference_bridge.h:13-15 */
typedef struct InferenceManager_t* InferenceManagerHandle;
typedef struct Model_t* ModelHandle;
typedef struct Tensor_t* TensorHandle;

/* error string, released with FreeErrorMessage — reference inference_bridge.h:18 */
typedef char* ErrorMessage;

/* reference inference_bridge.h:21-31 */
typedef enum {
    DATATYPE_FLOAT32 = 0,
    DATATYPE_INT32 = 1,
    DATATYPE_INT64 = 2,
    DATATYPE_UINT8 = 3,
    DATATYPE_INT8 = 4,
    DATATYPE_STRING = 5,
    DATATYPE_BOOL = 6,
    DATATYPE_FP16 = 7,
    DATATYPE_UNKNOWN = 8
} DataType;

/* reference inference_bridge.h:34-37 */
typedef enum { DEVICE_CPU = 0, DEVICE_GPU = 1 } DeviceType;

/* reference inference_bridge.h:40-47 */
typedef enum {
    MODEL_UNKNOWN = 0,
    MODEL_TENSORFLOW = 1,
    MODEL_TENSORRT = 2,
    MODEL_ONNX = 3,
    MODEL_PYTORCH = 4,
    MODEL_CUSTOM = 5
} ModelType;

/* reference inference_bridge.h:50-53 */
typedef struct {
    int64_t* dims;
    int num_dims;
} Shape;

/* reference inference_bridge.h:56-62 */
typedef struct {
    const char* name;
    DataType data_type;
    Shape shape;
    void* data;
    size_t data_size;
} TensorData;

/* reference inference_bridge.h:65-76 */
typedef struct {
    const char* name;
    const char* version;
    ModelType type_;
    int max_batch_size;
    const char** input_names;
    int num_inputs;
    const char** output_names;
    int num_outputs;
    int instance_count;
    bool dynamic_batching;
} ModelConfig;

/* reference inference_bridge.h:79-89 */
typedef struct {
    const char* name;
    const char* version;
    ModelType model_type;
    const char** inputs;
    int num_inputs;
    const char** outputs;
    int num_outputs;
    const char* description;
    int64_t load_time_ns;
} ModelMetadata;

/* reference inference_bridge.h:92-97 */
typedef struct {
    int64_t inference_count;
    int64_t total_inference_time_ns;
    int64_t last_inference_time_ns;
    size_t memory_usage_bytes;
} ModelStats;

/* reference inference_bridge.h:100-104 */
typedef struct {
    size_t total;
    size_t free;
    size_t used;
} CudaMemoryInfo;

/* Device queries — reference inference_bridge.h:107-110 (impl cuda_utils.cu:17-57,152-176).  The exported names keep
 * the word "Cuda" because the Go binding calls C.IsCudaAvailable (inference_binding.go:135); they report HIP devices. */
bool IsCudaAvailable(void);
int GetDeviceCount(void);
const char* GetDeviceInfo(int device_id);           /* malloc'd; caller free()s (inference_binding.go:145-147) */
CudaMemoryInfo GetMemoryInfo(int device_id);        /* total == 0 signals failure (inference_binding.go:165-167) */

/* Manager — reference inference_bridge.h:113-119 (impl inference_bridge.cpp:254-515) */
InferenceManagerHandle InferenceInitialize(const char* model_repository_path);
void InferenceShutdown(InferenceManagerHandle handle);
bool InferenceLoadModel(InferenceManagerHandle handle, const char* model_name, const char* version, ErrorMessage* error);
bool InferenceUnloadModel(InferenceManagerHandle handle, const char* model_name, const char* version, ErrorMessage* error);
bool InferenceIsModelLoaded(InferenceManagerHandle handle, const char* model_name, const char* version);
char** InferenceListModels(InferenceManagerHandle handle, int* num_models);
void InferenceFreeModelList(char** models, int num_models);

/* Model — reference inference_bridge.h:122-129 (impl inference_bridge.cpp:528-971) */
ModelHandle ModelCreate(const char* model_path, ModelType type, const ModelConfig* config, DeviceType device, int device_id,
                        ErrorMessage* error);
void ModelDestroy(ModelHandle handle);
bool ModelIsLoaded(ModelHandle handle);
bool ModelInfer(ModelHandle handle, const TensorData* inputs, int num_inputs, TensorData* outputs, int num_outputs,
                ErrorMessage* error);
ModelMetadata* ModelGetMetadata(ModelHandle handle);
void ModelFreeMetadata(ModelMetadata* metadata);
ModelStats* ModelGetStats(ModelHandle handle);
void ModelFreeStats(ModelStats* stats);

/* Utilities — reference inference_bridge.h:132-133 */
void FreeErrorMessage(ErrorMessage error);
ModelHandle GetModelHandle(InferenceManagerHandle handle, const char* model_name, const char* version, ErrorMessage* error);

/* Defined extern "C" by the reference but missing from its header (inference_bridge.cpp:603,636); declared here so a
 * handle made by ModelCreate can actually be loaded. */
bool ModelLoad(ModelHandle handle, ErrorMessage* error);
bool ModelUnload(ModelHandle handle, ErrorMessage* error);

#ifdef __cplusplus
}
#endif

#endif /* INFERENCE_BRIDGE_H */
