/*
 * Extensions of the MI355X engine beyond the reference ABI (nothing here is needed by the Go binding).
 * They exist for (a) device-resident benchmarking — the reference's ModelInfer (inference_bridge.cpp:692-828)
 * only takes host buffers, (b) per-kernel roofline measurement, (c) the RCCL weight broadcast that the
 * one-process-per-GPU launcher performs with torch.distributed, (d) host-only graph inspection for CPU tests.
 * Plain pointers and sizes only; strings are malloc'd and released with FreeErrorMessage().
 */
#ifndef INFERENCE_ENGINE_EXT_H
#define INFERENCE_ENGINE_EXT_H

#include "inference_bridge.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host-only (no GPU needed): parse <path>/model.onnx (or a .onnx file) and return a JSON document with the graph's
 * inputs/outputs, the reference's memory_usage_bytes estimate (model.cpp:979-1035) and, when batch > 0, the fused
 * execution plan for that batch (symbolic leading dims are replaced by `batch`).  NULL + *error on failure. */
char* EngineDescribeModel(const char* path, int batch, ErrorMessage* error);

/* Host-only: the packed fp32 weight blob of the plan (folded BatchNorm scale/shift vectors, biases, conv weights repacked
 * [Cout][kh][kw][Cin]) whose element offsets EngineDescribeModel's plan steps carry (w_off, bias_off, pre_scale_off, pre_shift_off).
 * The blob is batch independent and identical for every precision mode.  malloc'd, release with FreeErrorMessage(). */
float* EnginePlanWeights(const char* path, int batch, size_t* count, ErrorMessage* error);

/* Plan + allocate + capture the hipGraph for these input shapes (one Shape per graph input, graph order) and return
 * the engine-owned device buffers: d_inputs[i] is dense NCHW fp32 of input i, d_outputs[j] dense fp32 of output j. */
bool EnginePrepare(ModelHandle handle, const Shape* input_shapes, int num_inputs, void** d_inputs, void** d_outputs,
                   int num_outputs, ErrorMessage* error);
/* Enqueue `iters` forwards of the last prepared plan on the model's stream using the data already in its device
 * buffers; waits for completion when sync != 0. */
bool EngineRunPrepared(ModelHandle handle, int iters, int sync, ErrorMessage* error);
bool EngineSynchronize(ModelHandle handle, ErrorMessage* error);
/* hipStream_t the model launches on (for HIP-event timing by the caller). */
void* EngineGetStream(ModelHandle handle);
/* Eager (non-graph) forwards of the prepared plan with HIP events around every kernel launch; JSON list of
 * {name, kernel, ms, flops, bytes} per step (ms averaged over iters). */
char* EngineProfile(ModelHandle handle, int iters, ErrorMessage* error);
/* Packed fp32 weight blob in HBM (folded BN scale/shift, repacked conv weights). */
bool EngineGetWeightBlob(ModelHandle handle, void** d_ptr, size_t* bytes, ErrorMessage* error);
/* Tell the engine the fp32 blob was rewritten in place (e.g. by an RCCL broadcast): in fp16 precision mode the half
 * mirror the MFMA path reads is re-derived from it.  No-op in fp32 mode. */
bool EngineWeightsUpdated(ModelHandle handle, ErrorMessage* error);
/* 0 = fp32, 1 = fp16, 2 = fp8 (IE_PRECISION or config.json "precision"), -1 = model not loaded. */
int EngineGetPrecision(ModelHandle handle);
/* Synchronous hipMemcpy on the model's device: kind 1 = host->device, 2 = device->host, 3 = device->device.
 * Lets a test or benchmark fill / read the engine-owned buffers returned by EnginePrepare. */
bool EngineMemcpy(ModelHandle handle, void* dst, const void* src, size_t bytes, int kind, ErrorMessage* error);
/* Calibration microbenchmark: TFLOP/s of a register-resident v_mfma_f32_32x32x2_f32 loop with `nacc` (1, 2 or 4)
 * independent accumulator chains per wave and `blocks_per_cu` 4-wave workgroups per CU; <= 0 on error. */
double EngineMfmaPeak(int nacc, int blocks_per_cu, int iters);
/* Dynamic request batcher counters: device batches run, caller requests folded into them, and the active row limit
 * (0 = batching off for this model).  Enable with IE_DYNAMIC_BATCH=<rows> or config.json {"dynamic_batching": true,
 * "max_batch_size": N} or ModelCreate's ModelConfig fields; IE_BATCH_WINDOW_US sets the coalescing window (default 200). */
bool EngineGetBatcherStats(ModelHandle handle, int64_t* device_batches, int64_t* coalesced_requests, int* max_batch);
/* In-process batch sharding: number of model replicas a ModelInfer request is cut over (1 = off) and the number of requests that
 * were sharded.  Enable with IE_GPUS=<n> or config.json {"gpus": n} (devices device_id .. device_id+n-1, clipped to the device count)
 * or IE_SHARD_DEVICES=<id,id,...> (explicit list, ids may repeat).  Needs a graph with a symbolic batch axis. */
bool EngineGetShardStats(ModelHandle handle, int* num_shards, int64_t* sharded_calls);
/* JSON description of a loaded model's runtime: precision, execution lanes (config.json "instance_count") and shard replicas with
 * their devices and which lanes share a weight allocation, the high-water mark of concurrent requests, the RCCL weight broadcast
 * done at load (ranks, bytes, milliseconds), HIP-event device time of the forwards with the planner's algorithmic TFLOP/s and GB/s
 * against the gfx950 peaks, and counters of the pipelined host path.  with_checksums != 0 adds an FNV-1a checksum of every lane's
 * packed weight blob as it sits in HBM (waits for in-flight requests).  The one-line form of the same facts is what
 * ModelGetMetadata() returns in `description`.  malloc'd, release with FreeErrorMessage(). */
char* EngineGetRuntimeInfo(ModelHandle handle, int with_checksums, ErrorMessage* error);
/* The device's OCP e4m3 conversion as the fp8 kernels use it: codes[i] = e4m3(src[i] / scale) (round to nearest even, saturating at
 * +-448), dst[i] = decode(codes[i]) * scale.  `codes` may be NULL.  Test support for the fp8 precision mode. */
bool EngineE4m3RoundTrip(const float* src, float* dst, unsigned char* codes, size_t n, float scale, ErrorMessage* error);
/* result = a + b on the GPU for host arrays (the reference's VectorAdd smoke test, cuda_utils.cu:59-149). */
bool EngineVectorAdd(const float* a, const float* b, float* result, size_t n, ErrorMessage* error);

#ifdef __cplusplus
}
#endif

#endif /* INFERENCE_ENGINE_EXT_H */
