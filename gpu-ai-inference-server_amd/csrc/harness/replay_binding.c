/*
 * C replay of the call sequence the reference's Go binding performs against the C ABI
 * (inference_engine/binding/inference_binding.go): Go is not in the build image, so "links and runs unchanged"
 * is demonstrated from C with the same struct fills, malloc'd staging buffers and frees.
 *
 *   NewInferenceManager  -> InferenceInitialize                         (:177-193)
 *   ListModels           -> InferenceListModels / InferenceFreeModelList (:361-385)
 *   LoadModel            -> InferenceLoadModel, GetModelHandle           (:227-285)
 *   GetModel             -> InferenceIsModelLoaded                       (:387-426)
 *   (*Model).Infer       -> ModelIsLoaded, ModelInfer                    (:521-734)
 *   GetStats/GetMetadata -> ModelGetStats / ModelGetMetadata (+frees)    (:739-799)
 *   UnloadModel          -> InferenceUnloadModel, ModelDestroy           (:292-338)
 *   Shutdown             -> InferenceShutdown                            (:195-212)
 *
 * usage: replay_binding <repo> <model> <input_name> <output_name> <n_out_elems> d0 d1 [d2 d3] [--no-gpu]
 * Prints one line per call ("CALL name -> result") and, on success, "OUTPUT v0 v1 ...".  With --no-gpu it
 * expects InferenceLoadModel to fail loudly (no CPU fallback) and exits 0 if it does.
 * The input payload is all ones (test/onnx_test.cpp:92 uses the same probe).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "inference_bridge.h"

int main(int argc, char** argv) {
    if (argc < 8) {
        fprintf(stderr, "usage: %s <repo> <model> <input> <output> <n_out> d0 d1 [d2 d3] [--no-gpu]\n", argv[0]);
        return 2;
    }
    const char *repo = argv[1], *model = argv[2], *in_name = argv[3], *out_name = argv[4];
    size_t n_out = (size_t)atoll(argv[5]);
    int64_t dims[8];
    int nd = 0, no_gpu = 0;
    for (int i = 6; i < argc; ++i) {
        if (strcmp(argv[i], "--no-gpu") == 0) no_gpu = 1;
        else if (nd < 8) dims[nd++] = atoll(argv[i]);
    }
    printf("CALL IsCudaAvailable -> %d\n", (int)IsCudaAvailable());
    printf("CALL GetDeviceCount -> %d\n", GetDeviceCount());
    const char* info = GetDeviceInfo(0);
    printf("CALL GetDeviceInfo -> %s\n", info);
    free((void*)info);                                   /* Go frees with C.free (:146) */
    CudaMemoryInfo mi = GetMemoryInfo(0);
    printf("CALL GetMemoryInfo -> total=%zu free=%zu used=%zu\n", mi.total, mi.free, mi.used);

    InferenceManagerHandle mgr = InferenceInitialize(repo);
    printf("CALL InferenceInitialize -> %s\n", mgr ? "ok" : "NULL");
    if (!mgr) return 1;
    int nm = 0;
    char** names = InferenceListModels(mgr, &nm);
    printf("CALL InferenceListModels -> %d:", nm);
    for (int i = 0; i < nm; ++i) printf(" %s", names[i]);
    printf("\n");
    InferenceFreeModelList(names, nm);

    ErrorMessage err = NULL;
    bool ok = InferenceLoadModel(mgr, model, NULL, &err);
    printf("CALL InferenceLoadModel -> %d%s%s\n", (int)ok, err ? " error=" : "", err ? err : "");
    if (!ok) {
        int rc = (no_gpu && err && strstr(err, "HIP device")) ? 0 : 1;
        FreeErrorMessage(err);
        printf("CALL InferenceIsModelLoaded -> %d\n", (int)InferenceIsModelLoaded(mgr, model, NULL));
        InferenceShutdown(mgr);
        return rc;
    }
    err = NULL;
    ok = InferenceLoadModel(mgr, model, NULL, &err);     /* second load of the same name must fail (bridge:320-325) */
    printf("CALL InferenceLoadModel(again) -> %d error=%s\n", (int)ok, err ? err : "");
    FreeErrorMessage(err);
    err = NULL;
    ModelHandle h = GetModelHandle(mgr, model, NULL, &err);
    printf("CALL GetModelHandle -> %s\n", h ? "ok" : "NULL");
    if (!h) { FreeErrorMessage(err); InferenceShutdown(mgr); return 1; }
    printf("CALL InferenceIsModelLoaded -> %d\n", (int)InferenceIsModelLoaded(mgr, model, NULL));
    printf("CALL ModelIsLoaded -> %d\n", (int)ModelIsLoaded(h));

    /* (*Model).Infer marshalling: malloc'd dims + data per tensor, output buffer uninitialised */
    size_t n_in = 1;
    for (int i = 0; i < nd; ++i) n_in *= (size_t)dims[i];
    TensorData in, out;
    in.name = in_name;
    in.data_type = DATATYPE_FLOAT32;
    in.shape.dims = (int64_t*)malloc(sizeof(int64_t) * (size_t)nd);
    memcpy(in.shape.dims, dims, sizeof(int64_t) * (size_t)nd);
    in.shape.num_dims = nd;
    in.data = malloc(n_in * sizeof(float));
    for (size_t i = 0; i < n_in; ++i) ((float*)in.data)[i] = 1.0f;
    in.data_size = n_in * sizeof(float);
    int64_t odims_cfg[2] = {dims[0], (int64_t)(n_out / (size_t)dims[0])};
    out.name = out_name;
    out.data_type = DATATYPE_FLOAT32;
    out.shape.dims = (int64_t*)malloc(sizeof(int64_t) * 4);
    out.shape.dims[0] = odims_cfg[0]; out.shape.dims[1] = odims_cfg[1]; out.shape.dims[2] = 1; out.shape.dims[3] = 1;
    out.shape.num_dims = 4;
    out.data = malloc(n_out * sizeof(float));
    memset(out.data, 0xAB, n_out * sizeof(float));      /* poison: the engine must overwrite every byte */
    out.data_size = n_out * sizeof(float);
    err = NULL;
    ok = ModelInfer(h, &in, 1, &out, 1, &err);
    printf("CALL ModelInfer -> %d%s%s\n", (int)ok, err ? " error=" : "", err ? err : "");
    int rc = ok ? 0 : 1;
    if (ok) {
        printf("OUTPUT_SHAPE");
        for (int i = 0; i < out.shape.num_dims; ++i) printf(" %lld", (long long)out.shape.dims[i]);
        printf("\nOUTPUT");
        for (size_t i = 0; i < n_out && i < 16; ++i) printf(" %.9g", ((float*)out.data)[i]);
        printf("\n");
    } else FreeErrorMessage(err);
    ModelStats* st = ModelGetStats(h);
    if (st) {
        printf("CALL ModelGetStats -> count=%lld mem=%zu\n", (long long)st->inference_count, st->memory_usage_bytes);
        ModelFreeStats(st);
    }
    ModelMetadata* md = ModelGetMetadata(h);
    if (md) {
        printf("CALL ModelGetMetadata -> name=%s version=%s type=%d inputs=%d(%s) outputs=%d(%s)\n", md->name, md->version,
               (int)md->model_type, md->num_inputs, md->num_inputs ? md->inputs[0] : "", md->num_outputs,
               md->num_outputs ? md->outputs[0] : "");
        ModelFreeMetadata(md);
    }
    free(in.shape.dims); free(in.data); free(out.shape.dims); free(out.data);
    err = NULL;
    ok = InferenceUnloadModel(mgr, model, NULL, &err);
    printf("CALL InferenceUnloadModel -> %d\n", (int)ok);
    FreeErrorMessage(err);
    printf("CALL ModelIsLoaded(after unload) -> %d\n", (int)ModelIsLoaded(h));
    ModelDestroy(h);                                     /* Go destroys its wrapper right after (:325-329) */
    err = NULL;
    ok = InferenceUnloadModel(mgr, model, NULL, &err);
    printf("CALL InferenceUnloadModel(again) -> %d error=%s\n", (int)ok, err ? err : "");
    FreeErrorMessage(err);
    InferenceShutdown(mgr);
    printf("CALL InferenceShutdown -> ok\n");
    return rc;
}
