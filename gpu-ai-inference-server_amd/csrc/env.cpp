#include "env.h"

#include <atomic>
#include <cstdlib>

namespace ie {

namespace {
// name, meaning.  Keep this list complete: Env::Read() sees nothing else.
const char* const kNames[] = {
    // ---- what runs -------------------------------------------------------------------------------------------------------------
    "IE_PRECISION",            // fp32 | fp16 | fp8 (config.json "precision")
    "IE_FP32_SPLIT",           // 1: allow the bf16x6 1x1 kernels in fp32 mode (config.json "fp32_split")
    "IE_DEVICE_ID",            // HIP device of a model created through ModelCreate
    "IE_GPUS",                 // shard replicas on devices 0..n-1 (config.json "gpus")
    "IE_SHARD_DEVICES",        // explicit device list of the shard replicas, e.g. 0,0,0 on a one-GPU box
    "IE_SHARD_PRIVATE_WEIGHTS",// 1: a same-device replica owns its weight blob (RCCL moves real bytes on one GPU)
    "IE_INSTANCES",            // execution lanes on the primary device (config.json "instance_count")
    "IE_DYNAMIC_BATCH",        // max coalesced rows of the request batcher (config.json "max_batch_size" + "dynamic_batching")
    "IE_BATCH_WINDOW_US",      // how long the batcher's leader waits for more callers
    "IE_VERSION_ORDER",        // go: numeric "latest version" like the Go server (default: the C++ repository's lexicographic order)
    "IE_ROCTX",                // 1: one ROCTX range per ModelInfer
    // ---- planner -----------------------------------------------------------------------------------------------------------------
    "IE_NO_POOL_SWAP", "IE_NO_DENSE_FUSE", "IE_NO_DENSE_BLOCK", "IE_DENSE_BAND", "IE_NO_DUAL_F8", "IE_NO_STEM_POOL", "IE_FUSE_MAX_M", "IE_FUSE_PB",
    "IE_FORCE_ALGO", "IE_FORCE_TILE", "IE_FORCE_SPLITK",      // tests: pin the kernel family / tile / split-K of every conv
    // ---- executor ----------------------------------------------------------------------------------------------------------------
    "IE_AUTOTUNE", "IE_TUNE_CACHE", "IE_TUNE_BATCHES", "IE_TUNE_ON_DEMAND", "IE_TUNE_HOT", "IE_TUNE_LOG",
    "IE_DISABLE_GRAPH", "IE_SPLITK_IN_LAUNCH", "IE_PIPELINE_CHUNKS", "IE_PIPELINE_HEAD", "IE_MAX_PLANS", "IE_NO_FRAG_WEIGHTS",
    "IE_F8_CALIB_BATCH", "IE_F8_MARGIN",
    "IE_MAX_INFLIGHT_REPLAYS", // EngineRunPrepared synchronises every n graph replays (profilers: deep un-synchronised queues crash rocprofv3)
    // ---- launch-path debugging knobs (LaunchKnobs) ------------------------------------------------------------------------------
    "IE_DEBUG_ABLATE", "IE_AS_PAD", "IE_NO_PERSISTENT",
    nullptr};

LaunchKnobs g_knobs;
std::atomic<bool> g_knobs_set{false};
}  // namespace

Env Env::Read() {
    Env e;
    for (const char* const* n = kNames; *n; ++n)
        if (const char* v = std::getenv(*n)) e.kv_[*n] = v;
    return e;
}

int Env::integer(const char* name, int dflt) const {
    const char* v = get(name);
    return v ? std::atoi(v) : dflt;
}

const char* const* Env::Names() { return kNames; }

const LaunchKnobs& Knobs() { return g_knobs; }

void SetLaunchKnobs(const Env& env) {
    LaunchKnobs k;
    k.debug_ablate = env.integer("IE_DEBUG_ABLATE", 0);
    const int pad = env.integer("IE_AS_PAD", 8);
    k.as_pad = (pad >= 4 && pad <= 68 && pad % 4 == 0) ? pad : 8;
    k.no_persistent = env.flag("IE_NO_PERSISTENT");
    g_knobs = k;             // plain stores of ints: lanes are constructed before they launch anything
    g_knobs_set.store(true);
}

}  // namespace ie
