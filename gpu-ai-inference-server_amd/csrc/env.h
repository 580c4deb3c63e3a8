// Every environment switch of the engine in ONE place.  The process environment is read by Env::Read() only -- once per model load
// (bridge.cpp ModelObj::Load), once per lane (DeviceModel's constructor) and once per plan build (BuildPlan) -- never on a launch or request
// path; the copies are immutable afterwards.  config.json keys (csrc/config.cpp) are the production interface; these switches exist for tests,
// A/B measurements and debugging, and an environment value overrides the config key where both exist.
#pragma once
#include <map>
#include <string>

namespace ie {

class Env {
public:
    static Env Read();                                     // snapshot of the switches listed in env.cpp (unknown IE_* names are ignored)
    const char* get(const char* name) const {              // nullptr = unset, like getenv
        auto it = kv_.find(name);
        return it == kv_.end() ? nullptr : it->second.c_str();
    }
    bool flag(const char* name) const {                    // set and not "0" / empty
        const char* v = get(name);
        return v && v[0] != 0 && !(v[0] == '0' && v[1] == 0);
    }
    int integer(const char* name, int dflt) const;
    static const char* const* Names();                     // null-terminated list of every switch (documentation / tests)

private:
    std::map<std::string, std::string> kv_;
};

// Launch-path knobs that used to be read with getenv inside the launchers: set once from a lane's Env (DeviceModel's constructor).
struct LaunchKnobs {
    int debug_ablate = 0;      // IE_DEBUG_ABLATE: timing-only ablation bits (0 in production)
    int as_pad = 8;            // IE_AS_PAD: LDS row pad (floats) of conv1x1_as_kernel (A/B of the bank-conflict fix)
    bool no_persistent = false;   // IE_NO_PERSISTENT: one tile per workgroup in the tiled implicit GEMM
};
const LaunchKnobs& Knobs();
void SetLaunchKnobs(const Env& env);

}  // namespace ie
