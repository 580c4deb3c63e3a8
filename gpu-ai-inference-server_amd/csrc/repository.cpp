#include "repository.h"
#include "env.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <filesystem>

namespace fs = std::filesystem;

namespace ie {

Repository::Repository(const std::string& root) : root_(root) {
    std::error_code ec;
    if (!root_.empty() && !fs::exists(root_, ec)) fs::create_directories(root_, ec);
}

static bool has_model_files(const fs::path& d) {
    std::error_code ec;
    for (const char* f : {"config.json", "model.onnx", "model.pt", "saved_model.pb", "model.plan"})
        if (fs::exists(d / f, ec)) return true;
    return false;
}

// "newest first": numeric-descending by the leading integer (std::stoi semantics), string-descending when a
// version is not numeric (model_repository.cpp:44-53).
static bool version_before(const std::string& a, const std::string& b) {
    try {
        return std::stoi(a) > std::stoi(b);
    } catch (const std::exception&) {
        return a > b;
    }
}

bool Repository::Scan() {
    std::map<std::string, std::vector<std::string>> found;
    std::error_code ec;
    if (!fs::exists(root_, ec)) return false;
    try {
        for (const auto& md : fs::directory_iterator(root_)) {
            if (!md.is_directory()) continue;
            std::vector<std::string> vs;
            for (const auto& vd : fs::directory_iterator(md.path()))
                if (vd.is_directory() && has_model_files(vd.path())) vs.push_back(vd.path().filename().string());
            // a comparator that throws for some pairs and not others is not a strict weak order; decide once
            bool numeric = true;
            for (auto& v : vs) { try { (void)std::stoi(v); } catch (const std::exception&) { numeric = false; } }
            // IE_VERSION_ORDER=go: the unchanged Go server picks the config.json of the LEXICOGRAPHICALLY last all-digit directory
            // (server/main.go:640-655: sort.Strings over isNumeric names), e.g. "2" over "10", while the reference's C++ side -- and
            // this engine by default -- loads the numerically latest (model_repository.cpp:45-53).  With versions >= 10 the two
            // disagree about which version "latest" is; this switch makes the engine follow the Go side's rule.
            const Env env = Env::Read();
            const char* ord = env.get("IE_VERSION_ORDER");
            if (ord && (std::strcmp(ord, "go") == 0 || std::strcmp(ord, "lexicographic") == 0)) {
                std::stable_sort(vs.begin(), vs.end(), [](const std::string& a, const std::string& b) {
                    auto digits = [](const std::string& v) { return !v.empty() && std::all_of(v.begin(), v.end(), [](unsigned char c) { return std::isdigit(c) != 0; }); };
                    const bool da = digits(a), db = digits(b);
                    if (da != db) return da;                  // Go only considers all-digit names
                    return a > b;
                });
            } else if (numeric) std::stable_sort(vs.begin(), vs.end(), version_before);
            else std::sort(vs.begin(), vs.end(), std::greater<std::string>());
            if (!vs.empty()) found[md.path().filename().string()] = vs;
        }
    } catch (const std::exception&) {
        return false;
    }
    std::lock_guard<std::mutex> g(mu_);
    versions_.swap(found);
    return true;
}

std::vector<std::string> Repository::Models() const {
    std::lock_guard<std::mutex> g(mu_);
    std::vector<std::string> out;
    for (auto& kv : versions_) out.push_back(kv.first);
    return out;
}

std::vector<std::string> Repository::Versions(const std::string& model) const {
    std::lock_guard<std::mutex> g(mu_);
    auto it = versions_.find(model);
    return it == versions_.end() ? std::vector<std::string>{} : it->second;
}

std::string Repository::LatestVersion(const std::string& model) const {
    std::lock_guard<std::mutex> g(mu_);
    auto it = versions_.find(model);
    return (it == versions_.end() || it->second.empty()) ? "" : it->second.front();
}

std::string Repository::ModelPath(const std::string& model, const std::string& version) const {
    std::lock_guard<std::mutex> g(mu_);
    auto it = versions_.find(model);
    if (it == versions_.end() || it->second.empty()) return "";
    std::string v = version;
    if (v.empty()) v = it->second.front();
    else if (std::find(it->second.begin(), it->second.end(), v) == it->second.end()) return "";
    return (fs::path(root_) / model / v).string();
}

RepoModelType Repository::DetectType(const std::string& dir) {
    std::error_code ec;
    fs::path d(dir);
    if (fs::exists(d / "model.onnx", ec)) return RepoModelType::Onnx;
    if (fs::exists(d / "saved_model.pb", ec)) return RepoModelType::TensorFlow;
    if (fs::exists(d / "model.plan", ec)) return RepoModelType::TensorRT;
    if (fs::exists(d / "model.pt", ec)) return RepoModelType::PyTorch;
    return RepoModelType::Unknown;
}

}  // namespace ie
