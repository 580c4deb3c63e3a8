// Model repository: <root>/<model>/<version>/{model.onnx, config.json}.
// Behavioural mirror of the reference's ModelRepository (inference_engine/src/model_repository.cpp:10-205),
// written fresh; thread-safe (the reference's is not).
#pragma once
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace ie {

enum class RepoModelType { Unknown = 0, TensorFlow = 1, TensorRT = 2, Onnx = 3, PyTorch = 4, Custom = 5 };

class Repository {
public:
    explicit Repository(const std::string& root);            // creates the directory when missing (ref :10-16)
    bool Scan();                                              // ref :18-66
    std::vector<std::string> Models() const;                  // sorted by name (std::map order, ref :68-74)
    std::vector<std::string> Versions(const std::string& model) const;
    std::string LatestVersion(const std::string& model) const;                         // ref :180-187
    std::string ModelPath(const std::string& model, const std::string& version) const; // "" when unknown (ref :91-113)
    static RepoModelType DetectType(const std::string& dir);                           // ref :161-178
    const std::string& root() const { return root_; }

private:
    std::string root_;
    mutable std::mutex mu_;
    std::map<std::string, std::vector<std::string>> versions_;
};

}  // namespace ie
