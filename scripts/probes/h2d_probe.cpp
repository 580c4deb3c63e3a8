// Microbenchmark behind the ModelInfer staging design: what does it cost to get a caller-owned pageable buffer into HBM?
//   hipcc -O2 --offload-arch=gfx950 scripts/probes/h2d_probe.cpp -o gpurun_out/h2d_probe -lpthread && gpurun_out/h2d_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t mb : {19, 77}) {
        const size_t n = mb * 1000 * 1000;
        char* host = static_cast<char*>(std::malloc(n));
        std::memset(host, 1, n);
        void* dev = nullptr;
        CK(hipMalloc(&dev, n));
        void* pinned = nullptr;
        CK(hipHostMalloc(&pinned, n, hipHostMallocDefault));
        std::memset(pinned, 2, n);
        // (a) pageable hipMemcpyAsync + sync
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            double t = now() - t0;
            if (rep) std::printf("%zu MB pageable hipMemcpyAsync: %.3f ms (%.1f GB/s)\n", mb, t * 1e3, n / t / 1e9);
        }
        // (b) pinned
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            double t = now() - t0;
            if (rep) std::printf("%zu MB pinned hipMemcpyAsync: %.3f ms (%.1f GB/s)\n", mb, t * 1e3, n / t / 1e9);
        }
        // (c) register + copy + unregister
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipHostRegister(host, n, hipHostRegisterDefault));
            double t1 = now();
            CK(hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            double t2 = now();
            CK(hipHostUnregister(host));
            double t3 = now();
            std::printf("%zu MB hipHostRegister %.3f ms, copy %.3f ms (%.1f GB/s), unregister %.3f ms\n", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3,
                        n / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
        }
        // (d) threaded memcpy pageable -> pinned
        for (int nt : {1, 2, 4, 8, 12, 16}) {
            double best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                double t0 = now();
                std::vector<std::thread> th;
                const size_t per = (n / nt + 63) & ~size_t(63);
                for (int i = 0; i < nt; ++i)
                    th.emplace_back([&, i] {
                        size_t b = std::min(n, per * i), e = std::min(n, per * (i + 1));
                        if (e > b) std::memcpy(static_cast<char*>(pinned) + b, host + b, e - b);
                    });
                for (auto& t : th) t.join();
                best = std::min(best, now() - t0);
            }
            std::printf("%zu MB memcpy pageable->pinned, %2d threads (spawned): %.3f ms (%.1f GB/s)\n", mb, nt, best * 1e3, n / best / 1e9);
        }
        // (a2) a FRESH pageable buffer per call (the Go binding C.mallocs its payload per request, inference_binding.go:590-640):
        //      how long does the call itself block, and how long until the data is in HBM?
        for (int rep = 0; rep < 4; ++rep) {
            char* fresh = static_cast<char*>(std::malloc(n));
            std::memset(fresh, rep, n);
            double t0 = now();
            CK(hipMemcpyAsync(dev, fresh, n, hipMemcpyHostToDevice, st));
            double t1 = now();
            CK(hipStreamSynchronize(st));
            double t2 = now();
            std::printf("%zu MB FRESH pageable hipMemcpyAsync: call %.3f ms, done after %.3f ms (%.1f GB/s)\n", mb, (t1 - t0) * 1e3, (t2 - t0) * 1e3, n / (t2 - t0) / 1e9);
            std::free(fresh);
        }
        // (a3) the same in 4 chunks (what a pipelined ModelInfer would issue)
        for (int rep = 0; rep < 3; ++rep) {
            char* fresh = static_cast<char*>(std::malloc(n));
            std::memset(fresh, rep, n);
            double t0 = now();
            double tc[4];
            for (int c = 0; c < 4; ++c) {
                CK(hipMemcpyAsync(static_cast<char*>(dev) + c * (n / 4), fresh + c * (n / 4), n / 4, hipMemcpyHostToDevice, st));
                tc[c] = now() - t0;
            }
            CK(hipStreamSynchronize(st));
            double t2 = now();
            std::printf("%zu MB FRESH pageable in 4 chunks: calls return at %.3f %.3f %.3f %.3f ms, done after %.3f ms\n", mb, tc[0] * 1e3, tc[1] * 1e3, tc[2] * 1e3,
                        tc[3] * 1e3, (t2 - t0) * 1e3);
            std::free(fresh);
        }
        // (a4) the same FRESH pageable buffer uploaded by several threads at once, each its own slice on its own stream
        for (int nt : {1, 2, 3, 4}) {
            std::vector<hipStream_t> sts(nt);
            for (auto& s2 : sts) CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            double best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                char* fresh = static_cast<char*>(std::malloc(n));
                std::memset(fresh, rep, n);
                const size_t per = (n / nt + 4095) & ~size_t(4095);
                double t0 = now();
                std::vector<std::thread> th;
                for (int i = 0; i < nt; ++i)
                    th.emplace_back([&, i] {
                        const size_t b = std::min(n, per * i), e = std::min(n, per * (i + 1));
                        if (e > b) {
                            (void)hipMemcpyAsync(static_cast<char*>(dev) + b, fresh + b, e - b, hipMemcpyHostToDevice, sts[i]);
                            (void)hipStreamSynchronize(sts[i]);
                        }
                    });
                for (auto& t : th) t.join();
                best = std::min(best, now() - t0);
                std::free(fresh);
            }
            std::printf("%zu MB FRESH pageable, %d threads x own stream: %.3f ms (%.1f GB/s)\n", mb, nt, best * 1e3, n / best / 1e9);
            for (auto& s2 : sts) (void)hipStreamDestroy(s2);
        }
        // (e) D2H small: 128 KB pinned
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(pinned, dev, 128 * 1000, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double t = now() - t0;
            if (rep) std::printf("128 KB D2H pinned + sync: %.1f us\n", t * 1e6);
        }
        CK(hipFree(dev));
        CK(hipHostFree(pinned));
        std::free(host);
    }
    std::printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    return 0;
}
