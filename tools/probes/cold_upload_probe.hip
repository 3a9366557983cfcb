// VERDICT r4 item 8 probe: a cold weather-model file (115 MB of f64 totals, a page-cache mapping) to the device.
//   A  hipMemcpy from the mapping (pageable: what stage_in does today)
//   B  T host threads copy 4 MB chunks of the mapping into a ring of 8 page-locked buffers, one thread submits the DMAs in order
// hipcc --offload-arch=gfx950 -O3 -pthread cold_upload_probe.hip -o cold_upload_probe.bin ;  ./cold_upload_probe.bin [MB] 
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 115) << 20;
    const char* path = "/tmp/cold_upload_probe.dat";
    { FILE* f = fopen(path, "wb"); std::vector<char> b(1 << 20, 1); for (size_t i = 0; i < bytes >> 20; ++i) fwrite(b.data(), 1, b.size(), f); fclose(f); }
    void* dev; CK(hipMalloc(&dev, bytes));
    constexpr size_t CH = 4 << 20; constexpr int S = 8;
    char* pin[S]; hipEvent_t ev[S]; hipStream_t st; CK(hipStreamCreate(&st));
    for (int i = 0; i < S; ++i) { CK(hipHostMalloc((void**)&pin[i], CH, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
    auto fresh_map = [&]() -> const char* {            // a NEW mapping of the (page-cache resident) file: first touch of every page pays its minor fault
        int fd = open(path, O_RDONLY); void* m = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0); close(fd); return (const char*)m; };
    printf("{\"bytes\": %zu", bytes);
    for (int rep = 0; rep < 3; ++rep) {
        const char* m = fresh_map(); const double t0 = now();
        CK(hipMemcpy(dev, m, bytes, hipMemcpyHostToDevice));
        const double t = now() - t0; munmap((void*)m, bytes);
        printf(", \"A_pageable_ms_%d\": %.3f, \"A_GBps_%d\": %.1f", rep, t * 1e3, rep, bytes / t / 1e9);
    }
    // which half of A is fast: the synchronous call, or the private mapping?  (the product: np.memmap = MAP_SHARED, hipMemcpyAsync on a stream)
    for (int shared = 0; shared < 2; ++shared) for (int async = 0; async < 2; ++async) for (int populate = 0; populate < 2; ++populate) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            int fd = open(path, O_RDONLY);
            const double t0 = now();
            void* m = mmap(nullptr, bytes, PROT_READ, (shared ? MAP_SHARED : MAP_PRIVATE) | (populate ? MAP_POPULATE : 0), fd, 0);
            if (async) { CK(hipMemcpyAsync(dev, m, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
            else CK(hipMemcpy(dev, m, bytes, hipMemcpyHostToDevice));
            best = std::min(best, now() - t0); munmap(m, bytes); close(fd);
        }
        printf(", \"%s_%s%s_ms\": %.3f", shared ? "shared" : "private", async ? "async" : "sync", populate ? "_populate" : "", best * 1e3);
    }
    const size_t C = (bytes + CH - 1) / CH;
    for (int T : {1, 2, 4, 8}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            const char* m = fresh_map();
            std::vector<std::atomic<int>> ready(C), done(C);
            for (size_t k = 0; k < C; ++k) { ready[k] = 0; done[k] = 0; }
            std::atomic<size_t> next{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&] {
                for (;;) {
                    const size_t k = next.fetch_add(1); if (k >= C) return;
                    if (k >= (size_t)S) while (!done[k - S].load(std::memory_order_acquire)) std::this_thread::yield();
                    const size_t len = std::min(CH, bytes - k * CH);
                    memcpy(pin[k % S], m + k * CH, len);
                    ready[k].store(1, std::memory_order_release);
                } });
            size_t polled = 0;
            auto poll = [&](size_t upto) { while (polled < upto && hipEventQuery(ev[polled % S]) == hipSuccess) { done[polled].store(1, std::memory_order_release); ++polled; } };
            for (size_t k = 0; k < C; ++k) {
                while (!ready[k].load(std::memory_order_acquire)) { poll(k); std::this_thread::yield(); }
                const size_t len = std::min(CH, bytes - k * CH);
                CK(hipMemcpyAsync((char*)dev + k * CH, pin[k % S], len, hipMemcpyHostToDevice, st));
                CK(hipEventRecord(ev[k % S], st));
                poll(k + 1);
            }
            CK(hipStreamSynchronize(st));
            for (size_t k = polled; k < C; ++k) done[k].store(1);
            for (auto& t : th) t.join();
            best = std::min(best, now() - t0); munmap((void*)m, bytes);
        }
        printf(", \"B_threads%d_ms\": %.3f, \"B_threads%d_GBps\": %.1f", T, best * 1e3, T, bytes / best / 1e9);
    }
    printf("}\n");
    unlink(path);
    return 0;
}
