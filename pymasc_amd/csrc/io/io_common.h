// Shared plumbing of libpymasc_io.so: error reporting, read-only file mapping, thread count.
#ifndef PMX_IO_COMMON_H
#define PMX_IO_COMMON_H

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/pymasc_amd_io.h"

namespace pmx_io {

struct Error {
    int code;
    std::string msg;
    Error(int c, std::string m) : code(c), msg(std::move(m)) {}
};

// Records msg as the calling thread's last error and returns code.
int fail(int code, const std::string &msg);

int pick_threads(int requested);

// Whole file mapped read-only (MAP_PRIVATE); pages are faulted in as the readers walk it.
struct MappedFile {
    const uint8_t *data = nullptr;
    size_t size = 0;
    void open(const char *path);     // throws Error(PMX_IO_ERR_OPEN)
    void close();
    ~MappedFile() { close(); }
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

// Runs fn(lo, hi, chunk) over [0, n) cut into chunks of `grain` on up to nthreads threads; rethrows the first failure.
template <class F>
void parallel_for(int nthreads, size_t n, size_t grain, F fn)
{
    if (n == 0) return;
    const size_t chunks = (n + grain - 1) / grain;
    int t = (int)std::min<size_t>((size_t)nthreads, chunks);
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    Error first(0, "");
    std::mutex mu;
    auto body = [&]() {
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= chunks || failed.load()) return;
            const size_t lo = c * grain, hi = std::min(n, lo + grain);
            try {
                fn(lo, hi, c);
            } catch (const Error &e) {
                std::lock_guard<std::mutex> g(mu);
                if (!failed.exchange(true)) first = e;
                return;
            } catch (const std::exception &e) {   // bad_alloc / length_error from a growing vector: an exception that
                std::lock_guard<std::mutex> g(mu);   // leaves a std::thread body ends the process (and its GPU context)
                if (!failed.exchange(true)) first = Error(PMX_IO_ERR_FORMAT, std::string("worker thread: ") + e.what());
                return;
            } catch (...) {
                std::lock_guard<std::mutex> g(mu);
                if (!failed.exchange(true)) first = Error(PMX_IO_ERR_FORMAT, "worker thread: unknown exception");
                return;
            }
        }
    };
    if (t <= 1) {
        body();
    } else {
        std::vector<std::thread> th;
        th.reserve(t - 1);
        for (int i = 1; i < t; i++) th.emplace_back(body);
        body();
        for (auto &x : th) x.join();
    }
    if (failed.load()) throw first;
}

}  // namespace pmx_io
#endif
