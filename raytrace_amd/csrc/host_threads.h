// Host-side worker threads for passes over large arrays (BVH build, occupancy map).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace rt {

// Host threads for the passes over all faces of a large mesh (bounds, binning, partition, leaf-order copies).  A GPU box
// gives a process a share of its CPUs, not all that hardware_concurrency() reports: 16 at most; RT_HOST_THREADS overrides.
inline int host_threads() {
    if (const char* e = std::getenv("RT_HOST_THREADS")) return std::max(1, std::min(64, std::atoi(e)));
    static const int n = std::max(1, std::min((int)std::thread::hardware_concurrency(), 16));
    return n;
}

// f(part, begin, end) over `parts` contiguous chunks of [lo, hi), one thread each (the caller's thread takes the last).
template <class F>
inline void parallel_chunks(uint64_t lo, uint64_t hi, int parts, F f) {
    const uint64_t n = hi - lo;
    parts = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)parts, n));
    auto begin_of = [&](int p) { return lo + n * (uint64_t)p / (uint64_t)parts; };
    std::vector<std::thread> pool;
    pool.reserve((size_t)parts - 1);
    for (int p = 0; p + 1 < parts; p++) pool.emplace_back([&, p] { f(p, begin_of(p), begin_of(p + 1)); });
    f(parts - 1, begin_of(parts - 1), hi);
    for (std::thread& t : pool) t.join();
}

}  // namespace rt
