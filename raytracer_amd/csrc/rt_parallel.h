// Host-side parallel loop over index ranges (scene ingest: per-primitive records, shape boxes, BVH partitioning).
#ifndef RT_PARALLEL_H
#define RT_PARALLEL_H

#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

// Worker threads for ingest loops: RT_AMD_THREADS, else OMP_NUM_THREADS, else the hardware count, capped at 16
// (a GPU box shows the whole machine's CPUs; one GPU's share is 16).
inline int rt_worker_count() {
    for (const char* name : { "RT_AMD_THREADS", "OMP_NUM_THREADS" }) {
        const char* v = std::getenv(name);
        if (v && *v) { int n = std::atoi(v); if (n >= 1) return std::min(n, 64); }
    }
    unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hw ? hw : 1u, 16u));
}

// fn(begin, end, worker) over [0, n) in contiguous chunks, one per worker; serial below min_per_worker items each.
template <class F> void rt_parallel_for(size_t n, size_t min_per_worker, F&& fn) {
    size_t workers = (size_t)rt_worker_count();
    if (min_per_worker < 1) min_per_worker = 1;
    workers = std::min(workers, std::max<size_t>(1, n / min_per_worker));
    if (workers <= 1) { if (n) fn((size_t)0, n, 0); return; }
    std::vector<std::thread> pool; pool.reserve(workers - 1);
    const size_t chunk = (n + workers - 1) / workers;
    for (size_t w = 1; w < workers; ++w) {
        size_t b = std::min(n, w * chunk), e = std::min(n, b + chunk);
        if (b < e) pool.emplace_back([&fn, b, e, w]() { fn(b, e, (int)w); });
    }
    fn((size_t)0, std::min(n, chunk), 0);
    for (auto& t : pool) t.join();
}

#endif
