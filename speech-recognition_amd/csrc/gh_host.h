// Host-side helpers shared by the C-ABI translation units (not used in device code).
#pragma once
#include "gh_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <numeric>
#include <utility>
#include <vector>


// carve 256-byte aligned pieces out of the context scratch
struct Carver {
    size_t total = 0;
    std::vector<std::pair<void**, size_t>> items;  // (destination pointer, offset)
    template <typename T> void add(T** dst, size_t count) {
        items.push_back({reinterpret_cast<void**>(dst), total});
        total += (count * sizeof(T) + 255) & ~size_t(255);
    }
    int commit(gh_ctx* ctx) {
        void* base = nullptr;
        int rc = gh_scratch(ctx, total ? total : 256, &base);
        if (rc) return rc;
        for (auto& it : items) *it.first = static_cast<char*>(base) + it.second;
        return GH_OK;
    }
};

// Several host vectors -> ONE device allocation and ONE copy (a model or a graph set used to cost 13-25 hipMalloc +
// hipMemcpy pairs, ~0.4 ms per creation -- paid every EM iteration by the trainer).  Every piece is 256-byte aligned.
struct UploadArena {
    std::vector<char> host;
    std::vector<std::pair<void**, size_t>> items;
    template <typename T> void add(T** dst, const std::vector<T>& src) {
        *dst = nullptr;
        if (src.empty()) return;
        const size_t off = (host.size() + 255) & ~size_t(255);
        host.resize(off + src.size() * sizeof(T));
        memcpy(host.data() + off, src.data(), src.size() * sizeof(T));
        items.push_back({reinterpret_cast<void**>(dst), off});
    }
    int commit(void** base_out) {
        *base_out = nullptr;
        if (host.empty()) return GH_OK;
        GH_HIP(hipMalloc(base_out, host.size()));
        GH_HIP(hipMemcpy(*base_out, host.data(), host.size(), hipMemcpyHostToDevice));
        for (auto& it : items) *it.first = static_cast<char*>(*base_out) + it.second;
        return GH_OK;
    }
};

// Pieces of ONE device region (scratch or an own allocation): some filled from host memory, some left as they are.
// The host pieces are gathered into one staging buffer and travel in one copy.
struct UploadLayout {
    struct item { void** dst; size_t off, bytes; const void* src; size_t src_bytes; };
    std::vector<item> items;
    size_t total = 0;
    void add(void** dst, size_t bytes, const void* src, size_t src_bytes = 0) {
        items.push_back({dst, total, bytes, src, src ? src_bytes : 0});
        total += (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
    }
    int commit(void* base, hipStream_t st, bool sync) {
        for (auto& it : items) *it.dst = static_cast<char*>(base) + it.off;
        // every maximal run of consecutive pieces WITH host data travels as one copy; a piece that is "left as it is"
        // (no source) between two runs is not touched (one copy over [first, last) would have zeroed it)
        // (the staging buffer holds the RUNS, not the region: a session arena is tens of MB to GBs of device-only
        //  pieces behind a few KB of tables -- zero-filling a region-sized buffer cost 14 ms per gh_fit_create)
        struct run { size_t first, last, lo, hi, at; };
        std::vector<run> runs;
        size_t need = 0;
        for (size_t i = 0; i < items.size();) {
            if (!items[i].src_bytes) { ++i; continue; }
            size_t j = i, hi = items[i].off;
            while (j < items.size() && items[j].src_bytes) { hi = items[j].off + items[j].src_bytes; ++j; }
            runs.push_back({i, j, items[i].off, hi, need});
            need += hi - items[i].off;
            i = j;
        }
        std::vector<char> stage(need, 0);
        bool any = false;
        for (const run& r : runs) {
            for (size_t j = r.first; j < r.last; ++j)
                memcpy(stage.data() + r.at + (items[j].off - r.lo), items[j].src, items[j].src_bytes);
            GH_HIP(hipMemcpyAsync(static_cast<char*>(base) + r.lo, stage.data() + r.at, r.hi - r.lo, hipMemcpyHostToDevice, st));
            any = true;
        }
        if (any || sync) GH_HIP(hipStreamSynchronize(st));    // (`stage` is pageable host memory of this call)
        return GH_OK;
    }
};

template <typename T> inline int upload(T** dst, const std::vector<T>& src) {
    *dst = nullptr;
    if (src.empty()) return GH_OK;
    GH_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
    GH_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return GH_OK;
}



// gh_frontend.hip: cepstra already on the device (scratch) -> resident batch (delta, delta-delta, standardise)
int gh_batch_from_device_cepstra(gh_ctx* ctx, gh_dtype dtype, int mode, int C, int64_t N, int64_t U,
                                 const int64_t* utt_offsets, size_t extra_scratch, void** extra,
                                 const std::function<hipError_t(double*, hipStream_t)>& fill, const char* who,
                                 gh_batch** out);
