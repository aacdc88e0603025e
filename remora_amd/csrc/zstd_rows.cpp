// Parallel zstd decompression of POD5 signal rows for the ingest (host code).
//
// A POD5 signal row is zstd( streamvbyte16( zigzag( delta( int16 samples )))) - pod5's C++ reader inflates it for
// io.iter_signal (src/remora/io.py:441-474).  Here the zstd layer of a whole batch of rows is inflated by a few
// threads straight into the staging buffer rmr_vbz_decode uploads (the layers below run on the GPU).  libzstd is
// taken from the system at run time (dlopen; its one-shot API is re-entrant).
#include <dlfcn.h>

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/remora_hip.h"
#include "rmr_internal.h"

using rmr::set_error;

namespace {

typedef unsigned long long (*fn_content_size)(const void *, size_t);
typedef size_t (*fn_decompress)(void *, size_t, const void *, size_t);
typedef unsigned (*fn_is_error)(size_t);
typedef void *(*fn_create_dctx)();
typedef size_t (*fn_free_dctx)(void *);
typedef size_t (*fn_decompress_dctx)(void *, void *, size_t, const void *, size_t);

struct Zstd {
    fn_content_size content_size = nullptr;
    fn_decompress decompress = nullptr;
    fn_is_error is_error = nullptr;
    fn_create_dctx create_dctx = nullptr;
    fn_free_dctx free_dctx = nullptr;
    fn_decompress_dctx decompress_dctx = nullptr;
    bool ok = false;
};

const Zstd &zstd() {
    static Zstd z = [] {
        Zstd r;
        void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
        if (h) {
            r.content_size = (fn_content_size)dlsym(h, "ZSTD_getFrameContentSize");
            r.decompress = (fn_decompress)dlsym(h, "ZSTD_decompress");
            r.is_error = (fn_is_error)dlsym(h, "ZSTD_isError");
            r.create_dctx = (fn_create_dctx)dlsym(h, "ZSTD_createDCtx");
            r.free_dctx = (fn_free_dctx)dlsym(h, "ZSTD_freeDCtx");
            r.decompress_dctx = (fn_decompress_dctx)dlsym(h, "ZSTD_decompressDCtx");
            r.ok = r.content_size && r.decompress && r.is_error && r.create_dctx && r.free_dctx && r.decompress_dctx;
        }
        return r;
    }();
    return z;
}

}  // namespace

extern "C" {

int rmr_zstd_frame_sizes(const uint8_t *const *src, const int64_t *src_len, int64_t n_rows, int64_t *sizes) {
    if (!src || !src_len || !sizes || n_rows < 0) RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    const Zstd &z = zstd();
    if (!z.ok) RMR_FAIL(RMR_ERR_INVALID, "libzstd.so.1 not found");
    for (int64_t i = 0; i < n_rows; ++i) {
        const unsigned long long s = z.content_size(src[i], (size_t)src_len[i]);
        if (s >= (1ull << 62)) RMR_FAIL(RMR_ERR_INVALID, "row %lld: not a zstd frame with a known content size", (long long)i);
        sizes[i] = (int64_t)s;
    }
    return 0;
}

int rmr_zstd_rows(const uint8_t *const *src, const int64_t *src_len, int64_t n_rows, uint8_t *out,
                  const int64_t *out_off, int n_threads) {
    if (!src || !src_len || !out || !out_off || n_rows < 0) RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    const Zstd &z = zstd();
    if (!z.ok) RMR_FAIL(RMR_ERR_INVALID, "libzstd.so.1 not found");
    if (n_rows == 0) return 0;
    std::atomic<int64_t> next{0}, bad{-1};
    auto work = [&] {
        void *ctx = z.create_dctx();  // one context per thread: the one-shot call allocates and frees one per frame
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n_rows) break;
            const size_t cap = (size_t)(out_off[i + 1] - out_off[i]);
            const size_t got = ctx ? z.decompress_dctx(ctx, out + out_off[i], cap, src[i], (size_t)src_len[i])
                                   : z.decompress(out + out_off[i], cap, src[i], (size_t)src_len[i]);
            if (z.is_error(got) || got != cap) bad.store(i);
        }
        if (ctx) z.free_dctx(ctx);
    };
    int nt = n_threads < 1 ? 1 : n_threads;
    if (nt > n_rows) nt = (int)n_rows;
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (bad.load() >= 0) RMR_FAIL(RMR_ERR_INVALID, "corrupt zstd frame in POD5 signal row %lld", (long long)bad.load());
    return 0;
}

}  // extern "C"
