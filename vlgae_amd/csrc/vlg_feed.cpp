// vlg_feed.cpp -- host side of the data feed: length-bucketed token-budget batching and the
// region-feature collate.  Pure host code (no device work): its job is to hand the GPU a batch already in the padded
// layout the kernels read, in pinned memory, so that the only thing left is one DMA.
//
//   vlg_feed_kmeans   -- the bucketing of ConstantTokenNumSampler.kmeans (src/datamodule/sampler.py:148-191): Lloyd iterations
//                        on sentence lengths.  Lengths take a few hundred distinct values, so assignment works on the value
//                        table (O(n + U k) per iteration instead of the reference's dense [n, k] distance matrix); the
//                        per-sentence state the reference's empty-cluster repair needs is kept beside it.
//   vlg_feed_batches  -- ConstantTokenNumSampler._init_iter / _process_batch (sampler.py:86-140): split each bucket's
//                        permutation into its chunks, order the batches by the batch permutation, peel off over-long
//                        sentences, sort each batch by decreasing length (stable).  Output is CSR (offsets + items).
//   vlg_feed_npy_shape / vlg_feed_collate_npy -- _COCODetFeatLazyLoader.__call__ (src/datamodule/task/vlparse.py:36-92):
//                        read each image's [rows, feat_dim + 4] .npy, keep the selected rows, split features | box, and
//                        write them straight into the zero-padded [n, max_len, *] batch buffers (float32) + mask.
//                        Files are read by a small thread pool with pread (no intermediate arrays).
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "vlg_common.h"

namespace {

using vlg::set_error;

// ---------------------------------------------------------------- k-means on lengths
struct ValueTable {
    std::vector<int32_t> value;   // distinct lengths, ascending
    std::vector<int32_t> vid;     // [n] index into `value`
};

ValueTable make_table(const int32_t* x, int64_t n) {
    ValueTable t;
    t.value.assign(x, x + n);
    std::sort(t.value.begin(), t.value.end());
    t.value.erase(std::unique(t.value.begin(), t.value.end()), t.value.end());
    t.vid.resize(n);
    for (int64_t p = 0; p < n; ++p) t.vid[p] = (int32_t)(std::lower_bound(t.value.begin(), t.value.end(), x[p]) - t.value.begin());
    return t;
}

// nearest centroid of every distinct value; ties go to the lowest cluster id (torch.min(-1) on CPU returns the first minimum)
void assign_values(const ValueTable& t, const std::vector<float>& c, std::vector<int32_t>& yv, std::vector<float>& dv) {
    const int k = (int)c.size();
    for (size_t u = 0; u < t.value.size(); ++u) {
        const float xv = (float)t.value[u];
        float best = std::fabs(xv - c[0]);
        int arg = 0;
        for (int j = 1; j < k; ++j) {
            const float d = std::fabs(xv - c[j]);
            if (d < best) { best = d; arg = j; }
        }
        yv[u] = arg;
        dv[u] = best;
    }
}

// ---------------------------------------------------------------- .npy
struct NpyInfo {
    int64_t rows = 0, cols = 0, offset = 0;
    int elem = 0;   // 2, 4, 8 bytes: '<f2', '<f4', '<f8'
};

int parse_npy(int fd, const char* path, NpyInfo* info) {
    unsigned char head[12];
    if (pread(fd, head, 10, 0) != 10 || memcmp(head, "\x93NUMPY", 6) != 0) return set_error(VLG_ERR_ARG, "%s: not a .npy file", path);
    size_t hlen, hoff;
    if (head[6] == 1) { hlen = head[8] | (head[9] << 8); hoff = 10; }
    else {
        if (pread(fd, head, 12, 0) != 12) return set_error(VLG_ERR_ARG, "%s: truncated header", path);
        hlen = head[8] | (head[9] << 8) | (head[10] << 16) | ((size_t)head[11] << 24);
        hoff = 12;
    }
    std::string h(hlen, '\0');
    if ((size_t)pread(fd, &h[0], hlen, hoff) != hlen) return set_error(VLG_ERR_ARG, "%s: truncated header", path);
    const size_t d = h.find("'descr'");
    const size_t q0 = d == std::string::npos ? d : h.find('\'', d + 7);
    const size_t q1 = q0 == std::string::npos ? q0 : h.find('\'', q0 + 1);
    if (q1 == std::string::npos) return set_error(VLG_ERR_ARG, "%s: no descr in header", path);
    const std::string descr = h.substr(q0 + 1, q1 - q0 - 1);
    if (descr == "<f4") info->elem = 4;
    else if (descr == "<f8") info->elem = 8;
    else if (descr == "<f2") info->elem = 2;
    else return set_error(VLG_ERR_DTYPE, "%s: dtype %s not supported (little-endian f2 / f4 / f8)", path, descr.c_str());
    if (h.find("'fortran_order': False") == std::string::npos) return set_error(VLG_ERR_ARG, "%s: fortran_order arrays not supported", path);
    const size_t s = h.find("'shape'");
    const size_t p0 = s == std::string::npos ? s : h.find('(', s);
    if (p0 == std::string::npos) return set_error(VLG_ERR_ARG, "%s: no shape in header", path);
    long long r = 0, c = 0;
    if (sscanf(h.c_str() + p0, "(%lld, %lld", &r, &c) != 2) return set_error(VLG_ERR_SHAPE, "%s: expected a 2-d array", path);
    info->rows = r;
    info->cols = c;
    info->offset = (int64_t)(hoff + hlen);
    return 0;
}

inline float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 31, man = h & 1023;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            uint32_t m = man;
            do { m <<= 1; ++e; } while (!(m & 1024));
            bits = sign | ((uint32_t)(112 - e) << 23) | ((m & 1023) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

void convert_row(const unsigned char* src, int elem, int64_t count, float* dst) {
    if (elem == 4) memcpy(dst, src, (size_t)count * 4);
    else if (elem == 8) {
        for (int64_t i = 0; i < count; ++i) { double v; memcpy(&v, src + 8 * i, 8); dst[i] = (float)v; }
    } else {
        for (int64_t i = 0; i < count; ++i) { uint16_t v; memcpy(&v, src + 2 * i, 2); dst[i] = half_to_float(v); }
    }
}

}  // namespace

extern "C" {

int vlg_feed_kmeans(const int32_t* seq_len, int64_t n, const float* init_centroids, int k, int max_it, float* centroids,
                    int32_t* assign, int* n_clusters) {
    if (!seq_len || !init_centroids || !centroids || !assign || !n_clusters) return set_error(VLG_ERR_ARG, "vlg_feed_kmeans: null argument");
    if (n < 1 || k < 1 || k > n) return set_error(VLG_ERR_ARG, "vlg_feed_kmeans: need 1 <= k <= n (n=%lld, k=%d)", (long long)n, k);
    const ValueTable t = make_table(seq_len, n);
    const size_t U = t.value.size();
    std::vector<float> c(init_centroids, init_centroids + k), old;
    std::vector<int32_t> yv(U), y(n);
    std::vector<float> dv(U);
    std::vector<int64_t> cnt(k), sum(k);
    auto spread = [&] { for (int64_t p = 0; p < n; ++p) y[p] = yv[t.vid[p]]; };
    assign_values(t, c, yv, dv);
    spread();
    for (int it = 0; it < max_it; ++it) {
        // an empty cluster takes the point of the biggest cluster that lies farthest from that cluster's centroid
        // (distances as of the last assignment; first maximum in sentence order) -- sampler.py:164-176
        for (;;) {
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int64_t p = 0; p < n; ++p) ++cnt[y[p]];
            std::vector<int> none;
            for (int j = 0; j < k; ++j) if (cnt[j] == 0) none.push_back(j);
            if (none.empty()) break;
            for (int e : none) {
                const int big = (int)(std::max_element(cnt.begin(), cnt.end()) - cnt.begin());   // first maximum
                int64_t far = -1;
                float fd = -1.f;
                for (int64_t p = 0; p < n; ++p)
                    if (y[p] == big && dv[t.vid[p]] > fd) { fd = dv[t.vid[p]]; far = p; }
                y[far] = e;
                --cnt[big];
                ++cnt[e];
            }
        }
        std::fill(sum.begin(), sum.end(), 0);
        for (int64_t p = 0; p < n; ++p) sum[y[p]] += seq_len[p];
        old = c;
        for (int j = 0; j < k; ++j) c[j] = (float)sum[j] / (float)cnt[j];   // exact float32 sums while a bucket holds < 2^24 tokens
        assign_values(t, c, yv, dv);
        spread();
        if (c == old) break;
    }
    // surviving clusters, renumbered in centroid-id order (sampler.py:184-189)
    std::vector<int> remap(k, -1);
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int64_t p = 0; p < n; ++p) ++cnt[y[p]];
    int m = 0;
    for (int j = 0; j < k; ++j)
        if (cnt[j]) { remap[j] = m; centroids[m] = c[j]; ++m; }
    for (int64_t p = 0; p < n; ++p) assign[p] = remap[y[p]];
    *n_clusters = m;
    return 0;
}

int vlg_feed_batches(const int32_t* seq_len, int64_t n, const int64_t* bucket_offsets, const int64_t* bucket_items, int n_buckets,
                     const int64_t* chunks, const int64_t* bucket_perms, const int64_t* batch_perm, int single_sent_threshold,
                     int sort_in_batch, int64_t* out_offsets, int64_t* out_items, int64_t* n_batches) {
    if (!seq_len || !bucket_offsets || !bucket_items || !chunks || !bucket_perms || !batch_perm || !out_offsets || !out_items || !n_batches)
        return set_error(VLG_ERR_ARG, "vlg_feed_batches: null argument");
    // raw batches: bucket i's permutation cut into chunks[i] pieces of sizes (len - j - 1) / chunks + 1  (sampler.py:97-103)
    std::vector<int64_t> raw_off;   // into the permuted item list
    std::vector<int64_t> permuted(bucket_offsets[n_buckets]);
    raw_off.push_back(0);
    for (int b = 0; b < n_buckets; ++b) {
        const int64_t lo = bucket_offsets[b], len = bucket_offsets[b + 1] - lo, ch = chunks[b];
        if (ch < 1 || ch > len) return set_error(VLG_ERR_ARG, "vlg_feed_batches: bucket %d has %lld items but %lld chunks", b, (long long)len, (long long)ch);
        for (int64_t q = 0; q < len; ++q) {
            const int64_t s = bucket_perms[lo + q];
            if (s < 0 || s >= len) return set_error(VLG_ERR_ARG, "vlg_feed_batches: bucket %d: permutation entry %lld out of range", b, (long long)s);
            const int64_t item = bucket_items[lo + s];
            if (item < 0 || item >= n) return set_error(VLG_ERR_ARG, "vlg_feed_batches: sentence index %lld out of range", (long long)item);
            permuted[lo + q] = item;
        }
        for (int64_t j = 0; j < ch; ++j) raw_off.push_back(raw_off.back() + (len - j - 1) / ch + 1);
    }
    const int64_t n_raw = (int64_t)raw_off.size() - 1;
    int64_t nb = 0, w = 0;
    out_offsets[0] = 0;
    std::vector<int64_t> singles;
    for (int64_t r = 0; r < n_raw; ++r) {
        const int64_t src = batch_perm[r];
        if (src < 0 || src >= n_raw) return set_error(VLG_ERR_ARG, "vlg_feed_batches: batch permutation entry %lld out of range", (long long)src);
        singles.clear();
        const int64_t start = w;
        for (int64_t q = raw_off[src]; q < raw_off[src + 1]; ++q) {
            const int64_t item = permuted[q];
            if (single_sent_threshold != -1 && seq_len[item] >= single_sent_threshold) singles.push_back(item);
            else out_items[w++] = item;
        }
        if (w > start) {
            if (sort_in_batch)
                std::stable_sort(out_items + start, out_items + w, [&](int64_t a, int64_t b) { return seq_len[a] > seq_len[b]; });
            out_offsets[++nb] = w;
        }
        for (int64_t s : singles) {   // each over-long sentence is a batch of its own, after the batch it was drawn with
            out_items[w++] = s;
            out_offsets[++nb] = w;
        }
    }
    *n_batches = nb;
    return 0;
}

int vlg_feed_npy_shape(const char* path, int64_t* rows, int64_t* cols) {
    if (!path || !rows || !cols) return set_error(VLG_ERR_ARG, "vlg_feed_npy_shape: null argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return set_error(VLG_ERR_ARG, "%s: %s", path, strerror(errno));
    NpyInfo info;
    const int rc = parse_npy(fd, path, &info);
    close(fd);
    if (rc) return rc;
    *rows = info.rows;
    *cols = info.cols;
    return 0;
}

int vlg_feed_collate_npy(const char* const* paths, int n, const int32_t* sel, int sel_stride, const int32_t* n_sel, int feat_dim,
                         int box_dim, int max_len, float* feat, float* box, uint8_t* mask, int n_threads) {
    if (!paths || !n_sel || !feat || !box || !mask) return set_error(VLG_ERR_ARG, "vlg_feed_collate_npy: null argument");
    if (n < 0 || feat_dim < 1 || box_dim < 0 || max_len < 0) return set_error(VLG_ERR_SHAPE, "vlg_feed_collate_npy: bad sizes");
    for (int i = 0; i < n; ++i)
        if (n_sel[i] < 0 || n_sel[i] > max_len) return set_error(VLG_ERR_SHAPE, "vlg_feed_collate_npy: image %d keeps %d rows, batch pads to %d", i, n_sel[i], max_len);
    std::atomic<int> next(0), failed(0);
    std::vector<std::string> errs(n_threads > 1 ? n_threads : 1);
    std::vector<int> codes(errs.size(), 0);
    auto work = [&](int tix) {
        std::vector<unsigned char> row;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || failed.load()) return;
            const int fd = open(paths[i], O_RDONLY);
            int rc = fd < 0 ? set_error(VLG_ERR_ARG, "%s: %s", paths[i], strerror(errno)) : 0;
            NpyInfo info;
            if (!rc) rc = parse_npy(fd, paths[i], &info);
            if (!rc && info.cols != feat_dim + box_dim)
                rc = set_error(VLG_ERR_SHAPE, "%s: %lld columns, expected %d + %d", paths[i], (long long)info.cols, feat_dim, box_dim);
            float* f = feat + (size_t)i * max_len * feat_dim;
            float* b = box + (size_t)i * max_len * box_dim;
            uint8_t* m = mask + (size_t)i * max_len;
            const size_t rb = (size_t)info.cols * info.elem;
            int64_t need = 0;   // rows to fetch: one pread covers every selected row
            for (int k = 0; !rc && k < n_sel[i]; ++k) {
                const int64_t r = sel ? sel[(size_t)i * sel_stride + k] : k;   // no selection = the leading rows
                if (r < 0 || r >= info.rows) rc = set_error(VLG_ERR_ARG, "%s: row %lld of %lld", paths[i], (long long)r, (long long)info.rows);
                need = std::max(need, r + 1);
            }
            if (!rc) {
                row.resize((size_t)need * rb);
                size_t got = 0;
                while (got < row.size()) {
                    const ssize_t g = pread(fd, row.data() + got, row.size() - got, info.offset + (int64_t)got);
                    if (g <= 0) { rc = set_error(VLG_ERR_ARG, "%s: short read", paths[i]); break; }
                    got += (size_t)g;
                }
            }
            int k = 0;
            for (; !rc && k < n_sel[i]; ++k) {
                const unsigned char* src = row.data() + (size_t)(sel ? sel[(size_t)i * sel_stride + k] : k) * rb;
                convert_row(src, info.elem, feat_dim, f + (size_t)k * feat_dim);
                if (box_dim) convert_row(src + (size_t)feat_dim * info.elem, info.elem, box_dim, b + (size_t)k * box_dim);
                m[k] = 1;
            }
            if (!rc) {   // padding rows
                memset(f + (size_t)k * feat_dim, 0, (size_t)(max_len - k) * feat_dim * 4);
                if (box_dim) memset(b + (size_t)k * box_dim, 0, (size_t)(max_len - k) * box_dim * 4);
                memset(m + k, 0, (size_t)(max_len - k));
            }
            if (fd >= 0) close(fd);
            if (rc) {
                codes[tix] = rc;
                errs[tix] = vlg_last_error();
                failed.store(1);
                return;
            }
        }
    };
    if (n_threads <= 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int tix = 0; tix < n_threads; ++tix) pool.emplace_back(work, tix);
        for (auto& th : pool) th.join();
    }
    for (size_t tix = 0; tix < codes.size(); ++tix)
        if (codes[tix]) return set_error(codes[tix], "%s", errs[tix].c_str());
    return 0;
}

}  // extern "C"
